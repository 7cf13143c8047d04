#!/usr/bin/env python3
"""BASELINE config 4 in small: NAFS-style adaptive k-hop smoothing with the adjacency ROW-SHARDED across the GPUs of a node.

Every rank generates (stands in for: loads) only ITS rows of a directed papers100M-shaped graph and ITS rows of the feature
matrix, normalises its block (the degree vector is the one all-reduce), propagates k hops -- between hops a rank receives only the
rows its block gathers (need-aware exchange, sgl_amd/dist/halo.py) -- and weights the hops of its own nodes with the
over-smoothing distance (OverSmoothDistanceWeightedOp, sgl/operators/message_op/over_smooth_distance_op.py).  No rank ever holds
the whole adjacency or the whole feature matrix.

    python examples/nafs_row_sharded.py                                               # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/nafs_row_sharded.py --nodes 20000000
"""
import argparse
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgl_amd import synthetic as sy  # noqa: E402
from sgl_amd.dist import RowBlock, ShardedGraphOp, canonicalize_block  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=2_000_000)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--hops", type=int, default=5)
    ap.add_argument("--mean-deg", type=float, default=30.0)
    ap.add_argument("--r", type=float, default=0.5)
    a = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    n = a.nodes
    lo, hi = n * rank // world, n * (rank + 1) // world
    t0 = time.perf_counter()
    table = sy.degree_table(a.mean_deg, 20_000)
    rp, c, v = sy.hashed_block_torch(0, lo, hi - lo, n, table, device=device)             # my rows of T = A^T (directed)
    block = canonicalize_block(RowBlock(lo, hi, n, rp, c, v))
    x_own = sy.hashed_features_torch(0, lo, hi - lo, a.feat, device=device)               # my rows of X
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0

    op = ShardedGraphOp(a.hops, r=a.r, symmetric=False, col_chunks=2, reorder=None)
    t0 = time.perf_counter()
    hops = op.propagate(block, x_own)                                                       # K + 1 shards [hi - lo, d]
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    del hops                                                                                # (or the allocator must find new room)
    t0 = time.perf_counter()
    hops = op.propagate(block, x_own)                                                       # normalised block and plan cached
    torch.cuda.synchronize()
    t_prop = time.perf_counter() - t0
    t0 = time.perf_counter()
    smoothed = op.over_smooth_aggregate(hops)
    torch.cuda.synchronize()
    t_nafs = time.perf_counter() - t0
    t_again = t_prop + t_nafs
    # the adaptive-k-hop SWEEP of the NAFS tasks (every hop count 0 .. k evaluated, tasks/node_clustering.py:176-178) from the SAME
    # propagation: every prefix of the hop list in one pass over this rank's shards, no communication
    t0 = time.perf_counter()
    sweep = op.over_smooth_sweep(hops, a.hops + 1)
    torch.cuda.synchronize()
    t_sweep = time.perf_counter() - t0
    assert sorted(sweep) == list(range(a.hops + 1)) and torch.allclose(sweep[a.hops], smoothed, rtol=1e-5, atol=1e-6)

    nnz = torch.tensor([op.a_hat_block.nnz], dtype=torch.int64, device=device)
    check = smoothed.double().sum().reshape(1)
    if world > 1:
        dist.all_reduce(nnz)
        dist.all_reduce(check)
    if rank == 0:
        plan = getattr(op, "halo_plan", None)
        what = (f"need-aware exchange: rank 0 receives {plan.n_ghost} of {plan.rows_in_full} foreign rows per hop "
                f"({plan.skipped_fraction:.1%} never travel)") if plan is not None and world > 1 else "single rank: no exchange"
        print(f"NAFS row-sharded x{world}: N={n} nnz(A_hat)={int(nnz)} d={a.feat} k={a.hops}; {what}")
        print(f"  load own rows {t_load:.2f} s, first propagate (normalise block + plan + {a.hops} hops) {t_first:.2f} s, "
              f"cached propagate {t_prop * 1e3:.1f} ms + over-smoothing weights {t_nafs * 1e3:.1f} ms "
              f"= {int(nnz) * a.feat * a.hops / t_again / 1e12:.3f}e12 edge*feat/s; checksum {float(check):.6e}")
        print(f"  hop sweep: the smoothed features of all {a.hops + 1} hop counts from the same propagation in {t_sweep * 1e3:.1f} ms "
              f"(the reference re-propagates per hop count: {a.hops * (a.hops + 1) // 2} hops instead of {a.hops})")
        assert np.isfinite(float(check))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
