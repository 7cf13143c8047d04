"""Row-sharded storage helpers (sgl_amd/dist/sharded_adj.py) under gloo on CPU tensors: scatter of row blocks, piece-bound
tables, re-assembly, feature all-gather, exact exchange checksums, device-side balanced bounds."""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph(n=500, seed=3):
    rng = np.random.default_rng(seed)
    deg = np.minimum(rng.lognormal(1.0, 1.0, n).astype(np.int64), 60)
    deg[rng.integers(0, n, 25)] = 0
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    col = np.concatenate([np.sort(rng.choice(n, int(k), replace=False)) for k in deg]).astype(np.int32)
    val = rng.uniform(-1, 1, len(col)).astype(np.float32)
    return rowptr, col, val


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sgl_amd.dist import (allgather_blocks, allgather_rows, balanced_bounds, balanced_bounds_device, exchange_checksums,
                              gather_piece_bounds, local_piece_bounds, scatter_row_blocks)
    rowptr, col, val = _graph()
    n = len(rowptr) - 1
    bounds = balanced_bounds(rowptr, world)
    assert np.array_equal(bounds, balanced_bounds_device(torch.from_numpy(rowptr), world))
    full = tuple(torch.from_numpy(a) for a in (rowptr, col, val)) if rank == 0 else None
    blk = scatter_row_blocks(full, bounds, n, torch.device("cpu"))
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    ok = blk.lo == lo and blk.hi == hi and blk.n == n
    ok = ok and np.array_equal(blk.rowptr.numpy(), rowptr[lo:hi + 1] - rowptr[lo])
    ok = ok and np.array_equal(blk.col.numpy(), col[rowptr[lo]:rowptr[hi]]) and np.array_equal(blk.val.numpy(), val[rowptr[lo]:rowptr[hi]])
    mine, _ = local_piece_bounds(blk, 3)
    pb = gather_piece_bounds(mine)
    ok = ok and pb.shape == (world, 4) and pb[0, 0] == 0 and pb[-1, -1] == n and (pb[:, 0] == bounds[:-1]).all()
    rp2, c2, v2 = allgather_blocks(blk)
    ok = ok and np.array_equal(rp2.numpy(), rowptr) and np.array_equal(c2.numpy(), col) and np.array_equal(v2.numpy(), val)
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((n, 7)).astype(np.float32))
    rep = allgather_rows(x[lo:hi].clone(), bounds, n)
    ok = ok and torch.equal(rep, x)
    ok = ok and exchange_checksums(rep, x[lo:hi], bounds)
    bad = rep.clone()
    other = (rank + 1) % world
    if bounds[other + 1] > bounds[other]:
        bad[int(bounds[other]), 0] += 1.0                       # one corrupted element in a peer's range
        ok = ok and not exchange_checksums(bad, x[lo:hi], bounds)
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump({"ok": bool(ok)}, f)
    dist.destroy_process_group()


def test_row_block_helpers_gloo(tmp_path):
    for world in (2, 3):
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
        assert all(json.load(open(tmp_path / f"r{r}.json"))["ok"] for r in range(world)), world


def test_hashed_generator_host_mirror_properties():
    """the host mirror of sgl_synth_* (the device generator is checked against it bit for bit in the gpu suite)"""
    from sgl_amd import synthetic as sy
    table = sy.degree_table(30.07, 20_000)
    assert table.shape == (8192,) and table.min() >= 1 and table.max() <= 20_000
    assert (np.diff(table[:4096]) >= 0).all() and (np.diff(table[4096:]) >= 0).all()
    n = 1_000_003
    rows = np.array([0, 1, 17, n - 1, 123_456])
    ip, col, val = sy.hashed_rows_numpy(5, rows, n, table)
    assert ip[0] == 0 and np.array_equal(np.diff(ip), sy.hashed_degrees_numpy(5, rows, table))
    assert col.min() >= 0 and col.max() < n and val.min() >= 0 and val.max() < 1 / 32
    # keyed by (seed, row): a row does not depend on which other rows are generated with it
    ip2, col2, val2 = sy.hashed_rows_numpy(5, rows[2:3], n, table)
    assert np.array_equal(col2, col[ip[2]:ip[3]]) and np.array_equal(val2, val[ip[2]:ip[3]])
    assert not np.array_equal(sy.hashed_rows_numpy(6, rows, n, table)[1][:50], col[:50])
    d = sy.hashed_degrees_numpy(5, np.arange(400_000), table)
    assert abs(d.mean() - 30.07) < 0.6                          # the law's mean
    x = sy.hashed_features_numpy(5, rows, 16)
    assert x.shape == (5, 16) and x.dtype == np.float32 and np.abs(x).max() < 1
    perm = sy._permute_id(np.arange(5000), 5000, 9)
    assert np.array_equal(np.sort(perm), np.arange(5000))
    # hub skew: the square of a uniform concentrates on small pre-permutation ids
    u = sy._hash4(1, 1, np.arange(200_000), 0)
    skew = sy._mulhi64(sy._mulhi64(u, u), np.uint64(n)).astype(np.int64)
    assert 0.08 < (skew < n // 100).mean() < 0.12                # sqrt(1 %) = 10 %
