#!/usr/bin/env python3
"""PaSca operator sweep at the ogbn-papers100M shape (BASELINE config 5; SURVEY 8(d) workload S4), as ONE rank of the 8-GPU
row-sharded job sees it, on one GPU: graph ops {Laplacian r=0.5, PPR alpha in {0.1, 0.2, 0.3}} x k = 10 x every MessageOp
(sgl/search/search_models.py:19-46, search_config.py:14-15).

The hashed directed graph is canonicalised and normalised per row block exactly as the job would do it
(sgl_coo_to_csr -> sgl_norm_block_*, symmetric=False): the global degree vector is the sum of the 8 blocks' column sums
(here the 8 blocks are generated one after the other on the same GPU, in the job it is one all-reduce).  Rank 0's block
then runs 10 hops against the full 111 M x 128 replica (the same replica every hop: a single GPU cannot produce the other
ranks' rows of the next one -- per-hop time does not depend on the values) and every aggregator runs over the 11 hop
shards [13.9 M, 128].  The per-hop all-gather (49.8 GB in-bound per rank) is not part of this single-GPU measurement."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import _lib, device as dev, synthetic as sy  # noqa: E402
from sgl_amd.dist import RowBlock, canonicalize_block  # noqa: E402
from sgl_amd.operators import message_op as M  # noqa: E402


def timed(fn, reps=3):
    r = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        r = None                 # outputs here are tens of GB: never hold two of them
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), r


def main():
    device = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    scale = 1 if free > 200e9 else 8
    wl = sy.WORKLOADS["S3_papers"]
    n, d, K, parts = wl["n"] // scale, wl["d"], 10, 8
    table = sy.degree_table(wl["mean_deg"], wl["d_max"])
    bounds = [n * i // parts for i in range(parts + 1)]
    # global degree vector of A + I (A = T^T): column sums of every rank's canonical block + its diagonal
    deg = torch.zeros(n, dtype=torch.float64, device=device)
    block0 = None
    import time
    t0 = time.time()
    for b in range(parts):
        lo, hi = bounds[b], bounds[b + 1]
        rp, c, v = sy.hashed_block_torch(0, lo, hi - lo, n, table, device=device)
        blk = canonicalize_block(RowBlock(lo, hi, n, rp, c, v))
        del rp, c, v
        m = ctypes.c_int64(0)
        _lib.check(_lib.lib().sgl_norm_block_prepare(hi - lo, lo, blk.nnz, _lib.ptr(blk.rowptr), _lib.ptr(blk.col), ctypes.byref(m),
                                                     _lib.current_stream_ptr()))
        o_ptr = torch.empty(hi - lo + 1, dtype=torch.int64, device=device)
        o_col = torch.empty(m.value, dtype=torch.int32, device=device)
        t64 = torch.empty(m.value, dtype=torch.float64, device=device)
        rs = torch.empty(hi - lo, dtype=torch.float64, device=device)
        _lib.check(_lib.lib().sgl_norm_block_build(hi - lo, lo, blk.nnz, _lib.ptr(blk.rowptr), _lib.ptr(blk.col), _lib.ptr(blk.val), m.value,
                                                   _lib.ptr(o_ptr), _lib.ptr(o_col), _lib.ptr(t64), _lib.ptr(rs), _lib.current_stream_ptr()))
        _lib.check(_lib.lib().sgl_norm_block_colsum(n, m.value, _lib.ptr(o_col), _lib.ptr(t64), _lib.ptr(deg), _lib.current_stream_ptr()))
        torch.cuda.synchronize()
        if b == 0:
            block0 = blk
        del o_ptr, o_col, t64, rs
        if b:
            del blk
    print(f"S4 setup: {parts} blocks generated + canonicalised + column sums in {time.time() - t0:.1f} s; block 0: rows={block0.n_local} "
          f"nnz={block0.nnz} (canonical) of n={n}", flush=True)
    x = sy.hashed_features_torch(0, 0, n, d, device=device)
    lo, hi = bounds[0], bounds[1]
    hops = [x[lo:hi]] + [dev.alloc_rows(hi - lo, d, device) for _ in range(K)]
    # the (r, alpha)-independent part of the block, once per graph (what ShardedGraphOp keeps on the RowBlock): T + I in fp64, degrees
    torch.cuda.synchronize()
    t0 = time.time()
    prep = dev.PreparedBlock(block0.rowptr, block0.col, block0.val, lo, n, symmetric=False, deg=deg)
    torch.cuda.synchronize()
    print(f"S4 prepare_block_ms={(time.time() - t0) * 1e3:8.1f}  (once per graph: T + I fp64 values, row sums; {prep.nnz_out} nnz)", flush=True)
    for name, r, alpha in (("laplacian r=0.5", 0.5, None), ("ppr a=0.1", 0.5, 0.1), ("ppr a=0.2", 0.5, 0.2), ("ppr a=0.3", 0.5, 0.3),
                           ("laplacian r=0.3", 0.3, None)):
        for route in ((True, "auto") if name in ("laplacian r=0.5", "laplacian r=0.3") else ("auto",)):
            dev.clear_power_cache() if route is True else None
            hits = dev.pow_stats["cache_hits"]
            torch.cuda.synchronize()
            t0 = time.time()
            rowptr, col, val = prep.normalize(r, alpha, host_pow=route)
            torch.cuda.synchronize()
            t_norm = (time.time() - t0) * 1e3
            # algorithmic bytes of the pass that ran: gather pass 4 + 8 + 8 + 4 per nnz (+ 8 when the fp64 Laplacian is kept), mix 4 + 8 + 4
            print(f"S4 graph_op={name:16s} host_pow={str(route):5s} normalise_block_ms={t_norm:8.2f}  (degree-power cache hit: "
                  f"{dev.pow_stats['cache_hits'] > hits}; stats {dev.pow_stats})", flush=True)
        if name == "laplacian r=0.3":
            break
        csr = dev.DeviceCSR(rowptr, col, val, (hi - lo, n))

        def prop():
            for h in range(1, K + 1):
                csr.spmm(x, out=hops[h])
        t_prop, _ = timed(prop, reps=2)
        nnz = col.numel()
        alg = nnz * d * 4 + nnz * 8 + (hi - lo + 1) * 4 + (hi - lo) * d * 4
        print(f"S4 graph_op={name:16s} normalise_block_ms={t_norm:8.2f} propagate_k10_ms={t_prop:8.1f} per_hop_ms={t_prop / K:7.2f} "
              f"({nnz * d * K / (t_prop * 1e-3) / 1e12:.3f}e12 edge*feat/s per GPU, roofline frac {alg / (t_prop / K * 1e-3) / 8e12:.3f})", flush=True)
        if alpha is None:
            # every PPR alpha of the sweep WITHOUT propagating again: a triangular mix of this chain's 11 hop shards (row-wise: no
            # communication in the row-sharded job either) -- 11 streams read, 10 written per alpha
            from sgl_amd.operators.graph_op import ppr_hops_from_laplacian
            for a_ in (0.1, 0.2, 0.3):
                t_mix, mixed = timed(lambda: ppr_hops_from_laplacian(hops, a_), reps=1)
                by = (2 * K + 1) * (hi - lo) * d * 4
                print(f"S4   ppr a={a_}: the k = 10 hop shards mixed from the Laplacian chain in {t_mix:7.2f} ms ({by / (t_mix * 1e-3) / 1e12:.2f} TB/s) "
                      f"instead of a propagation ({t_prop:.1f} ms + the exchange of 10 hops)", flush=True)
                del mixed
        if alpha not in (None, 0.1):
            del csr
            continue
        ops = [("last", M.LastMessageOp()), ("concat", M.ConcatMessageOp(0, K + 1)), ("mean", M.MeanMessageOp(0, K + 1)),
               ("sum", M.SumMessageOp(0, K + 1)), ("max", M.MaxMessageOp(0, K + 1)), ("min", M.MinMessageOp(0, K + 1)),
               ("simple_weighted a=.85", M.SimpleWeightedMessageOp(0, K + 1, "alpha", 0.85)),
               ("learnable simple", M.LearnableWeightedMessageOp(0, K + 1, "simple", K).to(device)),
               ("learnable gate", M.LearnableWeightedMessageOp(0, K + 1, "gate", d).to(device)),
               ("learnable ori_ref", M.LearnableWeightedMessageOp(0, K + 1, "ori_ref", d).to(device)),
               ("learnable jk", M.LearnableWeightedMessageOp(0, K + 1, "jk", K, d).to(device)),
               ("iterate recursive", M.IterateLearnableWeightedMessageOp(0, K + 1, "recursive", d).to(device)),
               ("nafs over_smooth", M.OverSmoothDistanceWeightedOp())]
        hb = (K + 1) * (hi - lo) * d * 4
        for oname, op in ops:
            with torch.no_grad():
                t, out = timed(lambda: op.aggregate(hops), reps=2)
            by = hb + out.numel() * 4 if oname != "last" else 0
            print(f"S4   msg_op={oname:24s} aggregate_ms={t:9.2f}" + (f"  ({by / (t * 1e-3) / 1e12:.2f} TB/s)" if by else ""), flush=True)
            del out
        del csr


if __name__ == "__main__":
    main()
