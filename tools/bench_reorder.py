#!/usr/bin/env python3
"""Plan-time locality ordering (sgl_amd/reorder.py, GraphOp(reorder="community")): can it recover the locality of a graph whose
node ids carry none?
A products-sized graph with 80 % of its edges inside communities of `bs` nodes, ids SHUFFLED (what a real dump looks like
after any relabelling), is (a) propagated as it is, (b) after sorting the nodes by a community label found with a few rounds
of label propagation on the device, (c) with the generator's own community order (the upper bound)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import device as dev, synthetic  # noqa: E402
from sgl_amd.reorder import community_order, permute_csr  # noqa: E402



def time_ms(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def community_graph(n, m, bs, device, seed=7):
    g = torch.Generator(device=device).manual_seed(seed)
    w = torch.exp(torch.randn(n, generator=g, device=device, dtype=torch.float64) * 1.2)
    cdf = torch.cumsum(w, 0); cdf /= cdf[-1].clone()
    a_ = torch.searchsorted(cdf, torch.rand(m, generator=g, device=device, dtype=torch.float64)).clamp_(0, n - 1)
    local = torch.rand(m, generator=g, device=device) < 0.8
    b_far = torch.searchsorted(cdf, torch.rand(m, generator=g, device=device, dtype=torch.float64)).clamp_(0, n - 1)
    b_near = ((a_ // bs) * bs + torch.randint(0, bs, (m,), generator=g, device=device)).clamp_(0, n - 1)
    b_ = torch.where(local, b_near, b_far)
    keep = a_ != b_
    lo_, hi_ = torch.minimum(a_, b_)[keep], torch.maximum(a_, b_)[keep]
    keys = torch.unique(lo_ * n + hi_)
    full = torch.sort(torch.cat([keys, (keys % n) * n + keys // n])).values
    rp = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rp[1:] = torch.cumsum(torch.bincount(full // n, minlength=n), 0)
    return rp, (full % n).to(torch.int32), torch.ones(full.numel(), device=device)


def main():
    device = torch.device("cuda", 0)
    wl = synthetic.WORKLOADS[os.environ.get("REORDER_WORKLOAD", "S1_products")]
    n, d = wl["n"], wl["d"]
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    y = torch.empty_like(x0)
    for bs in [int(b) for b in os.environ.get("REORDER_BLOCKS", "2048,16384").split(",")]:
        rp, cc, vv = community_graph(n, wl["m"], bs, device)
        g = torch.Generator(device=device).manual_seed(11)
        shuffle = torch.randperm(n, generator=g, device=device)            # new id of old node i
        rp_s, cc_s, vv_s = permute_csr(rp, cc, vv, shuffle)
        variants = [("generator order (upper bound)", rp, cc, vv), ("shuffled ids", rp_s, cc_s, vv_s)]
        t0 = time.perf_counter()
        order, info = community_order(rp_s, cc_s, n, rounds=int(os.environ.get("REORDER_ROUNDS", 8)))
        torch.cuda.synchronize()
        t_order = time.perf_counter() - t0
        t0 = time.perf_counter()
        rp_r, cc_r, vv_r = permute_csr(rp_s, cc_s, vv_s, order)
        torch.cuda.synchronize()
        t_perm = time.perf_counter() - t0
        variants.append((f"shuffled ids + label-propagation order ({info})", rp_r, cc_r, vv_r))
        for name, a, b, c in variants:
            rpn, ccn, vvn = dev.normalize_adj(a, b, c, n, 0.5, None)
            csr = dev.DeviceCSR(rpn, ccn, vvn, (n, n))
            ms = time_ms(lambda: csr.spmm(x0, out=y))
            nz = ccn.numel()
            alg = nz * d * 4 + nz * 8 + (n + 1) * 4 + n * d * 4
            print(f"EXP reorder community={bs} {name}: ms_per_hop={ms:.3f} frac={alg / (ms * 1e-3) / 8e12:.3f}", flush=True)
            if name == "shuffled ids":
                # what GraphOp(reorder="community") does: the same matrix, ids untouched, rows STORED and PROCESSED in the
                # label-propagation order (sgl_csr_permute_rows + sgl_csr_set_rowmap) -- bit-identical results
                y_plain = y.clone()
                rowmap = torch.argsort(order).to(torch.int32)
                rp2, c2, v2 = dev.permute_rows(rpn, ccn, vvn, rowmap)
                cm = dev.DeviceCSR(rp2, c2, v2, (n, n)).set_rowmap(rowmap)
                ms = time_ms(lambda: cm.spmm(x0, out=y))
                print(f"EXP reorder community={bs} shuffled ids, rows processed in label-propagation order (row map, ids untouched): "
                      f"ms_per_hop={ms:.3f} frac={alg / (ms * 1e-3) / 8e12:.3f} bit_identical={bool(torch.equal(y, y_plain))}", flush=True)
                del cm, rp2, c2, v2, y_plain
            del csr, rpn, ccn, vvn
        print(f"EXP reorder community={bs} plan-time cost: ordering {t_order * 1e3:.0f} ms + permuting the CSR {t_perm * 1e3:.0f} ms", flush=True)
        # end to end through the operator (features permuted in, every hop permuted out), k = 3, adjacency cached
        from sgl_amd.io import DeviceAdjacency
        from sgl_amd.operators.graph_op import LaplacianGraphOp
        dadj = DeviceAdjacency(rp_s, cc_s, vv_s, (n, n))
        for label, op in (("plain", LaplacianGraphOp(3, r=0.5)), ("reorder=community", LaplacianGraphOp(3, r=0.5, reorder="community"))):
            ms = time_ms(lambda: op.propagate(dadj, x0), reps=3, warm=1)
            print(f"EXP reorder community={bs} GraphOp.propagate k=3 on the shuffled graph, {label}: ms={ms:.2f}", flush=True)
        del dadj
    # the benchmark graph itself has no communities: the ordering must not make it slower
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    order, info = community_order(a_ptr, a_col, n, rounds=8)
    for name, (a, b, c) in (("as generated", (a_ptr, a_col, a_val)), (f"label-propagation order ({info})", permute_csr(a_ptr, a_col, a_val, order))):
        rpn, ccn, vvn = dev.normalize_adj(a, b, c, n, 0.5, None)
        csr = dev.DeviceCSR(rpn, ccn, vvn, (n, n))
        ms = time_ms(lambda: csr.spmm(x0, out=y))
        print(f"EXP reorder S1 benchmark graph (no communities) {name}: ms_per_hop={ms:.3f}", flush=True)
        del csr


if __name__ == "__main__":
    main()
