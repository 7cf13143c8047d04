#!/usr/bin/env python3
"""Why does a rank's SpMM of the 8-rank job cost 1.21 ms when an eighth of the single-GPU launch would be 1.09 ms?

profiles/r03_scale_model.md: the per-launch excess is roughly constant (0.07 / 0.12 / 0.13 ms at G = 2 / 4 / 8), i.e. it is a
property of a LAUNCH, not of the work: a rank's block is ~29 000 work items for 8 192 resident wavefronts -- three and a half
"rounds" -- so how the launch ends (a straggler item started late, the last partial round) weighs ten per cent there and one
per cent on the single-GPU launch.  This probe times rank 0's block of the G-rank job (compact table, relabelled columns:
exactly what HaloPropagator multiplies) under different execution plans of the SAME matrix:

    item_nnz      target non-zeros per work item (512 default): smaller items = more rounds, finer tail
    long_row_nnz  rows above it are cut into pieces scheduled first (2048 default; CHANGES the summation order of those rows)
    order         heavy_first (the default since this probe; tuning key spmm_heavy_first) = the items that hold >= 2 x item_nnz
                  non-zeros are issued first in every XCD range; row_order = the plan before (results unchanged either way --
                  items are whole rows).  The default item size is 256 below 1e8 non-zeros per launch since this probe, 512 above.

    python tools/probe_small_launch.py  > profiles/r03_probe_small_launch.log
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import _lib  # noqa: E402
from sgl_amd import device as dev  # noqa: E402
from sgl_amd import synthetic  # noqa: E402
from sgl_amd.dist import HaloPlan, balanced_bounds  # noqa: E402


def timed(fn, reps=9, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = synthetic.WORKLOADS["S1_products"]
    n, d = wl["n"], wl["d"]
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    del a_ptr, a_col, a_val
    rp_host = rowptr.cpu().numpy()
    x0 = synthetic.features_torch(n, d, seed=0, device=device)

    def make(rp_, c_, v_, shape, heavy_first=True, **kw):
        _lib.set_tuning("spmm_heavy_first", 1 if heavy_first else 0)      # read when the plan is built
        try:
            return dev.DeviceCSR(rp_, c_, v_, shape, **kw)
        finally:
            _lib.set_tuning("spmm_heavy_first", 1)

    print(f"# tools/probe_small_launch.py: S1 (N = {n}, nnz = {int(rp_host[-1])}, d = {d}); median / min of 9 launches, ms")
    whole = make(rowptr, col, val, (n, n), heavy_first=False, item_nnz=512)
    y = torch.empty((n, d), device=device)
    t_whole, t_whole_min = timed(lambda: whole.spmm(x0, out=y))
    print(f"SMALL G=1 whole graph, row_order item_nnz=512: {t_whole:.3f} / {t_whole_min:.3f} ms, items {whole.info()['n_items']}, pieces {whole.info()['n_pieces']}")
    for name, kw in (("heavy_first item_nnz=512 (default)", {}), ("heavy_first item_nnz=384", {"item_nnz": 384}),
                     ("heavy_first item_nnz=256", {"item_nnz": 256}), ("row_order item_nnz=256", {"item_nnz": 256, "heavy_first": False})):
        other = make(rowptr, col, val, (n, n), **kw)
        y2 = torch.empty((n, d), device=device)
        t_o, t_o_min = timed(lambda: other.spmm(x0, out=y2))
        print(f"SMALL G=1 {name:36s} items {other.info()['n_items']:6d} | w=100: {t_o:.3f} / {t_o_min:.3f} | "
              f"{'bit-identical to the first row' if torch.equal(y, y2) else 'DIFFERENT'}", flush=True)
        del other, y2
    del whole, y
    for G in (8, 4, 2):
        bounds = balanced_bounds(rp_host, G)
        r = 0
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        nb, ne = int(rp_host[lo]), int(rp_host[hi])
        plan = HaloPlan.offline(r, bounds, n, lambda q: col[int(rp_host[bounds[q]]):int(rp_host[bounds[q + 1]])])
        rp_local = (rowptr[lo:hi + 1] - rowptr[lo]).contiguous()
        ccol = plan.relabel(col[nb:ne])
        cval = val[nb:ne].contiguous()
        t0 = x0.index_select(0, plan.global_ids)
        ideal = t_whole * (ne - nb) / int(rp_host[-1])
        print(f"SMALL G={G} rank 0: rows {hi - lo}, nnz {ne - nb}; its share of the single-GPU launch would be {ideal:.3f} ms")
        ref = None
        ro = {"heavy_first": False}
        variants = [("row_order item_nnz=512 (before)", dict(ro, item_nnz=512)), ("row_order item_nnz=384", dict(ro, item_nnz=384)),
                    ("row_order item_nnz=256", dict(ro, item_nnz=256)), ("row_order item_nnz=128", dict(ro, item_nnz=128)),
                    ("row_order 512 long_row_nnz=1024", dict(ro, item_nnz=512, long_row_nnz=1024)),
                    ("row_order 512 long_row_nnz=512", dict(ro, item_nnz=512, long_row_nnz=512)),
                    ("row_order 256 long_row_nnz=512", dict(ro, item_nnz=256, long_row_nnz=512)),
                    ("heavy_first item_nnz=512", {"item_nnz": 512}), ("heavy_first item_nnz=384", {"item_nnz": 384}),
                    ("heavy_first item_nnz=256", {"item_nnz": 256}), ("heavy_first item_nnz=192", {"item_nnz": 192}),
                    ("library default (now)", {})]
        for name, kw in variants:
            csr = make(rp_local, ccol, cval, (hi - lo, plan.n_compact), **kw)
            line = f"SMALL G={G} {name:36s} items {csr.info()['n_items']:6d} pieces {csr.info()['n_pieces']:4d}"
            for a, b in ((0, d), (0, 64), (64, d)):
                t = t0 if (a, b) == (0, d) else t0[:, a:b].contiguous()
                yy = torch.empty((hi - lo, b - a), device=device)
                med, mn = timed(lambda: csr.spmm(t, out=yy))
                line += f" | w={b - a:3d}: {med:.3f} / {mn:.3f}"
                if (a, b) == (0, d):
                    if ref is None:
                        ref = yy.clone()
                    same = bool(torch.equal(ref, yy))
                    line_same = "bit-identical to the first row" if same else f"max rel diff to the first row {float(((ref - yy).abs().max() / ref.abs().max())):.2e}"
                del yy
            print(line + " | " + line_same, flush=True)
            del csr
        del t0, plan


if __name__ == "__main__":
    main()
