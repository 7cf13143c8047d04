#!/usr/bin/env python3
"""Experiment: split feature layout for d = 100 -- 96 line-aligned main columns + a packed [N, 4]
tail table gathered by the idle lanes, against the plain pitch-100 layout.  Prints `EXP tail ...` lines.

    python tools/sweep_tail.py [--workload S1_products] [--hot]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sgl_amd import _lib, synthetic  # noqa: E402
from sgl_amd import device as dev  # noqa: E402


def time_ms(fn, reps=7, warm=2):
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
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S1_products")
    ap.add_argument("--hot", action="store_true", help="also: columns relabelled hottest-first (tail table hub-dense)")
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = synthetic.WORKLOADS[a.workload]
    n, d = wl["n"], wl["d"]
    dm = d // 32 * 32
    tw = (d - dm + 3) // 4 * 4
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    del a_ptr, a_col, a_val
    nnz = col.numel()
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    alg = nnz * d * 4 + nnz * 8 + (n + 1) * 4 + n * d * 4

    def line(name, ms, **kv):
        print(f"EXP tail {name} " + " ".join(f"{k}={v}" for k, v in kv.items()) +
              f" ms_per_hop={ms:.3f} frac={alg / (ms * 1e-3) / 8e12:.3f}", flush=True)

    def run_set(tag, rowptr, col, val, x0):
        csr = dev.DeviceCSR(rowptr, col, val, (n, n))
        y_ref = torch.empty_like(x0)
        line(f"{tag}base_pitch{d}", time_ms(lambda: csr.spmm(x0, out=y_ref)))
        # lower bound: the main block alone (3 lines per gathered row)
        xm = x0[:, :dm].contiguous()
        ym = torch.empty_like(xm)
        line(f"{tag}main_only_d{dm}", time_ms(lambda: csr.spmm(xm, out=ym)))
        xt = torch.zeros((n, tw), device=device)
        xt[:, :d - dm] = x0[:, dm:]
        yt = torch.empty_like(xt)
        for unroll in (2, 4):
            for nt in (0, 1):
                _lib.set_tuning("spmm_unroll", unroll)
                _lib.set_tuning("spmm_tail_nt", nt)
                ms = time_ms(lambda: csr.spmm_tail(xm, xt, ym, yt, d, dm))
                ok = bool(torch.equal(ym, y_ref[:, :dm]) and torch.equal(yt[:, :d - dm], y_ref[:, dm:]))
                line(f"{tag}split_pitch{dm}", ms, unroll=unroll, nt_main=nt, bit_equal=ok)
        _lib.set_tuning("spmm_unroll", 0)
        _lib.set_tuning("spmm_tail_nt", 0)
        # main rows at pitch 128 holding all d columns (ordinary [N, d] view for every consumer) + the tail table
        xp = torch.zeros((n, 128), device=device)
        xp[:, :d] = x0
        yp = torch.zeros((n, 128), device=device)
        for nt in (0, 1):
            _lib.set_tuning("spmm_tail_nt", nt)
            ms = time_ms(lambda: csr.spmm_tail(xp, xt, yp, yt, d, dm, tail_full=True))
            ok = bool(torch.equal(yp[:, :d], y_ref) and torch.equal(yt[:, :d - dm], y_ref[:, dm:]))
            line(f"{tag}split_pitch128_full", ms, nt_main=nt, bit_equal=ok)
        _lib.set_tuning("spmm_tail_nt", 0)
        # three chained hops in the split layout (what bench.py would time)
        bufs = [(torch.empty_like(xm), torch.empty_like(xt)) for _ in range(2)]

        def chain3():
            cm, ct = xm, xt
            for h in range(3):
                om, ot = bufs[h % 2]
                csr.spmm_tail(cm, ct, om, ot, d, dm)
                cm, ct = om, ot
        line(f"{tag}split_chain3", time_ms(chain3, reps=5) / 3)
        del csr

    run_set("", rowptr, col, val, x0)
    if a.hot:
        deg = rowptr[1:] - rowptr[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=device), deg)
        order = torch.argsort(deg, descending=True, stable=True)
        new_id = torch.empty_like(order)
        new_id[order] = torch.arange(n, device=device)
        c2 = new_id[col.long()]
        key, perm = torch.sort(rows * n + c2)
        c2 = (key % n).to(torch.int32)
        v2 = val[perm]
        del key, perm, rows
        x2 = x0[order].contiguous()
        for remap in (1, 0):
            _lib.set_tuning("spmm_xcd_remap", remap)
            run_set(f"hotcols_remap{remap}_", rowptr, c2, v2, x2)
        _lib.set_tuning("spmm_xcd_remap", 1)


if __name__ == "__main__":
    main()
