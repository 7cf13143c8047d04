#!/usr/bin/env python3
"""Experiment: d = 100 as a 96-column main pass (3 lines per gathered row) + an L2-blocked pass for the 4-column tail
(sgl_tailpass.hip), against the plain pitch-100 kernel.  Prints `EXP tailpass ...` lines.

    python tools/exp_tailpass.py [--workload S1_products]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import _lib, synthetic  # noqa: E402
from sgl_amd import device as dev  # noqa: E402
from sgl_amd._lib import check, current_stream_ptr, lib, ptr  # noqa: E402


def time_ms(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S1_products")
    a = ap.parse_args()
    L = lib()
    c_i64, c_i32, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
    L.sgl_exp_tailpass_offsets.argtypes = [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]
    L.sgl_exp_tailpass_offsets.restype = c_i32
    L.sgl_exp_tailpass_run.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]
    L.sgl_exp_tailpass_run.restype = c_i32
    device = torch.device("cuda", 0)
    wl = synthetic.WORKLOADS[a.workload]
    n, d = wl["n"], wl["d"]
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    del a_ptr, a_col, a_val
    nnz = col.numel()
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    alg = nnz * d * 4 + nnz * 8 + (n + 1) * 4 + n * d * 4
    csr = dev.DeviceCSR(rowptr, col, val, (n, n))
    y_ref = torch.empty_like(x0)
    base = time_ms(lambda: csr.spmm(x0, out=y_ref))
    print(f"EXP tailpass base_pitch{d} ms={base:.3f} frac={alg / (base * 1e-3) / 8e12:.3f}", flush=True)
    # main pass: rows at pitch 128 holding all d columns, the kernel run on the first 96
    xp = torch.zeros((n, 128), device=device)
    xp[:, :d] = x0
    yp = torch.zeros((n, 128), device=device)
    main = time_ms(lambda: csr.spmm(xp[:, :96], out=yp[:, :96]))
    print(f"EXP tailpass main_96_of_pitch128 ms={main:.3f}", flush=True)
    xt = x0[:, 96:100].contiguous()
    yt = torch.zeros((n, 4), device=device)
    for cb_log in (18, 17, 19, 16):
        cb = 1 << cb_log
        nb = (n + cb - 1) // cb
        off = torch.empty((nb + 1) * n, dtype=torch.int32, device=device)
        check(L.sgl_exp_tailpass_offsets(ptr(rowptr), ptr(col), n, nb, cb, ptr(off), current_stream_ptr()), "offsets")
        seg = (off.view(nb + 1, n)[1:] - off.view(nb + 1, n)[:-1])
        for long_len in (64, 256):
            long_rows = torch.nonzero((seg > long_len).any(0)).view(-1).to(torch.int32)
            for unroll in (2, 4, 8):
                def run():
                    check(L.sgl_exp_tailpass_run(ptr(rowptr), ptr(col), ptr(val), ptr(off), n, nb, ptr(long_rows) if long_rows.numel() else None,
                                                 long_rows.numel(), long_len, ptr(xt), ptr(yt), ptr(yp), 128, 96, unroll,
                                                 current_stream_ptr()), "run")
                ms = time_ms(run)
                err = (yt - y_ref[:, 96:100]).abs().max().item() / y_ref[:, 96:100].abs().max().item()
                same = (yt == y_ref[:, 96:100]).float().mean().item()
                tot = main + ms
                print(f"EXP tailpass cols_per_block=2^{cb_log} blocks={nb} long_len={long_len} long_rows={long_rows.numel()} unroll={unroll} "
                      f"tail_ms={ms:.3f} total_ms={tot:.3f} frac={alg / (tot * 1e-3) / 8e12:.3f} vs_base={base / tot:.3f} "
                      f"max_rel_err={err:.2e} bit_equal_share={same:.4f}", flush=True)
        del off, seg
    ok = torch.equal(yp[:, :96], y_ref[:, :96])
    print(f"EXP tailpass main block bit-equal: {ok}; tail written into the pitch-128 rows equal to the table: {torch.equal(yp[:, 96:100], yt)}")


if __name__ == "__main__":
    main()
