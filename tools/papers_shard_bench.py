#!/usr/bin/env python3
"""One rank's share of an ogbn-papers100M-shaped hop on ONE GPU (BASELINE configs 4/5 sizing): the rank owns 1/8 of the
rows (13.9 M rows, ~418 M non-zeros) but gathers from the FULL 111 M x 128 fp32 feature replica (56.9 GB, far beyond
the 256 MiB Infinity Cache).  Measures the per-rank SpMM time that the 8-GPU job's all-gather has to be weighed
against."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import device as dev  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    scale = 1.0 if free > 120e9 else 0.25
    n_cols = int(111_059_956 * scale)
    rows = n_cols // 8
    d = 128
    g = torch.Generator(device=device).manual_seed(0)
    deg = torch.exp(torch.randn(rows, generator=g, device=device) * 1.1 + 2.8).clamp_(1, 20000).long()   # mean ~30
    rowptr = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(deg, 0)
    nnz = int(rowptr[-1])
    w = torch.exp(torch.randn(n_cols, generator=g, device=device) * 1.2)
    cdf = torch.cumsum(w.double(), 0)
    cdf /= cdf[-1].clone()
    del w
    col = torch.empty(nnz, dtype=torch.int32, device=device)
    step = 1 << 26
    for s in range(0, nnz, step):
        e = min(nnz, s + step)
        col[s:e] = torch.searchsorted(cdf, torch.rand(e - s, generator=g, device=device, dtype=torch.float64)).clamp_(0, n_cols - 1).int()
    del cdf
    val = torch.rand(nnz, generator=g, device=device) * 0.1
    x = torch.empty((n_cols, d), device=device)
    x.normal_(generator=g)
    y = torch.empty((rows, d), device=device)
    t0 = time.time()
    csr = dev.DeviceCSR(rowptr, col, val, (rows, n_cols))
    torch.cuda.synchronize()
    print(f"PAPERS shard: rows={rows} n_cols={n_cols} nnz={nnz} d={d} X={n_cols * d * 4 / 1e9:.1f} GB  plan build {time.time() - t0:.2f}s  {csr.info()}", flush=True)
    for _ in range(2):
        csr.spmm(x, out=y)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        csr.spmm(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    alg = nnz * d * 4 + nnz * 8 + (rows + 1) * 4 + rows * d * 4
    print(f"PAPERS shard: {ms:.2f} ms per hop per rank  = {nnz * d / (ms * 1e-3) / 1e12:.3f}e12 edge*feat/s per GPU, "
          f"{nnz / (ms * 1e-3) / 1e9:.2f} G gathers/s, algorithmic-roofline fraction {alg / (ms * 1e-3) / 8e12:.3f}", flush=True)
    if os.environ.get("PAPERS_SWEEP", "0") == "group":
        # d = 128 as 32 lanes x float4 with TWO non-zeros per step (R = 2) instead of 64 lanes of which 32 idle
        from sgl_amd import _lib
        ref = y.clone()

        def timed_g():
            for _ in range(2):
                csr.spmm(x, out=y)
            torch.cuda.synchronize()
            tt = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                csr.spmm(x, out=y)
                e1.record()
                torch.cuda.synchronize()
                tt.append(e0.elapsed_time(e1))
            return float(np.median(tt))
        for group in (0, 32):
            for unroll in (0, 1, 3, 2, 4):            # default / 4 / 8 / 16 / (32) gathers in flight per lane
                _lib.set_tuning("spmm_group", group)
                _lib.set_tuning("spmm_unroll", unroll)
                ms_g = timed_g()
                err = float((y - ref).abs().max() / ref.abs().max())
                print(f"PAPERS group sweep group={group or 64} unroll_knob={unroll} ms={ms_g:.2f}  frac={alg / (ms_g * 1e-3) / 8e12:.3f}  "
                      f"max|dy|/max|y| vs default = {err:.2e}", flush=True)
        _lib.set_tuning("spmm_group", 0)
        _lib.set_tuning("spmm_unroll", 0)
        return
    if os.environ.get("PAPERS_SWEEP", "0") == "1":
        from sgl_amd import _lib
        def timed():
            for _ in range(2):
                csr.spmm(x, out=y)
            torch.cuda.synchronize()
            tt = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                csr.spmm(x, out=y)
                e1.record()
                torch.cuda.synchronize()
                tt.append(e0.elapsed_time(e1))
            return float(np.median(tt))
        for remap in (1, 0):
            for unroll in (1, 3, 2):        # 4 / 8 / 16 gathers in flight per lane
                _lib.set_tuning("spmm_xcd_remap", remap)
                _lib.set_tuning("spmm_unroll", unroll)
                print(f"PAPERS sweep xcd_remap={remap} unroll={unroll} ms={timed():.2f}", flush=True)
        _lib.set_tuning("spmm_unroll", 0)
        _lib.set_tuning("spmm_xcd_remap", 1)
        for waves in (1, 2):
            _lib.set_tuning("spmm_waves", waves)
            print(f"PAPERS sweep waves={waves} ms={timed():.2f}", flush=True)
        _lib.set_tuning("spmm_waves", 0)
        for remap in (0,):
            _lib.set_tuning("spmm_xcd_remap", remap)
            print(f"PAPERS sweep xcd_remap={remap} ms={timed():.2f}", flush=True)
        _lib.set_tuning("spmm_xcd_remap", 1)
        for item_nnz in (128, 2048):
            c2 = dev.DeviceCSR(rowptr, col, val, (rows, n_cols), item_nnz=item_nnz)
            csr_keep, csr = csr, c2
            print(f"PAPERS sweep item_nnz={item_nnz} ms={timed():.2f}", flush=True)
            csr = csr_keep
            del c2
    inbound = (7 / 8) * n_cols * d * 4
    print(f"PAPERS shard: all-gather in-bound per rank per hop {inbound / 1e9:.1f} GB -> >= {inbound / 537e9 * 1e3:.0f} ms at 7 x 76.8 GB/s", flush=True)
    # the other layouts of DESIGN.md section 6 on the same row block: column slices of the replica (16 columns = the
    # feature-sharded layout at 8 ranks, 32 columns = the 2 x 4 grid).  A feature-sharded rank multiplies ALL rows, i.e.
    # 8 blocks like this one, a grid rank 4 of them.
    del x, y
    if os.environ.get("PAPERS_SWEEP", "0") == "1":
        return
    for w, blocks, what in ((16, 8, "feature-sharded x8: no communication"),
                            (32, 4, f"grid 2 x 4: {n_cols / 2 * 32 * 4 / 1e9:.1f} GB in-bound per hop, relayed over 7 links")):
        xs = torch.empty((n_cols, w), device=device)
        xs.normal_(generator=g)
        ys = torch.empty((rows, w), device=device)
        for _ in range(2):
            csr.spmm(xs, out=ys)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            csr.spmm(xs, out=ys)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms_w = float(np.median(ts))
        print(f"PAPERS slice: {w} columns ({n_cols * w * 4 / 1e9:.1f} GB slice): {ms_w:.2f} ms for this row block, "
              f"{nnz / (ms_w * 1e-3) / 1e9:.2f} G gathers/s -> {blocks} blocks = {blocks * ms_w:.0f} ms per hop per rank ({what})", flush=True)
        del xs, ys


if __name__ == "__main__":
    main()
