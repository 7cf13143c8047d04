#!/usr/bin/env python3
"""What the MI355X memory system delivers to the two access patterns of the SpMM, measured with the bare probes of
sgl_probe.hip (no CSR stream, no stores): sequential 16-byte reads, and random whole-row gathers for several row
widths / table sizes / gathers in flight.  The SpMM kernel's gather rate is quoted against these ceilings in DESIGN.md.

    python tools/mem_ceilings.py [--big]      (--big adds the 57 GB papers100M-sized table)
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import _lib  # noqa: E402
from sgl_amd._lib import check, current_stream_ptr, lib, ptr  # noqa: E402,F401


def time_ms(fn, reps=5, warm=2):
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
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sink = torch.zeros(4, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)

    n_fl = 2 * (1 << 30)   # 8 GiB
    big = torch.empty(n_fl, device=dev)
    big.normal_(generator=g)
    ms = time_ms(lambda: _lib.check_probe(_lib.probe_lib().sgl_probe_stream_f32(ptr(big), n_fl, ptr(sink), current_stream_ptr())))
    print(f"CEIL stream_read bytes={n_fl * 4 / 1e9:.1f}GB ms={ms:.3f} TBps={n_fl * 4 / (ms * 1e-3) / 1e12:.2f}", flush=True)
    del big

    n_idx = 96 * (1 << 20)
    tables = [("S1_980MB", 2_449_029)]
    if a.big:
        tables += [("4GB", 8 << 20), ("papers_57GB", 111_059_956)]
    for tname, rows in tables:
        table = torch.empty((rows, 128), device=dev)
        table.normal_(generator=g)
        for dist in ("uniform", "hubs"):
            if dist == "uniform":
                idx = torch.randint(0, rows, (n_idx,), generator=g, device=dev, dtype=torch.int32)
            else:
                # endpoints drawn proportionally to log-normal weights (sigma 1.2): the gather stream of the benchmark graphs
                w = torch.exp(torch.randn(rows, generator=g, device=dev, dtype=torch.float64) * 1.2)
                cdf = torch.cumsum(w, 0)
                cdf /= cdf[-1].clone()
                idx = torch.empty(n_idx, dtype=torch.int32, device=dev)
                step = 1 << 25
                for s in range(0, n_idx, step):
                    e = min(n_idx, s + step)
                    idx[s:e] = torch.searchsorted(cdf, torch.rand(e - s, generator=g, device=dev, dtype=torch.float64)).clamp_(0, rows - 1).int()
                del w, cdf
            for ld, rf, what in ((128, 128, "512B rows, 4 lines"), (128, 96, "384B of 512B-pitch rows, 3 lines"),
                                 (100, 100, "400B rows at pitch 100, 4 lines"), (96, 96, "384B rows at pitch 96, 3 lines")):
                tv = table.view(-1)[: rows * ld].view(rows, ld)
                for fl in (8, 16, 32):
                    ms = time_ms(lambda: _lib.check_probe(_lib.probe_lib().sgl_probe_gather_f32(ptr(tv), ld, ptr(idx), n_idx, rf, fl, ptr(sink),
                                                                          current_stream_ptr())))
                    lines = (rf * 4 + 127) // 128 if (ld * 4) % 128 == 0 else 4
                    print(f"CEIL gather table={tname} dist={dist} [{what}] in_flight={fl} ms={ms:.3f} "
                          f"Ggather_per_s={n_idx / (ms * 1e-3) / 1e9:.2f} useful_TBps={n_idx * rf * 4 / (ms * 1e-3) / 1e12:.2f} "
                          f"line_TBps={n_idx * lines * 128 / (ms * 1e-3) / 1e12:.2f}", flush=True)
            del idx
        del table


if __name__ == "__main__":
    main()
