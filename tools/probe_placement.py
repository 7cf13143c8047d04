#!/usr/bin/env python3
"""Experiment: does the physical placement of the 57 GB gathered table change the gather rate?  (Two back-to-back runs of the
same papers-shard benchmark on one box differed by 10 %.)  One process, the same random row ids; the table is allocated, probed,
freed (torch.cuda.empty_cache -> hipFree) and allocated again, optionally with other allocations made first to move it."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd._lib import check, current_stream_ptr, lib, ptr  # noqa: E402


def time_ms(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(min(ts)), float(max(ts))


def main():
    dev = torch.device("cuda", 0)
    rows = int(os.environ.get("ROWS", 111_059_956))
    n_idx = 96 << 20
    g = torch.Generator(device=dev).manual_seed(3)
    idx = torch.randint(0, rows, (n_idx,), generator=g, device=dev, dtype=torch.int32)
    sink = torch.zeros(4, device=dev)
    spacers = []
    for trial in range(8):
        if trial in (3, 5):                        # move the table: park some memory first
            spacers.append(torch.empty((1 << 30) * (3 if trial == 3 else 7), dtype=torch.uint8, device=dev))
        table = torch.empty((rows, 128), device=dev)
        table.fill_(1.0)
        med, lo, hi = time_ms(lambda: check(lib().sgl_probe_gather_f32(ptr(table), 128, ptr(idx), n_idx, 128, 16, ptr(sink), current_stream_ptr())))
        print(f"PLACE trial={trial} table@0x{table.data_ptr():x} spacers={len(spacers)} ms med={med:.3f} min={lo:.3f} max={hi:.3f} "
              f"Ggather_per_s={n_idx / med / 1e6:.2f}", flush=True)
        del table
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
