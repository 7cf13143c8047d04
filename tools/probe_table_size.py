#!/usr/bin/env python3
"""Random 512-byte row gathers (the bare probe of sgl_probe.hip: no CSR stream, no arithmetic, no stores) from tables of 2 MB ... 2 GB:
which level of the memory system a gathered table lives in, and the line rate that level gives.  Result (profiles/
r06_gather_rate_vs_table_size.log): only the L2 is faster than the fabric -- a table that fits the 256 MB Infinity Cache gathers at
60-63 G lines/s, a 1 GB one at 57, HBM alone at 48; an L2-resident one at 140."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from sgl_amd import _lib
from sgl_amd._lib import current_stream_ptr, ptr
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sink = torch.zeros(4, device=dev); g = torch.Generator(device=dev).manual_seed(3)
def time_ms(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
n_idx = 96 * (1 << 20)
for mb in (2, 4, 8, 16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    rows = mb * (1 << 20) // 512
    table = torch.empty((rows, 128), device=dev); table.normal_(generator=g)
    idx = torch.randint(0, rows, (n_idx,), generator=g, device=dev, dtype=torch.int32)
    ms = time_ms(lambda: _lib.check_probe(_lib.probe_lib().sgl_probe_gather_f32(ptr(table), 128, ptr(idx), n_idx, 128, 16, ptr(sink), current_stream_ptr())))
    print(f"SIZE table={mb:5d} MB (512-byte rows): {n_idx / ms / 1e6:6.2f} G rows/s = {n_idx * 4 / ms / 1e6:6.1f} G lines/s = {n_idx * 512 / ms / 1e9:5.2f} TB/s", flush=True)
    del table, idx
