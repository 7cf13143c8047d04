#!/usr/bin/env python3
"""How many 128-byte lines does a gathered row cost?  The products-shaped SpMM (d = 100: 400-byte rows, always 4 lines) against
rows that are 3 whole lines (d = 96 at a 128-float or 96-float pitch), 4 whole lines (d = 128) and a compact 16-byte tail table
(d = 4).  Result (profiles/r05_probe_split_rows.txt): the kernel moves ~57 G lines/s whatever the row width -- a REQUEST
ceiling, not a byte one -- so a split-row layout (first 96 columns = 3 whole lines, columns 96..99 from a compact [n, 4] table)
was built, found bit-identical and exactly as fast as the plain kernel (8.835 vs 8.856 ms: the 16-byte gather is one more line
request), and removed (DESIGN K1)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import device as dev, synthetic  # noqa: E402


def main():
    wl = synthetic.WORKLOADS["S1_products"]
    n = wl["n"]
    device = torch.device("cuda", 0)
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    csr = dev.DeviceCSR(rowptr, col, val, (n, n))
    nnz = col.numel()
    gen = torch.Generator(device=device).manual_seed(0)
    for d, ld in ((100, 100), (100, 128), (96, 128), (96, 96), (128, 128), (4, 4), (4, 128), (64, 64), (32, 32)):
        xb = torch.randn(n * ld + 64, device=device, generator=gen)
        yb = torch.empty(n * ld + 64, device=device)
        off = (-xb.data_ptr() // 4) % 32                      # 128-byte aligned first row
        x = torch.as_strided(xb, (n, d), (ld, 1), off)
        offy = (-yb.data_ptr() // 4) % 32
        y = torch.as_strided(yb, (n, d), (ld, 1), offy)
        for _ in range(2):
            csr.spmm(x, out=y)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            csr.spmm(x, out=y)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = float(np.median(ts))
        lines = -(-(d * 4) // 128) if (ld * 4) % 128 == 0 else None
        print(f"SPLIT d={d:4d} ld={ld:4d} ms={t:7.3f} Ggathers/s={nnz / t / 1e6:7.2f} lines/row={lines} "
              f"line_TB/s={(nnz * (lines or 4) * 128 / t / 1e9):6.2f}", flush=True)
        del xb, yb, x, y


if __name__ == "__main__":
    main()
