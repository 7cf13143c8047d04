#!/usr/bin/env python3
"""Workload for a rocprofv3 PMC pass: the SpMM of ONE feature-sharded rank at 8 GPUs (all rows x 13 of 100 columns,
stored at a 16-float pitch) -- evidence that a 64-byte row still costs a whole 128-byte line.

    cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -o pmc -- python tools/slice_pmc.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sgl_amd import synthetic  # noqa: E402
from sgl_amd import device as dev  # noqa: E402

device = torch.device("cuda", 0)
wl = synthetic.WORKLOADS["S1_products"]
n = wl["n"]
a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
csr = dev.DeviceCSR(rowptr, col, val, (n, n))
width = int(os.environ.get("SGL_SLICE_WIDTH", "16"))
xs = torch.randn((n, width), device=device)
ys = torch.empty_like(xs)
for _ in range(4):
    csr.spmm(xs, out=ys)
torch.cuda.synchronize()
print(f"slice width {width}: nnz={col.numel()} expected line bytes {col.numel() * 128 / 1e9:.2f} GB + CSR {col.numel() * 8 / 1e9:.2f} GB")
