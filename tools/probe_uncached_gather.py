#!/usr/bin/env python3
"""Experiment: does placing the gathered matrix X in uncached / fine-grained device memory change the gather rate?

The SpMM is bound by 128-byte line fetches (DESIGN K1).  `hipExtMallocWithFlags(hipDeviceMallocUncached)` pages are not
allocated in L2, so narrow rows (d = 16 -> 64 B) might be fetched as 64-byte requests instead of whole lines.
Prints `EXP uncached kind=<default|finegrained|uncached> d=<d> ms_per_hop=<t>`.
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sgl_amd import _lib, synthetic  # noqa: E402
from sgl_amd import device as dev  # noqa: E402

FLAGS = {"default": 0x0, "finegrained": 0x1, "uncached": 0x3}


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    wl = synthetic.WORKLOADS["S1_products"]
    n = wl["n"]
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    del a_ptr, a_col, a_val
    csr = dev.DeviceCSR(rowptr, col, val, (n, n))
    L = _lib.lib()
    stream = _lib.current_stream_ptr()
    for d in (16, 32, 64, 100):
        ld = dev.row_pitch(d)
        x = dev.alloc_rows(n, d, device, zero_pad=True)
        x.copy_(synthetic.features_torch(n, d, seed=0, device=device))
        parent = dev.padded_parent(x)
        y = dev.alloc_rows(n, d, device, zero_pad=True)
        ref = None
        for kind, flag in FLAGS.items():
            p = ctypes.c_void_p()
            rc = hip.hipExtMallocWithFlags(ctypes.byref(p), parent.numel() * 4, flag)
            if rc != 0:
                print(f"EXP uncached kind={kind} d={d} alloc_failed rc={rc}", flush=True)
                continue
            torch.cuda.synchronize()
            assert hip.hipMemcpy(p, ctypes.c_void_p(parent.data_ptr()), parent.numel() * 4, 3) == 0

            def hop():
                _lib.check(L.sgl_spmm_f32(csr._h, p, ld, ctypes.c_void_p(y.data_ptr()), dev._ld(y), d, 0, stream), "spmm")
            for _ in range(2):
                hop()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                hop()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            if ref is None:
                ref = y.clone()
            same = bool(torch.equal(ref, y))
            print(f"EXP uncached kind={kind} d={d} pitch={ld} ms_per_hop={np.median(ts):.3f} same={same}", flush=True)
            hip.hipFree(p)


if __name__ == "__main__":
    main()
