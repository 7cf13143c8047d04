#!/usr/bin/env python3
"""Does address translation bound the 57 GB gather regime?  The bare row-gather probe (sgl_probe_gather_f32, 512-byte rows,
16 rows in flight per lane, uniform ids) against tables of 1 / 8 / 57 GB whose memory comes from different allocation paths:

    torch        torch.empty (hipMalloc through the caching allocator)
    default      sgl_mem_alloc(SGL_MEM_DEFAULT)        = hipMalloc
    contiguous   sgl_mem_alloc(SGL_MEM_CONTIGUOUS)     = one physically contiguous range
    vmm          sgl_mem_alloc(SGL_MEM_VMM, chunk 0)   = one hipMemCreate handle, one mapping
    vmm1g        ... chunk 1 GiB                        = 1 GiB physical chunks at 1 GiB-aligned addresses
    vmm2m        ... chunk 2 MiB                        = the controlled worst case: no fragment larger than 2 MiB can exist

    python tools/probe_tlb.py [--sizes 1,57] [--modes torch,contiguous,...] [--pmc]

--pmc runs each (size, mode) once without timing loops so that a rocprofv3 --pmc pass sees one probe dispatch per case, in the
order printed (tools/tlb_pmc.sh)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import device as dev  # noqa: E402
from sgl_amd import _lib  # noqa: E402
from sgl_amd._lib import check, current_stream_ptr, lib, ptr  # noqa: E402,F401

ROWS = {1: 2_449_029, 57: 111_059_956}     # the two benchmark tables; any other size (GB, may be fractional) is size * 2^30 / 512 rows
CHUNK = {"vmm": 0, "vmm1g": 1 << 30, "vmm2m": 2 << 20, "vmm64m": 64 << 20}


def table(rows, mode, device):
    if mode == "torch":
        return torch.empty((rows, 128), device=device)
    if mode in CHUNK:
        return dev.placed_empty((rows, 128), device, "vmm", CHUNK[mode])
    return dev.placed_empty((rows, 128), device, mode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1,57")
    ap.add_argument("--modes", default="torch,default,contiguous,vmm,vmm1g,vmm2m")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--reps", type=int, default=2, help="re-allocations per case (placement varies per allocation)")
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sink = torch.zeros(4, device=device)
    g = torch.Generator(device=device).manual_seed(3)
    n_idx = 96 << 20
    for size in [float(s) if "." in s else int(s) for s in a.sizes.split(",")]:
        rows = ROWS.get(size) or int(size * (1 << 30) / 512)
        idx = torch.randint(0, rows, (n_idx,), generator=g, device=device, dtype=torch.int32)
        for mode in a.modes.split(","):
            for rep in range(1 if a.pmc else a.reps):
                try:
                    t = table(rows, mode, device)
                except Exception as e:  # an allocation path the driver refuses is a result, not a crash
                    print(f"TLB table={size}GB mode={mode} rep={rep} ALLOC FAILED: {str(e)[:160]}", flush=True)
                    break
                flat = t.view(-1)
                step = 1 << 30
                for s in range(0, flat.numel(), step):
                    flat[s:s + step].uniform_(-1, 1, generator=g)

                def run():
                    _lib.check_probe(_lib.probe_lib().sgl_probe_gather_f32(ptr(t), 128, ptr(idx), n_idx, 128, 16, ptr(sink), current_stream_ptr()))
                run()
                torch.cuda.synchronize()
                if a.pmc:
                    print(f"TLB-PMC case table={size}GB mode={mode} addr=0x{t.data_ptr():x}", flush=True)
                else:
                    ts = []
                    for _ in range(5):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        run()
                        e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1))
                    ms = float(np.median(ts))
                    print(f"TLB table={size}GB mode={mode} rep={rep} addr=0x{t.data_ptr():x} ms={ms:.3f} "
                          f"Ggather_per_s={n_idx / (ms * 1e-3) / 1e9:.2f} line_TBps={n_idx * 512 / (ms * 1e-3) / 1e12:.2f}", flush=True)
                del t, flat
                torch.cuda.synchronize()
                if mode == "torch":
                    torch.cuda.empty_cache()
        del idx


if __name__ == "__main__":
    main()
