#!/usr/bin/env python3
"""One rank's COMPUTE share of every multi-GPU layout, measured on a single GPU (no communication): what the SpMM side
of a hop costs per rank for N = 2, 4, 8 with the products-shaped workload.  The 8-GPU job itself is the driver's to run;
this gives the compute floor of each layout that bench.py's auto-selection chooses from.

    EXP share layout=<cols|rows|grid2> world=<N> ms_per_hop=<t> speedup_bound=<single-GPU hop / t>
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sgl_amd import synthetic  # noqa: E402
from sgl_amd import device as dev  # noqa: E402
from sgl_amd.dist import all_piece_bounds, column_slices, device_piece_spmms  # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
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
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = synthetic.WORKLOADS[os.environ.get("SGL_WORKLOAD", "S1_products")]
    n, d = wl["n"], wl["d"]
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    del a_ptr, a_col, a_val
    rp_host = rowptr.cpu().numpy()
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    full = dev.DeviceCSR(rowptr, col, val, (n, n))

    def slice_of(a, b):
        w = b - a
        t = torch.zeros((n, dev.row_pitch(w, growth=2.0)), dtype=torch.float32, device=device)
        t[:, :w] = x0[:, a:b]
        return t

    y = dev.alloc_rows(n, d, device)
    base = timed(lambda: full.spmm(x0, out=y))
    print(f"EXP share layout=single world=1 ms_per_hop={base:.3f} speedup_bound=1.00", flush=True)
    for world in (2, 4, 8):
        a, b = column_slices(d, world)[0]
        xs = slice_of(a, b)
        ys = torch.empty_like(xs)
        t = timed(lambda: full.spmm(xs, out=ys))
        print(f"EXP share layout=cols world={world} width={b - a} pitch={xs.shape[1]} ms_per_hop={t:.3f} "
              f"speedup_bound={base / t:.2f}", flush=True)
        extra = [("grid2", 2, int(p)) for p in os.environ.get("SGL_GRID_PIECES", "").split(",") if p]
        for name, row_groups, pieces in [("rows", world, 2), ("grid2", 2, 4)] + extra:
            if name == "grid2" and world < 4:
                continue
            col_groups = world // row_groups
            a, b = column_slices(d, col_groups)[0]
            xs = slice_of(a, b) if col_groups > 1 else x0
            pb = all_piece_bounds(rp_host, row_groups, pieces)
            worst = 0.0
            for rg in (0, row_groups - 1):
                fns, handles = device_piece_spmms(rowptr, col, val, n, pb[rg], rowptr_host=rp_host)
                outs = [torch.empty((int(pb[rg, p + 1] - pb[rg, p]), xs.shape[1]), dtype=torch.float32, device=device)
                        for p in range(pieces)]

                def hop():
                    for p in range(pieces):
                        fns[p](xs, outs[p])
                worst = max(worst, timed(hop))
                if os.environ.get("SGL_TWO_STREAMS") and rg == 0:
                    main, aux = torch.cuda.current_stream(), torch.cuda.Stream()

                    def hop2():
                        aux.wait_stream(main)
                        for p in range(pieces):
                            with torch.cuda.stream(aux if p % 2 else main):
                                fns[p](xs, outs[p])
                        main.wait_stream(aux)
                    print(f"EXP share layout={name} world={world} pieces={pieces} two_streams ms_per_hop={timed(hop2):.3f} "
                          f"one_stream={timed(hop):.3f}", flush=True)
                del fns, handles, outs
            inbound = (row_groups - 1) / row_groups * n * xs.shape[1] * 4
            print(f"EXP share layout={name} world={world} grid={row_groups}x{col_groups} pieces={pieces} "
                  f"ms_per_hop={worst:.3f} speedup_bound={base / worst:.2f} inbound_MB_per_hop={inbound / 1e6:.0f}", flush=True)


if __name__ == "__main__":
    main()
