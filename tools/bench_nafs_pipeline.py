#!/usr/bin/env python3
"""The NAFS task pipeline (tricks/nafs_features.nafs_ensemble_features; reference tasks/node_clustering.py:205-258) at the
products shape on one GPU: 6 r values x `hops` SpMM hops + the per-node hop weighting + the ensemble, from a DeviceAdjacency.
Prints the wall time and what the SpMMs alone would take (the floor of the pipeline)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import device as dev, synthetic  # noqa: E402
from sgl_amd.io import DeviceAdjacency  # noqa: E402
from sgl_amd.tricks.nafs_features import nafs_ensemble_features  # noqa: E402


def main():
    wl = synthetic.WORKLOADS[os.environ.get("NAFS_WORKLOAD", "S1_products")]
    n, d = wl["n"], wl["d"]
    device = torch.device("cuda", 0)
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    adj = DeviceAdjacency(a_ptr, a_col, a_val, (n, n))
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    rp, c, v = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    csr = dev.DeviceCSR(rp, c, v, (n, n))
    y = csr.spmm(x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        csr.spmm(x0, out=y)
    torch.cuda.synchronize()
    hop_ms = (time.perf_counter() - t0) / 5 * 1e3
    del csr, y, rp, c, v
    r_list = (0.5, 0.4, 0.3, 0.2, 0.1, 0)
    for hops in (10, 20):
        for method in ("mean", "concat"):
            nafs_ensemble_features(adj, x0, 2, r_list=r_list[:2], method=method)         # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = nafs_ensemble_features(adj, x0, hops, r_list=r_list, method=method)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            floor = len(r_list) * hops * hop_ms
            print(f"NAFS pipeline hops={hops:2d} r x{len(r_list)} method={method:6s} ms={ms:8.1f}  "
                  f"(SpMM floor {floor:7.1f} ms = {len(r_list) * hops} hops x {hop_ms:.2f}; everything else {ms - floor:6.1f} ms) "
                  f"out={tuple(out.shape)}", flush=True)
            del out


if __name__ == "__main__":
    main()
