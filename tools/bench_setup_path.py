#!/usr/bin/env python3
"""What the one-off part of the path costs at the products shape (everything the reference redoes on the host at every
propagate call, operators/utils.py:76-88, base_op.py:20, data/base_data.py:29):
  ingest   COO edge list in file order -> canonical CSR   (sgl_coo_to_csr)
  prepare  A + I, degrees, symmetry check                 (PreparedAdjacency, once per graph)
  scale    one normalised adjacency per (r, alpha)
  plan     sgl_csr_create: items / long-row pieces of the SpMM plan
Prints HIP-event / wall times; used to look for slow spots off the timed SpMM loop."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import device as dev, io, synthetic  # noqa: E402


def wall(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), r


def main():
    wl = synthetic.WORKLOADS[os.environ.get("SETUP_WORKLOAD", "S1_products")]
    n = wl["n"]
    device = torch.device("cuda", 0)
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    nnz = a_col.numel()
    rows = torch.repeat_interleave(torch.arange(n, device=device), a_ptr[1:] - a_ptr[:-1])
    perm = torch.randperm(nnz, device=device)
    r64, c64, v = rows[perm].contiguous(), a_col.to(torch.int64)[perm].contiguous(), a_val[perm].contiguous()
    del rows, perm
    print(f"SETUP graph n={n} nnz={nnz}", flush=True)
    t, adj = wall(lambda: io.coo_to_csr_device(r64, c64, v, n, device=device))
    print(f"SETUP ingest  coo_to_csr (shuffled edge list)       ms={t:8.2f}  ({nnz / t / 1e6:.2f} G edges/s)", flush=True)
    assert torch.equal(adj.rowptr, a_ptr) and torch.equal(adj.col, a_col)
    del r64, c64, v, adj
    t, prep = wall(lambda: dev.PreparedAdjacency(a_ptr, a_col, a_val, n))
    print(f"SETUP prepare A + I, degrees, symmetry check        ms={t:8.2f}  symmetric={prep.symmetric}", flush=True)
    t, (rowptr, col, val) = wall(lambda: prep.normalize(0.5, None))
    print(f"SETUP scale   one (r, alpha)                        ms={t:8.2f}", flush=True)
    t, _ = wall(lambda: dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None, host_pow=False), reps=2)
    print(f"SETUP general directed pipeline (transpose by sort) ms={t:8.2f}", flush=True)
    # the same matrix with ONE value changed: not symmetric any more -> the directed route of PreparedAdjacency (the transpose is
    # built once, then every (r, alpha) is the single pass of the symmetric case)
    d_val = a_val.clone()
    d_val[12345] = 3.0
    t, dprep = wall(lambda: dev.PreparedAdjacency(a_ptr, a_col, d_val, n))
    print(f"SETUP directed: prepare (A + I, degrees, symmetry check) ms={t:8.2f}  symmetric={dprep.symmetric}", flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dprep.normalize(0.5, None)                                 # the very first call: builds the transpose (one stable sort)
    torch.cuda.synchronize()
    print(f"SETUP directed: first (r, alpha) incl. the transposition ms={(time.perf_counter() - t0) * 1e3:8.2f}", flush=True)
    t, _ = wall(lambda: dprep.normalize(0.3, None))
    print(f"SETUP directed: every further r                      ms={t:8.2f}", flush=True)
    t, _ = wall(lambda: dprep.normalize(0.4, 0.1))             # (warm-up call of wall(): gather pass + kept Laplacian; timed: the mix)
    print(f"SETUP directed: every further alpha of a PPR sweep   ms={t:8.2f}", flush=True)
    del dprep, d_val
    t, csr = wall(lambda: dev.DeviceCSR(rowptr, col, val, (n, n)))
    print(f"SETUP plan    sgl_csr_create                        ms={t:8.2f}", flush=True)
    x = synthetic.features_torch(n, wl["d"], seed=0, device=device)
    t, _ = wall(lambda: csr.spmm(x))
    print(f"SETUP (for scale) one SpMM hop                      ms={t:8.2f}", flush=True)


if __name__ == "__main__":
    main()
