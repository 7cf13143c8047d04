#!/usr/bin/env python3
"""End-to-end GraphOp.propagate at products scale FROM HOST INPUTS (scipy CSR + numpy features), the reference's exact
call shape: upload + device normalisation + k SpMMs (+ optional download), first call and cached second call."""
import os, sys, time
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgl_amd import synthetic
from sgl_amd.operators.graph_op import LaplacianGraphOp

wl = synthetic.WORKLOADS["S1_products"]; n, d = wl["n"], wl["d"]
rp, c, v = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device="cuda")
adj = sp.csr_matrix((v.cpu().numpy(), c.cpu().numpy(), rp.cpu().numpy().astype(np.int32)), shape=(n, n))
del rp, c, v
x = np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)
torch.cuda.synchronize()
for host_out in (False, True):
    op = LaplacianGraphOp(3, r=0.5, host_output=host_out)
    for call in ("first", "second (normalised adjacency cached)", "third (host result buffers recycled from the first)", "fourth"):
        t0 = time.perf_counter()
        hops = op.propagate(adj, x)
        torch.cuda.synchronize()
        print(f"E2E host_output={host_out} {call}: {time.perf_counter() - t0:.3f} s for k=3 on N={n}, nnz(A)={adj.nnz}, d={d}", flush=True)
    if not host_out:
        # a FRESH operator per trial over the same matrix (what a PaSca-style search does, search_models.py:19-46): the device copy
        # of A, A + I and the degrees are shared process-wide (operators.base_op.prepared_graph); a trial pays the content hash that
        # proves the matrix unchanged, one scaling pass, the plan and its k hops
        from sgl_amd.operators.graph_op import PprGraphOp
        for trial in (LaplacianGraphOp(3, r=0.3), PprGraphOp(3, r=0.5, alpha=0.15), PprGraphOp(3, r=0.5, alpha=0.3)):
            t0 = time.perf_counter()
            trial.propagate(adj, x)
            torch.cuda.synchronize()
            print(f"E2E fresh {type(trial).__name__} on the same matrix: {time.perf_counter() - t0:.3f} s for k=3", flush=True)
    if host_out:
        from sgl_amd import hostpool
        ref = LaplacianGraphOp(3, r=0.5).propagate(adj, x)
        print(f"E2E host pool: {hostpool.stats}; pinned results: {[bool(h.is_pinned()) for h in hops[1:]]}; "
              f"bit-equal to the device-resident hops: {all(torch.equal(h, r_.cpu()) for h, r_ in zip(hops[1:], ref[1:]))}", flush=True)
        t0 = time.perf_counter()
        from sgl_amd.operators.base_op import AdjIdentity
        fp = AdjIdentity.fingerprint(adj)
        print(f"E2E full fingerprint of the scipy matrix (indptr + indices + data, {(adj.indices.nbytes + adj.data.nbytes + adj.indptr.nbytes) / 1e9:.2f} GB): "
              f"{(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    del hops, op
