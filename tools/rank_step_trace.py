#!/usr/bin/env python3
"""Everything one rank of the G-rank S1 job does in a step EXCEPT the network, through the real code path, on one GPU.

Rank r's block, halo plan (HaloPlan.offline: identical to the collective one), compact tables and HaloPropagator are built as
in the job; only `begin_exchange` is cut short after its pack kernel (nothing is sent, the ghost rows keep their hop-0 values --
irrelevant for time).  Timed: the step as the bench issues it (propagate_chunked with caller-owned tables), wall clock per step
with the host running ahead and with a synchronisation after every call (host-bound or not?).  Under
`rocprofv3 --kernel-trace --stats` the kernel list shows what really runs: SpMM per chunk, pack per chunk, fix-up -- and
nothing else (no copies, no fills).

    python tools/rank_step_trace.py [--ranks 8] [--rank 0] [--chunks 2]  > profiles/r03_rank_step_trace.log
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import device as dev  # noqa: E402
from sgl_amd import synthetic  # noqa: E402
from sgl_amd.dist import HaloPlan, HaloPropagator, balanced_bounds, column_chunks  # noqa: E402


class _Done:
    def wait(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--chunks", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = synthetic.WORKLOADS["S1_products"]
    n, d, K = wl["n"], wl["d"], wl["k"]
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    del a_ptr, a_col, a_val
    rp_host = rowptr.cpu().numpy()
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    G, r = a.ranks, a.rank
    bounds = balanced_bounds(rp_host, G)
    lo, hi = int(bounds[r]), int(bounds[r + 1])
    nb, ne = int(rp_host[lo]), int(rp_host[hi])
    plan = HaloPlan.offline(r, bounds, n, lambda q: col[int(rp_host[bounds[q]]):int(rp_host[bounds[q + 1]])])
    rp_local = (rowptr[lo:hi + 1] - rowptr[lo]).contiguous()
    csr = dev.DeviceCSR(rp_local, plan.relabel(col[nb:ne]), val[nb:ne].contiguous(), (hi - lo, plan.n_compact))
    prop = HaloPropagator(plan, lambda x, out: csr.spmm(x, out=out))
    t0 = x0.index_select(0, plan.global_ids)
    del x0, rowptr, col, val
    chunks = column_chunks(d, a.chunks)
    tables = [t0[:, c0:c1].contiguous() for c0, c1 in chunks]
    del t0
    bufs = [[torch.empty_like(t) for _ in range(K - 1)] for t in tables]
    ylast = [[None] * (K - 1) + [torch.empty((plan.n_own, t.shape[1]), device=device)] for t in tables]

    def no_network(y_own, table_next, key=0):
        prop._pack(y_own, key)
        return _Done()
    prop.begin_exchange = no_network

    def step():
        return prop.propagate_chunked(tables, K, buffers=bufs, y_buffers=ylast, hops_in_buffers=True)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t_a = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_host = (time.perf_counter() - t_a) / a.steps
    torch.cuda.synchronize()
    t_run = (time.perf_counter() - t_a) / a.steps
    # the same with the device idle at every call: what the host needs to issue one step
    t_b = time.perf_counter()
    for _ in range(a.steps):
        torch.cuda.synchronize()
        step()
    torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t_b) / a.steps
    # kernels alone, from events around each call
    def timed(fn, reps=7):
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
    ys = prop.spmm_only(tables)
    spmm = [timed(lambda t=t, y=y: csr.spmm(t, out=y)) for t, y in zip(tables, ys)]
    pack = [timed(lambda y=y, c=c: prop._pack(y, c)) for c, y in enumerate(ys)]
    kernels = K * sum(spmm) + (K - 1) * sum(pack)
    print(f"# tools/rank_step_trace.py: rank {r} of {G}, S1 (own rows {plan.n_own}, ghosts {plan.n_ghost}, rows sent {int(plan.send_off[-1])}), "
          f"k = {K}, column chunks {chunks}")
    print(f"RANKSTEP spmm per chunk ms {[round(v, 3) for v in spmm]}, pack per chunk ms {[round(v, 3) for v in pack]} "
          f"(order chosen: {prop.pack_timing_ms})")
    print(f"RANKSTEP kernels of one step (k x spmm + (k-1) x pack): {kernels:.3f} ms")
    print(f"RANKSTEP step as issued by propagate_chunked, host running ahead: {t_run * 1e3:.3f} ms per step "
          f"(host time to issue it: {t_host * 1e3:.3f} ms); with the device idle at every step: {t_sync * 1e3:.3f} ms")
    print(f"RANKSTEP an eighth of the single-GPU step would be {26.1 / G:.3f} ms" if G == 8 else "")


if __name__ == "__main__":
    main()
