#!/usr/bin/env python3
"""The pack step of the need-aware exchange in isolation (sgl_gather_rows_f32 / sgl_scatter_rows_f32).

PACK        one row copy out[i] = x[idx[i]] (sorted index with repeats) at the shapes of the S1 job on 8 ranks (306 k own rows,
            1.8 M rows sent, 64 / 36 / 100 columns), the training-feed gather (200 k of 2.4 M rows), a whole-matrix copy and a
            papers100M-sized block, next to torch.index_select
PACK fill   pure writes (torch fill_ / zero_): the write-only ceiling
PACK2       the real index pattern -- every peer's sorted list of ~84 % of the own rows, peer after peer -- packed in PEER order
            (gather: sequential writes, every own row re-read once per peer) and in OWN-ROW order (scatter: every row read once,
            written to 7 places); HaloPropagator times both on the first pack of a shape and keeps the faster

    python tools/probe_pack.py > profiles/r03_pack_order.log
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgl_amd import device as dev  # noqa: E402


def timed(fn, reps=9, warm=3):
    for _ in range(warm):
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
    g = torch.Generator(device="cuda").manual_seed(1)
    print("# tools/probe_pack.py: median of 9 launches")
    for n_src, n_idx, d in ((306368, 1806000, 64), (306368, 1806000, 36), (306368, 1806000, 100), (2449029, 200000, 100),
                            (2449029, 2449029, 100), (13876991, 11300000, 64)):
        x = torch.randn((n_src, d), device="cuda")
        idx = torch.sort(torch.randint(0, n_src, (n_idx,), device="cuda", generator=g))[0]
        out = torch.empty((n_idx, d), device="cuda")
        t = timed(lambda: dev.gather_rows(x, idx, out=out))
        ok = torch.equal(out, x.index_select(0, idx))
        tt = timed(lambda: torch.index_select(x, 0, idx, out=out))
        print(f"PACK n_src={n_src} n_idx={n_idx} d={d}: {t:.3f} ms = {(n_idx * d * 4) / t / 1e9:.2f} TB/s written, "
              f"torch {tt:.3f} ms, equal={ok}", flush=True)
        del x, idx, out
    for mb in (260, 462, 980, 4000):
        buf = torch.empty(mb * 250000, device="cuda")
        t = timed(lambda: buf.fill_(1.0))
        t2 = timed(lambda: buf.zero_())
        print(f"PACK fill {mb} MB: fill_ {t:.3f} ms = {mb / t / 1e3:.2f} TB/s, zero_ {t2:.3f} ms = {mb / t2 / 1e3:.2f} TB/s", flush=True)
        del buf
    for n_src, n_idx, d in ((306368, 1806000, 64), (306368, 1806000, 36), (306368, 1806000, 32), (306368, 1806000, 100),
                            (306368, 1806000, 4), (13876991, 90000000, 64)):
        x = torch.randn((n_src, d), device="cuda")
        per = n_idx // 7
        idx = torch.cat([torch.sort(torch.randperm(n_src, device="cuda", generator=g)[:per])[0] for _ in range(7)])
        src, dst = torch.sort(idx, stable=True)
        out = torch.empty((idx.numel(), d), device="cuda")
        t = timed(lambda: dev.gather_rows(x, idx, out=out))
        ref = out.clone()
        t2 = timed(lambda: dev.scatter_rows(x, src, dst, out))
        print(f"PACK2 n_src={n_src} n_idx={idx.numel()} d={d}: gather (peer order) {t:.3f} ms, scatter (own-row order) {t2:.3f} ms, "
              f"equal={torch.equal(ref, out)}", flush=True)
        del x, out, ref, idx, src, dst


if __name__ == "__main__":
    main()
