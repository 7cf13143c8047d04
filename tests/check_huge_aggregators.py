#!/usr/bin/env python3
"""Robustness check at the largest shape of the contract: the aggregators over WHOLE ogbn-papers100M-sized hop matrices
([111 059 956, 128] = 14.2 G elements = 57 GB each) on one GPU -- more elements than a 32-bit index, more threads than one HIP
launch carries.  Checks sampled rows against numpy / the oracle and that kernels which cannot take the shape say so.
(A script, not a collected test: it needs 120 GB of HBM and a minute.  It lives under tests/ because it uses the oracle.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import _lib, synthetic  # noqa: E402
from sgl_amd import device as dev  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    n, d = 111_059_956, 128
    feats = []
    for h in range(2):
        t = torch.empty((n, d), dtype=torch.float32, device=device)
        synthetic.hashed_features_torch(5 + h, 0, n, d, device=device, out=t)
        feats.append(t)
    torch.cuda.synchronize()
    rows = np.concatenate([np.arange(0, 5), np.arange(n - 5, n), np.random.default_rng(0).integers(0, n, 200)])
    idx = torch.from_numpy(rows).to(device)
    host = [f[idx].cpu().numpy() for f in feats]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, op, ref in (("sum", _lib.SGL_REDUCE_SUM, lambda a, b: (np.float32(0) + a) + b), ("max", _lib.SGL_REDUCE_MAX, np.maximum)):
        e0.record()
        y = dev.hop_reduce(op, feats)
        e1.record()
        torch.cuda.synchronize()
        ok = np.array_equal(y[idx].cpu().numpy(), ref(host[0], host[1]))
        ms = e0.elapsed_time(e1)
        print(f"HUGE {name}: 2 x [{n}, {d}] -> ok={ok} ms={ms:.1f} ({3 * n * d * 4 / ms / 1e9:.2f} TB/s)", flush=True)
        del y
    w = torch.tensor([0.25, -1.5], device=device)
    y = dev.hop_reduce(_lib.SGL_REDUCE_WSUM, feats, w)
    ok = np.allclose(y[idx].cpu().numpy(), np.float32(0.25) * host[0] + np.float32(-1.5) * host[1], rtol=1e-6, atol=1e-6)
    print(f"HUGE wsum1d: ok={ok}", flush=True)
    del y
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    import oracle as orc
    try:
        e0.record()
        y = dev.nafs_aggregate(feats)
        e1.record()
        torch.cuda.synchronize()
        got = y[idx].cpu().numpy()
        want = orc.agg_over_smooth_distance(host)
        print(f"HUGE nafs: ok={bool(orc.parity_ok(got, want, 1e-5, rowwise=False))} ms={e0.elapsed_time(e1):.1f}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"HUGE nafs: refused loudly: {str(e)[:120]}", flush=True)


if __name__ == "__main__":
    main()
