#!/usr/bin/env python3
"""The NAFS task's hop SWEEP at the products shape on one GPU (reference: NodeClusteringNAFS._execute evaluates every hop count of
range(hops), tasks/node_clustering.py:139,176-178; hops = 20, 6 r values are its defaults).

  sweep      nafs_ensemble_sweep: 6 x 19 SpMMs + one prefix pass per r (sgl_nafs_prefix_f32), all 20 feature matrices
  per-count  what the same library did before: nafs_ensemble_features(hops = h) for every h -- 6 x (0 + ... + 19) = 1 140 SpMMs --
             measured on a few hop counts and summed over the range with the per-hop cost they show
Prints wall times, the SpMM floor, and the prefix kernel's own time and rate (bytes: every hop matrix read once + every emitted
matrix written once [+ read once when combined into the ensemble])."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import device as dev, synthetic  # noqa: E402
from sgl_amd.io import DeviceAdjacency  # noqa: E402
from sgl_amd.tricks.nafs_features import nafs_ensemble_features, nafs_ensemble_sweep  # noqa: E402


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


def main():
    wl = synthetic.WORKLOADS[os.environ.get("NAFS_WORKLOAD", "S1_products")]
    n, d = wl["n"], wl["d"]
    hops = int(os.environ.get("NAFS_HOPS", "20"))
    device = torch.device("cuda", 0)
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    adj = DeviceAdjacency(a_ptr, a_col, a_val, (n, n))
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    rp, c, v = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    csr = dev.DeviceCSR(rp, c, v, (n, n))
    y = csr.spmm(x0)
    hop_ms, _ = wall(lambda: [csr.spmm(x0, out=y) for _ in range(5)])
    hop_ms /= 5
    # the prefix kernel alone: `hops` hop matrices, every prefix emitted (store), then combined (add)
    feats = [dev.upload_rows(x0, device)] + [dev.alloc_rows(n, d, device).normal_() for _ in range(hops - 1)]
    emit = list(range(hops))
    outs = dev.nafs_prefix(feats, emit)
    for combine, name, streams in ((dev.NAFS_STORE, "store", 2 * hops), (dev.NAFS_ADD, "add", 3 * hops)):
        ms, _ = wall(lambda: [dev.nafs_prefix(feats, emit, outs=outs, combine=combine) for _ in range(3)])
        ms /= 3
        by = streams * n * d * 4
        print(f"prefix kernel  hops={hops} d={d} emit=all combine={name:5s} ms={ms:7.2f}  {by / ms / 1e9:6.2f} TB/s "
              f"({by / ms / 1e9 / 8.0:.3f} of peak; {streams} matrix streams of {n * d * 4 / 1e9:.2f} GB)", flush=True)
    ms, _ = wall(lambda: [dev.nafs_prefix(feats, [hops - 1], outs=outs[-1:]) for _ in range(3)])
    print(f"prefix kernel  hops={hops} d={d} emit=last only          ms={ms / 3:7.2f}  {(hops + 1) * n * d * 4 / (ms / 3) / 1e9:6.2f} TB/s", flush=True)
    del feats, outs, csr, y, rp, c, v
    r_list = (0.5, 0.4, 0.3, 0.2, 0.1, 0)
    nafs_ensemble_sweep(adj, x0, 3, r_list=r_list[:2], method="mean")                     # warm-up
    for method in ("mean", "max", "concat"):
        ms, out = wall(lambda: nafs_ensemble_sweep(adj, x0, hops, r_list=r_list, method=method))
        n_spmm = len(r_list) * (hops - 1)
        print(f"NAFS sweep     hops=range({hops}) r x{len(r_list)} method={method:6s} ms={ms:8.1f}  (SpMM floor {n_spmm * hop_ms:7.1f} ms = "
              f"{n_spmm} hops x {hop_ms:.2f}; everything else {ms - n_spmm * hop_ms:6.1f} ms; {len(out)} feature matrices)", flush=True)
        del out
    # per-count route on three hop counts, extrapolated over the range
    per = {}
    for h in (4, 10, hops - 1):
        per[h], out = wall(lambda: nafs_ensemble_features(adj, x0, h, r_list=r_list, method="mean"))
        del out
    per_hop = per[hops - 1] / (hops - 1)
    total = sum(per_hop * h for h in range(hops))
    print(f"per-count      measured ms {per}; ~{per_hop:.1f} ms per hop count unit -> range({hops}) ~ {total:8.1f} ms "
          f"({len(r_list) * sum(range(hops))} SpMMs)", flush=True)


if __name__ == "__main__":
    main()
