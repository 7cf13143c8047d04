#!/usr/bin/env python3
"""The NAFS clustering task's adaptive k-hop selection (NodeClusteringNAFS: hops = range(K), six r values, KMeans per hop count;
sgl/tasks/node_clustering.py:124-258) with ONE propagation per r.

The reference calls `_k_hop_cluster(hop)` for every hop count, each re-normalising and re-propagating from X_0 -- 6 x (0 + 1 + ...
+ K-1) SpMMs and an O(N x hops) Python loop per call.  Here `nafs_ensemble_sweep` propagates K-1 steps once per r and one kernel
emits the smoothed features of EVERY hop count while it streams the hop matrices (sgl_nafs_prefix_f32), combined into the
multi-r ensemble in the same pass; KMeans consumes them one at a time.

    python examples/nafs_hop_sweep.py [--nodes 20000] [--hops 8]
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgl_amd.tricks import nafs_ensemble_sweep  # noqa: E402


def planted_graph(n, classes, deg, p_in, d, seed):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, classes, n)
    members = [np.flatnonzero(y == c) for c in range(classes)]
    a = np.repeat(np.arange(n), deg)
    same = np.array([members[c][rng.integers(0, len(members[c]), (y == c).sum() * deg)] for c in range(classes)], dtype=object)
    b = rng.integers(0, n, a.size)
    pick = rng.random(a.size) < p_in
    for c in range(classes):
        m = (y[a] == c) & pick
        b[m] = same[c][: m.sum()]
    adj = sp.coo_matrix((np.ones(a.size, np.float32), (a, b)), shape=(n, n)).tocsr()
    adj = ((adj + adj.T) > 0).astype(np.float32).tocsr()
    x = rng.standard_normal((n, d)).astype(np.float32) + 1.0 * np.eye(classes, d, dtype=np.float32)[y]      # weak class signal
    return adj, x, y


def purity(y, pred, classes):
    return sum(np.bincount(y[pred == k], minlength=classes).max() for k in range(classes)) / len(y)


def main():
    from sklearn.cluster import KMeans
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=20_000)
    ap.add_argument("--hops", type=int, default=8)
    ap.add_argument("--classes", type=int, default=5)
    a = ap.parse_args()
    adj, x, y = planted_graph(a.nodes, a.classes, 10, 0.8, 32, seed=0)
    scores = {}

    def consume(hop, feats):                                   # called in hop order once the ensemble over r is complete
        pred = KMeans(n_clusters=a.classes, n_init=3, random_state=0).fit_predict(feats.cpu().numpy())
        scores[hop] = purity(y, pred, a.classes)
        return scores[hop]

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nafs_ensemble_sweep(adj, x, a.hops, r_list=(0.5, 0.4, 0.3, 0.2, 0.1, 0), method="mean", consume=consume)
    t = time.perf_counter() - t0
    best = max(scores, key=scores.get)
    print(f"NAFS hop sweep: N={a.nodes} hops=range({a.hops}) x 6 r: {6 * (a.hops - 1)} SpMMs instead of {6 * sum(range(a.hops))}; "
          f"{t:.2f} s incl. KMeans")
    print("  purity per hop count: " + " ".join(f"{h}:{s:.3f}" for h, s in sorted(scores.items())) + f"  -> best hop count {best}")
    assert sorted(scores) == list(range(a.hops)) and scores[best] > scores[0] + 0.1          # smoothing helps on a graph with communities


if __name__ == "__main__":
    main()
