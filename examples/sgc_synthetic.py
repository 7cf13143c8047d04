#!/usr/bin/env python3
"""Quick start, the MI355X counterpart of the reference's examples/sgc_pubmed.py (BASELINE config 1):
SGC(prop_steps=3) on a Pubmed-sized graph.  There is no dataset download on the GPU box, so the graph, features and
labels are synthetic (planted communities so that there is something to learn); everything after that is the
reference's flow: model.preprocess(adj, x) once, then train the logistic-regression head on row mini-batches.

    python examples/sgc_synthetic.py [--nodes 19717 --feat 500 --classes 3 --epochs 50]"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgl_amd.models.homo import SGC  # noqa: E402


def planted_graph(n, classes, avg_deg, p_in, seed):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, classes, n)
    m = n * avg_deg // 2
    a = rng.integers(0, n, m)
    same = rng.random(m) < p_in
    b = np.where(same, rng.permutation(n)[rng.integers(0, n, m)], rng.integers(0, n, m))
    # force intra-class partners for the `same` edges
    order = np.argsort(y, kind="stable")
    starts = np.searchsorted(y[order], np.arange(classes))
    counts = np.bincount(y, minlength=classes)
    pick = starts[y[a]] + (rng.integers(0, 1 << 30, m) % np.maximum(counts[y[a]], 1))
    b = np.where(same, order[pick], b)
    keep = a != b
    adj = sp.coo_matrix((np.ones(keep.sum(), np.float32), (a[keep], b[keep])), shape=(n, n)).tocsr()
    adj = ((adj + adj.T) > 0).astype(np.float32).tocsr()
    return adj, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=19717)
    ap.add_argument("--feat", type=int, default=500)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--prop-steps", type=int, default=3)
    a = ap.parse_args()
    device = torch.device("cuda")
    adj, y = planted_graph(a.nodes, a.classes, 5, 0.8, seed=0)
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((a.nodes, a.feat)) + 0.35 * np.eye(a.classes)[y] @ rng.standard_normal((a.classes, a.feat))).astype(np.float32)
    idx = rng.permutation(a.nodes)
    train_idx, test_idx = idx[: a.nodes // 5], idx[a.nodes // 5:]
    labels = torch.from_numpy(y).to(device)

    model = SGC(prop_steps=a.prop_steps, feat_dim=a.feat, output_dim=a.classes).to(device)
    t0 = time.time()
    model.preprocess(adj, x)                      # normalise + k SpMMs + aggregation: HIP kernels, resident in HBM
    torch.cuda.synchronize()
    print(f"Preprocessing done in {time.time() - t0:.4f}s")
    opt = torch.optim.Adam(model.parameters(), lr=0.1, weight_decay=5e-5)
    for epoch in range(a.epochs):
        model.train()
        opt.zero_grad()
        loss = F.cross_entropy(model.model_forward(train_idx, device), labels[train_idx])
        loss.backward()
        opt.step()
    model.eval()
    with torch.no_grad():
        acc = (model.model_forward(test_idx, device).argmax(1) == labels[test_idx]).float().mean().item()
    print(f"epochs {a.epochs}  train loss {loss.item():.4f}  test acc {acc:.4f}")
    assert acc > 1.5 / a.classes, "SGC failed to learn the planted communities"


if __name__ == "__main__":
    main()
