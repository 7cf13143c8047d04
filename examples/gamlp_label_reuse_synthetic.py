#!/usr/bin/env python3
"""The MI355X counterpart of the reference's examples/gamlp_products.py (BASELINE config 3): GAMLP with label use /
label reuse on an ogbn-products-shaped graph.  The reference's loop
(sgl/tasks/node_classification_with_label_use.py:58-137) re-runs model.preprocess(adj, [features || labels]) every
epoch and again for each label-reuse iteration -- `epochs x (1 + label_iters) x prop_steps` SpMMs over a
[N, d + C] = [2 449 029, 147] matrix -- which is where pre-propagation dominates the wall clock.  Here the adjacency is
built and normalised on the device once (cached), the feature matrix lives in HBM, and each preprocess() is
prop_steps HIP SpMMs.  Synthetic data (no dataset files on the GPU box).

    python examples/gamlp_label_reuse_synthetic.py [--workload S1_small] [--epochs 3] [--label-iters 2]"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgl_amd import synthetic  # noqa: E402
from sgl_amd.io import DeviceAdjacency  # noqa: E402
from sgl_amd.models.homo import GAMLP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S1_products")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--label-iters", type=int, default=2)
    ap.add_argument("--prop-steps", type=int, default=5)
    ap.add_argument("--classes", type=int, default=47)
    ap.add_argument("--batch", type=int, default=50_000)
    a = ap.parse_args()
    device = torch.device("cuda")
    wl = synthetic.WORKLOADS[a.workload]
    n, d, C = wl["n"], 100, a.classes
    rowptr, col, val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    adj = DeviceAdjacency(rowptr, col, val, (n, n))
    g = torch.Generator(device=device).manual_seed(0)
    y = torch.randint(0, C, (n,), generator=g, device=device)
    x = torch.randn((n, d), generator=g, device=device) + 0.5 * F.one_hot(y, C).float() @ torch.randn((C, d), generator=g, device=device)
    perm = torch.randperm(n, generator=g, device=device)
    train_idx, rest = perm[: n // 12], perm[n // 12:]

    model = GAMLP(a.prop_steps, d + C, C, 256, 3).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    from sgl_amd.tricks import add_labels, label_reuse
    feats = None
    t_prep = t_train = 0.0
    n_prep = 0
    calls = []
    real_pre = model.preprocess

    def timed_pre(adj_, f_):                                   # every preprocess() of the loop, timed: prop_steps SpMMs over [N, d + C]
        nonlocal t_prep, n_prep
        torch.cuda.synchronize(); t0 = time.time()
        real_pre(adj_, f_)
        torch.cuda.synchronize(); t_prep += time.time() - t0; n_prep += 1
        calls.append((time.time() - t0) * 1e3)
    model.preprocess = timed_pre
    epoch_ms = []
    for epoch in range(a.epochs):
        torch.cuda.synchronize(); t_e = time.time()
        # label use: one-hot labels of a random half of the training nodes as extra input columns (tasks/utils.py:33-36)
        keep = torch.rand(train_idx.numel(), generator=g, device=device) < 0.5
        feats = add_labels(x, y, train_idx[keep], C, out=feats, device=device)
        model.preprocess(adj, feats)
        # label reuse: predictions written back into the label columns of everything else, preprocess again (x label_iters)
        model.eval()
        label_reuse(model, adj, feats, torch.cat([train_idx[~keep], rest]), C, a.label_iters, device=device, batch_size=4 * a.batch)
        torch.cuda.synchronize(); epoch_ms.append((time.time() - t_e) * 1e3)
        torch.cuda.synchronize(); t0 = time.time()
        model.train()
        for s in range(0, train_idx.numel(), a.batch):
            b = train_idx[s:s + a.batch]
            opt.zero_grad()
            loss = F.cross_entropy(model.model_forward(b, device), y[b])
            loss.backward()
            opt.step()
        torch.cuda.synchronize(); t_train += time.time() - t0
        print(f"epoch {epoch}: loss {loss.item():.4f}")
    print("preprocess() calls, ms: " + " ".join(f"{c:.1f}" for c in calls))
    print("label use + reuse per epoch (add_labels, (1 + label_iters) x preprocess, label_iters x full prediction + write-back), ms: "
          + " ".join(f"{c:.1f}" for c in epoch_ms))
    di = getattr(model._pre_graph_op, "delta_info", None)
    what = "every column propagated in every call" if di is None else \
        (f"calls after the first re-propagate columns {di['columns_propagated'][0]}..{di['columns_propagated'][1]} only "
         f"(config.delta_propagate: the others are copied from the previous hop matrices)")
    print(f"{n_prep} preprocess() calls, {a.prop_steps} hops each over [N={n}, {d + C}]: {t_prep / n_prep * 1e3:.1f} ms per call incl. first-call "
          f"normalisation; {what}; mini-batch training {t_train:.2f}s")


if __name__ == "__main__":
    main()
