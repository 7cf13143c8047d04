"""Seeded synthetic graphs with ogbn-like degree laws: there are no dataset files and no
network on the GPU box, so bench.py and the full-size tests generate their inputs.

chung_lu(N, m, d_max): expected-degree (Chung-Lu) graph with log-normal weights (sigma = 1.2) clipped to
[1, d_max] and rescaled to sum 2m; endpoints drawn proportionally to the weights; self loops dropped, duplicates
removed, symmetrised.  Returned as canonical CSR (sorted columns) with unit weights (or `weight`)."""
import numpy as np
import torch

WORKLOADS = {
    # name: N, undirected edges m, d_max, feature dim d, prop_steps K
    "S0_pubmed": dict(n=19_717, m=44_324, d_max=171, d=500, k=3),
    "S1_products": dict(n=2_449_029, m=61_859_140, d_max=17_481, d=100, k=3),
    "S1_small": dict(n=200_000, m=5_000_000, d_max=5_000, d=100, k=3),
    "S2_gamlp": dict(n=2_449_029, m=61_859_140, d_max=17_481, d=147, k=5),
}


def chung_lu_numpy(n, m, d_max, seed=0, weight=1.0, sigma=1.2):
    """host version for small graphs (tests); returns (indptr int64, indices int32, data float32)"""
    rng = np.random.default_rng(seed)
    w = np.clip(rng.lognormal(0.0, sigma, n), None, None)
    w = np.clip(w / w.sum() * 2 * m, 1.0, d_max)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    draw = int(m * 1.05) + 16
    a = np.searchsorted(cdf, rng.random(draw)).clip(0, n - 1).astype(np.int64)
    b = np.searchsorted(cdf, rng.random(draw)).clip(0, n - 1).astype(np.int64)
    keep = a != b
    lo, hi = np.minimum(a, b)[keep], np.maximum(a, b)[keep]
    keys = np.unique(lo * n + hi)
    if len(keys) > m:
        keys = np.sort(rng.permutation(keys)[:m])
    lo, hi = keys // n, keys % n
    rows = np.concatenate([lo, hi])
    cols = np.concatenate([hi, lo])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
    return indptr, cols.astype(np.int32), np.full(len(cols), weight, dtype=np.float32)


def chung_lu_torch(n, m, d_max, seed=0, weight=1.0, sigma=1.2, device="cuda"):
    """device version for the full-size workloads; returns CUDA tensors (rowptr int64, col int32, val float32)"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w = torch.exp(torch.randn(n, generator=g, device=device, dtype=torch.float64) * sigma)
    w = torch.clamp(w / w.sum() * (2 * m), 1.0, float(d_max))
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    keys = torch.empty(0, dtype=torch.int64, device=device)
    need = m
    for _ in range(8):
        draw = int(need * 1.05) + 1024
        a = torch.searchsorted(cdf, torch.rand(draw, generator=g, device=device, dtype=torch.float64)).clamp_(0, n - 1)
        b = torch.searchsorted(cdf, torch.rand(draw, generator=g, device=device, dtype=torch.float64)).clamp_(0, n - 1)
        keep = a != b
        lo, hi = torch.minimum(a, b)[keep], torch.maximum(a, b)[keep]
        keys = torch.unique(torch.cat([keys, lo * n + hi]))
        del a, b, keep, lo, hi
        if keys.numel() >= m:
            break
        need = m - keys.numel()
    if keys.numel() > m:
        sel = torch.randperm(keys.numel(), generator=g, device=device)[:m]
        keys = torch.sort(keys[sel]).values
        del sel
    lo, hi = keys // n, keys % n
    del keys
    full = torch.cat([lo * n + hi, hi * n + lo])
    del lo, hi
    full = torch.sort(full).values
    rows = full // n
    col = (full % n).to(torch.int32)
    del full
    counts = torch.bincount(rows, minlength=n)
    del rows
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(counts, 0)
    val = torch.full((col.numel(),), float(weight), dtype=torch.float32, device=device)
    return rowptr, col, val


def features_torch(n, d, seed=0, device="cuda", kind="normal"):
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    if kind == "normal":
        return torch.randn((n, d), generator=g, device=device, dtype=torch.float32)
    # pubmed-like: 90 % zeros, rows normalised to sum 1 (dataset/planetoid.py:40-47)
    x = torch.rand((n, d), generator=g, device=device, dtype=torch.float32)
    x = x * (torch.rand((n, d), generator=g, device=device) < 0.1)
    return x / x.sum(1, keepdim=True).clamp_min(1e-12)
