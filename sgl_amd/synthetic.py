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
    "T_small": dict(n=20_000, m=150_000, d_max=800, d=100, k=3),     # launch rehearsals (N ranks on one GPU), seconds
    "S2_gamlp": dict(n=2_449_029, m=61_859_140, d_max=17_481, d=147, k=5),
    "S2_small": dict(n=200_000, m=5_000_000, d_max=5_000, d=147, k=5),
    # the S1 degree law with communities (degree-corrected planted partition, ids SHUFFLED: what a real co-purchase dump such as
    # ogbn-products looks like, dataset/ogbn.py:45-53 loads its ids as they come): the workload on which a locality ordering
    # (GraphOp(reorder=...)) has something to find.  The Chung-Lu workloads above have no communities by construction.
    "S1_community": dict(n=2_449_029, m=61_859_140, d_max=17_481, d=100, k=3, community=dict(block=4096, p_in=0.8)),
    "T_community": dict(n=40_000, m=400_000, d_max=800, d=100, k=3, community=dict(block=512, p_in=0.8)),
    # ogbn-papers100M-shaped: directed hash-generated graph, rows generated per shard on device
    # (`hashed=True`: sgl_synth_*, mirrored on the host below).  mean_deg 30.07 -> nnz ~ 3.34 G over all rows.
    # S3_papers_shard = ONE rank's share of the 8-GPU job on one GPU: rows [0, N/8) against the full 111 M x 128 replica.
    "S3_papers_shard": dict(n=111_059_956, d=128, k=1, hashed=True, mean_deg=30.07, d_max=20_000, row_block=(0, 8)),
    "S3_papers": dict(n=111_059_956, d=128, k=10, hashed=True, mean_deg=30.07, d_max=20_000),
    "S3_small": dict(n=400_000, d=128, k=3, hashed=True, mean_deg=30.07, d_max=2_000),
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


def planted_partition_torch(n, m, d_max, block, p_in=0.8, seed=0, weight=1.0, sigma=1.2, device="cuda", shuffle=True):
    """Degree-corrected planted partition with chung_lu's degree law: node i has the log-normal weight w_i of chung_lu_torch
    (same clipping and rescaling), communities are runs of `block` consecutive nodes of the GENERATOR's numbering, an edge's
    first endpoint is drawn proportionally to w and its partner proportionally to w INSIDE the first endpoint's community with
    probability p_in, over the whole graph otherwise.  Self loops dropped, duplicates removed, symmetrised; finally (shuffle)
    the node ids are permuted at random, so the ids carry no trace of the communities.  Returns canonical CSR tensors
    (rowptr int64, col int32, val float32) and `truth` (int64 [n]: the community of every node under the returned ids)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w = torch.exp(torch.randn(n, generator=g, device=device, dtype=torch.float64) * sigma)
    w = torch.clamp(w / w.sum() * (2 * m), 1.0, float(d_max))
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    cdf0 = torch.cat([torch.zeros(1, dtype=cdf.dtype, device=cdf.device), cdf])      # cdf0[i] = mass of nodes < i
    n_blocks = (n + block - 1) // block
    keys = torch.empty(0, dtype=torch.int64, device=device)
    need = m
    for _ in range(8):
        draw = int(need * 1.05) + 1024
        a = torch.searchsorted(cdf, torch.rand(draw, generator=g, device=device, dtype=torch.float64)).clamp_(0, n - 1)
        u = torch.rand(draw, generator=g, device=device, dtype=torch.float64)
        inside = torch.rand(draw, generator=g, device=device) < p_in
        blk = a // block
        lo_m = cdf0[blk * block]
        hi_m = cdf0[torch.clamp((blk + 1) * block, max=n)]
        target = torch.where(inside, lo_m + u * (hi_m - lo_m), u)
        b = torch.searchsorted(cdf, target).clamp_(0, n - 1)
        b = torch.where(inside, torch.minimum(torch.maximum(b, blk * block), torch.clamp((blk + 1) * block, max=n) - 1), b)
        del u, inside, blk, lo_m, hi_m, target
        keep = a != b
        lo, hi = torch.minimum(a, b)[keep], torch.maximum(a, b)[keep]
        keys = torch.unique(torch.cat([keys, lo * n + hi]))
        del a, b, keep, lo, hi
        if keys.numel() >= m:
            break
        need = m - keys.numel()
    if keys.numel() > m:
        sel = torch.randperm(keys.numel(), generator=g, device=device)[:m]
        keys = keys[sel]
        del sel
    lo, hi = keys // n, keys % n
    del keys
    truth = torch.arange(n, device=device, dtype=torch.int64) // block
    if shuffle:
        new_id = torch.randperm(n, generator=g, device=device)           # new id of generator node i
        lo, hi = new_id[lo], new_id[hi]
        t2 = torch.empty_like(truth)
        t2[new_id] = truth
        truth = t2
        del new_id
    full = torch.cat([lo * n + hi, hi * n + lo])
    del lo, hi
    full = torch.sort(full).values
    rows = full // n
    col = (full % n).to(torch.int32)
    del full
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    del rows
    val = torch.full((col.numel(),), float(weight), dtype=torch.float32, device=device)
    assert n_blocks >= 1
    return rowptr, col, val, truth


def features_torch(n, d, seed=0, device="cuda", kind="normal"):
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    if kind == "normal":
        return torch.randn((n, d), generator=g, device=device, dtype=torch.float32)
    # pubmed-like: 90 % zeros, rows normalised to sum 1 (dataset/planetoid.py:40-47)
    x = torch.rand((n, d), generator=g, device=device, dtype=torch.float32)
    x = x * (torch.rand((n, d), generator=g, device=device) < 0.1)
    return x / x.sum(1, keepdim=True).clamp_min(1e-12)


# ---- hash-keyed generator (device: csrc/sgl_synth.hip; host mirror here) -----------------------------------------------------
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_RK = np.uint64(0xD6E8FEB86659FD93)


def _u64(a):
    return np.asarray(a).astype(np.uint64)


def _mix64(z):
    with np.errstate(over="ignore"):
        z = _u64(z) + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _hash4(seed, stream, a, b):
    with np.errstate(over="ignore"):
        return _mix64(_mix64(_mix64(np.uint64(seed) * _GOLD + np.uint64(stream)) ^ _u64(a)) + _u64(b))


def _mulhi64(a, b):
    """high 64 bits of the 128-bit product (numpy has no 128-bit integers: 32-bit limbs)"""
    a, b = _u64(a), _u64(b)
    lo32 = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    a0, a1, b0, b1 = a & lo32, a >> s32, b & lo32, b >> s32
    with np.errstate(over="ignore"):
        p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = (p00 >> s32) + (p01 & lo32) + (p10 & lo32)
        return p11 + (p01 >> s32) + (p10 >> s32) + (mid >> s32)


def _feistel_half_bits(n):
    bits = 2
    while bits < 62 and (1 << bits) < n:
        bits += 1
    return (bits + 1) // 2


def _permute_id(x, n, seed):
    half = np.uint64(_feistel_half_bits(n))
    mask = np.uint64((1 << int(half)) - 1)
    key = _mix64(np.uint64(seed) ^ np.uint64(0xA5A5A5A5A5A5A5A5))
    x = _u64(x).copy()
    todo = np.ones(x.shape, dtype=bool)
    first = True
    while todo.any():
        cur = x[todo]
        l, r = cur >> half, cur & mask
        for rnd in range(4):
            with np.errstate(over="ignore"):
                f = _mix64(r + key + np.uint64(rnd) * _RK) & mask
            l, r = r, l ^ f
        cur = (l << half) | r
        x[todo] = cur
        nt = np.zeros_like(todo)
        nt[todo] = cur >= np.uint64(n)
        todo = nt
        first = False
    return x


def degree_table(mean_deg, d_max, sigma=1.1):
    """2 x 4096 quantiles of a log-normal degree law clipped to [1, d_max]: the body, and a refinement of its top bucket
    (the extreme tail, up to d_max: rows long enough to be split into pieces).  The mean over the law is (about)
    mean_deg.  The SAME array goes to the device generator and to the host mirror."""
    from scipy.special import ndtri
    qb = (np.arange(4096) + 0.5) / 4096
    qt = 1.0 - (1.0 - qb) / 4096.0
    zb, zt = ndtri(qb), ndtri(qt)

    def tables(mu):
        return (np.clip(np.rint(np.exp(mu + sigma * zb)), 1, d_max), np.clip(np.rint(np.exp(mu + sigma * zt)), 1, d_max))

    lo, hi = -5.0, 12.0
    for _ in range(60):                       # bisection on mu so that the clipped, rounded law has the wanted mean
        mu = 0.5 * (lo + hi)
        b, t = tables(mu)
        mean = (b[:-1].sum() + t.mean()) / 4096.0
        lo, hi = (mu, hi) if mean < mean_deg else (lo, mu)
    b, t = tables(0.5 * (lo + hi))
    return np.concatenate([b, t]).astype(np.int32)


def hashed_degrees_numpy(seed, rows, table):
    h = _hash4(seed, 0, rows, 0)
    t = (h >> np.uint64(52)).astype(np.int64)
    tail = 4096 + ((h >> np.uint64(40)) & np.uint64(4095)).astype(np.int64)
    return table[np.where(t == 4095, tail, t)].astype(np.int64)


def hashed_rows_numpy(seed, rows, n_cols, table):
    """host mirror of sgl_synth_degrees + sgl_synth_fill for the given GLOBAL row ids: (indptr int64, col int32, val f32)"""
    rows = np.asarray(rows, dtype=np.int64)
    deg = hashed_degrees_numpy(seed, rows, table)
    indptr = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    rr = np.repeat(rows, deg)
    jj = np.arange(indptr[-1], dtype=np.int64) - np.repeat(indptr[:-1], deg)
    u = _hash4(seed, 1, rr, jj)
    skew = _mulhi64(_mulhi64(u, u), np.uint64(n_cols))
    col = _permute_id(skew, n_cols, seed).astype(np.int32)
    val = ((_hash4(seed, 2, rr, jj) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -29)).astype(np.float32)
    return indptr, col, val


def hashed_features_numpy(seed, rows, d):
    rows = np.asarray(rows, dtype=np.int64)
    h = _hash4(seed, 3, rows[:, None], np.arange(d, dtype=np.int64)[None, :])
    q = (h >> np.uint64(40)).astype(np.int64) - (1 << 23)
    return (q.astype(np.float32) * np.float32(2.0 ** -23)).astype(np.float32)


def hashed_block_torch(seed, row0, n_rows, n_cols, table, device="cuda"):
    """rows [row0, row0 + n_rows) of the hashed graph generated on `device`: (rowptr int64 local, col int32, val f32)"""
    import ctypes
    from . import _lib
    _lib.require_gpu()
    tab = torch.from_numpy(np.ascontiguousarray(table, dtype=np.int32)).to(device)
    deg = torch.empty(n_rows, dtype=torch.int64, device=device)
    st = _lib.current_stream_ptr()
    with torch.cuda.device(tab.device):
        _lib.check_probe(_lib.probe_lib().sgl_synth_degrees(ctypes.c_uint64(seed), row0, n_rows, _lib.ptr(tab), _lib.ptr(deg), st), "sgl_synth_degrees")
        rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=device)
        torch.cumsum(deg, 0, out=rowptr[1:])
        del deg
        nnz = int(rowptr[-1])
        col = torch.empty(nnz, dtype=torch.int32, device=device)
        val = torch.empty(nnz, dtype=torch.float32, device=device)
        _lib.check_probe(_lib.probe_lib().sgl_synth_fill(ctypes.c_uint64(seed), row0, n_rows, n_cols, _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val),
                                             st), "sgl_synth_fill")
    return rowptr, col, val


def hashed_features_torch(seed, row0, n_rows, d, device="cuda", out=None):
    """X[row0 : row0 + n_rows, :d] of the hashed feature matrix into a fresh (or the given, possibly row-padded) buffer"""
    import ctypes
    from . import _lib
    from . import device as dev
    _lib.require_gpu()
    if out is None:
        out = dev.alloc_rows(n_rows, d, device, zero_pad=False)
    parent = dev.padded_parent(out)
    with torch.cuda.device(parent.device):
        _lib.check_probe(_lib.probe_lib().sgl_synth_features(ctypes.c_uint64(seed), row0, n_rows, d, parent.stride(0) if n_rows > 1 else parent.shape[1],
                                                 _lib.ptr(parent), _lib.current_stream_ptr()), "sgl_synth_features")
    return out
