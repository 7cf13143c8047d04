"""Parity of the HIP path (through the C ABI of libsgl_hip.so) against the CPU oracle and the committed golden
vectors recorded from the reference.  Every test here needs a real MI355X: run with `-m gpu`.

Bars: integer / index work (CSR structure, plans) bit-exact; the SpMM in strict-order mode bit-exact (same fmaf
chain as csrc/matmul.c:23-40); everything else within the SURVEY 8(c) tolerance (1e-5, three-way criterion)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import oracle
from inputs import hash_matrix
from sgl_amd import _lib
from sgl_amd import device as dev

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def cuda():
    from sgl_amd import _lib
    _lib.require_gpu()
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


def norm_graph(goldens, name, r=0.5, alpha=None):
    g = goldens.graph(name)
    n = g.shape[0]
    ptr, col, val = oracle.sym_norm_csr(g.indptr, g.indices, g.data, n, r, alpha)
    return n, ptr, col, val.astype(np.float32)


def device_csr(ptr, col, val, shape, dev, **kw):
    from sgl_amd.device import DeviceCSR
    return DeviceCSR(torch.from_numpy(np.asarray(ptr, np.int64)).to(dev), torch.from_numpy(np.asarray(col, np.int32)).to(dev),
                     torch.from_numpy(np.asarray(val, np.float32)).to(dev), shape, **kw)


def long_row_graph(n=1500, seed=5):
    """power-law rows plus three huge rows and some empty ones (canonical CSR, not symmetric)"""
    rng = np.random.default_rng(seed)
    deg = np.minimum(rng.lognormal(1.2, 1.0, n).astype(np.int64), 200)
    deg[rng.integers(0, n, 60)] = 0
    deg[[3, 700, n - 1]] = [1400, 900, 1499]
    rows, cols = [], []
    for i in range(n):
        c = np.sort(rng.choice(n, int(deg[i]), replace=False))
        rows.append(np.full(len(c), i))
        cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = rng.uniform(-1, 1, len(rows)).astype(np.float32)
    a = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    a.sort_indices()
    return a


D_LIST = [1, 2, 3, 4, 7, 8, 16, 32, 47, 64, 100, 128, 147, 256, 500]


@pytest.mark.parametrize("gname", ["pl2000", "dir40", "sym64"])
def test_spmm_strict_is_bit_exact(goldens, cuda, gname):
    from sgl_amd.device import alloc_rows, padded_parent
    n, ptr, col, val = norm_graph(goldens, gname)
    csr = device_csr(ptr, col, val, (n, n), cuda, strict=True)
    bad = []
    for d in D_LIST:
        x = hash_matrix(n, d, seed=d)
        ref = oracle.oracle_spmm(ptr, col, val, x)
        # (a) contiguous [n, d]: exercises the 4-/8-/16-byte lane paths depending on d
        y = csr.spmm(torch.from_numpy(x).to(cuda)).cpu().numpy()
        if not np.array_equal(y, ref):
            bad.append(("contig", d, oracle.parity_report(y, ref)))
        # (b) row-padded buffer (the layout GraphOp.propagate uses): always 16-byte lanes
        xp = alloc_rows(n, d, cuda)
        xp.copy_(torch.from_numpy(x))
        yp = csr.spmm(padded_parent(xp))[:, :d].cpu().numpy()
        if not np.array_equal(yp, ref):
            bad.append(("padded", d, oracle.parity_report(yp, ref)))
    assert not bad, bad


@pytest.mark.parametrize("gname", ["pl2000", "dir40"])
def test_spmm_fast_within_tolerance(goldens, cuda, gname):
    n, ptr, col, val = norm_graph(goldens, gname)
    csr = device_csr(ptr, col, val, (n, n), cuda, strict=False)
    exact = 0
    for d in D_LIST:
        x = hash_matrix(n, d, seed=d + 1)
        ref = oracle.oracle_spmm(ptr, col, val, x)
        scale = oracle.oracle_spmm(ptr, col, np.abs(val), np.abs(x))      # |A| . |X|: see oracle.parity_report
        y = csr.spmm(torch.from_numpy(x).to(cuda)).cpu().numpy()
        rep = oracle.parity_report(y, ref, TOL, scale=scale)
        assert rep["ok"], (d, rep)
        exact += rep["bit_equal"]
    print(f"fast mode: {exact}/{len(D_LIST)} widths bit-equal on {gname}")


@pytest.mark.parametrize("strict", [True, False])
def test_spmm_long_rows_empty_rows_and_splitting(cuda, strict):
    a = long_row_graph()
    n = a.shape[0]
    for d in (4, 100, 128, 37):
        x = hash_matrix(n, d, seed=11)
        ref = oracle.oracle_spmm(a.indptr, a.indices, a.data, x)
        scale = oracle.oracle_spmm(a.indptr, a.indices, np.abs(a.data), np.abs(x))
        for item_nnz, long_nnz in ((512, 2048), (64, 256), (8, 100), (100000, 64)):
            csr = device_csr(a.indptr, a.indices, a.data, (n, n), cuda, strict=strict, item_nnz=item_nnz, long_row_nnz=long_nnz)
            info = csr.info()
            if strict:
                assert info["n_pieces"] == 0
            elif long_nnz < 1400:
                assert info["n_long_rows"] >= 3 and info["n_pieces"] >= 3
            y = csr.spmm(torch.from_numpy(x).to(cuda)).cpu().numpy()
            if strict:
                assert np.array_equal(y, ref), (d, item_nnz, long_nnz, oracle.parity_report(y, ref))
            else:
                rep = oracle.parity_report(y, ref, TOL, scale=scale)
                assert rep["ok"], (d, item_nnz, long_nnz, rep)
            # deterministic: same bits on a second run (no float atomics)
            y2 = csr.spmm(torch.from_numpy(x).to(cuda)).cpu().numpy()
            assert np.array_equal(y, y2)


def test_spmm_accumulate_and_overwrite_semantics(goldens, cuda):
    n, ptr, col, val = norm_graph(goldens, "pl256")
    for strict in (True, False):
        csr = device_csr(ptr, col, val, (n, n), cuda, strict=strict, long_row_nnz=16)
        for d in (16, 100, 5):
            x = hash_matrix(n, d, seed=3)
            y0 = hash_matrix(n, d, seed=4)
            ref_acc = oracle.oracle_spmm(ptr, col, val, x, out=y0.copy())       # matmul.c:37 semantics
            ref_ovw = oracle.oracle_spmm(ptr, col, val, x)                      # cudamatmul.c:46 (beta = 0)
            y = torch.from_numpy(y0.copy()).to(cuda)
            csr.spmm(torch.from_numpy(x).to(cuda), out=y, accumulate=True)
            z = torch.from_numpy(y0.copy()).to(cuda)
            csr.spmm(torch.from_numpy(x).to(cuda), out=z, accumulate=False)
            if strict:
                assert np.array_equal(y.cpu().numpy(), ref_acc)
                assert np.array_equal(z.cpu().numpy(), ref_ovw)
            else:
                assert oracle.parity_ok(y.cpu().numpy(), ref_acc, TOL)
                assert oracle.parity_ok(z.cpu().numpy(), ref_ovw, TOL)


def test_nan_inf_stay_in_the_rows_that_reference_them(goldens, cuda):
    """a non-finite feature row must poison exactly the output rows whose adjacency references it -- no leakage through
    masked lanes, padded slots or the packed (R > 1) layouts (the reference never multiplies what it does not read)"""
    n, ptr, col, val = norm_graph(goldens, "pl2000")
    a = sp.csr_matrix((val, col, ptr), shape=(n, n))
    bad_node = int(np.argmax(np.diff(ptr)))                         # a hub: many rows reference it
    touched = np.zeros(n, bool)
    touched[a[:, bad_node].nonzero()[0]] = True
    for d in (1, 5, 16, 47, 100, 147, 500):
        for strict in (True, False):
            x = hash_matrix(n, d, seed=d)
            x[bad_node, :] = np.nan
            x[bad_node, 0] = np.inf
            csr = device_csr(ptr, col, val, (n, n), cuda, strict=strict, long_row_nnz=64)
            y = csr.spmm(torch.from_numpy(x).to(cuda)).cpu().numpy()
            nonfinite_rows = ~np.isfinite(y).all(axis=1)
            assert np.array_equal(nonfinite_rows, touched), (d, strict, nonfinite_rows.sum(), touched.sum())
            clean = oracle.oracle_spmm(ptr, col, val, np.nan_to_num(x, nan=0.0, posinf=0.0))
            assert oracle.parity_ok(y[~touched], clean[~touched], TOL, rowwise=False)


def test_spmm_on_a_side_stream(goldens, cuda):
    """every entry point is stream-ordered on the caller's stream: results are correct when launched on a non-default
    torch stream with the producer of X on that same stream"""
    n, ptr, col, val = norm_graph(goldens, "pl2000")
    csr = device_csr(ptr, col, val, (n, n), cuda, strict=True)
    xh = hash_matrix(n, 64, seed=3)
    ref = oracle.oracle_spmm(ptr, col, val, oracle.oracle_spmm(ptr, col, val, xh * np.float32(2.0)))
    side = torch.cuda.Stream()
    xd = torch.from_numpy(xh).to(cuda)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        x2 = xd * 2.0                      # produced on the side stream ...
        y = csr.spmm(csr.spmm(x2))         # ... and consumed there, twice, without any host sync in between
    side.synchronize()
    assert np.array_equal(y.cpu().numpy(), ref)


def test_spmm_fuzz_random_shapes(cuda):
    """60 random rectangular matrices (empty rows / columns, huge rows, random widths, random plan parameters,
    both layouts of X): strict order must be bit-exact, the fast layout within tolerance, accumulate must compose"""
    rng = np.random.default_rng(2024)
    for case in range(60):
        n_rows, n_cols = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
        d = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 31, 32, 33, 64, 65, 100, 128, 200, 257]))
        dens = float(rng.choice([0.0, 0.002, 0.02, 0.2]))
        deg = rng.binomial(n_cols, dens, n_rows)
        if n_rows > 3 and rng.random() < 0.5:
            deg[rng.integers(0, n_rows, 2)] = min(n_cols, int(rng.integers(200, 1500)))
        rows = np.repeat(np.arange(n_rows), deg)
        cols = np.concatenate([np.sort(rng.choice(n_cols, k, replace=False)) for k in deg]) if deg.sum() else np.zeros(0, np.int64)
        vals = rng.standard_normal(len(rows)).astype(np.float32)
        ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        x = rng.standard_normal((n_cols, d)).astype(np.float32)
        ref = oracle.oracle_spmm(ptr, cols, vals, x, n_rows=n_rows)
        scale = oracle.oracle_spmm(ptr, cols, np.abs(vals), np.abs(x), n_rows=n_rows)
        item_nnz, long_nnz = int(rng.choice([0, 1, 7, 64, 512])), int(rng.choice([0, 16, 100, 2048]))
        xd = torch.from_numpy(x).to(cuda)
        ctx = (case, n_rows, n_cols, d, dens, item_nnz, long_nnz)
        ys = device_csr(ptr, cols, vals, (n_rows, n_cols), cuda, strict=True, item_nnz=item_nnz).spmm(xd)
        assert np.array_equal(ys.cpu().numpy(), ref), ctx
        fast = device_csr(ptr, cols, vals, (n_rows, n_cols), cuda, item_nnz=item_nnz, long_row_nnz=long_nnz)
        yf = fast.spmm(xd)
        rep = oracle.parity_report(yf.cpu().numpy(), ref, TOL, scale=scale)
        assert rep["ok"], (ctx, rep)
        # accumulate composes: A x + (A x) == 2 A x up to one rounding per element
        y2 = yf.clone()
        fast.spmm(xd, out=y2, accumulate=True)
        assert oracle.parity_ok(y2.cpu().numpy(), 2 * ref, TOL, scale=2 * scale), ctx


def test_hip_graph_replay_of_a_propagation(goldens, cuda):
    """capture_chain: the k-hop launch sequence replayed from a hipGraph gives the same bits, also after the input
    buffer has been refilled in place and on a side stream"""
    n, ptr, col, val = norm_graph(goldens, "pl2000")
    csr = device_csr(ptr, col, val, (n, n), cuda, long_row_nnz=64)
    x = torch.from_numpy(hash_matrix(n, 100, seed=1)).to(cuda)
    outs = [torch.empty((n, 100), device=cuda) for _ in range(3)]
    graph = csr.capture_chain(x, outs)
    ref = [t.clone() for t in csr.spmm_chain(x, 3)]
    for o in outs:
        o.fill_(float("nan"))
    got = graph.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, ref))
    x.copy_(torch.from_numpy(hash_matrix(n, 100, seed=2)).to(cuda))         # new features, same buffers
    ref2 = [t.clone() for t in csr.spmm_chain(x, 3)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph.replay()
    side.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(outs, ref2))
    # the capture bakes the value / row-map pointers in -- once they change the graph refuses to replay (it would
    # silently compute with the old ones) and a fresh capture works
    from sgl_amd._lib import SglHipError
    csr.set_values(csr.val.clone() * 2.0)
    with pytest.raises(SglHipError, match="changed after the capture"):
        graph.replay()
    graph.close()
    graph2 = csr.capture_chain(x, outs)
    got2 = graph2.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b * 2.0 ** (h + 1)) for h, (a, b) in enumerate(zip(got2, ref2)))   # doubling is exact in fp32
    graph2.close()


def test_spmm_multi_writes_every_replica(goldens, cuda):
    """sgl_spmm_multi_f32: the same product lands in up to 8 destination matrices (in a multi-GPU job 7 of them are
    peer replicas; here all are local), for regular rows, split rows and narrow / wide matrices"""
    a = long_row_graph()
    n = a.shape[0]
    for d in (100, 16, 500):
        x = torch.from_numpy(hash_matrix(n, d, seed=9)).to(cuda)
        for strict, long_nnz in ((True, 0), (False, 128)):
            csr = device_csr(a.indptr, a.indices, a.data, (n, n), cuda, strict=strict, long_row_nnz=long_nnz)
            ref = csr.spmm(x)
            for n_out in (1, 3, 8):
                outs = [torch.full((n, d), float("nan"), device=cuda) for _ in range(n_out)]
                csr.spmm_multi(x, [o.data_ptr() for o in outs], d)
                for o in outs:
                    assert torch.equal(o, ref), (d, strict, n_out)
                if n_out == 3:
                    # row mask: destination 1 gets the even rows only, destination 2 the rows divisible by 3
                    rows = torch.arange(n, device=cuda)
                    mask = ((rows % 2 == 0).to(torch.uint8) | ((rows % 3 == 0).to(torch.uint8) << 1)).contiguous()
                    outs = [torch.full((n, d), -7.0, device=cuda) for _ in range(3)]
                    csr.spmm_multi(x, [o.data_ptr() for o in outs], d, row_mask=mask)
                    assert torch.equal(outs[0], ref)
                    for k, sel in ((1, rows % 2 == 0), (2, rows % 3 == 0)):
                        assert torch.equal(outs[k][sel], ref[sel]) and bool((outs[k][~sel] == -7.0).all())
    with pytest.raises(ValueError):
        csr.spmm_multi(x, [], d)


def test_spmm_chain_equals_repeated_spmm(goldens, cuda):
    n, ptr, col, val = norm_graph(goldens, "pl2000")
    csr = device_csr(ptr, col, val, (n, n), cuda, strict=True)
    for d in (100, 7):
        x = torch.from_numpy(hash_matrix(n, d, seed=5)).to(cuda)
        chain = csr.spmm_chain(x, 4)
        cur = x
        for k in range(4):
            cur = csr.spmm(cur)
            assert torch.equal(chain[k], cur)
        pre = [torch.empty((n, d), device=cuda) for _ in range(2)]
        got = csr.spmm_chain(x, 2, outs=pre)
        assert got[1].data_ptr() == pre[1].data_ptr() and torch.equal(pre[1], chain[1])
    assert csr.spmm_chain(x, 0) == []
    rect = device_csr(ptr[:11] - ptr[0], col[:ptr[10]], val[:ptr[10]], (10, n), cuda)
    with pytest.raises(ValueError):
        rect.spmm_chain(x, 2)


def test_spmm_rectangular_shard_and_degenerate_shapes(goldens, cuda):
    n, ptr, col, val = norm_graph(goldens, "pl2000")
    x = hash_matrix(n, 100, seed=8)
    ref = oracle.oracle_spmm(ptr, col, val, x)
    xd = torch.from_numpy(x).to(cuda)
    # a row shard [r0, r1) of the matrix is a rectangular CSR over all n columns (what each rank owns)
    for (r0, r1) in ((0, 700), (700, 1999), (1999, 2000), (500, 500)):
        rp = (ptr[r0:r1 + 1] - ptr[r0]).astype(np.int64)
        nb, ne = int(ptr[r0]), int(ptr[r1])
        csr = device_csr(rp, col[nb:ne], val[nb:ne], (r1 - r0, n), cuda, strict=True)
        y = csr.spmm(xd).cpu().numpy()
        assert y.shape == (r1 - r0, 100) and np.array_equal(y, ref[r0:r1])
    # all-empty matrix -> zeros; d = 0 columns -> empty
    z = device_csr(np.zeros(n + 1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32), (n, n), cuda)
    assert not z.spmm(xd).cpu().numpy().any()
    csr = device_csr(ptr, col, val, (n, n), cuda)
    assert csr.spmm(torch.empty((n, 0), device=cuda)).shape == (n, 0)
    with pytest.raises(ValueError):
        csr.spmm(xd[:10])


def test_spmm_every_tuning_variant(goldens, cuda):
    from sgl_amd import _lib
    n, ptr, col, val = norm_graph(goldens, "pl2000")
    saved = {k: _lib.get_tuning(k) for k in ("spmm_unroll", "spmm_nt", "spmm_group", "spmm_waves", "spmm_xcd_remap")}
    try:
        for d in (100, 16, 500):
            x = hash_matrix(n, d, seed=21)
            ref = oracle.oracle_spmm(ptr, col, val, x)
            xd = torch.from_numpy(x).to(cuda)
            for unroll in (0, 1, 2):
                for nt in (0, 1):
                    for group in (0, 32, 64):
                        for waves in (1, 4):
                            for remap in (0, 1):
                                for k, v in (("spmm_unroll", unroll), ("spmm_nt", nt), ("spmm_group", group),
                                             ("spmm_waves", waves), ("spmm_xcd_remap", remap)):
                                    _lib.set_tuning(k, v)
                                csr = device_csr(ptr, col, val, (n, n), cuda, long_row_nnz=128)
                                rep = oracle.parity_report(csr.spmm(xd).cpu().numpy(), ref, TOL)
                                assert rep["ok"], (d, unroll, nt, group, waves, remap, rep)
    finally:
        for k, v in saved.items():
            _lib.set_tuning(k, v)


def test_reference_signature_shims(goldens, cuda):
    """FloatCSRMulDenseOMP / FloatCSRMulDense bound exactly like sgl/operators/utils.py:10-73 binds them"""
    from sgl_amd.operators.utils import csr_sparse_dense_matmul, cuda_csr_sparse_dense_matmul
    n, ptr, col, val = norm_graph(goldens, "pl2000")
    adj = sp.csr_matrix((val.astype(np.float64), col, ptr.astype(np.int32)), shape=(n, n))
    for d in (37, 100):
        x = hash_matrix(n, d, seed=2)
        ref = oracle.oracle_spmm(ptr, col, val, x)
        assert np.array_equal(csr_sparse_dense_matmul(adj, x), ref)
        assert np.array_equal(cuda_csr_sparse_dense_matmul(adj, x), ref)
    if oracle.load_reference_lib() is not None:
        assert np.array_equal(csr_sparse_dense_matmul(adj, x), oracle.reference_spmm(ptr, col, val, x))
    # the K calls of one propagate() pass the same adjacency: uploaded CSR, plan and buffers are re-used (keyed on the
    # index arrays' addresses AND full content hashes -- the reference makes a fresh float32 copy of `data` per call)
    import ctypes

    def stats():
        h, m_ = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(_lib.lib().sgl_shim_cache_stats(ctypes.byref(h), ctypes.byref(m_)))
        return h.value, m_.value
    h0, m0 = stats()
    y1 = csr_sparse_dense_matmul(adj, x)
    y2 = csr_sparse_dense_matmul(adj, y1)                    # hop 2 of a propagate: same matrix, new dense operand
    h1, m1 = stats()
    assert h1 - h0 >= 2 and m1 == m0
    assert np.array_equal(y2, oracle.oracle_spmm(ptr, col, val, ref))
    adj.data[7] *= 3.0                                       # edited in place: same addresses, different content -> no stale hit
    val2 = adj.data.astype(np.float32)
    assert np.array_equal(csr_sparse_dense_matmul(adj, x), oracle.oracle_spmm(ptr, col, val2, x))
    assert stats()[1] == m1 + 1
    acc = np.ones((n, x.shape[1]), dtype=np.float32)         # accumulate-into-answer semantics of the CPU symbol (matmul.c:37)
    lib = _lib.lib()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    ip32, ix32 = adj.indptr.astype(np.int32), adj.indices.astype(np.int32)
    lib.FloatCSRMulDenseOMP(p(acc), p(val2), p(ix32), p(ip32), p(x), n, x.shape[1])
    want = np.ones_like(acc)
    oracle.oracle_spmm(ptr, col, val2, x, out=want)
    assert np.array_equal(acc, want)


G1_VARIANTS = [("lap", r, None) for r in (0.0, 0.3, 0.5, 1.0)] + \
              [("ppr", 0.5, a) for a in (0.1, 0.15, 0.2, 0.3)] + [("ppr", 0.3, 0.15)]


@pytest.mark.parametrize("gname", ["sym64", "dir40", "pl2000"])
def test_device_normalisation_matches_reference_goldens(goldens, cuda, gname):
    from sgl_amd.operators.utils import adj_to_symmetric_norm_device, canonical_csr
    g = goldens.graph(gname)
    g1 = goldens.npz("g1_norm")
    worst, n_diff32, total = 0.0, 0, 0
    for kind, r, a in G1_VARIANTS:
        key = f"{gname}|{kind}|{r}" + ("" if a is None else f"|{a}")
        ptr, col, v32, v64 = adj_to_symmetric_norm_device(g, r, a, device=cuda, return_fp64=True)
        assert np.array_equal(ptr.cpu().numpy(), g1[gname + "|indptr"]), key          # structure: bit exact
        assert np.array_equal(col.cpu().numpy(), g1[gname + "|indices"]), key
        ref = g1[key]
        got = v64.cpu().numpy()
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
        worst = max(worst, float(rel.max()))
        assert rel.max() <= 1e-14, (key, rel.max())                                   # fp64, same operation order
        d32 = v32.cpu().numpy() != ref.astype(np.float32)
        n_diff32 += int(d32.sum())
        total += d32.size
        assert np.allclose(v32.cpu().numpy(), ref.astype(np.float32), rtol=1.2e-7, atol=0), key   # <= 1 ulp(fp32)
        # device pow() for the degree factors (host_pow=False) stays within 1 ulp(fp32) of the reference
        gc = canonical_csr(g)
        _, _, v32_dev = dev.normalize_adj(*[t.to(cuda) for t in (torch.from_numpy(gc.indptr.astype(np.int64)),
                                                                 torch.from_numpy(gc.indices.astype(np.int32)),
                                                                 torch.from_numpy(gc.data.astype(np.float32)))],
                                          g.shape[0], r, a, host_pow=False)
        assert np.allclose(v32_dev.cpu().numpy(), ref.astype(np.float32), rtol=1.2e-7, atol=0), key
    print(f"{gname}: worst fp64 rel err {worst:.2e}; fp32-rounded values differing: {n_diff32}/{total}")
    # the degree powers come from the host's libm like the reference's: the rounded A_hat is bit-identical to scipy's
    assert n_diff32 == 0, (gname, n_diff32, total)


def test_prepared_adjacency_symmetric_fast_path(goldens, cuda):
    """PreparedAdjacency: A + I, degrees and the symmetry fingerprint once per graph; a symmetric A is normalised without
    any transposition and is bit-identical to the general pipeline (and hence to the reference goldens); a directed or
    value-asymmetric A is detected and takes the general path; one preparation serves every (r, alpha)."""
    from sgl_amd.operators.utils import canonical_csr
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(cuda)  # noqa: E731
    g1 = goldens.npz("g1_norm")
    for gname, want_sym in (("sym64", True), ("pl2000", True), ("dir40", False)):
        g = canonical_csr(goldens.graph(gname))
        n = g.shape[0]
        prep = dev.PreparedAdjacency(to(g.indptr, np.int64), to(g.indices, np.int32), to(g.data, np.float32), n)
        assert prep.symmetric == want_sym, gname
        for kind, r, a in G1_VARIANTS:
            key = f"{gname}|{kind}|{r}" + ("" if a is None else f"|{a}")
            p_, c_, v32, v64 = prep.normalize(r, a, return_fp64=True)
            assert np.array_equal(p_.cpu().numpy(), g1[gname + "|indptr"]) and np.array_equal(c_.cpu().numpy(), g1[gname + "|indices"])
            assert np.array_equal(v32.cpu().numpy(), g1[key].astype(np.float32)), key      # bit-identical to scipy's rounding
            assert np.abs(v64.cpu().numpy() - g1[key]).max() <= 1e-14 * np.abs(g1[key]).max()
    # symmetric structure but ONE asymmetric value: not symmetric
    g = canonical_csr(goldens.graph("pl2000")).copy()
    i = int(np.nonzero(np.diff(g.indptr) > 0)[0][5])
    g.data[g.indptr[i]] = 3.0
    prep = dev.PreparedAdjacency(to(g.indptr, np.int64), to(g.indices, np.int32), to(g.data, np.float32), g.shape[0])
    assert not prep.symmetric
    ref = oracle.sym_norm_csr(g.indptr, g.indices, g.data, g.shape[0], 0.5, None)
    _, _, v32 = prep.normalize(0.5)
    assert np.array_equal(v32.cpu().numpy(), ref[2].astype(np.float32))


def test_operators_over_one_matrix_share_its_preparation(goldens, cuda):
    """a fresh GraphOp per trial over the same scipy matrix (a PaSca-style search) re-uses ONE device copy of A, A + I and the
    degrees (operators.base_op.prepared_graph: keyed on object identity + content); results stay bit-identical to the goldens; an
    in-place edit of the matrix is noticed; clear_graph_cache() releases the copies; cache_prepared = False restores the old path"""
    from sgl_amd import config
    from sgl_amd.operators import base_op
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    g = goldens.graph("pl2000")
    g1 = goldens.npz("g1_norm")
    x = hash_matrix(2000, 16, seed=3)
    base_op.clear_graph_cache()
    ops = [LaplacianGraphOp(2, r=0.5, strict_order=True), PprGraphOp(2, r=0.5, alpha=0.15, strict_order=True),
           LaplacianGraphOp(2, r=0.3, strict_order=True)]
    hops = [op.propagate(g, x) for op in ops]
    assert len(base_op._GRAPHS) == 1
    prep = base_op._GRAPHS[0][2]
    assert base_op.prepared_graph(g, cuda) is prep
    for op, key in zip(ops, ("pl2000|lap|0.5", "pl2000|ppr|0.5|0.15", "pl2000|lap|0.3")):
        ref = oracle.propagate((g1["pl2000|indptr"], g1["pl2000|indices"], g1[key].astype(np.float32)), x, 2)
        assert all(np.array_equal(h.cpu().numpy(), r_) for h, r_ in zip(hops[ops.index(op)], ref)), key
    g2 = g.copy()
    LaplacianGraphOp(1).propagate(g2, x)
    assert len(base_op._GRAPHS) == 2                                    # another object: its own entry
    g2.data[7] = 3.0                                                    # edited in place: the old preparation must not be served
    y = LaplacianGraphOp(1, strict_order=True).propagate(g2, x)[1]
    want = oracle.propagate(oracle.laplacian_adj(g2.indptr, g2.indices, g2.data, 2000, 0.5), x, 1)[1]
    assert np.array_equal(y.cpu().numpy(), want) and len(base_op._GRAPHS) == 2
    del g2
    import gc
    gc.collect()
    base_op.prepared_graph(g, cuda)
    assert len(base_op._GRAPHS) == 1                                    # the entry of the matrix that is gone was dropped
    # an entry dies WITH its matrix (weak-reference callback), not at the next lookup
    g3 = g.copy()
    LaplacianGraphOp(1).propagate(g3, x)
    assert len(base_op._GRAPHS) == 2
    del g3
    gc.collect()
    assert len(base_op._GRAPHS) == 1
    # what stays resident: A + I in fp64 + degrees (12 bytes per non-zero + 16 per row), the raw copy of a symmetric matrix is
    # released, and the fp64 Laplacian a PPR request leaves behind is dropped unless a sweep is asked for (config.keep_sweep_values)
    prep = base_op._GRAPHS[0][2]
    m = prep.nnz_out
    base_bytes = m * 12 + (2000 + 1) * 8 + 2000 * 8
    assert prep.src is None and prep.cached and prep.nbytes() in (base_bytes, base_bytes + 2000 * 8)   # (+ the diagonal positions a PPR request leaves)
    PprGraphOp(1, r=0.5, alpha=0.2).propagate(g, x)
    assert prep.__dict__.get("_hat64") is None
    config.keep_sweep_values = True
    try:
        PprGraphOp(1, r=0.5, alpha=0.2).propagate(g, x)
        assert prep.__dict__["_hat64"] is not None and prep.nbytes() >= m * 20
        v64 = prep.normalize(0.5, None, return_fp64=True)[3]
        v64.zero_()                                                      # a caller's edit of what it was handed ...
        y_ppr = PprGraphOp(1, r=0.5, alpha=0.2, strict_order=True).propagate(g, x)[1]
        ref = oracle.propagate((g1["pl2000|indptr"], g1["pl2000|indices"], g1["pl2000|ppr|0.5|0.2"].astype(np.float32)), x, 1)[1]
        assert np.array_equal(y_ppr.cpu().numpy(), ref)                  # ... never reaches the cached Laplacian
    finally:
        config.keep_sweep_values = False
    # the byte budget: a preparation that alone exceeds it is handed out but NOT kept (transient, as before the cache existed);
    # under a budget that fits one entry the least recently used one is evicted
    base_op.clear_graph_cache()
    old = config.cache_prepared_gb
    try:
        config.cache_prepared_gb = 1e-6
        big = LaplacianGraphOp(2, r=0.5, strict_order=True).propagate(g, x)
        assert not base_op._GRAPHS and all(torch.equal(a_, b_) for a_, b_ in zip(big, hops[0]))
        assert base_op.prepared_graph(g, cuda).cached is False
        config.cache_prepared_gb = base_bytes * 1.5 / 2 ** 30
        g4_, g5_ = g.copy(), g.copy()
        p4 = base_op.prepared_graph(g4_, cuda)
        p5 = base_op.prepared_graph(g5_, cuda)
        assert p4.cached and p5.cached and len(base_op._GRAPHS) == 1 and base_op._GRAPHS[0][2] is p5
    finally:
        config.cache_prepared_gb = old
    base_op.clear_graph_cache()
    assert not base_op._GRAPHS
    config.cache_prepared = False
    try:
        again = LaplacianGraphOp(2, r=0.5, strict_order=True).propagate(g, x)
        assert not base_op._GRAPHS and all(torch.equal(a_, b_) for a_, b_ in zip(again, hops[0]))
    finally:
        config.cache_prepared = True


def test_long_weighted_rows_are_summed_in_scipy_order(cuda):
    """the fp64 degrees of A + I are SEQUENTIAL sums in column order (what scipy's row sum forms); rows beyond 64 elements are summed
    by a whole wavefront fed from registers -- same operations, same order: bit-identical degrees for real-valued weights, and the
    normalised matrix bit-identical to the oracle's"""
    rng = np.random.default_rng(3)
    n = 3000
    rows = np.concatenate([np.full(700, 5), np.full(65, 9), np.full(64, 10), np.full(2999, 11), rng.integers(0, n, 6000)])
    cols = np.concatenate([rng.choice(n, 700, replace=False), rng.choice(n, 65, replace=False), rng.choice(n, 64, replace=False),
                           np.delete(np.arange(n), 11), rng.integers(0, n, 6000)])
    a = sp.coo_matrix((rng.random(rows.size).astype(np.float32) + 0.01, (rows, cols)), shape=(n, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    to = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x, dtype=dt)).to(cuda)  # noqa: E731
    prep = dev.PreparedAdjacency(to(a.indptr, np.int64), to(a.indices, np.int32), to(a.data, np.float32), n)
    ap = (a.astype(np.float64) + sp.eye(n, format="csr")).tocsr()
    ap.sort_indices()
    want = np.zeros(n)
    for i in range(n):                                           # the sequential sum, element by element
        s_ = 0.0
        for v_ in ap.data[ap.indptr[i]:ap.indptr[i + 1]]:
            s_ += v_
        want[i] = s_
    assert np.array_equal(prep.deg.cpu().numpy(), want)
    assert int(np.diff(ap.indptr).max()) == n and (np.diff(ap.indptr) > 64).sum() >= 3
    ref = oracle.sym_norm_csr(a.indptr, a.indices, a.data, n, 0.5, None)
    _, _, v32 = prep.normalize(0.5)
    assert np.array_equal(v32.cpu().numpy(), ref[2].astype(np.float32))


@pytest.mark.parametrize("gname", ["sym64", "dir40", "pl2000"])
def test_row_block_normalisation_matches_full(goldens, cuda, gname):
    """sgl_norm_block_*: every rank of a row-sharded job normalises only ITS rows; the blocks laid end to end are
    bit-identical to the single-GPU normalisation (and hence to the reference goldens)."""
    from sgl_amd.operators.utils import canonical_csr
    g = canonical_csr(goldens.graph(gname))
    n = g.shape[0]
    symmetric = gname != "dir40"
    t = g if symmetric else sp.csr_matrix(g.T)            # rows of A^T are what a rank holds for a directed graph
    t.sort_indices()
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(cuda)  # noqa: E731
    for kind, r, a in G1_VARIANTS:
        gp, gc, gv = to(g.indptr, np.int64), to(g.indices, np.int32), to(g.data, np.float32)
        fptr, fcol, fval, f64 = dev.normalize_adj(gp, gc, gv, n, r, a, return_fp64=True)
        deg = torch.empty(n, dtype=torch.float64, device=cuda)
        _lib.check(_lib.lib().sgl_norm_degrees(n, 0, _lib.ptr(gp), _lib.ptr(gc), _lib.ptr(gv), _lib.ptr(deg),
                                               _lib.current_stream_ptr()))
        torch.cuda.synchronize()
        for bounds in ([0, n], [0, n // 3, n // 3, n - 5, n]):     # one block; three blocks incl. an EMPTY one
            ptrs, cols, vals, v64s = [], [], [], []
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                rp = t.indptr[lo:hi + 1].astype(np.int64) - int(t.indptr[lo])
                cc = t.indices[t.indptr[lo]:t.indptr[hi]]
                vv = t.data[t.indptr[lo]:t.indptr[hi]]
                bp, bc, bv, b64 = dev.normalize_block(to(rp, np.int64), to(cc, np.int32), to(vv, np.float32), lo, n, r, a,
                                                      symmetric=symmetric, return_fp64=True,
                                                      deg=deg if len(bounds) > 2 else None)
                ptrs.append(bp.cpu().numpy()); cols.append(bc.cpu().numpy()); vals.append(bv.cpu().numpy()); v64s.append(b64.cpu().numpy())
            offs = np.concatenate([[0], np.cumsum([p[-1] for p in ptrs])])
            full_ptr = np.concatenate([p[:-1] + o for p, o in zip(ptrs, offs[:-1])] + [offs[-1:]])
            assert np.array_equal(full_ptr, fptr.cpu().numpy()), (gname, kind, r, a)
            assert np.array_equal(np.concatenate(cols), fcol.cpu().numpy())
            assert np.array_equal(np.concatenate(v64s), f64.cpu().numpy())      # same operations in the same order
            assert np.array_equal(np.concatenate(vals), fval.cpu().numpy())


def test_prepared_block_serves_an_r_alpha_sweep(goldens, cuda):
    """PreparedBlock: T + I and the degree vector once per row block; every (r, alpha) of a sweep is then ONE pass over the block.
    A PPR request keeps the fp64 Laplacian of its r, the next alpha is a pure stream over it (sgl_norm_block_mix) -- all of it
    bit-identical to the reference goldens; the degree powers are evaluated once per r (cache hit counted)."""
    from sgl_amd.operators.utils import canonical_csr
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(cuda)  # noqa: E731
    g1 = goldens.npz("g1_norm")
    for gname in ("pl2000", "dir40"):
        g = canonical_csr(goldens.graph(gname))
        n = g.shape[0]
        symmetric = gname != "dir40"
        t = g if symmetric else sp.csr_matrix(g.T)
        t.sort_indices()
        lo, hi = n // 4, n - 3                               # a proper block: the degrees of the other rows come from `deg`
        gp, gc, gv = to(g.indptr, np.int64), to(g.indices, np.int32), to(g.data, np.float32)
        deg = torch.empty(n, dtype=torch.float64, device=cuda)
        _lib.check(_lib.lib().sgl_norm_degrees(n, 0, _lib.ptr(gp), _lib.ptr(gc), _lib.ptr(gv), _lib.ptr(deg), _lib.current_stream_ptr()))
        rp = t.indptr[lo:hi + 1].astype(np.int64) - int(t.indptr[lo])
        cc, vv = t.indices[t.indptr[lo]:t.indptr[hi]], t.data[t.indptr[lo]:t.indptr[hi]]
        prep = dev.PreparedBlock(to(rp, np.int64), to(cc, np.int32), to(vv, np.float32), lo, n, symmetric=symmetric, deg=deg)
        full_ptr = g1[gname + "|indptr"]
        a0, a1 = int(full_ptr[lo]), int(full_ptr[hi])
        dev.clear_power_cache()
        hits0 = dev.pow_stats["cache_hits"]
        for kind, r, a in [("ppr", 0.5, 0.1), ("ppr", 0.5, 0.15), ("lap", 0.5, None), ("ppr", 0.5, 0.3), ("lap", 0.3, None),
                           ("ppr", 0.3, 0.15), ("ppr", 0.5, 0.2)]:
            key = f"{gname}|{kind}|{r}" + ("" if a is None else f"|{a}")
            p_, c_, v32, v64 = prep.normalize(r, a, return_fp64=True)
            assert np.array_equal(p_.cpu().numpy(), full_ptr[lo:hi + 1] - a0) and np.array_equal(c_.cpu().numpy(), g1[gname + "|indices"][a0:a1])
            assert np.array_equal(v32.cpu().numpy(), g1[key][a0:a1].astype(np.float32)), key
            assert np.abs(v64.cpu().numpy() - g1[key][a0:a1]).max() <= 1e-14 * np.abs(g1[key]).max(), key
            one_pass = dev.normalize_block(to(rp, np.int64), to(cc, np.int32), to(vv, np.float32), lo, n, r, a, symmetric=symmetric,
                                           deg=deg, return_fp64=True)
            assert torch.equal(one_pass[2], v32) and torch.equal(one_pass[3], v64), key          # cached and one-pass routes: same bits
        assert prep._hat64[0][0] == 0.5 and dev.pow_stats["cache_hits"] > hits0
        prep.drop_values()
        assert prep._hat64 is None
    # route selection: unit weights -> few distinct degrees -> the host route even under "auto"; real-valued weights -> device pow
    g = canonical_csr(goldens.graph("pl2000")).copy()
    n = g.shape[0]
    big = sp.block_diag([g] * 12).tocsr()                    # 24 000 nodes: beyond the small-vector shortcut
    big.data = (1.0 + np.random.default_rng(0).random(big.nnz)).astype(np.float32)
    big = ((big + big.T) * 0.5).tocsr()                      # real-valued symmetric weights: every node its own degree
    big.sort_indices()
    for mat, want in ((sp.block_diag([g] * 12).tocsr(), "host_unique"), (big, "device")):
        mat.sort_indices()
        nb = mat.shape[0]
        prep = dev.PreparedBlock(to(mat.indptr, np.int64), to(mat.indices, np.int32), to(mat.data, np.float32), 0, nb)
        dev.clear_power_cache()
        before = dict(dev.pow_stats)
        _, _, v_auto = prep.normalize(0.5, None, host_pow="auto")
        assert dev.pow_stats[want] == before[want] + 1, (want, dev.pow_stats, before)
        _, _, v_host = prep.normalize(0.5, None, host_pow=True)
        ref = oracle.sym_norm_csr(mat.indptr, mat.indices, mat.data, nb, 0.5, None)[2].astype(np.float32)
        assert np.array_equal(v_host.cpu().numpy(), ref)                                           # host route: bit-identical
        assert np.allclose(v_auto.cpu().numpy(), ref, rtol=1.2e-7, atol=0)                          # device pow: <= 1 ulp(fp32)
    # the host route of an all-distinct vector goes through the thread team in chunks: same values as one numpy call
    d = np.random.default_rng(1).random(3_000_000) * 50 + 0.5
    l1, r1 = dev._host_powers(d, 0.3)
    with np.errstate(divide="ignore"):
        assert np.array_equal(l1, np.power(d, 0.3 - 1)) and np.array_equal(r1, np.power(d, -0.3))
    dt = torch.from_numpy(d).to(cuda)
    lt, rt = dev.degree_powers(dt, 0.3, host_pow=True)
    assert np.array_equal(lt.cpu().numpy(), l1) and np.array_equal(rt.cpu().numpy(), r1)


def test_hashed_directed_blocks_normalise_like_the_whole_matrix(cuda):
    """S4 path in small: hashed (directed, unsorted, possibly duplicated) row blocks -> canonicalize_block ->
    sgl_norm_block_*(symmetric=False) with the degree vector summed over the blocks' column sums == the whole-matrix
    normalisation of A = T^T, bit for bit"""
    from sgl_amd import synthetic as sy
    from sgl_amd.dist import RowBlock, canonicalize_block
    n = 3000
    table = sy.degree_table(12.0, 300)
    bounds = [0, 1100, 1100, 2500, n]                          # incl. an empty block
    blocks = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        rp, c, v = sy.hashed_block_torch(5, lo, hi - lo, n, table, device=cuda)
        blocks.append(canonicalize_block(RowBlock(lo, hi, n, rp, c, v)))
    # the whole T as scipy, canonical
    T = sp.vstack([sp.csr_matrix((b.val.cpu().numpy(), b.col.cpu().numpy(), b.rowptr.cpu().numpy()), shape=(b.n_local, n))
                   for b in blocks]).tocsr()
    for b in blocks:                                           # canonical: sorted, unique columns
        cc, rp = b.col.cpu().numpy(), b.rowptr.cpu().numpy()
        assert all((np.diff(cc[rp[i]:rp[i + 1]]) > 0).all() for i in range(0, b.n_local, 97))
    A = sp.csr_matrix(T.T)
    A.sort_indices()
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(cuda)  # noqa: E731
    for r, alpha in ((0.5, None), (0.3, 0.2)):
        fptr, fcol, fval = dev.normalize_adj(to(A.indptr, np.int64), to(A.indices, np.int32), to(A.data, np.float32), n, r, alpha)
        deg = torch.zeros(n, dtype=torch.float64, device=cuda)
        parts = []
        for b in blocks:                                       # what the ranks' all-reduce would add up
            p_, c_, v_, v64 = dev.normalize_block(b.rowptr, b.col, b.val, b.lo, n, 0.0, None, symmetric=False, return_fp64=True,
                                                  deg=torch.ones(n, dtype=torch.float64))      # r = 0, deg = 1: T' itself
            deg.index_add_(0, c_.long(), v64)
        for b in blocks:
            parts.append(dev.normalize_block(b.rowptr, b.col, b.val, b.lo, n, r, alpha, symmetric=False, deg=deg))
        offs = np.concatenate([[0], np.cumsum([int(p[0][-1]) for p in parts])])
        ptr_all = np.concatenate([p[0].cpu().numpy()[:-1] + o for p, o in zip(parts, offs[:-1])] + [offs[-1:]])
        assert np.array_equal(ptr_all, fptr.cpu().numpy())
        assert np.array_equal(np.concatenate([p[1].cpu().numpy() for p in parts]), fcol.cpu().numpy())
        assert np.array_equal(np.concatenate([p[2].cpu().numpy() for p in parts]), fval.cpu().numpy())


def test_adj_to_symmetric_norm_scipy_contract(goldens, cuda):
    from sgl_amd.operators.utils import adj_to_symmetric_norm
    g = goldens.graph("dir40")
    out = adj_to_symmetric_norm(g.tocoo(), 0.5)
    assert sp.issparse(out) and out.dtype == np.float64 and out.shape == g.shape
    g1 = goldens.npz("g1_norm")
    assert np.array_equal(out.indices, g1["dir40|indices"])
    assert np.allclose(out.data, g1["dir40|lap|0.5"], rtol=1e-14, atol=0)


def test_propagate_matches_reference_goldens(goldens, cuda):
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    g2 = goldens.npz("g2_prop")
    meta = goldens.json("g2_prop")
    n_exact = n_total = 0
    for key, m in meta.items():
        g = goldens.graph(m["graph"])
        x = hash_matrix(g.shape[0], m["d"], seed=m["seed"], order=m["order"])
        norm = oracle.sym_norm_csr(g.indptr, g.indices, g.data, g.shape[0], m["r"], m["alpha"])
        scales = oracle.propagate((norm[0], norm[1], np.abs(norm[2])), np.abs(x), m["K"])
        for strict in (True, False):
            mk = (lambda K: LaplacianGraphOp(K, r=m["r"], strict_order=strict)) if m["kind"] == "lap" else \
                 (lambda K: PprGraphOp(K, r=m["r"], alpha=m["alpha"], strict_order=strict))
            hops = mk(m["K"]).propagate(g, x)
            assert len(hops) == m["K"] + 1 and all(h.is_cuda and h.dtype == torch.float32 for h in hops)
            assert np.array_equal(hops[0].cpu().numpy(), x)
            check = range(1, m["K"] + 1) if m["keep"] == "all" else [m["K"]]
            for h in check:
                rep = oracle.parity_report(hops[h].cpu().numpy(), g2[f"{key}|h{h}"], TOL, scale=scales[h])
                assert rep["ok"], (key, strict, h, rep)
                if strict:
                    n_total += 1
                    n_exact += rep["bit_equal"]
            sums = np.array([f.double().sum().item() for f in hops])
            assert np.allclose(sums, g2[f"{key}|sums"], rtol=1e-4, atol=1e-3), key
    print(f"propagate strict mode: {n_exact}/{n_total} golden hop matrices reproduced bit-for-bit")
    assert n_exact == n_total        # A_hat is bit-identical to scipy's (degree powers from the host's libm), the chain too


def test_propagate_host_output_and_input_types(goldens, cuda):
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    g = goldens.graph("pl256")
    x = hash_matrix(256, 16, seed=1)
    ref = oracle.propagate(oracle.laplacian_adj(g.indptr, g.indices, g.data, 256, 0.5), x, 3)
    host = LaplacianGraphOp(3, host_output=True, strict_order=True).propagate(g, x)
    assert all((not h.is_cuda) and h.dtype == torch.float32 and h.is_contiguous() for h in host)
    assert host[0].data_ptr() == x.ctypes.data                       # element 0 aliases the caller's array
    for h in range(4):
        assert oracle.parity_ok(host[h].numpy(), ref[h], TOL)
    # superset inputs: torch CPU tensor, CUDA tensor, float64 ndarray, F-order
    for feat in (torch.from_numpy(x), torch.from_numpy(x).to(cuda), x.astype(np.float64), np.asfortranarray(x)):
        hops = LaplacianGraphOp(3).propagate(g, feat)
        assert oracle.parity_ok(hops[3].cpu().numpy(), ref[3], TOL)
    # cache: same matrix object -> the normalised adjacency is reused; a different r is not served from it
    op = LaplacianGraphOp(1)
    op.propagate(g, x)
    first = op._adj
    op.propagate(g, x)
    assert op._adj is first
    g2 = g.copy()
    g2.data[:] = 2.0
    op.propagate(g2, x)
    assert op._adj is not first


def test_error_contract_gpu(goldens, cuda):
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    g5 = goldens.json("g5_errors")
    g = goldens.graph("sym64")
    x = hash_matrix(64, 4, seed=1)

    def expect(case, fn):
        with pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == case["raised"] and str(ei.value) == case["msg"], (ei.value, case)

    expect(g5["propagate_coo_adj"], lambda: LaplacianGraphOp(2).propagate(g.tocoo(), x))
    expect(g5["propagate_shape_mismatch"], lambda: LaplacianGraphOp(2).propagate(g, x[:10]))
    expect(g5["propagate_tensor_feature"], lambda: LaplacianGraphOp(2, strict_types=True).propagate(g, torch.from_numpy(x)))


# ---- aggregators ------------------------------------------------------------------------------------------
def g3_feats(goldens, dev, requires_grad=False):
    g3 = goldens.npz("g3_agg")
    feats = [torch.from_numpy(g3[f"feat{j}"]).to(dev) for j in range(5)]
    if requires_grad:
        feats = [f.clone().requires_grad_(True) for f in feats]
    return feats, g3


def test_simple_aggregators_bit_exact(goldens, cuda):
    from sgl_amd.operators import message_op as m
    feats, g3 = g3_feats(goldens, cuda)
    H = 5
    assert np.array_equal(m.LastMessageOp().aggregate(feats).cpu().numpy(), g3["last"])
    for (s, e) in ((0, H), (1, H - 1)):
        tag = f"{s}_{e}"
        for name, cls in (("concat", m.ConcatMessageOp), ("mean", m.MeanMessageOp), ("sum", m.SumMessageOp),
                          ("max", m.MaxMessageOp), ("min", m.MinMessageOp)):
            y = cls(s, e).aggregate(feats)
            assert y.is_cuda
            assert np.array_equal(y.cpu().numpy(), g3[f"{name}|{tag}"]), (name, tag)
    # CPU tensors in -> CPU tensor out (computed on the GPU)
    y = m.MeanMessageOp(0, H).aggregate([f.cpu() for f in feats])
    assert not y.is_cuda and np.array_equal(y.numpy(), g3["mean|0_5"])


@pytest.mark.parametrize("d", [100, 147, 7, 64])
def test_fused_aggregation_in_the_spmm_epilogue(goldens, cuda, d):
    """GraphOp.propagate_reduce (sgl_spmm_acc_f32: the running aggregate updated where each row is produced) against
    aggregate(propagate(...)): bit-identical for last / sum / mean / max / min / simple_weighted, any hop range, also when long rows
    are split (fix-up path) and when the range ends before the last hop."""
    from sgl_amd.operators import message_op as m
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    a = long_row_graph(n=1600, seed=9)
    a = (a + a.T).tocsr()
    a.data[:] = 1.0
    x = hash_matrix(a.shape[0], d, seed=3)
    K = 4
    ops = [m.LastMessageOp(), m.SumMessageOp(0, K + 1), m.SumMessageOp(1, 4), m.SumMessageOp(2, 3), m.MeanMessageOp(0, K + 1),
           m.MeanMessageOp(1, K + 1), m.MeanMessageOp(0, 10), m.MeanMessageOp(K, K + 1),
           m.SimpleWeightedMessageOp(0, K + 1, "alpha", 0.85), m.SimpleWeightedMessageOp(1, K + 1, "alpha", 0.3),
           m.SimpleWeightedMessageOp(0, 3, "hand_crafted", [0.5, -0.25, 2.0]),
           m.MaxMessageOp(0, K + 1), m.MaxMessageOp(1, 4), m.MinMessageOp(0, K + 1), m.MinMessageOp(2, K + 1), m.MaxMessageOp(3, 4)]
    for gop in (LaplacianGraphOp(K, r=0.5), PprGraphOp(K, r=0.3, alpha=0.2), LaplacianGraphOp(K, r=0.5, strict_order=True)):
        hops = gop.propagate(a, x)
        for op in ops:
            spec = op.fused_spec(K + 1)
            assert spec is not None, op
            fused = gop.propagate_reduce(a, x, **spec)
            want = op.aggregate(hops)
            assert fused.shape == want.shape and torch.equal(fused, want), (type(op).__name__, op._start, op._end, d)
    # ops that cannot ride on the SpMM say so; weird ranges fall back
    assert m.ConcatMessageOp(0, 3).fused_spec(5) is None
    # a NaN in any hop wins in max / min (torch's rule), also in the running form
    xn = x.copy()
    xn[7, 0] = np.nan
    hn = gop.propagate(a, xn)
    for op in (m.MaxMessageOp(0, K + 1), m.MinMessageOp(1, K + 1)):
        fused, want = gop.propagate_reduce(a, xn, **op.fused_spec(K + 1)), op.aggregate(hn)
        assert torch.isnan(want).any() and torch.equal(torch.isnan(fused), torch.isnan(want))
        assert torch.equal(torch.nan_to_num(fused), torch.nan_to_num(want))
    assert gop.propagate_reduce(a, x, kind="sum", start=3, end=2) is None
    # the reference's exceptions come first, exactly as in propagate()
    with pytest.raises(TypeError):
        gop.propagate_reduce(a.tocoo(), x, kind="sum")
    # end to end through a model: the aggregate is the same tensor, the hop list is simply not kept
    from sgl_amd import config
    from sgl_amd.models.homo import SGC, SSGC
    for cls, args in ((SGC, (3, d, 5)), (SSGC, (3, d, 5))):
        config.fuse_aggregate = True
        mf = cls(*args)
        mf.preprocess(a, x)
        config.fuse_aggregate = False
        mu = cls(*args)
        mu.preprocess(a, x)
        config.fuse_aggregate = "auto"
        # folded: no hop list is held ... until somebody asks (the reference's dist tasks read the attribute): then it is produced
        assert mf.__dict__["_hop_list"] is None and len(mu._processed_feat_list) == 4
        assert torch.equal(mf._processed_feature, mu._processed_feature), cls.__name__
        lazy = mf._processed_feat_list
        assert len(lazy) == 4 and all(torch.equal(p_, q_) for p_, q_ in zip(lazy, mu._processed_feat_list))
        assert mf._processed_feat_list is lazy                         # computed once
    # "auto": `last` is always folded (it is free); sum-like aggregates only when the hop list would be heavy
    ma = SGC(3, d, 5)
    ma.preprocess(a, x)
    assert ma.__dict__["_hop_list"] is None and len(ma._processed_feat_list) == 4
    mb = SSGC(3, d, 5)
    mb.preprocess(a, x)
    assert len(mb._processed_feat_list) == 4 and torch.equal(mb._processed_feature, mu._processed_feature)
    # the inputs kept for the lazy list do not travel with a pickled / copied model, and a list that would not fit is never produced
    # by an incidental read: the read gives None like the reference's attribute before preprocess() (hasattr / getmembers / `is None`
    # probes must not raise) and warns ONCE with the way out; materialize_hops() is the explicit request and refuses unless forced
    import copy
    import pickle
    mc = SGC(3, d, 5)
    mc.preprocess(a, x)
    assert mc.__dict__["_hop_source"] is not None
    for clone in (copy.deepcopy(mc), pickle.loads(pickle.dumps(mc))):
        assert clone.__dict__["_hop_source"] is None and clone._processed_feat_list is None
        assert torch.equal(clone._processed_feature.cpu(), mc._processed_feature.cpu())
    assert mc.hops_available() == "lazy"
    mc._hops_fit = lambda: False
    assert mc.hops_available() == "too_large"
    with pytest.warns(RuntimeWarning, match="materialize_hops"):         # said once, with the way out
        assert mc._processed_feat_list is None
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                   # ... and only once
        assert mc._processed_feat_list is None and hasattr(mc, "_processed_feat_list")
    assert mc.__dict__["_hop_list"] is None
    with pytest.raises(RuntimeError):
        mc.materialize_hops()
    assert len(mc.materialize_hops(force=True)) == 4 and mc._processed_feat_list is not None and mc.hops_available() == "kept"
    assert SGC(3, d, 5).hops_available() == "none"


def test_host_output_from_the_pinned_pool_keeps_the_contract(goldens, cuda):
    """host_output=True through the pooled, overlapped download (hops launched one by one, each travelling to a page-locked pooled
    destination while the next is computed): CPU FloatTensors bit-equal to the device-resident hops, hop 0 aliases the caller's
    array, results of an earlier call are never touched by a later one, and buffers nobody holds any more are recycled"""
    from sgl_amd import hostpool
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    g = goldens.graph("pl2000")
    for d in (100, 37):
        x = hash_matrix(2000, d, seed=d)
        dev_hops = LaplacianGraphOp(3, strict_order=True).propagate(g, x)
        op = LaplacianGraphOp(3, strict_order=True, host_output=True)
        first = op.propagate(g, x)
        assert all((not t.is_cuda) and t.dtype == torch.float32 and t.shape == (2000, d) for t in first)
        assert first[0].data_ptr() == x.ctypes.data and all(t.is_pinned() for t in first[1:])
        assert all(torch.equal(h, dh.cpu()) for h, dh in zip(first[1:], dev_hops[1:]))
        keep = [t.clone() for t in first]
        second = op.propagate(g, x * 2.0)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(first[1:], keep[1:]))            # the first call's results are intact
        assert {t.data_ptr() for t in first[1:]}.isdisjoint({t.data_ptr() for t in second[1:]})
        assert all(torch.equal(s_, dh.cpu() * 2.0) for s_, dh in zip(second[1:], dev_hops[1:]))
        ptrs = {t.data_ptr() for t in first[1:]}
        reused = hostpool.stats["reused"]
        del first, keep
        third = op.propagate(g, torch.from_numpy(x))                                         # tensor input: hop 0 is downloaded too
        assert hostpool.stats["reused"] >= reused + 3 and ptrs <= {t.data_ptr() for t in third}
        assert all(torch.equal(t_, dh.cpu()) for t_, dh in zip(third, dev_hops))
        del second, third
    hostpool.trim()


def test_trace_records_the_phases_of_a_propagation(goldens, cuda, capsys):
    """SGL_AMD_TRACE (tracing): propagate() records the wall time of its phases and prints one line; results are
    untouched; off by default"""
    from sgl_amd import config
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    g = goldens.graph("pl2000")
    x = hash_matrix(2000, 40, seed=1)
    plain = LaplacianGraphOp(3).propagate(g, x)
    op = LaplacianGraphOp(3)
    config.trace = True
    try:
        traced = op.propagate(g, x)
        again = op.propagate(g, x)
    finally:
        config.trace = False
    assert all(torch.equal(a, b) for a, b in zip(plain, traced)) and all(torch.equal(a, b) for a, b in zip(plain, again))
    t = op.last_trace
    assert set(t) == {"adjacency_s", "features_s", "hops_s", "output_s", "total_s"} and all(v >= 0 for v in t.values())
    assert abs(t["total_s"] - (t["adjacency_s"] + t["features_s"] + t["hops_s"] + t["output_s"])) < 1e-4
    err = capsys.readouterr().err
    assert err.count("[sgl_amd trace] LaplacianGraphOp.propagate:") == 2 and "hops_s=" in err
    assert not hasattr(LaplacianGraphOp(3), "last_trace")


def test_on_disk_hop_cache_is_keyed_on_content(goldens, cuda, tmp_path):
    """GraphOp(hop_cache_dir=...): a second operator (another process, another run) finds the hop matrices of the same adjacency
    CONTENT + features + parameters on disk and loads them bit for bit; any change of a value, a feature entry, r, alpha or
    prop_steps is another key; a half-written entry is a miss; the reference's exceptions still come first"""
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    g = goldens.graph("pl2000")
    x = hash_matrix(2000, 36, seed=3)
    plain = LaplacianGraphOp(3, r=0.5, strict_order=True).propagate(g, x)
    a = LaplacianGraphOp(3, r=0.5, strict_order=True, hop_cache_dir=tmp_path)
    first = a.propagate(g, x)
    assert (a._hop_cache.hits, a._hop_cache.misses) == (0, 1) and all(torch.equal(p_, q_) for p_, q_ in zip(plain, first))
    b = LaplacianGraphOp(3, r=0.5, strict_order=True, hop_cache_dir=tmp_path)
    again = b.propagate(g.copy(), x.copy())                           # other objects, same content
    assert (b._hop_cache.hits, b._hop_cache.misses) == (1, 0) and all(h.is_cuda for h in again)
    assert all(torch.equal(p_, q_) for p_, q_ in zip(plain, again))
    again_t = b.propagate(g, torch.from_numpy(x).to(cuda))            # device features: digest of the same bits -> other key family
    assert all(torch.equal(p_, q_) for p_, q_ in zip(plain, again_t))
    n_entries = len(os.listdir(tmp_path))
    x2 = x.copy()
    x2[1234, 5] += 1.0
    g2 = g.copy()
    g2.data[777] = 2.0
    for op, adj_, feat_ in ((b, g, x2), (b, g2, x), (LaplacianGraphOp(2, r=0.5, strict_order=True, hop_cache_dir=tmp_path), g, x),
                            (LaplacianGraphOp(3, r=0.3, strict_order=True, hop_cache_dir=tmp_path), g, x),
                            (PprGraphOp(3, r=0.5, alpha=0.1, strict_order=True, hop_cache_dir=tmp_path), g, x)):
        before = len(os.listdir(tmp_path))
        op.propagate(adj_, feat_)
        assert len(os.listdir(tmp_path)) == before + 1                # every variation is its own entry
    assert len(os.listdir(tmp_path)) == n_entries + 5
    # a half-written entry (no meta.json) is recomputed, not trusted
    key = a._hop_cache.key((type(a).__name__, a._norm_params(), 3, True), g, x)
    os.remove(os.path.join(tmp_path, key, "meta.json"))
    c = LaplacianGraphOp(3, r=0.5, strict_order=True, hop_cache_dir=tmp_path)
    redo = c.propagate(g, x)
    assert (c._hop_cache.hits, c._hop_cache.misses) == (0, 1) and all(torch.equal(p_, q_) for p_, q_ in zip(plain, redo))
    with pytest.raises(ValueError):
        a.propagate(g, x[:10])


def test_hop_ranges_match_reference_goldens(goldens, cuda):
    """G8 (recorded from the reference): Mean's divisor is (end - start) whatever the slice held, partial / single-hop
    ranges; aggregate(propagate()) and the fused propagate_reduce both reproduce the reference -- sum / mean / last bit for
    bit in strict order, simple_weighted within 1e-6"""
    from sgl_amd.operators import message_op as m
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    g8 = goldens.npz("g8_ranges")
    feats, _ = g3_feats(goldens, cuda)
    for key, want in g8.items():
        if key.startswith(("mean|", "sum|")):
            kind, rng = key.split("|")
            s_, e_ = (int(t) for t in rng.split("_"))
            op = m.MeanMessageOp(s_, e_) if kind == "mean" else m.SumMessageOp(s_, e_)
            assert np.array_equal(op.aggregate(feats).cpu().numpy(), want), key
    g = goldens.graph("pl256")
    x = hash_matrix(256, 20, seed=808)
    K = 4
    for name, gop in (("lap", LaplacianGraphOp(K, r=0.5, strict_order=True)), ("ppr", PprGraphOp(K, r=0.3, alpha=0.2, strict_order=True))):
        hops = gop.propagate(g, x)
        cases = [(m.LastMessageOp(), "last", True), (m.SumMessageOp(0, K + 1), f"sum|0_{K + 1}", True), (m.SumMessageOp(1, 3), "sum|1_3", True),
                 (m.MeanMessageOp(0, K + 1), f"mean|0_{K + 1}", True), (m.MeanMessageOp(0, 10), "mean|0_10", True),
                 (m.MeanMessageOp(2, 4), "mean|2_4", True),
                 (m.SimpleWeightedMessageOp(0, K + 1, "alpha", 0.85), f"simple_weighted|alpha0.85|0_{K + 1}", False),
                 (m.SimpleWeightedMessageOp(1, K + 1, "alpha", 0.3), f"simple_weighted|alpha0.3|1_{K + 1}", False)]
        for op, key, exact in cases:
            want = g8[f"prop|{name}|{key}"]
            for got in (op.aggregate(hops), gop.propagate_reduce(g, x, **op.fused_spec(K + 1))):
                got = got.cpu().numpy()
                assert (np.array_equal(got, want) if exact else oracle.parity_ok(got, want, 1e-6)), (name, key)


def test_slab_hops_make_concat_a_view(goldens, cuda):
    """slab_hops: hop k is produced in column slice k of ONE [n, (K+1) d] buffer; the hops are bit-identical to the
    separate-buffer propagation and ConcatMessageOp over consecutive hops returns a zero-copy view of the slab"""
    from sgl_amd.operators import message_op as m
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    g = goldens.graph("pl2000")
    for d in (100, 16, 147):
        x = hash_matrix(g.shape[0], d, seed=4)
        K = 3
        plain = LaplacianGraphOp(K).propagate(g, x)
        slab = LaplacianGraphOp(K, slab_hops=True).propagate(g, x)
        assert all(torch.equal(a, b) for a, b in zip(plain, slab))
        cat = m.ConcatMessageOp(0, K + 1).aggregate(slab)
        want = torch.hstack(plain)
        assert torch.equal(cat, want)
        same_storage = cat.untyped_storage().data_ptr() == slab[0].untyped_storage().data_ptr()
        assert same_storage == (d % 4 == 0)             # d % 4 != 0: separate padded buffers, concat copies (any-width kernel)
        part = m.ConcatMessageOp(1, 3).aggregate(slab)
        assert torch.equal(part, torch.hstack(plain[1:3]))
        if d % 4 == 0:
            assert part.untyped_storage().data_ptr() == slab[0].untyped_storage().data_ptr()
        # hops out of order / from different propagations are copied as before
        assert torch.equal(m.ConcatMessageOp(0, 2).aggregate([slab[2], slab[0]]), torch.hstack([plain[2], plain[0]]))


def test_non_learnable_aggregators_backpropagate_when_asked(goldens, cuda):
    """hop matrices that require grad (outputs of a learnable stage fed into Concat / Mean / ...): the reference's own
    differentiable expression is evaluated instead of the forward-only HIP kernel; values agree with the kernel path"""
    from sgl_amd.operators import message_op as m
    feats, _ = g3_feats(goldens, cuda)
    ops = [m.SumMessageOp(0, 5), m.MeanMessageOp(1, 4), m.MaxMessageOp(0, 5), m.MinMessageOp(0, 3), m.ConcatMessageOp(0, 5),
           m.OverSmoothDistanceWeightedOp()]
    for op in ops:
        want = op.aggregate(feats)
        leaf = [f.clone().requires_grad_(True) for f in feats]
        got = op.aggregate(leaf)
        assert got.requires_grad and oracle.parity_ok(got.detach().cpu().numpy(), want.cpu().numpy(), 1e-6), type(op).__name__
        got.sum().backward()
        used = leaf if op._start is None else leaf[op._start:op._end]
        assert all(f.grad is not None and torch.isfinite(f.grad).all() for f in used), type(op).__name__


def test_hip_graph_survives_a_wider_eager_call(cuda):
    """the split-row workspace of a handle is grow-only and outgrown buffers are retired, not freed -- a chain
    captured into a hipGraph keeps replaying correctly after an eager SpMM with a wider matrix re-sized the workspace"""
    a = long_row_graph()
    a = sp.csr_matrix((np.abs(a.data) / 50.0, a.indices, a.indptr), shape=a.shape)
    n = a.shape[0]
    csr = device_csr(a.indptr, a.indices, a.data, (n, n), cuda, item_nnz=64, long_row_nnz=128)
    assert csr.info()["n_pieces"] > 0
    x = torch.from_numpy(hash_matrix(n, 8, seed=1)).to(cuda)
    outs = [torch.empty_like(x) for _ in range(2)]
    g = csr.capture_chain(x, outs)
    want = [o.clone() for o in g.replay()]
    torch.cuda.synchronize()
    wide = torch.from_numpy(hash_matrix(n, 200, seed=2)).to(cuda)
    csr.spmm(wide)                                   # needs a larger workspace than the captured chain
    for o in outs:
        o.zero_()
    got = g.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(got, want))


def test_max_min_propagate_nan_like_torch(cuda):
    from sgl_amd.operators import message_op as m
    a = torch.tensor([[1.0, float("nan"), -2.0, 5.0]] * 3, device=cuda)
    b = torch.tensor([[float("nan"), 0.0, 3.0, float("inf")]] * 3, device=cuda)
    for cls, fn in ((m.MaxMessageOp, torch.max), (m.MinMessageOp, torch.min)):
        got = cls(0, 2).aggregate([a, b]).cpu()
        ref = fn(torch.stack([a.cpu(), b.cpu()], 0), dim=0)[0]
        assert torch.equal(torch.isnan(got), torch.isnan(ref))
        assert torch.equal(got[~torch.isnan(got)], ref[~torch.isnan(ref)])


def test_weighted_and_nafs_aggregators(goldens, cuda):
    from sgl_amd.device import nafs_aggregate
    from sgl_amd.operators import message_op as m
    feats, g3 = g3_feats(goldens, cuda)
    H = 5
    for (s, e) in ((0, H), (1, H)):
        y = m.SimpleWeightedMessageOp(s, e, "alpha", 0.85).aggregate(feats).cpu().numpy()
        assert oracle.parity_ok(y, g3[f"simple_weighted|alpha0.85|{s}_{e}"], 1e-6)
    w = [float(v) for v in g3["simple_weighted|hand_crafted|w"]]
    y = m.SimpleWeightedMessageOp(0, H, "hand_crafted", w).aggregate(feats).cpu().numpy()
    assert oracle.parity_ok(y, g3["simple_weighted|hand_crafted|0_5"], 1e-6)
    y = m.OverSmoothDistanceWeightedOp().aggregate(feats).cpu().numpy()
    rep = oracle.parity_report(y, g3["over_smooth"], TOL)
    assert rep["ok"], rep
    _, wts = nafs_aggregate(feats, return_weights=True)
    assert np.allclose(wts.cpu().numpy(), oracle.nafs_weights([f.cpu().numpy() for f in feats]), rtol=1e-5, atol=1e-6)
    assert np.allclose(wts.sum(1).cpu().numpy(), 1.0, atol=1e-5)


def _planted_communities(n, bs, deg, p_in, seed):
    rng = np.random.default_rng(seed)
    a = np.repeat(np.arange(n), deg)
    near = (a // bs) * bs + rng.integers(0, bs, a.size)
    far = rng.integers(0, n, a.size)
    b = np.where(rng.random(a.size) < p_in, np.minimum(near, n - 1), far)
    keep = a != b
    m = sp.coo_matrix((np.ones(keep.sum(), np.float32), (a[keep], b[keep])), shape=(n, n)).tocsr()
    m = ((m + m.T) > 0).astype(np.float32).tocsr()
    m.sort_indices()
    return m


def test_community_reorder_is_transparent(cuda):
    """GraphOp(reorder="community"): the plan-time locality ordering only changes the order in which the rows of A_hat are
    STORED and PROCESSED (sgl_csr_permute_rows + sgl_csr_set_rowmap) -- X, Y, the column ids and the order of every row's terms
    stay the caller's, so hops and fused aggregates are bit-identical with and without it, in strict order too; the order is
    a permutation that makes the planted communities contiguous; permute_csr (the relabelled P A P^T) is scipy's"""
    from sgl_amd import device as dev
    from sgl_amd.io import DeviceAdjacency
    from sgl_amd.operators import message_op as m
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    from sgl_amd.reorder import community_order, community_order_reference, permute_csr
    n, bs = 3000, 100
    adj0 = _planted_communities(n, bs, 12, 0.9, seed=4)
    shuffle = np.random.default_rng(5).permutation(n)                 # hide the communities in the ids
    P = sp.coo_matrix((np.ones(n, np.float32), (shuffle, np.arange(n))), shape=(n, n)).tocsr()
    adj = (P @ adj0 @ P.T).tocsr()
    adj.sort_indices()
    x = hash_matrix(n, 128, seed=6)
    d_adj = DeviceAdjacency.from_scipy(adj, device=cuda)
    order, info = community_order(d_adj.rowptr, d_adj.col, n)
    o = order.cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(n)), info
    ref_order, _ = community_order_reference(d_adj.rowptr, d_adj.col, n)      # no node has > 256 neighbours: the kernel
    assert torch.equal(order, ref_order)                                        # samples nothing and must agree exactly
    # the relabelled matrix keeps the planted communities together: most edges join nodes less than 2 communities apart
    rp, cc, vv = permute_csr(d_adj.rowptr, d_adj.col, d_adj.val, order)
    Q = sp.coo_matrix((np.ones(n, np.float32), (o, np.arange(n))), shape=(n, n)).tocsr()
    want = (Q @ adj @ Q.T).tocsr()
    want.sort_indices()
    assert np.array_equal(rp.cpu().numpy(), want.indptr) and np.array_equal(cc.cpu().numpy(), want.indices)
    assert np.array_equal(vv.cpu().numpy(), want.data)
    coo = want.tocoo()
    near_after = np.mean(np.abs(coo.row - coo.col) < 2 * bs)
    coo0 = adj.tocoo()
    near_before = np.mean(np.abs(coo0.row - coo0.col) < 2 * bs)
    assert near_before < 0.2 and near_after > 0.8, (near_before, near_after, info)
    from sgl_amd.tricks import label_propagation
    for plain, reord in ((LaplacianGraphOp(3, r=0.5), LaplacianGraphOp(3, r=0.5, reorder="community")),
                         (PprGraphOp(2, r=0.3, alpha=0.2), PprGraphOp(2, r=0.3, alpha=0.2, reorder="community")),
                         (LaplacianGraphOp(3, r=0.5, strict_order=True), LaplacianGraphOp(3, r=0.5, strict_order=True, reorder="community"))):
        ha, hb = plain.propagate(adj, x), reord.propagate(adj, x)
        assert len(ha) == len(hb)
        for a, b in zip(ha, hb):
            assert torch.equal(a, b)        # rows are only PROCESSED in another order (x is 128 wide: one fmaf chain per row)
        for op in (m.LastMessageOp(), m.MeanMessageOp(0, 3), m.MaxMessageOp(1, 3), m.SimpleWeightedMessageOp(0, 3, "alpha", 0.85)):
            spec = op.fused_spec(len(ha))
            fa, fb = plain.propagate_reduce(adj, x, **spec), reord.propagate_reduce(adj, x, **spec)
            assert torch.equal(fa, fb), type(op).__name__          # the running aggregate is addressed through the row map too
        assert reord._adj.rowmap is not None and plain._adj.rowmap is None
        assert np.array_equal(np.sort(reord._adj.rowmap.cpu().numpy()), np.arange(n))
        reord.propagate(adj, x)                                        # cached: the ordering is found once per adjacency
    # long rows (split into pieces, summed by the fix-up kernel) under a row map; residual and running aggregate of the epilogues
    big = long_row_graph(n=1600, seed=9)
    big = (big + big.T).tocsr()
    big.data[:] = 1.0
    nb = big.shape[0]
    b_adj = DeviceAdjacency.from_scipy(big, device=cuda)
    rp, cc, vv = dev.normalize_adj(b_adj.rowptr, b_adj.col, b_adj.val, nb, 0.5, None)
    rowmap = torch.from_numpy(np.random.default_rng(2).permutation(nb).astype(np.int32)).to(cuda)   # any permutation will do
    rp2, c2, v2 = dev.permute_rows(rp, cc, vv, rowmap)
    ca = dev.DeviceCSR(rp, cc, vv, (nb, nb), long_row_nnz=256)
    cb = dev.DeviceCSR(rp2, c2, v2, (nb, nb), long_row_nnz=256).set_rowmap(rowmap)
    assert cb.info()["n_long_rows"] > 0 and cb.info()["n_pieces"] > 0
    for dd in (100, 36, 16, 7):
        # d > 64: one non-zero per step, a row = one fmaf chain.  d <= 64 packs several non-zeros per step; a non-zero's slot is
        # its index WITHIN ITS ROW mod R, so the partial sums do not depend on where the row sits in its item: bit-identical too
        same = torch.equal
        xa = dev.upload_rows(hash_matrix(nb, dd, seed=8), cuda)
        assert same(ca.spmm(xa), cb.spmm(xa)), dd
        res = dev.upload_rows(hash_matrix(nb, dd, seed=12), cuda)
        assert same(ca.spmm_axpb_clamp(xa, 0.7, res, 0.0, 1.0), cb.spmm_axpb_clamp(xa, 0.7, res, 0.0, 1.0)), dd
        acc_a, acc_b = res.clone(), res.clone()
        ya, yb = torch.empty_like(xa), torch.empty_like(xa)
        ca.spmm_acc(xa, ya, acc_a, w=0.3, mode="wsum")
        cb.spmm_acc(xa, yb, acc_b, w=0.3, mode="wsum")
        assert same(ya, yb) and same(acc_a, acc_b), dd
    with pytest.raises(Exception):                                     # replicas refuse a mapped handle
        cb.spmm_multi(xa, [ya.data_ptr(), yb.data_ptr()], ya.stride(0))
    cb.set_rowmap(None)                                                # without the map the stored order is what it is
    assert not torch.equal(ca.spmm(xa), cb.spmm(xa))
    # the whole map is validated on the device -- out of range or not a permutation is an error, not a stray write
    bad = rowmap.clone()
    bad[5] = nb + 3
    with pytest.raises(Exception, match="outside"):
        cb.set_rowmap(bad)
    dup = rowmap.clone()
    dup[7] = dup[8]
    with pytest.raises(Exception, match="permutation"):
        cb.set_rowmap(dup)
    cb.set_rowmap(rowmap)
    assert same(ca.spmm(xa), cb.spmm(xa))
    with pytest.raises(ValueError):
        LaplacianGraphOp(2, reorder="rcm").propagate(adj, x)


@pytest.mark.parametrize("d", [4, 8, 16, 24, 36, 50, 64, 100])
def test_two_plans_of_one_matrix_agree_bit_for_bit(goldens, cuda, d):
    """item size, long-row threshold, XCD remap and gathers in flight only change HOW the rows are walked: every row's terms are
    added in the same order and -- in the packed layouts of narrow matrices -- by the same slots, so the products are bit-equal
    (pieces of split rows are cut relative to the row, so they agree as well)"""
    from sgl_amd import _lib
    from sgl_amd import device as dev
    n, ptr, col, val = norm_graph(goldens, "pl2000")
    x = dev.upload_rows(hash_matrix(n, d, seed=d), cuda)
    base = device_csr(ptr, col, val, (n, n), cuda).spmm(x)
    for kw in (dict(item_nnz=64), dict(item_nnz=2048), dict(item_nnz=200, xcd_remap=False)):
        assert torch.equal(device_csr(ptr, col, val, (n, n), cuda, **kw).spmm(x), base), kw
    for unroll in (1, 2, 3):
        _lib.set_tuning("spmm_unroll", unroll)
        try:
            assert torch.equal(device_csr(ptr, col, val, (n, n), cuda, item_nnz=300).spmm(x), base), unroll
        finally:
            _lib.set_tuning("spmm_unroll", 0)


def test_reorder_auto_and_row_sharded_blocks(cuda):
    """reorder="auto": the label-propagation order is kept only when it makes the graph measurably more local than its own ids
    (planted communities behind shuffled ids: applied; the same graph in its natural order or a random graph: left alone).
    Row-sharded storage: ShardedGraphOp(reorder=...) orders the rows inside the rank's block (found on the block's diagonal part)
    behind a row map, for the need-aware exchange and for the full-replica pieces alike -- hops bit-identical, any width."""
    from sgl_amd import reorder
    from sgl_amd.dist import RowBlock, ShardedGraphOp
    from sgl_amd.io import DeviceAdjacency
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    n, bs = 6000, 150
    adj0 = _planted_communities(n, bs, 14, 0.9, seed=11)
    shuffle = np.random.default_rng(3).permutation(n)
    P = sp.coo_matrix((np.ones(n, np.float32), (shuffle, np.arange(n))), shape=(n, n)).tocsr()
    adj = (P @ adj0 @ P.T).tocsr()
    adj.sort_indices()
    rnd = _planted_communities(n, bs, 14, 0.0, seed=12)               # no community structure at all
    for graph, expect in ((adj, True), (adj0, False), (rnd, False)):
        for d in (100, 24):
            x = hash_matrix(n, d, seed=d)
            plain = LaplacianGraphOp(2, r=0.5).propagate(graph, x)
            op = LaplacianGraphOp(2, r=0.5, reorder="auto")
            got = op.propagate(graph, x)
            assert all(torch.equal(a, b) for a, b in zip(plain, got))
            info = op.reorder_info
            assert info["applied"] is expect and (op._adj.rowmap is not None) == expect, info
            assert (info["edge_locality_after"] >= info["edge_locality_before"] + reorder.AUTO_MIN_GAIN) == expect, info
    # the rank's row block (world 1: the whole matrix) through both row-sharded code paths
    da = DeviceAdjacency.from_scipy(adj, device=cuda)
    blk = RowBlock(0, n, n, da.rowptr, da.col, da.val)
    for d in (128, 36):
        x = torch.from_numpy(hash_matrix(n, d, seed=d + 1)).to(cuda)
        want = ShardedGraphOp(3, r=0.5, pieces=2, col_chunks=1).propagate(blk, x)
        for kw in (dict(transport="halo"), dict(transport="p2p", pieces=3)):
            for mode in ("community", "auto"):
                op = ShardedGraphOp(3, r=0.5, col_chunks=1, reorder=mode, **{"pieces": 2, **kw})
                got = op.propagate(blk, x)
                assert all(torch.equal(a, b) for a, b in zip(want, got)), (d, kw, mode)
                if kw["transport"] == "halo":
                    assert op.halo_plan.reorder_info["applied"] is True and op._props["halo"][2].rowmap is not None


def test_round5_kernels_fuzz_random_shapes(cuda):
    """40 random (rows, width, hops) shapes through the kernels added in round 5: every NAFS prefix from one pass against the
    oracle per prefix (widths up to 512: every lane layout; 1 ... 40 hops: more than the fused kernel holds), the ensemble
    combinations, the max / min backward against torch's autograd of stack(...).max(0) with ties and NaNs planted, the padded row
    gather from aligned matrices, column views and duplicate / negative indices"""
    rng = np.random.default_rng(20260928)
    for case in range(40):
        n = int(rng.integers(1, 300))
        d = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 33, 64, 65, 100, 127, 128, 129, 147, 255, 256, 257, 300, 511, 512]))
        H = int(rng.integers(1, 41))
        host = [np.ascontiguousarray((rng.standard_normal((n, d)) * (1.0 - 0.01 * h)).astype(np.float32)) for h in range(H)]
        if n > 2:
            host[0][1] = 0.0                                             # a zero row: cosine 0 with every hop
        feats = []
        for x in host:
            t = dev.alloc_rows(n, d, cuda)
            t.copy_(torch.from_numpy(x))
            feats.append(t)
        tag = (case, n, d, H)
        emit = sorted(set(int(v) for v in rng.integers(0, H, size=min(H, 4))) | {H - 1})
        outs = dev.nafs_prefix(feats, emit)
        for h, o in zip(emit, outs):
            assert oracle.parity_ok(o.cpu().numpy(), oracle.agg_over_smooth_distance(host[:h + 1]), 2e-5, rowwise=False), (tag, h)
            if o.stride(0) != d and n > 1:
                assert float(dev.padded_parent(o)[:, d:].abs().max()) == 0.0, (tag, h)
        acc = [o.clone() * 0.25 for o in outs]
        accp = []
        for a in acc:                                                   # the combinations need padded outputs of their own
            t = dev.alloc_rows(n, d, cuda)
            t.copy_(a)
            accp.append(t)
        dev.nafs_prefix(feats, emit, outs=accp, combine=dev.NAFS_MAX, outs_padded=True)
        assert all(torch.equal(a, torch.maximum(o * 0.25, o)) for a, o in zip(accp, outs)), tag
        # max / min backward
        Hs = min(H, 9)
        base = [h_.copy() for h_ in host[:Hs]]
        if Hs > 2 and n > 3:
            base[2][0] = base[0][0]                                      # ties
            base[1][3, 0] = np.nan                                       # NaN
        g = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(cuda)
        for op, name in ((_lib.SGL_REDUCE_MAX, "max"), (_lib.SGL_REDUCE_MIN, "min")):
            leaves = [torch.from_numpy(b.copy()).to(cuda).requires_grad_(True) for b in base]
            dev.hop_reduce_grad(op, leaves).backward(g)
            ref = [torch.from_numpy(b.copy()).to(cuda).requires_grad_(True) for b in base]
            getattr(torch.stack(ref, 0), name)(0)[0].backward(g)
            assert all(torch.equal(a.grad, b.grad) for a, b in zip(leaves, ref)), (tag, name)
        # row gather: padded source, a column view of a wider matrix, duplicate and negative indices
        m = int(rng.integers(1, 2 * n + 2))
        idx = torch.from_numpy(rng.integers(-n, n, size=m)).to(cuda)
        assert torch.equal(dev.gather_rows(feats[0], idx), feats[0][idx]), tag
        wide = torch.from_numpy(rng.standard_normal((n, d + 9)).astype(np.float32)).to(cuda)
        got = dev.gather_rows(wide[:, 4:4 + d], idx)
        assert torch.equal(got, wide[:, 4:4 + d][idx]), tag
        if got.stride(0) != d and m > 1:
            assert float(dev.padded_parent(got)[:, d:].abs().max()) == 0.0, tag


def test_aggregators_fuzz_random_shapes(cuda):
    """40 random (rows, width, hops, padded / dense) shapes through every aggregator kernel family: the bit-exact ones
    (sum / max / concat) against numpy bit for bit, the weighted ones and their gradients within tolerance -- widths 1..600
    hit every lane layout (8 / 16 / 32 / 64 lanes, 1 / 2 chunks), the LDS concat tiles, the masked tails and the unaligned
    gradient path; hop counts 1..20 hit every register instantiation and the general fallbacks"""
    from sgl_amd import _lib
    from sgl_amd import device as dev
    rng = np.random.default_rng(20260927)
    for case in range(40):
        n = int(rng.integers(1, 400))
        d = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 33, 64, 65, 100, 127, 128, 129, 147, 255, 256, 257, 300, 511, 513, 600]))
        H = int(rng.integers(1, 21))
        padded = bool(rng.integers(0, 2))
        host = [np.ascontiguousarray(rng.standard_normal((n, d)).astype(np.float32)) for _ in range(H)]
        if padded:
            feats = []
            for x in host:
                t = dev.alloc_rows(n, d, cuda)
                t.copy_(torch.from_numpy(x))
                feats.append(t)
        else:
            feats = [torch.from_numpy(x).to(cuda) for x in host]
        tag = (case, n, d, H, padded)
        assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_SUM, feats).cpu().numpy(), oracle.agg_sum(host, 0, H)), tag
        assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_MAX, feats).cpu().numpy(), oracle.agg_max(host, 0, H)), tag
        assert np.array_equal(dev.hop_concat(feats).cpu().numpy(), np.hstack(host)), tag
        g = rng.standard_normal((n, d)).astype(np.float32)
        gt = torch.from_numpy(g).to(cuda)
        w1 = rng.standard_normal(H).astype(np.float32)
        w1t = torch.from_numpy(w1).to(cuda).requires_grad_(True)
        y1 = dev.hop_wsum1d(feats, w1t)
        assert oracle.parity_ok(y1.detach().cpu().numpy(), oracle.one_dim_weighted_add(host, w1), 1e-5, rowwise=False), tag
        y1.backward(gt)
        ref1 = np.array([(g.astype(np.float64) * x).sum() for x in host])
        assert np.allclose(w1t.grad.cpu().numpy(), ref1, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(ref1).max())), tag
        w2 = oracle.softmax32(rng.standard_normal((n, H)).astype(np.float32), 1)
        w2t = torch.from_numpy(w2).to(cuda).requires_grad_(True)
        y2 = dev.hop_wsum2d(feats, w2t)
        assert oracle.parity_ok(y2.detach().cpu().numpy(), oracle.two_dim_weighted_add(host, w2), 1e-5, rowwise=False), tag
        y2.backward(gt)
        ref2 = np.stack([(g.astype(np.float64) * x).sum(1) for x in host], 1)
        assert np.allclose(w2t.grad.cpu().numpy(), ref2, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref2).max())), tag
        yn, wn = dev.nafs_aggregate(feats, return_weights=True)
        assert np.allclose(wn.cpu().numpy(), oracle.nafs_weights(host), rtol=5e-5, atol=5e-6), tag
        assert oracle.parity_ok(yn.cpu().numpy(), oracle.agg_over_smooth_distance(host), 2e-5, rowwise=False), tag
        v = rng.standard_normal(d).astype(np.float32)
        sc = dev.hop_scores(feats, torch.from_numpy(v).to(cuda)).cpu().numpy()
        ref = np.stack([x.astype(np.float64) @ v for x in host], 1)
        assert np.allclose(sc, ref, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(ref).max())), tag
        if dev.gate_fusable(feats):
            # the single-pass learnable gate and the one-pass 'jk' / 'ori_ref' scores (register-resident row kernels)
            bias = float(rng.standard_normal())
            yg, wg = dev.hop_gate(feats, torch.from_numpy(v).to(cuda), torch.tensor([bias], device=cuda), return_weights=True)
            sg = 1.0 / (1.0 + np.exp(-(ref + bias)))
            wref = np.exp(sg - sg.max(1, keepdims=True))
            wref /= wref.sum(1, keepdims=True)
            assert np.allclose(wg.cpu().numpy(), wref, rtol=2e-4, atol=2e-5), tag
            want = sum(wref[:, h:h + 1] * host[h].astype(np.float64) for h in range(H))
            assert oracle.parity_ok(yg.cpu().numpy(), want.astype(np.float32), 2e-5, rowwise=False), tag
            u = rng.standard_normal((H, d)).astype(np.float32)
            mask = int(rng.integers(0, 1 << H))
            h0 = int(rng.integers(0, H))
            h1 = int(rng.integers(h0, H + 1))
            pp, aa = dev.hop_scores2(feats, torch.from_numpy(v).to(cuda), torch.from_numpy(u).to(cuda), mask, h0, h1)
            assert np.allclose(pp.cpu().numpy(), ref[:, h0:h1], rtol=2e-4, atol=2e-4 * max(1.0, np.abs(ref).max())), tag
            aref = sum((host[j].astype(np.float64) @ u[j]) for j in range(H) if (mask >> j) & 1) if mask else np.zeros(n)
            if mask:
                assert np.allclose(aa.cpu().numpy(), aref, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(aref).max())), tag


@pytest.mark.parametrize("n,d,H", [(3000, 128, 11), (2500, 100, 4), (700, 147, 6), (64, 16, 16), (1, 500, 2), (900, 260, 5)])
def test_single_pass_gate_matches_two_pass_and_autograd(cuda, n, d, H):
    """sgl_hop_gate_f32 (scores, sigmoid, softmax, weighted sum with the rows in registers) against (a) the two-pass route it
    replaces (row-dot pass + torch sigmoid / softmax + weighted-sum pass) and (b) a float64 torch statement of the reference
    expression (learnable_weighted_messahe_op.py:67-71), values and gradients w.r.t. the Linear's weight, bias and the hops"""
    from sgl_amd import device as dev
    g = torch.Generator(device="cpu").manual_seed(n + d + H)
    feats = [dev.alloc_rows(n, d, cuda) for _ in range(H)]
    for f in feats:
        f.copy_(torch.randn(n, d, generator=g))
    v = (torch.randn(d, generator=g) * 0.3).to(cuda).requires_grad_(True)
    b = torch.randn(1, generator=g).to(cuda).requires_grad_(True)
    gout = torch.randn(n, d, generator=g).to(cuda)
    assert dev.gate_fusable(feats)
    fx = []
    for f in feats:                                           # row-padded like the hops GraphOp.propagate produces: the fused kernel
        t = dev.alloc_rows(n, d, cuda)
        t.copy_(f)
        fx.append(t.requires_grad_(True))
    y, w = dev.hop_gate(fx, v, b, return_weights=True)
    # hops whose rows are not 16-byte aligned (a dense [n, 147] tensor) take the two-pass route behind the same call
    if d % 4:
        yd, wd = dev.hop_gate([f.contiguous().clone() for f in feats], v.detach(), b.detach(), return_weights=True)
        assert torch.allclose(wd, w.detach(), rtol=1e-5, atol=1e-6)
        assert oracle.parity_ok(yd.cpu().numpy(), y.detach().cpu().numpy(), 2e-6, rowwise=False)
    # the C entry point with the bias as a host float (the wrapper passes NaN = "read it from the device, after the padded vector":
    # no host synchronisation): bit-identical, and the output's pad columns are written as zeros (whole-line stores)
    vp = dev._padded_vec(v, d, cuda)
    yh = dev.alloc_rows(n, d, cuda, zero_pad=False)
    dev.padded_parent(yh).fill_(7.0)
    ptrs, lds = _lib.hop_arrays([f.detach() for f in fx])
    _lib.check(_lib.lib().sgl_hop_gate_f32(H, ptrs, lds, _lib.ptr(vp), float(b.detach().cpu()), _lib.ptr(yh), yh.stride(0) if n > 1 else d,
                                           None, 0, None, 0, n, d, _lib.current_stream_ptr()), "sgl_hop_gate_f32")
    assert torch.equal(yh, y.detach())
    if n > 1 and yh.stride(0) != d:
        pad = dev.padded_parent(yh)[:, d:]
        assert bool((pad == 7.0).all())                      # the un-suffixed entry point touches nothing beyond column d ...
        _lib.check(_lib.lib().sgl_hop_gate_padded_f32(H, ptrs, lds, _lib.ptr(vp), float(b.detach().cpu()), _lib.ptr(yh), yh.stride(0),
                                                      dev.own_pad(yh), None, 0, None, 0, n, d, _lib.current_stream_ptr()),
                   "sgl_hop_gate_padded_f32")
        assert torch.equal(yh, y.detach()) and bool((pad == 0).all())      # ... the padded one writes the declared pad as zeros
        with pytest.raises(_lib.SglHipError):                # a pad that does not fit the pitch is refused
            _lib.check(_lib.lib().sgl_hop_gate_padded_f32(H, ptrs, lds, _lib.ptr(vp), 0.0, _lib.ptr(yh), yh.stride(0),
                                                          yh.stride(0), None, 0, None, 0, n, d, _lib.current_stream_ptr()), "pad")
    # (a) two-pass route, same kernels' arithmetic for the dots and the FMA sum
    sc = dev.hop_scores(feats, v.detach()) + b.detach()
    w2 = torch.softmax(torch.sigmoid(sc), dim=1)
    y2 = dev.hop_wsum2d(feats, w2)
    assert torch.allclose(w, w2, rtol=1e-5, atol=1e-6)
    assert oracle.parity_ok(y.detach().cpu().numpy(), y2.cpu().numpy(), 2e-6, rowwise=False)
    # (b) the reference expression with autograd, in float64 (the truth) and in float32 (what the reference's own arithmetic gives):
    # the HIP gradients may be at most twice as far from the truth as the float32 expression is (oracle.truth_report; the gradients
    # of v and b are sums over all rows and hops of cancelling terms: the bound is condition-aware for them)
    def expression(dt):
        vv, bb = v.detach().to(dt).requires_grad_(True), b.detach().to(dt).requires_grad_(True)
        ff = [f.detach().to(dt).requires_grad_(True) for f in feats]
        ss = torch.stack([f @ vv + bb for f in ff], dim=1)
        cond = {}
        ss.register_hook(lambda g_: cond.update(b=g_.abs().sum(), v=sum(g_[:, h].abs() @ ff[h].detach().abs() for h in range(H))))
        ww = torch.softmax(torch.sigmoid(ss), dim=1)
        yy = sum(ww[:, h:h + 1] * ff[h] for h in range(H))
        (yy * gout.to(dt)).sum().backward()
        return yy.detach(), vv.grad, bb.grad, [f.grad for f in ff], cond
    y64, dv64, db64, df64, cond = expression(torch.float64)
    y32, dv32, db32, df32, _ = expression(torch.float32)
    assert oracle.parity_ok(y.detach().cpu().numpy(), y64.float().cpu().numpy(), 1e-5, rowwise=False)
    (y * gout).sum().backward()
    for name, got, r32, want, cnd in (("v", v.grad, dv32, dv64, cond["v"]), ("b", b.grad, db32, db64, cond["b"])):
        rep = oracle.truth_report(got.cpu().numpy(), r32.cpu().numpy(), want.cpu().numpy(), cond=cnd.cpu().numpy())
        assert rep["ok"], (name, rep)
    for h in range(H):
        rep = oracle.truth_report(fx[h].grad.cpu().numpy(), df32[h].cpu().numpy(), df64[h].cpu().numpy())
        assert rep["ok"], (h, rep)


def _recursive_step_by_step(feats, weight, bias, cond=None):
    """the reference's loop as written (iterate_learnable_weighted_message_op.py:28-51), any dtype, plain torch.  cond (a dict):
    receives the condition magnitudes of the Linear's gradients -- the sums of the ABSOLUTE terms of weight.grad and bias.grad"""
    d = feats[0].shape[1]
    acc, weights = feats[0], None
    for i in range(len(feats)):
        inp = torch.hstack((feats[i], acc))
        z = inp @ weight.view(-1, 1) + bias
        if cond is not None and z.requires_grad:
            def hook(g_, inp=inp.detach()):
                cond["bias"] = cond.get("bias", 0) + g_.abs().sum()
                cond["weight"] = cond.get("weight", 0) + g_.abs().t() @ inp.abs()
            z.register_hook(hook)
        score = torch.sigmoid(z)
        weights = score if weights is None else torch.hstack((weights, score))
        weights = torch.softmax(weights, dim=1)
        acc = sum(weights[:, j:j + 1] * feats[j] for j in range(i + 1))
    return acc, weights


@pytest.mark.parametrize("n,d,H", [(3000, 128, 11), (2500, 100, 4), (700, 147, 6), (64, 16, 16), (90, 20, 12), (1, 500, 2), (900, 260, 5),
                                   (333, 64, 1), (257, 36, 8), (4000, 600, 3)])
def test_single_pass_recursive_gate_matches_the_step_by_step_form(cuda, n, d, H):
    """sgl_hop_recursive_f32 (GAMLP-R's gate with the hop rows in registers: 2 H row-dots, the recursion on per-hop scalars, the
    weighted sum) against a float64 statement of the reference's step-by-step loop (iterate_learnable_weighted_message_op.py:28-51):
    values, final weights, gradients w.r.t. the Linear's weight, bias and the hops.  (90, 20, 12): more hops than lanes in a row's
    group (the array form of the recursion); (4000, 600, 3): rows the register-resident kernel does not take -> two row-dot passes,
    the [n, H] recursion in torch and one weighted-sum pass behind the same call."""
    from sgl_amd import device as dev
    g = torch.Generator(device="cpu").manual_seed(n + d + H)
    feats = [dev.alloc_rows(n, d, cuda) for _ in range(H)]
    for h, f in enumerate(feats):
        f.copy_(torch.randn(n, d, generator=g) * (1.0 - 0.04 * h))
    weight = (torch.randn(1, 2 * d, generator=g) * (0.5 / d ** 0.5)).to(cuda).requires_grad_(True)
    b = torch.randn(1, generator=g).to(cuda).requires_grad_(True)
    gout = torch.randn(n, d, generator=g).to(cuda)
    assert dev.gate_fusable(feats) == (d <= 512)
    fx = [f.detach().requires_grad_(True) for f in feats]
    y, w = dev.hop_recursive(fx, weight, b, return_weights=True)
    assert y.shape == (n, d) and w.shape == (n, H)
    # the reference's loop in float64 (the truth) and in float32 (the reference's own arithmetic): values, weights and gradients of the
    # single-pass kernels may be at most twice as far from the truth as the float32 loop is (oracle.truth_report).  The forward kernel
    # evaluates exp / reciprocal with the hardware approximations and no max subtraction, the backward kernel recomputes the step
    # weights with IEEE expf / division: they differ in the last bits, which is inside this bound by the same argument.
    cond = {}
    w64, b64 = weight.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    f64 = [f.detach().double().requires_grad_(True) for f in feats]
    y64, wt64 = _recursive_step_by_step(f64, w64, b64, cond)
    w32, b32 = weight.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    f32 = [f.detach().clone().requires_grad_(True) for f in feats]
    y32, wt32 = _recursive_step_by_step(f32, w32, b32)
    rep = oracle.truth_report(w.detach().cpu().numpy(), wt32.detach().cpu().numpy(), wt64.detach().cpu().numpy())
    assert rep["ok"], ("weights", rep)
    rep = oracle.truth_report(y.detach().cpu().numpy(), y32.detach().cpu().numpy(), y64.detach().cpu().numpy())
    assert rep["ok"], ("out", rep)
    assert oracle.parity_ok(y.detach().cpu().numpy(), y64.detach().float().cpu().numpy(), 1e-5, rowwise=False)
    (y * gout).sum().backward()
    (y64 * gout.double()).sum().backward()
    (y32 * gout).sum().backward()
    for name, got, r32, want in (("weight", weight.grad, w32.grad, w64.grad), ("bias", b.grad, b32.grad, b64.grad)):
        rep = oracle.truth_report(got.cpu().numpy(), r32.cpu().numpy(), want.cpu().numpy(), cond=cond[name].cpu().numpy())
        assert rep["ok"], (name, rep)
    for h in range(H):
        rep = oracle.truth_report(fx[h].grad.cpu().numpy(), f32[h].grad.cpu().numpy(), f64[h].grad.cpu().numpy())
        assert rep["ok"], (h, rep)
    if d <= 512:
        # the pad columns of an own output are written as zeros (whole-line stores); the scalar route gives the same weights
        if n > 1 and y.stride(0) != d:
            assert bool((dev.padded_parent(y.detach())[:, d:] == 0).all())
        wt = weight.detach().reshape(-1)
        w2 = dev.recursive_weights(dev.hop_scores(feats, wt[:d]), dev.hop_scores(feats, wt[d:]), b.detach())
        assert torch.allclose(w.detach(), w2, rtol=1e-5, atol=2e-6)
        # the operator routes device hops through it
        from sgl_amd.operators.message_op import IterateLearnableWeightedMessageOp
        op = IterateLearnableWeightedMessageOp(0, H, "recursive", d).to(cuda)
        op.load_state_dict({"_IterateLearnableWeightedMessageOp__learnable_weight.weight": weight.detach(),
                            "_IterateLearnableWeightedMessageOp__learnable_weight.bias": b.detach()})
        with torch.no_grad():
            assert torch.equal(op.aggregate(feats), y.detach())


@pytest.mark.parametrize("n,d,H", [(50_000, 147, 6), (3000, 128, 11), (777, 100, 4), (5, 7, 3), (1, 500, 2), (4099, 1024, 16), (300, 36, 1)])
def test_hop_colsum_is_the_weight_gradient_of_the_row_dots(cuda, n, d, H):
    """sgl_hop_colsum_f32: out[h] = sum_n w[n, h] X_h[n, :] (and with one weight per row shared by all hops) -- what torch computes
    as one transposed GEMV per hop in the backward of the gate / jk / ori_ref scores -- against float64, deterministic, and the
    fallback for hops the kernel cannot take gives the same answer"""
    from sgl_amd import device as dev
    g = torch.Generator(device="cpu").manual_seed(n + d + H)
    feats = []
    for _ in range(H):
        t = dev.alloc_rows(n, d, cuda)
        t.copy_(torch.randn(n, d, generator=g))
        feats.append(t)
    w = torch.randn(n, H, generator=g).to(cuda)
    ws = torch.randn(n, generator=g).to(cuda)
    got, got_s = dev.hop_colsum(feats, w), dev.hop_colsum(feats, ws, shared=True)
    assert got.shape == (H, d) and got_s.shape == (H, d)
    want = torch.stack([f.double().t() @ w[:, h].double() for h, f in enumerate(feats)])
    want_s = torch.stack([f.double().t() @ ws.double() for f in feats])
    mag = torch.stack([f.double().abs().t() @ w[:, h].double().abs() for h, f in enumerate(feats)]).clamp_min(1e-30)
    mag_s = torch.stack([f.double().abs().t() @ ws.double().abs() for f in feats]).clamp_min(1e-30)
    assert float(((got.double() - want).abs() / mag).max()) <= 2e-6 and float(((got_s.double() - want_s).abs() / mag_s).max()) <= 2e-6
    assert torch.equal(dev.hop_colsum(feats, w), got)                                    # no atomics: run-to-run identical
    dense = [f.contiguous().clone() for f in feats]                                      # [n, d] rows at 4-byte alignment for odd d
    alt = dev.hop_colsum(dense, w)
    assert float(((alt.double() - want).abs() / mag).max()) <= 2e-5


@pytest.mark.parametrize("d", [100, 13])
def test_aggregators_with_more_than_sixteen_hops(cuda, d):
    """21 hop matrices (a 20-hop NAFS run): beyond the register-resident kernels' 16-hop limit every aggregator must take
    its general path -- two-pass NAFS, the LDS / scalar row-dot, the scalar 1-D weight gradient -- and still match"""
    from sgl_amd import _lib
    from sgl_amd import device as dev
    n, H = 333, 21
    host = [np.ascontiguousarray(hash_matrix(n, d, seed=120 + h) * (1.0 - 0.03 * h), dtype=np.float32) for h in range(H)]
    feats = []
    for x in host:
        t = dev.alloc_rows(n, d, cuda)
        t.copy_(torch.from_numpy(x))
        feats.append(t)
    assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_SUM, feats).cpu().numpy(), oracle.agg_sum(host, 0, H))
    assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_MAX, feats).cpu().numpy(), oracle.agg_max(host, 0, H))
    assert np.array_equal(dev.hop_concat(feats).cpu().numpy(), np.hstack(host))
    w1 = np.linspace(0.5, -0.2, H).astype(np.float32)
    w1t = torch.from_numpy(w1).to(cuda).requires_grad_(True)
    y1 = dev.hop_wsum1d(feats, w1t)
    assert oracle.parity_ok(y1.detach().cpu().numpy(), oracle.one_dim_weighted_add(host, w1), 1e-6, rowwise=False)
    g = hash_matrix(n, d, seed=7)
    y1.backward(torch.from_numpy(g).to(cuda))
    ref1 = np.array([(g.astype(np.float64) * x).sum() for x in host])
    assert np.allclose(w1t.grad.cpu().numpy(), ref1, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref1).max()))
    w2 = oracle.softmax32(hash_matrix(n, H, seed=9), 1)
    w2t = torch.from_numpy(w2).to(cuda).requires_grad_(True)
    y2 = dev.hop_wsum2d(feats, w2t)
    assert oracle.parity_ok(y2.detach().cpu().numpy(), oracle.two_dim_weighted_add(host, w2), 1e-6, rowwise=False)
    y2.backward(torch.from_numpy(g).to(cuda))
    ref2 = np.stack([(g.astype(np.float64) * x).sum(1) for x in host], 1)
    assert np.allclose(w2t.grad.cpu().numpy(), ref2, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref2).max()))
    yn, wn = dev.nafs_aggregate(feats, return_weights=True)
    assert np.allclose(wn.cpu().numpy(), oracle.nafs_weights(host), rtol=2e-5, atol=2e-6)
    assert oracle.parity_ok(yn.cpu().numpy(), oracle.agg_over_smooth_distance(host), 1e-5, rowwise=False)
    v = hash_matrix(1, d, seed=3).reshape(-1)
    sc = dev.hop_scores(feats, torch.from_numpy(v).to(cuda)).cpu().numpy()
    assert np.allclose(sc, np.stack([x.astype(np.float64) @ v for x in host], 1), rtol=1e-4, atol=1e-4)
    # the recursive gate beyond 16 hops: row-dot passes + the [n, H] recursion + one weighted-sum pass, against the reference loop
    wt = torch.from_numpy(hash_matrix(1, 2 * d, seed=4) * (0.5 / d ** 0.5)).to(cuda)
    b = torch.tensor([0.1], device=cuda)
    y, w = dev.hop_recursive(feats, wt, b, return_weights=True)
    y64, w64 = _recursive_step_by_step([f.double() for f in feats], wt.double(), b.double())
    assert float((w.double() - w64).abs().max()) <= 2e-6
    assert oracle.parity_ok(y.cpu().numpy(), y64.float().cpu().numpy(), 1e-5, rowwise=False)


@pytest.mark.parametrize("d,H", [(147, 6), (147, 11), (501, 5), (65, 16), (257, 4), (86, 3), (85, 3), (1023, 2), (1025, 3), (1025, 4), (511, 16)])
def test_concat_any_width_lds_tiles(cuda, d, H):
    """any-width concat of long rows (assembled in LDS, 1024-float tiles): rows shorter / longer than a tile, hop
    boundaries that fall inside and exactly on tile boundaries, the last partial vector -- bit-equal to numpy's hstack
    and to the funnel-select kernel it replaces"""
    from sgl_amd import _lib
    from sgl_amd import device as dev
    n = 301
    host = [hash_matrix(n, d, seed=90 + h) for h in range(H)]
    feats = []
    for x in host:
        t = dev.alloc_rows(n, d, cuda)
        t.copy_(torch.from_numpy(x))
        feats.append(t)
    want = np.hstack(host)
    got = dev.hop_concat(feats)
    assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want)
    # the pad columns of the output's own pitch are written (as zeros): every line of a row is written whole
    assert dev.own_pad(got) == got.stride(0) - d * H and not bool(dev.padded_parent(got)[:, d * H:].any())
    for mode in (0, 2, 3):                                    # funnel-select, 1024-float tiles, whole rows per block: same bits
        _lib.set_tuning("concat_lds", mode)
        try:
            other = dev.hop_concat(feats)
            assert torch.equal(other, got) and not bool(dev.padded_parent(other)[:, d * H:].any()), mode
        finally:
            _lib.set_tuning("concat_lds", 1)
    # the un-suffixed C entry point (pad_cols = 0) touches nothing beyond the row -- an output may be a slice of a wider matrix
    wide = torch.full((n, d * H + 8), 7.0, device=cuda)
    ptrs, lds = _lib.hop_arrays(feats)
    _lib.check(_lib.lib().sgl_hop_concat_f32(H, ptrs, lds, _lib.ptr(wide), wide.stride(0), n, d, _lib.current_stream_ptr()), "sgl_hop_concat_f32")
    assert np.array_equal(wide[:, :d * H].cpu().numpy(), want) and bool((wide[:, d * H:] == 7.0).all())


@pytest.mark.parametrize("d", [147, 7, 100, 500, 1, 5, 12, 13, 33])
@pytest.mark.parametrize("padded", [True, False])
def test_aggregators_any_width_and_layout(cuda, d, padded):
    """every aggregator kernel against the numpy oracle for widths that are / are not multiples of 4, on row-padded
    buffers (16-byte lanes, masked tail) and on plain contiguous tensors (4-byte lanes)"""
    from sgl_amd import _lib
    from sgl_amd import device as dev
    n, H = 777, 6
    host = [hash_matrix(n, d, seed=50 + h) * (1.0 - 0.1 * h) + hash_matrix(n, d, seed=50) * (0.1 * h) for h in range(H)]
    host = [np.ascontiguousarray(x, dtype=np.float32) for x in host]
    if padded:
        feats = []
        for x in host:
            t = dev.alloc_rows(n, d, cuda)
            t.copy_(torch.from_numpy(x))
            feats.append(t)
    else:
        feats = [torch.from_numpy(x).to(cuda) for x in host]
    assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_SUM, feats).cpu().numpy(), oracle.agg_sum(host, 0, H))
    assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_MEAN, feats).cpu().numpy(), oracle.agg_mean(host, 0, H))
    assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_MAX, feats).cpu().numpy(), oracle.agg_max(host, 0, H))
    assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_MIN, feats).cpu().numpy(), oracle.agg_min(host, 0, H))
    assert np.array_equal(dev.hop_concat(feats).cpu().numpy(), oracle.agg_concat(host, 0, H))
    w1 = np.linspace(0.5, -0.2, H).astype(np.float32)
    y = dev.hop_reduce(_lib.SGL_REDUCE_WSUM, feats, torch.from_numpy(w1)).cpu().numpy()
    assert oracle.parity_ok(y, oracle.one_dim_weighted_add(host, w1), 1e-6, rowwise=False)
    w1t = torch.from_numpy(w1).to(cuda).requires_grad_(True)
    g1 = hash_matrix(n, d, seed=78)
    dev.hop_wsum1d(feats, w1t).backward(torch.from_numpy(g1).to(cuda))     # dw[h] = <dOut, X_h>: single-pass vector kernel
    dw1_ref = np.array([(g1.astype(np.float64) * x).sum() for x in host])
    assert np.allclose(w1t.grad.cpu().numpy(), dw1_ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(dw1_ref).max()))
    w2 = oracle.softmax32(hash_matrix(n, H, seed=9), 1)
    w2t = torch.from_numpy(w2).to(cuda).requires_grad_(True)
    out = dev.hop_wsum2d(feats, w2t)
    assert oracle.parity_ok(out.detach().cpu().numpy(), oracle.two_dim_weighted_add(host, w2), 1e-6, rowwise=False)
    g = hash_matrix(n, d, seed=77)
    out.backward(torch.from_numpy(g).to(cuda))
    dw_ref = np.stack([(g.astype(np.float64) * x).sum(1) for x in host], 1)
    assert np.allclose(w2t.grad.cpu().numpy(), dw_ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(dw_ref).max()))
    yn, wn = dev.nafs_aggregate(feats, return_weights=True)
    assert np.allclose(wn.cpu().numpy(), oracle.nafs_weights(host), rtol=2e-5, atol=2e-6)
    assert oracle.parity_ok(yn.cpu().numpy(), oracle.agg_over_smooth_distance(host), 1e-5, rowwise=False)
    # the two-pass NAFS path gives the same answer as the fused single-pass kernel
    _lib.set_tuning("nafs_fused", 0)
    try:
        y2 = dev.nafs_aggregate(feats)
    finally:
        _lib.set_tuning("nafs_fused", 1)
    assert oracle.parity_ok(y2.cpu().numpy(), yn.cpu().numpy(), 1e-6, rowwise=False)


@pytest.mark.parametrize("n,H,d", [(1, 1, 3), (2, 1, 1), (1, 5, 64), (65, 2, 2), (3, 12, 9)])
def test_aggregators_degenerate_shapes(cuda, n, H, d):
    """one row, one hop, one column: every aggregator kernel still matches the oracle"""
    from sgl_amd import _lib
    from sgl_amd import device as dev
    host = [np.ascontiguousarray(hash_matrix(n, d, seed=90 + h), dtype=np.float32) for h in range(H)]
    feats = []
    for x in host:
        t = dev.alloc_rows(n, d, cuda)
        t.copy_(torch.from_numpy(x))
        feats.append(t)
    assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_SUM, feats).cpu().numpy(), oracle.agg_sum(host, 0, H))
    assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_MEAN, feats).cpu().numpy(), oracle.agg_mean(host, 0, H))
    assert np.array_equal(dev.hop_reduce(_lib.SGL_REDUCE_MAX, feats).cpu().numpy(), oracle.agg_max(host, 0, H))
    assert np.array_equal(dev.hop_concat(feats).cpu().numpy(), oracle.agg_concat(host, 0, H))
    w2 = oracle.softmax32(hash_matrix(n, H, seed=9), 1)
    out = dev.hop_wsum2d(feats, torch.from_numpy(w2).to(cuda))
    assert oracle.parity_ok(out.cpu().numpy(), oracle.two_dim_weighted_add(host, w2), 1e-6, rowwise=False)
    yn = dev.nafs_aggregate(feats)
    assert oracle.parity_ok(yn.cpu().numpy(), oracle.agg_over_smooth_distance(host), 1e-5, rowwise=False)
    idx = torch.arange(n - 1, -1, -1, device=cuda)
    assert np.array_equal(dev.gather_rows(feats[0], idx).cpu().numpy(), host[0][::-1])


@pytest.mark.parametrize("kind", ["simple", "simple_allow_neg", "gate", "ori_ref", "jk"])
def test_learnable_aggregators_forward_and_backward(goldens, cuda, kind):
    from sgl_amd.operators.message_op import LearnableWeightedMessageOp
    _, g3 = g3_feats(goldens, cuda)
    gout = torch.from_numpy(hash_matrix(96, 12, seed=777)).to(cuda)
    args = {"simple": (4,), "simple_allow_neg": (4,), "gate": (12,), "ori_ref": (12,), "jk": (4, 12)}[kind]
    for (s, e) in ((0, 5), (1, 5)):
        tag = f"learnable|{kind}|{s}_{e}"
        op = LearnableWeightedMessageOp(s, e, kind, *args)
        op.load_state_dict({k[len(tag) + 7:]: torch.from_numpy(v) for k, v in g3.items() if k.startswith(tag + "|param|")})
        op = op.to(cuda)
        feats, _ = g3_feats(goldens, cuda, requires_grad=True)
        y = op.aggregate(feats)
        rep = oracle.parity_report(y.detach().cpu().numpy(), g3[tag + "|out"], TOL)
        assert rep["ok"], (tag, rep)
        # tolerances DERIVED from the float64 truth (G12: the reference's module in .double()): the HIP result may be at most twice
        # as far from it as the reference's own float32 result (oracle.truth_report)
        g12 = goldens.npz("g12_fp64_truth")
        rep = oracle.truth_report(y.detach().cpu().numpy(), g3[tag + "|out"], g12["g3|" + tag + "|out"])
        assert rep["ok"], (tag, "out", rep)
        (y * gout).sum().backward()
        for name, p in op.named_parameters():
            # (Linear gradients are sums of cancelling terms: the bound is also condition-aware, see oracle.truth_report)
            rep = oracle.truth_report(p.grad.cpu().numpy(), g3[tag + "|grad|" + name], g12["g3|" + tag + "|grad|" + name],
                                      cond=g12.get("g3|" + tag + "|gradcond|" + name))
            assert rep["ok"], (tag, name, rep)
        for j, f in enumerate(feats):
            got = (f.grad if f.grad is not None else torch.zeros_like(f)).cpu().numpy()
            rep = oracle.truth_report(got, g3[tag + f"|dfeat{j}"], g12["g3|" + tag + f"|dfeat{j}"])
            assert rep["ok"], (tag, j, rep)


def test_iterate_and_projected_concat(goldens, cuda):
    from sgl_amd.operators.message_op import IterateLearnableWeightedMessageOp, ProjectedConcatMessageOp
    _, g3 = g3_feats(goldens, cuda)
    gout = torch.from_numpy(hash_matrix(96, 12, seed=777)).to(cuda)
    op = IterateLearnableWeightedMessageOp(0, 5, "recursive", 12)
    op.load_state_dict({k[len("iterate|0_5|param|"):]: torch.from_numpy(v) for k, v in g3.items() if k.startswith("iterate|0_5|param|")})
    op = op.to(cuda)
    feats, _ = g3_feats(goldens, cuda, requires_grad=True)
    y = op.aggregate(feats)
    rep = oracle.parity_report(y.detach().cpu().numpy(), g3["iterate|0_5|out"], TOL)
    assert rep["ok"], rep
    g12 = goldens.npz("g12_fp64_truth")
    rep = oracle.truth_report(y.detach().cpu().numpy(), g3["iterate|0_5|out"], g12["g3|iterate|0_5|out"])
    assert rep["ok"], rep
    (y * gout).sum().backward()
    for name, p in op.named_parameters():
        rep = oracle.truth_report(p.grad.cpu().numpy(), g3["iterate|0_5|grad|" + name], g12["g3|iterate|0_5|grad|" + name],
                                  cond=g12.get("g3|iterate|0_5|gradcond|" + name))
        assert rep["ok"], (name, rep)
    for j, f in enumerate(feats):
        rep = oracle.truth_report(f.grad.cpu().numpy(), g3[f"iterate|0_5|dfeat{j}"], g12[f"g3|iterate|0_5|dfeat{j}"])
        assert rep["ok"], (j, rep)
    pc = ProjectedConcatMessageOp(0, 5, 12, 8, 2)
    pc.load_state_dict({k[len("proj_concat|0_5|param|"):]: torch.from_numpy(v) for k, v in g3.items() if k.startswith("proj_concat|0_5|param|")})
    pc = pc.to(cuda).eval()
    feats, _ = g3_feats(goldens, cuda)
    with torch.no_grad():
        y = pc.aggregate(feats)
    rep = oracle.parity_report(y.cpu().numpy(), g3["proj_concat|0_5|out"], 1e-4)
    assert rep["ok"], rep


def test_grad_carrying_stateless_ops_and_projected_concat_run_through_the_library(goldens, cuda):
    """hop matrices that REQUIRE GRAD through concat / sum / mean / max / min and ProjectedConcat's hstack: the same HIP kernels as
    the no-grad path inside autograd Functions (forward bit-identical), backward = the reference expression's own gradient --
    slices, broadcast (one true division for mean), torch's selection rule for max / min incl. ties and NaNs -- and no torch cat /
    stack kernel in a profiler trace of the ops (the hstack is the aggregator part of ProjectedConcat)."""
    from sgl_amd.operators import message_op as mo
    from sgl_amd.operators.message_op._common import torch_combine
    n, d, H = 257, 37, 5
    base = [hash_matrix(n, d, seed=50 + h) for h in range(H)]
    base[2][5:9] = base[0][5:9]                        # ties: torch gives the gradient to the FIRST extremal hop
    base[3][11, 3] = np.nan                            # NaN: the first NaN takes it
    base[1][11, 3] = np.nan
    gout = torch.from_numpy(hash_matrix(n, d, seed=99)).to(cuda)
    same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))  # noqa: E731

    def leaves():
        return [torch.from_numpy(b.copy()).to(cuda).requires_grad_(True) for b in base]

    for kind, cls in (("sum", mo.SumMessageOp), ("mean", mo.MeanMessageOp), ("max", mo.MaxMessageOp), ("min", mo.MinMessageOp)):
        for (s_, e_) in ((0, H), (1, 4)):
            fs = leaves()
            y = cls(s_, e_).aggregate(fs)
            assert y.requires_grad and same(y.detach(), cls(s_, e_).aggregate([f.detach() for f in fs]))
            y.backward(gout)
            fr = leaves()
            yr = torch_combine(kind, fr[s_:e_], divisor=(e_ - s_) if kind == "mean" else None)      # the reference's expression
            yr.backward(gout)
            for h in range(H):
                if s_ <= h < e_:
                    # (mean: the reference's CPU DivBackward is a true division; torch on the GPU multiplies by the reciprocal)
                    want = fr[h].grad if kind != "mean" else (gout.cpu() / float(e_ - s_)).to(cuda)
                    assert same(fs[h].grad, want), (kind, s_, e_, h)
                else:
                    assert fs[h].grad is None
    # mean over a slice shorter than (end - start): still divided by (end - start), forward and backward
    fs = leaves()
    y = mo.MeanMessageOp(2, 9).aggregate(fs)
    y.backward(gout)
    assert same(y.detach().cpu(), (fs[2].detach().cpu() + fs[3].detach().cpu() + fs[4].detach().cpu()) / 7.0)
    assert torch.equal(fs[3].grad.cpu(), gout.cpu() / 7.0) and fs[0].grad is None
    fs = leaves()
    y = mo.ConcatMessageOp(1, 4).aggregate(fs)
    assert y.shape == (n, 3 * d) and same(y.detach(), torch.hstack([f.detach() for f in fs[1:4]]))
    g3c = torch.from_numpy(hash_matrix(n, 3 * d, seed=98)).to(cuda)
    y.backward(g3c)
    assert all(torch.equal(fs[1 + k].grad, g3c[:, k * d:(k + 1) * d]) for k in range(3)) and fs[0].grad is None
    # ProjectedConcat with gradients: the projections are torch GEMMs (out of scope), their hstack is the library's kernel
    _, g3 = g3_feats(goldens, cuda)
    pc = mo.ProjectedConcatMessageOp(0, 5, 12, 8, 2)
    pc.load_state_dict({k[len("proj_concat|0_5|param|"):]: torch.from_numpy(v) for k, v in g3.items() if k.startswith("proj_concat|0_5|param|")})
    pc = pc.to(cuda).eval()
    feats, _ = g3_feats(goldens, cuda, requires_grad=True)
    y = pc.aggregate(feats)
    assert y.requires_grad and oracle.parity_ok(y.detach().cpu().numpy(), g3["proj_concat|0_5|out"], TOL, rowwise=False)
    y.square().sum().backward()
    got = [p_.grad.clone() for p_ in pc.parameters()]
    mlps = list(pc.children())[0]
    cols = [mlps[0](feats[0].detach())] + [torch.relu(m(f.detach())) for m, f in zip(list(mlps)[1:], feats[1:])]
    want = torch.autograd.grad(torch.hstack(cols).square().sum(), list(pc.parameters()))
    assert all(torch.allclose(a_, b_, rtol=1e-5, atol=1e-6) for a_, b_ in zip(got, want))
    # the trace of the ops alone (forward + backward, nothing of the comparisons above): our kernels, no torch cat / stack kernel
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        feats, _ = g3_feats(goldens, cuda, requires_grad=True)
        pc.aggregate(feats).square().sum().backward()
        for cls in (mo.ConcatMessageOp, mo.SumMessageOp, mo.MeanMessageOp, mo.MaxMessageOp, mo.MinMessageOp):
            fs = leaves()
            cls(0, H).aggregate(fs).nan_to_num(nan=0.0).sum().backward()
        torch.cuda.synchronize()
    names = {e.key for e in prof.key_averages()}
    assert any("hop_concat" in k for k in names) and any("hop_reduce_kernel" in k for k in names) and any("hop_select_bwd" in k for k in names), sorted(names)
    assert not [k for k in names if "CatArray" in k or k in ("aten::cat", "aten::stack", "aten::hstack", "aten::_foreach_add")], sorted(names)


# ---- BASELINE config 5 at its own hop count (G9: PPR / Laplacian k = 10; every MessageOp over H = 11 hop matrices) ----------------
def test_config5_ppr_and_laplacian_k10_match_reference_goldens(goldens, cuda):
    """PprGraphOp(10, r, alpha in {.1, .2, .3}) and LaplacianGraphOp(10) (search_models.py:19-46, ppr_graph_op.py:13-21) against
    hops recorded from the reference: strict order bit-for-bit from raw A, the default order within the SURVEY 8(c) tolerance;
    the fused propagate_reduce path (last / mean over all 11 hops) on top of the same propagation"""
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    from sgl_amd.operators.message_op import LastMessageOp, MeanMessageOp
    g9 = goldens.npz("g9_config5")
    meta = goldens.json("g9_config5")["prop"]
    n_exact = n_total = n_ppr10 = 0
    for key, m in meta.items():
        g = goldens.graph(m["graph"])
        x = hash_matrix(g.shape[0], m["d"], seed=m["seed"])
        norm = oracle.sym_norm_csr(g.indptr, g.indices, g.data, g.shape[0], m["r"], m["alpha"])
        scales = oracle.propagate((norm[0], norm[1], np.abs(norm[2])), np.abs(x), m["K"])
        n_ppr10 += m["kind"] == "ppr" and m["K"] == 10
        for strict in (True, False):
            mk = (lambda: LaplacianGraphOp(m["K"], r=m["r"], strict_order=strict)) if m["kind"] == "lap" else \
                 (lambda: PprGraphOp(m["K"], r=m["r"], alpha=m["alpha"], strict_order=strict))
            hops = mk().propagate(g, x)
            assert len(hops) == m["K"] + 1
            for h in m["keep"]:
                rep = oracle.parity_report(hops[h].cpu().numpy(), g9[f"prop|{key}|h{h}"], TOL, scale=scales[h])
                assert rep["ok"], (key, strict, h, rep)
                if strict:
                    n_total += 1
                    n_exact += rep["bit_equal"]
            sums = np.array([f.double().sum().item() for f in hops])
            assert np.allclose(sums, g9[f"prop|{key}|sums"], rtol=1e-4, atol=1e-3), key
            if strict and m["K"] == 10:
                last = mk().propagate_reduce(g, x, **LastMessageOp().fused_spec(11))
                assert torch.equal(last, hops[10])
                mean = mk().propagate_reduce(g, x, **MeanMessageOp(0, 11).fused_spec(11))
                assert torch.equal(mean, MeanMessageOp(0, 11).aggregate(hops))
    assert n_ppr10 >= 7 and n_exact == n_total
    print(f"config 5 propagation: {n_exact}/{n_total} golden hop matrices (PPR / Laplacian, k = 10) reproduced bit-for-bit")


def test_ppr_hops_are_a_mix_of_the_laplacian_chain(goldens, cuda):
    """BASELINE config 5's sweep over graph operators with ONE propagation: the hop matrices of PprGraphOp(K, r, alpha) are
    polynomials in the Laplacian's, sum_j C(k, j) (1 - alpha)^j alpha^(k - j) A_hat^j X, so every alpha follows from the
    LaplacianGraphOp(K, r) chain by a triangular mix (sgl_hop_lincomb_f32) -- against the hop matrices the REFERENCE recorded for its
    own PPR chains (G9: k = 10 and k = 1, three graphs incl. the directed one, d = 16 and 128) at 1e-5, and against this library's
    own chain; the mixing kernel itself with arbitrary weights, zero weights skipping a NaN hop, 1 ... 16 inputs, odd widths"""
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp, ppr_hops_from_laplacian
    g9 = goldens.npz("g9_config5")
    meta = goldens.json("g9_config5")["prop"]
    chains = {}
    n_checked = 0
    for key, m in meta.items():
        if m["kind"] != "ppr":
            continue
        g = goldens.graph(m["graph"])
        x = hash_matrix(g.shape[0], m["d"], seed=m["seed"])
        ck = (m["graph"], m["d"], m["seed"], m["r"], m["K"])
        if ck not in chains:
            chains[ck] = LaplacianGraphOp(m["K"], r=m["r"]).propagate(g, x)              # ONE chain per (graph, features, r, K)
        mixed = PprGraphOp(m["K"], r=m["r"], alpha=m["alpha"]).propagate_from_laplacian(chains[ck])
        assert len(mixed) == m["K"] + 1 and mixed[0] is chains[ck][0]
        for h in m["keep"]:
            rep = oracle.parity_report(mixed[h].cpu().numpy(), g9[f"prop|{key}|h{h}"], TOL)
            assert rep["ok"], (key, h, rep)
            n_checked += 1
        own = PprGraphOp(m["K"], r=m["r"], alpha=m["alpha"]).propagate(g, x)
        assert all(oracle.parity_ok(a_.cpu().numpy(), b_.cpu().numpy(), TOL) for a_, b_ in zip(mixed, own)), key
    assert n_checked >= 15 and len(chains) < n_checked
    with pytest.raises(ValueError):
        PprGraphOp(3, alpha=0.2, strict_order=True).propagate_from_laplacian(chains[ck][:4])
    with pytest.raises(ValueError):
        PprGraphOp(3, alpha=0.2).propagate_from_laplacian(chains[ck][:3])
    # more than 16 matrices: the deeper hops are one weighted sum each
    g = goldens.graph("pl256")
    x = hash_matrix(256, 20, seed=2)
    deep = ppr_hops_from_laplacian(LaplacianGraphOp(20, r=0.5).propagate(g, x), 0.25)
    own = PprGraphOp(20, r=0.5, alpha=0.25).propagate(g, x)
    assert len(deep) == 21 and all(oracle.parity_ok(a_.cpu().numpy(), b_.cpu().numpy(), TOL) for a_, b_ in zip(deep, own))
    # the kernel: out_k = sum_j W[k, j] X_j
    rng = np.random.default_rng(11)
    for n, d, n_in, n_out in ((300, 100, 11, 10), (257, 147, 6, 3), (64, 3, 16, 16), (1, 37, 1, 2), (500, 128, 4, 20)):
        host = [rng.standard_normal((n, d)).astype(np.float32) for _ in range(n_in)]
        feats = [dev.upload_rows(h_, cuda) for h_ in host]
        w = rng.standard_normal((n_out, n_in)).astype(np.float32)
        w[rng.random((n_out, n_in)) < 0.3] = 0.0
        outs = dev.hop_lincomb(feats, w)
        want = np.einsum("kj,jnd->knd", w.astype(np.float64), np.stack(host).astype(np.float64))
        scale = np.einsum("kj,jnd->knd", np.abs(w).astype(np.float64), np.abs(np.stack(host)).astype(np.float64))
        for k in range(n_out):
            assert np.abs(outs[k].cpu().numpy() - want[k]).max() <= 1e-6 * max(scale[k].max(), 1e-30), (n, d, n_in, k)
            if outs[k].stride(0) != d and n > 1:
                assert float(dev.padded_parent(outs[k])[:, d:].abs().max()) == 0.0
    feats[1].fill_(float("nan"))                                          # a hop with weight 0 cannot contaminate an output
    w = np.zeros((2, 4), np.float32)
    w[0, 0], w[1, 1], w[1, 2] = 1.0, 2.0, 1.0
    o = dev.hop_lincomb(feats, w)
    assert torch.equal(o[0], feats[0]) and bool(torch.isnan(o[1]).all())
    with pytest.raises(_lib.SglHipError):
        dev.hop_lincomb(feats, w, outs=[feats[0], dev.alloc_rows(500, 128, cuda)])       # an output aliasing an input


@pytest.mark.parametrize("d", [16, 128])
def test_config5_every_message_op_over_eleven_hops(goldens, cuda, d):
    """every op of sgl/operators/message_op/ over H = 11 hop matrices (k = 10) against reference-generated goldens: stateless ops
    bit-exact, weighted / NAFS within tolerance, the five learnable kinds (jk: Linear(d + 11 d, 1), learnable_weighted_messahe_op.py:
    56-57) + the iterate op with parameter and input gradients, projected concat"""
    from sgl_amd.operators import message_op as mo
    g9 = goldens.npz("g9_config5")
    g12 = goldens.npz("g12_fp64_truth")
    P, H = f"agg|d{d}|", 11
    n = goldens.json("g9_config5")["agg"]["dims"][str(d)]

    def feats(requires_grad=False):
        fs = [torch.from_numpy(g9[P + f"feat{j}"]).to(cuda) for j in range(H)]
        return [f.clone().requires_grad_(True) for f in fs] if requires_grad else fs

    fs = feats()
    assert np.array_equal(mo.LastMessageOp().aggregate(fs).cpu().numpy(), g9[P + "last"])
    for (s, e) in ((0, H), (1, H - 1)):
        for name, cls in (("concat", mo.ConcatMessageOp), ("mean", mo.MeanMessageOp), ("sum", mo.SumMessageOp),
                          ("max", mo.MaxMessageOp), ("min", mo.MinMessageOp)):
            assert np.array_equal(cls(s, e).aggregate(fs).cpu().numpy(), g9[P + f"{name}|{s}_{e}"]), (name, s, e)
    for (s, e) in ((0, H), (1, H)):
        y = mo.SimpleWeightedMessageOp(s, e, "alpha", 0.85).aggregate(fs).cpu().numpy()
        assert oracle.parity_ok(y, g9[P + f"simple_weighted|alpha0.85|{s}_{e}"], 1e-6)
    rep = oracle.parity_report(mo.OverSmoothDistanceWeightedOp().aggregate(fs).cpu().numpy(), g9[P + "over_smooth"], TOL)
    assert rep["ok"], rep
    gout = torch.from_numpy(hash_matrix(n, d, seed=778)).to(cuda)
    stored = goldens.json("g9_config5")["agg"]["dfeat_stored"]

    def check_learnable(tag, op):
        op.load_state_dict({k[len(tag) + 7:]: torch.from_numpy(v) for k, v in g9.items() if k.startswith(tag + "|param|")})
        op = op.to(cuda)
        fg = feats(requires_grad=True)
        y = op.aggregate(fg)
        rep = oracle.parity_report(y.detach().cpu().numpy(), g9[tag + "|out"], TOL)
        assert rep["ok"], (tag, rep)
        rep = oracle.truth_report(y.detach().cpu().numpy(), g9[tag + "|out"], g12["g9|" + tag + "|out"])     # derived from the fp64 truth
        assert rep["ok"], (tag, "out", rep)
        (y * gout).sum().backward()
        for name, p in op.named_parameters():
            rep = oracle.truth_report(p.grad.cpu().numpy(), g9[tag + "|grad|" + name], g12["g9|" + tag + "|grad|" + name],
                                      cond=g12.get("g9|" + tag + "|gradcond|" + name))
            assert rep["ok"], (tag, name, rep)
        grads = [(f.grad if f.grad is not None else torch.zeros_like(f)).cpu().numpy() for f in fg]
        for j in stored:
            rep = oracle.truth_report(grads[j], g9[tag + f"|dfeat{j}"], g12["g9|" + tag + f"|dfeat{j}"])
            assert rep["ok"], (tag, j, rep)
        sums = np.array([gr.astype(np.float64).sum() for gr in grads])
        assert np.allclose(sums, g9[tag + "|dfeat_sums"], rtol=0, atol=2e-4 * np.maximum(g9[tag + "|dfeat_abs_sums"], 1e-6)), tag

    for kind, args in (("simple", (10,)), ("simple_allow_neg", (10,)), ("gate", (d,)), ("ori_ref", (d,)), ("jk", (10, d))):
        for (s, e) in ((0, H), (1, H)):
            check_learnable(P + f"learnable|{kind}|{s}_{e}", mo.LearnableWeightedMessageOp(s, e, kind, *args))
    check_learnable(P + f"iterate|0_{H}", mo.IterateLearnableWeightedMessageOp(0, H, "recursive", d))
    tag = P + f"proj_concat|0_{H}"
    pc = mo.ProjectedConcatMessageOp(0, H, d, 8, 2)
    pc.load_state_dict({k[len(tag) + 7:]: torch.from_numpy(v) for k, v in g9.items() if k.startswith(tag + "|param|")})
    pc = pc.to(cuda).eval()
    with torch.no_grad():
        y = pc.aggregate(feats())
    rep = oracle.parity_report(y.cpu().numpy(), g9[tag + "|out"], 1e-4)
    assert rep["ok"], rep


def test_gather_rows(cuda):
    from sgl_amd.device import alloc_rows, gather_rows
    for d in (100, 147, 3, 500):
        x = alloc_rows(1000, d, cuda)
        x.copy_(torch.from_numpy(hash_matrix(1000, d, seed=d)))
        idx = torch.randint(0, 1000, (333,), generator=torch.Generator().manual_seed(1))
        assert torch.equal(gather_rows(x, idx).cpu(), x.cpu()[idx])
        assert torch.equal(gather_rows(x, range(5, 50)).cpu(), x.cpu()[5:50])
        assert torch.equal(gather_rows(x, np.array([-1, 0, 999])).cpu(), x.cpu()[[-1, 0, 999]])
        assert gather_rows(x, []).shape == (0, d)
    with pytest.raises(IndexError):
        gather_rows(x, [1000])


def test_gather_and_scatter_rows_many_rows_per_thread(cuda):
    """large index lists take the 4-rows-per-thread variant (ragged last block included); scatter_rows = the pack step of the
    need-aware exchange: (own row, send-buffer row) pairs in own-row order give the buffer the peer-ordered gather gives"""
    from sgl_amd.device import gather_rows, scatter_rows
    g = torch.Generator(device=cuda).manual_seed(3)
    for d, n_src, n_idx in ((64, 5000, 300_001), (36, 7777, 270_003), (100, 3001, 140_007), (7, 900, 600_005)):
        x = torch.randn((n_src, d), device=cuda, generator=g)
        idx = torch.randint(0, n_src, (n_idx,), device=cuda, generator=g)
        want = x.index_select(0, idx)
        assert torch.equal(gather_rows(x, idx), want)
        src, dst = torch.sort(idx, stable=True)
        out = torch.full((n_idx + 5, d), -1.0, device=cuda)
        scatter_rows(x, src, dst, out)
        assert torch.equal(out[:n_idx], want) and bool((out[n_idx:] == -1.0).all())
    with pytest.raises(ValueError):
        scatter_rows(x, src, dst[:-1].contiguous(), out)
    assert scatter_rows(x, src[:0], dst[:0], out) is out


def test_models_match_reference_goldens(goldens, cuda):
    from sgl_amd.models import homo
    g4 = goldens.npz("g4_models")
    g12 = goldens.npz("g12_fp64_truth")
    g = goldens.graph("pl2000")
    n, d, C, K = 2000, 16, 5, 3
    x = hash_matrix(n, d, seed=4242)
    idx = g4["idx"]
    ctor = {"SGC": (K, d, C), "SSGC": (K, d, C), "SIGN": (K, d, C, 32, 2), "GBP": (K, d, C, 32, 2),
            "GAMLP": (K, d, C, 32, 2), "GAMLPRecursive": (K, d, C, 32, 2), "NAFS": (K, d, C),
            "PASCA_V1": (K, d, C, 32, 3), "PASCA_V2": (K, d, C, 32, 3), "PASCA_V3": (K, 2, d, C, 32, 3)}
    for name, args in ctor.items():
        model = getattr(homo, name)(*args)
        model.load_state_dict({k.split("|param|")[1]: torch.from_numpy(v) for k, v in g4.items() if k.startswith(name + "|param|")})
        model = model.to(cuda).eval()
        model.preprocess(g, x)
        with torch.no_grad():
            y = model.model_forward(idx, cuda)
        # logits: at most twice as far from the float64 truth (G12: the reference's model in .double() on float64 hops) as the
        # reference's own float32 logits are -- and inside the 1e-5 contract against the reference itself
        rep = oracle.truth_report(y.cpu().numpy(), g4[f"{name}|out"], g12[f"g4|{name}|out"])
        assert rep["ok"], (name, rep)
        rep = oracle.parity_report(y.cpu().numpy(), g4[f"{name}|out"], TOL, rowwise=False)
        assert rep["ok"], (name, rep)
        if name == "PASCA_V3":
            with torch.no_grad():
                post = model.postprocess(g, model.model_forward(range(n), cuda))
            rep = oracle.truth_report(post.cpu().numpy()[idx], g4["PASCA_V3|post"], g12["g4|PASCA_V3|post"])
            assert rep["ok"], (name, "post", rep)
    # training step through the learnable aggregator on device
    model = homo.GAMLP(K, d, C, 32, 2).to(cuda)
    model.preprocess(g, x)
    out = model.model_forward(range(0, 512), cuda)
    out.logsumexp(1).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_forward_over_a_contiguous_row_range_reads_the_hops_in_place(goldens, cuda):
    """model_forward(range(a, b)) -- a full prediction pass in batches -- takes VIEWS of the device hop matrices (no row gather) and
    gives what the gathered path gives, for a learnable aggregator (hop list) and a fixed one (one aggregated matrix)"""
    from sgl_amd.models.base_model import take_rows
    from sgl_amd.models.homo import GAMLPRecursive, SSGC
    g = goldens.graph("pl2000")
    n, d, C = 2000, 37, 5
    x = hash_matrix(n, d, seed=9)
    for model in (GAMLPRecursive(3, d, C, 16, 2), SSGC(3, d, C)):
        model = model.to(cuda).eval()
        model.preprocess(g, x)
        feats = model._processed_feat_list if model._pre_msg_learnable else [model._processed_feature]
        assert all(f.is_cuda for f in feats)
        v = take_rows(feats[0], range(100, 300), cuda)
        assert v.data_ptr() == feats[0][100:300].data_ptr() and v.shape == (200, feats[0].shape[1])
        with torch.no_grad():
            a = model.model_forward(range(100, 300), cuda)
            b = model.model_forward(torch.arange(100, 300, device=cuda), cuda)
            c = model.model_forward(range(0, n), cuda)
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6) and torch.allclose(c[100:300], b, rtol=1e-5, atol=1e-6)
        assert take_rows(feats[0], range(0, 10, 2), cuda).data_ptr() != feats[0].data_ptr()      # a strided range is gathered


@pytest.mark.parametrize("delta", [False, True])
def test_label_reuse_loop_matches_reference_task(goldens, cuda, delta, monkeypatch):
    """BASELINE config 3's loop (tasks/node_classification_with_label_use.py:58-137: label use every epoch, label reuse from epoch 1
    on, `preprocess` re-run after every write-back) on the device -- sgl_amd.tricks.add_labels / label_reuse, GAMLP with the
    reference's saved parameters, d + C = 147 columns, K = 5 -- against G10, which the reference's own task code recorded: after
    EVERY preprocess call the label columns (sampled rows), the column sums of the feature matrix and the fp64 sums of all six hop
    matrices; at the end hop 1 / 3 / 5 rows and the logits.  Nothing leaves the GPU inside the loop."""
    from sgl_amd import config
    from sgl_amd.models.homo import GAMLP
    from sgl_amd.tricks import add_labels, label_reuse
    # delta: every preprocess call after the first re-propagates only the label columns (config.delta_propagate, forced on for this
    # small graph) -- the same goldens must hold; without it every call propagates all d + C columns
    monkeypatch.setattr(config, "delta_propagate", delta)
    monkeypatch.setattr(config, "delta_propagate_min_mb", 0.0)
    deltas = []
    g10 = goldens.npz("g10_label_reuse")
    g12 = goldens.npz("g12_fp64_truth")
    n, d, C, K = (int(g10[k]) for k in ("n", "d", "C", "K"))
    g = goldens.graph("pl2000")
    x = torch.from_numpy(hash_matrix(n, d, seed=1010)).to(cuda)
    labels = torch.from_numpy(g10["labels"]).to(cuda)
    sub = torch.from_numpy(g10["sub_rows"]).to(cuda)
    model = GAMLP(K, d + C, C, 64, 2)
    model.load_state_dict({k[len("param|"):]: torch.from_numpy(v) for k, v in g10.items() if k.startswith("param|")})
    model = model.to(cuda).eval()
    train = set(g10["train_idx"].tolist())
    calls = []
    real_pre = model.preprocess

    def spy(adj, features):
        i = len(calls)
        assert features.is_cuda and features.dtype == torch.float32          # the loop never went through the host
        got_cols = features[sub, d:].cpu().numpy()
        # derived tolerance: at most twice the reference's own float32 distance from the loop run in float64 (G12)
        rep = oracle.truth_report(got_cols, g10[f"call{i}|label_cols_sub"], g12[f"g10|call{i}|label_cols_sub"])
        assert rep["ok"], ("label columns before preprocess call", i, rep)
        cs = features.double().sum(0).cpu().numpy()
        assert np.allclose(cs, g10[f"call{i}|feature_colsum"], rtol=1e-4, atol=1e-2), i
        real_pre(adj, features)
        deltas.append(model._pre_graph_op.delta_info)
        sums = np.array([h.double().sum().item() for h in model._processed_feat_list])
        assert len(sums) == K + 1 and np.allclose(sums, g10[f"call{i}|hop_sums"], rtol=1e-4, atol=1e-2), (i, sums, g10[f"call{i}|hop_sums"])
        calls.append(i)
    model.preprocess = spy
    feats = None
    for epoch in range(3):
        lab_idx = g10[f"epoch{epoch}|train_labels_idx"]
        feats = add_labels(x, labels, torch.from_numpy(lab_idx), C, out=feats, device=cuda)
        model.preprocess(g, feats)
        if epoch > 0:
            train_pred = np.array(sorted(train - set(lab_idx.tolist())))
            # the reference's order: train nodes whose labels are hidden this epoch, then validation, then test nodes
            mask = np.isin(g10["train_idx"], lab_idx)
            unlabeled = np.concatenate([g10["train_idx"][~mask], g10["val_idx"], g10["test_idx"]])
            assert set(unlabeled[: (~mask).sum()].tolist()) == set(train_pred.tolist())
            label_reuse(model, g, feats, unlabeled, C, 2, device=cuda, batch_size=700)
    assert len(calls) == 7
    if delta:      # every call but the first found the previous hop matrices and re-propagated the label columns only
        assert deltas[0] is None and all(di is not None and di["columns_propagated"] == (d // 4 * 4, (d + C + 3) // 4 * 4) for di in deltas[1:]), deltas
    else:
        assert all(di is None for di in deltas)
    hops = model._processed_feat_list
    for h in (1, 3, 5):
        rep = oracle.truth_report(hops[h][sub].cpu().numpy(), g10[f"final|hop{h}_sub"], g12[f"g10|final|hop{h}_sub"])
        assert rep["ok"], (h, rep)
        assert oracle.parity_ok(hops[h][sub].cpu().numpy(), g10[f"final|hop{h}_sub"], TOL, rowwise=False), h
    with torch.no_grad():
        logits = model.model_forward(range(n), cuda)
    rep = oracle.truth_report(logits[sub].cpu().numpy(), g10["final|logits_sub"], g12["g10|final|logits_sub"])
    assert rep["ok"], rep
    assert oracle.parity_ok(logits[sub].cpu().numpy(), g10["final|logits_sub"], TOL, rowwise=False)
    assert np.allclose(logits.double().sum(0).cpu().numpy(), g10["final|logits_colsum"], rtol=1e-3, atol=1e-2)


def test_column_signature_matches_its_definition(cuda):
    """sgl_col_signature_f32 against a numpy statement of its definition: sig[c] = sum_r mix(bits(X[r, c]), r) mod 2^64"""
    def mix(bits, r):
        with np.errstate(over="ignore"):
            h = (bits.astype(np.uint64) ^ (r.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))) * np.uint64(0xBF58476D1CE4E5B9)
            h ^= h >> np.uint64(31)
            h *= np.uint64(0x94D049BB133111EB)
            return h ^ (h >> np.uint64(29))
    for n, d in ((1, 4), (7, 3), (1000, 147), (4097, 100), (300, 1030), (513, 64)):
        xp = dev.alloc_rows(n, d, cuda)
        xp.copy_(torch.from_numpy(hash_matrix(n, d, seed=n + d)).to(cuda))
        sig = dev.column_signature(xp)
        dw = (d + 3) // 4 * 4
        assert sig is not None and sig.numel() == dw
        full = dev.padded_parent(xp)[:, :dw].cpu().numpy()
        with np.errstate(over="ignore"):
            want = mix(full.view(np.uint32), np.arange(n, dtype=np.uint64)[:, None]).sum(axis=0, dtype=np.uint64)
        assert np.array_equal(sig.cpu().numpy().view(np.uint64), want), (n, d)
        # one element moved -> exactly its column's signature moves; two rows swapped -> every column that differs between them
        if n > 1:
            xp[n // 2, d - 1] += 1.0
            sig2 = dev.column_signature(xp)
            assert torch.nonzero(sig2 != sig).flatten().tolist() == [d - 1]
    assert dev.column_signature(torch.zeros((5, 3), device=cuda)) is None          # 12-byte rows: not 16-byte vectors


@pytest.mark.parametrize("strict", [True, False])
def test_delta_propagate_recomputes_only_the_changed_columns(goldens, cuda, strict, monkeypatch):
    """config.delta_propagate: the second propagate() over one adjacency re-propagates the column range whose content changed and
    copies the rest from the hop matrices of the first call -- into fresh tensors, bit-identical to a full propagation in strict
    order, within the tolerance otherwise; every reason not to trust the previous hop matrices leads to a full propagation."""
    from sgl_amd import config
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    monkeypatch.setattr(config, "delta_propagate", True)
    monkeypatch.setattr(config, "delta_propagate_min_mb", 0.0)
    g = goldens.graph("pl2000")
    n, d, C, K = g.shape[0], 147, 47, 5
    x = dev.alloc_rows(n, d, cuda)
    x.copy_(torch.from_numpy(hash_matrix(n, d, seed=77)).to(cuda))

    def full(xx, cls=LaplacianGraphOp, **kw):
        monkeypatch.setattr(config, "delta_propagate", False)
        try:
            return cls(K, strict_order=strict, **kw).propagate(g, xx.clone())
        finally:
            monkeypatch.setattr(config, "delta_propagate", True)

    for cls, kw in ((LaplacianGraphOp, {}), (PprGraphOp, {"alpha": 0.2})):
        x.copy_(torch.from_numpy(hash_matrix(n, d, seed=77)).to(cuda))
        op = cls(K, strict_order=strict, **kw)
        h1 = op.propagate(g, x)
        assert op.delta_info is None
        keep1 = [h.clone() for h in h1]
        x[:, d - C:] = torch.from_numpy(hash_matrix(n, C, seed=78)).to(cuda)          # in place, like features[unlabeled, -C:] = ...
        h2 = op.propagate(g, x)
        assert op.delta_info == {"columns_propagated": (100, 148), "columns_changed": C, "of": d}
        want = full(x, cls, **kw)
        for k in range(K + 1):
            assert h2[k].data_ptr() not in [t.data_ptr() for t in h1[1:]]              # fresh tensors
            if strict:
                assert torch.equal(h2[k], want[k]), (cls.__name__, k)
            else:
                assert torch.equal(h2[k][:, :d - C], want[k][:, :d - C]), (cls.__name__, k)    # copied columns: the same bits
                assert oracle.parity_ok(h2[k].cpu().numpy(), want[k].cpu().numpy(), TOL), (cls.__name__, k)
            if k:
                assert torch.equal(h1[k], keep1[k])                                     # nothing the caller holds was written
                assert float(dev.padded_parent(h2[k])[:, d:].abs().max()) == 0.0        # pad columns stay zero
        # unchanged X: a copy of the previous hops
        h3 = op.propagate(g, x)
        assert op.delta_info["columns_propagated"] == (0, 0) and all(torch.equal(a, b) for a, b in zip(h3, h2))
        # one column in the middle
        x[:, 50] += 1.0
        h4 = op.propagate(g, x)
        assert op.delta_info["columns_propagated"] == (48, 52)
        want = full(x, cls, **kw)
        assert all(oracle.parity_ok(a.cpu().numpy(), b.cpu().numpy(), TOL) for a, b in zip(h4, want))
        assert (not strict) or all(torch.equal(a, b) for a, b in zip(h4, want))
        # reasons for a full propagation: the caller wrote into a previous hop matrix; the previous list is gone; too many columns
        h4[2].mul_(1.0)
        x[:, 50] += 1.0
        h5 = op.propagate(g, x)
        assert op.delta_info is None
        del h1, h2, h3, h4, h5
        x[:, 50] += 1.0
        h6 = op.propagate(g, x)
        assert op.delta_info is None
        x[:, ::8] += 1.0                                                                # columns 0, 8, ..., 144: the range is everything
        h7 = op.propagate(g, x)
        assert op.delta_info is None
        want = full(x, cls, **kw)
        assert all(oracle.parity_ok(a.cpu().numpy(), b.cpu().numpy(), TOL) for a, b in zip(h7, want))
        # another adjacency object with the same operator: nothing of the old chain may be reused
        g2 = (g + sp.identity(n, dtype=g.dtype, format="csr")).tocsr()
        x[:, d - 1] += 1.0
        h8 = op.propagate(g2, x)
        assert op.delta_info is None
        monkeypatch.setattr(config, "delta_propagate", False)
        want = cls(K, strict_order=strict, **kw).propagate(g2, x.clone())
        monkeypatch.setattr(config, "delta_propagate", True)
        assert all(oracle.parity_ok(a.cpu().numpy(), b.cpu().numpy(), TOL) for a, b in zip(h8, want))
        # the same handle re-weighted behind the operator's back (sgl_csr_set_values): the previous chain belongs to other values
        op._adj.set_values(op._adj.val * 0.5)
        x[:, d - 1] += 1.0
        h9 = op.propagate(g2, x)
        assert op.delta_info is None
        assert oracle.parity_ok(h9[1].cpu().numpy() * 2.0, cls(1, strict_order=strict, **kw).propagate(g2, x.clone())[1].cpu().numpy(), TOL)
        del h6, h7, h8, h9


def test_spmm_axpb_clamp_epilogue(cuda):
    """fused Y = clamp(alpha * A X + RES): regular rows in the main kernel, split rows in the fix-up kernel"""
    a = long_row_graph()
    n = a.shape[0]
    for d in (5, 47, 100):
        x = hash_matrix(n, d, seed=13)
        res = hash_matrix(n, d, seed=14)
        ax = oracle.oracle_spmm(a.indptr, a.indices, a.data, x)
        scale = oracle.oracle_spmm(a.indptr, a.indices, np.abs(a.data), np.abs(x))
        for strict, long_nnz in ((True, 0), (False, 256)):
            csr = device_csr(a.indptr, a.indices, a.data, (n, n), cuda, strict=strict, long_row_nnz=long_nnz)
            xd, rd = torch.from_numpy(x).to(cuda), torch.from_numpy(res).to(cuda)
            ref = np.clip((np.float32(0.8) * ax + res).astype(np.float32), np.float32(-0.5), np.float32(0.75))
            y = csr.spmm_axpb_clamp(xd, 0.8, rd, -0.5, 0.75).cpu().numpy()
            if strict:
                assert np.array_equal(y, ref), (d, oracle.parity_report(y, ref))
            else:
                # the clamp hides the magnitude the sums were formed at: normalise by the pre-clamp values
                pre = np.abs(np.float32(0.8) * ax + res).max()
                assert np.abs(y - ref).max() <= TOL * pre and y.min() >= -0.5 and y.max() <= 0.75
            y = csr.spmm_axpb_clamp(xd, -1.5).cpu().numpy()                      # no residual, no clamp
            ref2 = (np.float32(-1.5) * ax).astype(np.float32)
            assert np.array_equal(y, ref2) if strict else oracle.parity_ok(y, ref2, TOL, scale=1.5 * scale)


def test_label_propagation_and_correct_smooth_match_reference(goldens, cuda):
    from sgl_amd.tricks import CorrectAndSmooth, label_propagation
    g6 = goldens.npz("g6_consumers")
    g = goldens.graph("pl2000")
    n = g.shape[0]
    ptr, col, val = oracle.sym_norm_csr(g.indptr, g.indices, g.data, n, 0.5)
    adj = sp.csr_matrix((val, col, ptr.astype(np.int32)), shape=(n, n))          # what the reference passes in
    lab = torch.from_numpy(g6["lp|labels"])
    mask = g6["lp|mask"]
    y = label_propagation(lab, adj, 5, 0.8, mask=torch.from_numpy(mask))
    assert y.is_cuda and oracle.parity_ok(y.cpu().numpy(), g6["lp|long_masked"], TOL, rowwise=False)
    y = label_propagation(lab, adj, 3, 0.5)
    assert oracle.parity_ok(y.cpu().numpy(), g6["lp|long_nomask"], TOL, rowwise=False)
    soft = torch.softmax(torch.from_numpy(hash_matrix(n, 5, seed=32)) * 3, 1)
    y = label_propagation(soft - 0.3, adj, 4, 0.9, post_process=(-1., 1.))
    assert oracle.parity_ok(y.cpu().numpy(), g6["lp|float_clamp11"], TOL, rowwise=False)
    y = label_propagation(soft - 0.3, adj, 4, 0.9, post_process=lambda t: t.clamp_(-1., 1.))   # callable form
    assert oracle.parity_ok(y.cpu().numpy(), g6["lp|float_clamp11"], TOL, rowwise=False)
    for autoscale in (True, False):
        cs = CorrectAndSmooth(4, 0.9, 3, 0.7, autoscale=autoscale, scale=1.5)
        y1 = cs.correct(soft.clone(), lab, mask, adj)
        rep = oracle.parity_report(y1.cpu().numpy(), g6[f"cs|autoscale{int(autoscale)}|correct"], TOL, rowwise=False)
        assert rep["ok"], (autoscale, rep)
        y2 = cs.smooth(y1.clone(), lab, mask, adj)
        rep = oracle.parity_report(y2.cpu().numpy(), g6[f"cs|autoscale{int(autoscale)}|smooth"], TOL, rowwise=False)
        assert rep["ok"], (autoscale, rep)


def test_nafs_task_feature_pipeline_matches_reference(goldens, cuda):
    from inputs import hash_positive
    from sgl_amd.tricks import nafs_ensemble_features
    g6 = goldens.npz("g6_consumers")
    g = goldens.graph("pl256")
    x = hash_positive(256, 8, seed=33)
    for method in ("mean", "max", "concat", "simple"):
        y = nafs_ensemble_features(g, x, 3, [0.5, 0.4, 0.3, 0.2, 0.1, 0], method)
        rep = oracle.parity_report(y.cpu().numpy(), g6[f"nafs_task|{method}|hops3"], TOL)
        assert rep["ok"], (method, rep)
    with pytest.raises(ValueError):
        nafs_ensemble_features(g, x, 3, method="median")
    # the adjacency already on the device: identical result, and nothing of it crosses PCIe again (one upload serves all r;
    # the per-r normalisations re-use one SpMM plan through sgl_csr_set_values)
    from sgl_amd.io import DeviceAdjacency
    dadj = DeviceAdjacency.from_scipy(g, device=cuda)
    y_host = nafs_ensemble_features(g, x, 3, [0.5, 0.3, 0], "mean")
    assert torch.equal(nafs_ensemble_features(dadj, torch.from_numpy(x).to(cuda), 3, [0.5, 0.3, 0], "mean"), y_host)
    # rows processed in the plan-time community order (one order for all r): the same features (d = 8 uses a packed lane
    # layout: to rounding; strict order: bit for bit)
    y_re = nafs_ensemble_features(dadj, x, 3, [0.5, 0.3, 0], "mean", reorder="community")
    assert oracle.parity_ok(y_re.cpu().numpy(), y_host.cpu().numpy(), 1e-5)
    assert torch.equal(nafs_ensemble_features(dadj, x, 3, [0.5, 0.3, 0], "mean", strict_order=True, reorder="community"),
                       nafs_ensemble_features(dadj, x, 3, [0.5, 0.3, 0], "mean", strict_order=True))
    n, ptr, col, val = norm_graph(goldens, "pl256", r=0.5)
    csr = device_csr(ptr, col, val, (n, n), cuda)
    xd = torch.from_numpy(x).to(cuda)
    y1 = csr.spmm(xd)
    _, _, _, val2 = norm_graph(goldens, "pl256", r=0.2)
    v2 = torch.from_numpy(val2).to(cuda)
    assert torch.equal(csr.set_values(v2).spmm(xd), device_csr(ptr, col, val2, (n, n), cuda).spmm(xd)) and not torch.equal(y1, csr.spmm(xd))


def test_nafs_hop_sweep_in_one_propagation(goldens, cuda):
    """nafs_ensemble_sweep: the feature matrices of EVERY hop count of the NAFS task's sweep from one propagation per r
    (sgl_nafs_prefix_f32: running numerator / denominator, every hop element read once) against what the reference hands to
    KMeans for each hop count separately (G11, 1e-5), every ensemble method; then the kernel itself at the widths of every lane
    layout against the per-prefix kernel, with the ensemble combinations."""
    from sgl_amd.tricks import nafs_ensemble_features, nafs_ensemble_sweep
    g11 = goldens.npz("g11_nafs_sweep")
    g = goldens.graph("pl256")
    x, hops, r_list = g11["x"], [int(h) for h in g11["hops"]], [float(r) for r in g11["r_list"]]
    for method in ("mean", "max", "concat", "simple"):
        sweep = nafs_ensemble_sweep(g, x, hops, r_list, method)
        assert sorted(sweep) == hops
        for h in hops:
            rep = oracle.parity_report(sweep[h].cpu().numpy(), g11[f"nafs_sweep|{method}|hops{h}"], TOL)
            assert rep["ok"], (method, h, rep)
            # and the one-hop-count entry point (one propagation + the register-resident kernel) agrees
            assert oracle.parity_ok(sweep[h].cpu().numpy(), nafs_ensemble_features(g, x, h, r_list, method).cpu().numpy(), TOL)
    # a width that is not a multiple of 4: the concat ensemble cannot write 16-byte aligned column slices and assembles per-r matrices
    x7 = np.ascontiguousarray(x[:, :7])
    want7 = oracle.nafs_task_sweep(g.indptr, g.indices, g.data, 256, x7, [0, 2, 5], r_list, "concat")
    got7 = nafs_ensemble_sweep(g, x7, [5, 0, 2], r_list, "concat")
    assert all(oracle.parity_ok(got7[h].cpu().numpy(), want7[h], TOL) for h in (0, 2, 5))
    seen = []
    res = nafs_ensemble_sweep(g, x, 4, r_list, "mean", consume=lambda h, f: seen.append(h) or float(f.sum()))     # int = range(hops)
    assert seen == [0, 1, 2, 3] and sorted(res) == seen and all(isinstance(v, float) for v in res.values())
    with pytest.raises(ValueError):
        nafs_ensemble_sweep(g, x, [1, 64], r_list, "mean")
    with pytest.raises(ValueError):
        nafs_ensemble_sweep(g, x, [1], r_list, "median")
    # the kernel at every lane layout (8 / 16 / 32 / 64 lanes x 1, 16 x 3, 32 x 2, 64 x 2), more hops than the fused kernel holds
    n, ptr, col, val = norm_graph(goldens, "pl2000", r=0.5)
    csr = device_csr(ptr, col, val, (n, n), cuda)
    for d in (3, 20, 36, 100, 128, 147, 200, 300, 500):
        feats = [dev.upload_rows(hash_matrix(n, d, seed=d), cuda)]
        for _ in range(21):
            feats.append(csr.spmm(feats[-1]))
        emit = [0, 1, 2, 5, 11, 20, 21]
        outs = dev.nafs_prefix(feats, emit)
        ref_feats = [f.cpu().numpy() for f in feats]
        for h, o in zip(emit, outs):
            want = oracle.agg_over_smooth_distance(ref_feats[:h + 1])
            assert oracle.parity_ok(o.cpu().numpy(), want, TOL, rowwise=False), (d, h)
            if h + 1 <= 16:
                assert oracle.parity_ok(o.cpu().numpy(), dev.nafs_aggregate(feats[:h + 1]).cpu().numpy(), 1e-6, rowwise=False), (d, h)
            assert float(dev.padded_parent(o)[:, d:].abs().max()) == 0.0 if dev.padded_parent(o).shape[1] > d else True
        # ensemble combinations: add, add + divide, max -- against torch on the stored outputs
        base = [o.clone() for o in outs]
        acc = [dev.alloc_rows(n, d, cuda) for _ in emit]
        for a, b in zip(acc, base):
            a.copy_(b * 0.5 - 1.0)
        keep = [a.clone() for a in acc]
        dev.nafs_prefix(feats, emit, outs=acc, combine=dev.NAFS_ADD)
        assert all(torch.equal(a, k + b) for a, k, b in zip(acc, keep, base))
        dev.nafs_prefix(feats, emit, outs=acc, combine=dev.NAFS_ADD_DIV, divisor=3.0)
        # (true division like the reference's CPU `sum(...) / len(...)`; torch on the GPU multiplies by the reciprocal instead)
        assert all(torch.equal(a.cpu(), ((k.cpu() + b.cpu()) + b.cpu()) / 3.0) for a, k, b in zip(acc, keep, base))
        for a, k in zip(acc, keep):
            a.copy_(k)
        dev.nafs_prefix(feats, emit, outs=acc, combine=dev.NAFS_MAX)
        assert all(torch.equal(a, torch.maximum(k, b)) for a, k, b in zip(acc, keep, base))
    with pytest.raises(ValueError):
        dev.nafs_prefix(feats, [3, 2])
    with pytest.raises(_lib.SglHipError):
        wide = [dev.upload_rows(hash_matrix(64, 516, seed=1), cuda)] * 2
        dev.nafs_prefix(wide, [1])
    # the C entry point refuses what it cannot do, with a message (no kernel is launched): a prefix beyond the hop list, an unknown
    # combination, a zero divisor, an output that is not 16-byte aligned
    small = [dev.alloc_rows(8, 12, cuda).normal_() for _ in range(3)]
    out = dev.alloc_rows(8, 12, cuda)
    ptrs, lds = _lib.hop_arrays(small)
    optrs, olds = _lib.hop_arrays([out])

    def raw(mask, combine=0, divisor=1.0, op=optrs, ol=olds):
        return _lib.lib().sgl_nafs_prefix_f32(3, ptrs, lds, mask, op, ol, 0, combine, divisor, 8, 12, _lib.current_stream_ptr())
    assert raw(0b100) == 0
    for bad in (raw(0b1000), raw(0), raw(0b1, combine=4), raw(0b1, combine=2, divisor=0.0)):
        assert bad != 0 and _lib.last_error()
    off = torch.empty(8 * 16 + 1, device=cuda)[1:].view(8, 16)[:, :12]            # rows 4 bytes off a 16-byte boundary
    op2, ol2 = _lib.hop_arrays([off])
    assert raw(0b1, op=op2, ol=ol2) != 0 and "16-byte aligned" in _lib.last_error()


def test_ingest_raw_files_to_device_adjacency(goldens, cuda, tmp_path):
    """Custom_Homo raw layout -> device COO->CSR build (sgl_coo_to_csr) == the reference's Edge/scipy build (G7),
    and a DeviceAdjacency drives GraphOp.propagate without touching the host"""
    from sgl_amd import io
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    g7 = goldens.npz("g7_ingest")
    n = int(g7["n"])
    x = hash_matrix(n, 20, seed=2)
    io.save_custom_homo_raw(str(tmp_path), g7["row"], g7["col"], g7["data"], x=x, labels=np.arange(n) % 3,
                            train_idx=np.arange(0, n, 2))
    ds = io.load_custom_homo_raw(str(tmp_path), device=cuda)
    adj = ds["adj"]
    assert np.array_equal(adj.rowptr.cpu().numpy(), g7["indptr"]) and np.array_equal(adj.col.cpu().numpy(), g7["indices"])
    v = adj.val.cpu().numpy()
    assert (v != g7["values"]).sum() <= 2 and np.allclose(v, g7["values"], rtol=3e-7, atol=0)
    assert ds["y"].shape == (n,) and len(ds["train_idx"]) == n // 2 and ds["val_idx"] is None
    hops = LaplacianGraphOp(2, strict_order=True).propagate(adj, ds["x"])
    ref = oracle.propagate(oracle.laplacian_adj(g7["indptr"], g7["indices"], v, n, 0.5), x, 2)
    assert oracle.parity_ok(hops[2].cpu().numpy(), ref[2], TOL)
    hops2 = LaplacianGraphOp(2, strict_order=True).propagate(adj.to_scipy(), ds["x"])        # same through scipy
    assert torch.equal(hops[2], hops2[2])
    with pytest.raises(Exception):
        io.coo_to_csr_device([0, n], [0, 1], [1.0, 1.0], n, device=cuda)                       # index out of range
    empty = io.coo_to_csr_device([], [], [], 5, device=cuda)
    assert empty.nnz == 0 and empty.rowptr.cpu().tolist() == [0] * 6


def test_sharded_ingest_builds_only_this_ranks_rows(goldens, cuda, tmp_path):
    """load_custom_homo_raw_sharded: the row blocks of every rank, stitched, are the CSR the whole-matrix ingest builds
    (= Edge's csr_matrix, G7) -- for several world sizes and with chunks so small that every pass takes many of them"""
    from sgl_amd import io
    g7 = goldens.npz("g7_ingest")
    n = int(g7["n"])
    x = hash_matrix(n, 12, seed=5)
    io.save_custom_homo_raw(str(tmp_path), g7["row"], g7["col"], g7["data"], x=x, labels=np.arange(n) % 3)
    whole = io.load_custom_homo_raw(str(tmp_path), device=cuda)["adj"]
    wp, wc, wv = whole.rowptr.cpu().numpy(), whole.col.cpu().numpy(), whole.val.cpu().numpy()
    for world, chunk in ((1, 1 << 20), (3, 97), (4, 1000)):
        parts = [io.load_custom_homo_raw_sharded(str(tmp_path), r, world, device=cuda, chunk_edges=chunk) for r in range(world)]
        b = parts[0]["bounds"]
        assert b[0] == 0 and b[-1] == n and all(np.array_equal(b, q["bounds"]) for q in parts)
        for r, q in enumerate(parts):
            blk = q["block"]
            assert (blk.lo, blk.hi, blk.n) == (int(b[r]), int(b[r + 1]), n) and blk.shape == (blk.hi - blk.lo, n)
            a0, a1 = int(wp[blk.lo]), int(wp[blk.hi])
            assert np.array_equal(blk.rowptr.cpu().numpy(), wp[blk.lo:blk.hi + 1] - a0)
            assert np.array_equal(blk.col.cpu().numpy(), wc[a0:a1])
            assert np.array_equal(blk.val.cpu().numpy(), wv[a0:a1])          # same sort, same summation order: bit-equal
            assert np.array_equal(q["x"], x[blk.lo:blk.hi]) and q["y"].shape == (n,)
        if world > 1:   # nnz-balanced: no block holds more than its share plus one row's worth
            sizes = [q["block"].nnz + q["block"].n_local for q in parts]
            assert max(sizes) <= (whole.nnz + n) / world + int(np.diff(wp).max()) + 1
    with pytest.raises(ValueError):
        io.load_custom_homo_raw_sharded(str(tmp_path), 3, 3, device=cuda)


def test_row_sharded_pieces_on_one_gpu(goldens, cuda):
    """every virtual rank's row pieces (rectangular device CSRs) reproduce the rows of the single-matrix result;
    world = 1 ShardedPropagator == plain k-step propagation (the exchange itself is covered by the gloo tests)"""
    from sgl_amd.dist import ShardedPropagator, all_piece_bounds, device_piece_spmms
    n, ptr, col, val = norm_graph(goldens, "pl2000")
    rp = torch.from_numpy(ptr.astype(np.int64)).to(cuda)
    cc = torch.from_numpy(col.astype(np.int32)).to(cuda)
    vv = torch.from_numpy(val).to(cuda)
    x = torch.from_numpy(hash_matrix(n, 100, seed=6)).to(cuda)
    ref = oracle.propagate((ptr, col, val), x.cpu().numpy(), 3)
    for world, pieces in ((1, 4), (3, 2), (8, 4)):
        pb = all_piece_bounds(ptr, world, pieces)
        assert pb[0, 0] == 0 and pb[-1, -1] == n and (pb[1:, 0] == pb[:-1, -1]).all()
        y = torch.empty_like(x)
        for g in range(world):
            fns, _h = device_piece_spmms(rp, cc, vv, n, pb[g], strict=True)
            for p, f in enumerate(fns):
                r0, r1 = int(pb[g, p]), int(pb[g, p + 1])
                if r1 > r0:
                    f(x, y[r0:r1])
        assert np.array_equal(y.cpu().numpy(), ref[1])
    fns, _h = device_piece_spmms(rp, cc, vv, n, all_piece_bounds(ptr, 1, 4)[0], strict=True)
    hops = ShardedPropagator(fns, all_piece_bounds(ptr, 1, 4), 0, 1, n).propagate(x, 3)
    for h in range(4):
        assert np.array_equal(hops[h].cpu().numpy(), ref[h])


def test_label_propagation_round_by_round_over_row_blocks_equals_the_whole_matrix_kernel(cuda):
    """sgl_reorder_lpa_round (one round of the plan-time label propagation on a ROW BLOCK, global labels in, the block's new labels
    out) run over the blocks of a partition, slices assembled between rounds = what sgl_amd/dist/redistribute.py does with an
    all-gather: the labels' stable sort equals sgl_reorder_community's order on the whole matrix -- also for rows with more than the
    256 neighbours the kernel samples (planted communities + hubs, shuffled ids, unequal blocks incl. an empty one)"""
    import ctypes
    from sgl_amd.reorder import community_order
    rng = np.random.default_rng(5)
    n, bs = 9000, 300
    a = np.repeat(np.arange(n), 10)
    near = np.minimum((a // bs) * bs + rng.integers(0, bs, a.size), n - 1)
    b = np.where(rng.random(a.size) < 0.85, near, rng.integers(0, n, a.size))
    hubs = rng.choice(n, 3, replace=False)
    a = np.concatenate([a, np.repeat(hubs, 900)])
    b = np.concatenate([b, rng.integers(0, n, 2700)])
    m = sp.coo_matrix((np.ones(a.size, np.float32), (a, b)), shape=(n, n)).tocsr()
    m = ((m + m.T + sp.eye(n)) > 0).astype(np.float32).tocsr()
    shuffle = rng.permutation(n)
    P = sp.coo_matrix((np.ones(n, np.float32), (shuffle, np.arange(n))), shape=(n, n)).tocsr()
    m = (P @ m @ P.T).tocsr()
    m.sort_indices()
    assert np.diff(m.indptr).max() > 256
    rp = torch.from_numpy(m.indptr.astype(np.int64)).to(cuda)
    cc = torch.from_numpy(m.indices.astype(np.int32)).to(cuda)
    want, _ = community_order(rp, cc, n)
    bounds = [0, 2500, 2500, 7001, n]
    labels = torch.arange(n, dtype=torch.int32, device=cuda)
    rounds = 8
    for it in range(rounds):
        nxt = torch.empty_like(labels)
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            if hi == lo:
                continue
            brp = (rp[lo:hi + 1] - rp[lo]).contiguous()
            bcc = cc[int(m.indptr[lo]):int(m.indptr[hi])].contiguous()
            out = torch.empty(hi - lo, dtype=torch.int32, device=cuda)
            _lib.check(_lib.lib().sgl_reorder_lpa_round(_lib.ptr(brp), _lib.ptr(bcc), hi - lo, lo, n, _lib.ptr(labels), _lib.ptr(out), it,
                                                        1 if it == rounds - 1 else 0, None, _lib.current_stream_ptr()), "sgl_reorder_lpa_round")
            nxt[lo:hi] = out
        labels = nxt
    perm = torch.argsort(labels, stable=True)
    order = torch.empty(n, dtype=torch.int64, device=cuda)
    order[perm] = torch.arange(n, device=cuda)
    assert torch.equal(order, want)
    # and through the module (one rank: its block is the whole matrix)
    from sgl_amd.dist import RowBlock, sharded_community_order
    got, text = sharded_community_order(RowBlock(0, n, n, rp, cc, torch.ones(cc.numel(), device=cuda)), [0, n])
    assert torch.equal(got, want) and "communities after 8 rounds" in text


def test_row_pieces_cut_long_rows_where_the_whole_matrix_does(cuda):
    """default (non-strict) order: where a long row is cut depends on the matrix's nnz (sgl_csr_create: 32 / 128 / 512 / 2048).  A
    row block of a sharded matrix has fewer non-zeros than the whole and would fall into another bracket; the distributed paths
    pass the threshold of the GLOBAL nnz, so the default-order result of the row pieces is bit-identical to the single-handle
    run (a matrix of ~6e5 non-zeros -- bracket 128 -- cut into 8 pieces of ~7e4 -- bracket 32 on their own)."""
    from sgl_amd.dist import all_piece_bounds, device_piece_spmms
    from sgl_amd.dist.sharded_adj import RowBlock, block_piece_spmms
    from sgl_amd.synthetic import chung_lu_numpy
    n = 30_000
    ip, ix, dt = chung_lu_numpy(n, 290_000, 3000, seed=2)
    ptr, col, val = oracle.laplacian_adj(ip, ix, dt, n, 0.5)
    nnz = int(ptr[-1])
    assert (1 << 18) <= nnz < (1 << 20) and np.diff(ptr).max() > 128
    assert dev.default_long_row_nnz(nnz) == 128 and dev.default_long_row_nnz(nnz // 8) == 32
    rp, cc, vv = (torch.from_numpy(a).to(cuda) for a in (ptr.astype(np.int64), col.astype(np.int32), val.astype(np.float32)))
    x = torch.from_numpy(hash_matrix(n, 100, seed=5)).to(cuda)
    whole = dev.DeviceCSR(rp, cc, vv, (n, n))
    assert whole.info()["n_long_rows"] > 0
    want = whole.spmm(x)
    pb = all_piece_bounds(ptr, 4, 2)
    y = torch.empty_like(want)
    y2 = torch.empty_like(want)
    for g in range(4):
        fns, hs = device_piece_spmms(rp, cc, vv, n, pb[g])
        lo, hi = int(pb[g, 0]), int(pb[g, -1])
        blk = RowBlock(lo, hi, n, (rp[lo:hi + 1] - rp[lo]).contiguous(), cc[int(ptr[lo]):int(ptr[hi])].contiguous(), vv[int(ptr[lo]):int(ptr[hi])].contiguous())
        fns2, hs2, mine = block_piece_spmms(blk, 2, total_nnz=nnz)
        assert [int(b) for b in mine] == [int(b) for b in pb[g]]
        for p in range(2):
            r0, r1 = int(pb[g, p]), int(pb[g, p + 1])
            fns[p](x, y[r0:r1])
            fns2[p](x, y2[r0:r1])
    assert torch.equal(y, want) and torch.equal(y2, want)
    # without the global threshold a block on its own cuts elsewhere: same values to rounding, not the same bits
    fns3, _, _ = block_piece_spmms(blk, 2)
    y3 = torch.empty((int(pb[3, 1]) - int(pb[3, 0]), 100), device=cuda)
    fns3[0](x, y3)
    assert oracle.parity_ok(y3.cpu().numpy(), want[int(pb[3, 0]):int(pb[3, 1])].cpu().numpy(), TOL)


def test_gather_rows_never_copies_a_source_tail_into_the_pad(cuda):
    """x may be a column view of a WIDER matrix whose columns beyond d are data: the gathered rows carry zeros in the pad columns of
    our own output (alloc_rows invariant) whatever lies behind column d of the source, also for a view that starts at a column
    offset (the vector read of the last row must stay inside the storage)."""
    n, d = 500, 147
    big = torch.from_numpy(hash_matrix(n, 160, seed=3)).to(cuda)          # 160 = row_pitch(147): the ambiguous case
    idx = torch.from_numpy(np.random.default_rng(0).integers(0, n, 300)).to(cuda)
    for view in (big[:, :d], big[:, 12:12 + d], big[:, 13:13 + d]):        # aligned start, aligned offset, unaligned offset
        got = dev.gather_rows(view, idx)
        assert torch.equal(got, view[idx])
        if got.stride(0) != d:
            assert float(dev.padded_parent(got)[:, d:].abs().max()) == 0.0
    tail = big[:, 160 - d:]                                              # ends at the end of the storage
    got = dev.gather_rows(tail, idx)
    assert torch.equal(got, tail[idx]) and float(dev.padded_parent(got)[:, d:].abs().max()) == 0.0
    # a caller's output: nothing beyond round_up(d, 4) is touched, the straddling vector's tail is zero
    out = torch.full((300, 160), 7.0, device=cuda)
    dev.gather_rows(big[:, :d], idx, out=out[:, :d])
    assert torch.equal(out[:, :d], big[:, :d][idx]) and bool((out[:, 148:] == 7.0).all()) and bool((out[:, 147:148] == 0.0).all())


def test_legacy_gate_entry_point_keeps_scalar_bias_semantics(cuda):
    """sgl_hop_gate_f32 (un-suffixed): the bias is the scalar passed -- a NaN bias gives NaN outputs, it is NOT read from behind the
    vector (only sgl_hop_gate_padded_f32 has that convention); d_vec of exactly round_up(d, 4) floats is enough"""
    n, d, H = 64, 20, 3
    feats = [dev.alloc_rows(n, d, cuda) for _ in range(H)]
    for h, f in enumerate(feats):
        f.copy_(torch.from_numpy(hash_matrix(n, d, seed=h)))
    vec = torch.zeros(dev.round_up(d, 4), device=cuda)
    vec[:d] = 0.1
    out = dev.alloc_rows(n, d, cuda)
    ptrs, lds = _lib.hop_arrays(feats)
    _lib.check(_lib.lib().sgl_hop_gate_f32(H, ptrs, lds, _lib.ptr(vec), float("nan"), _lib.ptr(out), out.stride(0), None, 0, None, 0,
                                           n, d, _lib.current_stream_ptr()), "sgl_hop_gate_f32")
    assert bool(torch.isnan(out).all())
    _lib.check(_lib.lib().sgl_hop_gate_f32(H, ptrs, lds, _lib.ptr(vec), 0.25, _lib.ptr(out), out.stride(0), None, 0, None, 0,
                                           n, d, _lib.current_stream_ptr()), "sgl_hop_gate_f32")
    sc = torch.stack([f @ vec[:d] + 0.25 for f in feats], 1)
    want = sum(torch.softmax(torch.sigmoid(sc), 1)[:, h:h + 1] * feats[h] for h in range(H))
    assert oracle.parity_ok(out.cpu().numpy(), want.cpu().numpy(), TOL)


def test_sharded_graph_op_world1_and_nafs_on_shards(goldens, cuda):
    """BASELINE config 4 flow on one rank: ShardedGraphOp == LaplacianGraphOp, NAFS on the local rows == NAFS on all"""
    from sgl_amd.dist import ShardedGraphOp
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    from sgl_amd.operators.message_op import OverSmoothDistanceWeightedOp
    g = goldens.graph("pl2000")
    x = hash_matrix(2000, 100, seed=12)
    for op, ref_op in ((ShardedGraphOp(4, r=0.5, strict_order=True), LaplacianGraphOp(4, r=0.5, strict_order=True)),
                       (ShardedGraphOp(3, r=0.5, alpha=0.2, strict_order=True, pieces=3), PprGraphOp(3, r=0.5, alpha=0.2, strict_order=True))):
        hops = op.propagate(g, x)
        ref = ref_op.propagate(g, x)
        assert (op.lo, op.hi) == (0, 2000) and len(hops) == len(ref)
        for a, b in zip(hops, ref):
            assert torch.equal(a, b)
        nafs = OverSmoothDistanceWeightedOp()
        assert torch.equal(nafs.aggregate([h.contiguous() for h in hops]), nafs.aggregate(ref))
        assert op.gather_rows(hops[-1]) is hops[-1]


def test_bench_single_gpu_line_validates_itself(cuda, monkeypatch):
    """bench.py's N = 1 line carries config.validated: after the timed region, sampled rows of the last hop are recomputed in fp64
    and the last launch is repeated in strict order (benchlib/engine.py: validate_single).  A matrix whose values are changed
    behind the kernel's back -- the check recomputes from the corrupted array, the hop matrix was produced from the original --
    makes it false."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    import bench
    tiny = {"T_small": dict(n=30_000, m=300_000, d_max=900, d=100, k=3)}
    args = bench.parse_args(["--steps", "2", "--warmup", "1", "--workload", "T_small", "--no-cpu-baseline"])
    lines = []
    bench.run(args, workloads=tiny, emit=lines.append)
    j = json.loads(lines[0])
    v = j["config"]["validation"]
    assert j["config"]["validated"] is True and v["ok"] is True and v["sampled_rows"] == 4096 and v["sampled_rows_fp64_ok"] is True
    assert 0 <= v["strict_vs_fast_max_row_rel_l2"] <= TOL and 0 <= v["strict_vs_fast_max_abs_rel"] <= TOL

    class Corrupting(bench.GpuEngine):
        def validate_single(self, samples=4096, tol=1e-5):
            self._single["val"].mul_(1.0 + 1e-3)              # every entry of A_hat off by 0.1 % AFTER the hops were computed
            return super().validate_single(samples, tol)

    lines = []
    bench.run(args, engine_cls=Corrupting, workloads=tiny, emit=lines.append)
    j = json.loads(lines[0])
    assert j["config"]["validated"] is False and j["config"]["validation"]["sampled_rows_fp64_ok"] is False and j["value"] > 0


def test_library_writes_bump_tensor_versions(goldens, cuda):
    """the library writes into caller tensors through raw pointers; every wrapper then bumps the tensor's torch version counter
    (device._wrote), so that (data_ptr, _version)-keyed memos -- hopcache.SharedHops' content key, degree_powers' cache -- and
    autograd's saved-tensor checks see the write: a feature buffer refilled by a kernel between two propagate() calls maps to a NEW
    key under share_hops, and the hop store keys on the device the chain lives on"""
    from sgl_amd import config
    from sgl_amd.hopcache import SHARED
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    g = goldens.graph("pl2000")
    n, d = 2000, 16
    ptr_, col_, val_ = oracle.laplacian_adj(g.indptr, g.indices, g.data, n, 0.5)
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(cuda)  # noqa: E731
    csr = dev.DeviceCSR(to(ptr_, np.int64), to(col_, np.int32), to(val_, np.float32), (n, n))
    x = torch.from_numpy(hash_matrix(n, d, seed=1)).to(cuda)
    out, acc = torch.zeros_like(x), torch.zeros_like(x)

    def bumped(t, fn):
        v = t._version
        fn()
        return t._version > v
    assert bumped(out, lambda: csr.spmm(x, out=out))
    assert bumped(acc, lambda: csr.spmm_acc(x, out, acc))
    outs = [torch.zeros_like(x) for _ in range(2)]
    assert bumped(outs[1], lambda: csr.spmm_chain(x, 2, outs=outs))
    assert bumped(out, lambda: csr.spmm_axpb_clamp(x, 0.5, out=out))
    idx = torch.arange(0, n, 7, device=cuda)
    sub = torch.zeros((idx.numel(), d), device=cuda)
    assert bumped(sub, lambda: dev.gather_rows(x, idx, out=sub))
    big = torch.zeros_like(x)
    assert bumped(big, lambda: dev.scatter_rows(x, idx, idx, big))
    mixed = [torch.zeros_like(x)]
    assert bumped(mixed[0], lambda: dev.hop_lincomb([x, out], np.array([[0.25, 0.75]]), outs=mixed))
    graph = csr.capture_chain(x, outs)
    assert bumped(outs[0], graph.replay)
    # the hop store: a kernel-refilled feature buffer is another key; the device is part of the key
    old = config.share_hops
    config.share_hops = True
    try:
        SHARED.entries.clear()
        feat = x.clone()
        op = LaplacianGraphOp(2)
        h1 = op.propagate(g, feat)
        k1 = SHARED.data_key(g, feat, cuda)
        dev.gather_rows(x, torch.arange(n - 1, -1, -1, device=cuda), out=feat)        # refilled in place by a library kernel
        assert SHARED.data_key(g, feat, cuda) != k1
        h2 = LaplacianGraphOp(2).propagate(g, feat)
        ref = oracle.propagate((ptr_, col_, val_), feat.cpu().numpy(), 2)
        assert oracle.parity_ok(h2[2].cpu().numpy(), ref[2], TOL) and not torch.equal(h1[2], h2[2])
        assert SHARED.data_key(g, feat, "cuda:1") != SHARED.data_key(g, feat, "cuda:0")
    finally:
        config.share_hops = old
        SHARED.entries.clear()


def test_model_written_against_the_reference_names_runs_on_the_alias(goldens, cuda):
    """sgl_amd.compat.install(): a model whose source uses only the REFERENCE's module names (the body of sgl/models/homo/gamlp.py:1-13
    and sgc.py:1-13, typed here -- the reference's files themselves do not exist on this box; tests/test_host_cpu.py loads them
    where they do) preprocesses and predicts on the GPU and reproduces the logits recorded from the reference (G4)."""
    from sgl_amd import compat
    try:
        compat.install()
        from sgl.models.base_model import BaseSGAPModel
        from sgl.models.simple_models import LogisticRegression, MultiLayerPerceptron
        from sgl.operators.graph_op import LaplacianGraphOp
        from sgl.operators.message_op import LastMessageOp, LearnableWeightedMessageOp

        class SGC(BaseSGAPModel):
            def __init__(self, prop_steps, feat_dim, output_dim):
                super(SGC, self).__init__(prop_steps, feat_dim, output_dim)
                self._pre_graph_op = LaplacianGraphOp(prop_steps, r=0.5)
                self._pre_msg_op = LastMessageOp()
                self._base_model = LogisticRegression(feat_dim, output_dim)

        class GAMLP(BaseSGAPModel):
            def __init__(self, prop_steps, feat_dim, output_dim, hidden_dim, num_layers):
                super(GAMLP, self).__init__(prop_steps, feat_dim, output_dim)
                self._pre_graph_op = LaplacianGraphOp(prop_steps, r=0.5)
                self._pre_msg_op = LearnableWeightedMessageOp(0, prop_steps + 1, "jk", prop_steps, feat_dim)
                self._base_model = MultiLayerPerceptron(feat_dim, hidden_dim, num_layers, output_dim)
        g4 = goldens.npz("g4_models")
        g = goldens.graph("pl2000")
        x = hash_matrix(2000, 16, seed=4242)
        for name, model in (("SGC", SGC(3, 16, 5)), ("GAMLP", GAMLP(3, 16, 5, 32, 2))):
            model.load_state_dict({k.split("|param|")[1]: torch.from_numpy(v) for k, v in g4.items() if k.startswith(name + "|param|")})
            model = model.to(cuda).eval()
            model.preprocess(g, x)
            with torch.no_grad():
                y = model.model_forward(g4["idx"], cuda)
            rep = oracle.parity_report(y.cpu().numpy(), g4[f"{name}|out"], TOL, rowwise=False)
            assert rep["ok"], (name, rep)
    finally:
        compat.uninstall()


def test_bench_secondary_sections_cover_every_baseline_config(cuda, monkeypatch, tmp_path):
    """bench.py's single-GPU line carries one section per BASELINE config the headline does not cover (benchlib/extras.py), here on
    the small graphs of --extras-scale small: config 1 compared with the CPU oracle (strict order bit for bit), config 3's three
    preprocess calls, config 5's operator sweep with every MessageOp checked on sampled rows against the float64 formula, the
    per-rank share of configs 4/5, and the community graph with reorder=None / "auto" (bit-identical).  A check that is handed the
    wrong formula parameters fails its op and its section -- the validation has teeth."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    import bench
    from benchlib import extras
    detail = str(tmp_path / "detail.json")
    args = bench.parse_args(["--steps", "2", "--warmup", "1", "--workload", "S1_small", "--extras-scale", "small", "--no-cpu-baseline",
                             "--detail-out", detail])
    lines = []
    bench.run(args, emit=lines.append)
    j = json.loads(lines[0])
    assert list(j)[-1] == "sections" and j["config"]["validated"] is True and "papers100M" not in j
    sec = j["sections"]
    assert set(sec) == {"S0_pubmed", "S2_gamlp", "S4_products", "S1_community", "S4_papers_shard"}
    for name, s_ in sec.items():
        assert "failed" not in s_ and "skipped" not in s_, (name, s_)
        assert all(k in s_ for k in extras.REQUIRED_KEYS), (name, list(s_))
        assert s_["validated"] is True and s_["ms"] > 0 and 0 < s_["roofline"]["frac"] < 2, (name, s_)
    assert sec["S0_pubmed"]["strict_order_bit_equal_to_cpu_oracle"] is True and sec["S0_pubmed"]["fast_order_row_l2_rel"] <= TOL
    assert sec["S0_pubmed"]["cpu_baseline"]["propagate_ms"] > 0
    assert len(sec["S2_gamlp"]["preprocess_calls_ms"]) == 3 and sec["S2_gamlp"]["train_feed_ms"] > 0
    cpc = sec["S2_gamlp"]["columns_propagated_per_call"]              # calls 2 and 3 re-propagate the label columns only
    assert cpc[0] > cpc[1] == cpc[2] == 48 and sec["S2_gamlp"]["full_recompute_call_ms"] > 0, cpc
    for name in ("S4_products", "S4_papers_shard"):
        assert len(sec[name]["message_ops"]) == 10 and all(r[3] is True for r in sec[name]["message_ops"]), sec[name]["message_ops"]
        assert all(g[4] is True for g in sec[name]["graph_ops"])
    assert len(sec["S4_products"]["graph_ops"]) == 4 and all(g[5] is not None for g in sec["S4_products"]["graph_ops"][1:])
    assert set(sec["S4_products"]["cpu_baseline_combine"]["ms"]) == {"mean", "max", "concat", "nafs"}
    com = sec["S1_community"]["reorder_auto"]
    assert com["applied"] is True and com["bit_identical"] is True and com["edge_locality"][1] > com["edge_locality"][0] + 0.15
    full = json.load(open(detail))["sections"]
    assert full["S4_products"]["graph_ops"][0]["message_ops"][2]["preprocess_objective_ms"] > 0
    assert "sample" in full["S0_pubmed"]["cpu_baseline"] and full["S2_gamlp"]["validation"]["ok"] is True

    # the aggregate check fails when it is handed another formula than the one the op computes
    real = extras._search_space_ops

    def wrong(K, d, device, with_nafs=True):
        ops = real(K, d, device, with_nafs)
        return [(n_, k_, op, dict(p, alpha=0.8) if k_ == "simple_weighted" else p) for n_, k_, op, p in ops]
    monkeypatch.setattr(extras, "_search_space_ops", wrong)
    args2 = bench.parse_args(["--steps", "1", "--warmup", "1", "--workload", "S1_small", "--extras-scale", "small", "--no-cpu-baseline"])
    eng = bench.GpuEngine(0)
    d2 = {}
    extras.run_extras(args2, eng, d2, budget_s=120, which=("S4_products",))
    s4 = d2["sections"]["S4_products"]
    assert s4["validated"] is False
    bad = [t["msg_op"] for t in s4["graph_ops"][0]["message_ops"] if not t["validated"]]
    assert bad == ["simple_weighted a=.85"], bad


def test_c_abi_from_plain_c_program(cuda, tmp_path):
    """examples/c_abi_propagate.c: a C99 program with raw hipMalloc'ed buffers (no Python, no PyTorch) normalises a
    graph and runs the k-hop chain through include/sgl_hip.h, and checks every hop bit for bit against the
    reference's loop order on the host"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(ROOT, "sgl_amd", "csrc")
    exe = str(tmp_path / "c_abi_propagate")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", os.path.join(ROOT, "examples", "c_abi_propagate.c"), "-o", exe,
                           "-L", libdir, "-lsgl_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("C-ABI OK"), (out.returncode, out.stdout, out.stderr)


def test_c_abi_exchange_moves_bytes_through_a_real_rccl_communicator(cuda):
    """The C ABI's RCCL binding against the REAL librccl: a one-rank communicator created by the caller (RCCL refuses two ranks on
    one device) carries grouped ncclSend / ncclRecv pairs whose peer is the rank itself -- sgl_exchange_selftest posts them through
    the same post_group(), dlsym'ed function table and ncclFloat32 constant as sgl_allgather_rows / sgl_exchange_rows
    (sgl_exchange.hip), so the bytes that arrive prove the argument order, the data type and the stream / communicator hand-over.
    (The multi-peer bookkeeping itself is checked with 2-8 ranks against the mock RCCL, tests/native/exchange_mock.cpp.)"""
    import ctypes

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    try:
        rccl = ctypes.CDLL("librccl.so.1", mode=ctypes.RTLD_GLOBAL)
    except OSError:
        pytest.skip("no librccl.so.1 on this box")
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    lib = _lib.lib()
    try:
        assert lib.sgl_exchange_backend() in (b"process", b"librccl.so") and lib.sgl_exchange_backend() != b""
        g = torch.Generator(device=cuda).manual_seed(5)
        # a feature block's worth of rows (whole padded rows travel): 4096 x 128 floats, as 1, 3 and 7 pairs in one group
        src = torch.randn((4096, 128), generator=g, device=cuda)
        for n_ops in (1, 3, 7):
            dst = torch.full_like(src, float("nan"))
            _lib.check(lib.sgl_exchange_selftest(comm, 0, _lib.ptr(src), _lib.ptr(dst), src.numel(), n_ops, _lib.current_stream_ptr()),
                       "sgl_exchange_selftest")
            torch.cuda.synchronize()
            assert torch.equal(dst, src), n_ops                              # every byte arrived, bit for bit
        # on a side stream, ordered behind the kernel that produces the source (stream-ordered like the exchanges)
        side = torch.cuda.Stream(device=cuda)
        dst = torch.zeros(1000, device=cuda)
        with torch.cuda.stream(side):
            part = torch.arange(1000, dtype=torch.float32, device=cuda) * 0.5
            _lib.check(lib.sgl_exchange_selftest(comm, 0, _lib.ptr(part), _lib.ptr(dst), 1000, 2, ctypes.c_void_p(side.cuda_stream)),
                       "sgl_exchange_selftest")
        side.synchronize()
        assert torch.equal(dst, torch.arange(1000, dtype=torch.float32, device=cuda) * 0.5)
        # an odd count (not a multiple of anything) and the refusals
        a, b = torch.randn(1237, generator=g, device=cuda), torch.zeros(1237, device=cuda)
        _lib.check(lib.sgl_exchange_selftest(comm, 0, _lib.ptr(a), _lib.ptr(b), 1237, 5, _lib.current_stream_ptr()), "sgl_exchange_selftest")
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        assert lib.sgl_exchange_selftest(comm, 0, _lib.ptr(a), _lib.ptr(a), 16, 1, _lib.current_stream_ptr()) != 0
        assert b"overlap" in lib.sgl_last_error()
        assert lib.sgl_exchange_selftest(None, 0, _lib.ptr(a), _lib.ptr(b), 16, 1, _lib.current_stream_ptr()) != 0
        # the exchanges themselves on this communicator: a one-rank world has nothing to move and leaves the replica alone
        x = torch.arange(40 * 16, dtype=torch.float32, device=cuda).view(40, 16)
        keep = x.clone()
        bounds = (ctypes.c_int64 * 2)(0, 40)
        _lib.check(lib.sgl_allgather_rows(comm, 0, 1, bounds, _lib.ptr(x), 16, _lib.current_stream_ptr()), "sgl_allgather_rows")
        torch.cuda.synchronize()
        assert torch.equal(x, keep)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_c_abi_row_sharded_program(cuda, tmp_path):
    """examples/c_abi_row_sharded.c: the contract multi-GPU layout from plain C -- per-GPU row blocks normalised with
    sgl_norm_block_*, rectangular plans, sgl_allgather_rows on RCCL communicators created by the program itself; every hop
    on every GPU bit-equal to the reference's loop order.  Runs with however many GPUs the box shows (one here)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/lib/librccl.so"):
        pytest.skip("needs gcc and RCCL")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "sgl_amd", "csrc")
    exe = str(tmp_path / "c_abi_row_sharded")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"),
                           "-I", "/opt/rocm/include", os.path.join(root, "examples", "c_abi_row_sharded.c"), "-o", exe,
                           "-L", libdir, "-lsgl_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lrccl", "-lm",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C-ABI row-sharded OK" in out.stdout, (out.returncode, out.stdout[-500:], out.stderr[-1500:])
    assert "exchange backend: process" in out.stdout            # the library used the program's own RCCL, not a second copy


def test_c_abi_sweeps_program(cuda, tmp_path):
    """examples/c_abi_sweeps.c: the round-5 sweep entry points from plain C99 -- one preparation serving two r x (Laplacian + three
    alpha) normalisations (sgl_norm_block_prepare / build once, sgl_norm_degree_powers, sgl_norm_block_scale, sgl_norm_block_mix) and
    every NAFS prefix of a 6-hop propagation from one pass (sgl_nafs_prefix_f32), each checked on the host inside the program"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "sgl_amd", "csrc")
    exe = str(tmp_path / "c_abi_sweeps")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"),
                           "-I", "/opt/rocm/include", os.path.join(root, "examples", "c_abi_sweeps.c"), "-o", exe,
                           "-L", libdir, "-lsgl_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C-ABI sweeps OK" in out.stdout, (out.returncode, out.stdout[-500:], out.stderr[-1500:])


def test_int64_offsets_beyond_2_31_elements(cuda):
    """papers100M-shard shape: the dense operand has more than 2^31 elements (the reference's `int` offsets overflow
    there, matmul.c:29,33) and the gathered rows sit at byte offsets beyond 8 GiB"""
    from sgl_amd.device import DeviceCSR
    free, _ = torch.cuda.mem_get_info()
    n_cols, d = (40_000_000, 64) if free > 40e9 else (9_000_000, 256)
    assert n_cols * d > 2 ** 31
    rows, deg = 4096, 24
    g = torch.Generator(device=cuda).manual_seed(3)
    x = torch.empty((n_cols, d), device=cuda)
    x.normal_(generator=g)
    # columns concentrated at the far end of the table, plus a few at the start
    hi = torch.randint(n_cols - 1000, n_cols, (rows, deg - 2), generator=g, device=cuda)
    lo = torch.randint(0, 1000, (rows, 2), generator=g, device=cuda)
    cols = torch.cat([lo, hi], 1).sort(dim=1).values.to(torch.int32).reshape(-1).contiguous()
    vals = torch.rand(rows * deg, generator=g, device=cuda) + 0.5
    rowptr = torch.arange(0, rows + 1, device=cuda, dtype=torch.int64) * deg
    for strict in (True, False):
        y = DeviceCSR(rowptr, cols, vals, (rows, n_cols), strict=strict).spmm(x)
        ref = (x[cols.long()].double() * vals.double()[:, None]).view(rows, deg, d).sum(1)
        err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-6, err
    del x


# ---- full-size, size-independent properties ------------------------------------------------------------------
def test_products_scale_properties(cuda):
    """ogbn-products-shaped graph (N = 2.45 M, nnz ~ 126 M, d = 100): sampled rows against the oracle, linearity,
    row-stochasticity of D^-1 (A+I), strict == fast within tolerance."""
    from sgl_amd import device as dev
    from sgl_amd import synthetic
    free, _ = torch.cuda.mem_get_info()
    wl = synthetic.WORKLOADS["S1_products" if free > 60e9 else "S1_small"]
    n, d = wl["n"], wl["d"]
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=cuda)
    # symmetric, no self loops, sorted
    assert int(a_ptr[-1]) == a_col.numel() == 2 * wl["m"]
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    assert col.numel() == a_col.numel() + n
    x = synthetic.features_torch(n, d, seed=0, device=cuda)
    fast = dev.DeviceCSR(rowptr, col, val, (n, n))
    strict = dev.DeviceCSR(rowptr, col, val, (n, n), strict=True)
    y = fast.spmm(x)
    ys = strict.spmm(x)
    # (1) sampled rows vs the CPU oracle, bit-exact in strict mode
    rows = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:20000].sort().values
    rp, cc, vv = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    xh = x.cpu().numpy()
    sub_ptr = np.zeros(len(rows) + 1, np.int64)
    segs = [(int(rp[r]), int(rp[r + 1])) for r in rows.tolist()]
    sub_ptr[1:] = np.cumsum([e - b for b, e in segs])
    sub_col = np.concatenate([cc[b:e] for b, e in segs])
    sub_val = np.concatenate([vv[b:e] for b, e in segs])
    ref = oracle.oracle_spmm(sub_ptr, sub_col, sub_val, xh)
    assert np.array_equal(ys[rows.to(cuda)].cpu().numpy(), ref)
    rep = oracle.parity_report(y[rows.to(cuda)].cpu().numpy(), ref, TOL)
    assert rep["ok"], rep
    # (2) strict vs fast over ALL rows (row-wise L2 criterion on device)
    dn = (y - ys).norm(dim=1)
    rn = ys.norm(dim=1).clamp_min(1e-30)
    assert float((dn / rn).max()) <= TOL and float((y - ys).abs().max() / ys.abs().max()) <= TOL
    # (3) linearity: A(2x + 3z) == 2Ax + 3Az
    z = synthetic.features_torch(n, d, seed=5, device=cuda)
    lhs = fast.spmm(2 * x + 3 * z)
    rhs = 2 * y + 3 * fast.spmm(z)
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) <= 5e-6
    # (4) r = 1 gives the random-walk matrix (A+I)^T D^-1... columns scaled; use r = 0: rows of D^-1 (A+I) sum to 1
    rp0, c0, v0 = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.0, None)
    ones = torch.ones((n, 4), device=cuda)
    rs = dev.DeviceCSR(rp0, c0, v0, (n, n)).spmm(ones)
    assert float((rs - 1).abs().max()) <= 1e-5
    # (5) k-step propagation through the operator API reproduces repeated SpMM
    hop2 = strict.spmm(ys)
    assert torch.equal(hop2, strict.spmm(strict.spmm(x)))
    # (6) BASELINE config 3's width at the same size: d + C = 147 columns on the 160-float pitch (five whole lines per gathered row):
    #     the sampled rows bit-exact in strict order, the default order within tolerance over ALL rows, pad columns stay zero
    del z, lhs, rhs, rs, ones, hop2
    x147 = dev.upload_rows(synthetic.features_torch(n, 147, seed=3, device=cuda), cuda)
    xp = dev.padded_parent(x147)                           # what GraphOp.propagate multiplies: the whole pitch, pad columns zero
    assert xp.shape == (n, 160)
    y147s, y147 = strict.spmm(xp), fast.spmm(xp)
    ref147 = oracle.oracle_spmm(sub_ptr, sub_col, sub_val, x147.cpu().numpy())
    assert np.array_equal(y147s[rows.to(cuda)][:, :147].cpu().numpy(), ref147)
    dn = (y147 - y147s).norm(dim=1)
    assert float((dn / y147s.norm(dim=1).clamp_min(1e-30)).max()) <= TOL
    assert not bool(y147s[:, 147:].any()) and not bool(y147[:, 147:].any())


def test_hashed_generator_device_equals_host_mirror(cuda):
    """sgl_synth_* (rows generated per shard on device, keyed by (seed, row)) against the numpy mirror, bit for bit: small
    blocks anywhere in a papers100M-sized id range, and a block large enough (> 2^30 threads) that the kernels must
    stride -- a launch cannot carry 2^32 threads, and a truncated one leaves rows unwritten without any error."""
    from sgl_amd import synthetic as sy
    n = 111_059_956
    table = sy.degree_table(30.07, 20_000)
    for seed, row0, cnt in ((0, 0, 3000), (7, n - 2500, 2500), (1, 55_000_123, 1500)):
        rp, c, v = sy.hashed_block_torch(seed, row0, cnt, n, table, device=cuda)
        hp, hc, hv = sy.hashed_rows_numpy(seed, np.arange(row0, row0 + cnt), n, table)
        assert np.array_equal(rp.cpu().numpy(), hp) and np.array_equal(c.cpu().numpy(), hc) and np.array_equal(v.cpu().numpy(), hv)
        x = sy.hashed_features_torch(seed, row0, cnt, 100, device=cuda)
        assert np.array_equal(x.cpu().numpy(), sy.hashed_features_numpy(seed, np.arange(row0, row0 + cnt), 100))
        assert x.stride(0) == 100 or not dev.padded_parent(x)[:, 100:].any()
    big = 20_000_000                                          # 1.28e9 generator threads > the 2^30 the launch carries
    rp, c, v = sy.hashed_block_torch(3, 1000, big, n, table, device=cuda)
    rows = np.concatenate([np.arange(0, 64), np.arange(big // 2, big // 2 + 64), np.arange(big - 64, big)])
    hp, hc, hv = sy.hashed_rows_numpy(3, 1000 + rows, n, table)
    rph = rp.cpu().numpy()
    assert np.array_equal(np.diff(hp), (rph[rows + 1] - rph[rows]))
    sel = np.concatenate([np.arange(rph[r], rph[r + 1]) for r in rows])
    idx = torch.from_numpy(sel).to(cuda)
    assert np.array_equal(c[idx].cpu().numpy(), hc) and np.array_equal(v[idx].cpu().numpy(), hv)
    assert int(c.min()) >= 0 and int(c.max()) < n
    del rp, c, v
    xb = sy.hashed_features_torch(3, 0, 12_000_000, 128, device=cuda)          # 1.5e9 elements
    tail = np.arange(12_000_000 - 100, 12_000_000)
    assert np.array_equal(xb[tail[0]:].cpu().numpy(), sy.hashed_features_numpy(3, tail, 128))


def test_papers_shard_million_sampled_rows_vs_oracle(cuda):
    """BASELINE configs 4/5 at their own shape: one rank's 1/8 row block of the ogbn-papers100M-shaped hashed graph
    (13.9 M rows, ~418 M non-zeros) against the full 111 M x 128 feature replica (56.9 GB).  2^20 sampled rows are
    recomputed by the CPU oracle (the gathered X rows travel compacted) and must be bit-equal wherever the row is one
    fmaf chain (<= 2048 non-zeros), within 1e-5 for the few split rows; the sampled rows' generator output is pinned to
    the host mirror."""
    from sgl_amd import synthetic as sy
    free, _ = torch.cuda.mem_get_info()
    n = 111_059_956 if free > 120e9 else 111_059_956 // 8
    d = 128
    table = sy.degree_table(30.07, 20_000)
    rows_blk = n // 8
    rp, c, v = sy.hashed_block_torch(0, 0, rows_blk, n, table, device=cuda)
    x = sy.hashed_features_torch(0, 0, n, d, device=cuda)
    csr = dev.DeviceCSR(rp, c, v, (rows_blk, n))
    y = csr.spmm(x)
    info = csr.info()
    assert info["nnz"] == c.numel() and info["n_pieces"] > 0          # the block does contain rows that get split
    m = 1 << 20
    rows = torch.randperm(rows_blk, generator=torch.Generator().manual_seed(1))[:m].sort().values.to(cuda)
    b, e = rp[rows], rp[rows + 1]
    cnt = e - b
    srp = torch.zeros(m + 1, dtype=torch.int64, device=cuda)
    torch.cumsum(cnt, 0, out=srp[1:])
    pos = torch.arange(int(srp[-1]), device=cuda) - torch.repeat_interleave(srp[:-1], cnt) + torch.repeat_interleave(b, cnt)
    sc, sv = c[pos], v[pos]
    # generator pinned to the host mirror on the first 3000 sampled rows
    k = 3000
    hp, hc, hv = sy.hashed_rows_numpy(0, rows[:k].cpu().numpy(), n, table)
    assert np.array_equal(hp, srp[:k + 1].cpu().numpy()) and np.array_equal(hc, sc[:hp[-1]].cpu().numpy())
    assert np.array_equal(hv, sv[:hp[-1]].cpu().numpy())
    uniq, inv = torch.unique(sc.long(), return_inverse=True)
    assert np.array_equal(x[uniq[:500]].cpu().numpy(), sy.hashed_features_numpy(0, uniq[:500].cpu().numpy(), d))
    xc = x[uniq].cpu().numpy()                                          # only the gathered rows leave the GPU
    ref = oracle.oracle_spmm(srp.cpu().numpy(), inv.to(torch.int32).cpu().numpy(), sv.cpu().numpy(), xc, n_rows=m)
    got = y[rows].cpu().numpy()
    whole = (cnt <= 2048).cpu().numpy()
    assert whole.sum() >= m - 2000 and np.array_equal(got[whole], ref[whole])   # one fmaf chain per (row, column): bit-exact
    if (~whole).any():
        scale = oracle.oracle_spmm(srp.cpu().numpy(), inv.to(torch.int32).cpu().numpy(), np.abs(sv.cpu().numpy()), np.abs(xc), n_rows=m)
        rep = oracle.parity_report(got[~whole], ref[~whole], TOL, scale=scale[~whole])
        assert rep["ok"], rep
    print(f"papers shard: {int(whole.sum())} sampled rows bit-equal, {int((~whole).sum())} split rows within {TOL}")


@pytest.mark.gpu
def test_shared_hop_store_serves_fresh_operators(goldens, cuda):
    """BASELINE config 5 (a fresh model per search trial, sgl/search/search_models.py:19-46): with sgl_amd.config.share_hops a fresh
    GraphOp returns the chain an earlier operator produced for the same CONTENT of (adjacency, features), a PprGraphOp is mixed
    from the LaplacianGraphOp chain of the same r (against the reference's recorded PPR hops, 1e-5), a strict_order request never
    takes a mixed chain, edited features are another key, and the byte budget evicts the least recently used chain"""
    from sgl_amd import config, hopcache
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    g9 = goldens.npz("g9_config5")
    meta = goldens.json("g9_config5")["prop"]
    store = hopcache.SHARED
    store.clear()
    before = dict(store.stats)
    old = (config.share_hops, config.share_hops_gb)
    config.share_hops = True
    try:
        done = set()
        n_checked = 0
        for key, m in meta.items():
            if m["kind"] != "ppr":
                continue
            g = goldens.graph(m["graph"])
            x = hash_matrix(g.shape[0], m["d"], seed=m["seed"])
            ck = (m["graph"], m["d"], m["seed"], m["r"], m["K"])
            if ck not in done:
                lap = LaplacianGraphOp(m["K"], r=m["r"]).propagate(g, x)
                again = LaplacianGraphOp(m["K"], r=m["r"]).propagate(g.copy(), x.copy())      # other objects, same content
                assert all(a_ is b_ for a_, b_ in zip(lap, again))
                done.add(ck)
            d0 = store.stats["derived"]
            hops = PprGraphOp(m["K"], r=m["r"], alpha=m["alpha"]).propagate(g, x)
            assert store.stats["derived"] == d0 + 1, key
            for h in m["keep"]:
                rep = oracle.parity_report(hops[h].cpu().numpy(), g9[f"prop|{key}|h{h}"], TOL)
                assert rep["ok"], (key, h, rep)
                n_checked += 1
            h0 = store.stats["hits"]
            fresh = PprGraphOp(m["K"], r=m["r"], alpha=m["alpha"])
            assert all(a_ is b_ for a_, b_ in zip(hops, fresh.propagate(g, x)))
            assert store.stats["hits"] == h0 + 1
            assert fresh._adj is None             # a hit builds neither the normalised adjacency nor an SpMM plan
        assert n_checked >= 15
        # the reference's exceptions still come first, hit or miss
        with pytest.raises(TypeError, match="scipy csr sparse matrix"):
            LaplacianGraphOp(2).propagate(g.tocoo(), x)
        with pytest.raises(ValueError, match="Dimension mismatch"):
            LaplacianGraphOp(2).propagate(g, x[:-1])
        # strict order: propagates itself (bit-identical to the strict chain without the store), and then answers relaxed requests
        g = goldens.graph("pl256")
        x = hash_matrix(256, 24, seed=5)
        LaplacianGraphOp(4, r=0.5).propagate(g, x)
        d0, m0 = store.stats["derived"], store.stats["misses"]
        strict = PprGraphOp(4, r=0.5, alpha=0.3, strict_order=True).propagate(g, x)
        assert store.stats["derived"] == d0 and store.stats["misses"] == m0 + 1
        config.share_hops = False
        plain = PprGraphOp(4, r=0.5, alpha=0.3, strict_order=True).propagate(g, x)
        config.share_hops = True
        assert all(torch.equal(a_, b_) for a_, b_ in zip(strict, plain))
        relaxed = PprGraphOp(4, r=0.5, alpha=0.3).propagate(g, x)
        assert all(a_ is b_ for a_, b_ in zip(relaxed, strict))
        # a shorter chain is a prefix of a longer one; a longer one is a miss
        assert len(LaplacianGraphOp(2, r=0.5).propagate(g, x)) == 3
        m0 = store.stats["misses"]
        assert len(LaplacianGraphOp(6, r=0.5).propagate(g, x)) == 7 and store.stats["misses"] == m0 + 1
        # edited content is another key -- features and matrix -- and a device tensor passed as features is not aliased by the store
        x2 = x.copy()
        x2[17, 3] += 1.0
        m0 = store.stats["misses"]
        LaplacianGraphOp(4, r=0.5).propagate(g, x2)
        g2 = g.copy().tolil()
        g2[3, 200] = 1.0
        g2[200, 3] = 1.0
        LaplacianGraphOp(4, r=0.5).propagate(g2.tocsr(), x)
        assert store.stats["misses"] == m0 + 2
        xt = torch.from_numpy(hash_matrix(256, 8, seed=9)).to(cuda)
        kept = LaplacianGraphOp(2, r=0.5).propagate(g, xt)
        want0 = kept[0].clone()
        xt.add_(1.0)
        assert torch.equal(kept[0], want0)
        # device inputs: their content key is remembered while buffers and version counters stand, recomputed after an in-place write
        from sgl_amd.io import DeviceAdjacency
        da = DeviceAdjacency(torch.from_numpy(g.indptr.astype(np.int64)).to(cuda), torch.from_numpy(g.indices.astype(np.int32)).to(cuda),
                             torch.from_numpy(g.data.astype(np.float32)).to(cuda), g.shape)
        m0, h0 = store.stats["misses"], store.stats["hits"]
        edited = LaplacianGraphOp(2, r=0.5).propagate(da, xt)                 # xt was edited: not the chain kept above
        assert store.stats["misses"] == m0 + 1 and not torch.equal(edited[1], kept[1])
        again = LaplacianGraphOp(2, r=0.5).propagate(da, xt)
        assert store.stats["hits"] == h0 + 1 and all(a_ is b_ for a_, b_ in zip(again, edited))
        config.share_hops = False
        plain = LaplacianGraphOp(2, r=0.5).propagate(da, xt)
        config.share_hops = True
        assert all(torch.equal(a_, b_) for a_, b_ in zip(plain, edited))
        da.val.mul_(2.0)                                                      # the matrix edited in place: another key
        m0 = store.stats["misses"]
        LaplacianGraphOp(2, r=0.5).propagate(da, xt)
        assert store.stats["misses"] == m0 + 1
        # the reference's exceptions come before the lookup
        with pytest.raises(TypeError):
            LaplacianGraphOp(2).propagate(g.toarray(), x)
        # a PPR request nobody can serve propagates the Laplacian chain of its r ONCE; every alpha is then a mixing pass
        store.clear()
        d0 = store.stats["derived"]
        got = {a_: PprGraphOp(3, r=0.4, alpha=a_).propagate(g, x) for a_ in (0.1, 0.2, 0.35)}
        assert store.stats["derived"] == d0 + 3 and sum(1 for k_ in store.entries if k_[1] == "LaplacianGraphOp") == 1
        config.share_hops = False
        for a_, hops in got.items():
            own = PprGraphOp(3, r=0.4, alpha=a_).propagate(g, x)
            assert all(oracle.parity_ok(p_.cpu().numpy(), q_.cpu().numpy(), TOL) for p_, q_ in zip(hops, own)), a_
        config.share_hops = True
        # the budget: one chain of 4 hops x 256 x 24 floats is 98 KB; a 150 KB budget keeps one
        store.clear()
        config.share_hops_gb = 150e3 / (1 << 30)
        e0 = store.stats["evicted"]
        LaplacianGraphOp(4, r=0.5).propagate(g, x)
        LaplacianGraphOp(4, r=0.3).propagate(g, x)
        assert store.stats["evicted"] == e0 + 1 and len(store.entries) == 1
    finally:
        config.share_hops, config.share_hops_gb = old
        store.clear()
    assert store.stats["hits"] > before["hits"]


@pytest.mark.gpu
def test_gather_hops_uploads_the_indices_once(cuda, monkeypatch):
    """the training feed `[feat[idx].to(device) for feat in feat_list]` (sgl/models/base_model.py:58-60): bit-identical to torch
    indexing for every index kind torch takes, widths with and without a padded pitch, 1 ... 11 hops, mixed pitches; the indices are
    validated and uploaded ONCE for all hop matrices, also through a learnable model's forward"""
    rng = np.random.default_rng(3)
    for n, d, H in ((5000, 100, 4), (3000, 147, 6), (2048, 128, 11), (700, 3, 2), (900, 500, 3), (64, 37, 1)):
        host = [rng.standard_normal((n, d)).astype(np.float32) for _ in range(H)]
        feats = [dev.upload_rows(h_, cuda) for h_ in host]
        picks = rng.integers(-n, n, size=1501)
        mask = rng.random(n) < 0.3
        for idx in (picks, picks.tolist(), torch.from_numpy(picks), torch.from_numpy(picks).to(cuda), mask, torch.from_numpy(mask), range(3, n, 7)):
            got = dev.gather_hops(feats, idx)
            ref_idx = list(idx) if isinstance(idx, range) else (idx.cpu() if torch.is_tensor(idx) else idx)
            for h in range(H):
                want = torch.from_numpy(host[h])[ref_idx]
                assert got[h].shape == want.shape and torch.equal(got[h].cpu(), want), (n, d, H, type(idx), h)
                assert not bool(dev.padded_parent(got[h])[:, d:].abs().sum())            # the outputs' own padding is zeros
        # hop matrices of 16-byte rows (what propagate() returns) go through ONE launch for all hops (sgl_gather_hops_padded_f32: the
        # indices are read once per row); bit-identical to the hop-by-hop form; packed odd widths and single matrices fall back
        dv_idx = torch.from_numpy(np.where(picks < 0, picks + n, picks)).to(cuda)
        one = dev._gather_hops_one_launch(feats, dv_idx) if H > 1 else None
        assert (one is not None) == (H > 1), (n, d, H)                  # every multi-hop shape of this list is eligible
        if one is not None:
            per_hop = dev.gather_hops(feats, dv_idx, one_launch=False)
            assert all(torch.equal(a_, b_) and torch.equal(dev.padded_parent(a_), dev.padded_parent(b_)) for a_, b_ in zip(one, per_hop))
            for grid_ in (1, 0):            # 1: the hop in blockIdx.y (default); 0: the hop loop inside the thread
                _lib.set_tuning("gather_hops_grid", grid_)
                for u_ in (0, 1, 2, 4):
                    _lib.set_tuning("gather_rows_per_thread", u_)
                    assert all(torch.equal(a_, b_) and torch.equal(dev.padded_parent(a_), dev.padded_parent(b_))
                               for a_, b_ in zip(dev._gather_hops_one_launch(feats, dv_idx), per_hop)), (n, d, H, grid_, u_)
            _lib.set_tuning("gather_rows_per_thread", 0)
            _lib.set_tuning("gather_hops_grid", 1)
    with pytest.raises(IndexError):
        dev.gather_hops(feats, [0, n])
    assert [tuple(t.shape) for t in dev.gather_hops(feats, [])] == [(0, 37)] and dev.gather_hops([], [1]) == []
    # a column view (pitch not the allocator's) and a packed odd-width matrix next to padded ones
    wide = dev.upload_rows(rng.standard_normal((800, 110)).astype(np.float32), cuda)
    packed = torch.from_numpy(rng.standard_normal((800, 50)).astype(np.float32)).to(cuda)
    mixed = [wide[:, 3:53], packed, dev.upload_rows(packed.cpu().numpy(), cuda)]
    ids = rng.integers(0, 800, size=333)
    for g_, f_ in zip(dev.gather_hops(mixed, ids), mixed):
        assert torch.equal(g_, f_[torch.from_numpy(ids).to(cuda)])
    # one upload: host indices become a device tensor once, whatever the number of hop matrices -- also through the model
    calls = []
    real = dev._device_index

    def counting(idx, n_rows, device):
        if not (torch.is_tensor(idx) and idx.is_cuda):
            calls.append(type(idx).__name__)
        return real(idx, n_rows, device)
    monkeypatch.setattr(dev, "_device_index", counting)
    dev.gather_hops(feats * 5, [1, 2, 3])
    assert len(calls) == 1
    from sgl_amd.models.homo import GAMLP
    g = long_row_graph(1500, seed=4)
    g = (abs(g) + abs(g).T).tocsr().astype(np.float32)
    x = hash_matrix(1500, 24, seed=1)
    model = GAMLP(3, 24, 5, 16, 2).to(cuda).eval()
    model.preprocess(g, x)
    batch = rng.integers(0, 1500, size=400)
    with torch.no_grad():
        want = model._pre_msg_op.aggregate([f_[torch.from_numpy(batch).to(cuda)] for f_ in model._processed_feat_list])
        del calls[:]
        out = model.model_forward(batch, cuda)
        assert len(calls) == 1 and len(model._processed_feat_list) == 4
        assert torch.equal(out, model._base_model(want))


@pytest.mark.gpu
def test_training_steps_record_into_a_hip_graph(cuda):
    """The per-step work of the learnable models -- row gather of every hop matrix, aggregator forward and backward -- captured once
    in a HIP graph (torch.cuda.CUDAGraph) and replayed: the library's launches are stream-ordered, allocate through torch's allocator
    only, upload nothing and never synchronise.  Every MessageOp kind with grad-carrying inputs: same outputs and gradients as the
    eager step, and a replay follows indices and inputs rewritten in place.  (tests/capture_step_check.py, in a process of its own: a
    capture that goes wrong ends the process, not the test.)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.join(os.path.dirname(os.path.abspath(__file__)), "capture_step_check.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CAPTURE-OK 15" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
