"""Property-based CPU tests (hypothesis) of the integer / index logic around the kernels: the SpMM execution plan,
the row-shard arithmetic and the column chunking.  Index work must be exact for EVERY input, so it gets generated
inputs rather than a handful of hand-picked ones."""
import numpy as np
from hypothesis import given, settings, strategies as st

from sgl_amd.dist import all_piece_bounds, balanced_bounds, column_chunks
from test_host_cpu import build_plan

degrees = st.lists(st.one_of(st.integers(0, 6), st.integers(0, 80), st.integers(0, 3000)), min_size=0, max_size=400)


@settings(max_examples=150, deadline=None)
@given(deg=degrees, item_nnz=st.integers(1, 600), long_nnz=st.one_of(st.just(-1), st.integers(1, 700)))
def test_plan_is_an_exact_partition(deg, item_nnz, long_nnz):
    deg = np.asarray(deg, dtype=np.int64)
    n = len(deg)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    items, pb, pl, pr, lr, lf, counts = build_plan(rowptr, item_nnz, long_nnz)
    covered = np.zeros(n, dtype=np.int64)
    nnz_seen = 0
    last_end = 0
    for b, e in items:
        assert 0 <= b < e <= n and e - b <= 63 and b >= last_end      # ordered, disjoint, window of <= 63 rows
        last_end = e
        covered[b:e] += 1
        nnz_seen += rowptr[e] - rowptr[b]
        if long_nnz > 0:
            assert deg[b:e].max() <= long_nnz
    for k, r in enumerate(lr):
        assert long_nnz > 0 and deg[r] > long_nnz
        covered[r] += 1
        ps = list(range(lf[k], lf[k + 1]))
        assert len(ps) == -(-deg[r] // long_nnz)
        assert pb[ps[0]] == rowptr[r] and pb[ps[-1]] + pl[ps[-1]] == rowptr[r + 1]
        for q in ps:
            assert pr[q] == r and 0 < pl[q] <= long_nnz
            nnz_seen += pl[q]
    assert (covered == 1).all() and nnz_seen == rowptr[-1]
    assert counts[0] == len(items) and counts[1] == len(pb) and counts[2] == len(lr)


@settings(max_examples=150, deadline=None)
@given(deg=degrees, world=st.integers(1, 9), pieces=st.integers(1, 5))
def test_row_shards_tile_the_matrix(deg, world, pieces):
    rowptr = np.concatenate([[0], np.cumsum(np.asarray(deg, dtype=np.int64))]).astype(np.int64)
    n = len(deg)
    b = balanced_bounds(rowptr, world)
    assert b[0] == 0 and b[-1] == n and len(b) == world + 1 and (np.diff(b) >= 0).all()
    pb = all_piece_bounds(rowptr, world, pieces)
    assert pb.shape == (world, pieces + 1) and pb[0, 0] == 0 and pb[-1, -1] == n
    assert (np.diff(pb, axis=1) >= 0).all() and (pb[1:, 0] == pb[:-1, -1]).all() and (pb[:, 0] == b[:-1]).all()
    if n:
        # balance: no shard exceeds the ideal share by more than one row's worth of (nnz + 1)
        cost = np.diff(rowptr[b]) + np.diff(b)
        assert cost.max() <= (rowptr[-1] + n) / world + (max(deg) + 1) + 1


@settings(max_examples=200, deadline=None)
@given(d=st.integers(1, 3000), k=st.integers(1, 6))
def test_column_chunks_cover_and_align(d, k):
    ch = column_chunks(d, k)
    assert ch[0][0] == 0 and ch[-1][1] == d and all(a < b for a, b in ch)
    assert all(ch[i][1] == ch[i + 1][0] for i in range(len(ch) - 1))
    assert all((b - a) % 32 == 0 for a, b in ch[:-1])          # whole 128-byte lines except the last chunk
    assert len(ch) <= max(k, 1) + 1
