"""numpy / C restatement of the reference's SGAP pre-propagation algorithms.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function cites the
reference file:line it follows (paths relative to /root/reference).  Pure numpy
(fp64 for the normalisation, fp32 for everything the reference does in fp32);
the SpMM goes through the C restatement in spmm_ref.c.
"""
import ctypes
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

__all__ = [
    "load_oracle_lib", "load_reference_lib", "oracle_spmm", "oracle_spmm_scalar", "reference_spmm",
    "canonical_csr", "sym_norm_csr", "laplacian_adj", "ppr_adj", "propagate",
    "agg_last", "agg_concat", "agg_mean", "agg_sum", "agg_max", "agg_min",
    "alpha_weights", "one_dim_weighted_add", "two_dim_weighted_add",
    "agg_simple_weighted", "learnable_weights", "agg_learnable_weighted",
    "agg_iterate_learnable", "nafs_weights", "agg_over_smooth_distance",
    "sigmoid32", "softmax32", "parity_ok", "parity_report", "truth_report", "TRUTH_FLOOR",
    "label_propagation", "cs_correct", "cs_smooth", "nafs_task_features", "nafs_task_sweep", "coo_to_csr",
]

# ----------------------------------------------------------------------------------------------
# native pieces
# ----------------------------------------------------------------------------------------------
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, ndim=1, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, ndim=1, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, ndim=1, flags="C_CONTIGUOUS")

_oracle_lib = None
_ref_lib = None


def load_oracle_lib():
    """liboracle_spmm.so = oracle/spmm_ref.c (restates csrc/matmul.c:23-40)."""
    global _oracle_lib
    if _oracle_lib is None:
        path = os.path.join(_HERE, "liboracle_spmm.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.sgl_oracle_spmm_f32_i64.argtypes = [_f32p, _f32p, _i32p, _i64p, _f32p, ctypes.c_int64, ctypes.c_int64]
        lib.sgl_oracle_spmm_f32_i64.restype = None
        lib.sgl_oracle_spmm_f32_scalar.argtypes = lib.sgl_oracle_spmm_f32_i64.argtypes
        lib.sgl_oracle_spmm_f32_scalar.restype = None
        lib.sgl_oracle_FloatCSRMulDenseOMP.argtypes = [_f32p, _f32p, _i32p, _i32p, _f32p, ctypes.c_int, ctypes.c_int]
        lib.sgl_oracle_FloatCSRMulDenseOMP.restype = None
        _oracle_lib = lib
    return _oracle_lib


def load_reference_lib():
    """oracle/_ref/libmatmul.so = the reference's own csrc/matmul.c compiled in place; None if absent."""
    global _ref_lib
    if _ref_lib is None:
        path = os.path.join(_HERE, "_ref", "libmatmul.so")
        if not os.path.exists(path):
            return None
        lib = ctypes.CDLL(path)
        # matmul.h:5
        lib.FloatCSRMulDenseOMP.argtypes = [_f32p, _f32p, _i32p, _i32p, _f32p, ctypes.c_int, ctypes.c_int]
        lib.FloatCSRMulDenseOMP.restype = None
        _ref_lib = lib
    return _ref_lib


def oracle_spmm(indptr, indices, data, x, n_rows=None, out=None):
    """Y = A @ X, fp32, fmaf chain per (row, k) in CSR order (matmul.c:23-40).

    `x` is [n_cols, d]; `data` is rounded to float32 first, exactly where the reference
    rounds it (operators/utils.py:32).  Returns a new [n_rows, d] float32 array (or
    accumulates into `out`, like the reference accumulates into `answer`)."""
    lib = load_oracle_lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    d = x.shape[1]
    n_rows = len(indptr) - 1 if n_rows is None else n_rows
    y = np.zeros((n_rows, d), dtype=np.float32) if out is None else out
    lib.sgl_oracle_spmm_f32_i64(
        y.reshape(-1), np.ascontiguousarray(data, dtype=np.float32),
        np.ascontiguousarray(indices, dtype=np.int32), np.ascontiguousarray(indptr, dtype=np.int64),
        x.reshape(-1), n_rows, d)
    return y


def oracle_spmm_scalar(indptr, indices, data, x):
    lib = load_oracle_lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_rows, d = len(indptr) - 1, x.shape[1]
    y = np.zeros((n_rows, d), dtype=np.float32)
    lib.sgl_oracle_spmm_f32_scalar(
        y.reshape(-1), np.ascontiguousarray(data, dtype=np.float32),
        np.ascontiguousarray(indices, dtype=np.int32), np.ascontiguousarray(indptr, dtype=np.int64),
        x.reshape(-1), n_rows, d)
    return y


def reference_spmm(indptr, indices, data, x, n_rows=None):
    """Same product through the reference's own compiled kernel (oracle/_ref). int32 everything."""
    lib = load_reference_lib()
    if lib is None:
        raise RuntimeError("oracle/_ref/libmatmul.so not built (needs /root/reference)")
    x = np.ascontiguousarray(x, dtype=np.float32)
    d = x.shape[1]
    n_rows = len(indptr) - 1 if n_rows is None else n_rows
    y = np.zeros(n_rows * d, dtype=np.float32)
    lib.FloatCSRMulDenseOMP(
        y, np.ascontiguousarray(data, dtype=np.float32), np.ascontiguousarray(indices, dtype=np.int32),
        np.ascontiguousarray(indptr, dtype=np.int32), x.reshape(-1), n_rows, d)
    return y.reshape(n_rows, d)


# ----------------------------------------------------------------------------------------------
# normalisation  (operators/utils.py:76-88, graph_op/laplacian_graph_op.py:12-19,
#                 graph_op/ppr_graph_op.py:13-21)
# ----------------------------------------------------------------------------------------------
def canonical_csr(indptr, indices, data, n_cols=None):
    """Sort column indices inside each row and sum duplicates (fp64), the state scipy's
    `adj + sp.eye(n)` leaves the matrix in (operators/utils.py:77)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    data = np.asarray(data, dtype=np.float64)
    n = len(indptr) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    order = np.lexsort((indices, rows))          # stable: by row, then column
    rows, cols, vals = rows[order], indices[order], data[order]
    if len(rows):
        new = np.empty(len(rows), dtype=bool)
        new[0] = True
        new[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
        grp = np.cumsum(new) - 1
        vals = np.bincount(grp, weights=vals, minlength=grp[-1] + 1)   # sequential fp64 adds
        rows, cols = rows[new], cols[new]
    out_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=out_ptr[1:])
    return out_ptr, cols, vals


def sym_norm_csr(indptr, indices, data, n, r, alpha=None):
    """A_hat = D^{r-1} (A+I)^T D^{-r}   (operators/utils.py:76-88), optionally followed by
    (1-alpha) A_hat + alpha I   (graph_op/ppr_graph_op.py:20).

    Returns canonical CSR (indptr int64, indices int32, data float64) -- the object
    `_construct_adj(...).tocsr()` yields (laplacian_graph_op.py:19).  Entry-wise:
        A_hat[j, i] = fl( fl(A'[i, j] * deg_j^(r-1)) * deg_i^(-r) ),  A' = A + I,  deg = rowsum(A')
    all in fp64; the fp32 rounding happens later, at the SpMM call (operators/utils.py:32)."""
    ptr, col, val = canonical_csr(indptr, indices, data)
    # A' = A + I  (utils.py:77): union structure, diagonal += 1, zero results dropped (scipy binop)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr))
    rows = np.concatenate([rows, np.arange(n, dtype=np.int64)])
    cols = np.concatenate([col, np.arange(n, dtype=np.int64)])
    vals = np.concatenate([val, np.ones(n, dtype=np.float64)])
    ptr2 = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=ptr2[1:])
    order = np.lexsort((cols, rows))
    ptr2, col2, val2 = canonical_csr(ptr2, cols[order], vals[order])
    keep = val2 != 0.0
    if not keep.all():
        rows2 = np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr2))[keep]
        col2, val2 = col2[keep], val2[keep]
        ptr2 = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows2, minlength=n), out=ptr2[1:])
    rows2 = np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr2))
    # degrees = rowsum(A')  (utils.py:78) -- sequential fp64 in storage order
    deg = np.bincount(rows2, weights=val2, minlength=n).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        left = np.power(deg, r - 1)          # utils.py:79
        left[np.isinf(left)] = 0.0           # utils.py:80
        right = np.power(deg, -r)            # utils.py:83
        right[np.isinf(right)] = 0.0         # utils.py:84
    # (A' diag(left))^T diag(right)  (utils.py:87): entry (i,j) of A' lands at (j,i)
    v = (val2 * left[col2]) * right[rows2]
    t_rows, t_cols = col2, rows2
    order = np.lexsort((t_cols, t_rows))     # .tocsr() of the transposed product: sorted rows/cols
    t_rows, t_cols, v = t_rows[order], t_cols[order], v[order]
    if alpha is not None:                    # ppr_graph_op.py:20
        v = (1 - alpha) * v
        diag = t_rows == t_cols
        v[diag] = v[diag] + alpha
        # (the diagonal is always structurally present because A' = A + I; scipy's binop
        #  would drop an exactly-zero sum -- replicate)
        keep = v != 0.0
        if not keep.all():
            t_rows, t_cols, v = t_rows[keep], t_cols[keep], v[keep]
    out_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(t_rows, minlength=n), out=out_ptr[1:])
    return out_ptr, t_cols.astype(np.int32), v


def laplacian_adj(indptr, indices, data, n, r=0.5):
    """LaplacianGraphOp._construct_adj (graph_op/laplacian_graph_op.py:12-19)."""
    return sym_norm_csr(indptr, indices, data, n, r, None)


def ppr_adj(indptr, indices, data, n, r=0.5, alpha=0.15):
    """PprGraphOp._construct_adj (graph_op/ppr_graph_op.py:13-21)."""
    return sym_norm_csr(indptr, indices, data, n, r, alpha)


def propagate(norm_csr, x, prop_steps):
    """GraphOp.propagate's loop (operators/base_op.py:29-36): [X, AX, ..., A^K X], fp32."""
    ptr, col, val = norm_csr
    val32 = np.asarray(val).astype(np.float32)      # operators/utils.py:32
    feats = [np.ascontiguousarray(x, dtype=np.float32)]
    for _ in range(prop_steps):
        feats.append(oracle_spmm(ptr, col, val32, feats[-1]))
    return feats


# ----------------------------------------------------------------------------------------------
# aggregators  (operators/message_op/*.py; semantics table: SURVEY.md Appendix A)
# ----------------------------------------------------------------------------------------------
def sigmoid32(x):
    x = np.asarray(x, dtype=np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)


def softmax32(x, axis):
    x = np.asarray(x, dtype=np.float32)
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m).astype(np.float32)
    return (e / e.sum(axis=axis, keepdims=True, dtype=np.float32)).astype(np.float32)


def agg_last(feats):
    """last_message_op.py:9-10"""
    return feats[-1]


def agg_concat(feats, s, e):
    """concat_message_op.py:11-12"""
    return np.hstack(feats[s:e])


def agg_sum(feats, s, e):
    """sum_message_op.py:9-10 -- Python sum() over the SLICE feat_list[s:e] (an end beyond the list is clamped like any
    Python slice): ((0 + X_s) + X_{s+1}) + ..."""
    hops = feats[s:e]
    acc = hops[0].astype(np.float32, copy=True)
    for x in hops[1:]:
        acc = acc + x
    return acc


def agg_mean(feats, s, e):
    """mean_message_op.py:9-10 -- sum over the slice, then ONE true division by (e - s), whatever the slice held"""
    return agg_sum(feats, s, e) / np.float32(e - s)


def agg_max(feats, s, e):
    """max_message_op.py:11-12 (NaN propagates, as torch.max does)"""
    return np.maximum.reduce(np.stack(feats[s:e], 0), axis=0)


def agg_min(feats, s, e):
    """min_message_op.py:11-12"""
    return np.minimum.reduce(np.stack(feats[s:e], 0), axis=0)


def alpha_weights(alpha, n_feats, s, e):
    """simple_weighted_message_op.py:41-47 -- Python float64 recurrence, then FloatTensor slice"""
    w = [alpha]
    for _ in range(n_feats - 1):
        w.append((1 - alpha) * w[-1])
    return np.asarray(w[s:e], dtype=np.float32)


def one_dim_weighted_add(feats, w):
    """operators/utils.py:91-102: (vstack(flat X_h) * w[:,None]).sum(0)"""
    w = np.asarray(w, dtype=np.float32)
    acc = np.zeros_like(feats[0], dtype=np.float32)
    for h, f in enumerate(feats):
        acc = acc + f.astype(np.float32) * w[h]
    return acc


def two_dim_weighted_add(feats, w):
    """operators/utils.py:105-116: bmm(stack(X_h, 2) [n,d,H], W[:,:,None]) -> out[n,k] = sum_h X_h[n,k] W[n,h]"""
    w = np.asarray(w, dtype=np.float32)
    acc = np.zeros_like(feats[0], dtype=np.float32)
    for h, f in enumerate(feats):
        acc = acc + f.astype(np.float32) * w[:, h:h + 1]
    return acc


def agg_simple_weighted(feats, s, e, kind, arg):
    """simple_weighted_message_op.py:40-56"""
    w = alpha_weights(arg, len(feats), s, e) if kind == "alpha" else np.asarray(arg, dtype=np.float32)
    return one_dim_weighted_add(feats[s:e], w)


def _linear(x, weight, bias):
    return (x.astype(np.float32) @ np.asarray(weight, dtype=np.float32).T + np.asarray(bias, dtype=np.float32)).astype(np.float32)


def learnable_weights(feats, s, e, kind, param=None, weight=None, bias=None):
    """learnable_weighted_messahe_op.py:59-90 -- the weight tensor only.

    simple / simple_allow_neg return a 1-D [H] vector; gate / ori_ref / jk return [n, H].
    NB ori_ref and jk reshape the hop-major score vector with .view(-1, H) (:78,:86) which
    scrambles (node, hop) pairs; that IS the reference behaviour and is reproduced."""
    H = e - s
    if kind == "simple":
        return softmax32(sigmoid32(np.asarray(param, dtype=np.float32)[s:e]), 0)
    if kind == "simple_allow_neg":
        return np.asarray(param, dtype=np.float32)[s:e]
    if kind == "gate":
        stacked = np.vstack(feats[s:e])                                  # [H*n, d]
        sc = _linear(stacked, weight, bias)                              # [H*n, 1]
        return softmax32(sigmoid32(sc.reshape(H, -1).T), 1)              # view(H,-1).T  (:71)
    if kind == "ori_ref":
        ref = np.tile(feats[0], (H, 1))                                  # repeat(H,1)   (:74)
        adopted = np.hstack((ref, np.vstack(feats[s:e])))
        sc = _linear(adopted, weight, bias)
        return softmax32(sigmoid32(sc.reshape(-1, H)), 1)                # view(-1,H)    (:78)
    if kind == "jk":
        ref = np.tile(np.hstack(feats), (H, 1))                          # hstack ALL hops (:81)
        adopted = np.hstack((ref, np.vstack(feats[s:e])))
        sc = _linear(adopted, weight, bias)
        return softmax32(sigmoid32(sc.reshape(-1, H)), 1)                # view(-1,H)    (:86)
    raise ValueError(kind)


def agg_learnable_weighted(feats, s, e, kind, param=None, weight=None, bias=None):
    """learnable_weighted_messahe_op.py:59-101"""
    w = learnable_weights(feats, s, e, kind, param, weight, bias)
    if kind in ("simple", "simple_allow_neg"):
        return one_dim_weighted_add(feats[s:e], w)
    return two_dim_weighted_add(feats[s:e], w)


def agg_iterate_learnable(feats, s, e, weight, bias):
    """iterate_learnable_weighted_message_op.py:28-51 ('recursive')"""
    acc = feats[s]
    wl = None
    for i in range(s, e):
        sc = sigmoid32(_linear(np.hstack((feats[i], acc)), weight, bias))     # :33-34
        wl = sc if i == s else np.hstack((wl, sc))                            # :35-38
        wl = softmax32(wl, 1)                                                 # :39 (re-soft-maxed each step)
        acc = feats[s] * wl[:, 0:1]                                           # :41-42
        for j in range(1, i + 1):                                             # :43 (absolute i: only right for s=0)
            acc = acc + feats[s + j] * wl[:, j:j + 1]
    return acc


def nafs_weights(feats):
    """over_smooth_distance_op.py:12-22: W = softmax_h( (<X0,Xh>/(|Xh|+1e-10)) / (|X0|+1e-10) )"""
    x0 = feats[0].astype(np.float32)
    n0 = np.sqrt((x0 * x0).sum(1, dtype=np.float32)).astype(np.float32) + np.float32(1e-10)
    cols = []
    for f in feats:
        f = f.astype(np.float32)
        nh = np.sqrt((f * f).sum(1, dtype=np.float32)).astype(np.float32) + np.float32(1e-10)
        t = (x0 * f).sum(1, dtype=np.float32) / nh
        cols.append((t / n0)[:, None])
    return softmax32(np.concatenate(cols, 1), 1)


def agg_over_smooth_distance(feats):
    """over_smooth_distance_op.py:11-33 (the per-node Python loop :27-31, vectorised: same sums,
    accumulated in hop order starting from float 0.)"""
    w = nafs_weights(feats)
    acc = np.zeros_like(feats[0], dtype=np.float32)
    for h, f in enumerate(feats):
        acc = acc + w[:, h:h + 1] * f.astype(np.float32)
    return acc


# ----------------------------------------------------------------------------------------------
# SURVEY 8(f) rank 2 consumers: label propagation / Correct&Smooth / NAFS task pipeline
# ----------------------------------------------------------------------------------------------
def _one_hot(lab, C=None):
    lab = np.asarray(lab).reshape(-1)
    C = int(lab.max()) + 1 if C is None else C
    out = np.zeros((len(lab), C), dtype=np.float32)
    out[np.arange(len(lab)), lab] = 1.0
    return out


def label_propagation(labels, norm_csr, num_layers, alpha, clamp=(0.0, 1.0), mask=None, post=None):
    """sgl/tricks/utils.py:41-58.  norm_csr = (indptr, indices, data) of the already-normalised adjacency
    (float32-rounded like sparse_mx_to_torch_sparse_tensor does, :31-38)."""
    ptr, col, val = norm_csr
    val = np.asarray(val).astype(np.float32)
    labels = np.asarray(labels)
    if labels.dtype.kind in "iu":
        labels = _one_hot(labels)
    labels = labels.astype(np.float32)
    out = labels.copy()
    if mask is not None:
        out = np.zeros_like(labels)
        out[mask] = labels[mask]
    a = np.float32(alpha)
    res = (np.float32(1 - alpha) * out).astype(np.float32)           # :53
    for _ in range(num_layers):
        out = (a * oracle_spmm(ptr, col, val, out) + res).astype(np.float32)   # :55
        if post is not None:
            out = post(out)
        elif clamp is not None:
            out = np.clip(out, np.float32(clamp[0]), np.float32(clamp[1]))
    return out


def cs_correct(y_soft, y_true, mask, norm_csr, num_layers, alpha, autoscale=True, scale=1.0):
    """CorrectAndSmooth.correct (sgl/tricks/correct_and_smooth.py:18-45); mask = index array"""
    y_soft = np.asarray(y_soft, dtype=np.float32)
    yt = _one_hot(y_true, y_soft.shape[1]) if np.asarray(y_true).dtype.kind in "iu" else np.asarray(y_true, np.float32)
    error = np.zeros_like(y_soft)
    error[mask] = yt[mask] - y_soft[mask]
    num_true = len(mask)
    if autoscale:
        sm = label_propagation(error, norm_csr, num_layers, alpha, clamp=(-1.0, 1.0))
        sigma = np.float32(np.abs(error[mask]).sum(dtype=np.float32) / np.float32(num_true))
        with np.errstate(divide="ignore"):
            sc = sigma / np.abs(sm).sum(1, keepdims=True, dtype=np.float32)
        sc[np.isinf(sc) | (sc > 1000)] = 1.0
        return (y_soft + sm * sc).astype(np.float32)

    def fix(x):
        x = x.copy()
        x[mask] = error[mask]
        return x
    sm = label_propagation(error, norm_csr, num_layers, alpha, post=fix)
    return (y_soft + sm * np.float32(scale)).astype(np.float32)


def cs_smooth(y_soft, y_true, mask, norm_csr, num_layers, alpha):
    """CorrectAndSmooth.smooth (sgl/tricks/correct_and_smooth.py:47-62)"""
    y_soft = np.asarray(y_soft, dtype=np.float32).copy()
    yt = _one_hot(y_true, y_soft.shape[1]) if np.asarray(y_true).dtype.kind in "iu" else np.asarray(y_true, np.float32)
    y_soft[mask] = yt[mask]
    return label_propagation(y_soft, norm_csr, num_layers, alpha)


def nafs_task_features(indptr, indices, data, n, x, hops, r_list, method):
    """NodeClusteringNAFS._k_hop_cluster up to the KMeans call (sgl/tasks/node_clustering.py:205-252)"""
    x = np.asarray(x, dtype=np.float32)
    per_r = []
    for r in r_list:
        norm = sym_norm_csr(indptr, indices, data, n, r)
        feats = propagate(norm, x, hops)
        if method == "simple":
            return feats[-1]
        per_r.append(agg_over_smooth_distance(feats))
    if method == "mean":
        return agg_mean(per_r, 0, len(per_r))
    if method == "max":
        return agg_max(per_r, 0, len(per_r))
    return np.hstack(per_r)


def nafs_task_sweep(indptr, indices, data, n, x, hops_list, r_list, method):
    """NodeClusteringNAFS._execute's loop (sgl/tasks/node_clustering.py:139,176-178): _k_hop_cluster(hop) for every hop count of
    `hops` (an int means range(hops)), each one independently from X_0 exactly as the reference does -- the checker of the
    one-propagation sweep (sgl_amd.tricks.nafs_ensemble_sweep).  Returns {hop count: features}."""
    hops_list = range(hops_list) if isinstance(hops_list, int) else hops_list
    return {int(h): nafs_task_features(indptr, indices, data, n, x, int(h), r_list, method) for h in hops_list}


def coo_to_csr(row, col, data, n):
    """Edge's csr_matrix((w,(row,col)), shape=(n,n)) (sgl/data/base_data.py:29): float32, duplicates summed in input
    order, sorted columns.  Returns (indptr int64, indices int32, values float32)."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    data = np.asarray(data, dtype=np.float32)
    order = np.argsort(row * n + col, kind="stable")
    r, c, v = row[order], col[order], data[order]
    head = np.ones(len(r), dtype=bool)
    head[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
    starts = np.flatnonzero(head)
    ends = np.append(starts[1:], len(r))
    vals = np.empty(len(starts), dtype=np.float32)
    for k, (a, b) in enumerate(zip(starts, ends)):      # sequential fp32 adds, input order
        acc = v[a]
        for j in range(a + 1, b):
            acc = np.float32(acc + v[j])
        vals[k] = acc
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(r[head], minlength=n), out=indptr[1:])
    return indptr, c[head].astype(np.int32), vals


# ----------------------------------------------------------------------------------------------
# tolerance definition used everywhere
# ----------------------------------------------------------------------------------------------
TRUTH_FLOOR = 16 * 2.0 ** -24          # ~9.5e-7: sixteen float32 roundings relative to the largest entry of the truth


def truth_report(got, ref32, truth, factor=2.0, floor=TRUTH_FLOOR, cond=None):
    """A tolerance DERIVED from the reference's own float32 error instead of chosen: with `truth` the same quantity computed by the
    reference's modules in float64 (tests/golden/g12_fp64_truth.npz), pass iff for every element

        |got - truth| <= max(factor * max|ref32 - truth|, floor * max|truth|, floor * cond)

    i.e. the HIP path may be at most `factor` times as far from the truth as the reference's float32 result is -- both are float32
    evaluations of the same expression in different summation orders.  `floor` (default 16 float32 roundings of the largest entry)
    covers the cases where the reference happens to land within an ulp or two of the truth (its error is then a lucky sample, not a
    bound).  `cond` (same shape as truth, optional): for entries that are SUMS of many cancelling terms -- the weight / bias gradients
    of a Linear: sum over rows of dS x -- the sum of the ABSOLUTE terms (recorded with the truth): any summation order is within
    ~log2(N) roundings of that magnitude, while the error relative to the cancelled result is a single random draw for which no
    ratio to another single draw (the reference's) can be guaranteed."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref32 = np.asarray(ref32, dtype=np.float64).reshape(-1)
    truth = np.asarray(truth, dtype=np.float64).reshape(-1)
    if not (got.shape == ref32.shape == truth.shape):
        return {"ok": False, "why": f"shapes {got.shape} {ref32.shape} {truth.shape}"}
    if truth.size == 0:
        return {"ok": True, "err_got": 0.0, "err_ref": 0.0, "bound": floor, "ratio": 0.0}
    mx = float(np.abs(truth).max())
    mx = mx if mx > 0 else 1.0
    e_got = float(np.abs(got - truth).max() / mx)
    e_ref = float(np.abs(ref32 - truth).max() / mx)
    bound = np.full(truth.shape, max(factor * e_ref, floor) * mx)
    if cond is not None:
        bound = np.maximum(bound, floor * np.abs(np.asarray(cond, dtype=np.float64).reshape(-1)))
    ok = bool((np.abs(got - truth) <= bound).all())
    return {"ok": ok, "err_got": e_got, "err_ref": e_ref, "bound": float(max(factor * e_ref, floor)),
            "cond_bound_max": float((bound / mx).max()), "ratio": (e_got / e_ref) if e_ref > 0 else float("inf") if e_got > 0 else 0.0}


def parity_report(y, ref, tol=1e-5, scale=None, rowwise=True):
    """Three-way tolerance of SURVEY.md section 8(c).

    scale (optional, same shape as ref, >= 0): the condition-aware magnitude |A| . |X| of each output element.
    Any summation order of a length-n fp32 dot product satisfies |err| <= n * 2^-24 * scale, while the error
    relative to the RESULT is unbounded when the terms cancel (the reference itself is off by up to 1.3e0
    element-wise-relative against fp64 on cancelling entries).  When given, the row-wise
    criterion divides by max(|ref_row|_2, |scale_row|_2) instead of |ref_row|_2 alone.
    rowwise=False drops the row criterion.  BOTH are honoured for one-column outputs only (see below): every wider comparison is
    held to the unrelaxed three-way test."""
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if y.shape != ref.shape:
        return {"ok": False, "why": f"shape {y.shape} vs {ref.shape}"}
    if y.size == 0:
        return {"ok": True, "max_abs_over_max": 0.0, "row_l2_rel": 0.0, "allclose": True, "bit_equal": True}
    finite = np.isfinite(ref)
    same_nonfinite = np.array_equal(np.isnan(y), np.isnan(ref)) and np.array_equal(y[~finite & ~np.isnan(ref)], ref[~finite & ~np.isnan(ref)])
    yf, rf = np.where(finite, y, 0.0), np.where(finite, ref, 0.0)
    mx = np.abs(rf).max()
    diff = np.abs(yf - rf)
    g = float(diff.max() / mx) if mx > 0 else float(diff.max())
    y2, r2 = yf.reshape(len(yf), -1), rf.reshape(len(rf), -1)
    rn = np.sqrt((r2 * r2).sum(1))
    # The two relaxations only ever apply to ONE-COLUMN outputs, where a "row" is a single element and the row criterion degenerates
    # into the element-wise relative error that SURVEY 8(c) itself calls meaningless on cancelling entries.  The audit of the whole
    # GPU suite (profiles/r05_tolerance_audit.md: 712 relaxed comparisons) found that exactly those -- 10, all of width 1 -- need
    # it; every wider comparison passes the unrelaxed three-way test, which is therefore what it gets, whatever the caller passed.
    if r2.shape[1] > 1 and os.environ.get("SGL_PARITY_RELAX_ALL") != "1":
        scale, rowwise = None, True
    if scale is not None:
        s2 = np.asarray(scale, dtype=np.float64).reshape(len(rf), -1)
        rn = np.maximum(rn, np.sqrt((s2 * s2).sum(1)))
    dn = np.sqrt(((y2 - r2) ** 2).sum(1))
    with np.errstate(divide="ignore", invalid="ignore"):
        rr = np.where(rn > 0, dn / rn, np.where(dn > 0, np.inf, 0.0))
    row = float(rr.max()) if rr.size else 0.0
    ac = bool(np.allclose(yf, rf, rtol=tol, atol=tol * mx))
    ok = bool(same_nonfinite and g <= tol and (row <= tol or not rowwise) and ac)
    audit = os.environ.get("SGL_PARITY_AUDIT")
    if audit and ok and (scale is not None or not rowwise):       # (run with SGL_PARITY_RELAX_ALL=1 to audit every width again)
        # tolerance audit (profiles/r05_tolerance_audit.md): would this comparison also pass the UNRELAXED SURVEY 8(c) criterion --
        # row norm of the reference alone, row criterion on?  Every call that needs the relaxation is recorded with its test id.
        rn0 = np.sqrt((r2 * r2).sum(1))
        with np.errstate(divide="ignore", invalid="ignore"):
            rr0 = np.where(rn0 > 0, dn / rn0, np.where(dn > 0, np.inf, 0.0))
        row0 = float(rr0.max()) if rr0.size else 0.0
        with open(audit, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "?"), "shape": list(ref.shape), "tol": tol,
                                "relaxation": ("scale" if scale is not None else "") + ("" if rowwise else " rowwise=False"),
                                "needs_relaxation": bool(row0 > tol), "row_l2_rel_unrelaxed": row0, "row_l2_rel_used": row,
                                "max_abs_over_max": g}) + "\n")
    return {"ok": ok, "max_abs_over_max": g, "row_l2_rel": row, "allclose": ac,
            "nonfinite_match": bool(same_nonfinite), "bit_equal": bool(np.array_equal(y, ref))}


def parity_ok(y, ref, tol=1e-5, scale=None, rowwise=True):
    """pass iff max|d|/max|ref| <= tol AND max_rows |d_row|2/|ref_row|2 <= tol AND
    allclose(rtol=tol, atol=tol*max|ref|)   """
    return parity_report(y, ref, tol, scale, rowwise)["ok"]
