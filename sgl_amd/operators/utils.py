"""MI355X counterparts of sgl/operators/utils.py (same function names, argument meaning and errors).

reference                                   here
------------------------------------------  ---------------------------------------------------------
csr_sparse_dense_matmul      utils.py:10-40  same ctypes call shape, bound to libsgl_hip.so's
                                             FloatCSRMulDenseOMP shim (host pointers, accumulate semantics)
cuda_csr_sparse_dense_matmul utils.py:43-73  same, bound to FloatCSRMulDense (overwrite semantics)
adj_to_symmetric_norm        utils.py:76-88  device normalisation (sgl_norm_*), result as scipy CSR (fp64)
one_dim_weighted_add         utils.py:91-102 HIP kernel (+ autograd)
two_dim_weighted_add         utils.py:105-116 HIP kernel (+ autograd)
"""
import ctypes
from ctypes import c_int

import numpy as np
import numpy.ctypeslib as ctl
import scipy.sparse as sp
import torch
from torch import Tensor

from .. import _lib
from .. import device as dev


def _arr_types():
    arr_1d_int = ctl.ndpointer(dtype=np.int32, ndim=1, flags="CONTIGUOUS")
    arr_1d_float = ctl.ndpointer(dtype=np.float32, ndim=1, flags="CONTIGUOUS")
    return arr_1d_int, arr_1d_float


def csr_sparse_dense_matmul(adj, feature):
    """Drop-in for sgl/operators/utils.py:10-40: scipy CSR x float32 ndarray -> float32 ndarray, computed on the
    MI355X through the reference-signature symbol FloatCSRMulDenseOMP (matmul.h:5)."""
    _lib.require_gpu()
    ctl_lib = ctypes.CDLL(_lib.LIB_PATH)  # the library the reference would ctl.load_library()
    arr_1d_int, arr_1d_float = _arr_types()
    ctl_lib.FloatCSRMulDenseOMP.argtypes = [arr_1d_float, arr_1d_float, arr_1d_int, arr_1d_int, arr_1d_float, c_int, c_int]
    ctl_lib.FloatCSRMulDenseOMP.restype = None

    answer = np.zeros(feature.shape).astype(np.float32).flatten()
    data = adj.data.astype(np.float32)
    indices = adj.indices
    indptr = adj.indptr
    mat = feature.flatten()
    mat_row, mat_col = feature.shape

    ctl_lib.FloatCSRMulDenseOMP(answer, data, indices, indptr, mat, mat_row, mat_col)
    err = _lib.last_error()
    if err:
        raise _lib.SglHipError(f"FloatCSRMulDenseOMP: {err}")
    return answer.reshape(feature.shape)


def cuda_csr_sparse_dense_matmul(adj, feature):
    """Drop-in for sgl/operators/utils.py:43-73 (the reference's dead cuSPARSE wrapper), bound to FloatCSRMulDense."""
    _lib.require_gpu()
    ctl_lib = ctypes.CDLL(_lib.LIB_PATH)
    arr_1d_int, arr_1d_float = _arr_types()
    ctl_lib.FloatCSRMulDense.argtypes = [arr_1d_float, c_int, arr_1d_float, arr_1d_int, arr_1d_int, arr_1d_float, c_int, c_int]
    ctl_lib.FloatCSRMulDense.restype = c_int

    answer = np.zeros(feature.shape).astype(np.float32).flatten()
    data = adj.data.astype(np.float32)
    data_nnz = len(data)
    indices = adj.indices
    indptr = adj.indptr
    mat = feature.flatten()
    mat_row, mat_col = feature.shape

    rc = ctl_lib.FloatCSRMulDense(answer, data_nnz, data, indices, indptr, mat, mat_row, mat_col)
    if rc != 0:
        raise _lib.SglHipError(f"FloatCSRMulDense failed: {_lib.last_error()}")
    return answer.reshape(feature.shape)


def canonical_csr(adj):
    """scipy coo/csr -> canonical CSR (sorted columns, duplicates summed, float32) without touching the input"""
    if isinstance(adj, sp.coo_matrix):
        adj = adj.tocsr()
    if not adj.has_canonical_format:
        adj = adj.copy()
        adj.sum_duplicates()
    return adj


def adj_to_symmetric_norm_device(adj, r, alpha=None, device=None, return_fp64=False):
    """A_hat = D^{r-1} (A+I)^T D^{-r} [-> (1-alpha) A_hat + alpha I] computed on the GPU.
    adj: scipy coo/csr.  Returns device tensors (rowptr int64, col int32, val float32[, val float64])."""
    _lib.require_gpu()
    adj = canonical_csr(adj)
    n = adj.shape[0]
    if adj.shape[0] != adj.shape[1]:
        raise ValueError("the adjacency matrix must be square")
    device = torch.device(device or "cuda")
    rowptr = torch.from_numpy(adj.indptr.astype(np.int64)).to(device)
    col = torch.from_numpy(adj.indices.astype(np.int32)).to(device)
    val = torch.from_numpy(adj.data.astype(np.float32)).to(device)
    return dev.normalize_adj(rowptr, col, val, n, r, alpha, return_fp64=return_fp64)


def adj_to_symmetric_norm(adj, r):
    """Same contract as sgl/operators/utils.py:76-88 (returns a scipy sparse matrix with float64 values); the
    arithmetic runs on the MI355X."""
    n = adj.shape[0]
    rowptr, col, _, v64 = adj_to_symmetric_norm_device(adj, r, None, return_fp64=True)
    return sp.csr_matrix((v64.cpu().numpy(), col.cpu().numpy(), rowptr.cpu().numpy()), shape=(n, n))


def _as_device_list(feat_list):
    """CPU tensors are uploaded (the result goes back to their device): the kernels are GPU-only."""
    first = feat_list[0]
    if first.is_cuda:
        return list(feat_list), None
    _lib.require_gpu()
    return [dev.upload_rows(f, "cuda") if not f.requires_grad else f.to("cuda") for f in feat_list], first.device


def _rowmajor(f):
    if f.dim() != 2:
        raise ValueError("feature matrices must be 2-D")
    if f.dtype != torch.float32:
        f = f.float()
    if (f.shape[1] > 1 and f.stride(1) != 1) or (f.shape[0] > 1 and f.stride(0) < f.shape[1]):
        f = f.contiguous()
    return f


def one_dim_weighted_add(feat_list, weight_list):
    if not isinstance(feat_list, list) or not isinstance(weight_list, Tensor):
        raise TypeError("This function is designed for list(feature) and tensor(weight)!")
    elif len(feat_list) != weight_list.shape[0]:
        raise ValueError("The feature list and the weight list have different lengths!")
    elif len(weight_list.shape) != 1:
        raise ValueError("The weight list should be a 1d tensor!")
    feats, home = _as_device_list(feat_list)
    feats = [_rowmajor(f) for f in feats]
    out = dev.hop_wsum1d(feats, weight_list.to(feats[0].device))
    return out if home is None else out.to(home)


def two_dim_weighted_add(feat_list, weight_list):
    if not isinstance(feat_list, list) or not isinstance(weight_list, Tensor):
        raise TypeError("This function is designed for list(feature) and tensor(weight)!")
    elif len(feat_list) != weight_list.shape[1]:
        raise ValueError("The feature list and the weight list have different lengths!")
    elif len(weight_list.shape) != 2:
        raise ValueError("The weight list should be a 2d tensor!")
    feats, home = _as_device_list(feat_list)
    feats = [_rowmajor(f) for f in feats]
    out = dev.hop_wsum2d(feats, weight_list.to(feats[0].device))
    return out if home is None else out.to(home)
