"""Ingest: SGL's `Custom_Homo` raw layout -> device-resident adjacency, without PyG / OGB / scipy in the way.

Reference: sgl/dataset/custom_dataset.py:38-87 (raw files under `<root>/<name>/raw/`):
    x.npy              float features [N, d]                       (optional if num_node is given)
    adj_matrix.npz     COO edge list: arrays `row`, `col`, `data`  (required)
    label.npy          [N] class ids or [N, C] one-hot             (optional)
    indices.npz        train_idx / val_idx / test_idx              (optional)
and sgl/data/base_data.py:29, where `Edge` turns the COO arrays into csr_matrix((data,(row,col))) -- float32,
duplicates summed, columns sorted.  That scipy build is the only way a large graph (ogbn-papers100M has no loader in
the reference, dataset/ogbn.py:14) could enter SGL; here the same CSR is built on the GPU (sgl_coo_to_csr)."""
import ctypes
import os
from ctypes import c_int64

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream_ptr, lib, ptr

__all__ = ["DeviceAdjacency", "coo_to_csr_device", "load_custom_homo_raw", "save_custom_homo_raw"]


class DeviceAdjacency:
    """Canonical CSR of the (un-normalised) adjacency resident on one GPU: rowptr int64, col int32, val float32.
    GraphOp.propagate accepts it in place of a scipy matrix (everything then stays on the device)."""

    def __init__(self, rowptr, col, val, shape):
        self.rowptr, self.col, self.val = rowptr, col, val
        self.shape = (int(shape[0]), int(shape[1]))
        self.device = rowptr.device

    @property
    def nnz(self):
        return int(self.col.numel())

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val.cpu().numpy(), self.col.cpu().numpy(), self.rowptr.cpu().numpy()), shape=self.shape)

    @classmethod
    def from_scipy(cls, adj, device="cuda"):
        from .operators.utils import canonical_csr
        adj = canonical_csr(adj)
        return cls(torch.from_numpy(adj.indptr.astype(np.int64)).to(device),
                   torch.from_numpy(adj.indices.astype(np.int32)).to(device),
                   torch.from_numpy(adj.data.astype(np.float32)).to(device), adj.shape)


def coo_to_csr_device(row, col, data, num_node, device="cuda"):
    """Edge's csr_matrix((data,(row,col)), shape=(N,N)) on the GPU -> DeviceAdjacency.
    row / col: integer arrays or tensors; data: float array or tensor."""
    _lib.require_gpu()
    device = torch.device(device)
    r = torch.as_tensor(row).to(device=device, dtype=torch.int64).contiguous().view(-1)
    c = torch.as_tensor(col).to(device=device, dtype=torch.int64).contiguous().view(-1)
    v = torch.as_tensor(data).to(device=device, dtype=torch.float32).contiguous().view(-1)
    if not (r.numel() == c.numel() == v.numel()):
        raise ValueError("row, col and data must have the same length")
    nnz = r.numel()
    out_ptr = torch.empty(num_node + 1, dtype=torch.int64, device=device)
    out_col = torch.empty(max(nnz, 1), dtype=torch.int32, device=device)
    out_val = torch.empty(max(nnz, 1), dtype=torch.float32, device=device)
    n_out = c_int64(0)
    with torch.cuda.device(device):
        check(lib().sgl_coo_to_csr(num_node, num_node, nnz, ptr(r), ptr(c), ptr(v), ptr(out_ptr), ptr(out_col), ptr(out_val),
                                   ctypes.byref(n_out), current_stream_ptr()), "sgl_coo_to_csr")
    m = n_out.value
    return DeviceAdjacency(out_ptr, out_col[:m].clone(), out_val[:m].clone(), (num_node, num_node))


def load_custom_homo_raw(raw_dir, num_node=0, device="cuda"):
    """Read the Custom_Homo raw files and build the adjacency on the device.
    Returns dict(adj=DeviceAdjacency, x=ndarray|None, y=LongTensor|None, train_idx, val_idx, test_idx)."""
    def have(f):
        return os.path.exists(os.path.join(raw_dir, f))

    x = np.load(os.path.join(raw_dir, "x.npy")) if have("x.npy") else None
    if x is not None:
        if num_node:
            assert num_node == x.shape[0], 'every node should have a feature vector'
        else:
            num_node = x.shape[0]
    elif not num_node:
        raise ValueError('please provide either feature matrix or number of node')
    if not have("adj_matrix.npz"):
        raise ValueError('the adjacency matrix in coo-format is necessary')
    f = np.load(os.path.join(raw_dir, "adj_matrix.npz"))
    adj = coo_to_csr_device(f["row"], f["col"], f["data"], num_node, device=device)
    y = None
    if have("label.npy"):
        lab = np.load(os.path.join(raw_dir, "label.npy"))
        if lab.ndim == 2:
            lab = np.argmax(lab, 1)
        y = torch.LongTensor(lab)
    splits = {"train_idx": None, "val_idx": None, "test_idx": None}
    if have("indices.npz"):
        s = np.load(os.path.join(raw_dir, "indices.npz"))
        for k in splits:
            if k in s:
                splits[k] = s[k]
    return dict(adj=adj, x=x, y=y, num_node=num_node, **splits)


def save_custom_homo_raw(raw_dir, row, col, data, x=None, labels=None, train_idx=None, val_idx=None, test_idx=None):
    """write a graph in the Custom_Homo raw layout (used by tests and to hand synthetic graphs to the reference)"""
    os.makedirs(raw_dir, exist_ok=True)
    np.savez(os.path.join(raw_dir, "adj_matrix.npz"), row=np.asarray(row), col=np.asarray(col), data=np.asarray(data))
    if x is not None:
        np.save(os.path.join(raw_dir, "x.npy"), np.asarray(x))
    if labels is not None:
        np.save(os.path.join(raw_dir, "label.npy"), np.asarray(labels))
    idx = {k: v for k, v in (("train_idx", train_idx), ("val_idx", val_idx), ("test_idx", test_idx)) if v is not None}
    if idx:
        np.savez(os.path.join(raw_dir, "indices.npz"), **idx)
