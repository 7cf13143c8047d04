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

__all__ = ["DeviceAdjacency", "coo_to_csr_device", "load_custom_homo_raw", "load_custom_homo_raw_sharded",
           "save_custom_homo_raw"]


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


def coo_to_csr_device(row, col, data, num_node, device="cuda", num_col=None):
    """Edge's csr_matrix((data,(row,col)), shape=(N,N)) on the GPU -> DeviceAdjacency.
    row / col: integer arrays or tensors; data: float array or tensor.  `num_col` (default num_node) makes the result
    rectangular: a row block of a larger matrix (local row ids, global column ids)."""
    _lib.require_gpu()
    device = torch.device(device)
    num_col = num_node if num_col is None else int(num_col)
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
        check(lib().sgl_coo_to_csr(num_node, num_col, nnz, ptr(r), ptr(c), ptr(v), ptr(out_ptr), ptr(out_col), ptr(out_val),
                                   ctypes.byref(n_out), current_stream_ptr()), "sgl_coo_to_csr")
    m = n_out.value
    return DeviceAdjacency(out_ptr, out_col[:m].clone(), out_val[:m].clone(), (num_node, num_col))


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


def _edge_chunks(n_edges, chunk_edges):
    for a in range(0, n_edges, chunk_edges):
        yield a, min(n_edges, a + chunk_edges)


def load_custom_homo_raw_sharded(raw_dir, rank, world, num_node=0, device="cuda", chunk_edges=1 << 27, group=None):
    """The same raw files -> THIS rank's row block only ("direct sharded CSR build, int64-safe"), so
    a graph whose CSR does not fit one GPU -- or simply should never be replicated -- goes from disk straight into the
    row-sharded storage ShardedGraphOp / ShardedPropagator work on.

      pass 1  edges per row, counted on the device chunk by chunk (with a process group: every rank counts its share of
              the chunks, one all-reduce) -> nnz-balanced block boundaries, identical on every rank
      pass 2  chunk by chunk: keep the edges whose row lies in [lo, hi) -> local row ids, GLOBAL column ids
      build   sgl_coo_to_csr on the kept edges (duplicates summed, columns sorted: Edge's csr_matrix semantics)

    The device holds one chunk plus this rank's edges at a time (edge counts are 64-bit; one rank's share must stay
    below 2^32 - 1 entries).  NumPy's npz reader has no partial reads, so the HOST reads the three COO arrays whole.
    x.npy is memory-mapped and only rows [lo, hi) are read.
    Returns dict(block=RowBlock, bounds=int64[world+1], x=ndarray[hi-lo, d]|None, y, num_node, train/val/test_idx)."""
    _lib.require_gpu()
    from .dist.sharded_adj import RowBlock, balanced_bounds_device
    if not (0 <= rank < world):
        raise ValueError("rank must lie in [0, world)")
    device = torch.device(device)

    def have(f):
        return os.path.exists(os.path.join(raw_dir, f))

    xmap = np.load(os.path.join(raw_dir, "x.npy"), mmap_mode="r") if have("x.npy") else None
    if xmap is not None:
        if num_node:
            assert num_node == xmap.shape[0], 'every node should have a feature vector'
        else:
            num_node = xmap.shape[0]
    elif not num_node:
        raise ValueError('please provide either feature matrix or number of node')
    if not have("adj_matrix.npz"):
        raise ValueError('the adjacency matrix in coo-format is necessary')
    f = np.load(os.path.join(raw_dir, "adj_matrix.npz"))
    row, col, data = f["row"], f["col"], f["data"]
    if not (row.shape == col.shape == data.shape) or row.ndim != 1:
        raise ValueError("row, col and data must be 1-D arrays of the same length")
    n_edges = int(row.shape[0])
    chunks = list(_edge_chunks(n_edges, int(chunk_edges)))
    use_group = group is not None and world > 1
    counts = torch.zeros(num_node, dtype=torch.int64, device=device)
    for i, (a, b) in enumerate(chunks):
        if use_group and i % world != rank:
            continue
        r = torch.from_numpy(np.ascontiguousarray(row[a:b])).to(device=device, dtype=torch.int64)
        if r.numel() and (int(r.min()) < 0 or int(r.max()) >= num_node):
            raise ValueError("a row index lies outside [0, num_node)")
        counts += torch.bincount(r, minlength=num_node)
    if use_group:
        import torch.distributed as dist
        if counts.is_cuda and dist.get_backend(group) == "gloo":
            host = counts.cpu()
            dist.all_reduce(host, group=group)
            counts.copy_(host)
        else:
            dist.all_reduce(counts, group=group)
    rowptr_raw = torch.zeros(num_node + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rowptr_raw[1:])
    bounds = balanced_bounds_device(rowptr_raw, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    mine = int(rowptr_raw[hi] - rowptr_raw[lo])
    del counts, rowptr_raw
    kr = torch.empty(mine, dtype=torch.int64, device=device)
    kc = torch.empty(mine, dtype=torch.int64, device=device)
    kv = torch.empty(mine, dtype=torch.float32, device=device)
    at = 0
    for a, b in chunks:
        r = torch.from_numpy(np.ascontiguousarray(row[a:b])).to(device=device, dtype=torch.int64)
        keep = torch.nonzero((r >= lo) & (r < hi)).view(-1)
        m = int(keep.numel())
        if m == 0:
            continue
        kr[at:at + m] = r[keep] - lo
        del r
        kc[at:at + m] = torch.from_numpy(np.ascontiguousarray(col[a:b])).to(device=device, dtype=torch.int64)[keep]
        kv[at:at + m] = torch.from_numpy(np.ascontiguousarray(data[a:b])).to(device=device, dtype=torch.float32)[keep]
        at += m
    assert at == mine, "edge count changed between the two passes"
    part = coo_to_csr_device(kr, kc, kv, hi - lo, device=device, num_col=num_node)
    block = RowBlock(lo, hi, num_node, part.rowptr, part.col, part.val)
    x = np.ascontiguousarray(xmap[lo:hi]) if xmap is not None else None
    y = None
    if have("label.npy"):
        lab = np.load(os.path.join(raw_dir, "label.npy"))
        if lab.ndim == 2:
            lab = np.argmax(lab, 1)
        y = torch.LongTensor(lab)
    splits = {"train_idx": None, "val_idx": None, "test_idx": None}
    if have("indices.npz"):
        sidx = np.load(os.path.join(raw_dir, "indices.npz"))
        for k in splits:
            if k in sidx:
                splits[k] = sidx[k]
    return dict(block=block, bounds=bounds, x=x, y=y, num_node=num_node, **splits)


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
