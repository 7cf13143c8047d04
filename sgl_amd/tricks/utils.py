"""label_propagation on the MI355X.  Reference: sgl/tricks/utils.py:41-58.

    out = H0 (or H0 restricted to `mask`);  res = (1 - alpha) * out
    repeat num_layers:  out = post_process(alpha * (adj @ out) + res)

One fused HIP kernel per layer (sgl_spmm_axpb_clamp_f32: SpMM + scale + residual + clamp in the epilogue, same
rounding order as the reference's `alpha * torch.spmm(...) + res`); adj, H0 and both ping-pong buffers stay in HBM.
The reference's torch-COO `spmm` sums each row in storage order like the CSR kernel does."""
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from .. import _lib
from .. import device as dev

_adj_cache = {}


def device_adj(adj, device):
    """scipy sparse (already normalised by the caller, as in the reference) -> cached DeviceCSR"""
    if isinstance(adj, dev.DeviceCSR):
        return adj
    if not sp.issparse(adj):
        raise TypeError("adj must be a scipy sparse matrix (or a DeviceCSR)")
    key = (id(adj), adj.shape, adj.nnz, str(device))
    hit = _adj_cache.get(key)
    if hit is not None and hit[0] is adj:
        return hit[1]
    csr = adj.tocsr()
    if not csr.has_canonical_format:
        csr = csr.copy()
        csr.sum_duplicates()
    h = dev.DeviceCSR.from_scipy(csr, device=device)
    _adj_cache.clear()          # keep one entry: the loop calls this with the same matrix over and over
    _adj_cache[key] = (adj, h)
    return h


def _clamp_range(post_process):
    """(lo, hi) when the post-processing step is a clamp the kernel can fuse, else None"""
    if post_process is None:
        return (float("-inf"), float("inf"))
    if isinstance(post_process, (tuple, list)) and len(post_process) == 2:
        return (float(post_process[0]), float(post_process[1]))
    return None


@torch.no_grad()
def label_propagation(labels, adj, num_layers, alpha, post_process=(0., 1.), mask=None, device="cuda"):
    """Same contract as the reference function; `post_process` may be a (lo, hi) pair (fused clamp, the default
    reproduces the reference's `lambda x: x.clamp_(0., 1.)`), None, or any callable applied to the device tensor
    after every layer.  Returns a CUDA tensor [N, C]."""
    _lib.require_gpu()
    device = torch.device(device)
    if labels.dtype == torch.long:
        labels = F.one_hot(labels.reshape(-1)).to(torch.float)
    labels = labels.to(device=device, dtype=torch.float32)
    out = dev.upload_rows(labels, device)
    if mask is not None:
        m = torch.as_tensor(mask).to(device)
        kept = torch.zeros_like(out)
        kept[m] = out[m]
        out = dev.upload_rows(kept, device)
    csr = device_adj(adj, device)
    res = dev.upload_rows((1 - alpha) * out, device)      # H_0 term, rounded once like the reference (:53)
    rng = _clamp_range(post_process)
    bufs = [dev.alloc_rows(out.shape[0], out.shape[1], device) for _ in range(2)]
    cur = out
    for layer in range(num_layers):
        nxt = bufs[layer % 2]
        if rng is not None:
            csr.spmm_axpb_clamp(dev.padded_parent(cur), alpha, dev.padded_parent(res), rng[0], rng[1],
                                out=dev.padded_parent(nxt))
        else:
            csr.spmm_axpb_clamp(dev.padded_parent(cur), alpha, dev.padded_parent(res), out=dev.padded_parent(nxt))
            r = post_process(nxt)
            if r is not None and r is not nxt:
                nxt.copy_(r)
        cur = nxt
    return cur
