"""Plan-time node ordering for locality ("Locality ordering"): real co-purchase / citation graphs have communities, and a
gathered row of X that was fetched for one member of a community is fetched again for the next -- if the members are processed
close together it is still in L2 / the Infinity Cache.  A processing order that keeps communities together turns that into
hits; ids as they come out of a dump carry no such order.

    order, info = community_order(rowptr, col, n)      # order[i] = position of node i (a permutation), found on the device

GraphOp(reorder="community") applies it without relabelling anything: rowmap = argsort(order), the rows of A_hat are stored in
that order (device.permute_rows) and the SpMM handle writes storage row i to output row rowmap[i] (DeviceCSR.set_rowmap) --
bit-identical results.  permute_csr relabels the whole problem instead (P A P^T, canonical): the comparison point of
tools/bench_reorder.py and a utility for callers who want the relabelled graph.

The ordering is a few rounds of semi-synchronous label propagation (every node adopts the most frequent label among its
neighbours, ties to the smaller label; half of the nodes move per round so two-coloured structures cannot oscillate), then a
stable sort by label: sgl_reorder_community in the HIP library (one wavefront per node, the neighbours' labels counted in LDS).
It is a heuristic that runs once per graph; nothing of the propagation path depends on it.  `community_order_reference` is the same algorithm in plain tensor
code -- the readable statement the tests pin the kernel to (identical labels when no node has more than 256 neighbours; the
kernel samples longer rows)."""
import ctypes

import torch

from . import _lib, io
from ._lib import check, current_stream_ptr, lib, ptr

__all__ = ["community_order", "community_order_reference", "permute_csr", "edge_locality", "plan_rowmap", "plan_order", "local_rowmap"]


@torch.no_grad()
def community_order(rowptr, col, n, rounds=8):
    """rowptr int64 [n+1], col int32 [nnz] on the GPU (a symmetric adjacency; self-loops do not matter).
    Returns (order int64 [n] with order[i] = new id of node i, info string)."""
    _lib.require_gpu()
    if not (rowptr.is_cuda and col.is_cuda):
        raise ValueError("community_order runs on the device: pass device tensors (sgl_amd.io.DeviceAdjacency)")
    rowptr = rowptr.to(torch.int64).contiguous()
    col = col.to(torch.int32).contiguous()
    order = torch.empty(n, dtype=torch.int64, device=rowptr.device)
    info = (ctypes.c_int64 * 2)(0, 0)
    with torch.cuda.device(rowptr.device):
        check(lib().sgl_reorder_community(ptr(rowptr), ptr(col), n, int(rounds), ptr(order), info, current_stream_ptr()),
              "sgl_reorder_community")
    return order, f"{info[0]} communities after {int(rounds)} rounds, {info[1]} nodes moved in the last"


def _mix(x, salt):
    x = (x ^ salt) * 0x9E3779B97F4A7C15
    x = x & 0x7FFFFFFFFFFFFFFF
    return (x >> 29) ^ x


@torch.no_grad()
def community_order_reference(rowptr, col, n, rounds=8):
    """the algorithm of sgl_reorder_community in tensor code (any device; every neighbour counted, no sampling)"""
    device = rowptr.device
    deg = rowptr[1:] - rowptr[:-1]
    row = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg)
    colq = col.to(torch.int64)
    labels = torch.arange(n, device=device, dtype=torch.int64)
    ids = torch.arange(n, device=device, dtype=torch.int64)
    moved = n
    for it in range(rounds):
        key = row * n + labels[colq]                                  # (node, neighbour's label)
        key, _ = torch.sort(key)
        run, cnt = torch.unique_consecutive(key, return_counts=True)   # one entry per (node, label) with its multiplicity
        node, lab = run // n, run % n
        score = cnt * n + (n - 1 - lab)                               # most frequent, ties to the smaller label
        best = torch.full((n,), -1, dtype=torch.int64, device=device)
        best.scatter_reduce_(0, node, score, reduce="amax", include_self=True)
        new = torch.where(best >= 0, n - 1 - best % n, labels)
        active = (_mix(ids, it * 2654435761 + 12345) & 1) == (it & 1) if it < rounds - 1 else torch.ones_like(ids, dtype=torch.bool)
        upd = active & (new != labels)
        moved = int(upd.sum())
        labels = torch.where(upd, new, labels)
        del key, run, cnt, node, lab, score, best, new
    # communities in the order of their smallest label, members in id order
    perm = torch.argsort(labels, stable=True)                          # perm[k] = old id at new position k
    order = torch.empty_like(perm)
    order[perm] = ids
    n_comm = int(torch.unique(labels).numel())
    return order, f"{n_comm} communities after {it + 1} rounds, {moved} nodes moved in the last"


@torch.no_grad()
def permute_csr(rowptr, col, val, order):
    """P A P^T for the permutation order[i] = new id of node i; canonical CSR out (rows and columns relabelled, columns
    sorted within a row) through the library's COO -> CSR build."""
    n = rowptr.numel() - 1
    device = rowptr.device
    deg = rowptr[1:] - rowptr[:-1]
    row = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg)
    out = io.coo_to_csr_device(order[row], order[col.to(torch.int64)], val, n, device=device)
    return out.rowptr, out.col, out.val


@torch.no_grad()
def edge_locality(rowptr, col, order=None, window=None, row0=0):
    """Fraction of the non-zeros (i, j) whose two ends sit within `window` positions of each other in the processing order
    (order[i] = position of node i; None = the ids as they are).  What the plan-time ordering is judged by: a gathered row of X
    is re-used from L2 / the Infinity Cache when its readers are processed close together.  rowptr / col may describe a
    rectangular row block whose first row is node `row0`; columns outside the order's range count as far."""
    n = rowptr.numel() - 1
    nnz = int(col.numel())
    if n == 0 or nnz == 0:
        return 0.0
    m = order.numel() if order is not None else None
    window = int(window or max(256, min(65536, (m or n) // 16)))
    device = rowptr.device
    hits = 0
    step = 1 << 26
    rp = rowptr.to(torch.int64)
    for s0 in range(0, nnz, step):
        e0 = min(nnz, s0 + step)
        pos = torch.arange(s0, e0, dtype=torch.int64, device=device)
        row = torch.searchsorted(rp, pos, right=True) - 1 + row0
        c = col[s0:e0].to(torch.int64)
        if order is None:
            hits += int(((row - c).abs() < window).sum())
        else:
            inside = (c >= 0) & (c < m) & (row < m)
            pr = order[row.clamp(0, m - 1)]
            pc = order[c.clamp(0, m - 1)]
            hits += int((inside & ((pr - pc).abs() < window)).sum())
    return hits / nnz


AUTO_MIN_LOCALITY = 0.30     # reorder="auto": at least this share of the edges must end up local ...
AUTO_MIN_GAIN = 0.15         # ... and at least this much more than in the order the ids come in


@torch.no_grad()
def plan_rowmap(rowptr, col, n, mode):
    """The row map (int32 [n]: rowmap[k] = node processed k-th) a plan should use, or None.  mode: None / "community" / "auto".
    "auto" runs the label propagation (70 ms at products size) and keeps its order only if it makes the graph measurably more
    local than its own ids do (edge_locality: >= 30 % of the edges local and >= 15 points more than before): a random graph
    -- the benchmark workloads -- or one whose ids already follow its communities is left alone.  Returns (rowmap, info)."""
    if mode is None:
        return None, {"reorder": None}
    if mode not in ("community", "auto"):
        raise ValueError("reorder must be None, 'community' or 'auto'")
    order, text = community_order(rowptr, col, n)
    info = {"reorder": mode, "communities": text}
    if mode == "auto":
        before, after = edge_locality(rowptr, col), edge_locality(rowptr, col, order)
        use = after >= AUTO_MIN_LOCALITY and after >= before + AUTO_MIN_GAIN
        info.update({"edge_locality_before": round(before, 4), "edge_locality_after": round(after, 4), "applied": bool(use)})
        if not use:
            return None, info
    else:
        info["applied"] = True
    return torch.argsort(order).to(torch.int32), info


@torch.no_grad()
def plan_order(rowptr, col, n, mode):
    """order[i] = new id of node i for a RELABELLING of the problem (P A P^T: what a community-aware partition cuts), or None.
    Same decision rule as plan_rowmap ("auto": only when the graph becomes measurably more local).  Returns (order, info)."""
    if mode is None:
        return None, {"partition": None}
    if mode not in ("community", "auto"):
        raise ValueError("partition must be None, 'community' or 'auto'")
    order, text = community_order(rowptr, col, n)
    info = {"partition": mode, "communities": text, "applied": True}
    if mode == "auto":
        before, after = edge_locality(rowptr, col), edge_locality(rowptr, col, order)
        use = after >= AUTO_MIN_LOCALITY and after >= before + AUTO_MIN_GAIN
        info.update({"edge_locality_before": round(before, 4), "edge_locality_after": round(after, 4), "applied": bool(use)})
        if not use:
            return None, info
    return order, info


@torch.no_grad()
def local_rowmap(rowptr, col, lo, hi, mode):
    """plan_rowmap for a RECTANGULAR row block (rows [lo, hi) of the matrix, global or compact column ids in which the block's own
    nodes are columns [lo, hi)): the ordering is found on the block's diagonal part -- the edges between its own nodes -- which is
    all a rank of the row-sharded layout can see without communication; rows whose community shows there are processed
    together and gather the same foreign rows.  Returns (rowmap over LOCAL rows or None, info)."""
    if mode is None:
        return None, {"reorder": None}
    n_loc = hi - lo
    if n_loc <= 1:
        return None, {"reorder": mode, "applied": False}
    c = col.to(torch.int64)
    own = (c >= lo) & (c < hi)
    cnt = torch.zeros(n_loc + 1, dtype=torch.int64, device=rowptr.device)
    rows = torch.searchsorted(rowptr.to(torch.int64), torch.arange(c.numel(), dtype=torch.int64, device=c.device), right=True) - 1
    cnt[1:] = torch.bincount(rows[own], minlength=n_loc)
    d_ptr = torch.cumsum(cnt, 0)
    d_col = (c[own] - lo).to(torch.int32)
    return plan_rowmap(d_ptr, d_col, n_loc, mode)
