"""Community-aware partitioning of a matrix whose storage is ALREADY row-sharded, without any rank ever holding the whole of it.

`ShardedGraphOp(partition="community")` relabels the nodes in a plan-time community order before the matrix is cut into row
blocks, so that a block references mostly its own rows and the need-aware exchange moves a fraction of the foreign rows
(profiles/r03_partition_locality.log).  For a full adjacency on every rank that is a local computation.  For `RowBlock` input the
first implementation assembled the normalised matrix on every rank to find and apply the relabelling -- 27 GB of CSR per rank at
papers100M size, and the full feature matrix (57 GB) next to it.  Here every step works on the ranks' own rows:

  labels    semi-synchronous label propagation -- on the device the very per-node kernel of sgl_reorder_community, one round at a time
            (sgl_reorder_lpa_round); on CPU tensors the tensor-code statement of the same algorithm
            (sgl_amd.reorder.community_order_reference): a rank updates the labels of ITS rows from the labels of their neighbours; what is replicated is the label
            VECTOR (one integer per node: 0.9 GB at papers100M size, against 27 GB of matrix), refreshed by an all-gather of
            the ranks' slices per round
  order     the stable sort by label of that vector: identical on every rank, no communication
  bounds    nnz-balanced row blocks of the RELABELLED matrix from the all-gathered row degrees (one integer per node)
  rows      every rank sends each of its rows -- row id, column ids (relabelled) and values -- to the rank that owns the row's new
            id (one variable-size exchange) and builds its new block with the library's COO -> CSR kernel (sgl_coo_to_csr:
            sorted columns, the canonical form the whole-matrix path produces)
  features  a rank asks the owners (old ids) for exactly the feature rows its new compact table holds (fetch_rows)

The result -- block boundaries, the block's CSR arrays, the node ids of its rows -- equals what the whole-matrix path computes
(tests: both inputs give the same hops): on the device both run the same per-node kernel."""
import numpy as np
import torch
import torch.distributed as dist

from .layout import balanced_bounds
from .sharded_adj import RowBlock, _world

__all__ = ["exchange_var", "sharded_community_order", "redistribute_rows", "fetch_rows", "sharded_edge_locality"]


def _staged(group, t):
    return bool(t.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo")


def _gather_ints(values, group):
    """[world, len(values)] int64 table of every rank's small integer list (identical everywhere)"""
    world = _world(group)[1]
    if world == 1:
        return np.asarray([values], dtype=np.int64)
    everyone = [None] * world
    dist.all_gather_object(everyone, [int(v) for v in values], group=group)
    return np.asarray(everyone, dtype=np.int64)


def exchange_var(sends, group=None):
    """Variable-size exchange: sends[q] (a tensor, possibly empty; same dtype and trailing shape on every rank) goes to rank q of
    the group; returns the list of what every rank sent here (recv[q] from rank q; the own share is passed through).  One table
    of counts (all-gather), then one batch of point-to-point transfers -- staged through the host where the process group cannot
    move device memory (gloo)."""
    rank, world = _world(group)
    if world == 1:
        return [sends[0]]
    counts = _gather_ints([int(t.shape[0]) for t in sends], group)          # counts[src, dst]
    like = sends[rank]
    staged = _staged(group, like)
    recvs = [None] * world
    ops, keep = [], []
    for q in range(world):
        gq = dist.get_global_rank(group, q) if group is not None else q
        if q == rank:
            recvs[q] = sends[q]
            continue
        n_in = int(counts[q, rank])
        buf = torch.empty((n_in,) + tuple(like.shape[1:]), dtype=like.dtype, device="cpu" if staged else like.device)
        recvs[q] = buf
        if n_in:
            ops.append(dist.P2POp(dist.irecv, buf, gq, group=group))
        if sends[q].shape[0]:
            s = sends[q].contiguous()
            if staged:
                s = s.cpu()
            keep.append(s)
            ops.append(dist.P2POp(dist.isend, s, gq, group=group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    if staged:
        recvs = [r if q == rank else r.to(like.device) for q, r in enumerate(recvs)]
    return recvs


def _allgather_slices(local, bounds, n, group):
    """[n] vector on every rank from the ranks' slices (slice of rank q = entries [bounds[q], bounds[q + 1]))"""
    rank, world = _world(group)
    out = torch.empty(n, dtype=local.dtype, device=local.device)
    out[int(bounds[rank]):int(bounds[rank + 1])] = local
    if world == 1:
        return out
    staged = _staged(group, local)
    for q in range(world):
        gq = dist.get_global_rank(group, q) if group is not None else q
        a, b = int(bounds[q]), int(bounds[q + 1])
        if b == a:
            continue
        piece = out[a:b]
        if staged:
            h = piece.cpu() if q == rank else torch.empty(b - a, dtype=local.dtype)
            dist.broadcast(h, gq, group=group)
            if q != rank:
                piece.copy_(h)
        else:
            dist.broadcast(piece, gq, group=group)
    return out


def _mix(x, salt):
    x = (x ^ salt) * 0x9E3779B97F4A7C15
    x = x & 0x7FFFFFFFFFFFFFFF
    return (x >> 29) ^ x


@torch.no_grad()
def sharded_community_order(block, bounds, group=None, rounds=8):
    """order[i] = new id of node i (int64 [n], identical on every rank) from label propagation over the ranks' row blocks.
    block: this rank's rows [lo, hi) with GLOBAL column ids (structure only is used); bounds: the ranks' row boundaries.
    The same rounds, activity masks and tie rule as sgl_amd.reorder.community_order_reference on the whole matrix."""
    n, lo, hi = block.n, block.lo, block.hi
    dev = block.device
    if block.rowptr.is_cuda:
        return _sharded_community_order_device(block, bounds, group, rounds)
    deg = (block.rowptr[1:] - block.rowptr[:-1]).to(torch.int64)
    row = torch.repeat_interleave(torch.arange(hi - lo, device=dev, dtype=torch.int64), deg)      # LOCAL row of every non-zero
    colq = block.col.to(torch.int64)
    labels = torch.arange(n, device=dev, dtype=torch.int64)
    ids = torch.arange(lo, hi, device=dev, dtype=torch.int64)
    moved = 0
    for it in range(rounds):
        mine = labels[lo:hi]
        if row.numel():
            key, _ = torch.sort(row * n + labels[colq])                   # (local node, neighbour's label)
            run, cnt = torch.unique_consecutive(key, return_counts=True)
            node, lab = run // n, run % n
            score = cnt * n + (n - 1 - lab)                               # most frequent, ties to the smaller label
            best = torch.full((hi - lo,), -1, dtype=torch.int64, device=dev)
            best.scatter_reduce_(0, node, score, reduce="amax", include_self=True)
            new = torch.where(best >= 0, n - 1 - best % n, mine)
            del key, run, cnt, node, lab, score, best
        else:
            new = mine
        active = (_mix(ids, it * 2654435761 + 12345) & 1) == (it & 1) if it < rounds - 1 else torch.ones_like(ids, dtype=torch.bool)
        upd = active & (new != mine)
        moved = int(upd.sum())
        labels = _allgather_slices(torch.where(upd, new, mine), bounds, n, group)
    perm = torch.argsort(labels, stable=True)                              # communities by smallest label, members in id order
    order = torch.empty_like(perm)
    order[perm] = torch.arange(n, device=dev, dtype=torch.int64)
    total_moved = int(_gather_ints([moved], group).sum())
    n_comm = int(torch.unique(labels).numel())
    return order, f"{n_comm} communities after {rounds} rounds, {total_moved} nodes moved in the last"


@torch.no_grad()
def _sharded_community_order_device(block, bounds, group, rounds):
    """the device form: every round is sgl_reorder_lpa_round -- the per-node kernel of sgl_reorder_community (one wavefront per
    node, a strided sample of at most 256 neighbours) -- on this rank's rows, the slices all-gathered between rounds; the stable sort
    by label is torch's.  Identical to what the whole-matrix kernel computes for ANY graph, long rows included."""
    import ctypes
    from .. import _lib
    n, lo, hi = block.n, block.lo, block.hi
    dev = block.device
    rowptr = block.rowptr.to(torch.int64).contiguous()
    col = block.col.to(torch.int32).contiguous()
    labels = torch.arange(n, device=dev, dtype=torch.int32)
    moved = torch.zeros(1, dtype=torch.int64, device=dev)
    for it in range(rounds):
        new = torch.empty(hi - lo, dtype=torch.int32, device=dev)
        moved.zero_()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().sgl_reorder_lpa_round(_lib.ptr(rowptr), _lib.ptr(col), hi - lo, lo, n, _lib.ptr(labels), _lib.ptr(new), it,
                                                        1 if it == rounds - 1 else 0, ctypes.c_void_p(moved.data_ptr()),
                                                        _lib.current_stream_ptr()), "sgl_reorder_lpa_round")
        labels = _allgather_slices(new, bounds, n, group)
    perm = torch.argsort(labels, stable=True)
    order = torch.empty(n, dtype=torch.int64, device=dev)
    order[perm] = torch.arange(n, device=dev, dtype=torch.int64)
    total_moved = int(_gather_ints([int(moved.item())], group).sum())
    n_comm = int(torch.unique(labels).numel())
    return order, f"{n_comm} communities after {rounds} rounds, {total_moved} nodes moved in the last"


@torch.no_grad()
def sharded_edge_locality(block, order, group=None):
    """reorder.edge_locality of the whole matrix from the ranks' blocks (hits summed over the ranks)"""
    from ..reorder import edge_locality
    window = max(256, min(65536, block.n // 16))                             # the window the whole-matrix measure uses
    frac = edge_locality(block.rowptr, block.col, order, window=window, row0=block.lo) if block.nnz else 0.0
    t = _gather_ints([round(frac * block.nnz), block.nnz], group)
    return float(t[:, 0].sum()) / max(float(t[:, 1].sum()), 1.0)


@torch.no_grad()
def redistribute_rows(block, order, group=None):
    """The relabelled matrix P A P^T cut into nnz-balanced row blocks, this rank's block of it, from the ranks' blocks of A --
    no rank holds more than its old and its new rows.  block: rows [lo, hi) of A (global column ids, any values -- e.g. the
    NORMALISED block); order: int64 [n], order[i] = new id of node i (identical on every rank).
    Returns (new RowBlock in the relabelled ids, bounds [world + 1] of the new blocks)."""
    import ctypes
    from .. import _lib
    rank, world = _world(group)
    n, dev = block.n, block.device
    if not block.rowptr.is_cuda:
        raise RuntimeError("redistribute_rows builds the new block with the library's device kernel: the row block must live on a GPU")
    deg = (block.rowptr[1:] - block.rowptr[:-1]).to(torch.int64)
    old_bounds = _gather_ints([block.lo, block.hi], group)
    ob = [int(v) for v in old_bounds[:, 0]] + [int(old_bounds[-1, 1])]
    deg_all = _allgather_slices(deg, ob, n, group)                           # one integer per node
    deg_new = torch.empty_like(deg_all)
    deg_new[order] = deg_all
    rowptr_new = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr_new[1:] = torch.cumsum(deg_new, 0)
    bounds = balanced_bounds(rowptr_new.cpu().numpy(), world)                # the cut the whole-matrix path makes
    nlo, nhi = int(bounds[rank]), int(bounds[rank + 1])
    # every non-zero as (new row, new col, value), sorted by destination rank
    new_row = torch.repeat_interleave(order[block.lo:block.hi], deg)
    new_col = order[block.col.to(torch.int64)]
    bt = torch.from_numpy(np.asarray(bounds[1:-1], dtype=np.int64)).to(dev)
    dest = torch.searchsorted(bt, new_row, right=True) if world > 1 else torch.zeros_like(new_row)
    by_dest = torch.argsort(dest, stable=True)
    cnt = torch.bincount(dest, minlength=world).cpu().tolist()
    new_row, new_col, val = new_row[by_dest], new_col[by_dest], block.val[by_dest]
    offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    rc = torch.stack([new_row, new_col], dim=1)                              # one [m, 2] int64 stream + one float stream
    got_rc = exchange_var([rc[int(offs[q]):int(offs[q + 1])] for q in range(world)], group)
    got_v = exchange_var([val[int(offs[q]):int(offs[q + 1])] for q in range(world)], group)
    rc = torch.cat(got_rc, 0) if world > 1 else got_rc[0]
    vv = torch.cat(got_v, 0) if world > 1 else got_v[0]
    m = int(rc.shape[0])
    rows_l = (rc[:, 0] - nlo).contiguous()
    cols_g = rc[:, 1].contiguous()
    vv = vv.contiguous()
    n_loc = nhi - nlo
    out_ptr = torch.empty(n_loc + 1, dtype=torch.int64, device=dev)
    out_col = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    out_val = torch.empty(max(m, 1), dtype=torch.float32, device=dev)
    n_out = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().sgl_coo_to_csr(n_loc, n, m, _lib.ptr(rows_l), _lib.ptr(cols_g), _lib.ptr(vv), _lib.ptr(out_ptr),
                                             _lib.ptr(out_col), _lib.ptr(out_val), ctypes.byref(n_out), _lib.current_stream_ptr()),
                   "sgl_coo_to_csr")
    k = n_out.value
    return RowBlock(nlo, nhi, n, out_ptr, out_col[:k].clone(), out_val[:k].clone()), bounds


@torch.no_grad()
def fetch_rows(x_local, owner_bounds, want_ids, group=None):
    """x[want_ids] ([len(want_ids), d]) where the rows of x are spread over the ranks: rank q holds rows
    [owner_bounds[q], owner_bounds[q + 1]) as x_local.  Every rank sends each owner the list of ids it wants from it, the owners
    gather those rows (one kernel) and send them back; rows come out in the order of want_ids.  Collective."""
    from .. import device as dev_
    rank, world = _world(group)
    dev = x_local.device
    want = want_ids.to(device=dev, dtype=torch.int64)
    lo = int(owner_bounds[rank])
    if world == 1:
        return dev_.gather_rows(x_local, want - lo)
    bt = torch.from_numpy(np.asarray(owner_bounds[1:-1], dtype=np.int64)).to(dev)
    owner = torch.searchsorted(bt, want, right=True)
    by_owner = torch.argsort(owner, stable=True)
    cnt = torch.bincount(owner, minlength=world).cpu().tolist()
    offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    asked = want[by_owner]
    requests = exchange_var([asked[int(offs[q]):int(offs[q + 1])] for q in range(world)], group)     # ids others want from me
    replies = []
    for q in range(world):
        ids = requests[q]
        replies.append(dev_.gather_rows(x_local, ids - lo).contiguous() if ids.numel()
                       else torch.empty((0, x_local.shape[1]), dtype=x_local.dtype, device=dev))
    got = exchange_var(replies, group)
    rows = torch.cat(got, 0)                                                   # in `asked` order (grouped by owner)
    out = torch.empty((want.numel(), x_local.shape[1]), dtype=x_local.dtype, device=dev)
    out[by_owner] = rows
    return out
