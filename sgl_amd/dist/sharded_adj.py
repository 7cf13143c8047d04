"""Row-sharded STORAGE of the adjacency ("each GPU holds its CSR shard -- local row pointers, global
column ids"): helpers that move, cut and re-assemble row blocks so that no rank ever needs the whole matrix.

    RowBlock                    rows [lo, hi) of an n x n matrix on one device (local rowptr, global col ids)
    scatter_row_blocks          one rank holds the matrix (e.g. it generated it), every rank receives ITS rows only
    block_piece_spmms           cut the local block into nnz-balanced row pieces -> one SpMM callable per piece
    gather_piece_bounds         the [world, pieces+1] table of absolute piece boundaries ShardedPropagator wants
    allgather_blocks            re-assemble the whole CSR on every rank (only for layouts / checks that need a replica)
    balanced_bounds_device      nnz-balanced block boundaries from a row-pointer vector that lives on the device

Everything is plain torch + torch.distributed on whatever device the tensors live on (HIP with RCCL, or CPU with gloo in
the tests); the arithmetic -- normalisation and SpMM -- stays in the HIP library."""
from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

from .layout import piece_bounds


@dataclass
class RowBlock:
    lo: int
    hi: int
    n: int                      # global number of rows = number of columns
    rowptr: torch.Tensor        # int64 [hi - lo + 1], local (rowptr[0] == 0)
    col: torch.Tensor           # int32 [nnz_local], GLOBAL column ids
    val: torch.Tensor           # float32 [nnz_local]

    @property
    def n_local(self):
        return self.hi - self.lo

    @property
    def shape(self):
        return (self.hi - self.lo, self.n)

    @property
    def nnz(self):
        return int(self.col.numel())

    @property
    def device(self):
        return self.rowptr.device


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _host_staged(group, t):
    """gloo moves device memory only in broadcast / all_reduce: everything else goes through the host there (tests with
    several ranks on one GPU; RCCL takes device tensors directly)"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _send(t, dst, group):
    dist.send(t.cpu() if _host_staged(group, t) else t, dst, group=group)


def _recv(t, src, group):
    if _host_staged(group, t):
        buf = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(buf, src, group=group)
        t.copy_(buf)
    else:
        dist.recv(t, src, group=group)


def balanced_bounds_device(rowptr, parts):
    """balanced_bounds() for a row-pointer vector on the device: only the parts+1 boundaries cross to the host"""
    n = rowptr.numel() - 1
    cost = rowptr + torch.arange(n + 1, dtype=rowptr.dtype, device=rowptr.device)
    share = torch.arange(1, parts, dtype=torch.float64, device=rowptr.device) / parts
    target = (cost[-1].to(torch.float64) * share).ceil().to(cost.dtype)
    cuts = torch.searchsorted(cost, target, right=False).clamp_(0, n).cpu().numpy().astype(np.int64)
    b = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


def scatter_row_blocks(full, bounds, n, device, src=0, group=None):
    """Rank `src` holds `full` = (rowptr int64 [n+1], col int32, val float32) on `device`; every rank (src included) gets
    the RowBlock [bounds[rank], bounds[rank+1]).  Nobody but src ever sees more than its own rows.  Point-to-point
    transfers (works on RCCL and gloo)."""
    rank, world = _world(group)
    bounds = np.asarray(bounds, dtype=np.int64)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    if world == 1:
        rowptr, col, val = full
        return RowBlock(0, n, n, rowptr, col, val)
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    offs = torch.zeros(world + 1, dtype=torch.int64, device=device)
    if rank == src:
        rowptr, col, val = full
        offs.copy_(rowptr[torch.from_numpy(bounds).to(device)])
    dist.broadcast(offs, gsrc, group=group)
    offs_h = offs.cpu().numpy()
    if rank == src:
        for q in range(world):
            if q == src:
                continue
            gq = dist.get_global_rank(group, q) if group is not None else q
            a, b = int(bounds[q]), int(bounds[q + 1])
            _send((rowptr[a:b + 1] - rowptr[a]).contiguous(), gq, group)
            if offs_h[q + 1] > offs_h[q]:
                _send(col[offs_h[q]:offs_h[q + 1]].contiguous(), gq, group)
                _send(val[offs_h[q]:offs_h[q + 1]].contiguous(), gq, group)
        mine = ((rowptr[lo:hi + 1] - rowptr[lo]).contiguous(), col[offs_h[rank]:offs_h[rank + 1]].clone(),
                val[offs_h[rank]:offs_h[rank + 1]].clone())
        return RowBlock(lo, hi, n, *mine)
    nnz = int(offs_h[rank + 1] - offs_h[rank])
    rp = torch.empty(hi - lo + 1, dtype=torch.int64, device=device)
    cc = torch.empty(nnz, dtype=torch.int32, device=device)
    vv = torch.empty(nnz, dtype=torch.float32, device=device)
    _recv(rp, gsrc, group)
    if nnz:
        _recv(cc, gsrc, group)
        _recv(vv, gsrc, group)
    return RowBlock(lo, hi, n, rp, cc, vv)


def canonicalize_block(block):
    """Sort the columns inside every row of a block and sum duplicate (row, col) entries -- what the normalisation needs
    and what a generated / ingested row block does not guarantee (sgl_coo_to_csr: 64-bit keys, stable radix sort,
    duplicate sums in input order).  Returns a new RowBlock; the input is not modified."""
    import ctypes
    from .. import _lib
    _lib.require_gpu()
    dev = block.device
    n_loc, nnz = block.n_local, block.nnz
    rows = torch.repeat_interleave(torch.arange(n_loc, dtype=torch.int64, device=dev), block.rowptr[1:] - block.rowptr[:-1])
    cols = block.col.to(torch.int64)
    out_ptr = torch.empty(n_loc + 1, dtype=torch.int64, device=dev)
    out_col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    out_val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
    n_out = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().sgl_coo_to_csr(n_loc, block.n, nnz, _lib.ptr(rows), _lib.ptr(cols), _lib.ptr(block.val), _lib.ptr(out_ptr),
                                             _lib.ptr(out_col), _lib.ptr(out_val), ctypes.byref(n_out), _lib.current_stream_ptr()),
                   "sgl_coo_to_csr")
    m = n_out.value
    return RowBlock(block.lo, block.hi, block.n, out_ptr, out_col[:m].clone(), out_val[:m].clone())


def local_piece_bounds(block, pieces, weights=None):
    """absolute boundaries [pieces+1] of the nnz-balanced row pieces of this rank's block"""
    rp_host = block.rowptr.cpu().numpy()
    return piece_bounds(rp_host, 0, block.n_local, pieces, weights) + block.lo, rp_host


def global_nnz(block, group=None):
    """non-zeros of the whole matrix = the sum of the ranks' blocks (one all-reduce of one integer; collective)"""
    t = torch.tensor([block.nnz], dtype=torch.int64, device=block.device)
    rank, world = _world(group)
    if world > 1:
        if t.is_cuda and dist.get_backend(group) == "gloo":
            t = t.cpu()
        dist.all_reduce(t, group=group)
    return int(t.item())


def block_piece_spmms(block, pieces, weights=None, strict=False, reorder=None, total_nnz=None):
    """One DeviceCSR (rows of the piece x all n columns) per local row piece, built from the rank's OWN rows only.
    Returns (callables f(x_full, out), handles, absolute piece boundaries [pieces+1]).
    reorder (None / "community" / "auto"): every piece's rows are stored and processed in a locality order found on the piece's
    own diagonal part, behind a row map (sgl_amd.reorder.local_rowmap): bit-identical results.
    total_nnz: non-zeros of the WHOLE matrix (global_nnz(block)): long rows are then cut where the unsharded matrix would cut them."""
    from ..device import DeviceCSR, default_long_row_nnz, permute_rows
    pb, rp_host = local_piece_bounds(block, pieces, weights)
    fns, handles = [], []
    for p in range(pieces):
        r0, r1 = int(pb[p]) - block.lo, int(pb[p + 1]) - block.lo
        nb, ne = int(rp_host[r0]), int(rp_host[r1])
        rp_local = (block.rowptr[r0:r1 + 1] - block.rowptr[r0]).contiguous()
        cc, vv, rowmap = block.col[nb:ne], block.val[nb:ne], None
        if reorder and r1 > r0 and cc.is_cuda:
            from ..reorder import local_rowmap
            rowmap, _ = local_rowmap(rp_local, cc, int(pb[p]), int(pb[p + 1]), reorder)
            if rowmap is not None:
                rp_local, cc, vv = permute_rows(rp_local, cc.contiguous(), vv.contiguous(), rowmap)
        h = DeviceCSR(rp_local, cc, vv, (r1 - r0, block.n), strict=strict,
                      long_row_nnz=default_long_row_nnz(total_nnz) if total_nnz is not None else 0)
        if rowmap is not None:
            h.set_rowmap(rowmap)
        handles.append(h)
        fns.append(lambda x, out, h=h: h.spmm(x, out=out))
    return fns, handles, pb


def gather_piece_bounds(my_bounds, group=None):
    """[world, pieces+1] table of every rank's absolute piece boundaries (identical on all ranks)"""
    rank, world = _world(group)
    mine = [int(v) for v in my_bounds]
    if world == 1:
        return np.asarray([mine], dtype=np.int64)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    pb = np.asarray(everyone, dtype=np.int64)
    if not (pb[1:, 0] == pb[:-1, -1]).all() or pb[0, 0] != 0:
        raise RuntimeError("the ranks' row blocks do not tile the matrix")
    return pb


def allgather_blocks(block, group=None):
    """(rowptr [n+1], col, val) of the WHOLE matrix on every rank, assembled from the ranks' blocks (one broadcast per
    rank and array).  Only the layouts that multiply all rows (feature-sharded, grid) and replica-based checks need it."""
    rank, world = _world(group)
    if world == 1:
        return block.rowptr, block.col, block.val
    sizes = [None] * world
    dist.all_gather_object(sizes, (block.lo, block.hi, block.nnz), group=group)
    dev = block.device
    total = sum(s[2] for s in sizes)
    rowptr = torch.zeros(block.n + 1, dtype=torch.int64, device=dev)
    col = torch.empty(total, dtype=torch.int32, device=dev)
    val = torch.empty(total, dtype=torch.float32, device=dev)
    off = 0
    for q, (lo, hi, nnz) in enumerate(sizes):
        gq = dist.get_global_rank(group, q) if group is not None else q
        rp = (block.rowptr + off) if q == rank else torch.empty(hi - lo + 1, dtype=torch.int64, device=dev)
        dist.broadcast(rp, gq, group=group)
        rowptr[lo:hi + 1] = rp
        if nnz:
            cs, vs = col[off:off + nnz], val[off:off + nnz]
            if q == rank:
                cs.copy_(block.col)
                vs.copy_(block.val)
            dist.broadcast(cs, gq, group=group)
            dist.broadcast(vs, gq, group=group)
        off += nnz
    return rowptr, col, val


def allgather_rows(local, bounds, n, group=None, out=None):
    """[n, d] matrix on every rank from the ranks' row shards (the replica of the features a row-sharded job starts from)"""
    rank, world = _world(group)
    if out is None:
        out = torch.empty((n, local.shape[1]), dtype=local.dtype, device=local.device)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    out[lo:hi].copy_(local)
    if world == 1:
        return out
    for q in range(world):
        gq = dist.get_global_rank(group, q) if group is not None else q
        a, b = int(bounds[q]), int(bounds[q + 1])
        if b > a:
            dist.broadcast(out[a:b], gq, group=group)
    return out


def exchange_checksums(replica, my_shard, bounds, group=None):
    """Exact integrity check of an all-gathered feature replica: every rank publishes the wrapping int64 sum of the raw
    bits of ITS rows, and compares the sums of all row ranges of its replica against what the owners published.
    Order-independent and exact (integer arithmetic on the bit patterns).  Returns True iff every range matches."""
    rank, world = _world(group)

    def bits_sum(t):
        # int64 accumulation in bounded chunks: torch widens the operand before reducing, and the replica can be tens of GB
        total = torch.zeros((), dtype=torch.int64, device=t.device)
        flat = t.contiguous().view(torch.int32).view(-1)
        for part in flat.split(1 << 27):
            total += part.to(torch.int64).sum()
        return total

    mine = bits_sum(my_shard).reshape(1)
    if world == 1:
        return bool(bits_sum(replica[int(bounds[0]):int(bounds[1])]) == mine[0])
    everyone = [None] * world
    dist.all_gather_object(everyone, int(mine.item()), group=group)
    ok = True
    for q in range(world):
        ok = ok and int(bits_sum(replica[int(bounds[q]):int(bounds[q + 1])]).item()) == everyone[q]
    return ok
