"""Row-sharded multi-GPU pre-propagation: one process per GPU, adjacency rows partitioned across ranks,
the dense feature block all-gathered between hops over xGMI (RCCL through torch.distributed).

The reference has no multi-GPU propagation at all; this is new capability with the
same mathematical result: rank g owns the contiguous row block [b_g, b_{g+1}) of A_hat (balanced by non-zeros)
and of every hop matrix.  Per hop:

    Y_g = A_hat[b_g:b_{g+1}, :] @ X            local HIP SpMM, computed in `pieces` row pieces
    X'  = concat_g(Y_g)                        direct all-gather: every rank pushes each finished piece to its
                                               peers with grouped point-to-point send/recv (all xGMI links busy
                                               at once, no ring), while the next piece is still being computed

The last hop needs no exchange.  Aggregators are row-wise, so they run on the local shards with zero traffic.
Nothing here touches the data path on the host: buffers stay in HBM; torch.distributed is plumbing.

Grid layouts (GridLayout): SpMM is separable over feature columns, and 288 GB of HBM hold a replica of A_hat on every
GPU, so the G ranks can also be arranged as Gr row blocks x Gc column slices.  A rank then multiplies its row block of
A_hat with ITS column slice only and exchanges rows only inside its column group (Gr ranks):

    Gc = G (feature-sharded): every rank runs the whole k-hop chain on d/G columns -- no exchange at all;
    Gr = G (row-sharded):     the scheme above;
    in between:               in-bound bytes per rank per hop drop to (Gr-1)/Gr * N * d/Gc * 4.

xGMI is a point-to-point mesh, so an exchange inside a small column group would use only Gr-1 of a GPU's 7 links.  The
"relay" transport spreads it over all of them: each row piece is cut into G stripes; phase 1 sends stripe q to rank q,
phase 2 has q forward it to the ranks that need it (two link crossings per byte, but 7 links in parallel)."""
import contextlib

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["balanced_bounds", "piece_bounds", "all_piece_bounds", "tapered_weights", "device_piece_spmms", "column_chunks",
           "column_slices", "GridLayout", "ShardedPropagator", "ShardedGraphOp"]


def balanced_bounds(rowptr, parts, weights=None):
    """Cut rows [0, n) into `parts` contiguous blocks holding ~equal numbers of non-zeros (+1 per row so
    empty rows still count), or shares proportional to `weights`.  rowptr: host int64 array [n+1].
    Returns int64 array [parts+1]."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n = len(rowptr) - 1
    cost = rowptr + np.arange(n + 1, dtype=np.int64)
    if weights is None:
        share = np.arange(1, parts, dtype=np.float64) / parts
    else:
        w = np.asarray(weights, dtype=np.float64)
        if len(w) != parts or (w <= 0).any():
            raise ValueError("one positive weight per part")
        share = np.cumsum(w)[:-1] / w.sum()
    cuts = np.searchsorted(cost, cost[-1] * share, side="left").astype(np.int64)
    b = np.concatenate([[0], np.clip(cuts, 0, n), [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


def tapered_weights(pieces):
    """piece sizes for a hop whose LAST piece's transfer cannot hide behind compute: equal pieces, the last one half
    as large (4 pieces -> 2:2:2:1, the exposed transfer is 1/7 instead of 1/4 of the hop's traffic)"""
    return [2.0] * (pieces - 1) + [1.0] if pieces > 1 else [1.0]


def piece_bounds(rowptr, lo, hi, pieces, weights=None):
    """split the row block [lo, hi) into `pieces` nnz-balanced sub-blocks -> absolute row boundaries [pieces+1]"""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    local = rowptr[lo:hi + 1] - rowptr[lo]
    return balanced_bounds(local, pieces, weights) + lo


def all_piece_bounds(rowptr_host, world, pieces, weights=None):
    """[world, pieces+1] absolute row boundaries: rank blocks balanced by non-zeros, each cut into `pieces`
    (equal, or in the proportions `weights`)"""
    bounds = balanced_bounds(rowptr_host, world)
    return np.stack([piece_bounds(rowptr_host, int(bounds[g]), int(bounds[g + 1]), pieces, weights) for g in range(world)])


def device_piece_spmms(rowptr, col, val, n_cols, my_bounds, rowptr_host=None, strict=False):
    """One DeviceCSR (rectangular: rows of the piece x all columns) per local row piece, as `f(x_full, out)`
    callables for ShardedPropagator.  rowptr/col/val: the FULL normalised adjacency on this rank's device (only
    views of the local rows are kept alive); my_bounds: this rank's row of all_piece_bounds()."""
    from .device import DeviceCSR
    if rowptr_host is None:
        rowptr_host = rowptr.cpu().numpy()
    fns, handles = [], []
    for p in range(len(my_bounds) - 1):
        r0, r1 = int(my_bounds[p]), int(my_bounds[p + 1])
        nb, ne = int(rowptr_host[r0]), int(rowptr_host[r1])
        rp_local = (rowptr[r0:r1 + 1] - rowptr[r0]).contiguous()
        h = DeviceCSR(rp_local, col[nb:ne].contiguous(), val[nb:ne].contiguous(), (r1 - r0, n_cols), strict=strict)
        handles.append(h)
        fns.append(lambda x, out, h=h: h.spmm(x, out=out))
    return fns, handles


def column_chunks(d, n_chunks=2):
    """Split the feature dimension into `n_chunks` column ranges whose widths are multiples of 32 floats (one
    128-byte line) except the last: stored as separate contiguous matrices, a chunk row then covers whole cache
    lines and the chunks together touch no more lines than the unsplit row (d = 100 -> 64 + 36: 2 + 2 lines)."""
    if n_chunks <= 1 or d <= 32:
        return [(0, d)]
    width = max(32, ((d + n_chunks - 1) // n_chunks + 31) // 32 * 32)
    out, c = [], 0
    while c < d:
        out.append((c, min(d, c + width)))
        c += width
    return out


def column_slices(d, parts, line=32):
    """Split d feature columns into `parts` contiguous slices (the column groups of a GridLayout).

    Slices are whole 128-byte lines (`line` floats) wherever the column count allows: d = 100 over 4 groups is
    32 + 32 + 32 + 4, not 4 x 25 -- every group still gathers one line per non-zero, but no group stores or exchanges
    7 pad floats per row (-22 % bytes on the wire for the grid layout).  With more parts than lines (d = 100 over 8)
    the split is simply even.  Slices may be empty when d < parts."""
    d, parts = int(d), int(parts)
    units = -(-d // line)
    if parts <= units:
        base, extra = divmod(units, parts)             # lines per part, the first `extra` parts get one more
        out, c = [], 0
        for q in range(parts):
            w = min(d - c, (base + (1 if q < extra else 0)) * line)
            out.append((c, c + w))
            c += w
        return out
    base, extra = divmod(d, parts)
    out, c = [], 0
    for q in range(parts):
        w = base + (1 if q < extra else 0)
        out.append((c, c + w))
        c += w
    return out


class GridLayout:
    """world = row_groups x col_groups ranks.  Rank g works on row block g % row_groups of column slice
    g // row_groups; the ranks of one column group exchange rows between hops, different column groups never talk
    (except as relays of each other's traffic)."""

    def __init__(self, world, row_groups):
        world, row_groups = int(world), int(row_groups)
        if row_groups < 1 or world % row_groups:
            raise ValueError("row_groups must divide the world size")
        self.world, self.row_groups, self.col_groups = world, row_groups, world // row_groups

    def coords(self, g):
        """(row block index, column group index) of global rank g"""
        return g % self.row_groups, g // self.row_groups

    def members(self, cg):
        """global ranks of column group cg, ordered by row block"""
        return [cg * self.row_groups + r for r in range(self.row_groups)]

    def __repr__(self):
        return f"GridLayout({self.row_groups} row blocks x {self.col_groups} column slices)"


class _Works:
    """a set of outstanding transfers: anything with .wait()"""

    def __init__(self, works):
        self.works = list(works)

    def advance(self):
        pass

    def wait(self):
        for w in self.works:
            w.wait()


class _HostStagedXfer:
    """device -> host copies are sent, host receive buffers are copied into their device views on wait()"""

    def __init__(self, works, landings, keep):
        self.works, self.landings, self.keep = works, landings, keep

    def advance(self):
        pass

    def wait(self):
        for w in self.works:
            w.wait()
        for dst, buf in self.landings:
            dst.copy_(buf)


def _post(group, sends, recvs, staged=False):
    """Post sends [(tensor, global peer)] and receives [(tensor view, global peer)] as one batch of point-to-point ops.
    Per pair of ranks the order of the sends equals the order of the matching receives on the other side."""
    sends = [(t, peer) for t, peer in sends if t.numel()]         # zero-width slices (d < column groups): both sides
    recvs = [(t, peer) for t, peer in recvs if t.numel()]         # know the size, both skip
    if not sends and not recvs:
        return _Works([])
    if not staged:
        ops = [dist.P2POp(dist.isend, t, peer, group=group) for t, peer in sends] + \
              [dist.P2POp(dist.irecv, t, peer, group=group) for t, peer in recvs]
        return _Works(dist.batch_isend_irecv(ops))
    # process groups that cannot move device memory (gloo): device -> host -> send/recv -> device.  Slow by
    # construction (PCIe both ways, synchronises the stream); exists so that the sharded paths also run where RCCL
    # is unavailable -- and so that several ranks can be exercised end to end on ONE GPU in the tests.
    keep = [t.detach().cpu() for t, _ in sends]
    ops = [dist.P2POp(dist.isend, h, peer, group=group) for h, (_, peer) in zip(keep, sends)]
    landings = []
    for t, peer in recvs:
        buf = torch.empty(t.shape, dtype=t.dtype)
        ops.append(dist.P2POp(dist.irecv, buf, peer, group=group))
        landings.append((t, buf))
    return _HostStagedXfer(dist.batch_isend_irecv(ops), landings, keep)


# ---- transports: how the rows a rank has just computed reach the ranks of its column group ---------------------------
# begin(p, y_piece, x_next) starts moving row piece p (y_piece = my new rows, x_next = the next hop's replica, where my
# peers' rows must land) and returns a handle with advance() (cheap; called after more compute has been queued) and
# wait().  `prop` is the ShardedPropagator: bounds, ranks, group.

class _DirectTransport:
    """my piece to every rank of my column group and theirs to me, one grouped batch, one link per peer ("p2p";
    "staged" = the same through host memory for process groups that cannot move device memory)"""

    def __init__(self, prop, staged):
        self.prop, self.staged = prop, staged

    def exchanging(self):
        return self.prop.world > 1

    def begin(self, p, y_piece, x_next):
        pr = self.prop
        sends, recvs = [], []
        # stagger the peer order per rank so that at any moment every link carries one transfer
        for k in range(1, pr.world):
            dst, src = (pr.rank + k) % pr.world, (pr.rank - k) % pr.world
            if y_piece.numel():
                sends.append((y_piece, pr.members[dst]))
            r0, r1 = int(pr.pb[src, p]), int(pr.pb[src, p + 1])
            if r1 > r0:
                recvs.append((x_next[r0:r1], pr.members[src]))
        return _post(pr.group, sends, recvs, self.staged)


class _AllGatherTransport:
    """RCCL all_gather_into_tensor on equal-size padded pieces plus a local scatter of the valid rows ("allgather")"""

    class _Work:
        def __init__(self, work, staged, x_next, spans, max_rows):
            self.work, self.staged, self.x_next, self.spans, self.max_rows = work, staged, x_next, spans, max_rows

        def wait(self):
            self.work.wait()
            for q, (r0, r1) in enumerate(self.spans):
                if r1 > r0:
                    self.x_next[r0:r1].copy_(self.staged[q * self.max_rows:q * self.max_rows + (r1 - r0)])

    def __init__(self, prop):
        self.prop, self._buf = prop, {}

    def exchanging(self):
        return self.prop.world > 1

    def begin(self, p, y_piece, x_next):
        pr = self.prop
        w = x_next.shape[1]
        spans = [(int(pr.pb[q, p]), int(pr.pb[q, p + 1])) for q in range(pr.world)]
        max_rows = max(max(r1 - r0 for r0, r1 in spans), 1)
        key = (p, x_next.data_ptr(), w, max_rows)       # one staging pair per (piece, destination buffer): never shared
                                                          # between transfers that can be in flight together
        if key not in self._buf:
            self._buf[key] = (torch.zeros((max_rows, w), dtype=x_next.dtype, device=x_next.device),
                              torch.empty((pr.world * max_rows, w), dtype=x_next.dtype, device=x_next.device))
        inp, out = self._buf[key]
        inp[:y_piece.shape[0]].copy_(y_piece)
        work = dist.all_gather_into_tensor(out, inp, group=pr.group, async_op=True)
        mine = spans[pr.rank]
        spans_remote = [(r0, r1) if q != pr.rank else (mine[0], mine[0]) for q, (r0, r1) in enumerate(spans)]
        return _Works([_AllGatherTransport._Work(work, out, x_next, spans_remote, max_rows)])


class _RelayTransport:
    """Two-phase exchange over ALL ranks of the job (grid layouts): each row piece is cut into `world` stripes; phase 1
    scatters stripe q to rank q -- straight into the replica of a rank of my own column group, into a relay buffer
    elsewhere -- and phase 2 has every rank forward what it holds to the ranks of the owner's column group (and send
    the stripe it kept).  Every link carries 1/world of the piece per phase ("relay"; "relay_staged" through host
    memory).  With 2 row blocks each rank exchanges exactly ONE tensor with every other rank per phase, so on RCCL a
    phase is a single all_to_all on tensor lists instead of a batch of point-to-point ops (the host-side issue cost is
    what bounds the number of pieces a hop can be cut into); its per-piece plan is computed once."""

    class _Exchange:
        """On a GPU process group both phases are issued at once from a side stream (which first waits for the
        piece's SpMM): phase 2 then follows phase 1 on the communicator's stream without ever making the compute
        stream wait, so the SpMM of the next pieces overlaps both.  Elsewhere (gloo: transfers complete asynchronously
        on host threads) phase 2 is posted by advance() once phase 1 has landed."""

        def __init__(self, tr, p, y_piece, x_next):
            self.tr, self.p, self.y_piece, self.x_next = tr, p, y_piece, x_next
            self.phase2 = None
            if x_next.is_cuda and not tr.staged:
                main = torch.cuda.current_stream(x_next.device)
                side = tr.side_stream(x_next.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self.phase1 = tr.phase1(p, y_piece, x_next)
                    self.phase1.wait()                  # the SIDE stream waits; the compute stream runs on
                    self.phase2 = tr.phase2(p, y_piece, x_next)
            else:
                self.phase1 = tr.phase1(p, y_piece, x_next)

        def advance(self):
            if self.phase2 is None:
                self.phase1.wait()
                self.phase2 = self.tr.phase2(self.p, self.y_piece, self.x_next)

        def wait(self):
            self.advance()
            self.phase2.wait()

    def __init__(self, prop, staged):
        self.prop, self.staged = prop, staged
        self._bufs, self._plans, self._side = {}, {}, None

    def exchanging(self):                                 # every rank relays, even if its own column group were one rank
        return self.prop.layout.world > 1 and self.prop.layout.row_groups > 1

    def begin(self, p, y_piece, x_next):
        return _RelayTransport._Exchange(self, p, y_piece, x_next)

    def side_stream(self, device):
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def collective(self):
        pr = self.prop
        if pr.relay_collective is not None:
            return bool(pr.relay_collective)
        return not self.staged and pr.layout.row_groups == 2 and dist.get_backend(pr.group) == "nccl"

    def stripe(self, rg, p, q):
        """absolute rows of stripe q of row block rg's piece p"""
        pr = self.prop
        r0, r1 = int(pr.pb[rg, p]), int(pr.pb[rg, p + 1])
        W = pr.layout.world
        return r0 + (r1 - r0) * q // W, r0 + (r1 - r0) * (q + 1) // W

    def relay_buf(self, p, g, rows, width, like):
        key = (p, g)
        buf = self._bufs.get(key)
        if buf is None or buf.shape != (rows, width) or buf.device != like.device:
            buf = self._bufs[key] = torch.empty((rows, width), dtype=like.dtype, device=like.device)
        return buf

    def dummy(self, like, key):
        buf = self._bufs.get(key)
        if buf is None or buf.device != like.device:
            buf = self._bufs[key] = torch.zeros(1, dtype=like.dtype, device=like.device)
        return buf

    # -- point-to-point form (any number of row blocks, any process group) --
    def phase1(self, p, y_piece, x_next):
        if self.collective():
            return self._a2a(p, 0, y_piece, x_next)
        pr = self.prop
        L, me = pr.layout, pr.me
        my_cg = L.coords(me)[1]
        base = int(pr.pb[pr.rank, p])
        sends, recvs = [], []
        for k in range(1, L.world):
            q, g = (me + k) % L.world, (me - k) % L.world
            a, b = self.stripe(pr.rank, p, q)
            if b > a:
                sends.append((y_piece[a - base:b - base], q))
            rg_g, cg_g = L.coords(g)
            a, b = self.stripe(rg_g, p, me)
            if b > a:
                dst = x_next[a:b] if cg_g == my_cg else self.relay_buf(p, g, b - a, pr.widths[cg_g], x_next)
                recvs.append((dst, g))
        return _post(pr.group, sends, recvs, self.staged)

    def phase2(self, p, y_piece, x_next):
        if self.collective():
            return self._a2a(p, 1, y_piece, x_next)
        pr = self.prop
        L, me = pr.layout, pr.me
        my_cg = L.coords(me)[1]
        base = int(pr.pb[pr.rank, p])
        sends, recvs = [], []
        for k in range(1, L.world):
            dst, src = (me + k) % L.world, (me - k) % L.world
            cg_d = L.coords(dst)[1]
            for g in L.members(cg_d):
                if g == dst:
                    continue
                a, b = self.stripe(L.coords(g)[0], p, me)
                if b <= a:
                    continue
                if g == me:
                    t = y_piece[a - base:b - base]            # my own stripe never left
                elif cg_d == my_cg:
                    t = x_next[a:b]                           # I needed it myself: phase 1 put it into my replica
                else:
                    t = self.relay_buf(p, g, b - a, pr.widths[cg_d], x_next)
                sends.append((t, dst))
            for g in pr.members:
                if g == me:
                    continue
                a, b = self.stripe(L.coords(g)[0], p, src)
                if b > a:
                    recvs.append((x_next[a:b], src))
        return _post(pr.group, sends, recvs, self.staged)

    # -- all_to_all form (2 row blocks) --
    def plan(self, p):
        """Static part of the two all_to_all calls of piece p, computed once: per peer what to send and where to
        receive, as (kind, a, b, aux) with kind 0 = 1-element placeholder (nothing to move: the same on both sides),
        1 = rows [a, b) of my piece (relative), 2 = rows [a, b) of the next-hop replica, 3 = relay buffer
        aux = (source rank, rows, width)."""
        plan = self._plans.get(p)
        if plan is not None:
            return plan
        pr = self.prop
        L, me = pr.layout, pr.me
        my_cg = L.coords(me)[1]
        partner = [q for q in pr.members if q != me][0]
        base = int(pr.pb[pr.rank, p])
        ins1, outs1, ins2, outs2 = [], [], [], []
        for g in range(L.world):
            if g == me:                                   # my own stripe stays where it is
                ins1.append((0, 0, 0, "in"))
                outs1.append((0, 0, 0, "out"))
                ins2.append((0, 0, 0, "in"))
                outs2.append((0, 0, 0, "out"))
                continue
            rg_g, cg_g = L.coords(g)
            # phase 1, to g: stripe g of my piece; from g: stripe `me` of its piece
            a, b = self.stripe(pr.rank, p, g)
            ins1.append((1, a - base, b - base, None) if b > a and pr.widths[my_cg] else (0, 0, 0, "in"))
            a, b = self.stripe(rg_g, p, me)
            if b <= a or not pr.widths[cg_g]:
                outs1.append((0, 0, 0, ("out", g)))
            elif cg_g == my_cg:
                outs1.append((2, a, b, None))
            else:
                outs1.append((3, 0, 0, (g, b - a, pr.widths[cg_g])))
            # phase 2, to g: stripe `me` of the piece of g's partner (my own kept stripe if that partner is me)
            owner = [m for m in L.members(cg_g) if m != g][0]
            a, b = self.stripe(L.coords(owner)[0], p, me)
            if b <= a or not pr.widths[cg_g]:
                ins2.append((0, 0, 0, "in"))
            elif owner == me:
                ins2.append((1, a - base, b - base, None))
            else:
                ins2.append((3, 0, 0, (owner, b - a, pr.widths[cg_g])))
            # from g: stripe g of my partner's piece
            a, b = self.stripe(L.coords(partner)[0], p, g)
            outs2.append((2, a, b, None) if b > a and pr.widths[my_cg] else (0, 0, 0, ("out", g)))
        plan = self._plans[p] = ((ins1, outs1), (ins2, outs2))
        return plan

    def _tensors(self, p, entries, y_piece, x_next):
        out = []
        for kind, a, b, aux in entries:
            if kind == 1:
                out.append(y_piece[a:b])
            elif kind == 2:
                out.append(x_next[a:b])
            elif kind == 3:
                out.append(self.relay_buf(p, aux[0], aux[1], aux[2], x_next))
            else:
                out.append(self.dummy(x_next, aux))
        return out

    def _a2a(self, p, phase, y_piece, x_next):
        ins, outs = self.plan(p)[phase]
        return _Works([dist.all_to_all(self._tensors(p, outs, y_piece, x_next), self._tensors(p, ins, y_piece, x_next),
                                       group=self.prop.group, async_op=True)])


class ShardedPropagator:
    """K-hop propagation of a row-sharded adjacency.

    spmm_pieces: list of callables, one per local row piece: f(x_full [N, d]) -> y [rows_of_piece, d] written into
                 the tensor passed as `out`;  signature f(x_full, out).
    all_piece_bounds: int64 array [world, pieces+1] of absolute row boundaries of every rank's pieces
                 (identical on all ranks)."""

    def __init__(self, spmm_pieces, all_piece_bounds, rank, world, n_rows, group=None, transport="p2p",
                 layout=None, me=None, widths=None):
        """rank / world: this rank's row block index and the number of row blocks (= the size of its column group).
        layout / me / widths: only for grid jobs -- the GridLayout, this rank's GLOBAL rank and the (padded) slice
        width of every column group; without a layout the job is row-sharded over the whole process group and
        global rank == row block index.  `group` is the process group the transports run on (the whole job's; the
        "allgather" transport needs it to contain exactly the column group).
        transport: "p2p" | "allgather" | "staged" | "relay" | "relay_staged" (assignable at any time)."""
        self.piece_streams = True             # propagate(): alternate the row pieces between two streams (GPU only)
        self.relay_collective = None          # relay phases as one all_to_all each: None = when the backend is RCCL
        self.spmm_pieces = spmm_pieces
        self.pb = np.asarray(all_piece_bounds, dtype=np.int64)
        self.rank, self.world, self.n = rank, world, int(n_rows)
        self.pieces = self.pb.shape[1] - 1
        self.group = group
        assert self.pb.shape[0] == world and len(spmm_pieces) == self.pieces
        self.lo, self.hi = int(self.pb[rank, 0]), int(self.pb[rank, -1])
        self.layout = layout
        if layout is None:
            self.me, self.members = rank, list(range(world))
        else:
            assert layout.row_groups == world and me is not None and layout.coords(me)[0] == rank
            self.me, self.members = int(me), layout.members(layout.coords(me)[1])
        self.widths = widths
        self._transports = {}
        self.transport = transport

    @property
    def transport(self):
        return self._transport_name

    @transport.setter
    def transport(self, name):
        if name not in self._transports:
            if name in ("p2p", "staged"):
                tr = _DirectTransport(self, staged=name == "staged")
            elif name == "allgather":
                tr = _AllGatherTransport(self)
            elif name in ("relay", "relay_staged"):
                if self.layout is None:
                    raise ValueError("the relay transport needs a GridLayout")
                tr = _RelayTransport(self, staged=name == "relay_staged")
            else:
                raise ValueError(f"unknown transport {name!r}")
            self._transports[name] = tr       # kept: staging / relay buffers and plans survive a switch back
        self._transport_name, self._transport = name, self._transports[name]

    def _exchange_piece(self, p, y_piece, x_next):
        """start moving my piece p to the ranks of my column group (and theirs to me).  Returns an object with
        advance() (cheap; call it after more compute has been queued) and wait()."""
        return self._transport.begin(p, y_piece, x_next)

    def _exchanging(self):
        """does a hop need an exchange?"""
        return self._transport.exchanging()

    def _aux_stream(self, device):
        if getattr(self, "_aux", None) is None:
            self._aux = torch.cuda.Stream(device=device)
        return self._aux

    def propagate(self, x_full, prop_steps, x_buffers=None, y_buffers=None):
        """x_full: [N, d] replica of the input features on this rank's device (row-major, contiguous).
        Returns the list of K+1 LOCAL hop shards [hi-lo, d] (hop 0 is a view of x_full).
        y_buffers: optional K preallocated [hi-lo, d] outputs (a loop that calls this repeatedly then allocates
        nothing: with asynchronous transfers holding references, a host running ahead of the GPU would otherwise keep
        the allocator from recycling the previous calls' outputs)."""
        n, d = x_full.shape
        assert n == self.n
        hops = [x_full[self.lo:self.hi]]
        if prop_steps == 0:
            return hops
        if x_buffers is None:
            x_buffers = [torch.empty_like(x_full) for _ in range(min(2, max(prop_steps - 1, 0)))]
        cur = x_full
        # Consecutive row pieces alternate between the caller's stream and an auxiliary one: the tail of one piece's
        # launch (CUs draining) overlaps the head of the next, so cutting a hop into pieces costs no compute time
        # (measured: 8 pieces 1.50 -> 1.23 ms = the single-launch time, profiles/r01_layout_shares.log).  Each piece's
        # exchange is issued from the stream its SpMM ran on.
        two = x_full.is_cuda and self.pieces > 1 and self.piece_streams
        if two:
            main = torch.cuda.current_stream(x_full.device)
            aux = self._aux_stream(x_full.device)
        for h in range(1, prop_steps + 1):
            last = h == prop_steps
            y_local = y_buffers[h - 1] if y_buffers is not None else \
                torch.empty((self.hi - self.lo, d), dtype=x_full.dtype, device=x_full.device)
            x_next = None if last else x_buffers[(h - 1) % len(x_buffers)]
            if x_next is not None and x_next.numel() and x_next.data_ptr() == cur.data_ptr():
                raise RuntimeError("need two distinct full-size buffers to ping-pong between hops")
            if two:
                # no record_stream on y_local: the caller's stream waits for `aux` at the end of this hop, before
                # anything that could recycle the block, so stream order already protects it
                aux.wait_stream(main)             # the previous hop (and its exchange) is complete for both streams
            works = []
            for p in range(self.pieces):
                r0, r1 = int(self.pb[self.rank, p]) - self.lo, int(self.pb[self.rank, p + 1]) - self.lo
                y_piece = y_local[r0:r1]
                with torch.cuda.stream(aux if p % 2 else main) if two else contextlib.nullcontext():
                    if r1 > r0:
                        self.spmm_pieces[p](cur, y_piece)
                    for w in works:               # two-phase transports: earlier pieces move on while this one computed
                        w.advance()
                    if not last and self._exchanging():
                        works.append(self._exchange_piece(p, y_piece, x_next))
            if two:
                main.wait_stream(aux)
            if not last:
                x_next[self.lo:self.hi].copy_(y_local)
                for w in works:
                    w.advance()
                for w in works:
                    w.wait()
                cur = x_next
            hops.append(y_local)
        return hops

    # ---- diagnostics: the two halves of a hop in isolation (bench.py reports them next to the job time) ----------
    def spmm_only(self, x_chunks):
        """this rank's SpMM over every column chunk, no exchange -> list of local results"""
        outs = []
        for x in x_chunks:
            y_local = torch.empty((self.hi - self.lo, x.shape[1]), dtype=x.dtype, device=x.device)
            for p in range(self.pieces):
                r0, r1 = int(self.pb[self.rank, p]) - self.lo, int(self.pb[self.rank, p + 1]) - self.lo
                if r1 > r0:
                    self.spmm_pieces[p](x, y_local[r0:r1])
            outs.append(y_local)
        return outs

    def exchange_only(self, y_chunks, x_next_chunks):
        """one hop's all-gather of already computed local rows (no SpMM); blocks the stream until it has landed"""
        for y_local, x_next in zip(y_chunks, x_next_chunks):
            works = []
            for p in range(self.pieces):
                r0, r1 = int(self.pb[self.rank, p]) - self.lo, int(self.pb[self.rank, p + 1]) - self.lo
                if self._exchanging():
                    works.append(self._exchange_piece(p, y_local[r0:r1], x_next))
            x_next[self.lo:self.hi].copy_(y_local)
            for w in works:
                w.advance()
            for w in works:
                w.wait()

    # ---- fused push: the SpMM kernel itself writes each finished row into every peer's replica --------------------
    def enable_push(self, chunk_widths, handles, device, group=None):
        """Allocate the ping-pong feature replicas (two per column chunk) and map every peer's replicas into this
        process through CUDA/HIP IPC (torch.multiprocessing's tensor sharing: hipIpcGetMemHandle / OpenMemHandle with
        lazy peer access -- needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver).  Collective.
        handles: the DeviceCSR objects of this rank's row pieces (device_piece_spmms)."""
        from torch.multiprocessing.reductions import reduce_tensor
        grp = group or self.group
        self._push_handles = handles
        self._push_group = grp
        self._push_local, self._push_ptrs, self._push_keep = [], [], []
        for w in chunk_widths:
            self._push_local.append([torch.empty((self.n, int(w)), dtype=torch.float32, device=device) for _ in range(2)])
        if self.world == 1:
            self._push_ptrs = [[[t.data_ptr()] for t in slots] for slots in self._push_local]
            self._push_masks, self.push_error = None, None
            return self
        # export (local, may fail) -> exchange (collective, every rank takes part even after a local failure, so
        # nobody is left waiting) -> import (local, may fail).  The caller agrees on the outcome with agree().
        self.push_error = None
        try:
            mine = [[reduce_tensor(t) for t in slots] for slots in self._push_local]
        except Exception as e:  # noqa: BLE001
            mine, self.push_error = None, e
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=grp)
        try:
            if self.push_error is None and any(m is None for m in everyone):
                raise RuntimeError("a peer could not export its replicas")
            for c, slots in enumerate(self._push_local):
                per_slot = []
                for b, t in enumerate(slots):
                    ptrs = []
                    for q in range(self.world):
                        if q == self.rank:
                            ptrs.append(t.data_ptr())
                        else:
                            fn, args = everyone[q][c][b]
                            peer = fn(*args)                 # a tensor aliasing rank q's replica (IPC mapping)
                            if tuple(peer.shape) != tuple(t.shape):
                                raise RuntimeError("peer replica has an unexpected shape")
                            self._push_keep.append(peer)
                            ptrs.append(peer.data_ptr())
                    per_slot.append(ptrs)
                self._push_ptrs.append(per_slot)
        except Exception as e:  # noqa: BLE001
            self.push_error = self.push_error or e
        # which of MY rows does each peer actually gather?  rank q needs row i iff column i occurs in its shard, so the
        # kernel skips the peer stores nobody would read (all-gather of one byte per node, once per graph)
        self._push_masks = None
        try:
            needed = torch.zeros(self.n, dtype=torch.uint8, device=device)
            for hd in handles:
                if hd.col.numel():
                    needed[hd.col.long()] = 1
            gathered = [torch.empty_like(needed) for _ in range(self.world)]
            dist.all_gather(gathered, needed, group=grp)
            order = [q for q in range(self.world) if q != self.rank]
            masks = []
            for p in range(self.pieces):
                r0, r1 = int(self.pb[self.rank, p]), int(self.pb[self.rank, p + 1])
                m = torch.zeros(r1 - r0, dtype=torch.uint8, device=device)
                for k, q in enumerate(order):
                    m |= (gathered[q][r0:r1] << k)
                masks.append(m.contiguous())
            self._push_masks = masks
            self.push_skipped_fraction = 1.0 - float(sum(int(torch.count_nonzero(gathered[q][self.lo:self.hi])) for q in order)) / \
                max(1, (self.hi - self.lo) * len(order))
        except Exception as e:  # noqa: BLE001  (the masks are an optimisation: without them every row goes everywhere)
            self.push_error = self.push_error or e
        return self

    def agree(self, ok, device):
        """True iff `ok` holds on EVERY rank (all-reduce MIN): keeps the ranks' control flow identical"""
        if self.world == 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item())

    def _push_barrier(self, device):
        """hop boundary of the push transport: my kernels (and with them my posted peer stores) have completed, then
        every rank has said so -- after that all replicas hold the complete hop"""
        torch.cuda.synchronize(device)
        if self.world > 1:
            dist.barrier(group=self._push_group)

    def propagate_push(self, x_chunks, prop_steps):
        """Same result as propagate_chunked, different transport: no send/recv at all.  Every rank's SpMM kernel
        stores its output rows into ALL ranks' next-hop replicas (sgl_spmm_multi_f32: local store + up to 7 posted
        peer stores per row over xGMI); a device synchronise + process-group barrier closes the hop.  Compute and
        communication are the same instruction stream, so they overlap perfectly and no CU runs a copy kernel.
        Needs enable_push().  Returns hops[h][c] = local shard [hi-lo, w_c] (copies: the replicas are recycled)."""
        C = len(x_chunks)
        assert hasattr(self, "_push_ptrs") and len(self._push_ptrs) == C
        device = x_chunks[0].device
        hops = [[x[self.lo:self.hi] for x in x_chunks]]
        cur = list(x_chunks)
        order = [self.rank] + [q for q in range(self.world) if q != self.rank]          # local replica first
        for h in range(1, prop_steps + 1):
            last = h == prop_steps
            slot = (h - 1) % 2
            outs = []
            for c in range(C):
                w_c = x_chunks[c].shape[1]
                if last:
                    y_local = torch.empty((self.hi - self.lo, w_c), dtype=torch.float32, device=device)
                    for p in range(self.pieces):
                        r0, r1 = int(self.pb[self.rank, p]) - self.lo, int(self.pb[self.rank, p + 1]) - self.lo
                        if r1 > r0:
                            self.spmm_pieces[p](cur[c], y_local[r0:r1])
                    outs.append(y_local)
                    continue
                ptrs = self._push_ptrs[c][slot]
                for p in range(self.pieces):
                    r0, r1 = int(self.pb[self.rank, p]), int(self.pb[self.rank, p + 1])
                    if r1 > r0:
                        mask = self._push_masks[p] if getattr(self, "_push_masks", None) else None
                        self._push_handles[p].spmm_multi(cur[c], [ptrs[q] + r0 * w_c * 4 for q in order], w_c, row_mask=mask)
                outs.append(None)
            if not last:
                self._push_barrier(device)
                for c in range(C):
                    t = self._push_local[c][slot]
                    outs[c] = t[self.lo:self.hi].clone()
                    cur[c] = t
            hops.append(outs)
        return hops

    def propagate_chunked(self, x_chunks, prop_steps, buffers=None, y_buffers=None):
        """Software-pipelined variant: the feature block is held as C column chunks (separate contiguous [N, w_c]
        matrices, see column_chunks()).  SpMM is separable over columns, so while chunk c's new rows are in flight
        to the peers, chunk c+1 is being multiplied, and hop h+1 of chunk c only waits for chunk c's own exchange:

            compute  A1 B1 A2 B2 A3 B3
            exchange    A1 B1 A2 B2            (A_h = chunk A of hop h; the last hop needs no exchange)

        The dependency stall of the plain scheme (next hop cannot start before the whole all-gather landed)
        disappears; in the communication-bound regime the hop time is the transfer time.
        x_chunks: list of C replicas [N, w_c]; returns hops[h][c] = LOCAL shard [hi-lo, w_c].
        y_buffers[c][h-1]: optional preallocated outputs (see propagate)."""
        C = len(x_chunks)
        n = x_chunks[0].shape[0]
        assert n == self.n and self.pieces >= 1
        hops = [[x[self.lo:self.hi] for x in x_chunks]]
        if prop_steps == 0:
            return hops
        if buffers is None:
            buffers = [[torch.empty_like(x) for _ in range(min(2, max(prop_steps - 1, 0)))] for x in x_chunks]
        cur = list(x_chunks)
        pending = [[] for _ in range(C)]          # outstanding transfers that fill cur[c]
        for h in range(1, prop_steps + 1):
            last = h == prop_steps
            outs = []
            for c in range(C):
                for w in pending[c]:              # chunk c of the previous hop must have fully arrived
                    w.wait()
                pending[c] = []
                w_c = x_chunks[c].shape[1]
                y_local = y_buffers[c][h - 1] if y_buffers is not None else \
                    torch.empty((self.hi - self.lo, w_c), dtype=x_chunks[c].dtype, device=x_chunks[c].device)
                x_next = None if last else buffers[c][(h - 1) % len(buffers[c])]
                if x_next is not None and x_next.numel() and x_next.data_ptr() == cur[c].data_ptr():
                    raise RuntimeError("need two distinct buffers per chunk to ping-pong between hops")
                for p in range(self.pieces):
                    r0, r1 = int(self.pb[self.rank, p]) - self.lo, int(self.pb[self.rank, p + 1]) - self.lo
                    y_piece = y_local[r0:r1]
                    if r1 > r0:
                        self.spmm_pieces[p](cur[c], y_piece)
                    if not last and self._exchanging():
                        pending[c].append(self._exchange_piece(p, y_piece, x_next))
                if not last:
                    x_next[self.lo:self.hi].copy_(y_local)
                    cur[c] = x_next
                outs.append(y_local)
            hops.append(outs)
        return hops



class ShardedGraphOp:
    """GraphOp.propagate for one rank of a multi-GPU job (BASELINE configs 4/5: NAFS / PaSca sweeps over the GPUs of
    a node).

    Every rank passes the SAME full adjacency (scipy CSR or sgl_amd.io.DeviceAdjacency; it is normalised on the
    rank's own GPU -- identical kernels on identical inputs, so all ranks hold bit-identical A_hat) and the same full
    feature matrix, and gets back the K+1 hop matrices restricted to ITS block: rows `[self.lo, self.hi)` x columns
    `[self.c0, self.c1)`.  `row_groups` picks the layout (see GridLayout): None = row-sharded over all ranks (the
    block is full-width; MessageOps are row-wise, so they apply to the local shards unchanged, e.g.
    OverSmoothDistanceWeightedOp for NAFS), 1 = feature-sharded (all rows, d/G columns, no communication at all;
    column-wise aggregators -- last/sum/mean/max/min/simple_weighted -- apply unchanged), anything between = grid.
    288 GB per GPU make the replication affordable up to ogbn-papers100M (27 GB of CSR, two 57 GB feature replicas).

    Works without torch.distributed (world size 1); with it, uses the default process group unless `group` is given
    (row-sharded layout only: grid layouts address ranks of the default group)."""

    def __init__(self, prop_steps, r=0.5, alpha=None, pieces=2, col_chunks=2, strict_order=False, group=None,
                 device=None, row_groups=None, transport=None):
        self.prop_steps, self.r, self.alpha = prop_steps, r, alpha
        self.pieces, self.col_chunks, self.strict_order, self.group = pieces, col_chunks, strict_order, group
        self.device = device
        self.row_groups, self.transport = row_groups, transport
        self.lo = self.hi = self.c0 = self.c1 = None
        self._cache = None
        self._props = {}

    def _ranks(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def _gloo(self):
        return dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "gloo"

    def propagate(self, adj, feature):
        from . import device as dev
        from .io import DeviceAdjacency
        rank, world = self._ranks()
        device = torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        if not isinstance(adj, DeviceAdjacency):
            adj = DeviceAdjacency.from_scipy(adj, device=device)
        n = adj.shape[0]
        row_groups = world if self.row_groups is None else int(self.row_groups)
        layout = GridLayout(world, row_groups)
        rg, cg = layout.coords(rank)
        key = (id(adj), adj.col.data_ptr(), adj.nnz, world, rank, row_groups)
        if self._cache is None or self._cache[0] != key:
            rowptr, col, val = dev.normalize_adj(adj.rowptr, adj.col, adj.val, n, self.r, self.alpha)
            rp_host = rowptr.cpu().numpy()
            pb = all_piece_bounds(rp_host, row_groups, self.pieces)
            fns, handles = device_piece_spmms(rowptr, col, val, n, pb[rg], rowptr_host=rp_host, strict=self.strict_order)
            self._cache = (key, fns, pb, handles)
            self._props = {}
        _, fns, pb, handles = self._cache
        x = feature if torch.is_tensor(feature) else torch.from_numpy(np.ascontiguousarray(feature, dtype=np.float32))
        x = x.to(device=device, dtype=torch.float32)
        if x.shape[0] != n:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        d = x.shape[1]
        slices = column_slices(d, layout.col_groups)
        self.c0, self.c1 = slices[cg]
        if layout.col_groups == 1:
            transport = self.transport or ("staged" if world > 1 and self._gloo() else "p2p")
            prop = self._props.get(("rows", transport))
            if prop is None:
                prop = self._props[("rows", transport)] = ShardedPropagator(fns, pb, rank, world, n, group=self.group,
                                                                            transport=transport)
            self._prop = prop
            self.lo, self.hi = prop.lo, prop.hi
            x = x.contiguous()
            chunks = column_chunks(d, self.col_chunks if world > 1 else 1)
            if len(chunks) == 1:
                return prop.propagate(x, self.prop_steps)
            hops = prop.propagate_chunked([x[:, a:b].contiguous() for a, b in chunks], self.prop_steps)
            return [torch.cat(h, dim=1) for h in hops]
        # grid / feature-sharded: the slice is stored zero-padded to a line-friendly pitch and multiplied at that width
        # (the pad columns stay zero and cost no extra cache lines)
        pitch = [dev.row_pitch(b - a, growth=2.0) if b > a else 0 for a, b in slices]
        transport = self.transport or (("relay_staged" if self._gloo() else "relay") if row_groups > 1 else "p2p")
        prop = self._props.get((d, transport))               # keeps its relay buffers / streams across calls
        if prop is None:
            prop = self._props[(d, transport)] = ShardedPropagator(fns, pb, rg, row_groups, n, group=self.group,
                                                                   transport=transport, layout=layout, me=rank, widths=pitch)
        self._prop = prop
        self.lo, self.hi = prop.lo, prop.hi
        w = self.c1 - self.c0
        xs = torch.zeros((n, pitch[cg]), dtype=torch.float32, device=device)
        xs[:, :w] = x[:, self.c0:self.c1]
        hops = prop.propagate(xs, self.prop_steps)
        return [h[:, :w] for h in hops]

    def gather_full(self, local):
        """assemble the full [N, d] matrix on every rank from the ranks' blocks (e.g. the final aggregated features).
        `local` is this rank's block: rows [lo, hi), any width (the same inside a column group); the column groups'
        blocks are laid side by side in column-group order."""
        rank, world = self._ranks()
        if world == 1:
            return local
        prop = self._prop
        layout = prop.layout or GridLayout(world, world)
        staged = local.is_cuda and self._gloo()
        send = (local.detach().cpu() if staged else local).contiguous()
        widths = [None] * world
        dist.all_gather_object(widths, int(local.shape[1]), group=self.group)
        col_w = [int(widths[layout.members(cg)[0]]) for cg in range(layout.col_groups)]
        col_off = np.concatenate([[0], np.cumsum(col_w)])
        total = int(col_off[-1])
        full = torch.zeros((prop.n, total), dtype=local.dtype, device=send.device)
        ops, landings = [], []
        for k in range(1, world):
            dst, src = (rank + k) % world, (rank - k) % world
            if send.numel():
                ops.append(dist.P2POp(dist.isend, send, dst, group=self.group))
            rg_s, cg_s = layout.coords(src)
            r0, r1 = int(prop.pb[rg_s, 0]), int(prop.pb[rg_s, -1])
            if (r1 - r0) * col_w[cg_s]:
                buf = torch.empty((r1 - r0, col_w[cg_s]), dtype=local.dtype, device=send.device)
                ops.append(dist.P2POp(dist.irecv, buf, src, group=self.group))
                landings.append((r0, r1, int(col_off[cg_s]), buf))
        works = dist.batch_isend_irecv(ops) if ops else []
        rg, cg = layout.coords(rank)
        full[prop.lo:prop.hi, int(col_off[cg]):int(col_off[cg]) + local.shape[1]].copy_(send)
        for w in works:
            w.wait()
        for r0, r1, c, buf in landings:
            full[r0:r1, c:c + buf.shape[1]].copy_(buf)
        return full.to(local.device) if staged else full

    def gather_rows(self, local):
        """all-gather a local [hi-lo, d] shard into the full [N, d] matrix (row-sharded layout)"""
        return self.gather_full(local)

    def over_smooth_aggregate(self, hops):
        """OverSmoothDistanceWeightedOp (NAFS, message_op/over_smooth_distance_op.py:6-33) on this rank's blocks, for any
        layout.  The weights need whole rows (cosine of X_0[n] and X_h[n]); a rank that owns only a column slice
        contributes the partial sums of its columns -- X_0.X_h and |X_h|^2 per row and hop, one all-reduce of
        [N, 2H] floats for the whole job -- and then combines its own columns with the shared weights.  Row-sharded
        ranks own whole rows and use the fused single-pass kernel directly."""
        from . import device as dev
        prop = self._prop
        feats = [h.contiguous() for h in hops]
        if prop.layout is None or prop.layout.col_groups == 1:
            return dev.nafs_aggregate(feats)
        H = len(feats)
        part = torch.zeros((prop.n, 2 * H), dtype=torch.float32, device=feats[0].device)
        blk = part[prop.lo:prop.hi]
        for h, xh in enumerate(feats):
            blk[:, h] = (feats[0] * xh).sum(dim=1)
            blk[:, H + h] = (xh * xh).sum(dim=1)
        if self._gloo() and part.is_cuda:
            host = part.cpu()
            dist.all_reduce(host, group=self.group)
            part.copy_(host)
        else:
            dist.all_reduce(part, group=self.group)
        norms = blk[:, H:].sqrt() + 1e-10                    # the reference adds 1e-10 to each norm (:14, :16)
        w = torch.softmax(blk[:, :H] / norms / norms[:, :1], dim=1).contiguous()
        return dev.hop_wsum2d(feats, w)
