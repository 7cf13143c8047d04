"""How the rows a rank has just computed reach the ranks that gather them in the next hop (see the package docstring)."""
import torch
import torch.distributed as dist


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
