"""ShardedPropagator: the k-hop loop of one rank -- row pieces, exchange, buffer ping-pong, software pipelining."""
import contextlib

import numpy as np
import torch
import torch.distributed as dist

from .transports import _AllGatherTransport, _DirectTransport, _RelayTransport


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

    @staticmethod
    def _hops_in_buffers(flag, buffers, prop_steps):
        """the explicit aliasing request (see propagate_chunked) checked against the buffers it needs"""
        if not flag:
            return False
        if buffers is None or not all(len(b) >= prop_steps - 1 for b in buffers):
            raise ValueError("hops_in_buffers=True needs prop_steps - 1 caller-owned buffers per chunk (one per exchanged hop)")
        return True

    def _aux_stream(self, device):
        if getattr(self, "_aux", None) is None:
            self._aux = torch.cuda.Stream(device=device)
        return self._aux

    def propagate(self, x_full, prop_steps, x_buffers=None, y_buffers=None, in_place=False, hops_in_buffers=False):
        """x_full: [N, d] replica of the input features on this rank's device (row-major, contiguous).
        Returns the list of K+1 LOCAL hop shards [hi-lo, d] (hop 0 is a view of x_full).
        in_place: hops 1..K-1 are written straight into this rank's rows of the next replica (no separate shard, no
        copy) -- their entries in the returned list are views that the hop after next overwrites, i.e. only the LAST hop
        is retained.  For jobs that need just A^K X, or whose K+1 full-size shards would not fit (papers100M on few GPUs).
        y_buffers: optional K preallocated [hi-lo, d] outputs (a loop that calls this repeatedly then allocates
        nothing: with asynchronous transfers holding references, a host running ahead of the GPU would otherwise keep
        the allocator from recycling the previous calls' outputs)."""
        n, d = x_full.shape
        assert n == self.n
        hops = [x_full[self.lo:self.hi]]
        if prop_steps == 0:
            return hops
        keep_in_replicas = self._hops_in_buffers(hops_in_buffers, None if x_buffers is None else [x_buffers], prop_steps)
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
            x_next = None if last else x_buffers[(h - 1) % len(x_buffers)]
            if x_next is not None and x_next.numel() and x_next.data_ptr() == cur.data_ptr():
                raise RuntimeError("need two distinct full-size buffers to ping-pong between hops")
            direct = (in_place or keep_in_replicas) and not last
            if direct:
                y_local = x_next[self.lo:self.hi]
            else:
                y_local = y_buffers[h - 1] if y_buffers is not None else \
                    torch.empty((self.hi - self.lo, d), dtype=x_full.dtype, device=x_full.device)
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
                if not direct:
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

    def busiest_link_rows(self):
        """rows the busiest link of this rank carries per hop in one direction: the largest peer block (full all-gather)"""
        sizes = [int(self.pb[q, -1] - self.pb[q, 0]) for q in range(self.world)]
        if self.world == 1:
            return 0
        # in-bound: the largest peer block; out-bound: my own block to every peer
        return max(max(s for q, s in enumerate(sizes) if q != self.rank), sizes[self.rank])

    def exchange_only(self, y_chunks, x_next_chunks, keys=None):
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

    def propagate_chunked(self, x_chunks, prop_steps, buffers=None, y_buffers=None, in_place=False, hops_in_buffers=False):
        """Software-pipelined variant: the feature block is held as C column chunks (separate contiguous [N, w_c]
        matrices, see column_chunks()).  SpMM is separable over columns, so while chunk c's new rows are in flight
        to the peers, chunk c+1 is being multiplied, and hop h+1 of chunk c only waits for chunk c's own exchange:

            compute  A1 B1 A2 B2 A3 B3
            exchange    A1 B1 A2 B2            (A_h = chunk A of hop h; the last hop needs no exchange)

        The dependency stall of the plain scheme (next hop cannot start before the whole all-gather landed)
        disappears; in the communication-bound regime the hop time is the transfer time.
        x_chunks: list of C replicas [N, w_c]; returns hops[h][c] = LOCAL shard [hi-lo, w_c].
        y_buffers[c][h-1]: optional preallocated outputs (see propagate).
        in_place: as in propagate() -- hops 1..K-1 are produced directly in this rank's rows of the next replica (views that
        the hop after next overwrites); only the last hop is retained.
        hops_in_buffers: OPT-IN aliasing for callers that own one buffer per exchanged hop (prop_steps - 1 or more per chunk):
        nothing is overwritten inside a step, so hops 1..K-1 are returned as VIEWS of this rank's rows of those buffers and
        nothing is copied -- they stay valid only until the caller reuses the buffers (e.g. calls this again with them).
        Without the flag the returned hops are separate matrices whatever buffers are passed."""
        C = len(x_chunks)
        n = x_chunks[0].shape[0]
        assert n == self.n and self.pieces >= 1
        hops = [[x[self.lo:self.hi] for x in x_chunks]]
        if prop_steps == 0:
            return hops
        # (the caller's replicas, one per exchanged hop: the hop shards can simply be this rank's rows of them -- no copy)
        keep_in_replicas = self._hops_in_buffers(hops_in_buffers, buffers, prop_steps)
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
                x_next = None if last else buffers[c][(h - 1) % len(buffers[c])]
                if x_next is not None and x_next.numel() and x_next.data_ptr() == cur[c].data_ptr():
                    raise RuntimeError("need two distinct buffers per chunk to ping-pong between hops")
                direct = (in_place or keep_in_replicas) and not last
                if direct:
                    y_local = x_next[self.lo:self.hi]
                elif y_buffers is not None and y_buffers[c][h - 1] is not None:
                    y_local = y_buffers[c][h - 1]
                else:
                    y_local = torch.empty((self.hi - self.lo, w_c), dtype=x_chunks[c].dtype, device=x_chunks[c].device)
                for p in range(self.pieces):
                    r0, r1 = int(self.pb[self.rank, p]) - self.lo, int(self.pb[self.rank, p + 1]) - self.lo
                    y_piece = y_local[r0:r1]
                    if r1 > r0:
                        self.spmm_pieces[p](cur[c], y_piece)
                    if not last and self._exchanging():
                        pending[c].append(self._exchange_piece(p, y_piece, x_next))
                if not last:
                    if not direct:
                        x_next[self.lo:self.hi].copy_(y_local)
                    cur[c] = x_next
                outs.append(y_local)
            hops.append(outs)
        return hops
