"""Need-aware exchange for the row-sharded layout: a rank receives only the rows its block actually gathers.

The plain all-gather moves every row to every rank: (G-1)/G * N * d * 4 bytes in-bound per rank per hop.  But rank g reads
row i of the feature block only if column i occurs in ITS rows of A_hat, and on the benchmark degree laws a rank's block
leaves a sixth (ogbn-products shape, 8 ranks) to a quarter (ogbn-papers100M shape) of the nodes unreferenced.  The set is
a property of the graph, so it is found ONCE:

    HaloPlan          per peer q: `need[q]`, the sorted global ids of q's rows this rank gathers, and -- after one exchange of
                      those lists -- `send_rows[q]`, the rows of this rank's block that q gathers.
    compact table     the rank's gather source is no longer an [N, d] replica but [own rows | ghosts of peer 0 | peer 1 | ...]
                      ([n_own + n_ghost, d]); the block's column ids are relabelled once to positions in that table.  The order
                      of a row's terms is untouched, so every hop is bit-identical to the full-replica run.
    per hop           SpMM (compact columns) -> own rows of the next table; ONE row-copy kernel packs the rows the peers need
                      into a send buffer, peer after peer (sgl_gather_rows_f32 in peer order or sgl_scatter_rows_f32 in own-row
                      order -- every row read once --, whichever is faster for the shape); a grouped send / recv delivers every peer's
                      share straight into its ghost range of the next table.  Nothing is unpacked: ghosts are stored packed.

A rank therefore never holds the whole feature matrix (57 GB at papers100M size): it starts from its OWN feature rows and
fetches the ghosts with the same exchange.  The reference has no counterpart (its NCCL use is DDP training only,
sgl/tasks/node_classification_dist.py:61,70); SURVEY.md section 8(e) is the contract."""
import numpy as np
import torch
import torch.distributed as dist

from .sharded_adj import _world
from .transports import _post


_CHUNK = 1 << 26     # elements per pass of the marking / relabelling loops (bounds the int64 temporaries; tests shrink it)


def _global_rank(group, q):
    return dist.get_global_rank(group, q) if (group is not None and dist.is_initialized()) else q


def _is_staged(group, t):
    return bool(t.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo")


class HaloPlan:
    """Who needs which rows.  Collective over `group` (every rank of the row-sharded job calls it with its own block).

    lo, hi   : this rank's row range;  n: global number of rows (= columns)
    col      : int32 GLOBAL column ids of the rank's block (any device)
    bounds   : [world + 1] row boundaries of all ranks (identical everywhere)"""

    def __init__(self, lo, hi, n, col, bounds, group=None):
        rank, world = _world(group)
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi, self.n = int(lo), int(hi), int(n)
        self.bounds = np.asarray(bounds, dtype=np.int64)
        assert len(self.bounds) == world + 1 and self.bounds[rank] == lo and self.bounds[rank + 1] == hi
        dev = col.device
        self.n_own = self.hi - self.lo
        # 1. which columns does my block touch?  one byte per node, marked in bounded chunks
        mark = self._marks(self.n, col)
        self.need = []
        for q in range(world):
            a, b = int(self.bounds[q]), int(self.bounds[q + 1])
            if q == rank or b <= a:
                self.need.append(torch.empty(0, dtype=torch.int64, device=dev))
            else:
                self.need.append(torch.nonzero(mark[a:b]).flatten() + a)         # sorted global ids
        del mark
        # 2. tell every owner which of its rows I gather (counts by object all-gather, lists point to point)
        my_counts = [int(t.numel()) for t in self.need]
        if world > 1:
            table = [None] * world
            dist.all_gather_object(table, my_counts, group=group)
        else:
            table = [my_counts]
        self.counts = np.asarray(table, dtype=np.int64)                           # counts[r, q]: rows rank r needs from q
        self.send_rows = [torch.empty(int(self.counts[q, rank]), dtype=torch.int32, device=dev) for q in range(world)]
        if world > 1:
            staged = _is_staged(group, col)
            outgoing = [(self.need[q] - int(self.bounds[q])).to(torch.int32) for q in range(world)]
            sends, recvs = [], []
            for k in range(1, world):
                dst, src = (rank + k) % world, (rank - k) % world
                sends.append((outgoing[dst], _global_rank(group, dst)))
                recvs.append((self.send_rows[src], _global_rank(group, src)))
            w = _post(group, sends, recvs, staged)
            w.wait()
            if col.is_cuda:
                torch.cuda.current_stream(dev).synchronize()
        self._finish(dev)

    @staticmethod
    def _marks(n, col):
        mark = torch.zeros(n, dtype=torch.bool, device=col.device)
        for part in col.split(_CHUNK):
            mark[part.long()] = True
        return mark

    @classmethod
    def offline(cls, rank, bounds, n, block_cols):
        """The plan of `rank` computed WITHOUT a process group, from the column ids of every rank's block (block_cols(q) ->
        int32 tensor; called once per rank, so blocks can be generated one at a time).  For cost models and tests on one
        device: identical to what the collective constructor produces on rank `rank`."""
        self = cls.__new__(cls)
        world = len(bounds) - 1
        self.rank, self.world, self.group = int(rank), world, None
        self.bounds = np.asarray(bounds, dtype=np.int64)
        self.lo, self.hi, self.n = int(self.bounds[rank]), int(self.bounds[rank + 1]), int(n)
        self.n_own = self.hi - self.lo
        self.need, self.send_rows = [None] * world, [None] * world
        counts = np.zeros((world, world), dtype=np.int64)
        dev = None
        for q in range(world):
            col = block_cols(q)
            dev = col.device
            mark = cls._marks(self.n, col)
            del col
            if q == rank:
                for p in range(world):
                    a, b = int(self.bounds[p]), int(self.bounds[p + 1])
                    self.need[p] = torch.empty(0, dtype=torch.int64, device=dev) if (p == rank or b <= a) else \
                        torch.nonzero(mark[a:b]).flatten() + a
                    counts[rank, p] = self.need[p].numel()
                self.send_rows[q] = torch.empty(0, dtype=torch.int32, device=dev)
            else:
                self.send_rows[q] = torch.nonzero(mark[self.lo:self.hi]).flatten().to(torch.int32)
                counts[q, rank] = self.send_rows[q].numel()
            del mark
        self.counts = counts
        self._finish(dev)
        return self

    def _finish(self, dev):
        world, rank = self.world, self.rank
        my_counts = [int(t.numel()) for t in self.need]
        # 3. the compact table: [own rows | ghosts of peer 0 | ghosts of peer 1 | ...]
        off = [self.n_own]
        for q in range(world):
            off.append(off[-1] + my_counts[q])
        self.ghost_off = off                                                      # rows [ghost_off[q], ghost_off[q+1]) hold peer q's
        self.n_ghost = off[-1] - self.n_own
        self.n_compact = off[-1]
        self.global_ids = torch.cat([torch.arange(self.lo, self.hi, dtype=torch.int64, device=dev)] + self.need)
        # 4. what I pack for the peers: one index list, peer after peer
        soff = [0]
        for q in range(world):
            soff.append(soff[-1] + int(self.send_rows[q].numel()))
        self.send_off = soff
        self.send_idx = torch.cat([t.to(torch.int64) for t in self.send_rows]) if world > 1 else \
            torch.empty(0, dtype=torch.int64, device=dev)
        if self.send_idx.numel() and (int(self.send_idx.min()) < 0 or int(self.send_idx.max()) >= self.n_own):
            raise RuntimeError("halo plan: a peer asked for a row outside this rank's block")
        # the same pairs (own row -> send-buffer row) in OWN-ROW order: a row that several peers gather is then read once by the
        # pack kernel and served from cache for the others (each peer's list is sorted, so this is a merge of world - 1 runs)
        self.pack_src, self.pack_dst = torch.sort(self.send_idx, stable=True)

    def relabel(self, col):
        """the block's GLOBAL column ids as positions in the compact table (int32, same order)"""
        dev = col.device
        lut = torch.full((self.n,), -1, dtype=torch.int32, device=dev)
        lut[self.lo:self.hi] = torch.arange(self.n_own, dtype=torch.int32, device=dev)
        for q in range(self.world):
            m = int(self.need[q].numel())
            if m:
                lut[self.need[q]] = torch.arange(self.ghost_off[q], self.ghost_off[q] + m, dtype=torch.int32, device=dev)
        out = torch.empty_like(col)
        step = _CHUNK
        for s in range(0, col.numel(), step):
            out[s:s + step] = lut[col[s:s + step].long()]
        if out.numel() and int(out.min()) < 0:
            raise RuntimeError("halo plan: a column of the block is neither an own row nor a ghost")
        return out

    # ---- figures for the bench line / the cost model -----------------------------------------------------------------
    @property
    def rows_in_full(self):
        """rows a plain all-gather would deliver to this rank per hop"""
        return self.n - self.n_own

    @property
    def skipped_fraction(self):
        return 1.0 - self.n_ghost / max(self.rows_in_full, 1)

    def describe(self):
        return {"own_rows": self.n_own, "ghost_rows": self.n_ghost, "compact_rows": self.n_compact,
                "rows_sent": int(self.send_off[-1]), "exchange_skipped_fraction": round(self.skipped_fraction, 4)}


class HaloPropagator:
    """The k-hop loop of one rank on compact tables.

    spmm(x_compact [n_compact, w], out [n_own, w]) : the local SpMM on relabelled columns
    Tables are [n_compact, w]; rows [0, n_own) are the rank's own, the rest ghosts."""

    def __init__(self, plan, spmm, staged=None):
        self.plan, self.spmm = plan, spmm
        self.staged = staged
        self.lo, self.hi, self.n = plan.lo, plan.hi, plan.n
        self.rank, self.world, self.group = plan.rank, plan.world, plan.group
        self.layout = None                                   # what ShardedGraphOp.gather_full / over_smooth_aggregate look at
        self.pb = np.stack([plan.bounds[:-1], plan.bounds[1:]], axis=1)
        self._send = {}
        self.pack_mode = "auto"                              # "gather" | "scatter" | "auto" (CUDA: measured once per shape)
        self._pack_choice, self.pack_timing_ms = {}, {}
        # The send buffer holds the peers' shares in rank order and so do the ghost ranges of a table: the whole exchange is ONE
        # all_to_all_single with split sizes (an all-to-all-v) -- one call instead of 2 (G - 1) point-to-point operations, which is
        # what bounds how finely a hop can be cut on the host side.  Off by default (gloo has no all_to_all; bench.py times it
        # as a candidate on RCCL and keeps it only if it validates and wins).
        self.collective = False
        self._splits = ([int(plan.send_off[q + 1] - plan.send_off[q]) for q in range(plan.world)],
                        [int(plan.ghost_off[q + 1] - plan.ghost_off[q]) for q in range(plan.world)])

    # ---- building blocks -------------------------------------------------------------------------------------------
    def _pack(self, y_own, key):
        """rows of y_own every peer needs, peer after peer, in one reusable buffer per (chunk) key"""
        pl = self.plan
        rows, w = int(pl.send_off[-1]), y_own.shape[1]
        buf = self._send.get(key)
        if buf is None or buf.shape != (rows, w) or buf.device != y_own.device:
            buf = self._send[key] = torch.empty((rows, w), dtype=y_own.dtype, device=y_own.device)
        if rows == 0:
            return buf
        if y_own.is_cuda:
            src = y_own if y_own.stride(1) == 1 else y_own.contiguous()
            self._pack_device(src, buf, (key, rows, w))
        elif self.pack_mode == "scatter":
            buf.index_copy_(0, pl.pack_dst, y_own.index_select(0, pl.pack_src))
        else:
            torch.index_select(y_own, 0, pl.send_idx, out=buf)
        return buf

    def _pack_device(self, src, buf, key):
        """Two orders of the same copy: peer after peer (gather: sequential writes, every own row re-read once per peer that
        gathers it) or own row after own row (scatter: every row read once, rows written to world - 1 places).  Which is faster
        depends on the row width -- scattered rows must be whole 128-byte lines -- and on whether the send buffer still fits the
        caches (profiles/r03_pack_order.log: 0.098 vs 0.145 ms at 64 columns, 0.108 vs 0.095 ms at 36, S1 on 8 ranks; equal at
        the papers100M size), so with pack_mode "auto" the first pack of a shape times both (identical bytes either way)."""
        from .. import device as dev
        pl = self.plan

        def gather():
            dev.gather_rows(src, pl.send_idx, out=buf)

        def scatter():
            dev.scatter_rows(src, pl.pack_src, pl.pack_dst, buf)

        mode = self.pack_mode if self.pack_mode != "auto" else self._pack_choice.get(key)
        if mode is None:
            best = {}
            for name, fn in (("gather", gather), ("scatter", scatter)):
                fn()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                fn()
                fn()
                ev[1].record()
                ev[1].synchronize()
                best[name] = ev[0].elapsed_time(ev[1])
            mode = self._pack_choice[key] = min(best, key=best.get)
            self.pack_timing_ms[key[1:]] = {k: round(v / 2, 4) for k, v in best.items()}
        (scatter if mode == "scatter" else gather)()

    def begin_exchange(self, y_own, table_next, key=0):
        """pack my rows for the peers and start the grouped send / recv that fills the ghost ranges of table_next.
        Returns an object with wait() (stream-level on RCCL)."""
        pl = self.plan
        if self.world == 1:
            return _post(self.group, [], [])
        buf = self._pack(y_own, key)
        staged = _is_staged(self.group, table_next) if self.staged is None else self.staged
        if self.collective and not staged:
            from .transports import _Works
            work = dist.all_to_all_single(table_next[pl.n_own:], buf, output_split_sizes=self._splits[1],
                                          input_split_sizes=self._splits[0], group=self.group, async_op=True)
            return _Works([work])
        sends, recvs = [], []
        for k in range(1, self.world):                 # staggered peer order: every link busy in both directions
            dst, src = (self.rank + k) % self.world, (self.rank - k) % self.world
            a, b = pl.send_off[dst], pl.send_off[dst + 1]
            if b > a:
                sends.append((buf[a:b], _global_rank(self.group, dst)))
            a, b = pl.ghost_off[src], pl.ghost_off[src + 1]
            if b > a:
                recvs.append((table_next[a:b], _global_rank(self.group, src)))
        return _post(self.group, sends, recvs, staged)

    def new_table(self, w, like):
        return torch.empty((self.plan.n_compact, int(w)), dtype=like.dtype, device=like.device)

    def table_from_own(self, x_own, key="init"):
        """compact table from this rank's OWN feature rows: the ghosts come through the exchange (blocking)"""
        assert x_own.shape[0] == self.plan.n_own
        t = self.new_table(x_own.shape[1], x_own)
        t[:self.plan.n_own].copy_(x_own)
        self.begin_exchange(t[:self.plan.n_own], t, key).wait()
        return t

    def table_from_full(self, x_full):
        """compact table cut out of a full [N, w] matrix this rank happens to hold (no communication)"""
        assert x_full.shape[0] == self.plan.n
        ids = self.plan.global_ids
        if x_full.is_cuda:
            from .. import device as dev
            return dev.gather_rows(x_full if x_full.stride(1) == 1 else x_full.contiguous(), ids,
                                   out=self.new_table(x_full.shape[1], x_full))
        return x_full.index_select(0, ids)

    # ---- the hop loop ----------------------------------------------------------------------------------------------
    def propagate_chunked(self, tables, prop_steps, buffers=None, y_buffers=None, in_place=False, hops_in_buffers=False):
        """tables: list of C compact tables [n_compact, w_c] (the column chunks of hop 0).  Software-pipelined like
        ShardedPropagator.propagate_chunked: while chunk c's rows travel, chunk c+1 is multiplied, and hop h+1 of chunk c
        waits only for chunk c's own exchange.  Returns hops[h][c] = LOCAL shard [n_own, w_c]; with in_place only the last
        hop is retained (earlier entries are views the hop after next overwrites).  `buffers[c]`: the caller's tables for the
        exchanged hops.  hops_in_buffers (opt-in, needs prop_steps - 1 or more tables per chunk): hops 1..K-1 are returned as
        VIEWS of the own rows of those tables -- valid until the caller reuses the tables -- and nothing is copied; without the
        flag the returned hops are separate matrices whatever buffers are passed."""
        C = len(tables)
        n_own = self.plan.n_own
        hops = [[t[:n_own] for t in tables]]
        if prop_steps == 0:
            return hops
        # the CALLER's tables, one per exchanged hop: nothing is overwritten inside a step, so the hop matrices can simply BE the
        # own rows of those tables (no copy into the table, no separate output); with fewer (ping-pong) tables or our own
        # temporaries the hops are separate matrices, as before
        from .propagator import ShardedPropagator
        keep_in_tables = ShardedPropagator._hops_in_buffers(hops_in_buffers, buffers, prop_steps)
        if buffers is None:
            buffers = [[torch.empty_like(t) for _ in range(min(2, max(prop_steps - 1, 0)))] for t in tables]
        cur = list(tables)
        pending = [None] * C
        for h in range(1, prop_steps + 1):
            last = h == prop_steps
            outs = []
            for c in range(C):
                if pending[c] is not None:
                    pending[c].wait()
                    pending[c] = None
                w_c = tables[c].shape[1]
                t_next = None if last else buffers[c][(h - 1) % len(buffers[c])]
                if t_next is not None and t_next.numel() and t_next.data_ptr() == cur[c].data_ptr():
                    raise RuntimeError("need two distinct tables per chunk to ping-pong between hops")
                direct = (in_place or keep_in_tables) and not last
                if direct:
                    y_own = t_next[:n_own]
                elif y_buffers is not None and y_buffers[c][h - 1] is not None:
                    y_own = y_buffers[c][h - 1]
                else:
                    y_own = torch.empty((n_own, w_c), dtype=tables[c].dtype, device=tables[c].device)
                if n_own:
                    self.spmm(cur[c], y_own)
                if not last:
                    if not direct:
                        t_next[:n_own].copy_(y_own)
                    pending[c] = self.begin_exchange(y_own, t_next, key=c)
                    cur[c] = t_next
                outs.append(y_own)
            hops.append(outs)
        return hops

    def propagate(self, table, prop_steps, buffers=None, y_buffers=None, in_place=False, hops_in_buffers=False):
        hops = self.propagate_chunked([table], prop_steps, None if buffers is None else [buffers],
                                      None if y_buffers is None else [y_buffers], in_place, hops_in_buffers)
        return [h[0] for h in hops]

    def _exchanging(self):
        return self.world > 1

    # ---- diagnostics: the halves of a hop in isolation -------------------------------------------------------------
    def spmm_only(self, tables):
        outs = []
        for t in tables:
            y = torch.empty((self.plan.n_own, t.shape[1]), dtype=t.dtype, device=t.device)
            if self.plan.n_own:
                self.spmm(t, y)
            outs.append(y)
        return outs

    def exchange_only(self, ys, tables_next, keys=None):
        """keys: the chunk index of every entry (default 0, 1, ...): a caller that times ONE chunk passes its index so the
        chunk's own send buffer is used"""
        keys = range(len(ys)) if keys is None else keys
        works = [self.begin_exchange(y, t, key=c) for c, y, t in zip(keys, ys, tables_next)]
        for w in works:
            w.wait()

    def pack_only(self, ys, keys=None):
        keys = range(len(ys)) if keys is None else keys
        for c, y in zip(keys, ys):
            self._pack(y, c)

    def busiest_link_rows(self):
        """rows the busiest link of this rank carries per hop in one direction (to or from a single peer)"""
        pl = self.plan
        rows_in = max([int(t.numel()) for t in pl.need] + [0])
        rows_out = max([int(t.numel()) for t in pl.send_rows] + [0])
        return max(rows_in, rows_out)


def block_halo(block, bounds, group=None, strict=False, reorder=None):
    """Plan, propagator and SpMM handle for a NORMALISED RowBlock (rows [lo, hi) of A_hat, global column ids): the columns are
    relabelled to the rank's compact table and the handle multiplies [n_own x n_compact].  Collective.
    reorder: None / "community" / "auto" -- the rank's rows are STORED and PROCESSED in a locality order found on the block's own
    diagonal part (sgl_amd.reorder.local_rowmap; no communication) behind a row map: outputs, ids and every row's terms keep
    their order, the hops are bit-identical (plan.reorder_info says what was decided)."""
    from ..device import DeviceCSR, default_long_row_nnz, permute_rows
    from ..reorder import local_rowmap
    from .sharded_adj import global_nnz
    plan = HaloPlan(block.lo, block.hi, block.n, block.col, bounds, group)
    plan.reorder_info = {"reorder": reorder}
    total = global_nnz(block, group)          # long rows are cut where the WHOLE matrix would cut them (same bits at any world size)
    if block.n_local == 0:
        return plan, HaloPropagator(plan, lambda x, out: None), None
    rowptr, ccol, val = block.rowptr, plan.relabel(block.col), block.val
    rowmap = None
    if reorder and ccol.is_cuda:
        rowmap, plan.reorder_info = local_rowmap(rowptr, ccol, 0, plan.n_own, reorder)   # own nodes = compact columns [0, n_own)
        if rowmap is not None:
            rowptr, ccol, val = permute_rows(rowptr, ccol, val, rowmap)
    handle = DeviceCSR(rowptr, ccol, val, (block.n_local, plan.n_compact), strict=strict, long_row_nnz=default_long_row_nnz(total))
    if rowmap is not None:
        handle.set_rowmap(rowmap)
    return plan, HaloPropagator(plan, lambda x, out: handle.spmm(x, out=out)), handle


def halo_checksums(plan, table, y_own):
    """Exact integrity check of one exchange: for every peer the owner publishes the wrapping int64 sum of the raw bits of the
    rows it packed for that peer, the receiver re-sums its ghost range.  Order-independent, exact.  Collective."""
    def bits_sum(t):
        total = torch.zeros((), dtype=torch.int64, device=t.device)
        flat = t.contiguous().view(torch.int32).view(-1)
        for part in flat.split(1 << 27):
            total += part.to(torch.int64).sum()
        return int(total.item())

    world, rank = plan.world, plan.rank
    if world == 1:
        return True
    mine = []
    for q in range(world):
        rows = plan.send_rows[q]
        mine.append(bits_sum(y_own.index_select(0, rows.to(torch.int64))) if rows.numel() else 0)
    table_all = [None] * world
    dist.all_gather_object(table_all, mine, group=plan.group)
    ok = True
    for q in range(world):
        a, b = plan.ghost_off[q], plan.ghost_off[q + 1]
        got = bits_sum(table[a:b]) if b > a else 0
        ok = ok and got == table_all[q][rank]
    return ok
