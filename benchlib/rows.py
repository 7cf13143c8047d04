"""The contract layout of an N-rank job: A_hat row-sharded in storage + per-hop all-gather (need-aware or of the full replica),
the selection of its exchange and of its pipelining granularity, and the job state the layout builders share."""
import os
import sys
import time

import numpy as np
import torch

from .common import _phase  # noqa: F401


class _Job:
    """What the layout builders and the timing code share: the workload replica of this rank, the ranks' agreement
    helpers and the knobs.  One instance per bench.run()."""

    def __init__(self, args, engine, rank, world, wl):
        self.args, self.engine, self.device = args, engine, engine.device
        self.rank, self.world = rank, world
        self.n, self.d, self.K = wl["n"], wl["d"], wl["k"]
        self.nbuf = min(2, max(self.K - 1, 0))           # ping-pong replicas a multi-hop exchange needs
        self.rowptr = self.col = self.val = self.x0 = self.rp_host = None
        self.block = self.full = self.bounds = None
        self.t_setup = time.perf_counter()
        self.own_group = False
        self.info = {}                                    # -> config.plan of the JSON line
        # columns [a, b) of the feature block as the matrix a layout multiplies (engines may pad it to a line pitch)
        self.engine_pack = getattr(engine, "pack_slice", lambda x, a, b: x[:, a:b].contiguous())

    # ---- agreement between ranks ------------------------------------------------------------------------------------
    def sync_all(self):
        import torch.distributed as dist
        self.engine.sync()
        if self.world > 1:
            dist.barrier()
            self.engine.sync()

    def agree(self, ok):
        """True iff `ok` holds on every rank: keeps the ranks' control flow identical"""
        import torch.distributed as dist
        if self.world == 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def max_over_ranks(self, v):
        import torch.distributed as dist
        if self.world == 1:
            return float(v)
        tt = torch.tensor([v], dtype=torch.float64, device=self.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def timed_s(self, fn, reps=2, warm=1):
        """seconds per call, MAX over ranks, bracketed by barriers"""
        for _ in range(warm):
            fn()
        self.sync_all()
        t_a = time.perf_counter()
        for _ in range(reps):
            fn()
        self.sync_all()
        return self.max_over_ranks((time.perf_counter() - t_a) / reps)

    def ensure_group(self):
        """a process group even for --force-sharded on one GPU (the push transport's setup is collective)"""
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group(self.engine.backend, rank=0, world_size=1, **self.engine.init_kwargs())
            self.own_group = True

    # ---- workload -----------------------------------------------------------------------------------------------
    def load_workload(self, wl):
        """Single rank: the whole A_hat.  Several ranks: ROW-SHARDED STORAGE -- every rank ends up with its own
        nnz-balanced row block of A_hat (self.block) and a replica of the features; the whole matrix exists only on
        the rank that generated the raw graph (rank 0, Chung-Lu workloads) or nowhere at all (hashed workloads)."""
        import torch.distributed as dist
        from sgl_amd.dist import RowBlock, scatter_row_blocks
        n, d, device, engine = self.n, self.d, self.device, self.engine
        self.block = self.full = None
        if self.world == 1 and not self.args.force_sharded:
            rowptr, col, val, x0 = engine.build_workload(self.args, wl)
            self.full = (rowptr, col, val)
            self.rowptr, self.col, self.val, self.x0 = rowptr, col, val, x0
            self.nnz = int(col.numel())
            engine.sync()
            return
        if wl.get("hashed"):
            bounds, self.nnz = engine.hashed_bounds(self.args, wl, self.world)
            self.bounds = bounds
            self.block = engine.hashed_block(self.args, wl, int(bounds[self.rank]), int(bounds[self.rank + 1]))
            self.x0 = engine.features(self.args, wl)                  # generated locally on every rank: no traffic
            self.info["adjacency_storage"] = f"row block per rank, generated in place ({self.block.nnz} of {self.nnz} nnz on rank 0)"
            engine.sync()
            return
        raw = None
        if self.rank == 0:
            raw = engine.build_raw(self.args, wl)
            # A_hat has the rows of A plus one diagonal entry each (Chung-Lu graphs have no self loops)
            rp = raw[0].cpu().numpy() + np.arange(n + 1, dtype=np.int64)
            from sgl_amd.dist import balanced_bounds
            bounds = [int(b) for b in balanced_bounds(rp, self.world)]
        else:
            bounds = None
        if self.world > 1:
            box = [bounds]
            dist.broadcast_object_list(box, 0)
            bounds = box[0]
        self.bounds = np.asarray(bounds, dtype=np.int64)
        raw_block = scatter_row_blocks(raw, self.bounds, n, device) if self.world > 1 else RowBlock(0, n, n, *raw)
        del raw
        self.block = engine.normalize_block(raw_block, 0.5, None, symmetric=True)
        del raw_block
        if self.rank == 0:
            x0 = engine.features(self.args, wl)
        else:
            x0 = torch.empty((n, d), dtype=torch.float32, device=device)
        if self.world > 1:
            dist.broadcast(x0, 0)
        self.x0 = x0
        nnz = torch.tensor([self.block.nnz], dtype=torch.int64, device=device)
        if self.world > 1:
            dist.all_reduce(nnz)
        self.nnz = int(nnz.item())
        self.info["adjacency_storage"] = (f"row block per rank: rank 0 holds {self.block.nnz} of {self.nnz} nnz of A_hat "
                                          f"(normalised per block, degrees by all-reduce)")
        engine.sync()

    def full_adj(self):
        """the whole A_hat on this rank (layouts that multiply all rows, the single-GPU reference chain): gathered from
        the ranks' blocks on first use"""
        from sgl_amd.dist import allgather_blocks
        if self.full is None:
            self.full = allgather_blocks(self.block)
            self.info["adjacency_replicated_for"] = "alternative layout candidates and their single-GPU reference chain"
        self.rowptr, self.col, self.val = self.full
        return self.full

    def drop_full(self):
        self.full = self.rowptr = self.col = self.val = self.rp_host = None

    def piece_spmms(self, bounds):
        rowptr, col, val = self.full_adj()
        if self.rp_host is None:
            self.rp_host = rowptr.cpu().numpy()
        return self.engine.piece_spmms(self.args, rowptr, col, val, self.n, bounds, self.rp_host)

    def budget_left(self):
        return self.args.setup_budget - (time.perf_counter() - self.t_setup)


def _select_exchange(job, full, halo):
    """row-sharded layout: the transport of the per-hop all-gather when the flags leave a choice (--exchange auto / push; the default
    --exchange halo never comes here).  auto = time one hop's exchange with every candidate -- the need-aware packed exchange
    (halo: pack kernel + grouped send/recv of the rows each peer gathers) and the full-replica process-group transports (p2p,
    allgather) -- and keep the fastest (decision on the MAX over ranks, so every rank picks the same).  push = the fused
    transport (the SpMM kernel stores finished rows into the peers' replicas over HIP IPC): it must map all peers AND reproduce
    the process-group result of a whole k-hop step on every rank before it may run; otherwise the job goes on with the
    process-group transport and says why.  No timing decides whether push runs: it was asked for."""
    args, engine, info, K, device = job.args, job.engine, job.info, job.K, job.device
    if full is None:
        return "halo"
    prop, handles, x_chunks, cbufs = full["prop"], full["handles"], full["x_chunks"], full["cbufs"]
    transports = getattr(engine, "transports", ("p2p", "allgather"))

    def setup_push():
        """collective; returns True iff every rank mapped every peer's replicas"""
        job.ensure_group()
        prop.enable_push([xc.shape[1] for xc in x_chunks], handles, device)
        ok = prop.agree(prop.push_error is None, device)
        if ok and getattr(prop, "push_skipped_fraction", None) is not None:
            info["push_peer_rows_skipped"] = round(prop.push_skipped_fraction, 4)
        if not ok and prop.push_error is not None:
            sys.stderr.write(f"[bench] push transport unavailable on rank {job.rank}: {prop.push_error!r}\n")
        return ok

    exchange = args.exchange
    if exchange == "push":
        base = transports[0]
        prop.transport = base
        if not handles or not setup_push():
            info["push_rejected"] = "mapping failed"
            return base
        ref_hops = prop.propagate_chunked(x_chunks, K, buffers=cbufs)
        got_hops = prop.propagate_push(x_chunks, K)
        same = True
        for a_, b_ in zip(ref_hops[K], got_hops[K]):
            scale_ = float(a_.abs().max()) if a_.numel() else 0.0
            same = same and (a_.numel() == 0 or float((a_ - b_).abs().max()) <= 1e-5 * max(scale_, 1e-30))
        if not prop.agree(same, device):
            info["push_rejected"] = "result mismatch"
            return base
        info["push_validated_against"] = base
        return "push"
    if exchange != "auto":
        return exchange
    exchange = transports[0]
    if job.world == 1 or job.nbuf == 0:
        return exchange
    ys0 = [torch.zeros((prop.hi - prop.lo, xc.shape[1]), dtype=xc.dtype, device=device) for xc in x_chunks]
    cand = {}
    for tname in transports:
        prop.transport = tname
        cand[tname] = job.timed_s(lambda: prop.exchange_only(ys0, [b[0] for b in cbufs]))
    if halo is not None:
        hp = halo["prop"]
        first = [b[0] for b in halo["bufs"]]
        cand["halo"] = job.timed_s(lambda: hp.exchange_only(ys0, first))
        if getattr(engine, "halo_collective", False):
            # the same exchange as ONE all_to_all_single with split sizes: a candidate only if it delivers the right rows (exact
            # bit-checksums of every ghost range on random data) on every rank
            from sgl_amd.dist import halo_checksums
            good = True
            probe = [torch.rand_like(y) for y in ys0]
            try:
                hp.collective = True
                hp.exchange_only(probe, first)
                engine.sync()
            except Exception as e:  # noqa: BLE001
                good = False
                sys.stderr.write(f"[bench] all_to_all form of the need-aware exchange unavailable on rank {job.rank}: {e!r}\n")
            if job.agree(good):                               # every rank got through the call: now the (collective) check
                good = all([halo_checksums(halo["plan"], t, y) for t, y in zip(first, probe)])   # a list: every collective runs
                good = job.agree(good)
            else:
                good = False
            if good:
                cand["halo_a2a"] = job.timed_s(lambda: hp.exchange_only(ys0, first))
            else:
                info["halo_a2a_rejected"] = True
            hp.collective = False
    exchange = min(cand, key=cand.get)
    info["exchange_candidates_ms"] = {k: round(v * 1e3, 3) for k, v in cand.items()}
    info["exchange_selected_by"] = "measured exchange-only time (--exchange auto, opt-in)"
    return exchange


def _rows_full_replica(job, chunks):
    """row-sharded layout on full feature replicas: every rank's new rows go to every rank"""
    from sgl_amd.dist import ShardedPropagator, gather_piece_bounds
    args, K, x0, blk = job.args, job.K, job.x0, job.block
    pieces, handles, mine = job.engine.block_piece_spmms(args, blk, args.pieces)
    pb = gather_piece_bounds(mine) if job.world > 1 else np.asarray([[int(v) for v in mine]], dtype=np.int64)
    prop = ShardedPropagator(pieces, pb, job.rank, job.world, job.n)
    # column chunks live as separate contiguous matrices: whole cache lines per gathered chunk row
    x_chunks = [x0] if len(chunks) == 1 else [x0[:, a:b].contiguous() for a, b in chunks]
    cbufs = [[torch.empty_like(xc) for _ in range(job.nbuf)] for xc in x_chunks]
    ybufs = [[torch.empty((prop.hi - prop.lo, xc.shape[1]), dtype=xc.dtype, device=xc.device) for _ in range(K)]
             for xc in x_chunks]
    return {"prop": prop, "handles": handles, "x_chunks": x_chunks, "cbufs": cbufs, "ybufs": ybufs,
            "bounds": [int(v) for v in pb[:, 0]] + [int(pb[-1, -1])]}


def _rows_halo(job, chunks):
    """row-sharded layout on compact tables: a rank holds its own rows and the rows of each peer its block gathers, and receives
    only those between hops"""
    K, x0, blk = job.K, job.x0, job.block
    bounds = [int(v) for v in job.bounds]
    plan, prop, cblk = job.engine.block_halo(job.args, blk, bounds)
    t0 = prop.table_from_full(x0)                            # every rank of the bench holds x0; a real job passes its own rows
    tables = [t0] if len(chunks) == 1 else [t0[:, a:b].contiguous() for a, b in chunks]
    if len(chunks) > 1:
        del t0
    bufs = [[torch.empty_like(t) for _ in range(job.nbuf)] for t in tables]
    ybufs = [[torch.empty((plan.n_own, t.shape[1]), dtype=t.dtype, device=t.device) for _ in range(K)] for t in tables]
    return {"plan": plan, "prop": prop, "cblk": cblk, "tables": tables, "bufs": bufs, "ybufs": ybufs}


def _build_rows(job, ref=None):
    """The contract layout.  The default is a FIXED path (--exchange halo --col-chunks 3): nothing about the headline run is
    decided by a timing race, so two runs of the same command do the same thing.  --col-chunks auto (opt-in): how finely the feature
    block is cut for the pipelined exchange trades the un-overlapped head and tail of a step against per-chunk launch cost and
    against the gather efficiency of narrow chunks (profiles/r03_scale_model.md: 3 or 4 chunks win in the model when the links are
    the bound, 2 when compute is), and that depends on what the links deliver -- so 2, 3 and 4 chunks (d = 100: 64 + 36,
    32 + 32 + 36, 32 + 32 + 32 + 4; always 4 lines per gathered row) are built, validated and timed (untimed setup; the exchange is
    selected once, with the first) and the fastest is kept.  Whatever is chosen is validated before it is timed; a need-aware
    exchange that fails to build or to validate on ANY rank is replaced by the full-replica p2p exchange on ALL ranks
    (plan.halo_rejected says so)."""
    args = job.args
    auto = str(args.col_chunks) == "auto"
    measure = auto and job.world > 1 and job.nbuf > 0
    counts = [2, 3, 4] if measure else [2 if auto else int(args.col_chunks)]
    from sgl_amd.dist import column_chunks
    seen = []
    for nc in list(counts):                               # narrow feature blocks: several counts give the same cut
        cut = column_chunks(job.d, nc)
        if cut in seen:
            counts.remove(nc)
        seen.append(cut)
    base_info = dict(job.info)
    live = job.info                                       # callers hold a reference to this dict: it is edited in place

    def set_info(d_):
        live.clear()
        live.update(d_)

    def attempt(exchange):
        """build (and, when there is a choice or the exchange is the need-aware one, validate and time) every chunk count"""
        best, timing = None, {}
        for nc in counts:
            set_info(base_info)
            if getattr(job, "rows_inbound_bytes", None) is not None:
                job.rows_inbound_bytes = None
            cand, good = None, True
            try:
                cand = _build_rows_for(job, ref, nc, exchange)
            except Exception as e:  # noqa: BLE001  (same code on every rank, so an error is too; agree() settles it)
                good = False
                sys.stderr.write(f"[bench] rows with {nc} column chunks ({exchange or args.exchange}) could not be built on rank "
                                 f"{job.rank}: {e!r}\n")
            if not job.agree(good):
                exchange = job.info.get("exchange") or exchange or args.exchange
                continue
            exchange = job.info["exchange"]
            for k in ("exchange_candidates_ms", "exchange_selected_by", "push_peer_rows_skipped", "push_validated_against",
                      "push_rejected", "halo_a2a_rejected"):
                if k in live:
                    base_info[k] = live[k]                # the selection happens once: its record goes with every candidate
            if measure or str(exchange).startswith("halo"):
                # two phases, each closed by an agreement every rank reaches: a rank whose step raised never leaves the others
                # alone inside the collectives of the check
                good = True
                try:
                    cand["step"]()
                    job.engine.sync()
                except Exception as e:  # noqa: BLE001
                    good = False
                    sys.stderr.write(f"[bench] rows with {nc} column chunks ({exchange}) failed on rank {job.rank}: {e!r}\n")
                if not job.agree(good):
                    continue
                try:
                    job.sync_all()
                    good = bool(cand["check"]())
                except Exception as e:  # noqa: BLE001
                    good = False
                    sys.stderr.write(f"[bench] rows with {nc} column chunks ({exchange}): check failed on rank {job.rank}: {e!r}\n")
                if not job.agree(good):
                    continue
                timing[nc] = job.timed_s(cand["step"], reps=2, warm=0)
            if best is None or (nc in timing and timing[nc] < timing.get(best[0], float("inf"))):
                best = (nc, cand, dict(job.info), getattr(job, "rows_inbound_bytes", None))
            del cand
        return best, timing, exchange

    best, timing, exchange = attempt(None)
    if best is None and str(exchange).startswith("halo"):
        # the need-aware exchange did not reproduce itself on this system: the run goes on with the full-replica exchange
        base_info["halo_rejected"] = f"{exchange}: validation failed, fell back to the full-replica exchange"
        best, timing, exchange = attempt(getattr(job.engine, "transports", ("p2p",))[0])
    if best is None:
        raise RuntimeError("no column chunking of the row-sharded layout passed validation")
    nc, cand, info, inbound = best
    set_info(info)
    job.rows_inbound_bytes = inbound
    if measure and timing:
        job.info["col_chunks_candidates_ms"] = {str(k): round(v * 1e3, 3) for k, v in timing.items()}
        job.info["col_chunks_selected_by"] = "measured step time (--col-chunks auto, opt-in)"
    job.col_chunks_chosen = nc
    return cand


def _build_rows_for(job, ref, n_chunks, exchange_fixed=None):
    """A_hat row-sharded IN STORAGE (every rank multiplies the block it alone holds) + per-hop
    all-gather -- need-aware (halo) or of the full replica --, column chunks software-pipelined across hops.  Validated
    without any replica of A_hat: the exchanged rows by exact bit-checksums, the local SpMM by sampled rows recomputed in fp64."""
    from sgl_amd.dist import column_chunks, exchange_checksums, halo_checksums
    args, K, blk = job.args, job.K, job.block
    chunks = column_chunks(job.d, n_chunks)
    job.info.update({"row_pieces": args.pieces, "col_chunks": chunks})
    can_halo = job.world > 1 and job.nbuf > 0 and getattr(job.engine, "block_halo", None) is not None
    want = exchange_fixed or args.exchange
    if want == "staged":
        want = "p2p"
    want_a2a = want == "halo_a2a"
    if want_a2a:
        want = "halo"
    if want == "halo" and not can_halo:
        # one rank, a single hop (nothing is exchanged) or an engine without the need-aware exchange: the full-replica transport
        want = exchange_fixed = getattr(job.engine, "transports", ("p2p",))[0]
        want_a2a = False
    full = halo = None
    if want != "halo" or not can_halo:
        full = _rows_full_replica(job, chunks)
    if can_halo and want in ("auto", "halo"):
        halo = _rows_halo(job, chunks)
        if want_a2a:
            exchange_fixed = "halo_a2a"
    exchange = exchange_fixed if exchange_fixed in ("halo", "halo_a2a", "p2p", "allgather", "staged") else _select_exchange(job, full, halo)
    job.info["exchange"] = exchange
    check_fn = getattr(job.engine, "sampled_rows_check", None)
    if exchange in ("halo", "halo_a2a"):
        full = None                                           # the replicas of the other candidate are released
        plan, prop, cblk, tables, hbufs, ybufs = (halo[k] for k in ("plan", "prop", "cblk", "tables", "bufs", "ybufs"))
        prop.collective = exchange == "halo_a2a"
        frac = torch.tensor([plan.skipped_fraction, float(plan.n_ghost)], dtype=torch.float64, device=job.device)
        if job.world > 1:
            import torch.distributed as dist
            mx = frac.clone()
            dist.all_reduce(frac)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            frac /= job.world
        else:
            mx = frac
        job.info["halo"] = dict(plan.describe(), exchange_skipped_fraction_mean=round(float(frac[0]), 4),
                                ghost_rows_max_rank=int(mx[1]))
        job.rows_inbound_bytes = float(mx[1]) * job.d * 4

        def step():
            return prop.propagate_chunked(tables, K, buffers=hbufs, y_buffers=ybufs, hops_in_buffers=job.nbuf >= K - 1)

        def check():
            hops = step()
            job.engine.sync()
            ok = True
            for c in range(len(tables)):
                t_prev = tables[c] if K == 1 else hbufs[c][(K - 2) % job.nbuf]
                if K >= 2:                                    # the ghosts of hop K-1 are the owners' rows, bit for bit
                    ok = halo_checksums(plan, t_prev, hops[K - 1][c]) and ok     # collective: never behind a short circuit
                if check_fn is not None:
                    ok = check_fn(cblk, t_prev, hops[K][c]) and ok
            if ref is not None:
                ok = ok and all(ref.close(t, prop.lo, prop.hi, a, b) for t, (a, b) in zip(hops[K], chunks))
            return ok
        return {"step": step, "check": check, "halves": (prop, tables, hbufs),
                "describe": f"row-sharded x{job.world} (A_hat stored as one row block per GPU) + per-hop need-aware all-gather "
                            f"({'halo as one all_to_all_single' if prop.collective else 'halo'}: {plan.n_ghost} of {plan.rows_in_full} "
                            f"foreign rows gathered on rank 0, packed), "
                            f"{len(chunks)} column chunks pipelined across hops"}
    halo = None
    prop, x_chunks, cbufs, ybufs, bounds = (full[k] for k in ("prop", "x_chunks", "cbufs", "ybufs", "bounds"))
    x0 = job.x0
    if exchange in ("p2p", "allgather", "staged"):
        prop.transport = exchange
    if exchange == "push":
        def step():
            return prop.propagate_push(x_chunks, K)
    elif len(chunks) == 1:
        def step():
            return [[t] for t in prop.propagate(x0, K, x_buffers=cbufs[0], y_buffers=ybufs[0], hops_in_buffers=job.nbuf >= K - 1)]
    else:
        def step():
            return prop.propagate_chunked(x_chunks, K, buffers=cbufs, y_buffers=ybufs, hops_in_buffers=job.nbuf >= K - 1)

    def check():
        hops = step()
        job.engine.sync()
        ok = True
        for c, xc in enumerate(x_chunks):
            # what the last hop read: the replica of hop K-1 (the input itself when K == 1)
            x_prev = xc if K == 1 else (prop._push_local[c][(K - 2) % 2] if exchange == "push" else cbufs[c][(K - 2) % job.nbuf])
            if K >= 2 and exchange != "push":     # every rank's rows of hop K-1 arrived intact in my replica (the push
                ok = exchange_checksums(x_prev, hops[K - 1][c], bounds) and ok   # transport skips rows this rank never gathers);
                                                                                 # collective: never behind a short circuit
            if check_fn is not None:
                ok = check_fn(blk, x_prev, hops[K][c]) and ok
        if ref is not None:  # a replica-based reference chain exists anyway (alternative layouts were asked for)
            ok = ok and all(ref.close(t, prop.lo, prop.hi, a, b) for t, (a, b) in zip(hops[K], chunks))
        return ok
    return {"step": step, "check": check, "halves": (prop, x_chunks, cbufs),
            "describe": f"row-sharded x{job.world} (A_hat stored as one row block per GPU) + per-hop all-gather ({exchange}), "
                        f"{args.pieces} row pieces x {len(chunks)} column chunks"}
