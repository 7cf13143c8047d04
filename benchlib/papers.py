"""The papers100M-shaped section of an S1 bench line: the same measurement on a hashed 111 M-node graph, row-sharded in storage
over the same ranks (full replicas or, with the need-aware exchange, compact tables built from each rank's own feature rows)."""
import time

import numpy as np
import torch

from .common import HBM_PEAK_BYTES, algorithmic_bytes_per_hop, workload_text


def _n_chunks(args):
    """column chunks of the papers100M-shaped section: the count the S1 job settled on (--col-chunks auto), else the flag"""
    v = getattr(args, "col_chunks_chosen", None) or args.col_chunks
    return 2 if str(v) == "auto" else int(v)


def papers_section(args, engine, rank, world, exchange, wl=None):
    """The same measurement on an ogbn-papers100M-shaped graph, row-sharded in storage over the same
    ranks: every rank generates ITS nnz-balanced row block and the feature replica on its own GPU (hash keyed by
    (seed, row): no traffic, no rank ever sees the whole graph), k = 3 hops with the per-hop all-gather, only the last
    hop retained (hop shards are written straight into the next replica).  Returns the dict for the JSON line."""
    import torch.distributed as dist
    from sgl_amd import synthetic
    from sgl_amd.dist import ShardedPropagator, exchange_checksums, gather_piece_bounds
    # k = 3 by default (bounded: the section shares the bench's time budget); --papers-k 10 = BASELINE config 5's own hop count
    wl = dict(synthetic.WORKLOADS["S3_papers"], k=int(getattr(args, "papers_k", 3) or 3)) if wl is None else wl      # (tests pass a small hashed workload)
    n, d, K = wl["n"], wl["d"], wl["k"]
    t0 = time.perf_counter()
    bounds, nnz = engine.hashed_bounds(args, wl, world)
    blk = engine.hashed_block(args, wl, int(bounds[rank]), int(bounds[rank + 1]))
    if exchange in ("halo", "halo_a2a") and world > 1 and getattr(engine, "block_halo", None) is not None:
        return _papers_halo(args, engine, rank, world, wl, bounds, nnz, blk, t0, collective=exchange == "halo_a2a")
    x0 = engine.features(args, wl)
    pieces, handles, mine = engine.block_piece_spmms(args, blk, args.pieces)
    pb = gather_piece_bounds(mine) if world > 1 else np.asarray([[int(v) for v in mine]], dtype=np.int64)
    prop = ShardedPropagator(pieces, pb, rank, world, n, transport=exchange if exchange in ("p2p", "allgather", "staged") else "p2p")
    # N > 1: the feature block is held as two column chunks, software-pipelined across hops (chunk A's all-gather is in
    # flight while chunk B is multiplied and hop h+1 of chunk A only waits for A's own exchange): the job is communication
    # bound there (49.8 GB in-bound per rank per hop at 8 ranks) and this hides the SpMM behind the transfers.
    from sgl_amd.dist import column_chunks
    chunks = column_chunks(d, _n_chunks(args) if world > 1 else 1)
    if len(chunks) > 1:
        x_chunks = [x0[:, a:b].contiguous() for a, b in chunks]
        del x0
    else:
        x_chunks = [x0]
    cbufs = [[torch.empty_like(xc) for _ in range(2)] for xc in x_chunks]
    # the last hop reads replica (K-2) % 2, so its output can live in this rank's rows of the other one: no extra memory
    ylast = [cb[(K - 1) % 2][prop.lo:prop.hi] for cb in cbufs]

    def step():
        if len(x_chunks) == 1:
            return [[t] for t in prop.propagate(x_chunks[0], K, x_buffers=cbufs[0], y_buffers=[None] * (K - 1) + [ylast[0]],
                                               in_place=True)]
        return prop.propagate_chunked(x_chunks, K, buffers=cbufs, y_buffers=[[None] * (K - 1) + [yl] for yl in ylast],
                                      in_place=True)

    def sync_all():
        engine.sync()
        if world > 1:
            dist.barrier()
            engine.sync()

    hops = step()                                             # warm-up + validation
    sync_all()
    ok = True
    bnds = [int(v) for v in pb[:, 0]] + [int(pb[-1, -1])]
    for c in range(len(x_chunks)):
        x_prev = cbufs[c][(K - 2) % 2]
        ok = exchange_checksums(x_prev, x_prev[prop.lo:prop.hi], bnds) and ok      # collective: never behind a short circuit
        ok = engine.sampled_rows_check(blk, x_prev, hops[K][c]) and ok
    dev_ = x_chunks[0].device
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev_)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    steps = 2
    sync_all()
    t_a = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    el = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64, device=dev_)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    hop_s = elapsed / (K * steps)
    alg = algorithmic_bytes_per_hop(n, nnz, d) / world
    inbound = (world - 1) / world * n * d * 4
    return {"workload": workload_text("S3_papers", K), "n_nodes": n, "nnz": nnz, "feat_dim": d, "prop_steps": K,
            "n_gpus": world, "steps": steps, "validated": bool(flag.item()),
            "value": nnz * d * K * steps / elapsed, "unit": "edge\u00b7featdim/s", "ms_per_step": elapsed * 1e3 / steps,
            "ms_per_hop": hop_s * 1e3,
            "roofline": {"bound": "hbm", "achieved": alg / hop_s / 1e9, "peak": HBM_PEAK_BYTES / 1e9, "unit": "GB/s",
                         "frac": alg / hop_s / HBM_PEAK_BYTES, "algorithmic_bytes_per_launch": alg,
                         "note": "per-GPU share of one hop / wall time per hop (the all-gather is inside that time for N>1)"},
            "parallelism": "single GPU" if world == 1 else
                           f"row-sharded x{world} (A_hat stored as one row block per GPU) + per-hop all-gather ({prop.transport}), "
                           f"{args.pieces} row pieces x {len(x_chunks)} column chunks pipelined across hops, "
                           f"{inbound / 1e9:.1f} GB in-bound per rank per hop",
            "hops_retained": "last only (hop shards are written into the next replica in place)",
            "setup_s": round(time.perf_counter() - t0 - elapsed * (steps + 1) / steps, 2)}


def _papers_halo(args, engine, rank, world, wl, bounds, nnz, blk, t0, collective=False):
    """papers100M-shaped section with the need-aware exchange: no rank ever holds the 57 GB feature matrix -- it generates its
    OWN feature rows, fetches the rows its block gathers from their owners (the same exchange that runs between hops) and keeps
    compact tables [own rows | ghosts per peer]; k hops in place, only the last retained."""
    import torch.distributed as dist
    from sgl_amd.dist import column_chunks, halo_checksums
    n, d, K = wl["n"], wl["d"], wl["k"]
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    plan, prop, cblk = engine.block_halo(args, blk, [int(b) for b in bounds])
    prop.collective = bool(collective)
    x_own = engine.feature_rows(args, wl, lo, hi)
    chunks = column_chunks(d, _n_chunks(args))
    tables = [prop.table_from_own(x_own if len(chunks) == 1 else x_own[:, a:b].contiguous(), key=("init", c))
              for c, (a, b) in enumerate(chunks)]
    del x_own
    prop._send.clear()
    bufs = [[torch.empty_like(t) for _ in range(2)] for t in tables]
    ylast = [b[(K - 1) % 2][:plan.n_own] for b in bufs]

    def step():
        return prop.propagate_chunked(tables, K, buffers=bufs, y_buffers=[[None] * (K - 1) + [yl] for yl in ylast], in_place=True)

    def sync_all():
        engine.sync()
        dist.barrier()
        engine.sync()

    hops = step()                                             # warm-up + validation
    sync_all()
    ok = True
    for c in range(len(tables)):
        t_prev = bufs[c][(K - 2) % 2] if K >= 2 else tables[c]
        if K >= 2:
            ok = halo_checksums(plan, t_prev, t_prev[:plan.n_own]) and ok           # collective: never behind a short circuit
        ok = engine.sampled_rows_check(cblk, t_prev, hops[K][c]) and ok
    dev_ = tables[0].device
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev_)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    stats = torch.tensor([float(plan.n_ghost), plan.skipped_fraction], dtype=torch.float64, device=dev_)
    mx = stats.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(stats)
    steps = 2
    sync_all()
    t_a = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    el = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64, device=dev_)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    hop_s = elapsed / (K * steps)
    alg = algorithmic_bytes_per_hop(n, nnz, d) / world
    inbound = float(mx[0]) * d * 4
    return {"workload": workload_text("S3_papers", K), "n_nodes": n, "nnz": nnz, "feat_dim": d, "prop_steps": K,
            "n_gpus": world, "steps": steps, "validated": bool(flag.item()),
            "value": nnz * d * K * steps / elapsed, "unit": "edge\u00b7featdim/s", "ms_per_step": elapsed * 1e3 / steps,
            "ms_per_hop": hop_s * 1e3,
            "roofline": {"bound": "hbm", "achieved": alg / hop_s / 1e9, "peak": HBM_PEAK_BYTES / 1e9, "unit": "GB/s",
                         "frac": alg / hop_s / HBM_PEAK_BYTES, "algorithmic_bytes_per_launch": alg,
                         "note": "per-GPU share of one hop / wall time per hop (pack kernel and exchange are inside that time)"},
            "parallelism": f"row-sharded x{world} (A_hat stored as one row block per GPU) + per-hop need-aware all-gather (halo), "
                           f"{len(chunks)} column chunks pipelined across hops, {inbound / 1e9:.1f} GB in-bound per rank per hop "
                           f"(a full all-gather: {(world - 1) / world * n * d * 4 / 1e9:.1f} GB)",
            "halo": dict(plan.describe(), exchange_skipped_fraction_mean=round(float(stats[1]) / world, 4),
                         ghost_rows_max_rank=int(mx[0])),
            "hops_retained": "last only (hop shards are written into the next table in place)",
            "setup_s": round(time.perf_counter() - t0 - elapsed * (steps + 1) / steps, 2)}
