"""After the timed region, never part of `value`: the halves of an exchanging layout's hop in isolation and a link probe."""
import torch


def _hop_halves(job, prop, x_chunks, cbufs, inbound):
    """the two halves of an exchanging layout's hop in isolation: SpMM only / exchange only (MAX over ranks)"""
    ys = prop.spmm_only(x_chunks)
    spmm_ms = job.timed_s(lambda: prop.spmm_only(x_chunks), reps=3) * 1e3
    xnext = [b[0] for b in cbufs]
    exch_ms = job.timed_s(lambda: prop.exchange_only(ys, xnext), reps=3) * 1e3 if prop._exchanging() else 0.0
    return {"spmm_only_ms_per_hop_max_rank": spmm_ms, "exchange_only_ms_per_hop_max_rank": exch_ms,
            "inbound_bytes_per_rank_per_hop": inbound,
            "exchange_inbound_GBps_per_rank": (inbound / (exch_ms * 1e-3) / 1e9) if exch_ms > 0 else None}


def _per_hop(job, prop, x_chunks, cbufs, measured_step_ms):
    """Per column chunk, MAX over ranks, each in isolation: the SpMM, the pack kernel (need-aware exchange only) and the wire
    time of one grouped exchange (exchange minus pack); the bytes the busiest link carries; what fraction of the exchange the
    pipelined step hid; and the schedule model's prediction from those very numbers next to the measured step
    (benchlib/model.py) -- a scaling line that explains itself against profiles/r03_scale_model.md."""
    from .model import explain
    if not prop._exchanging() or job.K < 2:
        return None
    spmm, pack, xfer, link_bytes = [], [], [], []
    rows_link = job.max_over_ranks(float(prop.busiest_link_rows()))
    has_pack = hasattr(prop, "pack_only")
    for c, xc in enumerate(x_chunks):
        ys = prop.spmm_only([xc])
        spmm.append(job.timed_s(lambda: prop.spmm_only([xc]), reps=3) * 1e3)
        nxt = [cbufs[c][0]]
        ex = job.timed_s(lambda: prop.exchange_only(ys, nxt, keys=[c]), reps=3) * 1e3
        pk = job.timed_s(lambda: prop.pack_only(ys, keys=[c]), reps=3) * 1e3 if has_pack else 0.0
        pack.append(pk)
        xfer.append(max(ex - pk, 0.0))
        link_bytes.append(rows_link * xc.shape[1] * 4)
    return explain(spmm, pack, xfer, link_bytes, job.K, measured_step_ms, job.world,
                   single_gpu_ms=getattr(job, "single_gpu_ms_replayed", None))


def _link_probe(job):
    """What the links of this node deliver to the two communication patterns the layouts use (reporting only, a few
    tens of milliseconds): one all_to_all with S bytes per peer (the relay's phases) and a pairwise exchange between
    ranks 2i and 2i+1 (what a 2-rank column group would get from its single direct link)."""
    import torch.distributed as dist
    world, device = job.world, job.device
    out = {}
    for mb in (1, 8, 32):
        elems = mb * (1 << 20) // 4
        try:
            src = torch.zeros(world * elems, dtype=torch.float32, device=device)
            dst = torch.empty_like(src)
            t = job.timed_s(lambda: dist.all_to_all_single(dst, src), reps=3)
            out[f"all_to_all_{mb}MB_per_peer_GBps_per_link"] = mb * (1 << 20) / t / 1e9
        except Exception as e:  # noqa: BLE001
            out[f"all_to_all_{mb}MB_per_peer_GBps_per_link"] = f"unavailable: {e!r}"[:120]
            break
    try:
        peer = job.rank ^ 1
        buf_s = torch.zeros(16 << 20, dtype=torch.float32, device=device)       # 64 MiB each way
        buf_r = torch.empty_like(buf_s)

        def pair():
            if peer < world:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf_s, peer), dist.P2POp(dist.irecv, buf_r, peer)]):
                    w.wait()
        t = job.timed_s(pair, reps=3)
        out["pair_exchange_64MB_GBps_per_direction"] = (64 << 20) / t / 1e9
    except Exception as e:  # noqa: BLE001
        out["pair_exchange_64MB_GBps_per_direction"] = f"unavailable: {e!r}"[:120]
    return out


def _diagnostics(job, halves, measured_step_ms=None):
    """after the timed region, never part of `value`.  Row-sharded layout: its SpMM and all-gather halves and the
    achieved rate per link; grid layout: the same two halves of the relayed exchange (every byte crosses two links)."""
    diag = None
    if "rows" in halves and job.nbuf > 0:
        inbound = getattr(job, "rows_inbound_bytes", None) or (job.world - 1) / job.world * job.n * job.d * 4
        diag = _hop_halves(job, *halves["rows"], inbound)
        ms = diag["exchange_only_ms_per_hop_max_rank"]
        diag["exchange_GBps_per_link"] = (inbound / max(job.world - 1, 1) / (ms * 1e-3) / 1e9) if ms > 0 else None
        prop = halves["rows"][0]
        if measured_step_ms is not None and job.info.get("layout", "rows") == "rows":
            try:
                diag["per_hop"] = _per_hop(job, *halves["rows"], measured_step_ms)
            except Exception as e:  # noqa: BLE001  (reporting only)
                diag["per_hop"] = {"failed": repr(e)[:200]}
        if hasattr(prop, "pack_only"):                        # need-aware exchange: the pack kernel alone (inside exchange_only too)
            ys = prop.spmm_only(halves["rows"][1])
            diag["pack_only_ms_per_hop_max_rank"] = job.timed_s(lambda: prop.pack_only(ys), reps=3) * 1e3
            timing = getattr(prop, "pack_timing_ms", None)
            if timing:                                        # rank 0's choice per (rows, columns) of a chunk: peer order or own-row order
                diag["pack_order_ms_rank0"] = {f"{r}x{w}": t for (r, w), t in timing.items()}
    if "grid" in halves and job.nbuf > 0:
        prop, x_chunks, cbufs = halves["grid"]
        inbound = (prop.world - 1) / prop.world * job.n * x_chunks[0].shape[1] * 4
        g = _hop_halves(job, prop, x_chunks, cbufs, inbound)
        ms = g["exchange_only_ms_per_hop_max_rank"]
        # two phases, each moving 1/world of the block over every one of the world-1 links
        g["relay_GBps_per_link"] = (2 * inbound / job.world / (ms * 1e-3) / 1e9) if ms > 0 else None
        diag = dict(diag or {}, grid=g)
    if job.world > 1 and getattr(job.engine, "probe_links", True):
        diag = dict(diag or {}, links=_link_probe(job))
    return diag
