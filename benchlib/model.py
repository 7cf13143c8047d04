"""The schedule model of the row-sharded job: what a k-hop step of the chunk-pipelined loop costs, given the per-chunk SpMM and
pack times of a rank and the time one grouped exchange of a chunk occupies the links.  tools/scale_model.py feeds it with
one-GPU measurements and a link rate as the free parameter (profiles/r03_scale_model.md); bench.py feeds it with what an N-rank
run measured itself and prints the prediction NEXT TO the measured step (config.diagnostics.per_hop.model), so a scaling line
explains itself: measured link rate against the rate the 5x target needs."""

XGMI_PEAK_GBPS_PER_DIRECTION = 76.8     # one xGMI link of an MI355X, one direction (7 links x 153.6 GB/s bidirectional per GPU)
MODEL_RATES_GBPS = (35.0, 45.0, 55.0, 65.0, XGMI_PEAK_GBPS_PER_DIRECTION)


def simulate(spmm, pack, xfer, k):
    """ms per step of the chunk-pipelined schedule HaloPropagator / ShardedPropagator.propagate_chunked issue.
    spmm[c], pack[c]: ms on the compute stream; xfer[c]: ms one grouped exchange of chunk c occupies the links (all peers in
    parallel: the busiest link decides).  One compute resource, one link resource; hop h of chunk c starts when the compute
    stream is free AND chunk c's exchange of hop h-1 has landed; the last hop needs no exchange."""
    C = len(spmm)
    t_comp = t_link = 0.0
    landed = [0.0] * C
    for h in range(1, k + 1):
        for c in range(C):
            start = max(t_comp, landed[c])
            t_comp = start + spmm[c]
            if h < k:
                t_comp += pack[c]
                t_link = max(t_link, t_comp) + xfer[c]
                landed[c] = t_link
    return t_comp


def rate_for_speedup(spmm, pack, link_bytes, k, single_gpu_ms, speedup, lo=1.0, hi=4000.0):
    """the link rate (GB/s per direction on the busiest link) at which the modelled step reaches single_gpu_ms / speedup;
    None when the schedule cannot get there even with free links"""
    def step(B):
        return simulate(spmm, pack, [b / (B * 1e9) * 1e3 for b in link_bytes], k)
    target = single_gpu_ms / speedup
    if step(hi) > target:
        return None
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if step(mid) <= target:
            hi = mid
        else:
            lo = mid
    return hi


def explain(spmm, pack, xfer, link_bytes, k, measured_step_ms, world, single_gpu_ms=None):
    """The per-hop breakdown of an exchanging job and the model's view of it.  All inputs are MAX-over-ranks measurements of
    the run itself: per-chunk SpMM, pack and wire (exchange minus pack) milliseconds, the bytes the busiest link carries per
    chunk and hop.  Returns the dict bench.py prints under config.diagnostics.per_hop."""
    spmm, pack, xfer = [float(v) for v in spmm], [float(v) for v in pack], [float(v) for v in xfer]
    # what the compute stream does whatever the links deliver: K hops of SpMM and the K - 1 pack passes between them
    compute_stream = k * sum(spmm) + (k - 1) * sum(pack)
    wire_total = (k - 1) * sum(xfer)
    exposed = max(measured_step_ms - compute_stream, 0.0)
    wire = sum(xfer)
    rate = (sum(link_bytes) / (wire * 1e-3) / 1e9) if wire > 0 else None
    out = {
        "spmm_ms": sum(spmm), "pack_ms": sum(pack), "exchange_wire_ms": wire,
        "per_chunk": {"spmm_ms": spmm, "pack_ms": pack, "exchange_wire_ms": xfer, "busiest_link_bytes": [int(b) for b in link_bytes]},
        "link_GBps_per_direction_busiest_link": rate,
        "link_frac_of_xgmi_peak": (rate / XGMI_PEAK_GBPS_PER_DIRECTION) if rate else None,
        # share of the step's wire time (the K - 1 exchanged hops) that the pipelined schedule hid behind compute-stream work
        "overlap_fraction": (min(max(1.0 - exposed / wire_total, 0.0), 1.0) if wire_total > 0 else None),
        "compute_stream_ms_per_step": compute_stream,
        "exposed_exchange_ms_per_step": exposed,
    }
    model = {"schedule": "compute stream: spmm(c), pack(c) per chunk; one grouped exchange per chunk and hop on the links; hop h+1 "
                         "of a chunk waits for that chunk's exchange only (benchlib/model.py)",
             "predicted_ms_per_step_at_measured_rates": simulate(spmm, pack, xfer, k),
             "measured_ms_per_step": measured_step_ms,
             "predicted_ms_per_step_free_links": simulate(spmm, pack, [0.0] * len(spmm), k),
             "predicted_ms_per_step_by_link_GBps": {
                 f"{B:g}": simulate(spmm, pack, [b / (B * 1e9) * 1e3 for b in link_bytes], k) for B in MODEL_RATES_GBPS}}
    if single_gpu_ms:
        model["single_gpu_ms_per_step"] = single_gpu_ms
        model["speedup_measured"] = single_gpu_ms / measured_step_ms if measured_step_ms > 0 else None
        model["speedup_by_link_GBps"] = {B: single_gpu_ms / t for B, t in model["predicted_ms_per_step_by_link_GBps"].items()}
        if world == 8:
            model["link_GBps_needed_for_5x"] = rate_for_speedup(spmm, pack, link_bytes, k, single_gpu_ms, 5.0)
    out["model"] = model
    return out
