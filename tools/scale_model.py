#!/usr/bin/env python3
"""What the row-sharded job (the contract layout) will cost per step at G = 2 / 4 / 8, from quantities measured on ONE GPU.

For every rank r of a G-rank job over the products-shaped workload (S1, k = 3, d = 100) this builds exactly what the rank would
hold -- its nnz-balanced row block of A_hat with the columns relabelled to its compact table [own rows | ghosts per peer]
(HaloPlan.offline: the same plan the collective constructor produces) -- and measures on this GPU:

    spmm_c   the rank's SpMM per column chunk (what runs between exchanges)
    pack_c   the pack kernel per chunk (rows the peers gather -> send buffer; the faster of peer order / own-row order)
    in / out bytes per peer link per hop (need-aware) next to the full all-gather volume

The exchange itself cannot be measured here; it enters as a link rate B (GB/s per direction per link).  A small event simulation
of the schedule HaloPropagator.propagate_chunked issues (compute stream: spmm, pack; links: one grouped exchange per chunk and hop;
hop h+1 of chunk c waits for chunk c's exchange only) turns the measurements into ms per step for a range of B, for the
need-aware exchange and for the full all-gather, and solves for the B at which the 8-GPU job reaches 5x the single-GPU step.

    python tools/scale_model.py [--papers]  > profiles/r03_scale_model.md
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sgl_amd import device as dev  # noqa: E402
from sgl_amd import synthetic  # noqa: E402
from sgl_amd.dist import HaloPlan, balanced_bounds, column_chunks  # noqa: E402

K = 3
ALT_CHUNKS = (1, 2, 3, 4)     # column-chunk counts of the pipelined schedule modelled next to each other (bench.py times 2, 3 and 4)


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def simulate(spmm, pack, xfer, k=K):
    """ms per step of the chunk-pipelined schedule: the model bench.py prints next to a measured N-rank step (benchlib/model.py)"""
    from benchlib.model import simulate as sim
    return sim(spmm, pack, xfer, k)


def rank_measurements(rowptr, col, val, rp_host, bounds, r, n, d, x0, chunks, device):
    lo, hi = int(bounds[r]), int(bounds[r + 1])
    nb, ne = int(rp_host[lo]), int(rp_host[hi])
    plan = HaloPlan.offline(r, bounds, n, lambda q: col[int(rp_host[bounds[q]]):int(rp_host[bounds[q + 1]])])
    rp_local = (rowptr[lo:hi + 1] - rowptr[lo]).contiguous()
    csr = dev.DeviceCSR(rp_local, plan.relabel(col[nb:ne]), val[nb:ne], (hi - lo, plan.n_compact))
    t0 = x0.index_select(0, plan.global_ids)
    out = {"rank": r, "own": plan.n_own, "ghost": plan.n_ghost, "skipped": plan.skipped_fraction,
           "in_peer_max": max(int(t.numel()) for t in plan.need) * d * 4, "in_total": plan.n_ghost * d * 4,
           "out_peer_max": max(int(t.numel()) for t in plan.send_rows) * d * 4, "out_total": int(plan.send_off[-1]) * d * 4,
           "full_in_total": (n - plan.n_own) * d * 4,
           "full_peer_max": max(int(bounds[q + 1] - bounds[q]) for q in range(len(bounds) - 1) if q != r) * d * 4,
           "spmm": [], "pack": []}
    y_full = torch.empty((hi - lo, d), device=device)
    out["spmm_whole"] = timed(lambda: csr.spmm(t0, out=y_full))

    def per_chunk(chs):
        sp, pk = [], []
        for a, b in chs:
            t = t0[:, a:b].contiguous()
            y = torch.empty((hi - lo, b - a), device=device)
            sp.append(timed(lambda: csr.spmm(t, out=y)))
            buf = torch.empty((int(plan.send_off[-1]), b - a), device=device)
            pk.append(min(timed(lambda: dev.scatter_rows(y, plan.pack_src, plan.pack_dst, buf)),
                          timed(lambda: dev.gather_rows(y, plan.send_idx, out=buf))) if buf.shape[0] else 0.0)   # the faster order, like HaloPropagator
            del t, y, buf
        return sp, pk
    out["spmm"], out["pack"] = per_chunk(chunks)
    out["alt"] = {nc: per_chunk(column_chunks(d, nc)) for nc in ALT_CHUNKS}
    # the same block against the FULL replica (global column ids): what the plain all-gather layout multiplies
    csr_g = dev.DeviceCSR(rp_local, col[nb:ne], val[nb:ne], (hi - lo, n))
    out["spmm_whole_full_replica"] = timed(lambda: csr_g.spmm(x0, out=y_full))
    out["spmm_full_replica"] = []
    for a, b in chunks:
        t = x0[:, a:b].contiguous()
        y = torch.empty((hi - lo, b - a), device=device)
        out["spmm_full_replica"].append(timed(lambda: csr_g.spmm(t, out=y)))
        del t, y
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--papers", action="store_true", help="add rank 0 of the 8-rank papers100M-shaped job (57 GB regime)")
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = synthetic.WORKLOADS["S1_products"]
    n, d = wl["n"], wl["d"]
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    del a_ptr, a_col, a_val
    rp_host = rowptr.cpu().numpy()
    nnz = int(rp_host[-1])
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    full = dev.DeviceCSR(rowptr, col, val, (n, n))
    outs = [dev.alloc_rows(n, d, device) for _ in range(K)]
    t1 = timed(lambda: full.spmm_chain(x0, K, outs=outs))
    del outs
    chunks = column_chunks(d, 2)
    rates = (25, 35, 45, 55, 65, 76.8)

    print("# Scaling model of the row-sharded job (contract layout), from one-GPU measurements\n")
    print(f"`python tools/scale_model.py`: workload S1 (N = {n}, nnz(A_hat) = {nnz}, d = {d}, k = {K}); single-GPU step measured here: "
          f"**{t1:.2f} ms** (3 hops).  Column chunks {chunks}.  Every rank's block, compact table, SpMM and pack kernel are built and timed on this GPU; "
          "the exchange enters as a link rate B per direction per link (xGMI: 76.8 GB/s peak).\n")
    print("| G | rank | own rows | ghost rows | skipped | spmm whole (compact / full replica) ms | spmm per chunk ms | pack per chunk ms | in-bound MB (need-aware / full) | busiest link in / out MB |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    model = {}
    for G in (2, 4, 8):
        bounds = balanced_bounds(rp_host, G)
        per_rank = [rank_measurements(rowptr, col, val, rp_host, bounds, r, n, d, x0, chunks, device) for r in range(G)]
        for m in per_rank:
            print(f"| {G} | {m['rank']} | {m['own']} | {m['ghost']} | {m['skipped']:.3f} | {m['spmm_whole']:.3f} / {m['spmm_whole_full_replica']:.3f} | "
                  f"{' + '.join(f'{t:.3f}' for t in m['spmm'])} | {' + '.join(f'{t:.3f}' for t in m['pack'])} | "
                  f"{m['in_total'] / 1e6:.0f} / {m['full_in_total'] / 1e6:.0f} | {m['in_peer_max'] / 1e6:.1f} / {m['out_peer_max'] / 1e6:.1f} |")
        model[G] = per_rank
    print()
    print("## Predicted ms per step (k = 3) and speed-up over the single-GPU step\n")
    print("Schedule simulated: compute stream = spmm(chunk A), pack(A), spmm(B), pack(B), ...; one grouped exchange per chunk and hop whose duration is "
          "the busiest link's bytes / B; hop h+1 of a chunk waits only for that chunk's exchange; the slowest rank decides.  "
          "`halo` = need-aware packed exchange, `full` = every row to every rank (no pack kernel).\n")
    print("| G | exchange | " + " | ".join(f"B = {b} GB/s" for b in rates) + " | no exchange at all |")
    print("|---|---|" + "---|" * (len(rates) + 1))

    def step_ms(G, B, kind):
        worst = 0.0
        for m in model[G]:
            if kind == "halo":
                link = max(m["in_peer_max"], m["out_peer_max"])
                xfer = [link * (b - a) / d / (B * 1e9) * 1e3 for a, b in chunks]
                t = simulate(m["spmm"], m["pack"], xfer)
            else:
                link = m["full_peer_max"]
                xfer = [link * (b - a) / d / (B * 1e9) * 1e3 for a, b in chunks]
                t = simulate(m["spmm_full_replica"], [0.0] * len(chunks), xfer)
            worst = max(worst, t)
        return worst

    for G in (2, 4, 8):
        for kind in ("halo", "full"):
            cells = []
            for B in rates:
                t = step_ms(G, B, kind)
                cells.append(f"{t:.2f} ms ({t1 / t:.2f}x)")
            t_inf = step_ms(G, 1e9, kind)
            print(f"| {G} | {kind} | " + " | ".join(cells) + f" | {t_inf:.2f} ms ({t1 / t_inf:.2f}x) |")
    print()
    print("### Pipelining granularity: the need-aware exchange with 1 / 2 / 3 / 4 column chunks\n")
    print("More chunks shorten the un-overlapped head (first chunk's SpMM + pack) and tail (last chunk's last SpMM) of the step; every cut keeps "
          "whole 128-byte lines per chunk row (d = 100 -> 64 + 36, 32 + 32 + 36, 32 + 32 + 32 + 4: four lines per gathered row either way), so the "
          "SpMM total moves little.\n")
    print("| G | chunks | spmm per chunk ms (slowest rank) | " + " | ".join(f"B = {b} GB/s" for b in rates) + " |")
    print("|---|---|---|" + "---|" * len(rates))
    for G in (2, 4, 8):
        for nc in ALT_CHUNKS:
            chs = column_chunks(d, nc)
            cells, slow = [], None
            for B in rates:
                worst = 0.0
                for m in model[G]:
                    link = max(m["in_peer_max"], m["out_peer_max"])
                    xfer = [link * (b - a) / d / (B * 1e9) * 1e3 for a, b in chs]
                    t = simulate(m["alt"][nc][0], m["alt"][nc][1], xfer)
                    if t > worst:
                        worst, slow = t, m
                cells.append(f"{worst:.2f} ms ({t1 / worst:.2f}x)")
            print(f"| {G} | {len(chs)} {chs} | {' + '.join(f'{v:.3f}' for v in slow['alt'][nc][0])} | " + " | ".join(cells) + " |")
    print()
    for kind in ("halo", "full"):
        lo_b, hi_b = 1.0, 2000.0
        if t1 / step_ms(8, hi_b, kind) < 5.0:
            print(f"* `{kind}`: 5x at 8 GPUs is out of reach at any link rate with this schedule ({t1 / step_ms(8, hi_b, kind):.2f}x with free links).")
            continue
        for _ in range(60):
            mid = 0.5 * (lo_b + hi_b)
            if t1 / step_ms(8, mid, kind) >= 5.0:
                hi_b = mid
            else:
                lo_b = mid
        print(f"* `{kind}`: the 8-GPU job reaches **5x** ({t1 / 5:.2f} ms per step) when every link delivers **{hi_b:.1f} GB/s per direction** "
              f"({hi_b / 76.8:.0%} of the xGMI peak).")
    del full, x0, rowptr, col, val
    torch.cuda.empty_cache()

    if a.papers:
        papers(device)


def papers(device):
    """rank 0 of the 8-rank papers100M-shaped job: compact table instead of the 57 GB replica"""
    import ctypes
    from sgl_amd import _lib
    from sgl_amd.dist import balanced_bounds_device
    wl = synthetic.WORKLOADS["S3_papers"]
    n, d, G, seed = wl["n"], wl["d"], 8, 0
    tab_h = synthetic.degree_table(wl["mean_deg"], wl["d_max"])
    tab = torch.from_numpy(tab_h).to(device)
    deg = torch.empty(n, dtype=torch.int64, device=device)
    _lib.check_probe(_lib.probe_lib().sgl_synth_degrees(ctypes.c_uint64(seed), 0, n, _lib.ptr(tab), _lib.ptr(deg), _lib.current_stream_ptr()))
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    del deg
    bounds = balanced_bounds_device(rowptr, G)
    nnz = int(rowptr[-1])
    del rowptr

    def block(q):
        return synthetic.hashed_block_torch(seed, int(bounds[q]), int(bounds[q + 1] - bounds[q]), n, tab_h, device=device)

    plan = HaloPlan.offline(0, bounds, n, lambda q: block(q)[1])
    rp, col, val = block(0)
    csr = dev.DeviceCSR(rp, plan.relabel(col), val, (plan.n_own, plan.n_compact))
    del col
    chunks = column_chunks(d, 2)
    print(f"\n## papers100M-shaped job, rank 0 of 8 (N = {n}, nnz = {nnz}, d = {d})\n")
    print(f"own rows {plan.n_own}, ghost rows {plan.n_ghost} of {plan.rows_in_full} foreign rows (skipped {plan.skipped_fraction:.3f}); compact table "
          f"{plan.n_compact * d * 4 / 1e9:.1f} GB instead of the {n * d * 4 / 1e9:.1f} GB replica; rows sent {plan.send_off[-1]} "
          f"({plan.send_off[-1] * d * 4 / 1e9:.1f} GB packed per hop); in-bound {plan.n_ghost * d * 4 / 1e9:.1f} GB per hop instead of "
          f"{plan.rows_in_full * d * 4 / 1e9:.1f} GB; busiest link in {max(int(t.numel()) for t in plan.need) * d * 4 / 1e9:.2f} GB, out "
          f"{max(int(t.numel()) for t in plan.send_rows) * d * 4 / 1e9:.2f} GB.\n")
    spmm, pack = [], []
    for a, b in chunks:
        t = torch.empty((plan.n_compact, b - a), device=device).uniform_(-1, 1)
        y = torch.empty((plan.n_own, b - a), device=device)
        spmm.append(timed(lambda: csr.spmm(t, out=y), reps=3, warm=1))
        buf = torch.empty((int(plan.send_off[-1]), b - a), device=device)
        pack.append(min(timed(lambda: dev.scatter_rows(y, plan.pack_src, plan.pack_dst, buf), reps=3, warm=1),
                        timed(lambda: dev.gather_rows(y, plan.send_idx, out=buf), reps=3, warm=1)))
        del t, y, buf
    t = torch.empty((plan.n_compact, d), device=device).uniform_(-1, 1)
    y = torch.empty((plan.n_own, d), device=device)
    whole = timed(lambda: csr.spmm(t, out=y), reps=3, warm=1)
    del t, y
    print(f"measured: SpMM whole width {whole:.2f} ms (the same block against the full 57 GB replica: 37-43 ms, profiles/r02_*), per chunk "
          f"{' + '.join(f'{v:.2f}' for v in spmm)} ms, pack per chunk {' + '.join(f'{v:.2f}' for v in pack)} ms.\n")
    link = max(max(int(t_.numel()) for t_ in plan.need), max(int(t_.numel()) for t_ in plan.send_rows)) * d * 4
    full_link = max(int(bounds[q + 1] - bounds[q]) for q in range(1, G)) * d * 4
    t1 = 3 * 320.0
    print("| exchange | " + " | ".join(f"B = {b} GB/s" for b in (25, 35, 45, 55, 65, 76.8)) + " |")
    print("|---|" + "---|" * 6)
    for kind in ("halo", "full"):
        cells = []
        for B in (25, 35, 45, 55, 65, 76.8):
            lk = link if kind == "halo" else full_link
            xfer = [lk * (b - a) / d / (B * 1e9) * 1e3 for a, b in chunks]
            ts = simulate(spmm, pack if kind == "halo" else [0.0, 0.0], xfer)
            cells.append(f"{ts:.0f} ms ({t1 / ts:.2f}x)")
        print(f"| {kind} | " + " | ".join(cells) + " |")
    print(f"\n(speed-up against 3 x 320 ms = the whole graph on one GPU, `profiles/r02_bench_S1.json`; rank 0 only, SpMM times of the compact "
          "table used for both rows -- the full-replica SpMM is 5-15 % slower.)")


if __name__ == "__main__":
    main()
