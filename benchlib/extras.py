"""The BASELINE configs the headline does not cover, as secondary sections of the ONE bench line (single GPU, after the timed
region; never part of `value`).  BASELINE.json lists five configs; the headline is config 2 (S1), `papers100M` is the SpMM of
configs 4/5 on the whole graph.  The sections here add, each with its own workload string, time, roofline and validation:

  S0_pubmed        config 1: SGC prop_steps=3 on the Pubmed-sized graph -- which BASELINE quotes ON THE REFERENCE CPU PATH, so the
                   section times that path too (scipy normalisation + the reference's C kernel) and compares the GPU result with it
  S2_gamlp         config 3: d = 147, k = 5, the three `preprocess` calls of one label-reuse epoch
                   (tasks/node_classification_with_label_use.py:79,104) + the learnable aggregate's training feed
  S4_products      config 5 on the graph that fits one GPU: Laplacian + PPR alpha in {.1,.2,.3}, k = 10, normalise / propagate /
                   aggregate per MessageOp of the search space (search/search_models.py:19-46) = PaSca's time objective
                   (search_config.py:42-48), + the torch-CPU `_combine` of mean / max / concat / NAFS on a row sample
  S4_papers_shard  configs 4/5 as one rank of the 8-GPU job sees them: its row block of the 111 M-node graph normalised per block,
                   k = 10 hops against the full replica, NAFS + the same MessageOps over the 11 hop shards
  S1_community     the S1 degree law WITH communities and shuffled ids (what the real co-purchase graph looks like): the hop with
                   reorder=None and reorder="auto" -- bit-identical outputs, the locality ordering's effect

Every aggregate is checked on sampled rows against the reference's formula evaluated in float64 with plain torch indexing (a
check that shares no code with the kernels); every propagation by sampled rows recomputed in float64."""
import argparse
import os
import time

import numpy as np
import torch

from .common import HBM_PEAK_BYTES, algorithmic_bytes_per_hop, workload_text

TOL = 1e-5
# which synthetic workloads the sections run on: "full" = the BASELINE shapes; "small" = the same code on graphs a test finishes in seconds
SCALES = {"full": {"S1": "S1_products", "S2": "S2_gamlp", "S3": "S3_papers", "COMM": "S1_community"},
          "small": {"S1": "S1_small", "S2": "S2_small", "S3": "S3_small", "COMM": "T_community"}}
REQUIRED_KEYS = ("workload", "ms", "roofline", "validated", "wall_s")     # of every section that ran (tests/test_bench_orchestration.py)


def _names(args):
    return SCALES[getattr(args, "extras_scale", "full") or "full"]


def _ms(fn, reps=3, warm=1):
    """median HIP-event time of fn() on the current stream (the stream the library launches on) -> (ms, last result)"""
    r = None
    for _ in range(warm):
        r = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        r = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), r


def _close(a, b, tol=TOL):
    """SURVEY 8(c)'s three-way criterion between two device matrices of one shape (b is the reference)"""
    if a.shape != b.shape:
        return False
    if a.numel() == 0 or torch.equal(a, b):
        return True
    diff = (a.double() - b.double())
    ref = b.double()
    mx = float(ref.abs().max())
    ok1 = float(diff.abs().max()) <= tol * mx
    rn = ref.norm(dim=1)
    ok2 = bool(((diff.norm(dim=1) <= tol * rn) | (rn == 0)).all())
    ok3 = bool((diff.abs() <= tol * mx + tol * ref.abs()).all())
    return bool(ok1 and ok2 and ok3)


def _roof(bytes_, ms):
    a = bytes_ / (ms * 1e-3)
    return {"bound": "hbm", "achieved": a / 1e9, "peak": HBM_PEAK_BYTES / 1e9, "unit": "GB/s", "frac": a / HBM_PEAK_BYTES,
            "algorithmic_bytes_per_launch": int(bytes_)}


def _free_all():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _with_workload(args, name):
    ns = argparse.Namespace(**vars(args))
    ns.workload = name
    return ns


# ---- float64 statements of the MessageOps on sampled rows (reference: sgl/operators/message_op/*.py) ---------------------------
def _f64_aggregate(kind, rows64, params):
    """rows64: list of H float64 [m, d] tensors (the sampled rows of the hop matrices the op was given, i.e. ALL hops);
    returns the float64 [m, *] aggregate of the reference's formula"""
    s, e = params.get("start", 0), params.get("end", len(rows64))
    hops = rows64[s:e]
    if kind == "last":
        return rows64[-1]
    if kind == "concat":
        return torch.cat(hops, 1)
    if kind in ("sum", "mean"):
        acc = hops[0].clone()
        for h in hops[1:]:
            acc = acc + h
        return acc / (e - s) if kind == "mean" else acc
    if kind in ("max", "min"):
        st = torch.stack(hops, 0)
        return st.max(0)[0] if kind == "max" else st.min(0)[0]
    if kind == "simple_weighted":
        a = params["alpha"]
        w = [a]
        for _ in range(len(rows64) - 1):
            w.append((1 - a) * w[-1])
        w = [float(np.float32(v)) for v in w[s:e]]                     # the reference rounds the weights to float32
        return sum(wi * h for wi, h in zip(w, hops))
    if kind == "learnable_simple":
        w = torch.softmax(torch.sigmoid(params["p"].double()[s:e]), 0)
        return sum(w[i] * h for i, h in enumerate(hops))
    if kind == "learnable_gate":
        v, b = params["weight"].double().view(-1), params["bias"].double()
        sc = torch.stack([h @ v + b for h in hops], 1)                 # [m, H]: score of (node, hop) -- .view(H, -1).T
        w = torch.softmax(torch.sigmoid(sc), 1)
        return sum(w[:, i:i + 1] * h for i, h in enumerate(hops))
    if kind == "nafs":
        x0 = rows64[0]
        n0 = x0.norm(dim=1) + 1e-10
        cs = torch.stack([((x0 * h).sum(1) / (h.norm(dim=1) + 1e-10)) / n0 for h in rows64], 1)
        w = torch.softmax(cs, 1)
        return sum(w[:, i:i + 1] * h for i, h in enumerate(rows64))
    raise KeyError(kind)


def _check_aggregate(kind, hops, out, params, rows):
    """the aggregate's sampled rows against the float64 formula: SURVEY 8(c) criterion (row-wise L2 and max-norm, 1e-5)"""
    rows64 = [h[rows].double() for h in hops]
    want = _f64_aggregate(kind, rows64, params)
    got = out[rows].double()
    if want.shape != got.shape:
        return {"ok": False, "why": f"shape {tuple(got.shape)} != {tuple(want.shape)}"}
    den = want.norm(dim=1).clamp_min(1e-30)
    row_rel = float(((got - want).norm(dim=1) / den).max())
    abs_rel = float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))
    return {"ok": bool(row_rel <= TOL and abs_rel <= TOL), "max_row_rel_l2": row_rel, "max_abs_rel": abs_rel}


def _search_space_ops(K, d, device, with_nafs=True):
    """the nine MessageOps of the PaSca search space (search_models.py:28-46; mesg_types 7 / 8 start at hop 1 and are handed feat_dim
    where prop_steps is meant -- reproduced) + the NAFS weighting of config 4"""
    from sgl_amd.operators import message_op as M
    torch.manual_seed(0)
    gate = M.LearnableWeightedMessageOp(1, K + 1, "gate", d).to(device)
    simple = M.LearnableWeightedMessageOp(1, K + 1, "simple", d).to(device)
    gl = [p for p in gate.parameters()]
    ops = [("last", "last", M.LastMessageOp(), {}),
           ("concat", "concat", M.ConcatMessageOp(0, K + 1), {"start": 0, "end": K + 1}),
           ("mean", "mean", M.MeanMessageOp(0, K + 1), {"start": 0, "end": K + 1}),
           ("sum", "sum", M.SumMessageOp(0, K + 1), {"start": 0, "end": K + 1}),
           ("max", "max", M.MaxMessageOp(0, K + 1), {"start": 0, "end": K + 1}),
           ("min", "min", M.MinMessageOp(0, K + 1), {"start": 0, "end": K + 1}),
           ("simple_weighted a=.85", "simple_weighted", M.SimpleWeightedMessageOp(0, K + 1, "alpha", 0.85), {"start": 0, "end": K + 1, "alpha": 0.85}),
           ("learnable gate (hops 1..K)", "learnable_gate", gate, {"start": 1, "end": K + 1, "weight": gl[0].detach(), "bias": gl[1].detach()}),
           ("learnable simple (hops 1..K)", "learnable_simple", simple, {"start": 1, "end": K + 1, "p": next(simple.parameters()).detach()})]
    if with_nafs:
        ops.append(("nafs over_smooth_distance", "nafs", M.OverSmoothDistanceWeightedOp(), {}))
    return ops


def _aggregate_sweep(hops, n_rows, d, K, device, reps=3):
    """every MessageOp over the K + 1 hop matrices: ms, fraction of the HBM peak on the aggregators' byte model
    ((hops read + output written) x n x d x 4), validated on sampled rows"""
    g = torch.Generator(device="cpu").manual_seed(99)
    rows = torch.randint(0, n_rows, (min(2048, n_rows),), generator=g).to(device)
    table, all_ok = [], True
    for name, kind, op, params in _search_space_ops(K, d, device):
        with torch.no_grad():
            ms, out = _ms(lambda: op.aggregate(hops), reps=reps)
        h_in = len(hops[params.get("start", 0):params.get("end", K + 1)]) if kind != "last" else 0
        by = (h_in * n_rows * d + out.numel()) * 4 if kind != "last" else 0
        chk = _check_aggregate(kind, hops, out, params, rows)
        all_ok = all_ok and chk["ok"]
        row = {"msg_op": name, "ms": ms, "validated": chk["ok"], "max_row_rel_l2": chk.get("max_row_rel_l2")}
        if by:
            row["roofline"] = _roof(by, ms)
        else:
            row["note"] = "returns the last hop matrix itself: no kernel, no bytes"
        table.append(row)
        del out
    return table, all_ok


def _sampled_spmm_ok(engine, rowptr, col, val, n_cols, x_prev, y, lo=0, d=None):
    from sgl_amd.dist import RowBlock
    blk = RowBlock(lo, lo + rowptr.numel() - 1, n_cols, rowptr, col, val)
    d = d or y.shape[1]
    return bool(engine.sampled_rows_check(blk, x_prev[:, :d], y[:, :d], samples=2048, tol=TOL))


def _host_cores():
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    return len(cpus)


# ---- config 1 ------------------------------------------------------------------------------------------------------------------
def section_s0(args, engine):
    """SGC prop_steps=3 on the Pubmed-sized graph: the k launches replayed as one hipGraph (the S0 workload of `--workload
    S0_pubmed`), the whole result compared with the CPU oracle's propagate() of the same raw graph -- normalisation included,
    strict order bit for bit -- and the reference CPU path (what BASELINE config 1 is quoted on) timed on this host."""
    from sgl_amd import synthetic
    from sgl_amd.io import DeviceAdjacency
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    wl = synthetic.WORKLOADS["S0_pubmed"]
    n, d, K = wl["n"], wl["d"], wl["k"]
    a0 = _with_workload(args, "S0_pubmed")
    a_ptr, a_col, a_val = engine.build_raw(a0, wl)
    x0 = engine.features(a0, wl)
    rowptr, col, val, _ = engine.build_workload(a0, wl)                 # A_hat normalised on the device (same seed: same graph)
    step, info = engine.single_step(a0, rowptr, col, val, x0, n, d, K)
    nnz = int(col.numel())
    reps = 200
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms_step = e0.elapsed_time(e1) / reps
    out = {"workload": workload_text("S0_pubmed", K), "baseline_config": 1, "n_nodes": n, "nnz_a_hat": nnz, "feat_dim": d, "prop_steps": K,
           "ms": ms_step, "ms_per_hop": ms_step / K, "value": nnz * d * K / (ms_step * 1e-3), "unit": "edge·featdim/s",
           "roofline": _roof(algorithmic_bytes_per_hop(n, nnz, d), ms_step / K),
           "how": f"{reps} replays of the hipGraph holding the {K} launches (+ fix-ups), HIP events", "plan": info,
           "short": f"SGC k={K}, Pubmed-sized graph N={n} d={d}, hipGraph replay", "ms_is_short": f"{K} hops"}
    y_fast = engine._single["y_last"][:, :d].cpu().numpy()
    engine._single = None
    # the plugin API on the same raw graph, strict order
    adj = DeviceAdjacency(a_ptr, a_col, a_val, (n, n))
    hops_strict = [h.cpu().numpy() for h in LaplacianGraphOp(K, r=0.5, strict_order=True).propagate(adj, x0)]
    t0 = time.perf_counter()
    LaplacianGraphOp(K, r=0.5).propagate(adj, x0)
    torch.cuda.synchronize()
    out["propagate_call_ms"] = (time.perf_counter() - t0) * 1e3           # a fresh operator: normalise (prepared graph shared) + plan + k hops
    # ---- the reference CPU path of config 1, on this host: checker and baseline at once (benchlib/cpu_legs.py, the cpu_baseline leg)
    from .cpu_legs import config1_reference_path
    cpu, check = config1_reference_path(a_ptr.cpu().numpy(), a_col.cpu().numpy(), a_val.cpu().numpy(), x0.cpu().numpy(), n, K, y_fast, hops_strict, TOL)
    out["validated"] = bool(check["fast_order"]["ok"] and check["strict_order_all_hops_bit_equal"])
    out["validation"] = check
    out["cpu_baseline"] = cpu
    return out


# ---- config 3 ------------------------------------------------------------------------------------------------------------------
def _label_columns(n, c, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.softmax(torch.randn((n, c), generator=g, device=device), 1)


def section_s2(args, engine, raw):
    """GAMLP + label reuse at the products shape (examples/gamlp_products.py; tasks/node_classification_with_label_use.py:58-137):
    one epoch calls model.preprocess(adj, features) 1 + label_iters = 3 times, the last 47 of the 147 feature columns rewritten
    in between (:103).  Timed through the plugin API (LaplacianGraphOp(5).propagate on a device adjacency); the per-hop kernel time
    at d = 147 gives the roofline; then the learnable aggregate's training feed (gather |train| rows of the 6 hops + `jk`)."""
    from sgl_amd import device as dev, synthetic
    from sgl_amd.io import DeviceAdjacency
    from sgl_amd.operators import message_op as M
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    wl = synthetic.WORKLOADS[_names(args)["S2"]]
    n, d, K, C = wl["n"], wl["d"], wl["k"], 47
    device = engine.device
    a_ptr, a_col, a_val = raw
    adj = DeviceAdjacency(a_ptr, a_col, a_val, (n, n))
    feats = dev.alloc_rows(n, d, device)
    g = torch.Generator(device=device).manual_seed(args.seed + 5)
    feats[:, :d - C] = torch.randn((n, d - C), generator=g, device=device)
    feats[:, d - C:] = _label_columns(n, C, 1, device)
    from sgl_amd import config as sgl_config
    gop = LaplacianGraphOp(K, r=0.5)
    calls, cols = [], []
    hops = None
    keep_min = sgl_config.delta_propagate_min_mb
    if getattr(args, "extras_scale", "full") == "small":
        sgl_config.delta_propagate_min_mb = 0.0         # the test-sized graph is below the size from which the column delta is used
    for it in range(3):
        if it:
            feats[:, d - C:] = _label_columns(n, C, 1 + it, device)      # features[unlabeled, -C:] = softmax(pred)  (:103)
        held = hops                   # the model still holds its previous hop list while preprocess() runs (base_model.py:25-33)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hops = gop.propagate(adj, feats)
        torch.cuda.synchronize()
        calls.append((time.perf_counter() - t0) * 1e3)
        di = getattr(gop, "delta_info", None)
        cols.append(d if di is None else int(di["columns_propagated"][1] - di["columns_propagated"][0]))
        del held
    # the same third call with every column propagated again (config.delta_propagate off): what the calls cost without the column delta
    keep_delta = sgl_config.delta_propagate
    sgl_config.delta_propagate = False
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        full_hops = gop.propagate(adj, feats)
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t0) * 1e3
    finally:
        sgl_config.delta_propagate = keep_delta
        sgl_config.delta_propagate_min_mb = keep_min
    delta_ok = all(_close(a, b) for a, b in zip(hops, full_hops))
    del full_hops
    csr = gop._adj
    nnz = csr.nnz
    src = dev.padded_parent(feats)
    outs = [dev.padded_parent(h) for h in hops[1:]]
    ms_chain, _ = _ms(lambda: csr.spmm_chain(src, K, outs=outs), reps=3)
    ok = _sampled_spmm_ok(engine, csr.rowptr, csr.col, csr.val, n, outs[K - 2], outs[K - 1], d=d)
    out = {"workload": workload_text(_names(args)["S2"], K), "baseline_config": 3, "n_nodes": n, "nnz_a_hat": nnz, "feat_dim": d, "prop_steps": K,
           "preprocess_calls_ms": calls, "ms": calls[1] + calls[2] + calls[0],
           "columns_propagated_per_call": cols, "full_recompute_call_ms": full_ms, "delta_equals_full": delta_ok,
           "note": "call 1 normalises A on the device and builds the plan; calls 2-3 find both cached (the reference redoes the scipy "
                   "normalisation in every call, base_op.py:20) and re-propagate only the column range whose content changed "
                   "(config.delta_propagate: the product is separable by columns; the other columns of the new hop matrices are copied "
                   "from the previous ones); full_recompute_call_ms = the same call with all columns propagated",
           "ms_per_hop": ms_chain / K, "value": nnz * d * K / (ms_chain * 1e-3), "unit": "edge·featdim/s",
           "roofline": _roof(algorithmic_bytes_per_hop(n, nnz, d), ms_chain / K),
           "row_pitch_floats": int(src.stride(0)), "validated": bool(ok and delta_ok),
           "short": f"GAMLP label reuse, products shape N={n} d={d} k={K}, 3 preprocess calls", "ms_is_short": "the 3 preprocess calls of one epoch",
           "validation": {"how": "2048 rows of hop K recomputed in float64 from hop K-1", "ok": ok}}
    # the training feed of the learnable aggregate: |train| = 196 615 rows (8 %) of the 6 hop matrices, then `jk`
    idx = torch.randperm(n, generator=g, device=device)[:max(1, int(n * 0.0803))]
    torch.manual_seed(0)
    jk = M.LearnableWeightedMessageOp(0, K + 1, "jk", K, d).to(device)

    def feed():
        with torch.no_grad():
            return jk.aggregate(dev.gather_hops(hops, idx))
    ms_feed, _ = _ms(feed, reps=5)
    out["train_feed"] = {"rows": int(idx.numel()), "hops": K + 1, "ms": ms_feed,
                         "what": "gather_hops (one index upload, H gathers) + LearnableWeightedMessageOp('jk').aggregate, no_grad"}
    return out


# ---- config 5 at the products shape ----------------------------------------------------------------------------------------------
def section_s4_products(args, engine, raw):
    from sgl_amd import device as dev, synthetic
    from sgl_amd.operators.graph_op import ppr_hops_from_laplacian
    wl = synthetic.WORKLOADS[_names(args)["S1"]]
    n, d, K = wl["n"], wl["d"], 10
    device = engine.device
    a_ptr, a_col, a_val = raw
    x0 = engine.features(_with_workload(args, _names(args)["S1"]), wl)
    t_prep, prep = _ms(lambda: dev.PreparedAdjacency(a_ptr, a_col, a_val, n), reps=1, warm=0)
    hops = [x0] + [dev.alloc_rows(n, d, device) for _ in range(K)]
    src = [dev.padded_parent(h) if h.stride(0) % 4 == 0 else h for h in hops]
    x_in = dev.upload_rows(x0, device) if dev.row_pitch(d) != d else x0
    hops[0] = x_in
    src[0] = dev.padded_parent(x_in)
    graph_ops, ok_all = [], True
    lap_hops_ms = None
    for name, r, alpha in (("LaplacianGraphOp r=0.5", 0.5, None), ("PprGraphOp r=0.5 alpha=0.1", 0.5, 0.1),
                           ("PprGraphOp r=0.5 alpha=0.2", 0.5, 0.2), ("PprGraphOp r=0.5 alpha=0.3", 0.5, 0.3)):
        t_norm, (rowptr, col, val) = _ms(lambda: prep.normalize(r, alpha), reps=2)
        csr = dev.DeviceCSR(rowptr, col, val, (n, n))
        nnz = csr.nnz
        t_prop, _ = _ms(lambda: csr.spmm_chain(src[0], K, outs=src[1:]), reps=2)
        ok = _sampled_spmm_ok(engine, rowptr, col, val, n, src[K - 1], src[K], d=d)
        ok_all = ok_all and ok
        row = {"graph_op": name, "normalise_ms": t_norm, "propagate_ms": t_prop, "ms_per_hop": t_prop / K,
               "value": nnz * d * K / (t_prop * 1e-3), "unit": "edge·featdim/s",
               "roofline": _roof(algorithmic_bytes_per_hop(n, nnz, d), t_prop / K), "validated": ok}
        if alpha is None:
            lap_hops_ms = t_prop
            table, ok_a = _aggregate_sweep(hops, n, d, K, device)
            ok_all = ok_all and ok_a
            row["message_ops"] = table
            for t in table:
                # PaSca's time objective for (this graph op, this MessageOp): preprocess = normalise + propagate + aggregate
                t["preprocess_objective_ms"] = t_norm + t_prop + t["ms"]
            from .cpu_legs import combine_baseline
            cpu = combine_baseline(hops, d)
            by_name = {t["msg_op"].split()[0]: t["ms"] for t in table}
            cpu["gpu_rows_per_s"] = {k: n / (by_name[k] * 1e-3) for k in ("mean", "max", "concat", "nafs")}
            row["cpu_baseline_combine"] = cpu
            lap = [h.clone() for h in hops]
        else:
            # the same hop matrices without propagating: a triangular mix of the Laplacian chain (sgl_hop_lincomb_f32)
            t_mix, mixed = _ms(lambda: ppr_hops_from_laplacian(lap, alpha), reps=2)
            err = max(float((m_ - f_).abs().max() / f_.abs().max()) for m_, f_ in zip(mixed, hops))
            row["mixed_from_laplacian_chain"] = {"ms": t_mix, "max_abs_rel_vs_propagated": err, "ok": bool(err <= TOL),
                                                 "roofline": _roof((2 * K + 1) * n * d * 4, t_mix)}
            ok_all = ok_all and err <= TOL
            del mixed
        graph_ops.append(row)
        del csr, rowptr, col, val
    return {"workload": f"S4_products ({_names(args)['S1']}): PaSca operator sweep (BASELINE config 5's search space, search/search_models.py:19-46) on the "
                        f"ogbn-products-shaped Chung-Lu graph: {{Laplacian, PPR alpha in .1/.2/.3}} x k = {K} x nine MessageOps + NAFS",
            "baseline_config": 5, "n_nodes": n, "feat_dim": d, "prop_steps": K, "prepare_ms": t_prep,
            "short": f"PaSca op sweep, products shape N={n} d={d} k={K}", "ms_is_short": "4 x (normalise + 10 hops) + the 10 aggregates",
            "ms": sum(g["normalise_ms"] + g["propagate_ms"] for g in graph_ops) + sum(t["ms"] for t in graph_ops[0]["message_ops"]),
            "roofline": graph_ops[0]["roofline"], "graph_ops": graph_ops, "validated": bool(ok_all),
            "laplacian_chain_ms": lap_hops_ms}


# ---- configs 4/5 as one rank of the 8-GPU job --------------------------------------------------------------------------------------
def section_s4_papers_shard(args, engine, parts=8):
    """Rank 0's row block of the ogbn-papers100M-shaped hashed graph, canonicalised and normalised PER BLOCK as the row-sharded job
    does it (sgl_norm_block_*; the global degree vector is the sum of the blocks' column sums -- here the 8 blocks are generated one
    after the other on this GPU, in the job it is one all-reduce), then k = 10 hops against the full 111 M x 128 replica (the same
    replica every hop: one GPU cannot produce the other ranks' rows of the next; per-hop time does not depend on the values) and
    every MessageOp + the NAFS weighting over the 11 hop shards.  The per-hop exchange is not part of a single-GPU measurement."""
    import ctypes
    from sgl_amd import _lib, device as dev, synthetic as sy
    from sgl_amd.dist import RowBlock, canonicalize_block
    device = engine.device
    free, _ = torch.cuda.mem_get_info()
    wl = sy.WORKLOADS[_names(args)["S3"]]
    scale = 1 if (free > 200e9 or wl["n"] < 10_000_000) else 8
    n, d, K = wl["n"] // scale, wl["d"], 10
    table = sy.degree_table(wl["mean_deg"], wl["d_max"])
    bounds = [n * i // parts for i in range(parts + 1)]
    deg = torch.zeros(n, dtype=torch.float64, device=device)
    block0 = None
    t0 = time.perf_counter()
    for b in range(parts):
        lo, hi = bounds[b], bounds[b + 1]
        rp, c, v = sy.hashed_block_torch(args.seed, lo, hi - lo, n, table, device=device)
        blk = canonicalize_block(RowBlock(lo, hi, n, rp, c, v))
        del rp, c, v
        m = ctypes.c_int64(0)
        _lib.check(_lib.lib().sgl_norm_block_prepare(hi - lo, lo, blk.nnz, _lib.ptr(blk.rowptr), _lib.ptr(blk.col), ctypes.byref(m),
                                                     _lib.current_stream_ptr()))
        o_ptr = torch.empty(hi - lo + 1, dtype=torch.int64, device=device)
        o_col = torch.empty(m.value, dtype=torch.int32, device=device)
        t64 = torch.empty(m.value, dtype=torch.float64, device=device)
        rs = torch.empty(hi - lo, dtype=torch.float64, device=device)
        _lib.check(_lib.lib().sgl_norm_block_build(hi - lo, lo, blk.nnz, _lib.ptr(blk.rowptr), _lib.ptr(blk.col), _lib.ptr(blk.val), m.value,
                                                   _lib.ptr(o_ptr), _lib.ptr(o_col), _lib.ptr(t64), _lib.ptr(rs), _lib.current_stream_ptr()))
        _lib.check(_lib.lib().sgl_norm_block_colsum(n, m.value, _lib.ptr(o_col), _lib.ptr(t64), _lib.ptr(deg), _lib.current_stream_ptr()))
        torch.cuda.synchronize()
        if b == 0:
            block0 = blk
        del o_ptr, o_col, t64, rs, blk
    setup_s = time.perf_counter() - t0
    x = sy.hashed_features_torch(args.seed, 0, n, d, device=device)
    lo, hi = bounds[0], bounds[1]
    n_loc = hi - lo
    hops = [x[lo:hi]] + [dev.alloc_rows(n_loc, d, device) for _ in range(K)]
    t_prep, prep = _ms(lambda: dev.PreparedBlock(block0.rowptr, block0.col, block0.val, lo, n, symmetric=False, deg=deg), reps=1, warm=0)
    graph_ops, ok_all = [], True
    for name, r, alpha in (("LaplacianGraphOp r=0.5", 0.5, None), ("PprGraphOp r=0.5 alpha=0.1", 0.5, 0.1)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rowptr, col, val = prep.normalize(r, alpha, host_pow="auto")
        torch.cuda.synchronize()
        t_norm = (time.perf_counter() - t0) * 1e3
        csr = dev.DeviceCSR(rowptr, col, val, (n_loc, n))
        nnz = csr.nnz

        def prop():
            for h in range(1, K + 1):
                csr.spmm(x, out=hops[h])
        t_prop, _ = _ms(prop, reps=2)
        ok = _sampled_spmm_ok(engine, rowptr, col, val, n, x, hops[K], lo=lo, d=d)
        ok_all = ok_all and ok
        alg = nnz * d * 4 + nnz * 8 + (n_loc + 1) * 4 + n_loc * d * 4
        ceiling = engine.gather_ceiling(col, x, d) if alpha is None else None       # the bare pattern of THIS block on THIS table, measured now
        row = {"graph_op": name, "normalise_block_ms": t_norm, "propagate_ms": t_prop, "ms_per_hop": t_prop / K,
               "gather_ceiling_Ggathers_per_s": ceiling, "kernel_Ggathers_per_s": nnz / (t_prop / K * 1e-3) / 1e9,
               "frac_of_gather_ceiling": (nnz / (t_prop / K * 1e-3) / 1e9 / ceiling) if ceiling else None,
               "value": nnz * d * K / (t_prop * 1e-3), "unit": "edge·featdim/s per GPU", "nnz_block": nnz,
               "roofline": _roof(alg, t_prop / K), "validated": ok}
        if alpha is None:
            tab, ok_a = _aggregate_sweep(hops, n_loc, d, K, device, reps=2)
            ok_all = ok_all and ok_a
            row["message_ops"] = tab
            for t in tab:
                t["preprocess_objective_ms"] = t_norm + t_prop + t["ms"]
            from sgl_amd.operators.graph_op import ppr_hops_from_laplacian
            t_mix, mixed = _ms(lambda: ppr_hops_from_laplacian(hops, 0.1), reps=1)
            row["ppr_alpha_0.1_mixed_from_this_chain"] = {"ms": t_mix, "roofline": _roof((2 * K + 1) * n_loc * d * 4, t_mix)}
            del mixed
        graph_ops.append(row)
        del csr, rowptr, col, val
    nafs = next(t for t in graph_ops[0]["message_ops"] if t["msg_op"].startswith("nafs"))
    return {"workload": f"S4_papers_shard ({_names(args)['S3']}): NAFS (config 4) and the PaSca MessageOp sweep (config 5), k = {K}, on rank 0's 1/{parts} row block "
                        f"of the ogbn-papers100M-shaped hashed graph ({n_loc} rows, normalised per block) against the full {n} x {d} replica"
                        + ("" if scale == 1 else f" [scaled 1/{scale}: this GPU has {free / 1e9:.0f} GB free]"),
            "baseline_config": [4, 5], "n_nodes": n, "rows_block": n_loc, "feat_dim": d, "prop_steps": K,
            "setup_s": round(setup_s, 2), "prepare_block_ms": t_prep,
            "ms": graph_ops[0]["normalise_block_ms"] + graph_ops[0]["propagate_ms"] + nafs["ms"],
            "ms_is": "config 4 on this rank: normalise block + 10 hops + NAFS weighting (exchange excluded: single GPU)",
            "short": f"NAFS + PaSca ops k={K} on rank 0's 1/{parts} row block ({n_loc} rows) of the papers100M-shaped graph vs the full {n} x {d} replica",
            "ms_is_short": "NAFS on this rank: normalise block + 10 hops + weighting (no exchange)",
            "roofline": graph_ops[0]["roofline"], "graph_ops": graph_ops, "validated": bool(ok_all)}


# ---- the S1 degree law with communities ----------------------------------------------------------------------------------------------
def section_s1_community(args, engine):
    """The products-sized graph WITH communities and shuffled ids: one hop as the ids come (reorder=None) and with the plan-time
    locality ordering chosen by reorder="auto" (rows stored and processed community by community, ids untouched).  Outputs are
    bit-identical; the fraction is on the NO-REUSE byte model and may therefore exceed what that model allows."""
    from sgl_amd import device as dev, synthetic
    from sgl_amd.reorder import plan_rowmap
    name = _names(args)["COMM"]
    wl = synthetic.WORKLOADS[name]
    n, d, K = wl["n"], wl["d"], wl["k"]
    device = engine.device
    cm = wl["community"]
    a_ptr, a_col, a_val, _truth = synthetic.planted_partition_torch(n, wl["m"], wl["d_max"], cm["block"], cm["p_in"], seed=args.seed,
                                                                     device=device)
    del _truth
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    del a_ptr, a_col, a_val
    x0 = engine.features(_with_workload(args, _names(args)["S1"]), wl)
    x_in = dev.padded_parent(dev.upload_rows(x0, device) if dev.row_pitch(d) != d else x0)
    y0, y1 = dev.padded_parent(dev.alloc_rows(n, d, device)), dev.padded_parent(dev.alloc_rows(n, d, device))
    nnz = int(col.numel())
    alg = algorithmic_bytes_per_hop(n, nnz, d)
    plain = dev.DeviceCSR(rowptr, col, val, (n, n))
    ms_plain, _ = _ms(lambda: plain.spmm(x_in, out=y0), reps=5, warm=2)
    ok = _sampled_spmm_ok(engine, rowptr, col, val, n, x_in, y0, d=d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rowmap, info = plan_rowmap(rowptr, col, n, "auto")
    torch.cuda.synchronize()
    t_plan = (time.perf_counter() - t0) * 1e3
    out = {"workload": f"{name}: one hop of LaplacianGraphOp r=0.5 on a degree-corrected planted-partition graph with the S1 degree law "
                       f"(communities of {cm['block']} nodes, p_in={cm['p_in']}, node ids shuffled), d={d}",
           "n_nodes": n, "nnz_a_hat": nnz, "feat_dim": d,
           "reorder_none": {"ms_per_hop": ms_plain, "roofline": _roof(alg, ms_plain), "validated": ok},
           "reorder_auto": {"plan": info, "plan_ms": t_plan}, "ms": ms_plain,
           "short": f"one hop, products degree law WITH communities ({cm['block']} nodes, p_in={cm['p_in']}), ids shuffled, N={n} d={d}",
           "ms_is_short": "one hop with reorder=auto"}
    if rowmap is not None:
        rp2, c2, v2 = dev.permute_rows(rowptr, col, val, rowmap)
        ordered = dev.DeviceCSR(rp2, c2, v2, (n, n)).set_rowmap(rowmap)
        ms_ord, _ = _ms(lambda: ordered.spmm(x_in, out=y1), reps=5, warm=2)
        same = bool(torch.equal(y0[:, :d], y1[:, :d]))
        out["reorder_auto"].update({"ms_per_hop": ms_ord, "roofline": _roof(alg, ms_ord), "bit_identical_to_reorder_none": same,
                                    "note": "fraction on the no-reuse byte model: > 0.76 means the gathers were served from cache"})
        out["ms"] = ms_ord
        ok = ok and same
    out["validated"] = bool(ok)
    out["roofline"] = out["reorder_auto"].get("roofline", out["reorder_none"]["roofline"])
    # fabric-side bytes per launch from the builder's separate rocprofv3 --pmc passes over `bench.py --workload S1_community
    # [--reorder auto]` (profiles/traffic.json): REPLAYED, not measured in this run -- against the algorithmic bytes of the no-reuse model
    from .common import _replayed_profile
    for key, tag in (("reorder_none", "none"), ("reorder_auto", "auto")):
        prof = _replayed_profile(f"{name}_{tag}", 1)
        if prof and prof.get("hbm_bytes_per_launch"):
            out[key]["traffic_replayed"] = {"bytes_per_launch": prof["hbm_bytes_per_launch"], "over_algorithmic": prof["hbm_bytes_per_launch"] / alg,
                                            "source": prof.get("source")}
    return out


# ---- what goes into the ONE line ----------------------------------------------------------------------------------------------------
def _r(v, sig=4):
    """4 significant digits: the line stays short enough to be read whole"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    try:
        return float(f"{float(v):.{sig}g}")
    except (TypeError, ValueError):
        return v


def _roof_c(r):
    return {"frac": _r(r["frac"]), "achieved": _r(r["achieved"]), "unit": r["unit"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"]}


def _ops_c(table):
    """[msg_op, ms, frac of the HBM peak on the aggregators' byte model, validated]"""
    return [[t["msg_op"], _r(t["ms"]), _r(t["roofline"]["frac"]) if "roofline" in t else None, t["validated"]] for t in table]


def compact_sections(sections):
    """the sections as they appear in the bench line: workload, ms, roofline, validated + the tables as short rows.  The full
    dictionaries (every field run_extras produced) go to --detail-out."""
    out = {}
    for name, s in sections.items():
        if "failed" in s or "skipped" in s:
            out[name] = {k: s[k] for k in ("failed", "where", "skipped", "wall_s") if k in s}
            continue
        c = {"workload": s["workload"].split(":")[0] + ": " + s.get("short", ""), "baseline_config": s.get("baseline_config"),
             "ms": _r(s["ms"]), "ms_is": s.get("ms_is_short"), "roofline": _roof_c(s["roofline"]), "validated": s["validated"],
             "wall_s": s.get("wall_s")}
        if name == "S0_pubmed":
            cb = s["cpu_baseline"]
            c.update({"ms_per_hop": _r(s["ms_per_hop"]), "value": _r(s["value"]), "strict_order_bit_equal_to_cpu_oracle": s["validation"]["strict_order_all_hops_bit_equal"],
                      "fast_order_row_l2_rel": _r(s["validation"]["fast_order"]["row_l2_rel"]),
                      "cpu_baseline": {"kind": cb["kind"], "cores": cb["cores"], "normalise_ms": _r(cb["normalise_ms"]),
                                       "propagate_ms": _r(cb["propagate_ms"]), "value": _r(cb["value"])}})
        elif name == "S2_gamlp":
            c.update({"preprocess_calls_ms": [_r(v) for v in s["preprocess_calls_ms"]], "columns_propagated_per_call": s.get("columns_propagated_per_call"),
                      "full_recompute_call_ms": _r(s.get("full_recompute_call_ms")), "ms_per_hop": _r(s["ms_per_hop"]), "value": _r(s["value"]),
                      "train_feed_ms": _r(s["train_feed"]["ms"]), "train_feed_rows": s["train_feed"]["rows"]})
        elif name in ("S4_products", "S4_papers_shard"):
            nk = "normalise_ms" if name == "S4_products" else "normalise_block_ms"
            c["graph_ops_cols"] = ["graph_op", nk, "propagate_k10_ms", "frac", "validated", "ppr_mixed_from_laplacian_chain_ms"]
            c["graph_ops"] = [[g["graph_op"], _r(g[nk]), _r(g["propagate_ms"]), _r(g["roofline"]["frac"]), g["validated"],
                               _r((g.get("mixed_from_laplacian_chain") or g.get("ppr_alpha_0.1_mixed_from_this_chain") or {}).get("ms"))]
                              for g in s["graph_ops"]]
            c["message_ops_cols"] = ["msg_op", "ms", "frac", "validated"]
            c["message_ops"] = _ops_c(s["graph_ops"][0]["message_ops"])
            cb = s["graph_ops"][0].get("cpu_baseline_combine")
            if cb:
                c["cpu_baseline_combine"] = {"kind": cb["kind"], "what": "the reference's torch-CPU _combine on the first rows of the same hops",
                                             "threads": cb["threads"], "rows": cb["rows"], "nafs_rows": cb["nafs_rows"],
                                             "ms": {k: _r(cb[k]["ms"]) for k in ("mean", "max", "concat", "nafs")},
                                             "rows_per_s": {k: _r(cb[k]["rows_per_s"]) for k in ("mean", "max", "concat", "nafs")},
                                             "gpu_rows_per_s": {k: _r(v) for k, v in cb["gpu_rows_per_s"].items()}}
            if name == "S4_papers_shard":
                c["rows_block"] = s["rows_block"]
                c["frac_of_gather_ceiling"] = _r(s["graph_ops"][0].get("frac_of_gather_ceiling"))
        elif name == "S1_community":
            ra = s["reorder_auto"]
            c.update({"reorder_none": {"ms_per_hop": _r(s["reorder_none"]["ms_per_hop"]), "frac": _r(s["reorder_none"]["roofline"]["frac"])},
                      "reorder_auto": {"ms_per_hop": _r(ra.get("ms_per_hop")), "frac": _r(ra["roofline"]["frac"]) if "roofline" in ra else None,
                                       "applied": ra["plan"].get("applied"), "edge_locality": [ra["plan"].get("edge_locality_before"), ra["plan"].get("edge_locality_after")],
                                       "plan_ms": _r(ra["plan_ms"]), "bit_identical": ra.get("bit_identical_to_reorder_none")},
                      "frac_model": "no-reuse bytes: may exceed 1 when gathers hit in cache",
                      "traffic_over_algorithmic_replayed": [_r((s["reorder_none"].get("traffic_replayed") or {}).get("over_algorithmic")),
                                                            _r((ra.get("traffic_replayed") or {}).get("over_algorithmic"))]})
        out[name] = {k: v for k, v in c.items() if v is not None}
    return out


# ---- driver -------------------------------------------------------------------------------------------------------------------------
def run_extras(args, engine, out, budget_s=240.0, which=None):
    """Run the sections into out["sections"][name]; every section is independent (its own inputs, freed afterwards), a failure or
    an exhausted time budget is recorded in place of its numbers and never touches the headline."""
    from sgl_amd import synthetic
    t_all = time.perf_counter()
    sections = out.setdefault("sections", {})
    raw_box = [None]

    def raw():
        if raw_box[0] is None:
            raw_box[0] = engine.build_raw(_with_workload(args, _names(args)["S1"]), synthetic.WORKLOADS[_names(args)["S1"]])
        return raw_box[0]

    plan = [("S0_pubmed", lambda: section_s0(args, engine)),
            ("S2_gamlp", lambda: section_s2(args, engine, raw())),
            ("S4_products", lambda: section_s4_products(args, engine, raw())),
            ("S1_community", lambda: section_s1_community(args, engine)),
            ("S4_papers_shard", lambda: section_s4_papers_shard(args, engine))]
    for name, fn in plan:
        if which is not None and name not in which:
            continue
        if name in ("S1_community", "S4_papers_shard"):
            raw_box[0] = None                                  # the products graph is not needed any more
            _free_all()
        left = budget_s - (time.perf_counter() - t_all)
        if left < 20:
            sections[name] = {"skipped": f"extras budget of {budget_s:.0f} s exhausted"}
            continue
        t0 = time.perf_counter()
        try:
            sections[name] = fn()
        except Exception as e:  # noqa: BLE001  (reporting only)
            import traceback
            sections[name] = {"failed": repr(e)[:300], "where": traceback.format_exc()[-600:]}
        sections[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        _free_all()
    return sections
