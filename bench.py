#!/usr/bin/env python3
"""bench.py -- SGAP pre-propagation SpMM throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S1_products] [--no-cpu-baseline] [--no-papers] [--no-extras]

One "step" = one full k-hop propagation (prop_steps SpMM launches, k=3 for the headline config
"SGC prop_steps=3 on ogbn-products") over a synthetic ogbn-products-shaped graph (Chung-Lu, N=2 449 029,
61.86 M undirected edges, d=100; sgl_amd/synthetic.py).  A_hat, X and all hop buffers are resident in HBM when
the timed region starts.  value = nnz(A_hat) * d * k * steps / time  [edge*featdim/s], whole job.

N>1, one rank per GPU: either launched by torch.distributed.run (RANK / WORLD_SIZE in the environment), or as a bare
`python bench.py --gpus N`, in which case this process starts the N ranks itself (benchlib/launch.py; the reference's
analogue spawns its ranks the same way, tasks/node_classification_dist.py:42-61), relays rank 0's line and exits non-zero with
a `value: null` line if any rank fails.  What `value` measures is the contract layout (north_star,
SURVEY 8(e)): A_hat ROW-SHARDED IN STORAGE -- rank 0 generates the raw graph and hands every rank only its nnz-balanced row
block, each rank normalises its own block (sgl_norm_block_*, one all-reduce of the degree vector) and keeps nothing else --
plus a per-hop all-gather of the feature block over RCCL, column chunks software-pipelined across hops.  Two things are
measured during setup, not assumed: the exchange (need-aware packed exchange -- a rank receives only the rows its block
gathers, sgl_amd/dist/halo.py -- as grouped send/recv or as one all_to_all_single, grouped p2p of full replicas, RCCL
all-gather) and the pipelining granularity (2, 3 or 4 column chunks).  Validated without any replica of A_hat (exact
bit-checksums of the exchanged rows + sampled rows recomputed in fp64).  --layout auto / cols / grid / all additionally
build alternatives that REPLICATE A_hat (reported under config.plan.alternatives).  Total work is fixed -> "scaling": "strong".
The helper modules live in benchlib/ (engine, rows = the contract layout, layouts = the alternatives, diagnostics, papers).

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  roofline     : dominant kernel (spmm_kernel) algorithmic bytes per launch / measured launch time vs 8 TB/s HBM
  cpu_baseline : the reference's own CPU kernel (oracle/_ref, else the C restatement) timed on this node's host
                 cores on a bounded row sample of the same workload (N=1 only)
  sections     : (S1 single-GPU runs only, unless --no-extras) the BASELINE configs the headline does not cover, each with workload,
                 ms, roofline, validated: S0_pubmed (config 1, compared with the CPU oracle and timed beside the reference CPU
                 path), S2_gamlp (config 3), S4_products (config 5's operator sweep + a torch-CPU aggregator baseline),
                 S1_community (the products degree law WITH communities: reorder=None vs "auto"), S4_papers_shard (configs 4/5 as
                 one rank of the 8-GPU job: NAFS + every MessageOp); benchlib/extras.py
  papers100M   : (S1 runs only, unless --no-papers) the same measurement on an ogbn-papers100M-shaped graph
                 (111 M nodes, ~3.34 G non-zeros, d=128, k=3, rows generated per rank on device), row-sharded in
                 storage over the same N ranks: value, ms per hop, roofline fraction.  Bounded by a watchdog: if it
                 overruns, the line is printed without it.
"""
import argparse
import os as _os
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # multi-process GPU work needs dmabuf IPC on this driver
import json
import os
import sys
import time

import numpy as np  # noqa: F401
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import (HBM_PEAK_BYTES, _PHASE, _QuietStdout, _phase, _replayed_profile, algorithmic_bytes_per_hop,  # noqa: E402,F401
                             baseline_metric, workload_text)
from benchlib.diagnostics import _diagnostics  # noqa: E402
from benchlib.engine import GpuEngine, cpu_baseline, engine_from_env  # noqa: E402,F401
from benchlib.launch import needs_self_launch, self_launch  # noqa: E402
from benchlib.layouts import _select_layout  # noqa: E402
from benchlib.papers import papers_section  # noqa: E402,F401
from benchlib.rows import _Job  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("SGL_BENCH_WORKLOAD", "S1_products"))
    ap.add_argument("--pieces", type=int, default=2, help="row pieces per rank (N>1): transfers start per piece")
    ap.add_argument("--col-chunks", default="3",
                    help="column chunks of the feature block for the software-pipelined exchange (N>1); 1 = plain; default 3 "
                         "(32 + 32 + 36 columns at d = 100: what profiles/r03_scale_model.md predicts fastest below the link peak); "
                         "auto (opt-in) = 2, 3 and 4 are all built, validated and timed during setup and the fastest runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strict", action="store_true", help="bit-exact reference summation order")
    ap.add_argument("--exchange", choices=("auto", "halo", "halo_a2a", "p2p", "allgather", "push"),
                    default=os.environ.get("SGL_BENCH_EXCHANGE", "halo"),
                    help="N>1 transport of the per-hop all-gather: halo = need-aware (a rank receives only the rows its block "
                         "gathers, packed into a compact table; sgl_amd/dist/halo.py), halo_a2a = the same as one all_to_all_single, p2p = every row to every rank by grouped "
                         "RCCL send/recv, allgather = RCCL all-gather on padded pieces.  Default halo: a fixed, reproducible path (if it fails "
                         "to build or to validate on some rank the job falls back to p2p and says so).  Opt-in: auto = time one "
                         "hop's exchange with each during setup and keep the fastest; push = stores into peer replicas from the "
                         "SpMM kernel over HIP IPC, validated against p2p before it may run")
    ap.add_argument("--layout", choices=("rows", "auto", "cols", "grid", "all"), default=os.environ.get("SGL_BENCH_LAYOUT", "rows"),
                    help="N>1: rows (default) = A_hat row-sharded in storage + per-hop all-gather: the contract layout, the only "
                         "one whose figure is `value` unless another is asked for.  Alternatives that REPLICATE A_hat, opt-in: "
                         "cols = feature-sharded (each GPU runs the whole chain on d/N columns, no communication); grid = 2 row "
                         "blocks x N/2 column slices, pair exchange relayed over all links; auto = rows + ONE alternative (cols "
                         "up to 4 ranks, grid from 8), the faster runs and the other is listed under plan.alternatives; all = "
                         "every layout")
    ap.add_argument("--grid-pieces", default="4",
                    help="row pieces per rank of the grid layout; a comma list is tried and the fastest count kept")
    ap.add_argument("--setup-budget", type=float, default=float(os.environ.get("SGL_BENCH_SETUP_BUDGET", "120")),
                    help="seconds of untimed setup after which further layout candidates are skipped (and listed)")
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("SGL_BENCH_WATCHDOG", "600")),
                    help="N>1: seconds after which a job that is stuck (a collective some rank never entered) is ended: rank 0 "
                         "prints a JSON line with value null, the phase it was in and exits non-zero instead of hanging")
    ap.add_argument("--no-papers", action="store_true",
                    help="skip the ogbn-papers100M-shaped secondary measurement of an S1_products run")
    ap.add_argument("--papers-budget", type=float, default=float(os.environ.get("SGL_BENCH_PAPERS_BUDGET", "150")),
                    help="watchdog (s) of the papers100M-shaped section: on overrun the JSON line is printed without it")
    ap.add_argument("--papers-k", type=int, default=3,
                    help="prop_steps of the papers100M-shaped section (3 by default; 10 = BASELINE config 5's hop count: raise --papers-budget)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary sections of an S1_products single-GPU run (BASELINE configs 1, 3, 5 and the per-rank share "
                         "of 4/5: benchlib/extras.py); with --no-papers and --no-cpu-baseline this is the fast path")
    ap.add_argument("--extras-budget", type=float, default=float(os.environ.get("SGL_BENCH_EXTRAS_BUDGET", "240")),
                    help="seconds the secondary sections may take in total: a section that would start later is recorded as skipped")
    ap.add_argument("--detail-out", default=os.environ.get("SGL_BENCH_DETAIL_OUT"),
                    help="file that receives the secondary sections in full (the line carries their short form)")
    ap.add_argument("--extras-scale", choices=("full", "small"), default=os.environ.get("SGL_BENCH_EXTRAS_SCALE", "full"),
                    help="small = the same sections on graphs that finish in seconds (tests)")
    ap.add_argument("--reorder", choices=("community", "auto"), default=None,
                    help="N = 1: process the rows of A_hat in a plan-time locality order (GraphOp(reorder=...): bit-identical results); "
                         "auto keeps the order only when it makes the graph measurably more local (workload S1_community: yes; S1_products: no)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dup2", action="store_true",
                    help="edge weight 2.0 instead of 1.0: the reference's Ogbn loader symmetrises an already "
                         "bidirectional edge list and CSR construction sums the duplicates (dataset/ogbn.py:45-53)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="debug: run the row-piece (multi-GPU) code path even with one GPU")
    return ap.parse_args(argv)


def _step_stats(step_ms):
    if not step_ms:
        return None
    a = np.sort(np.asarray(step_ms, dtype=np.float64))
    return {"n": int(a.size), "median": float(np.median(a)), "min": float(a[0]), "max": float(a[-1]),
            "p10": float(np.percentile(a, 10)), "p90": float(np.percentile(a, 90)),
            "spread_pct": float((a[-1] - a[0]) / np.median(a) * 100.0)}


def run(args, engine_cls=None, workloads=None, emit=print):
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if engine_cls is None:
        engine_cls = engine_from_env()
    if world != args.gpus:
        # main() starts the ranks itself when no launcher did; only a caller of run() with a half-set environment ends up here
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1: call bench.main() (it starts the N ranks) or launch with torch.distributed.run")
        args.gpus = world
    quiet = _QuietStdout()
    if emit is print:
        quiet.mute()
    engine = engine_cls(local_rank)
    device = engine.device
    if workloads is None:
        from sgl_amd import synthetic
        workloads = synthetic.WORKLOADS
    wl = workloads[args.workload]
    job = _Job(args, engine, rank, world, wl)
    guard = None
    line_box = [None]                         # rank 0's JSON line once the timed region is over (the watchdog prints it if later phases hang)
    timed_done = [False]                      # every rank: the timed region is behind us
    if world > 1 and emit is print and args.watchdog > 0:
        # a rank that fails before a collective leaves the others waiting in it for ever: end the job with a diagnosis
        import threading

        def stuck():
            if rank == 0:
                quiet.unmute()
                if line_box[0] is not None:
                    # the timed region is over: the measured line goes out as it stands, only what came after it is missing
                    line_box[0]["watchdog"] = f"no progress after {args.watchdog:.0f} s in phase {_PHASE[0]!r}; later sections missing"
                    print(json.dumps(line_box[0]), flush=True)
                    os._exit(0)
                print(json.dumps({"metric": baseline_metric(), "value": None, "unit": "edge\u00b7featdim/s", "n_gpus": world,
                                  "error": f"watchdog: no progress after {args.watchdog:.0f} s", "phase": _PHASE[0]}), flush=True)
            os._exit(0 if timed_done[0] else 3)       # (ranks other than 0 never hold the line: they go by the phase)
        guard = threading.Timer(args.watchdog, stuck)
        guard.daemon = True
        guard.start()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _phase("init_process_group")
        dist.init_process_group(engine.backend, rank=rank, world_size=world, **engine.init_kwargs())
        job.own_group = True
    n, d, K = job.n, job.d, job.K

    t_setup = time.perf_counter()
    job.t_setup = t_setup
    _phase("load_workload (generate, scatter row blocks, normalise per block)")
    job.load_workload(wl)
    nnz = job.nnz
    sharded = world > 1 or args.force_sharded
    halves = {}
    if not sharded:
        step, info0 = engine.single_step(args, job.rowptr, job.col, job.val, job.x0, n, d, K)
        job.info.update(info0)
    else:
        _phase("select_layout")
        step, halves = _select_layout(job)
    info = job.info
    setup_s = time.perf_counter() - t_setup

    # ---- the timed region: W warm-up steps, then exactly K steps between barrier + device synchronise ------------
    _phase("timed region")
    for _ in range(args.warmup):
        step()
    job.sync_all()
    t_start, t_stop, t_elapsed_ms = engine.timer()
    marks = engine.step_marks(args.steps) if hasattr(engine, "step_marks") else None
    t0 = time.perf_counter()
    t_start()
    for i in range(args.steps):
        step()
        if marks is not None:
            marks.mark(i)                     # one event record per step on the launch stream: the spread of the K steps (below)
    t_stop()
    job.sync_all()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    gpu_ms = t_elapsed_ms()
    timed_done[0] = True
    step_ms = marks.durations_ms(t_start.event if hasattr(t_start, "event") else None) if marks is not None else None

    # ---- the line exists from here on: whatever follows (self-validation, diagnostics, baselines, the papers100M-shaped section)
    # only ADDS to it, and the watchdog prints it as it stands instead of losing a measured value to a stuck collective ------------
    out = None
    if rank == 0:
        value = nnz * d * K * args.steps / elapsed
        hop_s = (gpu_ms * 1e-3) / (K * args.steps)           # average launch duration from HIP events
        n_rows_local = (job.rowptr.numel() - 1) if (not sharded and job.rowptr is not None) else n
        alg = nnz * d * 4 + nnz * 8 + (n_rows_local + 1) * 4 + n_rows_local * d * 4   # SURVEY 8(d) no-reuse gather model
        if world > 1:
            alg = alg / world                                 # per-GPU share of one hop
        achieved = alg / hop_s
        prof = _replayed_profile(args.workload, world)
        traffic = prof.get("hbm_bytes_per_launch") if prof else None
        out = {
            "metric": baseline_metric(),
            "value": value, "unit": "edge\u00b7featdim/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            # the K timed steps one by one (HIP events on rank 0's launch stream, inside the one timed region `value` comes from):
            # median and spread -- what one run can say about its own noise; box-to-box spread is a separate matter
            "ms_per_step_stats": _step_stats(step_ms),
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(args.workload, K),
                       "n_nodes": n, "nnz_a_hat": nnz, "feat_dim": d, "prop_steps": K,
                       "parallelism": info.get("parallelism", "single GPU") if sharded else "single GPU",
                       "summation": "strict (no row splitting: bit-exact reference order)" if args.strict else "reference order per row; long rows split into pieces (> 2048 nnz at this size: the threshold follows the GLOBAL nnz, 32 ... 2048)",
                       "plan": info, "setup_s": round(setup_s, 2), "validated": None, "validation": None, "diagnostics": None},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_BYTES, "traffic": traffic,
                         # PMC counters need their own rocprofv3 passes: the figure is the builder's pass over this very
                         # command, replayed from the file named here -- not a measurement of this run
                         "traffic_source": (prof.get("source", "profiles/traffic.json") + " (rocprofv3 --pmc pass of the same "
                                            "command, replayed; not measured in this run)") if prof else None,
                         "kernel": "spmm_kernel", "algorithmic_bytes_per_launch": alg,
                         "avg_launch_ms": hop_s * 1e3,
                         "kernel_ms_profile": prof.get("kernel_avg_ms_rocprof") if prof else None,
                         # NOT an HBM utilisation: the counters sit on the fabric side of L2 and count every request, also those the
                         # Infinity Cache serves -- requests per second against the HBM peak (can exceed what HBM itself delivers)
                         "fabric_request_frac": (traffic / hop_s / HBM_PEAK_BYTES) if traffic else None,
                         # the bare-gather ceiling of THIS access pattern measured in this run (probe kernel: the same
                         # column ids, row width and pitch; no CSR stream, arithmetic or stores) and the kernel's gather rate
                         "gather_ceiling_Ggathers_per_s": None,
                         "kernel_Ggathers_per_s": nnz / world / hop_s / 1e9,
                         "frac_of_gather_ceiling": None},
            "cpu_baseline": None,
        }
        line_box[0] = out

    # ---- self-validation of what the timed steps left behind (never part of `value`) -------------------------------------------
    _phase("validation")
    if not sharded:
        if hasattr(engine, "validate_single"):
            try:
                v = engine.validate_single()
            except Exception as e:  # noqa: BLE001
                v = {"ok": False, "failed": repr(e)[:300]}
            if rank == 0:
                out["config"]["validated"], out["config"]["validation"] = v.get("ok"), v
    elif rank == 0:
        # N > 1 layouts were validated during setup (before they could be timed): exact bit-checksums of the exchanged rows +
        # sampled rows of every rank recomputed in fp64 (benchlib/rows.py); a layout that fails never reaches the timed region
        out["config"]["validated"] = True
        how = ("bit-checksums of exchanged rows + 512 sampled rows per rank and column chunk recomputed in fp64"
               if info.get("layout", "rows") == "rows" else "last hop against the single-GPU chain on a gathered replica of A_hat (1e-5)")
        out["config"]["validation"] = {"when": "setup, before timing", "how": how, "ok": True}

    _phase("diagnostics")
    job.single_gpu_ms_replayed = (_replayed_profile(args.workload, 1) or {}).get("single_gpu_ms_per_step")
    try:
        diag = _diagnostics(job, halves, measured_step_ms=elapsed * 1e3 / args.steps) if job.budget_left() > -60 else \
            {"skipped": "setup budget exhausted"}
    except Exception as e:  # noqa: BLE001  (reporting only; the measured value is already in hand)
        diag = {"failed": repr(e)}
    if rank == 0:
        out["config"]["diagnostics"] = diag

    if not sharded and rank == 0 and hasattr(engine, "gather_ceiling"):
        ceiling = engine.gather_ceiling(job.col, job.x0, d)
        out["roofline"]["gather_ceiling_Ggathers_per_s"] = ceiling
        out["roofline"]["frac_of_gather_ceiling"] = (nnz / world / hop_s / 1e9 / ceiling) if ceiling else None

    if not sharded and not args.no_cpu_baseline and rank == 0:
        try:
            cpu = cpu_baseline(job.rowptr, job.col, job.val, job.x0, d)
        except Exception as e:  # noqa: BLE001  (baseline is reporting only; never blocks the GPU number)
            cpu = {"value": None, "unit": "edge\u00b7featdim/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        out["cpu_baseline"] = cpu

    # ---- secondary sections (bounded; never endanger the line above): the other BASELINE configs on this GPU (benchlib/extras.py)
    # and the papers100M-shaped graph on the same ranks.  ONE watchdog covers them all: on overrun the line goes out with whatever
    # sections are finished.
    def emit_line():
        if rank == 0:
            quiet.unmute()
            try:
                text = json.dumps(out)
            except RuntimeError:                  # the watchdog thread caught the main thread adding a key: once more
                time.sleep(0.2)
                text = json.dumps(out)
            emit(text)
            sys.stdout.flush()
            if emit is print:
                quiet.mute()                  # late library chatter (communicator teardown) goes to stderr too

    small = getattr(args, "extras_scale", "full") == "small" and args.workload == "S1_small"      # (tests: the same sections, in seconds)
    can_extra = (args.workload == "S1_products" or small) and not args.force_sharded and hasattr(engine, "hashed_block")
    want_papers = can_extra and not small and not args.no_papers
    want_extras = can_extra and world == 1 and not getattr(args, "no_extras", False)
    if want_papers or want_extras:
        import threading
        _phase("secondary sections")
        done = threading.Event()
        budget = (args.papers_budget if want_papers else 0) + (args.extras_budget + 60 if want_extras else 0)

        def overrun():
            if done.is_set():
                return
            if rank == 0:
                out["secondary_sections_watchdog"] = f"not finished within {budget:.0f} s (phase {_PHASE[0]!r}): the line is printed with what was done"
                if want_papers and "papers100M" not in out:
                    out["papers100M"] = {"skipped": f"did not finish within the {budget:.0f} s watchdog"}
                emit_line()
            os._exit(0)                       # a collective may be stuck: do not wait for it

        timer = threading.Timer(budget, overrun)
        timer.daemon = True
        timer.start()
        exchange = info.get("exchange", "p2p")
        args.col_chunks_chosen = getattr(job, "col_chunks_chosen", None)
        del step, halves
        job.drop_full()
        job.block = job.x0 = None
        engine._single = None             # the operands validate_single() looked at
        import gc
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        # Nothing below may cost the headline: every step is fenced, a failure is recorded in the line instead of raised.
        extras_fn, detail, t_x = None, {}, 0.0

        def publish_sections():
            if not (want_extras and rank == 0 and detail.get("sections")):
                return
            try:
                # LAST key of the line and short on purpose (4 significant digits, tables as rows): whoever keeps only the tail of a
                # long line still reads every section.  The complete dictionaries go to --detail-out.
                out.pop("sections", None)
                out["sections"] = compact_sections(detail["sections"])
                if getattr(args, "detail_out", None):
                    with open(args.detail_out, "w") as f:
                        json.dump({"head": {k: out[k] for k in ("metric", "value", "ms_per_step", "n_gpus", "steps", "warmup")},
                                   "sections": detail["sections"]}, f, indent=1)
            except Exception as e:  # noqa: BLE001
                out["sections"] = {"failed": f"could not render the sections: {e!r}"[:300]}

        if want_extras:
            try:
                from benchlib.extras import compact_sections, run_extras as extras_fn
                _phase("extras: S0 / S2 / S4_products / S1_community")
                t_x = time.perf_counter()
                extras_fn(args, engine, detail, budget_s=args.extras_budget, which=("S0_pubmed", "S2_gamlp", "S4_products", "S1_community"))
                t_x = time.perf_counter() - t_x
            except Exception as e:  # noqa: BLE001
                out["secondary_sections_error"] = repr(e)[:300]
                extras_fn = None
            publish_sections()               # (the watchdog prints whatever is published if the papers100M section overruns)
        if want_papers:
            _phase("papers100M section")
            try:
                papers = papers_section(args, engine, rank, world, exchange)
            except Exception as e:  # noqa: BLE001
                import traceback
                papers = {"failed": repr(e)[:200], "where": traceback.format_exc()[-700:]}
            if rank == 0:
                out["papers100M"] = papers
        if want_extras and extras_fn is not None:
            try:
                _phase("extras: S4_papers_shard")
                gc.collect()
                torch.cuda.empty_cache()
                extras_fn(args, engine, detail, budget_s=args.extras_budget - t_x, which=("S4_papers_shard",))
            except Exception as e:  # noqa: BLE001
                out["secondary_sections_error"] = repr(e)[:300]
        publish_sections()                   # again: `sections` stays the LAST key whatever was added since
        done.set()
        timer.cancel()
    emit_line()
    if guard is not None:
        guard.cancel()
    if world > 1:
        dist.barrier()
    if job.own_group and dist.is_initialized():
        dist.destroy_process_group()
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if needs_self_launch(args):
        # `python bench.py --gpus N` with no launcher: start the N ranks here (benchlib/launch.py), relay rank 0's line
        sys.exit(self_launch(args, argv, os.path.abspath(__file__), baseline_metric()))
    run(args)


if __name__ == "__main__":
    main()
