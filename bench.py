#!/usr/bin/env python3
"""bench.py -- SGAP pre-propagation SpMM throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S1_products] [--no-cpu-baseline] [--no-papers]

One "step" = one full k-hop propagation (prop_steps SpMM launches, k=3 for the headline config
"SGC prop_steps=3 on ogbn-products") over a synthetic ogbn-products-shaped graph (Chung-Lu, N=2 449 029,
61.86 M undirected edges, d=100; sgl_amd/synthetic.py).  A_hat, X and all hop buffers are resident in HBM when
the timed region starts.  value = nnz(A_hat) * d * k * steps / time  [edge*featdim/s], whole job.

N>1 (launched by torch.distributed.run, one rank per GPU).  The contract layout (north_star) is "rows":
A_hat ROW-SHARDED IN STORAGE -- rank 0 generates the raw graph and hands every rank only its nnz-balanced row block,
each rank normalises its own block (sgl_norm_block_*, one all-reduce of the degree vector) and keeps nothing else --
plus a per-hop all-gather of the feature block over RCCL, overlapped with the SpMM of the next row piece / column
chunk.  It is always built, validated (exact bit-checksums of the exchanged replicas + sampled rows recomputed in
fp64) and timed, and its figures are always in the JSON line (config.plan.rows).  --layout auto (default) additionally
tries ONE alternative that replicates A_hat -- "cols" (feature-sharded, no communication) up to 4 ranks, "grid"
(2 row blocks x N/2 column slices, relayed exchange) from 8 -- validates it against the single-GPU chain and runs the
faster of the two in the timed region; candidates that do not fit the setup budget are skipped and listed.
Total work is fixed -> "scaling": "strong".

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  roofline     : dominant kernel (spmm_kernel) algorithmic bytes per launch / measured launch time vs 8 TB/s HBM
  cpu_baseline : the reference's own CPU kernel (oracle/_ref, else the C restatement) timed on this node's host
                 cores on a bounded row sample of the same workload (N=1 only)
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

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_BYTES = 8.0e12  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def baseline_metric():
    """the metric string of BASELINE.json, verbatim (the file travels with the repo snapshot)"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:  # noqa: BLE001
        return "pre-prop SpMM throughput (edge\u00b7featdim/s), ogbn-products k=3, 1/2/4/8 GPU"


def algorithmic_bytes_per_hop(n, nnz, d):
    """SURVEY.md section 8(d) no-reuse gather model: gathered X rows + (col,val) + rowptr + Y write"""
    return nnz * d * 4 + nnz * 8 + (n + 1) * 4 + n * d * 4


def cpu_baseline(rowptr, col, val, x, d, budget_s=20.0):
    """time the reference CPU kernel (csrc/matmul.c:23-40) on the first rows of the same A_hat / X"""
    import oracle  # test infrastructure: allowed here as the reported baseline only
    n = rowptr.numel() - 1
    rows = min(n, 400_000)
    rp = rowptr[:rows + 1].cpu().numpy()
    nnz_s = int(rp[-1])
    compacted = ""
    if x.shape[0] * d * 4 > (8 << 30):
        # papers100M-sized replica (57 GB): only the rows the sample gathers travel to the host, columns re-indexed
        uniq, inv = torch.unique(col[:nnz_s].long(), return_inverse=True)
        c = inv.to(torch.int32).cpu().numpy()
        xh = x[uniq][:, :d].contiguous().cpu().numpy()
        compacted = f" (X compacted to the {uniq.numel()} gathered rows)"
    else:
        c = col[:nnz_s].cpu().numpy()
        xh = x.cpu().numpy()
        xh = np.ascontiguousarray(xh[:, :d])
    v = val[:nnz_s].cpu().numpy()
    kind = "reference" if oracle.load_reference_lib() is not None and xh.shape[0] * d < 2 ** 31 else "port"
    fn = (lambda: oracle.reference_spmm(rp, c, v, xh, n_rows=rows)) if kind == "reference" else \
         (lambda: oracle.oracle_spmm(rp, c, v, xh, n_rows=rows))
    fn()  # warm-up
    times = []
    t_all = time.perf_counter()
    for _ in range(5):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s:
            break
    t = float(np.median(times))
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    threads = int(os.environ.get("OMP_NUM_THREADS", len(cpus)))
    cpu_model, physical = "unknown CPU", None
    try:
        # hardware threads -> physical cores: distinct (package, core id) pairs among the CPUs this process may run on
        seen, cur = set(), {}
        with open("/proc/cpuinfo") as f:
            for line in f:
                if ":" in line:
                    key_, text_ = (p_.strip() for p_ in line.split(":", 1))
                    cur[key_] = text_
                    if key_ == "model name" and cpu_model == "unknown CPU":
                        cpu_model = text_
                elif cur:
                    if int(cur.get("processor", -1)) in cpus and "core id" in cur:
                        seen.add((cur.get("physical id", "0"), cur["core id"]))
                    cur = {}
        physical = len(seen) or None
    except Exception:  # noqa: BLE001
        pass
    cores = min(threads, physical) if physical else threads
    out = {"value": nnz_s * d / t, "unit": "edge\u00b7featdim/s", "cores": cores, "threads": threads, "kind": kind,
           "sample": f"first {rows} rows of A_hat ({nnz_s} nnz) x d={d}{compacted}, one hop, median of {len(times)} reps, "
                     f"OpenMP static schedule, {threads} threads on {cores} physical cores of {cpu_model}",
           "ms_per_hop_sample": t * 1e3}
    # B2 of BASELINE.md: the reference's non-Linux branch `adj.dot(x)` (base_op.py:34), scipy, single thread, on a
    # smaller slice of the same rows (bounded: a few seconds)
    try:
        import scipy.sparse as sp
        r2 = min(rows, 50_000)
        a = sp.csr_matrix((v[:int(rp[r2])], c[:int(rp[r2])], rp[:r2 + 1]), shape=(r2, xh.shape[0]))
        t0 = time.perf_counter()
        a.dot(xh)
        ts = time.perf_counter() - t0
        out["scipy_dot"] = {"value": int(rp[r2]) * d / ts, "unit": "edge\u00b7featdim/s", "cores": 1,
                            "sample": f"scipy csr.dot on the first {r2} rows ({int(rp[r2])} nnz)"}
    except Exception as e:  # noqa: BLE001
        out["scipy_dot"] = {"value": None, "sample": f"failed: {e}"}
    return out


class _QuietStdout:
    """RCCL prints a version banner through C stdio (flushed at exit, i.e. AFTER our JSON line).  The driver reads ONE
    JSON line from stdout, so everything except that line is routed to stderr at the file-descriptor level."""

    def __init__(self):
        self.saved = None

    def mute(self):
        if self.saved is None:
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)

    def unmute(self):
        if self.saved is not None:
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)      # push buffered C-level output out while fd 1 still is stderr
            except Exception:  # noqa: BLE001
                pass
            sys.stdout.flush()
            os.dup2(self.saved, 1)
            os.close(self.saved)
            self.saved = None


class GpuEngine:
    """Everything device-specific in the bench: workload construction, the two step functions, timing.
    tests/test_bench_orchestration.py substitutes a CPU/gloo engine to exercise the distributed orchestration
    (broadcast, shard bounds, exchange, barrier/MAX timing, JSON contract) without GPUs."""
    backend = "nccl"
    transports = ("p2p", "allgather")     # process-group transports the auto-selection may choose from
    halo_collective = True                # RCCL has all_to_all_single with split sizes: the need-aware exchange in one call

    def __init__(self, local_rank):
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the sgl_amd hot path has no CPU fallback)")
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)

    def init_kwargs(self):
        return {"device_id": self.device}

    # ---- workload pieces ---------------------------------------------------------------------------------------------
    def build_raw(self, args, wl):
        """the raw (un-normalised) symmetric adjacency A of a Chung-Lu workload on this device: (rowptr, col, val)"""
        from sgl_amd import synthetic
        return synthetic.chung_lu_torch(wl["n"], wl["m"], wl["d_max"], seed=args.seed, device=self.device,
                                        weight=2.0 if getattr(args, "dup2", False) else 1.0)

    def features(self, args, wl):
        from sgl_amd import synthetic
        if wl.get("hashed"):
            return synthetic.hashed_features_torch(args.seed, 0, wl["n"], wl["d"], device=self.device)
        return synthetic.features_torch(wl["n"], wl["d"], seed=args.seed, device=self.device,
                                        kind="pubmed" if args.workload.startswith("S0") else "normal")

    def build_workload(self, args, wl):
        """single GPU: the whole A_hat (LaplacianGraphOp r = 0.5, normalised on device) + features.  Hashed workloads
        (papers100M-shaped) come out of the generator directly: directed, values in [0, 1/32), throughput only."""
        from sgl_amd import device as dev
        n = wl["n"]
        if wl.get("hashed"):
            lo, hi = self.hashed_rows(wl)
            blk = self.hashed_block(args, wl, lo, hi)
            return blk.rowptr, blk.col, blk.val, self.features(args, wl)
        a_ptr, a_col, a_val = self.build_raw(args, wl)
        rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
        return rowptr, col, val, self.features(args, wl)

    @staticmethod
    def hashed_rows(wl):
        """the row block a single-GPU hashed workload multiplies: everything, or share i of `row_block` = (i, parts)"""
        if "row_block" in wl:
            i, parts = wl["row_block"]
            return wl["n"] * i // parts, wl["n"] * (i + 1) // parts
        return 0, wl["n"]

    def hashed_table(self, wl):
        from sgl_amd import synthetic
        return synthetic.degree_table(wl["mean_deg"], wl["d_max"])

    def hashed_bounds(self, args, wl, parts):
        """nnz-balanced row-block boundaries of a hashed graph: every rank derives them from the (hash-generated) degrees
        of ALL rows on its own device -- identical everywhere, nothing is communicated"""
        import ctypes
        from sgl_amd import _lib
        from sgl_amd.dist import balanced_bounds_device
        n = wl["n"]
        tab = torch.from_numpy(self.hashed_table(wl)).to(self.device)
        deg = torch.empty(n, dtype=torch.int64, device=self.device)
        _lib.check(_lib.lib().sgl_synth_degrees(ctypes.c_uint64(args.seed), 0, n, _lib.ptr(tab), _lib.ptr(deg),
                                                _lib.current_stream_ptr()), "sgl_synth_degrees")
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(deg, 0, out=rowptr[1:])
        del deg
        return balanced_bounds_device(rowptr, parts), int(rowptr[-1])

    def hashed_block(self, args, wl, lo, hi):
        from sgl_amd import synthetic
        from sgl_amd.dist import RowBlock
        rowptr, col, val = synthetic.hashed_block_torch(args.seed, lo, hi - lo, wl["n"], self.hashed_table(wl), device=self.device)
        return RowBlock(lo, hi, wl["n"], rowptr, col, val)

    def normalize_block(self, blk, r=0.5, alpha=None, symmetric=True):
        """rows [lo, hi) of A_hat from the same rows of the raw symmetric A: collective only in the degree vector"""
        from sgl_amd import device as dev
        from sgl_amd.dist import RowBlock
        rowptr, col, val = dev.normalize_block(blk.rowptr, blk.col, blk.val, blk.lo, blk.n, r, alpha, symmetric=symmetric)
        return RowBlock(blk.lo, blk.hi, blk.n, rowptr, col, val)

    def block_piece_spmms(self, args, blk, pieces, weights=None):
        from sgl_amd.dist import block_piece_spmms
        return block_piece_spmms(blk, pieces, weights, strict=args.strict)

    def block_halo(self, args, blk, bounds):
        """need-aware exchange of the row-sharded layout (sgl_amd/dist/halo.py): plan, propagator on compact tables and the
        block with its columns relabelled to the compact table (for the sampled-row check)"""
        from sgl_amd.dist import RowBlock
        from sgl_amd.dist.halo import block_halo
        plan, prop, handle = block_halo(blk, bounds, strict=args.strict)
        cblk = RowBlock(blk.lo, blk.hi, plan.n_compact, blk.rowptr, handle.col if handle is not None else blk.col, blk.val)
        return plan, prop, cblk

    def feature_rows(self, args, wl, lo, hi):
        """rows [lo, hi) of a hashed workload's feature matrix (a rank of the need-aware layout generates only its own)"""
        from sgl_amd import synthetic
        return synthetic.hashed_features_torch(args.seed, lo, hi - lo, wl["d"], device=self.device)

    def gather_ceiling(self, col, x, d, max_idx=64 << 20):
        """What the memory system gives the bare access pattern of this workload (sgl_probe_gather_f32: whole-row gathers
        at the workload's own column ids, row width and pitch; no CSR stream, no arithmetic, no stores): the ceiling the
        SpMM's gather rate is quoted against, measured in this run.  Returns G gathers/s or None."""
        from sgl_amd import _lib
        try:
            idx = col[: min(int(col.numel()), max_idx)]
            rf = (d + 3) // 4 * 4
            ld = x.stride(0) if x.shape[0] > 1 else rf
            if rf > 256 or ld % 4 or x.data_ptr() % 16 or ld < rf:
                return None
            sink = torch.zeros(4, device=self.device)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def go():
                _lib.check(_lib.lib().sgl_probe_gather_f32(_lib.ptr(x), ld, _lib.ptr(idx), idx.numel(), rf, 16, _lib.ptr(sink),
                                                           _lib.current_stream_ptr()), "sgl_probe_gather_f32")
            go()
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                ev0.record()
                go()
                ev1.record()
                torch.cuda.synchronize()
                ts.append(ev0.elapsed_time(ev1))
            return idx.numel() / (sorted(ts)[1] * 1e-3) / 1e9
        except Exception:  # noqa: BLE001  (reporting only)
            return None

    def sampled_rows_check(self, blk, x_prev, y_local, samples=512, tol=1e-5):
        """kernel-independent check of this rank's SpMM: `samples` of its rows recomputed in fp64 with plain torch
        indexing from the replica the hop read (x_prev) and compared with what the kernel wrote (y_local)"""
        n_loc = blk.n_local
        if n_loc == 0:
            return True
        g = torch.Generator(device="cpu").manual_seed(1234 + blk.lo)
        rows = torch.randint(0, n_loc, (min(samples, n_loc),), generator=g).to(blk.device)
        b, e = blk.rowptr[rows], blk.rowptr[rows + 1]
        cnt = e - b
        if int(cnt.sum()) == 0:
            return bool((y_local[rows] == 0).all())
        seg = torch.repeat_interleave(torch.arange(rows.numel(), device=blk.device), cnt)
        pos = torch.arange(int(cnt.sum()), device=blk.device) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt) + \
            torch.repeat_interleave(b, cnt)
        contrib = blk.val[pos].double().unsqueeze(1) * x_prev[blk.col[pos].long()].double()
        want = torch.zeros((rows.numel(), x_prev.shape[1]), dtype=torch.float64, device=blk.device).index_add_(0, seg, contrib)
        mag = torch.zeros_like(want).index_add_(0, seg, contrib.abs())
        err = (y_local[rows].double() - want).abs()
        return bool((err <= tol * mag.clamp_min(1e-30) + 1e-30).all())

    def single_step(self, args, rowptr, col, val, x0, n, d, K):
        from sgl_amd import device as dev
        csr = dev.DeviceCSR(rowptr, col, val, (rowptr.numel() - 1, n), strict=args.strict)
        n_out = rowptr.numel() - 1
        free, _ = torch.cuda.mem_get_info()
        pingpong = n_out == n and K > 2 and K * n_out * dev.row_pitch(d) * 4 > free // 2
        # K hop matrices that would not fit (the whole papers100M-shaped graph: 57 GB each): two buffers, alternating
        bufs = [dev.alloc_rows(n_out, d, self.device) for _ in range(2 if pingpong else K)]
        if pingpong:
            bufs = [bufs[h % 2] for h in range(K)]
        src0 = dev.upload_rows(x0, self.device) if dev.row_pitch(d) != d else x0   # re-pack into the line-aware pitch

        x_in = dev.padded_parent(src0)
        outs = [dev.padded_parent(b) for b in bufs]

        info = csr.info()
        if pingpong:
            info["hops_retained"] = "last two only (K hop matrices of this size do not fit one GPU)"
        if n_out != n:
            # a row block against the full replica (S3_papers_shard): K launches of the same hop
            def step():
                for h in range(K):
                    csr.spmm(x_in, out=outs[h])
            return step, info
        if info["nnz"] < 5_000_000:
            # small graph: the k launches are captured in a hipGraph and replayed (launch-bound regime)
            graph = csr.capture_chain(x_in, outs)
            info["hip_graph"] = True
            return graph.replay, info

        def step():
            csr.spmm_chain(x_in, K, outs=outs)     # the k SpMM launches of one propagate(), issued from one call
        return step, info

    def piece_spmms(self, args, rowptr, col, val, n, my_bounds, rp_host):
        from sgl_amd.dist import device_piece_spmms
        return device_piece_spmms(rowptr, col, val, n, my_bounds, rowptr_host=rp_host, strict=args.strict)

    relay_transport = "relay"             # the grid layout's two-phase exchange over the process group

    def pack_slice(self, x0, a, b):
        """columns [a, b) of x0 as a contiguous matrix, zero-padded to a line-friendly row pitch (the pad columns are
        multiplied too: zeros in, zeros out, no extra cache lines)"""
        from sgl_amd import device as dev
        w = b - a
        out = torch.zeros((x0.shape[0], dev.row_pitch(w, growth=2.0) if w else 0), dtype=x0.dtype, device=x0.device)
        if w:
            out[:, :w] = x0[:, a:b]
        return out

    def sync(self):
        torch.cuda.synchronize()

    def timer(self):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        return (lambda: ev0.record()), (lambda: ev1.record()), (lambda: ev0.elapsed_time(ev1))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("SGL_BENCH_WORKLOAD", "S1_products"))
    ap.add_argument("--pieces", type=int, default=2, help="row pieces per rank (N>1): transfers start per piece")
    ap.add_argument("--col-chunks", default="auto",
                    help="column chunks of the feature block for the software-pipelined exchange (N>1); 1 = plain; auto = 2 and 4 "
                         "are both built, validated and timed during setup and the faster runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strict", action="store_true", help="bit-exact reference summation order")
    ap.add_argument("--exchange", choices=("auto", "halo", "halo_a2a", "p2p", "allgather", "push"),
                    default=os.environ.get("SGL_BENCH_EXCHANGE", "auto"),
                    help="N>1 transport of the per-hop all-gather: halo = need-aware (a rank receives only the rows its block "
                         "gathers, packed into a compact table; sgl_amd/dist/halo.py), halo_a2a = the same as one all_to_all_single, p2p = every row to every rank by grouped "
                         "RCCL send/recv, allgather = RCCL all-gather on padded pieces, auto = time one hop's exchange with each "
                         "during setup and keep the fastest, push = stores into peer replicas from the SpMM kernel (opt-in)")
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
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dup2", action="store_true",
                    help="edge weight 2.0 instead of 1.0: the reference's Ogbn loader symmetrises an already "
                         "bidirectional edge list and CSR construction sums the duplicates (dataset/ogbn.py:45-53)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="debug: run the row-piece (multi-GPU) code path even with one GPU")
    return ap.parse_args(argv)


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


class _Reference:
    """The single-GPU k-hop chain computed on this rank's own replica: what every multi-GPU layout must reproduce."""

    def __init__(self, job):
        self.full_spmm = job.piece_spmms(np.array([0, job.n], dtype=np.int64))[0][0]
        bufs = [torch.empty_like(job.x0) for _ in range(job.K)]

        def chain():
            last = job.x0
            for h in range(job.K):
                self.full_spmm(last, bufs[h])
                last = bufs[h]
            return last
        chain()
        job.engine.sync()
        t0 = time.perf_counter()
        self.last = chain()
        job.engine.sync()
        self.ms = (time.perf_counter() - t0) * 1e3          # one rank's single-GPU step: the yardstick of the fallback rule
        del bufs[:job.K - 1]
        self.scale = max(float(self.last.abs().max()), 1e-30)

    def close(self, block, r0, r1, c0, c1):
        want = self.last[r0:r1, c0:c1]
        return want.numel() == 0 or float((block - want).abs().max()) <= 1e-5 * self.scale


# ---- the layout candidates of an N-rank job (sgl_amd/dist/).  Each builder returns {"step", "check", "describe"} ------

def _build_cols(job, ref):
    """feature-sharded: every rank runs the whole chain on d/N columns, no communication"""
    from sgl_amd.dist import column_slices
    a, b = column_slices(job.d, job.world)[job.rank]
    w, K = b - a, job.K
    xs = job.engine_pack(job.x0, a, b)
    outs = [torch.empty_like(xs) for _ in range(K)]

    def step():
        cur = xs
        for h in range(K if w else 0):
            ref.full_spmm(cur, outs[h])
            cur = outs[h]
    return {"step": step, "check": lambda: K == 0 or ref.close(outs[K - 1][:, :w], 0, job.n, a, b),
            "describe": f"feature-sharded x{job.world} (each GPU: all rows x {w} of {job.d} columns, no communication)"}


def _build_grid(job, ref, row_groups):
    """row_groups row blocks x N/row_groups column slices, the exchange inside a column group relayed over all ranks.
    How many row pieces a hop is cut into trades exposed transfer time (the last piece's) against per-piece launch and
    issue cost, and the optimum depends on what the links deliver -- so every count in --grid-pieces is built,
    validated and timed (untimed setup), and the fastest one is this layout's candidate."""
    from sgl_amd.dist import GridLayout, ShardedPropagator, all_piece_bounds, column_slices, tapered_weights
    K = job.K
    layout = GridLayout(job.world, row_groups)
    rg, cg = layout.coords(job.rank)
    slices = column_slices(job.d, layout.col_groups)
    job.full_adj()
    if job.rp_host is None:
        job.rp_host = job.rowptr.cpu().numpy()
    a, b = slices[cg]
    w = b - a
    xs = job.engine_pack(job.x0, a, b)
    widths = [job.engine_pack(job.x0[:1], sa, sb).shape[1] for sa, sb in slices]
    bufs = [torch.empty_like(xs) for _ in range(job.nbuf)]

    def variant(pieces):
        # the last piece's transfer is the one nothing can hide: make it half as large as the others
        pb = all_piece_bounds(job.rp_host, row_groups, pieces, tapered_weights(pieces))
        fns, _handles = job.piece_spmms(pb[rg])
        prop = ShardedPropagator(fns, pb, rg, row_groups, job.n, transport=getattr(job.engine, "relay_transport", "relay"),
                                 layout=layout, me=job.rank, widths=widths)
        ybufs = [torch.empty((prop.hi - prop.lo, xs.shape[1]), dtype=xs.dtype, device=xs.device) for _ in range(K)]

        def step():
            return prop.propagate(xs, K, x_buffers=bufs, y_buffers=ybufs)   # every buffer preallocated: no allocator traffic
        return prop, step

    counts = [int(t) for t in str(job.args.grid_pieces).split(",") if t.strip()]
    best, timing = None, {}
    for pieces in counts:
        good, made = True, None
        try:
            made = variant(pieces)
            good = bool(ref.close(made[1]()[K][:, :w], made[0].lo, made[0].hi, a, b))
        except Exception as e:  # noqa: BLE001  (same code on every rank, so an error is too; agree() settles it)
            good = False
            sys.stderr.write(f"[bench] grid with {pieces} pieces failed on rank {job.rank}: {e!r}\n")
        if not job.agree(good):
            continue
        timing[pieces] = job.timed_s(made[1], reps=2, warm=0)
        if best is None or timing[pieces] < timing[best[0]]:
            best = (pieces,) + made
    if best is None:
        raise RuntimeError("no grid variant reproduced the single-GPU result")
    pieces, prop, step = best
    job.info["grid_pieces"] = pieces
    job.info["grid_pieces_candidates_ms"] = {str(k): round(v * 1e3, 3) for k, v in timing.items()}
    return {"step": step, "check": lambda: ref.close(step()[K][:, :w], prop.lo, prop.hi, a, b), "halves": (prop, [xs], [bufs]),
            "describe": f"grid {row_groups} row blocks x {layout.col_groups} column slices, pair exchange relayed over all "
                        f"{job.world} ranks, {pieces} row pieces"}


def _select_exchange(job, full, halo):
    """row-sharded layout: pick the transport of the per-hop all-gather.  auto = time one hop's exchange with every candidate --
    the need-aware packed exchange (halo: pack kernel + grouped send/recv of the rows each peer gathers) and the full-replica
    process-group transports (p2p, allgather) -- and keep the fastest (decision on the MAX over ranks, so every rank picks the
    same); the fused push transport is an opt-in further candidate that must map all peers, reproduce the process-group result
    and be >= 3 % faster."""
    args, engine, info, K, device = job.args, job.engine, job.info, job.K, job.device
    if full is None:
        return "halo"
    prop, handles, x_chunks, cbufs = full["prop"], full["handles"], full["x_chunks"], full["cbufs"]

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
    if exchange == "push" and not setup_push():
        exchange = "p2p"
    if exchange != "auto":
        return exchange
    transports = getattr(engine, "transports", ("p2p", "allgather"))
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
            try:
                hp.collective = True
                probe = [torch.rand_like(y) for y in ys0]
                hp.exchange_only(probe, first)
                engine.sync()
                good = all([halo_checksums(halo["plan"], t, y) for t, y in zip(first, probe)])   # a list: every collective runs
            except Exception as e:  # noqa: BLE001
                good = False
                sys.stderr.write(f"[bench] all_to_all form of the need-aware exchange unavailable on rank {job.rank}: {e!r}\n")
            if job.agree(good):
                cand["halo_a2a"] = job.timed_s(lambda: hp.exchange_only(ys0, first))
            else:
                info["halo_a2a_rejected"] = True
            hp.collective = False
    exchange = min(cand, key=cand.get)
    info["exchange_candidates_ms"] = {k: round(v * 1e3, 3) for k, v in cand.items()}
    # opt-in (SGL_BENCH_TRY_PUSH=1): a fault in a peer store would take the whole job down
    if os.environ.get("SGL_BENCH_TRY_PUSH", "0") != "1" or not handles or exchange.startswith("halo"):
        return exchange
    prop.transport = exchange
    if not setup_push():
        info["push_rejected"] = "mapping failed"
        return exchange
    ref_hops = prop.propagate_chunked(x_chunks, K, buffers=cbufs)
    got_hops = prop.propagate_push(x_chunks, K)
    same = True
    for a_, b_ in zip(ref_hops[K], got_hops[K]):
        scale_ = float(a_.abs().max()) if a_.numel() else 0.0
        same = same and (a_.numel() == 0 or float((a_ - b_).abs().max()) <= 1e-5 * max(scale_, 1e-30))
    if not prop.agree(same, device):
        info["push_rejected"] = "result mismatch"
        return exchange
    fullt = {exchange: job.timed_s(lambda: prop.propagate_chunked(x_chunks, K, buffers=cbufs)),
             "push": job.timed_s(lambda: prop.propagate_push(x_chunks, K))}
    info["full_step_candidates_ms"] = {k: round(v * 1e3, 3) for k, v in fullt.items()}
    return "push" if fullt["push"] < 0.97 * fullt[exchange] else exchange


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
    """The contract layout.  --col-chunks auto (default): how finely the feature block is cut for the pipelined exchange trades
    the un-overlapped head and tail of a step against per-chunk launch / issue cost (profiles/r03_scale_model.md: 4 chunks win
    in the model when the links are the bound, 2 when compute is), and that depends on what the links deliver -- so both are
    built, validated and timed (untimed setup; the exchange is selected once, with the first) and the faster is kept."""
    args = job.args
    auto = str(args.col_chunks) == "auto"
    counts = [2, 4] if (auto and job.world > 1 and job.nbuf > 0) else [2 if auto else int(args.col_chunks)]
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
            cand = _build_rows_for(job, ref, nc, exchange)
            exchange = job.info["exchange"]
            for k in ("exchange_candidates_ms", "push_peer_rows_skipped", "full_step_candidates_ms", "push_rejected",
                      "halo_a2a_rejected"):
                if k in live:
                    base_info[k] = live[k]                # the selection happens once: its record goes with every candidate
            if len(counts) > 1 or str(exchange).startswith("halo"):
                good = True
                try:
                    cand["step"]()
                    job.sync_all()
                    good = bool(cand["check"]())
                except Exception as e:  # noqa: BLE001
                    good = False
                    sys.stderr.write(f"[bench] rows with {nc} column chunks ({exchange}) failed on rank {job.rank}: {e!r}\n")
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
        best, timing, exchange = attempt("p2p")
    if best is None:
        raise RuntimeError("no column chunking of the row-sharded layout passed validation")
    nc, cand, info, inbound = best
    set_info(info)
    job.rows_inbound_bytes = inbound
    if timing:
        job.info["col_chunks_candidates_ms"] = {str(k): round(v * 1e3, 3) for k, v in timing.items()}
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
            return prop.propagate_chunked(tables, K, buffers=hbufs, y_buffers=ybufs)

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
            return [[t] for t in prop.propagate(x0, K, x_buffers=cbufs[0], y_buffers=ybufs[0])]
    else:
        def step():
            return prop.propagate_chunked(x_chunks, K, buffers=cbufs, y_buffers=ybufs)

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


def _alternatives(job):
    """which replica-based layouts --layout asks for besides the contract one"""
    world, layout = job.world, job.args.layout
    grid_ok = world >= 4 and world % 2 == 0
    if layout == "grid" and not grid_ok:
        raise SystemExit("--layout grid needs an even number of at least 4 ranks")
    if layout == "rows" or world == 1:
        return []
    if layout == "auto":
        return ["grid"] if (world >= 8 and grid_ok) else ["cols"]
    if layout == "all":
        return ["cols"] + (["grid"] if grid_ok else [])
    return [layout]


def _select_layout(job):
    """Build the contract layout (rows) first, then the alternatives --layout asks for while the setup budget lasts;
    validate each, time a full step (MAX over ranks), run the fastest.  The row-sharded figures are always reported.
    Returns (step, {layout: (propagator, x_chunks, buffers)} for the layouts that exchange rows)."""
    args, info, world = job.args, job.info, job.world
    alts = _alternatives(job)
    wanted = (["rows"] if args.layout in ("auto", "all", "rows") else []) + alts
    ref = None
    cands, timing, rejected, skipped = {}, {}, [], []
    for name in wanted:
        if name != "rows" and cands and not job.agree(job.budget_left() > 0):
            skipped.append(name)                          # out of setup budget: the contract layout is already in hand
            continue
        # a candidate that raises is dropped on EVERY rank (the code path is the same on all of them, so an error is
        # too; agree() keeps the control flow identical even if it is not)
        c, good = None, True
        _phase(f"select_layout: candidate {name!r}")
        try:
            if name != "rows" and ref is None:
                ref = _Reference(job)
            c = _build_rows(job) if name == "rows" else (_build_cols(job, ref) if name == "cols" else _build_grid(job, ref, 2))
            c["step"]()                                   # warm: plans, communicators, staging buffers
            job.sync_all()
            good = bool(c["check"]())
        except Exception as e:  # noqa: BLE001
            good = False
            sys.stderr.write(f"[bench] layout {name!r} failed on rank {job.rank}: {e!r}\n")
        if not job.agree(good):
            rejected.append(name)
            continue
        timing[name] = job.timed_s(c["step"], reps=3 if name == "rows" else 2, warm=0)
        cands[name] = c
    # Fallback rule (auto, >= 8 ranks): the communication-free feature-sharded layout is a known quantity -- every rank runs the
    # whole chain on d/N columns, measured at 0.26 of the single-GPU step for 8 ranks (profiles/r01_layout_shares.log).  It is
    # only built when neither exchanging layout beats that estimate (links slower than assumed), and the budget allows.
    if (args.layout == "auto" and "cols" not in wanted and ref is not None and timing and world >= 8
            and job.agree(min(timing.values()) * 1e3 > 0.26 * ref.ms and job.budget_left() > 0)):
        _phase("select_layout: fallback candidate 'cols'")
        c, good = None, True
        try:
            c = _build_cols(job, ref)
            c["step"]()
            job.sync_all()
            good = bool(c["check"]())
        except Exception as e:  # noqa: BLE001
            good = False
            sys.stderr.write(f"[bench] layout 'cols' failed on rank {job.rank}: {e!r}\n")
        if job.agree(good):
            timing["cols"] = job.timed_s(c["step"], reps=2, warm=0)
            cands["cols"] = c
            info["cols_fallback"] = "built because no exchanging layout beat the feature-sharded estimate"
        else:
            rejected.append("cols")
    if not cands:
        raise SystemExit(f"no multi-GPU layout passed validation (tried {wanted}, rejected {rejected})")
    chosen = min(timing, key=timing.get)
    info["layout"] = chosen
    info["contract_layout"] = "rows"
    info["layout_candidates_ms"] = {k: round(v * 1e3, 3) for k, v in timing.items()}
    if rejected:
        info["layout_rejected"] = rejected
    if skipped:
        info["layout_skipped_setup_budget"] = skipped
    if "rows" in timing:
        info["rows"] = {"ms_per_step": round(timing["rows"] * 1e3, 3),
                        "value": job.nnz * job.d * job.K / timing["rows"], "unit": "edge\u00b7featdim/s",
                        "parallelism": cands["rows"]["describe"], "exchange": info.get("exchange"),
                        "exchange_skipped_fraction": (info.get("halo") or {}).get("exchange_skipped_fraction_mean", 0.0)}
    info["alternatives"] = {k: round(v * 1e3, 3) for k, v in timing.items() if k != "rows"}
    info["parallelism"] = cands[chosen]["describe"] + ("" if chosen == "rows" else " [contract layout rows: see plan.rows]")
    halves = {name: c["halves"] for name, c in cands.items() if "halves" in c}
    if chosen == "rows":
        job.drop_full()                                   # nothing replica-based runs in the timed region
        ref = None
    return cands[chosen]["step"], halves


def _hop_halves(job, prop, x_chunks, cbufs, inbound):
    """the two halves of an exchanging layout's hop in isolation: SpMM only / exchange only (MAX over ranks)"""
    ys = prop.spmm_only(x_chunks)
    spmm_ms = job.timed_s(lambda: prop.spmm_only(x_chunks), reps=3) * 1e3
    xnext = [b[0] for b in cbufs]
    exch_ms = job.timed_s(lambda: prop.exchange_only(ys, xnext), reps=3) * 1e3 if prop._exchanging() else 0.0
    return {"spmm_only_ms_per_hop_max_rank": spmm_ms, "exchange_only_ms_per_hop_max_rank": exch_ms,
            "inbound_bytes_per_rank_per_hop": inbound,
            "exchange_inbound_GBps_per_rank": (inbound / (exch_ms * 1e-3) / 1e9) if exch_ms > 0 else None}


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


def _diagnostics(job, halves):
    """after the timed region, never part of `value`.  Row-sharded layout: its SpMM and all-gather halves and the
    achieved rate per link; grid layout: the same two halves of the relayed exchange (every byte crosses two links)."""
    diag = None
    if "rows" in halves and job.nbuf > 0:
        inbound = getattr(job, "rows_inbound_bytes", None) or (job.world - 1) / job.world * job.n * job.d * 4
        diag = _hop_halves(job, *halves["rows"], inbound)
        ms = diag["exchange_only_ms_per_hop_max_rank"]
        diag["exchange_GBps_per_link"] = (inbound / max(job.world - 1, 1) / (ms * 1e-3) / 1e9) if ms > 0 else None
        prop = halves["rows"][0]
        if hasattr(prop, "pack_only"):                        # need-aware exchange: the pack kernel alone (inside exchange_only too)
            ys = prop.spmm_only(halves["rows"][1])
            diag["pack_only_ms_per_hop_max_rank"] = job.timed_s(lambda: prop.pack_only(ys), reps=3) * 1e3
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


WORKLOAD_TEXT = {
    "S0": "SGC prop_steps={K} pre-propagation on a Pubmed-sized Chung-Lu graph (BASELINE config 1), LaplacianGraphOp r=0.5",
    "S1": "SGC prop_steps={K} pre-propagation on an ogbn-products-shaped Chung-Lu graph (BASELINE config 2), LaplacianGraphOp r=0.5",
    "S2": "GAMLP label-reuse sized propagation (d=147, prop_steps={K}) on the ogbn-products-shaped graph (BASELINE config 3), "
          "LaplacianGraphOp r=0.5",
    "S3_papers_shard": "one rank's 1/8 row block of an ogbn-papers100M-shaped hashed graph against the full 111 M x 128 feature "
                       "replica (BASELINE configs 4/5, per-GPU share of the 8-GPU job), {K} hop launch(es) per step",
    "S3": "prop_steps={K} propagation on an ogbn-papers100M-shaped hashed graph (BASELINE configs 4/5; directed, generated per "
          "row block on device, values used as A_hat directly: throughput only)",
}


def workload_text(name, K):
    for key in sorted(WORKLOAD_TEXT, key=len, reverse=True):
        if name.startswith(key):
            return f"{name}: " + WORKLOAD_TEXT[key].format(K=K)
    return f"{name}: prop_steps={K} pre-propagation (test workload)"


def _replayed_profile(workload, world):
    """rocprofv3 figures of the same command kept under profiles/ (PMC counters cannot be collected inside the timed run)"""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tfile))
    except Exception:  # noqa: BLE001
        return None
    if "workload" in tj:                              # round-1 layout: a single entry
        tj = {tj["workload"]: tj}
    return tj.get(workload) if world == 1 else None


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
    wl = dict(synthetic.WORKLOADS["S3_papers"], k=3) if wl is None else wl      # (tests pass a small hashed workload)
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


_PHASE = ["start"]


def _phase(name):
    _PHASE[0] = name


def run(args, engine_cls=GpuEngine, workloads=None, emit=print):
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
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
    if world > 1 and emit is print and args.watchdog > 0:
        # a rank that fails before a collective leaves the others waiting in it for ever: end the job with a diagnosis
        import threading

        def stuck():
            if rank == 0:
                quiet.unmute()
                print(json.dumps({"metric": baseline_metric(), "value": None, "unit": "edge\u00b7featdim/s", "n_gpus": world,
                                  "error": f"watchdog: no progress after {args.watchdog:.0f} s", "phase": _PHASE[0]}), flush=True)
            os._exit(3)
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
    t0 = time.perf_counter()
    t_start()
    for _ in range(args.steps):
        step()
    t_stop()
    job.sync_all()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    gpu_ms = t_elapsed_ms()

    _phase("diagnostics")
    try:
        diag = _diagnostics(job, halves) if job.budget_left() > -60 else {"skipped": "setup budget exhausted"}
    except Exception as e:  # noqa: BLE001  (reporting only; the measured value is already in hand)
        diag = {"failed": repr(e)}

    ceiling = None
    if not sharded and rank == 0 and hasattr(engine, "gather_ceiling"):
        ceiling = engine.gather_ceiling(job.col, job.x0, d)

    cpu = None
    if not sharded and not args.no_cpu_baseline and rank == 0:
        try:
            cpu = cpu_baseline(job.rowptr, job.col, job.val, job.x0, d)
        except Exception as e:  # noqa: BLE001  (baseline is reporting only; never blocks the GPU number)
            cpu = {"value": None, "unit": "edge\u00b7featdim/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

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
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(args.workload, K),
                       "n_nodes": n, "nnz_a_hat": nnz, "feat_dim": d, "prop_steps": K,
                       "parallelism": info.get("parallelism", "single GPU") if sharded else "single GPU",
                       "summation": "strict (no row splitting: bit-exact reference order)" if args.strict else "reference order per row; rows > 2048 nnz split into pieces",
                       "plan": info, "setup_s": round(setup_s, 2), "diagnostics": diag},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_BYTES, "traffic": traffic,
                         # PMC counters need their own rocprofv3 passes: the figure is the builder's pass over this very
                         # command, replayed from the file named here -- not a measurement of this run
                         "traffic_source": (prof.get("source", "profiles/traffic.json") + " (rocprofv3 --pmc pass of the same "
                                            "command, replayed; not measured in this run)") if prof else None,
                         "kernel": "spmm_kernel", "algorithmic_bytes_per_launch": alg,
                         "avg_launch_ms": hop_s * 1e3,
                         "kernel_ms_profile": prof.get("kernel_avg_ms_rocprof") if prof else None,
                         "traffic_frac": (traffic / hop_s / HBM_PEAK_BYTES) if traffic else None,
                         # the bare-gather ceiling of THIS access pattern measured in this run (probe kernel: the same
                         # column ids, row width and pitch; no CSR stream, arithmetic or stores) and the kernel's gather rate
                         "gather_ceiling_Ggathers_per_s": ceiling,
                         "kernel_Ggathers_per_s": nnz / world / hop_s / 1e9,
                         "frac_of_gather_ceiling": (nnz / world / hop_s / 1e9 / ceiling) if ceiling else None},
            "cpu_baseline": cpu,
        }

    # ---- secondary: the papers100M-shaped graph on the same ranks (bounded; never endangers the line above) -------------
    def emit_line():
        if rank == 0:
            quiet.unmute()
            emit(json.dumps(out))
            sys.stdout.flush()
            if emit is print:
                quiet.mute()                  # late library chatter (communicator teardown) goes to stderr too

    want_papers = (args.workload == "S1_products" and not args.no_papers and not args.force_sharded
                   and hasattr(engine, "hashed_block"))
    if want_papers:
        import threading
        _phase("papers100M section")
        done = threading.Event()

        def overrun():
            if done.is_set():
                return
            if rank == 0:
                out["papers100M"] = {"skipped": f"did not finish within the {args.papers_budget:.0f} s watchdog"}
                emit_line()
            os._exit(0)                       # a collective may be stuck: do not wait for it

        timer = threading.Timer(args.papers_budget, overrun)
        timer.daemon = True
        timer.start()
        try:
            exchange = info.get("exchange", "p2p")
            args.col_chunks_chosen = getattr(job, "col_chunks_chosen", None)
            del step, halves
            job.drop_full()
            job.block = job.x0 = None
            import gc
            gc.collect()
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
            papers = papers_section(args, engine, rank, world, exchange)
        except Exception as e:  # noqa: BLE001
            import traceback
            papers = {"failed": repr(e)[:200], "where": traceback.format_exc()[-700:]}
        done.set()
        timer.cancel()
        if rank == 0:
            out["papers100M"] = papers
    emit_line()
    if guard is not None:
        guard.cancel()
    if world > 1:
        dist.barrier()
    if job.own_group and dist.is_initialized():
        dist.destroy_process_group()
    return out


def main():
    run(parse_args())


if __name__ == "__main__":
    main()
