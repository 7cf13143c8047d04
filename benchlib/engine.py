"""Everything device-specific in the bench (GpuEngine) and the CPU baseline leg."""
import os
import time

import numpy as np
import torch

from .common import ROOT  # noqa: F401  (puts the repo root on sys.path)


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
    t0 = time.perf_counter()
    fn()
    t_probe = time.perf_counter() - t0
    whole = ""
    nnz_all = int(rowptr[-1])
    if rows < n and not compacted and t_probe * nnz_all / max(nnz_s, 1) * 4 <= budget_s:
        # ONE WHOLE HOP fits the budget (warm-up + three repetitions): the figure is then no extrapolation from the first rows --
        # the static schedule hands every thread one contiguous block of rows, whose cost the first 16 % need not represent
        rows, rp = n, rowptr.cpu().numpy()
        nnz_s = nnz_all
        c, v = col.cpu().numpy(), val.cpu().numpy()
        fn = (lambda: oracle.reference_spmm(rp, c, v, xh, n_rows=rows)) if kind == "reference" else \
             (lambda: oracle.oracle_spmm(rp, c, v, xh, n_rows=rows))
        fn()  # warm-up of the full-size output
        whole = "ONE WHOLE HOP: "
    times = []
    t_all = time.perf_counter()
    for _ in range(3 if whole else 5):
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
           "sample": f"{whole}{'all' if whole else 'first'} {rows} rows of A_hat ({nnz_s} nnz) x d={d}{compacted}, one hop, median of {len(times)} reps "
                     f"({min(times) * 1e3:.0f} ... {max(times) * 1e3:.0f} ms), "
                     f"OpenMP static schedule, {threads} threads on {cores} physical cores of {cpu_model}",
           "ms_per_hop_sample": t * 1e3}
    # the build's OWN restatement of the same loop (oracle/spmm_ref.c: bit-equal to the reference binary, tests/test_oracle_golden.py)
    # beside it whenever the headline figure is the reference's compiled kernel: both travel, the line says which is which
    if kind == "reference":
        try:
            oracle.oracle_spmm(rp, c, v, xh, n_rows=rows)
            t0 = time.perf_counter()
            oracle.oracle_spmm(rp, c, v, xh, n_rows=rows)
            tp = time.perf_counter() - t0
            out["port"] = {"value": nnz_s * d / tp, "unit": "edge\u00b7featdim/s", "cores": cores, "threads": threads, "kind": "port",
                           "sample": "the same rows through oracle/liboracle_spmm.so (this repository's C restatement), one rep"}
        except Exception as e:  # noqa: BLE001
            out["port"] = {"value": None, "sample": f"failed: {e}"}
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
        if wl.get("community"):          # the same degree law with planted communities and shuffled ids (S1_community)
            cm = wl["community"]
            return synthetic.planted_partition_torch(wl["n"], wl["m"], wl["d_max"], cm["block"], cm["p_in"], seed=args.seed, device=self.device,
                                                     weight=2.0 if getattr(args, "dup2", False) else 1.0)[:3]
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
        _lib.check_probe(_lib.probe_lib().sgl_synth_degrees(ctypes.c_uint64(args.seed), 0, n, _lib.ptr(tab), _lib.ptr(deg),
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
        from sgl_amd.dist.sharded_adj import global_nnz
        return block_piece_spmms(blk, pieces, weights, strict=args.strict, total_nnz=global_nnz(blk))

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
                _lib.check_probe(_lib.probe_lib().sgl_probe_gather_f32(_lib.ptr(x), ld, _lib.ptr(idx), idx.numel(), rf, 16, _lib.ptr(sink),
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

    def validate_single(self, samples=4096, tol=1e-5):
        """Self-check of the single-GPU line, run AFTER the timed region on what the timed steps left behind: (1) `samples` rows of
        the last hop recomputed in fp64 with plain torch indexing from the matrix the last launch read (kernel-independent);
        (2) the same launch repeated in strict (reference) summation order into a scratch matrix: row-wise L2 and max-norm
        distance of the timed kernel's result from it, both against the SURVEY 8(c) tolerance.  Returns the dict that goes to
        config.validation; its "ok" is config.validated."""
        from sgl_amd import device as dev
        from sgl_amd.dist import RowBlock
        st = getattr(self, "_single", None)
        if st is None:
            return {"ok": None, "skipped": "no single-GPU step was built"}
        rowptr, col, val, d = st["rowptr"], st["col"], st["val"], st["d"]
        x_prev, y_last = st["x_prev"], st["y_last"]
        blk = RowBlock(0, rowptr.numel() - 1, st["n"], rowptr, col, val)
        ok_rows = bool(self.sampled_rows_check(blk, x_prev[:, :d], y_last[:, :d], samples=samples, tol=tol))
        out = {"sampled_rows": min(samples, blk.n_local), "sampled_rows_fp64_ok": ok_rows, "tolerance": tol}
        ok = ok_rows
        free, _ = torch.cuda.mem_get_info()
        if st["strict"]:
            out["strict_vs_fast"] = "the timed kernel already ran in strict order"
        elif y_last.numel() * 4 * 1.2 > free:
            out["strict_vs_fast"] = "skipped: no room for a scratch hop matrix"
        else:
            ref = torch.empty_like(y_last)
            dev.DeviceCSR(rowptr, col, val, (blk.n_local, st["n"]), strict=True).spmm(x_prev, out=ref)
            dn = (y_last[:, :d] - ref[:, :d]).norm(dim=1)
            rn = ref[:, :d].norm(dim=1).clamp_min(1e-30)
            out["strict_vs_fast_max_row_rel_l2"] = float((dn / rn).max())
            out["strict_vs_fast_max_abs_rel"] = float((y_last[:, :d] - ref[:, :d]).abs().max() / ref[:, :d].abs().max().clamp_min(1e-30))
            ok = ok and out["strict_vs_fast_max_row_rel_l2"] <= tol and out["strict_vs_fast_max_abs_rel"] <= tol
            del ref
        out["ok"] = bool(ok)
        return out

    def single_step(self, args, rowptr, col, val, x0, n, d, K):
        from sgl_amd import device as dev
        reorder_info = None
        if getattr(args, "reorder", None) and rowptr.numel() - 1 == n:
            # plan-time locality ordering (GraphOp(reorder=...)): rows stored and processed community by community behind a row map;
            # ids, X, Y and every row's term order untouched -- the validation below still recomputes rows of the ORIGINAL matrix
            from sgl_amd.reorder import plan_rowmap
            rowmap, reorder_info = plan_rowmap(rowptr, col, n, args.reorder)
            if rowmap is not None:
                rp2, c2, v2 = dev.permute_rows(rowptr, col, val, rowmap)
                csr = dev.DeviceCSR(rp2, c2, v2, (n, n), strict=args.strict).set_rowmap(rowmap)
            else:
                csr = dev.DeviceCSR(rowptr, col, val, (n, n), strict=args.strict)
        else:
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
        if reorder_info is not None:
            info["reorder"] = reorder_info
        if pingpong:
            info["hops_retained"] = "last two only (K hop matrices of this size do not fit one GPU)"
        # what validate_single() looks at after the timed region: the operand and the result of the LAST launch of a step
        self._single = {"rowptr": rowptr, "col": col, "val": val, "d": d, "n": n, "strict": bool(args.strict),
                        "x_prev": x_in if (K == 1 or n_out != n) else outs[K - 2], "y_last": outs[K - 1]}
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

        def start():
            ev0.record()
        start.event = ev0
        return start, (lambda: ev1.record()), (lambda: ev0.elapsed_time(ev1))

    def step_marks(self, n_steps):
        """one event per timed step on the stream the kernels are launched on (torch's current stream = what current_stream_ptr()
        hands the library): the per-step durations inside the timed region"""
        class Marks:
            def __init__(self):
                self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps)]

            def mark(self, i):
                self.ev[i].record()

            def durations_ms(self, start_event):
                out, prev = [], start_event
                for e in self.ev:
                    if prev is not None:
                        out.append(prev.elapsed_time(e))
                    prev = e
                return out
        return Marks()


class OneGpuGlooEngine(GpuEngine):
    """REHEARSAL engine: every rank on cuda:0, process group = gloo (which cannot move device memory, so the exchange is the
    host-staged transport).  The kernels, the row-sharded storage, the need-aware plan, the validation and the JSON line are
    the real ones; only the wire differs.  Selected by SGL_BENCH_ENGINE=one_gpu_gloo -- it exists so that an N-rank launch of
    bench.py can be run end to end on a one-GPU box (tests/test_zz_gpu_multiprocess.py, profiles/r05_launch_rehearsal.log)."""
    backend = "gloo"
    transports = ("staged",)
    relay_transport = "relay_staged"
    probe_links = False                   # the link micro-benchmark moves device tensors through the process group
    halo_collective = False               # gloo has no all_to_all_single

    def __init__(self, local_rank):
        super().__init__(0)

    def init_kwargs(self):
        return {}


def engine_from_env():
    """SGL_BENCH_ENGINE: unset / "gpu" = one rank per GPU over RCCL (the product path); "one_gpu_gloo" = the rehearsal engine"""
    import os
    name = os.environ.get("SGL_BENCH_ENGINE", "gpu")
    if name in ("", "gpu"):
        return GpuEngine
    if name == "one_gpu_gloo":
        return OneGpuGlooEngine
    raise SystemExit(f"SGL_BENCH_ENGINE={name!r}: expected gpu or one_gpu_gloo")
