"""bench.py's distributed orchestration (replica broadcast, shard bounds, piece exchange, barrier + MAX-over-ranks
timing, the one-line JSON contract) exercised with 2 gloo ranks on CPU.  The compute engine is substituted
(CPU oracle SpMM instead of the HIP kernels) -- the GPU engine itself is covered by `-m gpu` tests and by the
driver's bench runs; what cannot be run here, an N>1 launch, is exactly what this test pins down."""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TINY = {"T_tiny": dict(n=3000, m=20_000, d_max=300, d=40, k=3),
        "T_wide": dict(n=2000, m=12_000, d_max=200, d=100, k=3)}      # d = 100: 2 / 3 / 4 column chunks are three different cuts


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_engine():
    import time

    import oracle
    from sgl_amd.synthetic import chung_lu_numpy

    class CpuEngine:
        backend = "gloo"
        transports = ("p2p", "allgather")

        def __init__(self, local_rank):
            self.device = torch.device("cpu")

        def init_kwargs(self):
            return {}

        def build_raw(self, args, wl):
            ip, ix, dt = chung_lu_numpy(wl["n"], wl["m"], wl["d_max"], seed=args.seed)
            return torch.from_numpy(ip), torch.from_numpy(ix.astype(np.int32)), torch.from_numpy(dt.astype(np.float32))

        def features(self, args, wl):
            return torch.from_numpy(np.random.default_rng(0).standard_normal((wl["n"], wl["d"])).astype(np.float32))

        def build_workload(self, args, wl):
            ip, ix, dt = (t.numpy() for t in self.build_raw(args, wl))
            ptr, col, val = oracle.laplacian_adj(ip, ix, dt, wl["n"], 0.5)
            return (torch.from_numpy(ptr), torch.from_numpy(col.astype(np.int32)),
                    torch.from_numpy(val.astype(np.float32)), self.features(args, wl))

        def normalize_block(self, blk, r=0.5, alpha=None, symmetric=True):
            """numpy restatement of sgl_norm_block_* for a symmetric unit-weight block: rows of A + I, degrees by
            all-reduce, (T'[j,i] * L[j]) * R[i] rounded to fp32"""
            from sgl_amd.dist import RowBlock
            rp, cc, vv = blk.rowptr.numpy(), blk.col.numpy(), blk.val.numpy()
            rows = np.repeat(np.arange(blk.n_local), np.diff(rp))
            import scipy.sparse as sp
            t = sp.csr_matrix((vv.astype(np.float64), (rows, cc)), shape=(blk.n_local, blk.n))
            eye = sp.csr_matrix((np.ones(blk.n_local), (np.arange(blk.n_local), np.arange(blk.lo, blk.hi))), shape=t.shape)
            t = (t + eye).tocsr()
            t.sort_indices()
            deg = torch.zeros(blk.n, dtype=torch.float64)
            deg[blk.lo:blk.hi] = torch.from_numpy(np.asarray(t.sum(1)).ravel())
            if dist.is_initialized():
                dist.all_reduce(deg)
            dg = deg.numpy()
            left, right = np.power(dg, r - 1), np.power(dg, -r)
            vals = (t.data * np.repeat(left[blk.lo:blk.hi], np.diff(t.indptr))) * right[t.indices]
            return RowBlock(blk.lo, blk.hi, blk.n, torch.from_numpy(t.indptr.astype(np.int64)),
                            torch.from_numpy(t.indices.astype(np.int32)), torch.from_numpy(vals.astype(np.float32)))

        def block_piece_spmms(self, args, blk, pieces, weights=None):
            from sgl_amd.dist import local_piece_bounds
            pb, rp_host = local_piece_bounds(blk, pieces, weights)
            fns = []
            for p in range(pieces):
                r0, r1 = int(pb[p]) - blk.lo, int(pb[p + 1]) - blk.lo
                rp = (rp_host[r0:r1 + 1] - rp_host[r0]).astype(np.int64)
                c, v = blk.col[int(rp_host[r0]):int(rp_host[r1])].numpy(), blk.val[int(rp_host[r0]):int(rp_host[r1])].numpy()
                fns.append(lambda x, out, rp=rp, c=c, v=v, rows=r1 - r0:
                           out.copy_(torch.from_numpy(oracle.oracle_spmm(rp, c, v, x.numpy(), n_rows=rows))))
            return fns, None, pb

        def block_halo(self, args, blk, bounds):
            from sgl_amd.dist import HaloPlan, HaloPropagator, RowBlock
            plan = HaloPlan(blk.lo, blk.hi, blk.n, blk.col, bounds)
            ccol = plan.relabel(blk.col)
            rp, c, v = blk.rowptr.numpy(), ccol.numpy(), blk.val.numpy()
            prop = HaloPropagator(plan, lambda x, out: out.copy_(
                torch.from_numpy(oracle.oracle_spmm(rp, c, v, x.numpy(), n_rows=blk.n_local))))
            return plan, prop, RowBlock(blk.lo, blk.hi, plan.n_compact, blk.rowptr, ccol, blk.val)

        def sampled_rows_check(self, blk, x_prev, y_local, samples=64, tol=1e-5):
            rows = np.random.default_rng(blk.lo).integers(0, max(blk.n_local, 1), size=min(samples, blk.n_local))
            rp, c, v = blk.rowptr.numpy(), blk.col.numpy(), blk.val.numpy()
            for r in rows:
                sl = slice(int(rp[r]), int(rp[r + 1]))
                want = (v[sl, None].astype(np.float64) * x_prev.numpy()[c[sl]].astype(np.float64)).sum(0)
                mag = (np.abs(v[sl, None].astype(np.float64)) * np.abs(x_prev.numpy()[c[sl]])).sum(0)
                if not (np.abs(y_local.numpy()[r] - want) <= tol * np.maximum(mag, 1e-30) + 1e-30).all():
                    return False
            return True

        def single_step(self, args, rowptr, col, val, x0, n, d, K):
            rp, c, v = rowptr.numpy(), col.numpy(), val.numpy()

            def step():
                cur = x0.numpy()
                for _ in range(K):
                    cur = oracle.oracle_spmm(rp, c, v, cur)
                self.last = cur
            return step, {"n_items": 0}

        def piece_spmms(self, args, rowptr, col, val, n, my_bounds, rp_host):
            fns = []
            for p in range(len(my_bounds) - 1):
                r0, r1 = int(my_bounds[p]), int(my_bounds[p + 1])
                rp = (rp_host[r0:r1 + 1] - rp_host[r0]).astype(np.int64)
                nb, ne = int(rp_host[r0]), int(rp_host[r1])
                c, v = col[nb:ne].numpy(), val[nb:ne].numpy()
                fns.append(lambda x, out, rp=rp, c=c, v=v, rows=r1 - r0:
                           out.copy_(torch.from_numpy(oracle.oracle_spmm(rp, c, v, x.numpy(), n_rows=rows))))
            return fns, None

        def sync(self):
            pass

        def timer(self):
            t = {}
            return (lambda: t.__setitem__("a", time.perf_counter())), (lambda: t.__setitem__("b", time.perf_counter())), \
                   (lambda: (t["b"] - t["a"]) * 1e3)

    return CpuEngine


def _emulate_all_to_all_single(rank, world):
    """gloo has no all_to_all; give torch.distributed the contract of all_to_all_single with split sizes through point-to-point
    operations so that the one-call form of the need-aware exchange runs in the CPU orchestration tests"""
    def a2a_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        assert sum(output_split_sizes) == output.shape[0] and sum(input_split_sizes) == input.shape[0]
        oi, oo = np.concatenate([[0], np.cumsum(input_split_sizes)]), np.concatenate([[0], np.cumsum(output_split_sizes)])
        ops = []
        for k in range(1, world):
            dst, src = (rank + k) % world, (rank - k) % world
            if input_split_sizes[dst]:
                ops.append(dist.P2POp(dist.isend, input[oi[dst]:oi[dst + 1]], dst))
            if output_split_sizes[src]:
                ops.append(dist.P2POp(dist.irecv, output[oo[src]:oo[src + 1]], src))
        works = dist.batch_isend_irecv(ops) if ops else []

        class Work:
            def wait(self):
                for w_ in works:
                    w_.wait()
        return Work()
    dist.all_to_all_single = a2a_single


def _worker(rank, world, port, out_dir, extra=(), a2a=False, break_halo=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import bench
    if a2a:
        _emulate_all_to_all_single(rank, world)
    args = bench.parse_args(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--workload", "T_tiny",
                             "--pieces", "3", "--no-cpu-baseline", *extra])
    lines = []
    engine_cls = _make_engine()
    engine_cls.halo_collective = bool(a2a)
    if break_halo:
        # a need-aware exchange that delivers wrong rows on one rank (a transport that misbehaves on some system): the exact
        # checksums must catch it and the run must go on with the full-replica exchange
        good_halo = engine_cls.block_halo

        def bad_halo(self, args_, blk, bounds):
            plan, prop, cblk = good_halo(self, args_, blk, bounds)
            if break_halo == "raise":
                raise RuntimeError("need-aware exchange unavailable on this system (test)")
            if rank == 1:
                real = prop.begin_exchange

                def corrupt(y_own, table_next, key=0):
                    w = real(y_own, table_next, key)

                    class W:
                        def wait(self_inner):
                            w.wait()
                            if plan.n_ghost:
                                table_next[plan.n_own, 0] += 1.0
                    return W()
                prop.begin_exchange = corrupt
            return plan, prop, cblk
        engine_cls.block_halo = bad_halo
    out = bench.run(args, engine_cls=engine_cls, workloads=TINY, emit=lines.append)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"lines": lines, "returned": out is not None, "initialized_after": dist.is_initialized()}, f)


def test_bench_two_ranks_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--layout", "auto", "--exchange", "auto", "--col-chunks", "2")),
             nprocs=world, join=True)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert len(r0["lines"]) == 1 and r1["lines"] == [] and not r0["initialized_after"]
    j = json.loads(r0["lines"][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "strong"
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["unit"] == "edge\u00b7featdim/s" and j["metric"].startswith("pre-prop SpMM throughput") and j["vs_baseline"] is None
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] and j["cpu_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["config"]["plan"]["exchange"] in ("p2p", "allgather", "halo")       # opt-in selection: whichever won is valid
    assert set(j["config"]["plan"]["exchange_candidates_ms"]) == {"p2p", "allgather", "halo"}
    assert j["config"]["plan"]["exchange_selected_by"].startswith("measured exchange-only time")
    assert j["config"]["validated"] is True and j["config"]["validation"]["when"].startswith("setup")
    dg = j["config"]["diagnostics"]
    assert dg["spmm_only_ms_per_hop_max_rank"] > 0 and dg["exchange_only_ms_per_hop_max_rank"] > 0
    assert dg["exchange_inbound_GBps_per_rank"] > 0
    plan = j["config"]["plan"]
    assert set(plan["layout_candidates_ms"]) == {"cols", "rows"} and plan["layout"] in ("cols", "rows")
    assert "layout_rejected" not in plan and plan["layout"] == min(plan["layout_candidates_ms"], key=plan["layout_candidates_ms"].get)
    assert j["config"]["parallelism"].startswith("feature-sharded" if plan["layout"] == "cols" else "row-sharded")
    # the contract layout is always reported, whatever runs; A_hat is stored as one row block per rank
    assert plan["contract_layout"] == "rows" and plan["rows"]["value"] > 0 and plan["rows"]["ms_per_step"] > 0
    assert set(plan["alternatives"]) == {"cols"} and plan["rows"]["exchange"] == plan["exchange"]
    assert 0.0 <= plan["rows"]["exchange_skipped_fraction"] < 1.0
    assert plan["rows"]["parallelism"].startswith("row-sharded x2 (A_hat stored as one row block per GPU)")
    assert plan["adjacency_storage"].startswith("row block per rank") and j["config"]["setup_s"] < 120
    assert j["config"]["workload"].startswith("T_tiny: prop_steps=3")
    assert j["roofline"]["traffic"] is None and j["roofline"]["traffic_source"] is None and "papers100M" not in j


def test_bench_rows_only_keeps_no_replica_and_budget_skips(tmp_path):
    """default flags: the contract layout with the FIXED need-aware exchange and 3 column chunks -- nothing is selected by a timing
    race, nothing replica-based is ever built; the line explains itself per hop (SpMM / pack / wire ms, rate on the busiest link,
    overlap, the schedule model's prediction beside the measured step); with --layout auto a zero setup budget skips the
    alternative and says so; --exchange halo / p2p run exactly that transport"""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--workload", "T_wide")), nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    plan = j["config"]["plan"]
    assert list(plan["layout_candidates_ms"]) == ["rows"] and plan["layout"] == "rows" and "adjacency_replicated_for" not in plan
    assert plan["alternatives"] == {} and j["config"]["parallelism"].startswith("row-sharded x2")
    assert abs(j["value"] - plan["rows"]["value"]) / j["value"] < 0.9          # the headline IS the row-sharded job (separate timings)
    # the deterministic default: no candidates, no selection records
    assert plan["exchange"] == "halo" and plan["col_chunks"] == [[0, 32], [32, 64], [64, 100]]
    for key in ("exchange_candidates_ms", "exchange_selected_by", "col_chunks_candidates_ms", "col_chunks_selected_by",
                "full_step_candidates_ms", "halo_rejected"):
        assert key not in plan, key
    ph = j["config"]["diagnostics"]["per_hop"]
    assert ph["spmm_ms"] > 0 and ph["pack_ms"] > 0 and ph["exchange_wire_ms"] >= 0 and 0.0 <= ph["overlap_fraction"] <= 1.0
    assert len(ph["per_chunk"]["spmm_ms"]) == 3 and all(b > 0 for b in ph["per_chunk"]["busiest_link_bytes"])
    assert ph["link_GBps_per_direction_busiest_link"] is None or ph["link_GBps_per_direction_busiest_link"] > 0
    m = ph["model"]
    assert m["measured_ms_per_step"] == j["ms_per_step"] and m["predicted_ms_per_step_at_measured_rates"] > 0
    assert m["predicted_ms_per_step_free_links"] <= m["predicted_ms_per_step_at_measured_rates"] + 1e-9
    assert set(m["predicted_ms_per_step_by_link_GBps"]) == {"35", "45", "55", "65", "76.8"}
    for ex in ("halo", "p2p"):
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--exchange", ex, "--col-chunks", "2")), nprocs=world, join=True)
        j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
        plan = j["config"]["plan"]
        assert plan["exchange"] == ex and "exchange_candidates_ms" not in plan and plan["layout"] == "rows"
        assert ("halo" in plan) == (ex == "halo")
        dg = j["config"]["diagnostics"]
        assert dg["exchange_only_ms_per_hop_max_rank"] > 0 and ("pack_only_ms_per_hop_max_rank" in dg) == (ex == "halo")
        assert (dg["per_hop"]["pack_ms"] > 0) == (ex == "halo")
        if ex == "halo":
            h = plan["halo"]
            assert h["compact_rows"] == h["own_rows"] + h["ghost_rows"] and 0 <= h["exchange_skipped_fraction"] < 1
            assert plan["rows"]["exchange_skipped_fraction"] == h["exchange_skipped_fraction_mean"]
            assert "need-aware all-gather (halo" in j["config"]["parallelism"]
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--layout", "auto", "--setup-budget", "0")), nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    plan = j["config"]["plan"]
    assert plan["layout"] == "rows" and plan["layout_skipped_setup_budget"] == ["cols"] and "adjacency_replicated_for" not in plan


def test_bench_four_ranks_gloo_all_layouts(tmp_path):
    """N = 4, --layout all: row-sharded (contract), feature-sharded and the relayed 2 x 2 grid are all built, validated and
    timed; the default tries rows + one alternative only; an explicit --layout runs just that one"""
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--layout", "all", "--grid-pieces", "2,4,8")), nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    plan = j["config"]["plan"]
    assert set(plan["layout_candidates_ms"]) == {"cols", "rows", "grid"} and "layout_rejected" not in plan
    assert j["n_gpus"] == 4 and j["value"] > 0 and plan["rows"]["value"] > 0
    assert plan["adjacency_replicated_for"].startswith("alternative layout candidates")
    assert set(plan["grid_pieces_candidates_ms"]) == {"2", "4", "8"} and plan["grid_pieces"] in (2, 4, 8)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--layout", "auto")), nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    assert set(j["config"]["plan"]["layout_candidates_ms"]) == {"rows", "cols"}
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--layout", "grid", "--grid-pieces", "2")), nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    assert j["config"]["plan"]["layout"] == "grid" and j["config"]["parallelism"].startswith("grid 2 row blocks x 2 column slices")
    assert j["config"]["plan"]["grid_pieces"] == 2 and list(j["config"]["plan"]["grid_pieces_candidates_ms"]) == ["2"]
    g = j["config"]["diagnostics"]["grid"]          # only the grid was built: its two halves, no row-sharded entry
    assert g["spmm_only_ms_per_hop_max_rank"] > 0 and g["exchange_only_ms_per_hop_max_rank"] > 0 and g["relay_GBps_per_link"] > 0
    assert "spmm_only_ms_per_hop_max_rank" not in j["config"]["diagnostics"]


def test_bench_eight_ranks_gloo_default_is_the_contract_layout(tmp_path):
    """the driver's largest launch shape: 8 ranks, default flags: the headline is the row-sharded job (A_hat stored per rank,
    per-hop need-aware all-gather), on a FIXED path -- halo exchange, 3 column chunks, nothing chosen by timing -- and the
    per-hop breakdown with the model's prediction for this G is in the line"""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--workload", "T_wide")), nprocs=world, join=True)
    lines = [json.load(open(tmp_path / f"rank{r}.json"))["lines"] for r in range(world)]
    assert len(lines[0]) == 1 and all(l == [] for l in lines[1:])
    j = json.loads(lines[0][0])
    plan = j["config"]["plan"]
    from sgl_amd.dist import column_chunks
    assert plan["exchange"] == "halo" and plan["col_chunks"] == [list(c) for c in column_chunks(100, 3)]
    for key in ("exchange_candidates_ms", "col_chunks_candidates_ms", "halo_rejected", "push_rejected"):
        assert key not in plan, key
    assert j["n_gpus"] == 8 and plan["layout"] == "rows" and plan["contract_layout"] == "rows" and plan["alternatives"] == {}
    assert list(plan["layout_candidates_ms"]) == ["rows"] and "adjacency_replicated_for" not in plan
    assert plan["rows"]["exchange"] == "halo" and 0.0 <= plan["rows"]["exchange_skipped_fraction"] < 1.0
    assert j["config"]["parallelism"].startswith("row-sharded x8 (A_hat stored as one row block per GPU)")
    dg = j["config"]["diagnostics"]
    assert dg["exchange_GBps_per_link"] > 0 and j["value"] > 0 and j["config"]["validated"] is True
    ph = dg["per_hop"]
    for key in ("spmm_ms", "pack_ms", "exchange_wire_ms", "per_chunk", "link_GBps_per_direction_busiest_link", "link_frac_of_xgmi_peak",
                "overlap_fraction", "exposed_exchange_ms_per_step", "model"):
        assert key in ph, key
    for key in ("predicted_ms_per_step_at_measured_rates", "measured_ms_per_step", "predicted_ms_per_step_free_links",
                "predicted_ms_per_step_by_link_GBps", "schedule"):
        assert key in ph["model"], key


def test_bench_eight_ranks_gloo_opt_in_selection(tmp_path):
    """--exchange auto --col-chunks auto (opt-in): the transport is chosen among halo / p2p / allgather by their measured exchange
    time and the pipelining granularity by measured step time; whatever wins was validated, and the line says how it was chosen"""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--col-chunks", "auto", "--exchange", "auto")), nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    plan = j["config"]["plan"]
    # the distinct cuts of 2 / 3 / 4 column chunks are validated and timed (d = 40 here: every count cuts 32 + 8, so one remains)
    assert set(plan["col_chunks_candidates_ms"]) == {"2"} and plan["col_chunks_selected_by"].startswith("measured step time")
    from sgl_amd.dist import column_chunks
    chosen = int(min(plan["col_chunks_candidates_ms"], key=plan["col_chunks_candidates_ms"].get))
    assert plan["col_chunks"] == [list(c) for c in column_chunks(40, chosen)]
    assert set(plan["exchange_candidates_ms"]) == {"p2p", "allgather", "halo"} and plan["exchange"] in plan["exchange_candidates_ms"]
    assert plan["rows"]["exchange"] == plan["exchange"] and j["value"] > 0


def test_bench_eight_ranks_gloo(tmp_path):
    """8 ranks, --layout auto: the contract layout + the 2 x 4 grid as the opt-in alternative"""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--layout", "auto")), nprocs=world, join=True)
    lines = [json.load(open(tmp_path / f"rank{r}.json"))["lines"] for r in range(world)]
    assert len(lines[0]) == 1 and all(l == [] for l in lines[1:])
    j = json.loads(lines[0][0])
    plan = j["config"]["plan"]
    # rows (contract) + the grid; the communication-free layout joins ONLY through the fallback rule (no exchanging layout beat
    # its estimate -- always the case with gloo on CPU), and says so
    assert j["n_gpus"] == 8 and set(plan["layout_candidates_ms"]) - {"cols"} == {"rows", "grid"} and "layout_rejected" not in plan
    assert ("cols" in plan["layout_candidates_ms"]) == ("cols_fallback" in plan)
    assert plan["rows"]["value"] > 0 and plan["contract_layout"] == "rows" and list(plan["grid_pieces_candidates_ms"]) == ["4"]
    assert j["config"]["diagnostics"]["exchange_GBps_per_link"] > 0
    links = j["config"]["diagnostics"]["links"]     # gloo: the pairwise exchange works, all_to_all may not exist
    assert links["pair_exchange_64MB_GBps_per_direction"] > 0 and "all_to_all_1MB_per_peer_GBps_per_link" in links


def test_bench_single_rank_contract(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    import bench
    args = bench.parse_args(["--steps", "2", "--warmup", "1", "--workload", "T_tiny", "--no-cpu-baseline"])
    lines = []
    bench.run(args, engine_cls=_make_engine(), workloads=TINY, emit=lines.append)
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["config"]["parallelism"] == "single GPU" and j["value"] > 0
    # default CLI values finish quickly and match the documented contract
    d = bench.parse_args([])
    assert d.gpus == 1 and d.steps == 10 and d.warmup == 2 and d.workload == "S1_products"


def test_cpu_baseline_leg_reports_reference_kernel_cores_and_scipy():
    """bench.cpu_baseline on a small workload: the reference's own compiled kernel (oracle/_ref) when it is built, physical
    cores AND threads stated, the scipy single-thread figure present (a regression here once went unnoticed: the leg is
    reporting-only and swallows its own exceptions)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import bench
    import oracle
    from sgl_amd.synthetic import chung_lu_numpy
    n = 30_000
    ip, ix, dt = chung_lu_numpy(n, 200_000, 400, seed=1)
    ptr, col, val = oracle.laplacian_adj(ip, ix, dt, n, 0.5)
    out = bench.cpu_baseline(torch.from_numpy(ptr), torch.from_numpy(col.astype(np.int32)), torch.from_numpy(val.astype(np.float32)),
                             torch.randn(n, 16), 16, budget_s=1.0)
    assert out["value"] > 0 and out["kind"] in ("reference", "port") and out["unit"] == "edge\u00b7featdim/s"
    assert 1 <= out["cores"] <= out["threads"] and "physical cores" in out["sample"]
    assert out["scipy_dot"]["value"] and out["scipy_dot"]["cores"] == 1
    # the reference's compiled kernel and this repository's restatement of it are both reported, each under its own kind
    assert (out["kind"] == "reference") == ("port" in out) and (out["kind"] != "reference" or out["port"]["value"] > 0)


def test_bench_need_aware_exchange_as_one_all_to_all(tmp_path):
    """the one-call form of the need-aware exchange (all_to_all_single with split sizes; emulated over gloo): validated by exact
    checksums on random data before it may be timed, a candidate of the selection, and runnable on request through both column
    chunk candidates"""
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--col-chunks", "auto", "--exchange", "auto"), True), nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    plan = j["config"]["plan"]
    assert set(plan["exchange_candidates_ms"]) == {"p2p", "allgather", "halo", "halo_a2a"} and "halo_a2a_rejected" not in plan
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path),
                            ("--col-chunks", "auto", "--exchange", "halo_a2a", "--workload", "T_wide"), True), nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    plan = j["config"]["plan"]
    assert plan["exchange"] == "halo_a2a" and set(plan["col_chunks_candidates_ms"]) == {"2", "3", "4"} and j["value"] > 0
    from sgl_amd.dist import column_chunks
    chosen = int(min(plan["col_chunks_candidates_ms"], key=plan["col_chunks_candidates_ms"].get))
    assert plan["col_chunks"] == [list(c) for c in column_chunks(100, chosen)]
    assert "halo as one all_to_all_single" in j["config"]["parallelism"] and plan["rows"]["exchange"] == "halo_a2a"


def test_bench_falls_back_when_the_need_aware_exchange_misdelivers(tmp_path):
    """a need-aware exchange that corrupts one ghost value on one rank is caught by the exact checksums (on every chunking) and
    the job continues, validated, on the full-replica exchange -- and says so"""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ("--exchange", "halo", "--col-chunks", "auto"), False, True),
             nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    plan = j["config"]["plan"]
    assert plan["exchange"] == "p2p" and plan["halo_rejected"].startswith("halo: validation failed") and j["value"] > 0
    assert "halo" not in plan and plan["rows"]["exchange"] == "p2p" and plan["rows"]["exchange_skipped_fraction"] == 0.0


def test_bench_default_falls_back_when_the_need_aware_exchange_cannot_be_built(tmp_path):
    """default flags, and the need-aware exchange raises while it is built: every rank agrees, the job runs -- validated -- on the
    full-replica p2p exchange and the line says so"""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), (), False, "raise"), nprocs=world, join=True)
    j = json.loads(json.load(open(tmp_path / "rank0.json"))["lines"][0])
    plan = j["config"]["plan"]
    assert plan["exchange"] == "p2p" and plan["halo_rejected"].startswith("halo") and j["value"] > 0 and plan["layout"] == "rows"
    assert j["config"]["validated"] is True and j["config"]["diagnostics"]["per_hop"]["pack_ms"] == 0


# ---- `python bench.py --gpus N` without a launcher (benchlib/launch.py) ------------------------------------------------------------
_STUB = r'''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_RANK"] == str(rank)
mode = sys.argv[1]
print("library chatter of rank", rank)                      # only rank 0's JSON line may reach the parent's stdout
if mode == "ok":
    if rank == 0:
        print(json.dumps({"value": 1.0, "n_gpus": world}))
elif mode == "rank1_dies":
    if rank == 1:
        sys.stderr.write("boom on rank 1\n"); sys.exit(7)
    time.sleep(120)                                          # the survivors wait in a "collective" for ever
elif mode == "hangs":
    time.sleep(120)                                          # nobody dies, nobody finishes
elif mode == "dies_after_line":
    if rank == 0:
        print(json.dumps({"value": 2.0, "n_gpus": world}), flush=True)
    if rank == 1:
        time.sleep(0.5); sys.exit(5)
'''


def _launch(tmp_path, mode, n=3, grace=1.0):
    sys.path.insert(0, ROOT)
    import contextlib
    import io
    from benchlib import launch
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB)

    class A:
        gpus = n
    buf = io.StringIO()
    os.environ["SGL_BENCH_ENGINE"] = "one_gpu_gloo"           # (no device-count probe: the stub needs no GPU)
    try:
        with contextlib.redirect_stdout(buf):
            rc = launch.self_launch(A, [mode], str(stub), "m", grace_s=grace)
    finally:
        del os.environ["SGL_BENCH_ENGINE"]
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    return rc, lines


def test_self_launch_relays_exactly_one_line(tmp_path):
    rc, lines = _launch(tmp_path, "ok")
    assert rc == 0 and len(lines) == 1 and json.loads(lines[0]) == {"value": 1.0, "n_gpus": 3}


def test_self_launch_ends_the_job_when_a_rank_dies(tmp_path):
    """rank 1 exits 7 while the others sit in a collective: the parent ends them after the grace period, prints ONE line with
    value null that names the rank, its code and its stderr, and returns non-zero"""
    import time
    t0 = time.monotonic()
    rc, lines = _launch(tmp_path, "rank1_dies")
    assert time.monotonic() - t0 < 60
    j = json.loads(lines[0])
    assert rc != 0 and len(lines) == 1 and j["value"] is None and j["n_gpus"] == 3
    assert "rank 1 exited with code 7" in j["error"] and "boom on rank 1" in j["stderr_tail"]


def test_self_launch_ends_a_job_that_hangs(tmp_path, monkeypatch):
    """no rank dies and none finishes: after SGL_BENCH_LAUNCH_TIMEOUT the parent ends the ranks and prints ONE null line"""
    import time
    monkeypatch.setenv("SGL_BENCH_LAUNCH_TIMEOUT", "2")
    t0 = time.monotonic()
    rc, lines = _launch(tmp_path, "hangs", n=2)
    assert time.monotonic() - t0 < 60
    j = json.loads(lines[0])
    assert rc != 0 and len(lines) == 1 and j["value"] is None and "still running after 2 s" in j["error"]


def test_self_launch_keeps_a_measured_line_when_a_rank_fails_later(tmp_path):
    rc, lines = _launch(tmp_path, "dies_after_line", n=2)
    j = json.loads(lines[0])
    assert rc == 0 and len(lines) == 1 and j["value"] == 2.0 and j["launcher"]["failed_rank"] == 1


def test_bare_bench_command_with_more_ranks_than_gpus_prints_a_null_line():
    """`python bench.py --gpus 2` where fewer than 2 GPUs are visible (here: none): one JSON line, value null, non-zero rc --
    not a traceback, and not the SystemExit the bare command used to die with"""
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SGL_BENCH_ENGINE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert out.returncode != 0 and len(lines) == 1
    j = json.loads(lines[0])
    assert j["value"] is None and j["n_gpus"] == 2 and "GPU(s) visible" in j["error"]


# ---- secondary sections of the single-GPU line (benchlib/extras.py) ----------------------------------------------------------
def test_extras_runner_records_failures_and_budget_and_compacts(monkeypatch):
    """run_extras: a section that raises is recorded in place of its numbers, a section that would start after the budget is
    recorded as skipped, the others carry wall_s; compact_sections keeps the REQUIRED_KEYS of every section that ran and the
    failure text of those that did not -- none of it can touch the headline"""
    sys.path.insert(0, ROOT)
    import argparse
    import time
    from benchlib import extras

    def ok_section(name, cfg):
        return {"workload": f"{name}: long text", "short": "short text", "baseline_config": cfg, "ms": 1.23456789, "validated": True,
                "roofline": extras._roof(1e9, 1.0)}

    class Eng:
        def build_raw(self, args, wl):
            return ("raw",)
    monkeypatch.setattr(extras, "section_s0", lambda a, e: dict(ok_section("S0_pubmed", 1), ms_per_hop=0.4, value=1e12,
                        validation={"strict_order_all_hops_bit_equal": True, "fast_order": {"row_l2_rel": 2e-7}},
                        cpu_baseline={"kind": "port", "cores": 8, "normalise_ms": 1.0, "propagate_ms": 2.0, "value": 1e9}))
    monkeypatch.setattr(extras, "section_s2", lambda a, e, raw: (_ for _ in ()).throw(RuntimeError("boom")))
    monkeypatch.setattr(extras, "section_s4_products", lambda a, e, raw: (time.sleep(0.3), ok_section("S4_products", 5))[1])
    args = argparse.Namespace(seed=0, extras_scale="small")
    detail = {}
    extras.run_extras(args, Eng(), detail, budget_s=20.2, which=("S0_pubmed", "S2_gamlp", "S4_products", "S1_community"))
    sec = detail["sections"]
    assert sec["S0_pubmed"]["validated"] is True and "wall_s" in sec["S0_pubmed"]
    assert "boom" in sec["S2_gamlp"]["failed"] and "where" in sec["S2_gamlp"]
    assert "skipped" in sec["S1_community"] and "budget" in sec["S1_community"]["skipped"]        # 20.2 - 0.3 s < the 20 s a section needs
    assert "S4_papers_shard" not in sec                                                          # not asked for
    # S4_products' stub has no tables: compact what has the full shape
    c = extras.compact_sections({k: sec[k] for k in ("S0_pubmed", "S2_gamlp", "S1_community")})
    assert all(k in c["S0_pubmed"] for k in extras.REQUIRED_KEYS)
    assert c["S0_pubmed"]["ms"] == 1.235 and c["S0_pubmed"]["roofline"]["frac"] == extras._r(1e9 / 1e-3 / 8e12)
    assert c["S0_pubmed"]["workload"] == "S0_pubmed: short text" and c["S0_pubmed"]["cpu_baseline"]["cores"] == 8
    assert "failed" in c["S2_gamlp"] and "skipped" in c["S1_community"]


def test_recorded_bench_line_carries_every_baseline_config():
    """the bench line kept under profiles/ (the builder's run of the driver's command): sections for BASELINE configs 1, 3, 5 and the
    per-rank share of 4/5, each with workload / ms / roofline{frac, achieved, algorithmic_bytes_per_launch} / validated, the MessageOp
    tables, the CPU baselines -- the format a reader of BENCH_rNN.json relies on"""
    sys.path.insert(0, ROOT)
    from benchlib import extras
    path = os.path.join(ROOT, "profiles", "r06_bench_S1.json")
    assert os.path.exists(path), "profiles/r06_bench_S1.json is missing"
    j = json.loads(open(path).read().strip().splitlines()[-1])
    assert j["config"]["workload"].startswith("S1_products") and j["value"] > 0 and j["roofline"]["frac"] > 0.6
    sec = j["sections"]
    assert list(j)[-1] == "sections", "the sections must be the LAST key of the line"
    assert set(sec) == {"S0_pubmed", "S2_gamlp", "S4_products", "S1_community", "S4_papers_shard"}
    for name, s in sec.items():
        assert all(k in s for k in extras.REQUIRED_KEYS), (name, list(s))
        assert s["validated"] is True and s["ms"] > 0 and {"frac", "achieved", "algorithmic_bytes_per_launch"} <= set(s["roofline"]), name
    assert sec["S0_pubmed"]["baseline_config"] == 1 and sec["S0_pubmed"]["strict_order_bit_equal_to_cpu_oracle"] is True
    assert sec["S0_pubmed"]["cpu_baseline"]["kind"] in ("reference", "port") and sec["S0_pubmed"]["cpu_baseline"]["cores"] >= 1
    assert sec["S2_gamlp"]["baseline_config"] == 3 and len(sec["S2_gamlp"]["preprocess_calls_ms"]) == 3
    for name in ("S4_products", "S4_papers_shard"):
        ops = {r[0]: r for r in sec[name]["message_ops"]}
        assert len(ops) == 10 and all(r[3] is True for r in ops.values()), name                  # the nine ops of the search space + NAFS
        assert {"last", "concat", "mean", "sum", "max", "min"} <= set(ops) and any(k.startswith("nafs") for k in ops)
        assert all(g[4] is True for g in sec[name]["graph_ops"])
    assert len(sec["S4_products"]["graph_ops"]) == 4                                             # Laplacian + PPR alpha .1 / .2 / .3
    cb = sec["S4_products"]["cpu_baseline_combine"]
    assert set(cb["ms"]) == {"mean", "max", "concat", "nafs"} and cb["threads"] >= 1 and cb["rows"] > 0
    com = sec["S1_community"]
    assert com["reorder_auto"]["applied"] is True and com["reorder_auto"]["bit_identical"] is True
    assert com["reorder_auto"]["ms_per_hop"] < com["reorder_none"]["ms_per_hop"]
    assert len(json.dumps(sec)) < 6000, "the sections are meant to stay short enough to be read from the tail of the line"


def test_cpu_legs_of_the_secondary_sections():
    """benchlib/cpu_legs.py on small inputs: config 1's reference path returns a baseline AND a verdict on the candidate hops
    (a perturbed candidate fails, the oracle's own hops pass bit for bit); the torch-CPU combine leg reports all four aggregates"""
    sys.path.insert(0, ROOT)
    import oracle
    from benchlib import cpu_legs
    from sgl_amd.synthetic import chung_lu_numpy
    n, d, K = 3000, 24, 3
    ip, ix, dt = chung_lu_numpy(n, 20_000, 300, seed=2)
    x = np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)
    ref = oracle.propagate(oracle.laplacian_adj(ip, ix, dt, n, 0.5), x, K)
    cpu, chk = cpu_legs.config1_reference_path(ip, ix, dt, x, n, K, ref[K], ref, 1e-5)
    assert chk["fast_order"]["ok"] and chk["fast_order"]["bit_equal"] and chk["strict_order_all_hops_bit_equal"]
    assert cpu["value"] > 0 and cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["normalise_ms"] > 0
    bad = [h.copy() for h in ref]
    bad[2][5, 3] += 1e-3
    _, chk = cpu_legs.config1_reference_path(ip, ix, dt, x, n, K, ref[K] * (1 + 1e-4), bad, 1e-5)
    assert not chk["fast_order"]["ok"] and not chk["strict_order_all_hops_bit_equal"]
    hops = [torch.from_numpy(h) for h in ref]
    res = cpu_legs.combine_baseline(hops, d, rows_fast=2000, rows_loop=50)
    assert res["rows"] == 2000 and res["nafs_rows"] == 50 and all(res[k]["ms"] > 0 and res[k]["rows_per_s"] > 0 for k in ("mean", "max", "concat", "nafs"))


def test_planted_partition_generator_properties():
    """sgl_amd.synthetic.planted_partition_torch (workload S1_community): canonical symmetric CSR without self loops, the stated
    share of edges inside communities, and NO trace of the communities in the (shuffled) ids"""
    sys.path.insert(0, ROOT)
    from sgl_amd import synthetic as sy
    n, m, blk = 20_000, 200_000, 512
    rp, col, val, truth = sy.planted_partition_torch(n, m, 600, blk, 0.8, seed=3, device="cpu")
    rows = torch.repeat_interleave(torch.arange(n), rp[1:] - rp[:-1])
    c = col.long()
    assert int(rp[-1]) == col.numel() == val.numel() and bool((rows != c).all()) and bool((val == 1).all())
    k1, k2 = torch.sort(rows * n + c).values, torch.sort(c * n + rows).values
    assert torch.equal(k1, k2) and torch.unique(k1).numel() == k1.numel()                       # symmetric, no duplicates
    assert bool((torch.diff(rows * n + c) > 0).all())                                            # canonical: rows ascending, columns sorted
    inside = float((truth[rows] == truth[c]).float().mean())
    assert 0.7 < inside < 0.85
    near = float(((rows - c).abs() < blk).float().mean())
    assert near < 0.1                                                                            # ids are shuffled: neighbours are not id-neighbours
    rp2, col2, _, truth2 = sy.planted_partition_torch(n, m, 600, blk, 0.8, seed=3, device="cpu")
    assert torch.equal(rp, rp2) and torch.equal(col, col2) and torch.equal(truth, truth2)        # seeded
    assert int(truth.max()) == (n - 1) // blk and sy.WORKLOADS["S1_community"]["n"] == sy.WORKLOADS["S1_products"]["n"]


def test_a_crash_in_the_secondary_sections_cannot_cost_the_headline(monkeypatch):
    """bench.run: whatever happens after the timed region -- the section runner raising outright, a section failing, the renderer
    choking on a malformed section -- the ONE line is printed with the measured value, and says what went wrong"""
    sys.path.insert(0, ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    import bench
    from benchlib import extras

    class WithSections(_make_engine()):
        hashed_block = staticmethod(lambda *a, **k: None)        # (what marks an engine that can run the secondary sections)
    workloads = {"S1_products": TINY["T_tiny"]}
    args = bench.parse_args(["--steps", "2", "--warmup", "1", "--workload", "S1_products", "--no-cpu-baseline", "--no-papers"])

    def run(fn):
        monkeypatch.setattr(extras, "run_extras", fn)
        lines = []
        bench.run(args, engine_cls=WithSections, workloads=workloads, emit=lines.append)
        assert len(lines) == 1
        return json.loads(lines[0])

    def boom(*a, **k):
        raise RuntimeError("boom in the runner")
    j = run(boom)
    assert j["value"] > 0 and "boom in the runner" in j["secondary_sections_error"] and "sections" not in j

    def one_bad_section(args_, engine, detail, budget_s=0, which=None):
        sec = detail.setdefault("sections", {})
        for name in which:
            sec[name] = {"failed": "RuntimeError('section died')", "where": "traceback tail", "wall_s": 0.1}
    j = run(one_bad_section)
    assert j["value"] > 0 and list(j)[-1] == "sections" and "section died" in j["sections"]["S0_pubmed"]["failed"]
    assert set(j["sections"]) == {"S0_pubmed", "S2_gamlp", "S4_products", "S1_community", "S4_papers_shard"}

    def malformed(args_, engine, detail, budget_s=0, which=None):
        detail.setdefault("sections", {})["S0_pubmed"] = {"workload": "S0_pubmed: x"}          # no ms / roofline: the renderer raises
    j = run(malformed)
    assert j["value"] > 0 and "could not render" in j["sections"]["failed"]
    j = run(lambda *a, **k: None)                                                             # nothing produced: no key at all
    assert j["value"] > 0 and "sections" not in j
    args.no_extras = True
    assert "sections" not in run(boom) and "secondary_sections_error" not in run(boom)        # --no-extras: the runner is never called
