"""world_size-2/3 gloo tests (CPU) of the row-sharded propagation orchestration in sgl_amd/dist/:
shard arithmetic, piece-wise point-to-point all-gather, buffer ping-pong.  The local SpMM is injected (the CPU
oracle stands in for the HIP kernel here ONLY because this is a test of the exchange logic, which is
device-agnostic); the GPU kernels themselves are covered by tests/test_gpu_parity.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, pieces, K, d, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import oracle
    from inputs import hash_matrix
    from sgl_amd.dist import ShardedPropagator, balanced_bounds, piece_bounds

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", "graphs.npz")))
        indptr, indices, data = g["pl2000|indptr"], g["pl2000|indices"], g["pl2000|data"]
        n = len(indptr) - 1
        ptr, col, val = oracle.sym_norm_csr(indptr, indices, data, n, 0.5)
        val = val.astype(np.float32)
        bounds = balanced_bounds(ptr, world)
        pb = np.stack([piece_bounds(ptr, int(bounds[r]), int(bounds[r + 1]), pieces) for r in range(world)])

        def make_piece(r0, r1):
            rp = (ptr[r0:r1 + 1] - ptr[r0]).astype(np.int64)
            nb, ne = int(ptr[r0]), int(ptr[r1])
            c, v = col[nb:ne], val[nb:ne]

            def f(x, out):
                out.copy_(torch.from_numpy(oracle.oracle_spmm(rp, c, v, x.numpy(), n_rows=r1 - r0)))
            return f

        fns = [make_piece(int(pb[rank, p]), int(pb[rank, p + 1])) for p in range(pieces)]
        prop = ShardedPropagator(fns, pb, rank, world, n)
        x = torch.from_numpy(hash_matrix(n, d, seed=3))
        hops = prop.propagate(x, K)
        ref = oracle.propagate((ptr, col, val), x.numpy(), K)
        lo, hi = int(pb[rank, 0]), int(pb[rank, -1])
        ok = len(hops) == K + 1 and all(np.array_equal(hops[h].numpy(), ref[h][lo:hi]) for h in range(K + 1))
        # second call reuses nothing stale (buffers are ping-ponged per call)
        hops2 = prop.propagate(x, K)
        ok = ok and all(torch.equal(a, b) for a, b in zip(hops, hops2))
        # column-chunked, software-pipelined schedule gives the same numbers (SpMM is separable over columns)
        from sgl_amd.dist import column_chunks
        for chunks in (column_chunks(d, 2), [(0, 3), (3, 4), (4, d)]):
            xs = [x[:, a:b].contiguous() for a, b in chunks]
            hc = prop.propagate_chunked(xs, K)
            for h in range(K + 1):
                got = torch.cat(hc[h], dim=1)
                ok = ok and np.array_equal(got.numpy(), ref[h][lo:hi])
        # the collective transport (padded all_gather_into_tensor) is interchangeable with the send/recv one
        prop_ag = ShardedPropagator(fns, pb, rank, world, n, transport="allgather")
        hag = prop_ag.propagate(x, K)
        ok = ok and all(np.array_equal(hag[h].numpy(), ref[h][lo:hi]) for h in range(K + 1))
        hcg = prop_ag.propagate_chunked([x[:, :3].contiguous(), x[:, 3:].contiguous()], K)
        ok = ok and all(np.array_equal(torch.cat(hcg[h], 1).numpy(), ref[h][lo:hi]) for h in range(K + 1))
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,pieces,K", [(2, 4, 3), (2, 1, 1), (3, 2, 4)])
def test_sharded_propagation_matches_single_process(tmp_path, world, pieces, K):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, pieces, K, 8, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def _grid_worker(rank, world, port, row_groups, pieces, K, d, transport, out_dir, graph="pl2000"):
    """grid job: rank (rg, cg) multiplies row block rg with column slice cg; exchanges stay inside the column group
    (direct) or are spread over every rank of the job (relay)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import oracle
    from inputs import hash_matrix
    from sgl_amd.dist import GridLayout, ShardedPropagator, all_piece_bounds, column_slices

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", "graphs.npz")))
        indptr, indices, data = g[graph + "|indptr"], g[graph + "|indices"], g[graph + "|data"]
        n = len(indptr) - 1
        ptr, col, val = oracle.sym_norm_csr(indptr, indices, data, n, 0.5)
        val = val.astype(np.float32)
        layout = GridLayout(world, row_groups)
        rg, cg = layout.coords(rank)
        slices = column_slices(d, layout.col_groups)
        pb = all_piece_bounds(ptr, row_groups, pieces)

        def make_piece(r0, r1):
            rp = (ptr[r0:r1 + 1] - ptr[r0]).astype(np.int64)
            nb, ne = int(ptr[r0]), int(ptr[r1])
            c, v = col[nb:ne], val[nb:ne]

            def f(x, out):
                out.copy_(torch.from_numpy(oracle.oracle_spmm(rp, c, v, x.numpy(), n_rows=r1 - r0)))
            return f

        fns = [make_piece(int(pb[rg, p]), int(pb[rg, p + 1])) for p in range(pieces)]
        widths = [b - a for a, b in slices]
        collective = transport == "relay_all_to_all"
        if collective:
            # the RCCL fast path posts each relay phase as ONE all_to_all; gloo has no all_to_all, so emulate its
            # contract (tensor q of `ins` arrives as tensor `rank` of rank q's `outs`, sizes must match exactly)
            def all_to_all(outs, ins, group=None, async_op=False):
                outs[rank].copy_(ins[rank])
                ops = []
                for k in range(1, world):
                    dst, src = (rank + k) % world, (rank - k) % world
                    ops += [dist.P2POp(dist.isend, ins[dst], dst), dist.P2POp(dist.irecv, outs[src], src)]
                works = dist.batch_isend_irecv(ops)

                class Work:
                    def wait(self):
                        for w in works:
                            w.wait()
                return Work()
            dist.all_to_all = all_to_all
            transport = "relay"
        prop = ShardedPropagator(fns, pb, rg, row_groups, n, transport=transport, layout=layout, me=rank, widths=widths)
        prop.relay_collective = collective
        x = torch.from_numpy(hash_matrix(n, d, seed=5))
        a, b = slices[cg]
        ref = oracle.propagate((ptr, col, val), x.numpy(), K)
        ok = True
        for _ in range(2):                                    # second call: relay buffers and replicas are recycled
            hops = prop.propagate(x[:, a:b].contiguous(), K)
            ok = ok and len(hops) == K + 1 and all(
                np.array_equal(hops[h].numpy(), ref[h][prop.lo:prop.hi, a:b]) for h in range(K + 1))
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,row_groups,pieces,K,transport", [
    (4, 2, 2, 3, "relay"), (4, 2, 2, 3, "p2p"), (6, 3, 2, 3, "relay"), (8, 2, 3, 2, "relay"), (4, 4, 1, 2, "relay"),
    (3, 1, 2, 2, "p2p"), (6, 2, 1, 2, "relay"), (4, 2, 2, 2, "staged"), (4, 2, 2, 2, "relay_staged"),
    (4, 2, 2, 3, "relay_all_to_all"), (8, 2, 3, 2, "relay_all_to_all")])
def test_grid_layouts_match_single_process(tmp_path, world, row_groups, pieces, K, transport):
    port = _free_port()
    mp.spawn(_grid_worker, args=(world, port, row_groups, pieces, K, 11, transport, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


@pytest.mark.parametrize("transport", ["relay", "relay_all_to_all", "p2p"])
def test_grid_with_more_stripes_than_rows(tmp_path, transport):
    """40 rows, 8 ranks, 3 pieces: most stripes (and some whole pieces) are empty -- both sides must skip the same
    transfers, d = 3 < column groups leaves a column group without columns"""
    port = _free_port()
    mp.spawn(_grid_worker, args=(8, port, 2, 3, 2, 3, transport, str(tmp_path), "dir40"), nprocs=8, join=True)
    for r in range(8):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def test_grid_layout_arithmetic():
    from sgl_amd.dist import GridLayout, column_slices
    L = GridLayout(8, 2)
    assert (L.row_groups, L.col_groups) == (2, 4) and L.members(3) == [6, 7] and L.coords(5) == (1, 2)
    assert sorted(g for cg in range(4) for g in L.members(cg)) == list(range(8))
    with pytest.raises(ValueError):
        GridLayout(8, 3)
    assert column_slices(100, 8) == [(0, 13), (13, 26), (26, 39), (39, 52), (52, 64), (64, 76), (76, 88), (88, 100)]
    assert column_slices(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)] and column_slices(100, 1) == [(0, 100)]
    # whole 128-byte lines per slice wherever there are at least as many lines as parts
    assert column_slices(100, 4) == [(0, 32), (32, 64), (64, 96), (96, 100)] and column_slices(100, 2) == [(0, 64), (64, 100)]
    assert column_slices(128, 4) == [(0, 32), (32, 64), (64, 96), (96, 128)] and column_slices(11, 2) == [(0, 6), (6, 11)]
    for d in range(1, 300):
        for parts in range(1, 9):
            sl = column_slices(d, parts)
            assert len(sl) == parts and sl[0][0] == 0 and sl[-1][1] == d and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            assert max(-(-(b - a) // 32) for a, b in sl) == -(-(-(-d // parts)) // 32) or parts > -(-d // 32)


def _halo_worker(rank, world, port, K, d, out_dir, graph, bounds_override):
    """need-aware exchange: compact tables [own rows | ghosts per peer], relabelled columns, packed sends.  Every hop must be
    bit-equal to the single-process chain, ghosts must equal the owners' rows, and the exchanged volume must be what the
    plan says (rows nobody gathers never travel)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import oracle
    from inputs import hash_matrix
    from sgl_amd.dist import HaloPlan, HaloPropagator, balanced_bounds, halo, halo_checksums
    halo._CHUNK = 777            # the marking / relabelling loops run in many passes (at papers100M size they do: 1.7 G non-zeros per rank)

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", "graphs.npz")))
        indptr, indices, data = g[graph + "|indptr"], g[graph + "|indices"], g[graph + "|data"]
        n = len(indptr) - 1
        ptr, col, val = oracle.sym_norm_csr(indptr, indices, data, n, 0.5)
        val = val.astype(np.float32)
        bounds = np.asarray(bounds_override, dtype=np.int64) if bounds_override is not None else balanced_bounds(ptr, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        nb, ne = int(ptr[lo]), int(ptr[hi])
        rp = (ptr[lo:hi + 1] - ptr[lo]).astype(np.int64)
        c_glob = torch.from_numpy(col[nb:ne].astype(np.int32))
        plan = HaloPlan(lo, hi, n, c_glob, bounds)
        c_comp = plan.relabel(c_glob).numpy()
        v = val[nb:ne]
        # the offline planner (cost models on one device) reproduces the collective plan of this rank exactly
        off = HaloPlan.offline(rank, bounds, n, lambda q: torch.from_numpy(col[int(ptr[bounds[q]]):int(ptr[bounds[q + 1]])].astype(np.int32)))
        same_plan = all(torch.equal(a, b) for a, b in zip(off.need, plan.need)) and off.ghost_off == plan.ghost_off and \
            all(torch.equal(a, b) for a, b in zip(off.send_rows, plan.send_rows)) and torch.equal(off.send_idx, plan.send_idx) and \
            off.send_off == plan.send_off and np.array_equal(off.counts[rank], plan.counts[rank]) and \
            np.array_equal(off.counts[:, rank], plan.counts[:, rank])

        def spmm(x, out):
            out.copy_(torch.from_numpy(oracle.oracle_spmm(rp, c_comp, v, x.numpy(), n_rows=hi - lo)))

        prop = HaloPropagator(plan, spmm)
        x = torch.from_numpy(hash_matrix(n, d, seed=7))
        ref = oracle.propagate((ptr, col, val), x.numpy(), K)
        ok = same_plan and plan.n_compact == plan.n_own + plan.n_ghost and plan.n_ghost <= n - plan.n_own
        # the plan is exact: ghosts == the distinct foreign columns of my block
        foreign = np.unique(col[nb:ne][(col[nb:ne] < lo) | (col[nb:ne] >= hi)])
        ok = ok and plan.n_ghost == len(foreign) and np.array_equal(plan.global_ids.numpy()[plan.n_own:], foreign)
        # hop 0 from my OWN rows only (ghosts fetched) == cut out of the full matrix
        t_own = prop.table_from_own(x[lo:hi].contiguous())
        t_full = prop.table_from_full(x)
        ok = ok and torch.equal(t_own, t_full) and torch.equal(t_full, x[plan.global_ids])
        for in_place in (False, True):
            hops = prop.propagate(t_own.clone(), K, in_place=in_place)
            ok = ok and len(hops) == K + 1 and np.array_equal(hops[K].numpy(), ref[K][lo:hi])
            if not in_place:
                ok = ok and all(np.array_equal(hops[h].numpy(), ref[h][lo:hi]) for h in range(K + 1))
        # caller-owned tables, one per exchanged hop: the hop matrices ARE the own rows of those tables (nothing copied, all kept)
        bufs = [torch.empty_like(t_own) for _ in range(K - 1)]
        hops_v = prop.propagate(t_own.clone(), K, buffers=bufs, hops_in_buffers=True)
        ok = ok and all(np.array_equal(hops_v[h].numpy(), ref[h][lo:hi]) for h in range(K + 1))
        ok = ok and all(hops_v[h + 1].data_ptr() == bufs[h].data_ptr() for h in range(K - 1) if plan.n_own)
        # ... only when asked for: without the flag the hops are separate matrices whatever buffers are passed (a caller that reuses
        # its buffers across calls keeps its earlier results), and the flag without enough buffers is an error
        hops_c = prop.propagate(t_own.clone(), K, buffers=bufs)
        ok = ok and all(np.array_equal(hops_c[h].numpy(), ref[h][lo:hi]) for h in range(K + 1))
        ok = ok and all(hops_c[h + 1].data_ptr() != bufs[h].data_ptr() for h in range(K - 1) if plan.n_own)
        if K >= 3:
            try:
                prop.propagate(t_own.clone(), K, buffers=bufs[:1], hops_in_buffers=True)
                ok = False
            except ValueError:
                pass
        # the pack step in own-row order (every row read once, written to each peer's share) fills the same send buffer
        ok = ok and torch.equal(plan.send_idx[plan.pack_dst], plan.pack_src) and bool((plan.pack_src[1:] >= plan.pack_src[:-1]).all())
        prop.pack_mode = "scatter"
        hops_s = prop.propagate(t_own.clone(), K)
        prop.pack_mode = "auto"
        ok = ok and np.array_equal(hops_s[K].numpy(), ref[K][lo:hi])
        # the same exchange as ONE all_to_all_single with split sizes (the RCCL form; gloo has none, so its contract is emulated
        # with point-to-point operations: rows [sum(in[:q]), +in[q]) of the input go to rank q, rows from rank q land at
        # [sum(out[:q]), +out[q]) of the output)
        def a2a_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
            ok_shapes = sum(output_split_sizes) == output.shape[0] and sum(input_split_sizes) == input.shape[0]
            assert ok_shapes and output.is_contiguous() and input.is_contiguous()
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
        real_a2a = getattr(dist, "all_to_all_single")
        dist.all_to_all_single = a2a_single
        prop.collective = True
        try:
            hops_c = prop.propagate(t_own.clone(), K)
            ok = ok and all(np.array_equal(hops_c[h].numpy(), ref[h][lo:hi]) for h in range(K + 1))
        finally:
            prop.collective = False
            dist.all_to_all_single = real_a2a
        # column chunks, software-pipelined, with caller-owned tables: the ghosts of the last exchanged hop are the owners' rows
        chunks = [(0, 3), (3, d)]
        tabs = [t_own[:, a:b].contiguous() for a, b in chunks]
        bufs = [[torch.empty_like(t) for _ in range(2)] for t in tabs]
        hc = prop.propagate_chunked(tabs, K, buffers=bufs)
        for h in range(K + 1):
            ok = ok and np.array_equal(torch.cat(hc[h], 1).numpy(), ref[h][lo:hi])
        if K >= 2:
            for ci, (a, b) in enumerate(chunks):
                last_table = bufs[ci][(K - 2) % 2]
                ok = ok and np.array_equal(last_table.numpy(), ref[K - 1][plan.global_ids.numpy(), a:b])
                ok = ok and halo_checksums(plan, last_table, hc[K - 1][ci])
            bad = bufs[0][(K - 2) % 2].clone()
            if plan.n_ghost:
                bad[plan.n_own, 0] += 1.0
            flags = [None] * world
            dist.all_gather_object(flags, bool(halo_checksums(plan, bad, hc[K - 1][0])))
            ok = ok and (flags[rank] == (plan.n_ghost == 0))
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
        with open(os.path.join(out_dir, f"skip{rank}.txt"), "w") as f:
            f.write(repr(plan.describe()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,K,graph,bounds", [
    (2, 3, "pl2000", None), (3, 2, "pl2000", None), (4, 3, "pl2000", None), (8, 2, "pl2000", None),
    (3, 3, "dir40", [0, 0, 25, 40]),          # an empty block: its rank packs nothing, gathers nothing, still takes part
    (4, 1, "dir40", None), (2, 0, "dir40", None)])
def test_halo_exchange_matches_single_process(tmp_path, world, K, graph, bounds):
    port = _free_port()
    mp.spawn(_halo_worker, args=(world, port, K, 9, str(tmp_path), graph, bounds), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def _redistribute_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from sgl_amd.dist import RowBlock, balanced_bounds, exchange_var, sharded_community_order
    from sgl_amd.dist.redistribute import sharded_edge_locality
    from sgl_amd.reorder import community_order_reference, edge_locality

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        ok = True
        # variable-size exchange: rank r sends q + 2 r rows of its own marker to rank q (some shares empty)
        sends = [torch.full(((q + 2 * rank) % 4, 3), float(10 * rank + q)) for q in range(world)]
        got = exchange_var(sends)
        for q in range(world):
            want = torch.full(((rank + 2 * q) % 4, 3), float(10 * q + rank))
            ok = ok and got[q].shape == want.shape and torch.equal(got[q], want)
        ints = exchange_var([torch.arange(q + 1, dtype=torch.int64) + 100 * rank for q in range(world)])
        ok = ok and all(torch.equal(ints[q], torch.arange(rank + 1, dtype=torch.int64) + 100 * q) for q in range(world))
        # label propagation over row blocks == the same algorithm on the whole matrix (a graph with planted communities whose ids
        # are shuffled), for unequal blocks incl. an EMPTY one
        import scipy.sparse as sp
        rng = np.random.default_rng(0)
        n, bs = 1500, 60
        a = np.repeat(np.arange(n), 8)
        near = np.minimum((a // bs) * bs + rng.integers(0, bs, a.size), n - 1)
        b = np.where(rng.random(a.size) < 0.9, near, rng.integers(0, n, a.size))
        m = sp.coo_matrix((np.ones(a.size, np.float32), (a, b)), shape=(n, n)).tocsr()
        m = ((m + m.T + sp.eye(n)) > 0).astype(np.float32).tocsr()
        shuffle = np.random.default_rng(1).permutation(n)
        P = sp.coo_matrix((np.ones(n, np.float32), (shuffle, np.arange(n))), shape=(n, n)).tocsr()
        m = (P @ m @ P.T).tocsr()
        m.sort_indices()
        rp, cc = torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int32))
        want_order, _ = community_order_reference(rp, cc, n)
        bounds = [0, 400, 400, n] if world == 3 else [int(v) for v in balanced_bounds(m.indptr.astype(np.int64), world)]
        lo, hi = bounds[rank], bounds[rank + 1]
        blk = RowBlock(lo, hi, n, (rp[lo:hi + 1] - rp[lo]).contiguous(), cc[int(rp[lo]):int(rp[hi])].contiguous(),
                       torch.from_numpy(m.data[int(rp[lo]):int(rp[hi])].copy()))
        order, text = sharded_community_order(blk, bounds)
        ok = ok and torch.equal(order, want_order) and "communities after 8 rounds" in text
        ok = ok and abs(sharded_edge_locality(blk, None) - edge_locality(rp, cc)) < 1e-12
        ok = ok and abs(sharded_edge_locality(blk, order) - edge_locality(rp, cc, want_order)) < 1e-12
        ok = ok and edge_locality(rp, cc, want_order) > edge_locality(rp, cc) + 0.3          # the ordering found the communities
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_relabelling_is_found_on_row_blocks(world, tmp_path):
    """sgl_amd/dist/redistribute.py on CPU / gloo: the variable-size exchange, and the community order found by label propagation
    over the ranks' row blocks (one integer per node replicated, never the matrix) equals the whole-matrix algorithm's.  The row
    redistribution itself builds its blocks with the HIP COO -> CSR kernel: covered in the gpu suite with two real processes."""
    mp.spawn(_redistribute_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"rank{r}.txt").read() for r in range(world)] == ["ok"] * world
