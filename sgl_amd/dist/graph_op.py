"""ShardedGraphOp: GraphOp.propagate for one rank of a multi-GPU job, in any layout."""
import numpy as np
import torch
import torch.distributed as dist

from .layout import GridLayout, all_piece_bounds, column_chunks, column_slices, device_piece_spmms
from .propagator import ShardedPropagator
from .sharded_adj import RowBlock, allgather_rows, block_piece_spmms, gather_piece_bounds


class ShardedGraphOp:
    """GraphOp.propagate for one rank of a multi-GPU job (BASELINE configs 4/5: NAFS / PaSca sweeps over the GPUs of
    a node).

    Every rank passes the SAME full adjacency (scipy CSR or sgl_amd.io.DeviceAdjacency; it is normalised on the
    rank's own GPU -- identical kernels on identical inputs, so all ranks hold bit-identical A_hat) and the same full
    feature matrix, and gets back the K+1 hop matrices restricted to ITS block: rows `[self.lo, self.hi)` x columns
    `[self.c0, self.c1)`.  `row_groups` picks the layout (see GridLayout): None = row-sharded over all ranks (the
    block is full-width; MessageOps are row-wise, so they apply to the local shards unchanged, e.g.
    OverSmoothDistanceWeightedOp for NAFS), 1 = feature-sharded (all rows, d/G columns, no communication at all;
    column-wise aggregators -- last/sum/mean/max/min/simple_weighted -- apply unchanged), anything between = grid.
    288 GB per GPU make the replication affordable up to ogbn-papers100M (27 GB of CSR, two 57 GB feature replicas).

    Row-sharded STORAGE (the contract layout of SURVEY 8(e)): pass a `RowBlock` instead -- this rank's rows
    [lo, hi) of the raw adjacency (of A^T when A is not symmetric: `symmetric=False`), local row pointers, global column
    ids.  The block is normalised in place of the whole matrix (sgl_norm_block_*: the only communication is the degree
    vector) and NO rank ever holds more than its own rows of A or A_hat; the feature argument may be the full matrix or
    just this rank's rows (then the replica is all-gathered once).  Only the row-sharded layout applies.

    Works without torch.distributed (world size 1); with it, uses the default process group unless `group` is given
    (row-sharded layout only: grid layouts address ranks of the default group)."""

    def __init__(self, prop_steps, r=0.5, alpha=None, pieces=2, col_chunks=2, strict_order=False, group=None,
                 device=None, row_groups=None, transport=None, symmetric=True, reorder=None, partition=None):
        self.prop_steps, self.r, self.alpha = prop_steps, r, alpha
        self.pieces, self.col_chunks, self.strict_order, self.group = pieces, col_chunks, strict_order, group
        self.device = device
        self.row_groups, self.transport = row_groups, transport
        self.symmetric = symmetric
        self.reorder = reorder                # None / "community" / "auto": RowBlock inputs only (rows ordered inside the rank's block)
        # None / "community" / "auto": full-adjacency inputs, row-sharded layout: the node ids are RELABELLED in a community order
        # before the matrix is cut into row blocks, so that a block references mostly its own rows and the need-aware exchange
        # moves a fraction of the foreign rows (50 % instead of 98 % at 8 ranks on a graph with 80 % intra-community edges,
        # profiles/r03_partition_locality.log).  The hop shards then belong to the nodes `self.node_ids` (original ids of rows
        # [lo, hi) of the relabelled problem); gather_full(..., original_order=True) undoes the relabelling.  P A P^T sorts a row's
        # terms by NEW column id, so hops agree with the unpartitioned run to rounding (1e-5), not bit for bit: refused with strict_order.
        self.partition = partition
        self.node_ids = None
        self.lo = self.hi = self.c0 = self.c1 = None
        self._cache = None
        self._props = {}

    def _ranks(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def _gloo(self):
        return dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "gloo"

    def _normalize_block(self, blk):
        """this rank's rows of A_hat.  The (r, alpha)-independent part (T + I in fp64, the global degree vector: the only
        communication) is kept ON the RowBlock (sgl_amd.config.cache_prepared), so the next operator over the same block -- the
        other graph operators of a PaSca sweep, the r values of the NAFS ensemble -- pays one pass over its non-zeros.  The
        degree powers come from the host's numpy (bit-identical to the reference's A_hat) under strict_order and whenever the
        degrees have few distinct values (every unit-weight graph: microseconds); real-valued degrees outside strict_order use the
        device's pow() (A_hat within 1 ulp(fp32); no round trip of an N-vector through the host)."""
        from .. import config
        from .. import device as dev
        from ..operators.base_op import AdjIdentity
        prep = None
        if config.cache_prepared:
            held = getattr(blk, "_prepared", None)
            world = self._ranks()[1]
            if held is not None and held[0] == (self.symmetric, world) and held[1].matches(blk):
                prep = held[2]
            else:
                prep = dev.PreparedBlock(blk.rowptr, blk.col, blk.val, blk.lo, blk.n, symmetric=self.symmetric, group=self.group)
                try:
                    blk._prepared = ((self.symmetric, world), AdjIdentity(blk), prep)
                except AttributeError:
                    pass
        return dev.normalize_block(blk.rowptr, blk.col, blk.val, blk.lo, blk.n, self.r, self.alpha, symmetric=self.symmetric,
                                   group=self.group, host_pow=True if self.strict_order else "auto", prepared=prep)

    def _propagate_block(self, blk, feature):
        """row-sharded storage: normalise and multiply this rank's rows only"""
        from .. import device as dev
        from ..operators.base_op import AdjIdentity
        rank, world = self._ranks()
        if self.row_groups not in (None, world):
            raise ValueError("a RowBlock adjacency implies the row-sharded layout")
        key = ("block", world, rank, blk.lo, blk.hi)
        if self._cache is None or self._cache[0] != key or not self._cache_ident.matches(blk):
            rowptr, col, val = self._normalize_block(blk)
            nblk = RowBlock(blk.lo, blk.hi, blk.n, rowptr, col, val)
            self._cache = [key, None, None, None]
            self._cache_ident = AdjIdentity(blk)
            self._props = {}
            self.a_hat_block = nblk
        n = blk.n
        x = feature if torch.is_tensor(feature) else torch.from_numpy(np.ascontiguousarray(feature, dtype=np.float32))
        x = x.to(device=blk.device, dtype=torch.float32)
        transport = self.transport or ("halo" if world > 1 else "p2p")
        if transport == "halo":
            if self._cache[2] is None:                       # only the ranks' block boundaries are needed
                self._cache[2] = gather_piece_bounds([blk.lo, blk.hi], self.group)
            return self._propagate_halo(x, self._cache[2], rank, world)
        if self._cache[1] is None:                           # full-replica transports: SpMM handles on global column ids
            from .sharded_adj import global_nnz
            fns, handles, mine = block_piece_spmms(self.a_hat_block, self.pieces, strict=self.strict_order, reorder=self.reorder,
                                                   total_nnz=global_nnz(self.a_hat_block, self.group))
            self._cache[1:] = [fns, gather_piece_bounds(mine, self.group), handles]
        _, fns, pb, handles = self._cache
        if x.shape[0] == blk.n_local and x.shape[0] != n:
            x = allgather_rows(x.contiguous(), pb[:, 0].tolist() + [int(pb[-1, -1])], n, group=self.group)
        if x.shape[0] != n:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        if transport == "p2p" and world > 1 and self._gloo() and x.is_cuda:
            transport = "staged"
        prop = self._props.get(("rows", transport))
        if prop is None:
            prop = self._props[("rows", transport)] = ShardedPropagator(fns, pb, rank, world, n, group=self.group,
                                                                        transport=transport)
        self._prop = prop
        self.lo, self.hi, self.c0, self.c1 = prop.lo, prop.hi, 0, x.shape[1]
        x = x.contiguous()
        chunks = column_chunks(x.shape[1], self.col_chunks if world > 1 else 1)
        if len(chunks) == 1:
            return prop.propagate(x, self.prop_steps)
        hops = prop.propagate_chunked([x[:, a:b].contiguous() for a, b in chunks], self.prop_steps)
        return [torch.cat(h, dim=1) for h in hops]

    def _propagate_halo(self, x, pb, rank, world):
        """need-aware exchange (the default of the row-sharded storage path with several ranks): this rank gathers from a
        compact table [own rows | the rows of each peer its block references] and receives only those rows between hops
        (sgl_amd/dist/halo.py).  `x` may be the full matrix or just this rank's rows -- a rank never needs more than its own."""
        from .halo import block_halo
        blk = self.a_hat_block
        halo = self._props.get("halo")
        if halo is None:
            bounds = pb[:, 0].tolist() + [int(pb[-1, -1])]
            halo = self._props["halo"] = block_halo(blk, bounds, group=self.group, strict=self.strict_order, reorder=self.reorder)
        plan, prop, _ = halo
        self._prop = prop
        self.halo_plan = plan
        self.lo, self.hi, self.c0, self.c1 = blk.lo, blk.hi, 0, x.shape[1]
        if x.shape[0] == blk.n and (blk.n_local != blk.n or world == 1):
            table = prop.table_from_full(x.contiguous())
        elif x.shape[0] == blk.n_local:
            table = prop.table_from_own(x.contiguous())
        else:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        chunks = column_chunks(x.shape[1], self.col_chunks if world > 1 else 1)
        if len(chunks) == 1:
            return prop.propagate(table, self.prop_steps)
        hops = prop.propagate_chunked([table[:, a:b].contiguous() for a, b in chunks], self.prop_steps)
        return [torch.cat(h, dim=1) for h in hops]

    def propagate(self, adj, feature):
        from .. import device as dev
        from ..io import DeviceAdjacency
        if isinstance(adj, RowBlock) and not self.partition:
            return self._propagate_block(adj, feature)
        rank, world = self._ranks()
        device = torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        if isinstance(adj, RowBlock):
            if self.strict_order:
                raise ValueError("partition= relabels the problem (a row's terms are added in another order): not with strict_order")
            return self._propagate_partitioned(adj, feature, rank, world, adj.device)
        from ..operators.base_op import AdjIdentity
        caller_adj = adj                     # the cache is keyed on the CALLER's object, never on a temporary of ours
        n = adj.shape[0]
        row_groups = world if self.row_groups is None else int(self.row_groups)
        if self.partition:
            if row_groups != world:
                raise ValueError("partition= applies to the row-sharded layout only")
            if self.strict_order:
                raise ValueError("partition= relabels the problem (a row's terms are added in another order): not with strict_order")
            return self._propagate_partitioned(adj, feature, rank, world, device)
        layout = GridLayout(world, row_groups)
        rg, cg = layout.coords(rank)
        key = (world, rank, row_groups)
        if self._cache is None or self._cache[0] != key or not self._cache_ident.matches(caller_adj):
            if not isinstance(adj, DeviceAdjacency):
                adj = DeviceAdjacency.from_scipy(adj, device=device)
            self._cache_ident = AdjIdentity(caller_adj)
            rowptr, col, val = dev.normalize_adj(adj.rowptr, adj.col, adj.val, n, self.r, self.alpha)
            rp_host = rowptr.cpu().numpy()
            pb = all_piece_bounds(rp_host, row_groups, self.pieces)
            fns, handles = device_piece_spmms(rowptr, col, val, n, pb[rg], rowptr_host=rp_host, strict=self.strict_order)
            self._cache = (key, fns, pb, handles)
            self._props = {}
        _, fns, pb, handles = self._cache
        x = feature if torch.is_tensor(feature) else torch.from_numpy(np.ascontiguousarray(feature, dtype=np.float32))
        x = x.to(device=device, dtype=torch.float32)
        if x.shape[0] != n:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        d = x.shape[1]
        slices = column_slices(d, layout.col_groups)
        self.c0, self.c1 = slices[cg]
        if layout.col_groups == 1:
            transport = self.transport or ("staged" if world > 1 and self._gloo() else "p2p")
            prop = self._props.get(("rows", transport))
            if prop is None:
                prop = self._props[("rows", transport)] = ShardedPropagator(fns, pb, rank, world, n, group=self.group,
                                                                            transport=transport)
            self._prop = prop
            self.lo, self.hi = prop.lo, prop.hi
            x = x.contiguous()
            chunks = column_chunks(d, self.col_chunks if world > 1 else 1)
            if len(chunks) == 1:
                return prop.propagate(x, self.prop_steps)
            hops = prop.propagate_chunked([x[:, a:b].contiguous() for a, b in chunks], self.prop_steps)
            return [torch.cat(h, dim=1) for h in hops]
        # grid / feature-sharded: the slice is stored zero-padded to a line-friendly pitch and multiplied at that width
        # (the pad columns stay zero and cost no extra cache lines)
        pitch = [dev.row_pitch(b - a, growth=2.0) if b > a else 0 for a, b in slices]
        transport = self.transport or (("relay_staged" if self._gloo() else "relay") if row_groups > 1 else "p2p")
        prop = self._props.get((d, transport))               # keeps its relay buffers / streams across calls
        if prop is None:
            prop = self._props[(d, transport)] = ShardedPropagator(fns, pb, rg, row_groups, n, group=self.group,
                                                                   transport=transport, layout=layout, me=rank, widths=pitch)
        self._prop = prop
        self.lo, self.hi = prop.lo, prop.hi
        w = self.c1 - self.c0
        xs = torch.zeros((n, pitch[cg]), dtype=torch.float32, device=device)
        xs[:, :w] = x[:, self.c0:self.c1]
        hops = prop.propagate(xs, self.prop_steps)
        return [h[:, :w] for h in hops]

    def _propagate_partitioned(self, adj, feature, rank, world, device):
        """full adjacency in, community-aware row blocks: normalise, relabel in the plan-time community order (when it helps),
        cut the RELABELLED matrix into nnz-balanced row blocks, keep this rank's block and run the need-aware exchange on it"""
        from .. import device as dev
        from ..io import DeviceAdjacency
        from ..operators.base_op import AdjIdentity
        from ..reorder import permute_csr, plan_order
        from .halo import block_halo
        from .layout import balanced_bounds
        n = adj.n if isinstance(adj, RowBlock) else adj.shape[0]
        key = ("partitioned", world, rank, self.partition)
        if self._cache is None or self._cache[0] != key or not self._cache_ident.matches(adj):
            ident = AdjIdentity(adj)
            if isinstance(adj, RowBlock):
                # storage already row-sharded: normalise the block where it lies; the relabelling is found and applied on the ranks' own
                # rows (sgl_amd/dist/redistribute.py): labels by distributed label propagation (what is replicated is one integer per
                # node), then every row moves to the rank that owns its new id.  No rank ever holds the whole matrix.
                from ..reorder import AUTO_MIN_GAIN, AUTO_MIN_LOCALITY
                from .redistribute import _gather_ints, redistribute_rows, sharded_community_order, sharded_edge_locality
                rp_b, c_b, v_b = self._normalize_block(adj)
                nblk = RowBlock(adj.lo, adj.hi, n, rp_b, c_b, v_b)
                old = _gather_ints([adj.lo, adj.hi], self.group)
                old_bounds = [int(v) for v in old[:, 0]] + [int(old[-1, 1])]
                order, text = sharded_community_order(nblk, old_bounds, self.group)
                info = {"partition": self.partition, "communities": text, "applied": True, "found_on": "row blocks (no rank holds the matrix)"}
                if self.partition == "auto":
                    before, after = sharded_edge_locality(nblk, None, self.group), sharded_edge_locality(nblk, order, self.group)
                    use = after >= AUTO_MIN_LOCALITY and after >= before + AUTO_MIN_GAIN
                    info.update({"edge_locality_before": round(before, 4), "edge_locality_after": round(after, 4), "applied": bool(use)})
                    if not use:
                        order = None
                if order is None:
                    order = torch.arange(n, dtype=torch.int64, device=nblk.device)
                blk, bounds = redistribute_rows(nblk, order, self.group)
                perm = torch.argsort(order)                                 # perm[k] = original id of the node relabelled k
                self._old_bounds = old_bounds
                del nblk
            else:
                dadj = adj if isinstance(adj, DeviceAdjacency) else DeviceAdjacency.from_scipy(adj, device=device)
                rowptr, col, val = dev.normalize_adj(dadj.rowptr, dadj.col, dadj.val, n, self.r, self.alpha)
                order, info = plan_order(rowptr, col, n, self.partition)   # identical on every rank: same kernels, same input
                if order is not None:
                    rowptr, col, val = permute_csr(rowptr, col, val, order)
                    perm = torch.argsort(order)                             # perm[k] = original id of the node relabelled k
                else:
                    perm = torch.arange(n, dtype=torch.int64, device=rowptr.device)
                bounds = balanced_bounds(rowptr.cpu().numpy(), world)
                lo, hi = int(bounds[rank]), int(bounds[rank + 1])
                a0, a1 = int(rowptr[lo]), int(rowptr[hi])
                blk = RowBlock(lo, hi, n, (rowptr[lo:hi + 1] - rowptr[lo]).contiguous(), col[a0:a1].contiguous(), val[a0:a1].contiguous())
                del rowptr, col, val
            plan, prop, handle = block_halo(blk, [int(b) for b in bounds], group=self.group, strict=False, reorder=self.reorder)
            self._cache = (key, plan, prop, handle, perm, info)
            self._cache_ident = ident
            self.a_hat_block = blk
        _, plan, prop, _, perm, info = self._cache
        self._prop, self.halo_plan, self.partition_info = prop, plan, info
        self.lo, self.hi = plan.lo, plan.hi
        self.node_ids = perm[plan.lo:plan.hi]
        self._perm = perm
        x = feature if torch.is_tensor(feature) else torch.from_numpy(np.ascontiguousarray(feature, dtype=np.float32))
        x = x.to(device=device, dtype=torch.float32)
        if isinstance(adj, RowBlock) and x.shape[0] == adj.n_local and x.shape[0] != n:
            # this rank's feature rows only (old ids): the rows the relabelled block's compact table holds are scattered over all
            # ranks -- each is asked of its owner, exactly those rows travel (never the 57 GB matrix of a papers100M-sized job)
            from .redistribute import fetch_rows
            self.c0, self.c1 = 0, x.shape[1]
            table = fetch_rows(x.contiguous(), self._old_bounds, perm[plan.global_ids], self.group)
            chunks = column_chunks(x.shape[1], self.col_chunks if world > 1 else 1)
            if len(chunks) == 1:
                return prop.propagate(table, self.prop_steps)
            hops = prop.propagate_chunked([table[:, a:b].contiguous() for a, b in chunks], self.prop_steps)
            return [torch.cat(h, dim=1) for h in hops]
        if x.shape[0] != n:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        self.c0, self.c1 = 0, x.shape[1]
        table = dev.gather_rows(x.contiguous(), perm[plan.global_ids])      # the compact table of the RELABELLED problem, cut out of X
        chunks = column_chunks(x.shape[1], self.col_chunks if world > 1 else 1)
        if len(chunks) == 1:
            return prop.propagate(table, self.prop_steps)
        hops = prop.propagate_chunked([table[:, a:b].contiguous() for a, b in chunks], self.prop_steps)
        return [torch.cat(h, dim=1) for h in hops]

    def gather_full(self, local, original_order=False):
        """assemble the full [N, d] matrix on every rank from the ranks' blocks (e.g. the final aggregated features);
        original_order=True undoes the relabelling of partition= (row i = node i again).
        `local` is this rank's block: rows [lo, hi), any width (the same inside a column group); the column groups'
        blocks are laid side by side in column-group order."""
        rank, world = self._ranks()
        if world == 1:
            if original_order and getattr(self, "_perm", None) is not None and self.partition:
                out = torch.empty_like(local)
                out[self._perm.to(local.device)] = local
                return out
            return local
        prop = self._prop
        layout = prop.layout or GridLayout(world, world)
        staged = local.is_cuda and self._gloo()
        send = (local.detach().cpu() if staged else local).contiguous()
        widths = [None] * world
        dist.all_gather_object(widths, int(local.shape[1]), group=self.group)
        col_w = [int(widths[layout.members(cg)[0]]) for cg in range(layout.col_groups)]
        col_off = np.concatenate([[0], np.cumsum(col_w)])
        total = int(col_off[-1])
        full = torch.zeros((prop.n, total), dtype=local.dtype, device=send.device)
        ops, landings = [], []
        for k in range(1, world):
            dst, src = (rank + k) % world, (rank - k) % world
            if send.numel():
                ops.append(dist.P2POp(dist.isend, send, dst, group=self.group))
            rg_s, cg_s = layout.coords(src)
            r0, r1 = int(prop.pb[rg_s, 0]), int(prop.pb[rg_s, -1])
            if (r1 - r0) * col_w[cg_s]:
                buf = torch.empty((r1 - r0, col_w[cg_s]), dtype=local.dtype, device=send.device)
                ops.append(dist.P2POp(dist.irecv, buf, src, group=self.group))
                landings.append((r0, r1, int(col_off[cg_s]), buf))
        works = dist.batch_isend_irecv(ops) if ops else []
        rg, cg = layout.coords(rank)
        full[prop.lo:prop.hi, int(col_off[cg]):int(col_off[cg]) + local.shape[1]].copy_(send)
        for w in works:
            w.wait()
        for r0, r1, c, buf in landings:
            full[r0:r1, c:c + buf.shape[1]].copy_(buf)
        full = full.to(local.device) if staged else full
        if original_order and getattr(self, "_perm", None) is not None and self.partition:
            out = torch.empty_like(full)
            out[self._perm.to(full.device)] = full          # row k of the relabelled problem is node perm[k]
            return out
        return full

    def gather_rows(self, local):
        """all-gather a local [hi-lo, d] shard into the full [N, d] matrix (row-sharded layout)"""
        return self.gather_full(local)

    def over_smooth_sweep(self, hops, hops_list):
        """The NAFS aggregate of EVERY requested prefix X_0..X_h of this rank's hop shards in one pass (the adaptive-k-hop sweep of the
        NAFS tasks, tasks/node_clustering.py:139,176-178, on row-sharded storage: BASELINE config 4): {h: [n_local, d]}.  The
        over-smoothing weights are per node and a row-sharded rank owns whole rows, so this is sgl_nafs_prefix_f32 on the local
        shards with no communication at all; column-sliced layouts fall back to one over_smooth_aggregate per prefix."""
        from .. import device as dev
        hops_list = sorted({int(h) for h in (range(hops_list) if isinstance(hops_list, int) else hops_list)})
        prop = self._prop
        feats = [h.contiguous() for h in hops]
        if (prop.layout is None or prop.layout.col_groups == 1) and feats[0].shape[1] <= 512:
            return dict(zip(hops_list, dev.nafs_prefix(feats, hops_list)))
        return {h: self.over_smooth_aggregate(hops[:h + 1]) for h in hops_list}

    def over_smooth_aggregate(self, hops):
        """OverSmoothDistanceWeightedOp (NAFS, message_op/over_smooth_distance_op.py:6-33) on this rank's blocks, for any
        layout.  The weights need whole rows (cosine of X_0[n] and X_h[n]); a rank that owns only a column slice
        contributes the partial sums of its columns -- X_0.X_h and |X_h|^2 per row and hop, one all-reduce of
        [N, 2H] floats for the whole job -- and then combines its own columns with the shared weights.  Row-sharded
        ranks own whole rows and use the fused single-pass kernel directly."""
        from .. import device as dev
        prop = self._prop
        feats = [h.contiguous() for h in hops]
        if prop.layout is None or prop.layout.col_groups == 1:
            return dev.nafs_aggregate(feats)
        H = len(feats)
        part = torch.zeros((prop.n, 2 * H), dtype=torch.float32, device=feats[0].device)
        blk = part[prop.lo:prop.hi]
        for h, xh in enumerate(feats):
            blk[:, h] = (feats[0] * xh).sum(dim=1)
            blk[:, H + h] = (xh * xh).sum(dim=1)
        if self._gloo() and part.is_cuda:
            host = part.cpu()
            dist.all_reduce(host, group=self.group)
            part.copy_(host)
        else:
            dist.all_reduce(part, group=self.group)
        norms = blk[:, H:].sqrt() + 1e-10                    # the reference adds 1e-10 to each norm (:14, :16)
        w = torch.softmax(blk[:, :H] / norms / norms[:, :1], dim=1).contiguous()
        return dev.hop_wsum2d(feats, w)
