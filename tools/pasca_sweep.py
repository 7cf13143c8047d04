#!/usr/bin/env python3
"""PaSca operator sweep (BASELINE config 5 shape, on the products-sized graph that fits one GPU):
graph ops {Laplacian r=0.5, PPR alpha in {0.1, 0.2, 0.3}} x k = 10 hops, then every MessageOp over the 11 hop
matrices.  Search space: sgl/search/search_config.py:14-15, search_models.py:19-46.  Prints wall time per stage
(HIP events) -- the quantity PaSca's second objective `time_preprocess` measures (auto_search.py:29,54)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import device as dev, synthetic  # noqa: E402
from sgl_amd.io import DeviceAdjacency  # noqa: E402
from sgl_amd.operators import message_op as M  # noqa: E402
from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), r


def main():
    wl = synthetic.WORKLOADS[os.environ.get("SWEEP_WORKLOAD", "S1_products")]
    n, d, K = wl["n"], wl["d"], 10
    device = torch.device("cuda", 0)
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    t_prep, prep = timed(lambda: dev.PreparedAdjacency(a_ptr, a_col, a_val, n), reps=1)
    print(f"SWEEP prepare (A + I, degrees, symmetry check; once per graph) ms={t_prep:8.2f} symmetric={prep.symmetric}", flush=True)
    for name, r, alpha in (("laplacian r=0.5", 0.5, None), ("ppr a=0.1", 0.5, 0.1), ("ppr a=0.2", 0.5, 0.2), ("ppr a=0.3", 0.5, 0.3)):
        t_norm, (rowptr, col, val) = timed(lambda: prep.normalize(r, alpha), reps=2)
        if name.startswith("lap"):
            t_full, _ = timed(lambda: dev.normalize_adj(a_ptr, a_col, a_val, n, r, alpha), reps=2)
            t_dev, _ = timed(lambda: dev.normalize_adj(a_ptr, a_col, a_val, n, r, alpha, host_pow=False), reps=2)
            print(f"SWEEP normalise from scratch ms={t_full:8.2f} (prepare + this r); general all-device pipeline with sort ms={t_dev:8.2f}", flush=True)
        csr = dev.DeviceCSR(rowptr, col, val, (n, n))

        def prop():
            feats = [x0]
            for _ in range(K):
                feats.append(csr.spmm(feats[-1]))
            return feats
        t_prop, feats = timed(prop)
        nnz = col.numel()
        print(f"SWEEP graph_op={name:16s} normalise_ms={t_norm:8.2f} propagate_k10_ms={t_prop:8.2f} "
              f"({nnz * d * K / (t_prop * 1e-3) / 1e12:.3f}e12 edge*feat/s)", flush=True)
        if alpha is None:
            lap_feats = feats
        else:
            # the same K + 1 hop matrices WITHOUT propagating: ((1 - a) A_hat + a I)^k X is a polynomial in A_hat, i.e. a triangular
            # mix of the Laplacian chain's hop matrices (ppr_hops_from_laplacian -> sgl_hop_lincomb_f32: 11 streams read, 10 written)
            from sgl_amd.operators.graph_op import ppr_hops_from_laplacian
            t_mix, mixed = timed(lambda: ppr_hops_from_laplacian(lap_feats, alpha))
            err = max(float((m_ - f_).abs().max() / f_.abs().max()) for m_, f_ in zip(mixed, feats))
            by = (2 * K + 1) * n * d * 4
            print(f"SWEEP   the same hop matrices mixed from the Laplacian chain: ms={t_mix:8.2f} ({by / (t_mix * 1e-3) / 1e12:.2f} TB/s; "
                  f"max |mixed - propagated| / max |propagated| = {err:.1e}) instead of {t_prop:8.2f}", flush=True)
            del mixed
        if alpha not in (None, 0.1):
            continue
        ops = [("last", M.LastMessageOp()), ("concat", M.ConcatMessageOp(0, K + 1)), ("mean", M.MeanMessageOp(0, K + 1)),
               ("sum", M.SumMessageOp(0, K + 1)), ("max", M.MaxMessageOp(0, K + 1)), ("min", M.MinMessageOp(0, K + 1)),
               ("simple_weighted a=.85", M.SimpleWeightedMessageOp(0, K + 1, "alpha", 0.85)),
               ("learnable simple", M.LearnableWeightedMessageOp(0, K + 1, "simple", K).to(device)),
               ("learnable gate", M.LearnableWeightedMessageOp(0, K + 1, "gate", d).to(device)),
               ("learnable ori_ref", M.LearnableWeightedMessageOp(0, K + 1, "ori_ref", d).to(device)),
               ("learnable jk", M.LearnableWeightedMessageOp(0, K + 1, "jk", K, d).to(device)),
               ("iterate recursive", M.IterateLearnableWeightedMessageOp(0, K + 1, "recursive", d).to(device)),
               ("nafs over_smooth", M.OverSmoothDistanceWeightedOp())]
        for oname, op in ops:
            with torch.no_grad():
                t, _ = timed(lambda: op.aggregate(feats))
            print(f"SWEEP   msg_op={oname:24s} aggregate_ms={t:8.3f}", flush=True)
        del csr
        if alpha is not None:
            del feats
        # the same aggregates with the aggregation folded into the propagation (round 2): sum / mean / simple_weighted ride
        # on the SpMM epilogue (GraphOp.propagate_reduce), concat is the layout the hops are produced in (slab_hops)
        adj = DeviceAdjacency(a_ptr, a_col, a_val, (n, n))
        gop = LaplacianGraphOp(K, r=r) if alpha is None else PprGraphOp(K, r=r, alpha=alpha)
        gop.propagate(adj, x0)                      # normalised adjacency cached on the operator from here on
        t_plain, _ = timed(lambda: gop.propagate(adj, x0))
        for oname, op in ops[:1] + ops[2:4] + ops[6:7]:
            t, _ = timed(lambda: gop.propagate_reduce(adj, x0, **op.fused_spec(K + 1)))
            print(f"SWEEP   fused  {oname:24s} propagate+aggregate_ms={t:8.2f} (propagate alone {t_plain:8.2f}: aggregate adds "
                  f"{t - t_plain:+.2f} ms)", flush=True)
        gslab = LaplacianGraphOp(K, r=r, slab_hops=True) if alpha is None else PprGraphOp(K, r=r, alpha=alpha, slab_hops=True)
        gslab.propagate(adj, x0)
        t_slab, hops = timed(lambda: gslab.propagate(adj, x0))
        with torch.no_grad():
            t_cat, cat = timed(lambda: ops[1][1].aggregate(hops))
        print(f"SWEEP   slab   concat: propagate_k10_ms={t_slab:8.2f} (separate hop buffers {t_plain:8.2f}) concat_ms={t_cat:8.3f} "
              f"view={cat.untyped_storage().data_ptr() == hops[0].untyped_storage().data_ptr()}", flush=True)
        del hops, cat, gop, gslab


if __name__ == "__main__":
    main()
