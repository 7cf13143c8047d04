#!/usr/bin/env python3
"""Throughput of the MessageOp kernels at the products shape (N = 2 449 029, d = 100, H = 4 hops) on the GPU box.
Prints achieved GB/s against the algorithmic bytes of each op.
Environment: AGG_N = rows (default 2 449 029), AGG_SHAPES = "dxH,dxH,..." replaces the default shapes (e.g. 250x11,200x6: the
two-chunks-per-lane instantiations of the row kernels)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import _lib  # noqa: E402
from sgl_amd import device as dev  # noqa: E402


def timeit(fn, reps=7, warm=2):
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


def main():
    n = int(os.environ.get("AGG_N", 2_449_029))
    if os.environ.get("AGG_LPR32X2"):
        _lib.set_tuning("row_lpr32x2", int(os.environ["AGG_LPR32X2"]))
    if os.environ.get("AGG_NARROW"):      # row layouts of the register-resident kernels: 0 = power-of-two groups only, 1 = auto, 2 / 3 = one candidate
        _lib.set_tuning("row_narrow_groups", int(os.environ["AGG_NARROW"]))
    if os.environ.get("AGG_WHOLE"):       # 0: row outputs end at column d (a partly written last line), 1: pad columns written too
        _lib.set_tuning("row_whole_lines", int(os.environ["AGG_WHOLE"]))
    if os.environ.get("AGG_CONCAT"):      # any-width concat: 1 = auto, 2 = 1024-float tiles, 3 = whole rows per block in LDS, 0 = funnel-select kernel
        _lib.set_tuning("concat_lds", int(os.environ["AGG_CONCAT"]))
    if os.environ.get("AGG_BLOCKS"):
        _lib.set_tuning("agg_blocks", int(os.environ["AGG_BLOCKS"]))
    device = torch.device("cuda", 0)
    shapes = ((100, 4), (128, 11), (147, 6), (128, 6), (16, 4))
    if os.environ.get("AGG_SHAPES"):                      # e.g. AGG_SHAPES=250x11,200x6
        shapes = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["AGG_SHAPES"].split(","))
    for d, H in shapes:
        feats = [dev.alloc_rows(n, d, device) for _ in range(H)]
        for f in feats:
            f.normal_()
        nb = n * d * 4

        def rep(name, ms, bytes_):
            print(f"AGG d={d} H={H} {name:44s} ms={ms:8.3f} GB/s={bytes_ / (ms * 1e-3) / 1e9:8.1f} frac_of_8TB/s={bytes_ / (ms * 1e-3) / 8e12:.3f}", flush=True)

        for name, op in (("sum", _lib.SGL_REDUCE_SUM), ("mean", _lib.SGL_REDUCE_MEAN), ("max", _lib.SGL_REDUCE_MAX)):
            rep(name, timeit(lambda: dev.hop_reduce(op, feats)), (H + 1) * nb)
        w1 = torch.rand(H, device=device)
        rep("wsum1d", timeit(lambda: dev.hop_reduce(_lib.SGL_REDUCE_WSUM, feats, w1)), (H + 1) * nb)
        rep("concat", timeit(lambda: dev.hop_concat(feats)), 2 * H * nb)
        if d % 4:
            _lib.set_tuning("concat_flat_read", 0)
            rep("concat (row-by-row reads)", timeit(lambda: dev.hop_concat(feats)), 2 * H * nb)
            _lib.set_tuning("concat_flat_read", 1)
        w2 = torch.softmax(torch.randn(n, H, device=device), 1)
        rep("wsum2d fwd", timeit(lambda: dev.hop_wsum2d(feats, w2)), (H + 1) * nb + n * H * 4)
        w2g = w2.clone().requires_grad_(True)
        out = dev.hop_wsum2d(feats, w2g)
        g = torch.randn_like(out)
        rep("wsum2d bwd (dW)", timeit(lambda: torch.autograd.grad(out, w2g, g, retain_graph=True)), (H + 1) * nb + n * H * 4)
        w1g = w1.clone().requires_grad_(True)
        out1 = dev.hop_wsum1d(feats, w1g)
        rep("wsum1d bwd (dw)", timeit(lambda: torch.autograd.grad(out1, w1g, g, retain_graph=True)), (H + 1) * nb)
        del out1
        v = torch.randn(d, device=device)
        rep("gate scores (rowdot)", timeit(lambda: dev.hop_scores(feats, v)), H * nb + n * H * 4)
        bb = torch.zeros(1, device=device)
        rep("gate (single pass)", timeit(lambda: dev.hop_gate(feats, v, bb)), (H + 1) * nb + 2 * n * H * 4)
        rep("gate (two-pass route)", timeit(lambda: dev.hop_wsum2d(feats, torch.softmax(torch.sigmoid(dev.hop_scores(feats, v) + bb), 1))),
            (H + 1) * nb + 2 * n * H * 4)
        uu = torch.randn(H, d, device=device)
        rep("jk scores (one pass)", timeit(lambda: dev.hop_scores2(feats, v, uu, (1 << H) - 1, 0, H)), H * nb + n * (H + 1) * 4)
        rep("jk scores (hstack+GEMV)", timeit(lambda: (torch.hstack(feats) @ uu.view(-1), dev.hop_scores(feats, v))), H * nb + n * (H + 1) * 4)
        from sgl_amd.operators.message_op import IterateLearnableWeightedMessageOp
        it = IterateLearnableWeightedMessageOp(0, H, "recursive", d).to(device)
        with torch.no_grad():                            # GAMLP-R's recursive gate; algorithmic bytes: every hop once, one output
            rep("recursive gate (iterate op)", timeit(lambda: it.aggregate(feats)), (H + 1) * nb)
        rep("nafs (weights + sum)", timeit(lambda: dev.nafs_aggregate(feats)), (H + 1) * nb)
        idx = torch.randint(0, n, (200_000,), device=device)
        rep("gather_rows 200k", timeit(lambda: dev.gather_rows(feats[0], idx)), 2 * 200_000 * d * 4)
        # the training feed of the learnable aggregators: the same 200k rows of EVERY hop matrix (models/base_model.py:58-60) -- hop by hop
        # against ONE index upload for all of them (device.gather_hops), device and host (numpy int64) indices
        idx_host = idx.cpu().numpy()
        rep(f"gather 200k rows of {H} hops, hop by hop", timeit(lambda: [dev.gather_rows(f_, idx) for f_ in feats]), 2 * H * 200_000 * d * 4)
        rep(f"gather 200k rows of {H} hops, gather_hops", timeit(lambda: dev.gather_hops(feats, idx)), 2 * H * 200_000 * d * 4)
        rep(f"gather 200k rows of {H} hops, gather_hops per hop", timeit(lambda: dev.gather_hops(feats, idx, one_launch=False)), 2 * H * 200_000 * d * 4)
        for u_ in (1, 4):
            _lib.set_tuning("gather_rows_per_thread", u_)
            rep(f"gather 200k rows of {H} hops, gather_hops rows/thread={u_}", timeit(lambda: dev.gather_hops(feats, idx)), 2 * H * 200_000 * d * 4)
        _lib.set_tuning("gather_rows_per_thread", 0)
        g_one = dev.gather_hops(feats, idx)
        g_ref = [dev.gather_rows(f_, idx) for f_ in feats]
        print(f"AGG   one-launch gather bit-identical to hop by hop: {all(torch.equal(a_, b_) for a_, b_ in zip(g_one, g_ref))}", flush=True)
        rep(f"  ... host indices, hop by hop", timeit(lambda: [dev.gather_rows(f_, idx_host) for f_ in feats]), 2 * H * 200_000 * d * 4)
        rep(f"  ... host indices, gather_hops", timeit(lambda: dev.gather_hops(feats, idx_host)), 2 * H * 200_000 * d * 4)
        # the kernel alone: ten launches queued back to back into a preallocated output (the host's ~20 us per call -- allocation,
        # index checks, ctypes -- hide behind the GPU as they do in a training loop; a single timed call includes them)
        gout_ = dev.alloc_rows(200_000, d, device)
        rep("  gather_rows 200k (10 queued, per launch)", timeit(lambda: [dev.gather_rows(feats[0], idx, out=gout_) for _ in range(10)]) / 10,
            2 * 200_000 * d * 4)
        rep(f"  gather_hops 200k x {H} hops (10 queued, per launch)", timeit(lambda: [dev.gather_hops(feats, idx) for _ in range(10)]) / 10,
            2 * H * 200_000 * d * 4)
        for g_, u_ in ((0, 0), (1, 1), (1, 2), (1, 4)):                  # 0: hop loop inside the thread (round 6, first form); 1: hop in blockIdx.y
            _lib.set_tuning("gather_hops_grid", g_)
            _lib.set_tuning("gather_rows_per_thread", u_)
            ok_ = all(torch.equal(a_, b_) for a_, b_ in zip(dev.gather_hops(feats, idx), g_ref))
            what_ = "hop loop in the thread" if g_ == 0 else f"hop in the grid, rows/thread={u_}"
            rep(f"  gather_hops {what_}, single call (bit-identical {ok_})", timeit(lambda: dev.gather_hops(feats, idx)), 2 * H * 200_000 * d * 4)
            rep(f"  gather_hops {what_} (10 queued, per launch)",
                timeit(lambda: [dev.gather_hops(feats, idx) for _ in range(10)]) / 10, 2 * H * 200_000 * d * 4)
        _lib.set_tuning("gather_hops_grid", 1)
        _lib.set_tuning("gather_rows_per_thread", 0)
        rep("  contiguous copy 200k (10 queued, per launch)",
            timeit(lambda: [dev.padded_parent(gout_).copy_(dev.padded_parent(feats[0][:200_000])) for _ in range(10)]) / 10, 2 * 200_000 * d * 4)
        # what a launch of THIS size can reach at all: the same bytes as one contiguous copy, and a sorted (nearly sequential) gather
        cont = feats[0][:200_000]
        dst = dev.alloc_rows(200_000, d, device)
        rep("  ceiling: contiguous copy 200k", timeit(lambda: dev.padded_parent(dst).copy_(dev.padded_parent(cont))), 2 * 200_000 * d * 4)
        for lpr_, u_ in ((0, 1), (0, 4), (0, 16), (64, 0), (32, 0), (16, 0), (8, 0)):      # the knobs behind the default choice
            _lib.set_tuning("gather_lpr", lpr_)
            _lib.set_tuning("gather_rows_per_thread", u_)
            rep(f"  gather_rows 200k lpr={lpr_ or 'auto'} rows/thread={u_ or 'auto'}", timeit(lambda: dev.gather_rows(feats[0], idx)), 2 * 200_000 * d * 4)
        _lib.set_tuning("gather_lpr", 0)
        _lib.set_tuning("gather_rows_per_thread", 0)
        idx_sorted = torch.sort(idx).values
        rep("  gather_rows 200k (sorted ids)", timeit(lambda: dev.gather_rows(feats[0], idx_sorted)), 2 * 200_000 * d * 4)
        idx2m = torch.randint(0, n, (2_000_000,), device=device)
        rep("gather_rows 2M", timeit(lambda: dev.gather_rows(feats[0], idx2m)), 2 * 2_000_000 * d * 4)
        # torch reference points for the same math (not part of the product): stack+sum, index_select
        rep("[torch] sum of hops", timeit(lambda: sum(feats)), (H + 1) * nb)
        rep("[torch] x[idx]", timeit(lambda: feats[0][idx]), 2 * 200_000 * d * 4)
        del feats


if __name__ == "__main__":
    main()
