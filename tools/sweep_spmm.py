#!/usr/bin/env python3
"""Experiment harness for the SpMM kernel on the GPU box (not part of the product or of bench.py).

    python tools/sweep_spmm.py --workload S1_products --exp knobs,plan,colblock,relabel,pad

Prints one line per configuration: `EXP <name> <key=value ...> ms_per_hop=<t> frac=<algorithmic roofline fraction>`.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sgl_amd import _lib, synthetic  # noqa: E402
from sgl_amd import device as dev  # noqa: E402


def time_hops(fn, reps=5, warm=2):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S1_products")
    ap.add_argument("--exp", default="knobs,plan,colblock,relabel,pad")
    ap.add_argument("--k", type=int, default=3)
    a = ap.parse_args()
    exps = set(a.exp.split(","))
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = synthetic.WORKLOADS[a.workload]
    n, d, K = wl["n"], wl["d"], a.k
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    rowptr, col, val = dev.normalize_adj(a_ptr, a_col, a_val, n, 0.5, None)
    del a_ptr, a_col, a_val
    nnz = col.numel()
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    alg = nnz * d * 4 + nnz * 8 + (n + 1) * 4 + n * d * 4
    bufs = [torch.empty_like(x0) for _ in range(2)]

    def report(name, ms_total, hops, **kv):
        ms = ms_total / hops
        frac = alg / (ms * 1e-3) / 8e12
        print(f"EXP {name} " + " ".join(f"{k}={v}" for k, v in kv.items()) + f" ms_per_hop={ms:.3f} frac={frac:.3f}", flush=True)

    def chain(csr, xin=x0, outs=bufs):
        def f():
            cur = xin
            for h in range(K):
                out = outs[h % 2]
                csr.spmm(cur, out=out)
                cur = out
        return f

    knobs = ("spmm_unroll", "spmm_nt", "spmm_group", "spmm_waves", "spmm_xcd_remap", "spmm_vec")
    saved = {k: _lib.get_tuning(k) for k in knobs}

    def set_knobs(**kw):
        for k in knobs:
            _lib.set_tuning(k, kw.get(k, saved[k]))

    base = dev.DeviceCSR(rowptr, col, val, (n, n))
    report("base", time_hops(chain(base)), K, **base.info())
    strict = dev.DeviceCSR(rowptr, col, val, (n, n), strict=True)
    report("strict", time_hops(chain(strict)), K)
    del strict

    if "knobs" in exps:
        for unroll in (0, 1, 2):
            for nt in (0, 1):
                set_knobs(spmm_unroll=unroll, spmm_nt=nt)
                report("knobs", time_hops(chain(base)), K, unroll=unroll, nt=nt)
        for group in (32, 64):
            set_knobs(spmm_group=group)
            report("knobs", time_hops(chain(base)), K, group=group)
        for waves in (1, 2, 4):
            set_knobs(spmm_waves=waves)
            report("knobs", time_hops(chain(base)), K, waves=waves)
        for remap in (0, 1):
            set_knobs(spmm_xcd_remap=remap)
            report("knobs", time_hops(chain(base)), K, xcd_remap=remap)
        set_knobs()

    if "plan" in exps:
        for item_nnz in (128, 256, 512, 1024, 2048, 4096):
            for long_nnz in (512, 2048, 8192):
                c = dev.DeviceCSR(rowptr, col, val, (n, n), item_nnz=item_nnz, long_row_nnz=long_nnz)
                report("plan", time_hops(chain(c)), K, item_nnz=item_nnz, long_nnz=long_nnz, items=c.info()["n_items"],
                       pieces=c.info()["n_pieces"])
                del c

    if "small" in exps:
        # small graphs (Pubmed-sized: launch- and latency-bound): item size x gathers in flight, eager and as a replayed hipGraph
        xin = dev.padded_parent(dev.upload_rows(x0, device)) if dev.row_pitch(d) != d else x0
        for item_nnz, long_nnz in ((8, 0), (16, 0), (32, 0), (64, 0), (128, 0), (32, 128), (32, 64), (32, 32), (16, 32), (16, 16), (8, 16)):
            for unroll in (0, 2):
                set_knobs(spmm_unroll=unroll)
                c = dev.DeviceCSR(rowptr, col, val, (n, n), item_nnz=item_nnz, long_row_nnz=long_nnz)
                outs = [dev.padded_parent(dev.alloc_rows(n, d, device)) for _ in range(K)]
                g = c.capture_chain(xin, outs)
                report("small", time_hops(g.replay), K, item_nnz=item_nnz, long_nnz=long_nnz, unroll=unroll, items=c.info()["n_items"],
                       pieces=c.info()["n_pieces"], graph=1)
                del g, c, outs
        set_knobs()

    deg = rowptr[1:] - rowptr[:-1]
    if "colblock" in exps or "relabel" in exps:
        rows = torch.repeat_interleave(torch.arange(n, device=device), deg)

    def col_blocks(rp, cc, vv, rws, edges):
        out = []
        for c0, c1 in zip(edges[:-1], edges[1:]):
            m = (cc >= c0) & (cc < c1)
            cnt = torch.bincount(rws[m], minlength=n)
            p = torch.zeros(n + 1, dtype=torch.int64, device=device)
            p[1:] = torch.cumsum(cnt, 0)
            out.append(dev.DeviceCSR(p, cc[m].contiguous(), vv[m].contiguous(), (n, n)))
        return out

    def chain_blocks(blocks, xin):
        def f():
            cur = xin
            for h in range(K):
                out = bufs[h % 2]
                for b, c in enumerate(blocks):
                    c.spmm(cur, out=out, accumulate=(b > 0))
                cur = out
        return f

    if "colblock" in exps:
        for B in (2, 3, 4, 6, 8):
            edges = [int(round(i * n / B)) for i in range(B + 1)]
            blocks = col_blocks(rowptr, col, val, rows, edges)
            report("colblock", time_hops(chain_blocks(blocks, x0)), K, B=B)
            del blocks

    if "relabel" in exps:
        # relabel nodes by descending degree: hot rows of X become one contiguous, cache-resident slab
        order = torch.argsort(deg, descending=True, stable=True)       # order[new] = old
        new_id = torch.empty_like(order)
        new_id[order] = torch.arange(n, device=device)
        key = new_id[rows] * n + new_id[col.long()]
        key, perm = torch.sort(key)
        r2 = key // n
        c2 = (key % n).to(torch.int32)
        v2 = val[perm]
        del key, perm
        p2 = torch.zeros(n + 1, dtype=torch.int64, device=device)
        p2[1:] = torch.cumsum(torch.bincount(r2, minlength=n), 0)
        x2 = x0[order].contiguous()
        c = dev.DeviceCSR(p2, c2, v2, (n, n))
        report("relabel", time_hops(chain(c, xin=x2)), K, order="degree_desc")
        if "narrow" in exps:
            # column slices of a feature-sharded / grid rank: do hub rows stay L2-resident when rows are 64-128 bytes?
            for dd in (16, 32):
                xs = torch.randn((n, dd), device=device)
                outs = [torch.empty_like(xs) for _ in range(2)]
                t_base = time_hops(chain(base, xin=xs, outs=outs)) / K
                t_rel = time_hops(chain(c, xin=xs, outs=outs)) / K
                print(f"EXP relabel_narrow d={dd} original_order_ms={t_base:.3f} degree_order_ms={t_rel:.3f}", flush=True)
                del xs, outs
        for B, hot_frac in ((2, 0.1), (2, 0.25), (3, 0.2), (4, 0.25)):
            # first block = the hottest rows (what fits the 256 MiB Infinity Cache), rest equal width
            hot = int(n * hot_frac)
            edges = [0, hot] + [hot + int(round(i * (n - hot) / (B - 1))) for i in range(1, B)]
            blocks = col_blocks(p2, c2, v2, r2, edges)
            report("relabel+colblock", time_hops(chain_blocks(blocks, x2)), K, B=B, hot_frac=hot_frac)
            del blocks
        del c, p2, c2, v2, r2, x2

    if "dsweep" in exps:
        # time per hop vs feature width on the same graph: separates per-line from per-gather cost
        for dd in (16, 32, 64, 96, 100, 128, 192, 256):
            xs = torch.randn((n, dd), device=device)
            outs = [torch.empty_like(xs) for _ in range(2)]
            for unroll in (0, 2):
                set_knobs(spmm_unroll=unroll)
                ms = time_hops(chain(base, xin=xs, outs=outs))
                algd = nnz * dd * 4 + nnz * 8 + (n + 1) * 4 + n * dd * 4
                print(f"EXP dsweep d={dd} unroll={unroll} ms_per_hop={ms / K:.3f} frac={algd / (ms / K * 1e-3) / 8e12:.3f} "
                      f"Ggather_per_s={nnz / (ms / K * 1e-3) / 1e9:.2f}", flush=True)
            del xs, outs
        set_knobs()

    if "wsweep" in exps:
        # uniform-random gathers from a table of T rows (d floats each): where does the memory system saturate?
        deg_u, rows_u = 48, 1 << 21
        for dd in (100, 128, 32):
            for T in (1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
                g = torch.Generator(device=device)
                g.manual_seed(1)
                cc = torch.randint(0, T, (rows_u * deg_u,), generator=g, device=device, dtype=torch.int32)
                cc = cc.view(rows_u, deg_u).sort(dim=1).values.reshape(-1).contiguous()
                pp = torch.arange(0, rows_u + 1, device=device, dtype=torch.int64) * deg_u
                vv = torch.ones(rows_u * deg_u, device=device)
                cu = dev.DeviceCSR(pp, cc, vv, (rows_u, T))
                xt = torch.randn((T, dd), device=device)
                yo = torch.empty((rows_u, dd), device=device)
                for unroll in (0, 2):
                    set_knobs(spmm_unroll=unroll)
                    ms = time_hops(lambda: cu.spmm(xt, out=yo), reps=5, warm=2)
                    print(f"EXP wsweep d={dd} table_MB={T * dd * 4 / 2**20:.1f} unroll={unroll} ms={ms:.3f} "
                          f"Ggather_per_s={rows_u * deg_u / (ms * 1e-3) / 1e9:.2f} "
                          f"gathered_TBps={rows_u * deg_u * dd * 4 / (ms * 1e-3) / 1e12:.2f}", flush=True)
                del cu, cc, pp, vv, xt, yo
        set_knobs()

    if "relabel2" in exps:
        rows2 = torch.repeat_interleave(torch.arange(n, device=device), deg)
        order = torch.argsort(deg, descending=True, stable=True)
        new_id = torch.empty_like(order)
        new_id[order] = torch.arange(n, device=device)
        # relabel COLUMNS only (X rows sorted by hotness) but keep the original row order of the matrix, so the
        # work distribution is unchanged and only the locality of the gathered table differs
        c2 = new_id[col.long()].to(torch.int32)
        key = rows2 * n + c2.long()
        key, perm = torch.sort(key)
        c2 = (key % n).to(torch.int32)
        v2 = val[perm]
        del key, perm, rows2
        x2 = x0[order].contiguous()
        c = dev.DeviceCSR(rowptr, c2, v2, (n, n))
        for remap in (0, 1):
            set_knobs(spmm_xcd_remap=remap)
            ms = time_hops(lambda: c.spmm(x2, out=bufs[0]))
            print(f"EXP relabel2 hot_columns_first xcd_remap={remap} ms_per_hop={ms:.3f}", flush=True)
        set_knobs()
        del c, c2, v2, x2

    if "community" in exps:
        # same size / degree law, but 80 % of the edges stay inside contiguous blocks of `bs` nodes (community-ordered ids,
        # as real co-purchase / citation graphs largely are): what the XCD-aware block->row mapping buys when there IS locality
        for bs in (2048, 16384):
            g = torch.Generator(device=device).manual_seed(7)
            m = wl["m"]
            w = torch.exp(torch.randn(n, generator=g, device=device, dtype=torch.float64) * 1.2)
            cdf = torch.cumsum(w, 0); cdf /= cdf[-1].clone()
            a_ = torch.searchsorted(cdf, torch.rand(m, generator=g, device=device, dtype=torch.float64)).clamp_(0, n - 1)
            local = torch.rand(m, generator=g, device=device) < 0.8
            b_far = torch.searchsorted(cdf, torch.rand(m, generator=g, device=device, dtype=torch.float64)).clamp_(0, n - 1)
            b_near = ((a_ // bs) * bs + torch.randint(0, bs, (m,), generator=g, device=device)).clamp_(0, n - 1)
            b_ = torch.where(local, b_near, b_far)
            keep = a_ != b_
            lo_, hi_ = torch.minimum(a_, b_)[keep], torch.maximum(a_, b_)[keep]
            keys = torch.unique(lo_ * n + hi_)
            full = torch.sort(torch.cat([keys, (keys % n) * n + keys // n])).values
            rp = torch.zeros(n + 1, dtype=torch.int64, device=device)
            rp[1:] = torch.cumsum(torch.bincount(full // n, minlength=n), 0)
            cc = (full % n).to(torch.int32)
            vv = torch.ones(cc.numel(), device=device)
            del a_, b_, b_far, b_near, keys, full, lo_, hi_, keep, local, cdf, w
            rpn, ccn, vvn = dev.normalize_adj(rp, cc, vv, n, 0.5, None)
            cm = dev.DeviceCSR(rpn, ccn, vvn, (n, n))
            nz = ccn.numel()
            for remap in (1, 0):
                set_knobs(spmm_xcd_remap=remap)
                ms = time_hops(lambda: cm.spmm(x0, out=bufs[0]), reps=7, warm=2)
                algc = nz * d * 4 + nz * 8 + (n + 1) * 4 + n * d * 4
                print(f"EXP community block={bs} nnz={nz} xcd_remap={remap} ms_per_hop={ms:.3f} frac={algc / (ms * 1e-3) / 8e12:.3f} "
                      f"Ggather_per_s={nz / (ms * 1e-3) / 1e9:.2f}", flush=True)
            set_knobs()
            del cm, rpn, ccn, vvn, rp, cc, vv

    if "vec" in exps:
        for vec in (0, 2):
            for unroll in (3, 2, 4):
                for nt in (0, 1):
                    set_knobs(spmm_vec=vec, spmm_unroll=unroll, spmm_nt=nt)
                    ms = time_hops(lambda: base.spmm(x0, out=bufs[0]), reps=7, warm=2)
                    print(f"EXP vec vec={vec} unroll={unroll} nt={nt} ms_per_hop={ms:.3f}", flush=True)
        set_knobs()

    if "uns" in exps:
        # unroll x group per width (default-selection table)
        for dd in (8, 16, 32, 48, 64, 100, 128, 256, 500):
            xs = torch.randn((n, dd), device=device)
            outs = [torch.empty_like(xs) for _ in range(2)]
            for group in (0, 64):
                for unroll in (1, 0, 2):
                    set_knobs(spmm_unroll=unroll, spmm_group=group)
                    ms = time_hops(lambda: base.spmm(xs, out=outs[0]), reps=5, warm=1)
                    print(f"EXP uns d={dd} group={group} unroll={unroll} ms_per_hop={ms:.3f}", flush=True)
            del xs, outs
        set_knobs()

    if "pad" in exps:
        # rows padded to 128 floats (512 B, line aligned) instead of the native 400 B
        xp = torch.zeros((n, 128), device=device)
        xp[:, :d] = x0
        outs = [torch.zeros((n, 128), device=device) for _ in range(2)]

        def f():
            cur = xp[:, :d]
            for h in range(K):
                out = outs[h % 2][:, :d]
                base.spmm(cur, out=out)
                cur = out
        report("pad", time_hops(f), K, ld=128, d=d)

        def g():
            cur = xp
            for h in range(K):
                out = outs[h % 2]
                base.spmm(cur, out=out)
                cur = out
        ms = time_hops(g)
        alg128 = nnz * 128 * 4 + nnz * 8 + (n + 1) * 4 + n * 128 * 4
        print(f"EXP d128 ms_per_hop={ms / K:.3f} frac={alg128 / (ms / K * 1e-3) / 8e12:.3f}", flush=True)


if __name__ == "__main__":
    main()
