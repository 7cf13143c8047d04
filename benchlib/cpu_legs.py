"""The CPU legs of the secondary bench sections: the reference's CPU path timed on this node's host cores next to the HIP path
(reported baselines, never part of a measured GPU figure).  Together with benchlib/engine.py:cpu_baseline these are the only
places outside tests/ and __graft_entry__.smoke() that touch oracle/ (tests/test_host_cpu.py guards that)."""
import os
import time

import numpy as np
import torch


def _host_cores():
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    return len(cpus)


def config1_reference_path(indptr, indices, data, x, n, K, y_fast, hops_strict, tol):
    """BASELINE config 1 IS the reference CPU path ("SGC prop_steps=3 on Pubmed via reference scipy.sparse CPU path"): the fp64 scipy
    normalisation (operators/utils.py:76-88, restated in oracle/ref_ops.py) + K hops through the reference's own matmul.c compiled in
    place (oracle/_ref; the C restatement where it is absent), timed here -- and its result is the checker of the GPU section:
    hop K of the fast-order run within the SURVEY 8(c) tolerance, every hop of the strict-order run bit for bit.
    Returns (cpu_baseline dict, validation dict)."""
    import oracle  # test infrastructure: the reported baseline and the checker
    t0 = time.perf_counter()
    norm = oracle.laplacian_adj(indptr, indices, data, n, 0.5)
    t_norm = time.perf_counter() - t0
    ref = oracle.propagate(norm, x, K)                                    # the checker's result (C restatement; also the warm-up)
    kind = "reference" if oracle.load_reference_lib() is not None else "port"
    spmm = oracle.reference_spmm if kind == "reference" else oracle.oracle_spmm
    val32 = np.asarray(norm[2]).astype(np.float32)                        # operators/utils.py:32
    cur = np.ascontiguousarray(x, dtype=np.float32)
    spmm(norm[0], norm[1], val32, cur)                                    # warm-up of the timed kernel
    t0 = time.perf_counter()
    for _ in range(K):                                                     # base_op.py:29-35
        cur = spmm(norm[0], norm[1], val32, cur)
    t_prop = time.perf_counter() - t0
    rep = oracle.parity_report(y_fast, ref[K], tol)
    rep = {k: rep.get(k) for k in ("ok", "max_abs_over_max", "row_l2_rel", "allclose", "bit_equal")}
    strict_equal = all(np.array_equal(hops_strict[h], ref[h]) for h in range(K + 1))
    nnz_hat, d = int(norm[0][-1]), x.shape[1]
    cpu = {"value": nnz_hat * d * K / t_prop, "unit": "edge\u00b7featdim/s", "cores": _host_cores(), "kind": kind,
           "normalise_ms": t_norm * 1e3, "propagate_ms": t_prop * 1e3,
           "sample": f"the whole config: scipy fp64 normalisation restated (operators/utils.py:76-88) + {K} hops through the "
                     f"{'reference matmul.c compiled in place' if kind == 'reference' else 'C restatement'}, OpenMP static schedule, "
                     f"one timed run after a warm-up"}
    check = {"against": "oracle.propagate(oracle.laplacian_adj(raw A)) on the host, all of hop K", "fast_order": rep,
             "strict_order_all_hops_bit_equal": bool(strict_equal)}
    return cpu, check


def combine_baseline(hops, d, rows_fast=200_000, rows_loop=4_000):
    """the reference's torch-CPU `_combine` of mean / max / concat / NAFS on the first rows of the same hop matrices
    (oracle/torch_combine.py restates the reference's torch calls, message_op/{mean,max,concat}_message_op.py and the per-node
    Python loop of over_smooth_distance_op.py:27-31): milliseconds and rows/s on this host"""
    from oracle import torch_combine as tc  # test infrastructure: the reported baseline
    H = len(hops)
    n = hops[0].shape[0]
    m = min(n, rows_fast)
    host = [h[:m].cpu().contiguous() for h in hops]
    threads = torch.get_num_threads()
    res = {"threads": threads, "cores": _host_cores(), "kind": "port", "rows": m, "nafs_rows": min(m, rows_loop),
           "sample": f"first {m} rows of the {H} hop matrices (d={d}) through the reference's torch-CPU expressions "
                     f"(oracle/torch_combine.py), one timed run after a warm-up; NAFS: first {min(m, rows_loop)} rows through the "
                     f"reference's per-node Python loop (over_smooth_distance_op.py:27-31)"}
    for name, fn in (("mean", lambda: tc.combine_mean(host, 0, H)), ("max", lambda: tc.combine_max(host, 0, H)),
                     ("concat", lambda: tc.combine_concat(host, 0, H))):
        fn()
        t0 = time.perf_counter()
        fn()
        t = time.perf_counter() - t0
        res[name] = {"ms": t * 1e3, "rows_per_s": m / t}
    small = [h[:min(m, rows_loop)] for h in host]
    t0 = time.perf_counter()
    tc.combine_over_smooth_distance(small)
    t = time.perf_counter() - t0
    res["nafs"] = {"ms": t * 1e3, "rows_per_s": small[0].shape[0] / t}
    return res
