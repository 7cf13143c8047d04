"""Pin the CPU oracle (oracle/) to the reference: every golden vector in tests/golden/ was
produced by importing the reference itself (make_goldens.py).  CPU-only; runs everywhere."""
import os

import numpy as np
import pytest

import oracle
from inputs import hash_matrix

G1_VARIANTS = [("lap", r, None) for r in (0.0, 0.3, 0.5, 1.0)] + \
              [("ppr", 0.5, a) for a in (0.1, 0.15, 0.2, 0.3)] + [("ppr", 0.3, 0.15)]


@pytest.mark.parametrize("gname", ["sym64", "dir40", "pl2000"])
def test_g1_normalisation_matches_reference(goldens, gname):
    g = goldens.graph(gname)
    g1 = goldens.npz("g1_norm")
    n = g.shape[0]
    for kind, r, a in G1_VARIANTS:
        key = f"{gname}|{kind}|{r}" + ("" if a is None else f"|{a}")
        ptr, col, val = oracle.sym_norm_csr(g.indptr, g.indices, g.data, n, r, a)
        # structure: integer work -> bit exact
        assert np.array_equal(ptr, g1[gname + "|indptr"]), key
        assert np.array_equal(col, g1[gname + "|indices"]), key
        ref = g1[key]
        # fp64 values: same operation order -> expect bit equality; allow 2 ulp(fp64) for libm pow
        assert np.allclose(val, ref, rtol=5e-16, atol=0), key
        # what the SpMM consumes (fp32-rounded, operators/utils.py:32) must be identical
        assert np.array_equal(val.astype(np.float32), ref.astype(np.float32)), key


def _g2_cases(goldens):
    return goldens.json("g2_prop")


def test_g2_propagate_bit_exact(goldens):
    g2 = goldens.npz("g2_prop")
    meta = goldens.json("g2_prop")
    for key, m in meta.items():
        g = goldens.graph(m["graph"])
        n = g.shape[0]
        x = hash_matrix(n, m["d"], seed=m["seed"], order=m["order"])
        norm = oracle.sym_norm_csr(g.indptr, g.indices, g.data, n, m["r"], m["alpha"])
        feats = oracle.propagate(norm, x, m["K"])
        hops = range(1, m["K"] + 1) if m["keep"] == "all" else [m["K"]]
        for h in hops:
            assert np.array_equal(feats[h], g2[f"{key}|h{h}"]), f"{key} hop {h} not bit-equal to reference"
        sums = np.array([f.astype(np.float64).sum() for f in feats])
        assert np.array_equal(sums, g2[f"{key}|sums"]), key


def test_spmm_oracle_equals_reference_binary(goldens):
    """C restatement == the reference's own compiled matmul.c (oracle/_ref), bit for bit."""
    if oracle.load_reference_lib() is None:
        pytest.skip("oracle/_ref/libmatmul.so not present (built only where /root/reference exists)")
    for gname, d in (("pl2000", 37), ("dir40", 5), ("pl256", 128)):
        g = goldens.graph(gname)
        n = g.shape[0]
        ptr, col, val = oracle.sym_norm_csr(g.indptr, g.indices, g.data, n, 0.5)
        x = hash_matrix(n, d, seed=9)
        a = oracle.oracle_spmm(ptr, col, val, x)
        b = oracle.reference_spmm(ptr, col, val, x)
        c = oracle.oracle_spmm_scalar(ptr, col, val, x)
        assert np.array_equal(a, b)
        assert np.array_equal(a, c)


def _feats(goldens):
    g3 = goldens.npz("g3_agg")
    return [g3[f"feat{j}"] for j in range(5)], g3


def test_g3_simple_aggregators(goldens):
    feats, g3 = _feats(goldens)
    H = 5
    assert np.array_equal(oracle.agg_last(feats), g3["last"])
    for (s, e) in ((0, H), (1, H - 1)):
        tag = f"{s}_{e}"
        assert np.array_equal(oracle.agg_concat(feats, s, e), g3[f"concat|{tag}"])
        assert np.array_equal(oracle.agg_sum(feats, s, e), g3[f"sum|{tag}"])
        assert np.array_equal(oracle.agg_mean(feats, s, e), g3[f"mean|{tag}"])
        assert np.array_equal(oracle.agg_max(feats, s, e), g3[f"max|{tag}"])
        assert np.array_equal(oracle.agg_min(feats, s, e), g3[f"min|{tag}"])


def test_g3_weighted_aggregators(goldens):
    feats, g3 = _feats(goldens)
    H = 5
    for (s, e) in ((0, H), (1, H)):
        y = oracle.agg_simple_weighted(feats, s, e, "alpha", 0.85)
        assert oracle.parity_ok(y, g3[f"simple_weighted|alpha0.85|{s}_{e}"], 1e-6)
    y = oracle.agg_simple_weighted(feats, 0, H, "hand_crafted", g3["simple_weighted|hand_crafted|w"])
    assert oracle.parity_ok(y, g3["simple_weighted|hand_crafted|0_5"], 1e-6)
    y = oracle.agg_over_smooth_distance(feats)
    assert oracle.parity_ok(y, g3["over_smooth"], 1e-6)


@pytest.mark.parametrize("kind", ["simple", "simple_allow_neg", "gate", "ori_ref", "jk"])
def test_g3_learnable_forward(goldens, kind):
    feats, g3 = _feats(goldens)
    H = 5
    for (s, e) in ((0, H), (1, H)):
        tag = f"learnable|{kind}|{s}_{e}"
        if kind in ("simple", "simple_allow_neg"):
            p = g3[tag + "|param|_LearnableWeightedMessageOp__learnable_weight"]
            y = oracle.agg_learnable_weighted(feats, s, e, kind, param=p)
        else:
            w = g3[tag + "|param|_LearnableWeightedMessageOp__learnable_weight.weight"]
            b = g3[tag + "|param|_LearnableWeightedMessageOp__learnable_weight.bias"]
            y = oracle.agg_learnable_weighted(feats, s, e, kind, weight=w, bias=b)
        rep = oracle.parity_report(y, g3[tag + "|out"], 1e-5)
        assert rep["ok"], (tag, rep)


def test_g3_iterate(goldens):
    feats, g3 = _feats(goldens)
    w = g3["iterate|0_5|param|_IterateLearnableWeightedMessageOp__learnable_weight.weight"]
    b = g3["iterate|0_5|param|_IterateLearnableWeightedMessageOp__learnable_weight.bias"]
    y = oracle.agg_iterate_learnable(feats, 0, 5, w, b)
    rep = oracle.parity_report(y, g3["iterate|0_5|out"], 1e-5)
    assert rep["ok"], rep


def test_parity_metric_rejects_wrong():
    ref = hash_matrix(50, 8, seed=3)
    bad = ref.copy()
    bad[7, 3] += 1e-3
    assert oracle.parity_ok(ref, ref)
    assert not oracle.parity_ok(bad, ref)
    assert not oracle.parity_ok(ref[:10], ref)


def test_g6_label_propagation_and_correct_smooth(goldens):
    g6 = goldens.npz("g6_consumers")
    g = goldens.graph("pl2000")
    n = g.shape[0]
    norm = oracle.sym_norm_csr(g.indptr, g.indices, g.data, n, 0.5)
    lab, mask = g6["lp|labels"], g6["lp|mask"]
    y = oracle.label_propagation(lab, norm, 5, 0.8, mask=mask)
    assert oracle.parity_ok(y, g6["lp|long_masked"], 1e-5, rowwise=False)
    y = oracle.label_propagation(lab, norm, 3, 0.5)
    assert oracle.parity_ok(y, g6["lp|long_nomask"], 1e-5, rowwise=False)
    soft = oracle.softmax32(hash_matrix(n, 5, seed=32) * np.float32(3), 1)
    y = oracle.label_propagation(soft - np.float32(0.3), norm, 4, 0.9, clamp=(-1.0, 1.0))
    assert oracle.parity_ok(y, g6["lp|float_clamp11"], 1e-5, rowwise=False)
    for autoscale in (True, False):
        y1 = oracle.cs_correct(soft, lab, mask, norm, 4, 0.9, autoscale=autoscale, scale=1.5)
        rep = oracle.parity_report(y1, g6[f"cs|autoscale{int(autoscale)}|correct"], 1e-5, rowwise=False)
        assert rep["ok"], (autoscale, rep)
        y2 = oracle.cs_smooth(y1, lab, mask, norm, 3, 0.7)
        rep = oracle.parity_report(y2, g6[f"cs|autoscale{int(autoscale)}|smooth"], 1e-5, rowwise=False)
        assert rep["ok"], (autoscale, rep)


def test_g6_nafs_task_pipeline(goldens):
    from inputs import hash_positive
    g6 = goldens.npz("g6_consumers")
    g = goldens.graph("pl256")
    x = hash_positive(256, 8, seed=33)
    for method in ("mean", "max", "concat", "simple"):
        y = oracle.nafs_task_features(g.indptr, g.indices, g.data, 256, x, 3, [0.5, 0.4, 0.3, 0.2, 0.1, 0], method)
        rep = oracle.parity_report(y, g6[f"nafs_task|{method}|hops3"], 1e-5)
        assert rep["ok"], (method, rep)


def test_g11_nafs_hop_sweep(goldens):
    """the reference's per-hop-count loop for hops in {0, 1, 3, 6}, every ensemble method; and the identity the one-pass device
    sweep rests on: softmax over a PREFIX of the cosine scores = running e^c numerator / denominator (|c| <= 1, no max needed)"""
    g11 = goldens.npz("g11_nafs_sweep")
    g = goldens.graph("pl256")
    x, hops, r_list = g11["x"], [int(h) for h in g11["hops"]], [float(r) for r in g11["r_list"]]
    for method in ("mean", "max", "concat", "simple"):
        sweep = oracle.nafs_task_sweep(g.indptr, g.indices, g.data, 256, x, hops, r_list, method)
        for h in hops:
            rep = oracle.parity_report(sweep[h], g11[f"nafs_sweep|{method}|hops{h}"], 1e-5)
            assert rep["ok"], (method, h, rep)
    feats = oracle.propagate(oracle.sym_norm_csr(g.indptr, g.indices, g.data, 256, 0.3), x, 6)
    c = np.stack([np.einsum("nd,nd->n", feats[0].astype(np.float64), f.astype(np.float64)) /
                  (np.linalg.norm(f.astype(np.float64), axis=1) + 1e-10) / (np.linalg.norm(feats[0].astype(np.float64), axis=1) + 1e-10)
                  for f in feats], 1)
    assert np.abs(c).max() <= 1 + 1e-12
    num, den = np.zeros_like(feats[0], dtype=np.float64), np.zeros(256)
    for h in range(7):
        num += np.exp(c[:, h])[:, None] * feats[h]
        den += np.exp(c[:, h])
        want = oracle.agg_over_smooth_distance(feats[:h + 1])
        assert oracle.parity_ok((num / den[:, None]).astype(np.float32), want, 1e-6), h


def test_g7_ingest_coo_to_csr(goldens):
    g7 = goldens.npz("g7_ingest")
    n = int(g7["n"])
    ptr, idx, val = oracle.coo_to_csr(g7["row"], g7["col"], g7["data"], n)
    assert np.array_equal(ptr, g7["indptr"]) and np.array_equal(idx, g7["indices"])     # structure: bit exact
    # scipy sorts duplicates with an UNSTABLE sort before adding them, so runs of >= 3 equal pairs may be summed in a
    # different order: 1 ulp there, exact everywhere else
    diff = val != g7["values"]
    assert diff.sum() <= 2 and np.allclose(val, g7["values"], rtol=3e-7, atol=0)


def test_hop_range_quirks_match_reference(goldens):
    """G8: Mean divides by (end - start) whatever the slice held, partial ranges, a single hop -- straight and on top of a
    real propagation (what the fused propagate_reduce path of sgl_amd must reproduce)"""
    g8 = goldens.npz("g8_ranges")
    g3 = goldens.npz("g3_agg")
    feats = [g3[f"feat{j}"] for j in range(5)]
    for key in g8:
        if key.startswith("mean|") or key.startswith("sum|"):
            kind, rng = key.split("|")
            s_, e_ = (int(t) for t in rng.split("_"))
            got = oracle.agg_mean(feats, s_, e_) if kind == "mean" else oracle.agg_sum(feats, s_, e_)
            assert np.array_equal(got, g8[key]), key
    g = goldens.graph("pl256")
    x = hash_matrix(256, 20, seed=808)
    for name, norm in (("lap", oracle.laplacian_adj(g.indptr, g.indices, g.data, 256, 0.5)),
                       ("ppr", oracle.ppr_adj(g.indptr, g.indices, g.data, 256, 0.3, 0.2))):
        hops = oracle.propagate(norm, x, 4)
        assert np.array_equal(oracle.agg_last(hops), g8[f"prop|{name}|last"])
        assert np.array_equal(oracle.agg_sum(hops, 0, 5), g8[f"prop|{name}|sum|0_5"])
        assert np.array_equal(oracle.agg_sum(hops, 1, 3), g8[f"prop|{name}|sum|1_3"])
        assert np.array_equal(oracle.agg_mean(hops, 0, 5), g8[f"prop|{name}|mean|0_5"])
        assert np.array_equal(oracle.agg_mean(hops, 0, 10), g8[f"prop|{name}|mean|0_10"])
        assert np.array_equal(oracle.agg_mean(hops, 2, 4), g8[f"prop|{name}|mean|2_4"])
        for a_, s_ in ((0.85, 0), (0.3, 1)):
            got = oracle.agg_simple_weighted(hops, s_, 5, "alpha", a_)
            assert oracle.parity_ok(got, g8[f"prop|{name}|simple_weighted|alpha{a_}|{s_}_5"], 1e-6)


# ---- G9: BASELINE config 5 at its own hop count (PPR / Laplacian k = 10; every MessageOp over H = 11 hops) ---------------------
def test_g9_ppr_and_laplacian_k10_bit_exact(goldens):
    g9 = goldens.npz("g9_config5")
    meta = goldens.json("g9_config5")["prop"]
    assert sum(1 for m in meta.values() if m["kind"] == "ppr" and m["K"] == 10) >= 7
    for key, m in meta.items():
        g = goldens.graph(m["graph"])
        n = g.shape[0]
        x = hash_matrix(n, m["d"], seed=m["seed"])
        norm = oracle.sym_norm_csr(g.indptr, g.indices, g.data, n, m["r"], m["alpha"])
        feats = oracle.propagate(norm, x, m["K"])
        for h in m["keep"]:
            assert np.array_equal(feats[h], g9[f"prop|{key}|h{h}"]), f"{key} hop {h} not bit-equal to reference"
        sums = np.array([f.astype(np.float64).sum() for f in feats])
        assert np.array_equal(sums, g9[f"prop|{key}|sums"]), key


@pytest.mark.parametrize("d", [16, 128])
def test_g9_every_message_op_over_eleven_hops(goldens, d):
    g9 = goldens.npz("g9_config5")
    P, H = f"agg|d{d}|", 11
    feats = [g9[P + f"feat{j}"] for j in range(H)]
    assert np.array_equal(oracle.agg_last(feats), g9[P + "last"])
    for (s, e) in ((0, H), (1, H - 1)):
        tag = f"{s}_{e}"
        assert np.array_equal(oracle.agg_concat(feats, s, e), g9[P + f"concat|{tag}"])
        assert np.array_equal(oracle.agg_sum(feats, s, e), g9[P + f"sum|{tag}"])
        assert np.array_equal(oracle.agg_mean(feats, s, e), g9[P + f"mean|{tag}"])
        assert np.array_equal(oracle.agg_max(feats, s, e), g9[P + f"max|{tag}"])
        assert np.array_equal(oracle.agg_min(feats, s, e), g9[P + f"min|{tag}"])
    for (s, e) in ((0, H), (1, H)):
        y = oracle.agg_simple_weighted(feats, s, e, "alpha", 0.85)
        assert oracle.parity_ok(y, g9[P + f"simple_weighted|alpha0.85|{s}_{e}"], 1e-6)
    assert oracle.parity_ok(oracle.agg_over_smooth_distance(feats), g9[P + "over_smooth"], 1e-6)
    for kind in ("simple", "simple_allow_neg", "gate", "ori_ref", "jk"):
        for (s, e) in ((0, H), (1, H)):
            tag = P + f"learnable|{kind}|{s}_{e}"
            if kind in ("simple", "simple_allow_neg"):
                p = g9[tag + "|param|_LearnableWeightedMessageOp__learnable_weight"]
                assert p.shape == (H,)
                y = oracle.agg_learnable_weighted(feats, s, e, kind, param=p)
            else:
                w = g9[tag + "|param|_LearnableWeightedMessageOp__learnable_weight.weight"]
                b = g9[tag + "|param|_LearnableWeightedMessageOp__learnable_weight.bias"]
                assert w.shape == (1, {"gate": d, "ori_ref": 2 * d, "jk": (H + 1) * d}[kind])      # jk: Linear(d + (K+1) d, 1)
                y = oracle.agg_learnable_weighted(feats, s, e, kind, weight=w, bias=b)
            rep = oracle.parity_report(y, g9[tag + "|out"], 1e-5)
            assert rep["ok"], (tag, rep)
    w = g9[P + f"iterate|0_{H}|param|_IterateLearnableWeightedMessageOp__learnable_weight.weight"]
    b = g9[P + f"iterate|0_{H}|param|_IterateLearnableWeightedMessageOp__learnable_weight.bias"]
    rep = oracle.parity_report(oracle.agg_iterate_learnable(feats, 0, H, w, b), g9[P + f"iterate|0_{H}|out"], 1e-5)
    assert rep["ok"], rep


def test_g10_label_use_features_and_their_propagation(goldens):
    """G10 (BASELINE config 3, recorded through the reference's own task loop): the epochs' fresh [x || one-hot] matrices
    (tasks/utils.py:33-36) and their K = 5 pre-propagation are pure path work -- the oracle reproduces the recorded column sums
    and hop sums exactly; the label-reuse iterations in between involve the dense head and are checked on the GPU."""
    g10 = goldens.npz("g10_label_reuse")
    n, d, C, K = (int(g10[k]) for k in ("n", "d", "C", "K"))
    g = goldens.graph("pl2000")
    norm = oracle.laplacian_adj(g.indptr, g.indices, g.data, n, 0.5)
    x = hash_matrix(n, d, seed=1010)
    labels, sub = g10["labels"], g10["sub_rows"]
    for epoch, call in ((0, 0), (1, 1), (2, 4)):
        idx = g10[f"epoch{epoch}|train_labels_idx"]
        onehot = np.zeros((n, C), dtype=np.float32)
        onehot[idx, labels[idx]] = 1
        f = np.concatenate([x, onehot], axis=1)
        assert np.array_equal(f[sub, d:], g10[f"call{call}|label_cols_sub"])
        assert np.array_equal(f.astype(np.float64).sum(0), g10[f"call{call}|feature_colsum"])
        hops = oracle.propagate(norm, f, K)
        assert np.array_equal(np.array([h.astype(np.float64).sum() for h in hops]), g10[f"call{call}|hop_sums"]), call


def test_truth_report_rule():
    """the derived tolerance: at most `factor` times the reference's own float32 distance from the float64 truth, with a floor of
    16 float32 roundings of the largest entry, and -- for sums of cancelling terms -- of the sum of their absolute values"""
    truth = np.array([1.0, -2.0, 1e-3])
    ref = truth + np.array([1e-6, 0.0, 0.0])                 # the reference is 1e-6 / 2 = 5e-7 (relative to max|truth|) off
    assert oracle.truth_report(truth + np.array([0.0, 1.9e-6, 0.0]), ref, truth)["ok"]                    # within 2 x
    floor_abs = oracle.TRUTH_FLOOR * 2.0
    assert oracle.truth_report(truth + np.array([0.0, 0.0, 0.9 * floor_abs]), truth, truth)["ok"]         # exact reference: the floor
    bad = oracle.truth_report(truth + np.array([0.0, 0.0, 3e-6]), ref, truth)
    assert not bad["ok"] and bad["err_got"] > bad["bound"] and abs(bad["err_ref"] - 5e-7) < 1e-12
    # a cancelled scalar: tiny value, large sum of absolute terms -> the condition-aware bound applies to that element only
    t2, c2 = np.array([1e-4, 5.0]), np.array([40.0, 5.0])
    assert not oracle.truth_report(t2 + np.array([2e-5, 0.0]), t2, t2)["ok"]
    assert oracle.truth_report(t2 + np.array([2e-5, 0.0]), t2, t2, cond=c2)["ok"]
    assert not oracle.truth_report(t2 + np.array([2e-5, 2e-5]), t2, t2, cond=c2)["ok"]
    assert not oracle.truth_report(np.zeros(2), np.zeros(3), np.zeros(3))["ok"]
    # the two relaxations of parity_report apply to one-column outputs only
    rng = np.random.default_rng(0)
    ref = rng.standard_normal((50, 8)).astype(np.float32)
    ref[3] *= 1e-6                                           # a tiny row: its relative row error is large for a fixed absolute error
    y = ref + np.float32(2e-6) * np.abs(ref).max() * np.sign(ref)
    assert not oracle.parity_ok(y, ref, 1e-5) and not oracle.parity_ok(y, ref, 1e-5, rowwise=False)       # wide output: unrelaxed
    assert oracle.parity_ok(y[:, :1], ref[:, :1], 1e-5, rowwise=False) and not oracle.parity_ok(y[:, :1], ref[:, :1], 1e-5)


def test_torch_combine_equals_the_numpy_oracle(goldens):
    """oracle/torch_combine.py (the torch-CPU execution of the reference's _combine bodies, timed by the bench's cpu_baseline leg)
    against the golden vectors recorded from the reference and against the numpy restatement"""
    import torch
    from oracle import torch_combine as tc
    feats, g3 = _feats(goldens)
    tf = [torch.from_numpy(np.ascontiguousarray(f)) for f in feats]
    H = 5
    for (s, e) in ((0, H), (1, H - 1)):
        tag = f"{s}_{e}"
        assert np.array_equal(tc.combine_concat(tf, s, e).numpy(), g3[f"concat|{tag}"])
        assert np.array_equal(tc.combine_sum(tf, s, e).numpy(), g3[f"sum|{tag}"])
        assert np.array_equal(tc.combine_mean(tf, s, e).numpy(), g3[f"mean|{tag}"])
        assert np.array_equal(tc.combine_max(tf, s, e).numpy(), g3[f"max|{tag}"])
        assert np.array_equal(tc.combine_min(tf, s, e).numpy(), g3[f"min|{tag}"])
    loop = tc.combine_over_smooth_distance(tf).numpy()
    assert np.array_equal(loop, g3["over_smooth"])                           # the reference's own loop, restated: bit for bit
    assert oracle.parity_ok(tc.combine_over_smooth_distance_vectorised(tf).numpy(), g3["over_smooth"], 1e-6)
    assert oracle.parity_ok(loop, oracle.agg_over_smooth_distance(feats), 1e-6)
