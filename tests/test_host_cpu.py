"""CPU-side tests of the host logic and of the C-ABI boundary (no GPU compute): symbol export, execution plan,
error contract, parameter layout, row-shard arithmetic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import oracle
from conftest import ROOT, has_gpu

from sgl_amd import _lib
from sgl_amd.dist import balanced_bounds, piece_bounds


def header_functions(header="sgl_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int64_t|int|void|const char)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 25, names
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/sgl_hip.h but not exported by libsgl_hip.so"
    assert sorted(_lib.PROTOTYPES) == names, (sorted(set(names) ^ set(_lib.PROTOTYPES)))
    assert _lib.lib().sgl_version() >= 100
    assert _lib.last_error() == "" or isinstance(_lib.last_error(), str)
    # measurement / test support lives in its own library behind its own header: nothing of it in the product ABI
    probe = header_functions("sgl_probe.h")
    assert probe and not set(probe) & set(names) and all(n.startswith(("sgl_probe_", "sgl_synth_", "sgl_mem_")) for n in probe), probe
    assert not [n for n in names if n.startswith(("sgl_probe_", "sgl_synth_", "sgl_mem_"))]
    ph = ctypes.CDLL(_lib.PROBE_LIB_PATH)
    for n in probe:
        assert hasattr(ph, n), f"{n} declared in include/sgl_probe.h but not exported by libsgl_probe.so"
    assert sorted(_lib.PROBE_PROTOTYPES) == probe
    assert not any(hasattr(handle, n) for n in probe), "the product library still exports measurement symbols"
    assert _lib.probe_lib().sgl_probe_last_error() == b""


def test_device_count_never_aborts():
    assert _lib.device_count() >= 0


def test_tuning_knobs():
    _lib.set_tuning("spmm_unroll", 1)
    assert _lib.get_tuning("spmm_unroll") == 1
    _lib.set_tuning("spmm_unroll", 0)
    with pytest.raises(_lib.SglHipError):
        _lib.set_tuning("no_such_knob", 1)


def build_plan(rowptr, item_nnz, long_row_nnz):
    lib = _lib.lib()
    rp = np.ascontiguousarray(rowptr, dtype=np.int64)
    h = ctypes.c_void_p()
    _lib.check(lib.sgl_plan_build(ctypes.byref(h), rp.ctypes.data_as(ctypes.c_void_p), len(rp) - 1, item_nnz, long_row_nnz))
    counts = (ctypes.c_int64 * 8)()
    _lib.check(lib.sgl_plan_counts(h, counts))
    ni, npc, nl = counts[0], counts[1], counts[2]
    items = np.zeros(2 * ni, np.int32)
    pb, pl, pr = np.zeros(npc, np.int64), np.zeros(npc, np.int32), np.zeros(npc, np.int32)
    lr, lf = np.zeros(nl, np.int32), np.zeros(nl + 1, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _lib.check(lib.sgl_plan_export(h, p(items), p(pb), p(pl), p(pr), p(lr), p(lf)))
    lib.sgl_plan_destroy(h)
    return items.reshape(-1, 2), pb, pl, pr, lr, lf, list(counts)


@pytest.mark.parametrize("seed,item_nnz,long_nnz", [(0, 64, 100), (1, 512, 2048), (2, 16, -1), (3, 1, 3)])
def test_plan_partitions_every_row_exactly_once(seed, item_nnz, long_nnz):
    rng = np.random.default_rng(seed)
    n = 3000
    deg = np.minimum(rng.lognormal(1.5, 1.4, n).astype(np.int64), 5000)
    deg[rng.integers(0, n, 200)] = 0            # empty rows
    deg[17] = 7001                               # a very long row
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    items, pb, pl, pr, lr, lf, counts = build_plan(rowptr, item_nnz, long_nnz)
    covered = np.zeros(n, np.int32)
    for b, e in items:
        assert 0 <= b < e <= n and e - b <= 63
        covered[b:e] += 1
        nnz = rowptr[e] - rowptr[b]
        # an item closes as soon as it reaches item_nnz, so it never exceeds item_nnz + one row
        assert nnz < max(item_nnz, 1) + deg[b:e].max() + 1
        if long_nnz > 0:
            assert deg[b:e].max() <= long_nnz
    if long_nnz > 0:
        assert len(lr) == int((deg > long_nnz).sum())
        for k, r in enumerate(lr):
            covered[r] += 1
            ps = range(lf[k], lf[k + 1])
            assert all(pr[q] == r for q in ps)
            assert pb[lf[k]] == rowptr[r] and pb[lf[k + 1] - 1] + pl[lf[k + 1] - 1] == rowptr[r + 1]
            for q in ps:
                assert 0 < pl[q] <= long_nnz
                if q + 1 < lf[k + 1]:
                    assert pb[q] + pl[q] == pb[q + 1]
    else:
        assert len(lr) == 0 and len(pb) == 0
    assert (covered == 1).all()
    assert counts[0] == len(items) and counts[5] == n


def test_plan_of_a_large_matrix_is_built_in_segments():
    """more than 2^21 rows: the rows are cut into 2^20-row segments partitioned by a team of host threads and laid end to end
    (sgl_core.cpp: build_plan).  Same guarantees -- every row exactly once, long rows cut per row -- and the plan does not depend
    on the number of threads: two builds are identical, and equal to the single-segment plans of the segments' row ranges"""
    rng = np.random.default_rng(7)
    n = (1 << 21) + (1 << 20) + 12345
    deg = np.minimum(rng.lognormal(1.0, 1.2, n).astype(np.int64), 4000)
    deg[[5, (1 << 20) - 1, 1 << 20, n - 1]] = 3000                       # long rows at the segment boundaries and at both ends
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    _lib.set_tuning("spmm_heavy_first", 0)
    try:
        items, pb, pl, pr, lr, lf, counts = build_plan(rowptr, 256, 2048)
        again = build_plan(rowptr, 256, 2048)
        parts = [build_plan(rowptr[a:b + 1] - 0, 256, 2048) for a, b in ((0, 1 << 20), (1 << 20, 1 << 21), (1 << 21, 3 << 20), (3 << 20, n))]
    finally:
        _lib.set_tuning("spmm_heavy_first", 1)
    assert np.array_equal(items, again[0]) and np.array_equal(pb, again[1]) and np.array_equal(lf, again[5])
    covered = np.zeros(n, np.int32)
    np.add.at(covered, lr, 1)
    diff = np.zeros(n + 1, np.int64)
    np.add.at(diff, items[:, 0], 1)
    np.add.at(diff, items[:, 1], -1)
    covered += np.cumsum(diff[:-1]).astype(np.int32)
    assert (covered == 1).all() and (items[:, 1] - items[:, 0] <= 63).all() and (np.diff(items[:, 0]) > 0).all()
    assert np.array_equal(lr, np.flatnonzero(deg > 2048)) and lf[0] == 0 and lf[-1] == len(pb)
    assert np.array_equal(pb[lf[:-1]], rowptr[lr]) and np.array_equal(pb[lf[1:] - 1] + pl[lf[1:] - 1], rowptr[lr + 1])
    # no item spans a segment boundary, and each segment's share is the plan the segment has on its own
    for k, (a, part) in enumerate(zip((0, 1 << 20, 1 << 21, 3 << 20), parts)):
        b = min(a + (1 << 20), n)
        mine = items[(items[:, 0] >= a) & (items[:, 0] < b)]
        assert (mine[:, 1] <= b).all() and np.array_equal(mine - a, part[0])
    assert counts[0] == len(items) and counts[1] == len(pb) and counts[5] == n


def test_plan_issues_heavy_items_first_inside_every_xcd_range():
    """an item that ends in a heavy row holds several times item_nnz non-zeros; issued late it is the tail of the launch
    (profiles/r03_probe_small_launch.log).  Inside every XCD's range of the item list the items with >= 2 x item_nnz non-zeros
    come first, longest first, the rest keep their row order -- and the tuning key turns it off"""
    rng = np.random.default_rng(5)
    n, item_nnz, long_nnz = 40000, 64, 600
    deg = np.minimum(rng.lognormal(1.2, 1.5, n).astype(np.int64), 3000)
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    items, *_ = build_plan(rowptr, item_nnz, long_nnz)
    _lib.set_tuning("spmm_heavy_first", 0)
    try:
        plain, *_ = build_plan(rowptr, item_nnz, long_nnz)
    finally:
        _lib.set_tuning("spmm_heavy_first", 1)
    assert (np.diff(plain[:, 0]) > 0).all()                                   # the greedy partition itself: row order
    assert sorted(map(tuple, items)) == sorted(map(tuple, plain))             # the same items, another issue order
    ni = len(items)
    rng_len = (((ni + 3) // 4 + 7) // 8) * 4                                  # what spmm_kernel's XCD remap walks (4 waves / block)
    nnz = rowptr[items[:, 1]] - rowptr[items[:, 0]]
    n_heavy = 0
    for lo in range(0, ni, rng_len):
        seg, seg_nnz = items[lo:lo + rng_len], nnz[lo:lo + rng_len]
        assert sorted(map(tuple, seg)) == sorted(map(tuple, plain[lo:lo + rng_len]))   # nobody leaves its XCD range
        heavy = seg_nnz >= 2 * item_nnz
        k = int(heavy.sum())
        n_heavy += k
        assert heavy[:k].all() and not heavy[k:].any()
        assert (np.diff(seg_nnz[:k]) <= 0).all()
        assert (np.diff(seg[k:, 0]) > 0).all()
    assert n_heavy > 50


def test_plan_rejects_bad_rowptr():
    lib = _lib.lib()
    rp = np.array([0, 5, 3], dtype=np.int64)
    h = ctypes.c_void_p()
    rc = lib.sgl_plan_build(ctypes.byref(h), rp.ctypes.data_as(ctypes.c_void_p), 2, 0, 0)
    assert rc != 0 and "decrease" in _lib.last_error()


def test_plan_empty_matrix():
    items, pb, *_ = build_plan(np.zeros(1, np.int64), 0, 0)
    assert len(items) == 0 and len(pb) == 0


# ---- error contract (tests/golden/g5_errors.json, recorded from the reference) --------------------------
def _raises_like(case, fn):
    if case["raised"] is None:
        r = fn()
        assert type(r).__name__ == case["returned_type"] and str(r) == case["returned_msg"]
    else:
        with pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == case["raised"], (type(ei.value).__name__, case)
        assert str(ei.value) == case["msg"]


def test_error_contract_cpu(goldens):
    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    from sgl_amd.operators.message_op import (IterateLearnableWeightedMessageOp, LearnableWeightedMessageOp,
                                              MeanMessageOp, SimpleWeightedMessageOp)
    g5 = goldens.json("g5_errors")
    x = np.zeros((64, 4), np.float32)
    g = goldens.graph("sym64")
    _raises_like(g5["propagate_dense_adj"], lambda: LaplacianGraphOp(2).propagate(g.toarray(), x))
    _raises_like(g5["ppr_dense_adj"], lambda: PprGraphOp(2).propagate(np.zeros((3, 3)), x))
    _raises_like(g5["aggregate_not_list"], lambda: MeanMessageOp(0, 2).aggregate(torch.zeros(2, 2)))
    _raises_like(g5["aggregate_not_tensor"], lambda: MeanMessageOp(0, 2).aggregate([np.zeros((2, 2)), np.zeros((2, 2))]))
    _raises_like(g5["simple_weighted_bad_type"], lambda: SimpleWeightedMessageOp(0, 2, "nope", 0.5))
    _raises_like(g5["simple_weighted_alpha_int"], lambda: SimpleWeightedMessageOp(0, 2, "alpha", 1))
    _raises_like(g5["simple_weighted_alpha_range"], lambda: SimpleWeightedMessageOp(0, 2, "alpha", 1.5))
    _raises_like(g5["simple_weighted_nargs"], lambda: SimpleWeightedMessageOp(0, 2, "alpha"))
    _raises_like(g5["simple_weighted_hand_bad"], lambda: SimpleWeightedMessageOp(0, 2, "hand_crafted", 3))
    _raises_like(g5["learnable_bad_type"], lambda: LearnableWeightedMessageOp(0, 2, "nope", 1))
    _raises_like(g5["learnable_simple_nargs"], lambda: LearnableWeightedMessageOp(0, 2, "simple"))
    _raises_like(g5["learnable_jk_nargs"], lambda: LearnableWeightedMessageOp(0, 2, "jk", 3))
    _raises_like(g5["iterate_bad_type"], lambda: IterateLearnableWeightedMessageOp(0, 2, "nope", 4))
    _raises_like(g5["iterate_nargs"], lambda: IterateLearnableWeightedMessageOp(0, 2, "recursive"))


def test_aggr_type_strings():
    from sgl_amd.operators import message_op as m
    assert m.LastMessageOp().aggr_type == "last"
    assert m.ConcatMessageOp(0, 2).aggr_type == "concat"
    assert m.MeanMessageOp(0, 2).aggr_type == "mean"
    assert m.SumMessageOp(0, 2).aggr_type == "sum"
    assert m.MaxMessageOp(0, 2).aggr_type == "max"
    assert m.MinMessageOp(0, 2).aggr_type == "min"
    assert m.SimpleWeightedMessageOp(0, 2, "alpha", 0.5).aggr_type == "simple_weighted"
    assert m.LearnableWeightedMessageOp(0, 2, "gate", 4).aggr_type == "learnable_weighted"
    assert m.IterateLearnableWeightedMessageOp(0, 2, "recursive", 4).aggr_type == "iterate_learnable_weighted"
    assert m.ProjectedConcatMessageOp(0, 2, 4, 8, 2).aggr_type == "proj_concat"
    assert m.OverSmoothDistanceWeightedOp().aggr_type == "over_smooth_dis_weighted"


def test_alpha_weights_bit_equal_to_reference_recurrence():
    from sgl_amd.operators.message_op import SimpleWeightedMessageOp
    for a in (0.85, 0.1, 0.5, 1.0, 0.0):
        for (s, e, n) in ((0, 5, 5), (1, 5, 5), (2, 4, 11)):
            w = SimpleWeightedMessageOp(s, e, "alpha", a).weights(n).numpy()
            assert np.array_equal(w, oracle.alpha_weights(a, n, s, e))


@pytest.mark.parametrize("kind", ["simple", "simple_allow_neg", "gate", "ori_ref", "jk"])
def test_learnable_hop_weights_match_reference(goldens, kind):
    """the gate-score decomposition (no repeat/hstack temporaries) reproduces the reference weights,
    including the .view(-1, H) pairing of 'ori_ref' / 'jk' -- pure torch, runs on CPU"""
    from sgl_amd.operators.message_op import LearnableWeightedMessageOp
    g3 = goldens.npz("g3_agg")
    feats = [torch.from_numpy(g3[f"feat{j}"]) for j in range(5)]
    args = {"simple": (4,), "simple_allow_neg": (4,), "gate": (12,), "ori_ref": (12,), "jk": (4, 12)}[kind]
    for (s, e) in ((0, 5), (1, 5)):
        tag = f"learnable|{kind}|{s}_{e}"
        op = LearnableWeightedMessageOp(s, e, kind, *args)
        sd = {k[len(tag) + 7:]: torch.from_numpy(v) for k, v in g3.items() if k.startswith(tag + "|param|")}
        op.load_state_dict(sd)
        w = op.hop_weights(feats).detach().numpy()
        if kind in ("simple", "simple_allow_neg"):
            ref = oracle.learnable_weights([f.numpy() for f in feats], s, e, kind, param=list(sd.values())[0].numpy())
        else:
            ref = oracle.learnable_weights([f.numpy() for f in feats], s, e, kind,
                                           weight=sd["_LearnableWeightedMessageOp__learnable_weight.weight"].numpy(),
                                           bias=sd["_LearnableWeightedMessageOp__learnable_weight.bias"].numpy())
        assert w.shape == ref.shape
        assert np.allclose(w, ref, rtol=1e-5, atol=1e-6), np.abs(w - ref).max()


def test_model_state_dicts_interchange_with_reference(goldens):
    from sgl_amd.models import homo
    g4 = goldens.npz("g4_models")
    K, d, C = 3, 16, 5
    ctor = {"SGC": (K, d, C), "SSGC": (K, d, C), "SIGN": (K, d, C, 32, 2), "GBP": (K, d, C, 32, 2),
            "GAMLP": (K, d, C, 32, 2), "GAMLPRecursive": (K, d, C, 32, 2), "NAFS": (K, d, C),
            "PASCA_V1": (K, d, C, 32, 3), "PASCA_V2": (K, d, C, 32, 3), "PASCA_V3": (K, 2, d, C, 32, 3)}
    for name, args in ctor.items():
        model = getattr(homo, name)(*args)
        ref_keys = sorted(k.split("|param|")[1] for k in g4 if k.startswith(name + "|param|"))
        assert sorted(model.state_dict().keys()) == ref_keys, name
        model.load_state_dict({k.split("|param|")[1]: torch.from_numpy(v) for k, v in g4.items()
                               if k.startswith(name + "|param|")})


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_hot_path_fails_loudly_without_gpu(goldens):
    from sgl_amd.operators.graph_op import LaplacianGraphOp
    from sgl_amd.operators.message_op import MeanMessageOp
    g = goldens.graph("sym64")
    with pytest.raises(_lib.SglHipError):
        LaplacianGraphOp(2).propagate(g, np.zeros((64, 4), np.float32))
    with pytest.raises(_lib.SglHipError):
        MeanMessageOp(0, 2).aggregate([torch.zeros(4, 4), torch.zeros(4, 4)])


def test_sgl_amd_never_imports_the_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "sgl_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_only_the_checkers_use_the_oracle():
    """outside tests/: __graft_entry__.smoke() and the bench's cpu_baseline legs (benchlib/engine.py: the headline's; benchlib/cpu_legs.py:
    the secondary sections') -- nothing under tools/, examples/, the rest of benchlib/ or bench.py itself imports the oracle or loads
    its library"""
    allowed = {os.path.join(ROOT, "benchlib", "engine.py"), os.path.join(ROOT, "benchlib", "cpu_legs.py"), os.path.join(ROOT, "__graft_entry__.py")}
    bad = []
    for top in ("tools", "examples", "benchlib", "sgl_amd"):
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                path = os.path.join(dp, f)
                if f.endswith(".py") and path not in allowed:
                    src = open(path).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src:
                        bad.append(path)
    for f in ("bench.py",):
        src = open(os.path.join(ROOT, f)).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
            bad.append(f)
    assert not bad, bad
    # and where it is allowed it is the baseline / the checker, inside one function each
    eng = open(os.path.join(ROOT, "benchlib", "engine.py")).read()
    assert len(re.findall(r"^\s*import oracle\b", eng, flags=re.M)) == 1 and "def cpu_baseline" in eng
    legs = open(os.path.join(ROOT, "benchlib", "cpu_legs.py")).read()
    assert len(re.findall(r"^\s*(from|import)\s+oracle\b", legs, flags=re.M)) == 2 and "def config1_reference_path" in legs and "def combine_baseline" in legs


# ---- row sharding arithmetic -----------------------------------------------------------------------------
def test_balanced_bounds_properties():
    rng = np.random.default_rng(0)
    deg = rng.lognormal(2.0, 1.3, 5000).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    for parts in (1, 2, 3, 8):
        b = balanced_bounds(rowptr, parts)
        assert b[0] == 0 and b[-1] == 5000 and (np.diff(b) >= 0).all() and len(b) == parts + 1
        nnz = np.diff(rowptr[b])
        assert nnz.max() <= rowptr[-1] / parts + deg.max() + 5000 / parts + 1
    pb = piece_bounds(rowptr, 1000, 3000, 4)
    assert pb[0] == 1000 and pb[-1] == 3000 and (np.diff(pb) >= 0).all()
    # degenerate: more parts than rows
    from sgl_amd.dist import all_piece_bounds, tapered_weights
    rp = np.arange(0, 70001, 7, dtype=np.int64)                      # 10 000 rows of 7 non-zeros
    pbw = all_piece_bounds(rp, 2, 4, tapered_weights(4))
    assert pbw.shape == (2, 5) and pbw[0, 0] == 0 and pbw[1, -1] == 10000 and pbw[0, -1] == pbw[1, 0]
    sizes = np.diff(pbw[0])
    assert abs(sizes[3] / sizes[0] - 0.5) < 0.01 and abs(sizes[1] / sizes[0] - 1) < 0.01
    with pytest.raises(ValueError):
        balanced_bounds(rp, 3, [1, 2])
    b = balanced_bounds(np.array([0, 2, 4], dtype=np.int64), 8)
    assert b[0] == 0 and b[-1] == 2 and (np.diff(b) >= 0).all()


def test_row_pitch_is_line_aware():
    """leading dimensions: always 16-byte aligned rows, never fewer floats than the row, and rounded up to whole
    lines / a power of two exactly when that makes a gathered row touch fewer 128-byte lines"""
    from sgl_amd.device import expected_lines, row_pitch
    for d in range(1, 700):
        for growth in (1.25, 2.0):
            ld = row_pitch(d, growth)
            assert ld >= d and ld % 4 == 0 and ld <= max(4, growth * (d + 3) // 4 * 4 + 4, 1.34 * ((d + 3) // 4 * 4))
            assert expected_lines(ld, d) <= expected_lines((d + 3) // 4 * 4, d)
    assert [row_pitch(d) for d in (100, 128, 147, 500, 12, 13, 16, 50)] == [100, 128, 160, 512, 16, 16, 16, 64]
    assert row_pitch(25) == 32 and row_pitch(20) == 20 and row_pitch(20, growth=2.0) == 32
    assert expected_lines(100, 100) == 4 and expected_lines(128, 100) == 4 and expected_lines(160, 147) == 5


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/sgl_hip.h is the drop-in boundary: it must compile as C99 and as C++17, and a plain C program built
    against it must link with libsgl_hip.so and call an entry point (no GPU needed: version / error / tuning calls)"""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "sgl_hip.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
    src = tmp_path / "use_abi.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "sgl_hip.h"\n'
                   'int main(void) {\n'
                   '  sgl_csr_t *h = NULL;\n'
                   '  int rc = sgl_csr_create(&h, -1, 1, 0, NULL, NULL, NULL, 0, 0, 0, NULL);\n'
                   '  printf("%d|%d|%s\\n", sgl_version(), rc != 0, strlen(sgl_last_error()) > 0 ? "msg" : "nomsg");\n'
                   '  return 0;\n}\n')
    exe = tmp_path / "use_abi"
    libdir = os.path.join(ROOT, "sgl_amd", "csrc")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lsgl_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,--allow-shlib-undefined"])
    out = subprocess.check_output([str(exe)], text=True).strip().split("|")
    assert int(out[0]) > 0 and out[1] == "1" and out[2] == "msg"          # bad arguments -> error code + message, no abort


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_host_side_under_address_and_ub_sanitizers(tmp_path, sanitizer):
    """SURVEY section 5 (sanitizers / race detection): the host side of the C ABI -- plan builder / export / error path / tuning
    table -- rebuilt with g++ -fsanitize=address,undefined and with -fsanitize=thread and driven with adversarial inputs from
    one and from several threads (tests/native/plan_asan.cpp); leaks, overflows, use-after-free, undefined behaviour and data
    races all fail the run."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("needs g++ and the HIP headers")
    exe = str(tmp_path / "plan_asan")
    cmd = [gxx, "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", *(["-fno-sanitize-recover=undefined"] if "undefined" in sanitizer else []),
           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "native", "plan_asan.cpp"), os.path.join(ROOT, "sgl_amd", "csrc", "sgl_core.cpp"),
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "plan_asan: OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_exchange_entry_point_argument_contract():
    """sgl_allgather_rows without a GPU: world 1 is a no-op, bad arguments are reported, nothing aborts"""
    lib = _lib.lib()
    b = (ctypes.c_int64 * 2)(0, 10)
    assert lib.sgl_allgather_rows(None, 0, 1, b, None, 16, None) == 0
    assert lib.sgl_allgather_rows(None, 1, 1, b, None, 16, None) != 0 and "rank" in _lib.last_error()
    b3 = (ctypes.c_int64 * 3)(0, 10, 5)
    assert lib.sgl_allgather_rows(None, 0, 2, b3, None, 16, None) != 0 and "bounds" in _lib.last_error()
    b3 = (ctypes.c_int64 * 3)(0, 10, 20)
    assert lib.sgl_allgather_rows(None, 0, 2, b3, None, 16, None) != 0      # NULL matrix / communicator
    assert isinstance(lib.sgl_exchange_backend(), bytes)

    # the need-aware form: same contract
    so, ro = (ctypes.c_int64 * 2)(0, 0), (ctypes.c_int64 * 2)(5, 5)
    assert lib.sgl_exchange_rows(None, 0, 1, None, so, None, ro, 16, None) == 0
    so3, ro3 = (ctypes.c_int64 * 3)(0, 0, 4), (ctypes.c_int64 * 3)(5, 5, 9)
    assert lib.sgl_exchange_rows(None, 0, 2, None, so3, None, ro3, 16, None) != 0      # NULL buffers / communicator
    assert lib.sgl_exchange_rows(None, 2, 2, None, so3, None, ro3, 16, None) != 0 and "rank" in _lib.last_error()


def test_multi_peer_exchange_against_a_mock_rccl(tmp_path):
    """sgl_allgather_rows / sgl_exchange_rows with 2..8 ranks and no GPU: tests/native/exchange_mock.cpp exports a mock RCCL
    (mailboxes + memcpy) that the library resolves through its dlsym(RTLD_DEFAULT) path, and drives one host thread per rank:
    grouped send / recv loop, peer staggering, offsets, unequal and empty blocks, packed ghost ranges, error propagation."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("needs g++")
    exe = str(tmp_path / "exchange_mock")
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-rdynamic", os.path.join(ROOT, "tests", "native", "exchange_mock.cpp"),
                        "-ldl", "-pthread", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, _lib.LIB_PATH], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "exchange_mock: OK (42 multi-rank cases" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])



def test_adjacency_fingerprint_hashes_every_entry():
    """the operator-level cache key hashes indptr, indices AND data in full (sgl_content_hash, the library's host
    thread team; no optional module) -- an in-place edit of ANY single value or index of a cached scipy matrix is noticed,
    a recycled temporary never matches"""
    import scipy.sparse as sp
    from sgl_amd.operators.base_op import AdjIdentity
    rng = np.random.default_rng(0)
    n, nnz = 4000, 900_000
    m = sp.csr_matrix((rng.random(nnz).astype(np.float32), (rng.integers(0, n, nnz), rng.integers(0, n, nnz))), shape=(n, n))
    ident = AdjIdentity(m)
    assert ident.matches(m) and not ident.matches(m.copy())
    for pos in (1, m.nnz // 2 + 7, m.nnz - 3):           # positions a strided 65 536-sample never looks at
        old = m.data[pos]
        m.data[pos] = old + 1.0
        assert not ident.matches(m)
        m.data[pos] = old
        assert ident.matches(m)
    j = m.nnz // 3 + 11
    old = m.indices[j]
    m.indices[j] = (old + 1) % n
    assert not ident.matches(m)
    m.indices[j] = old
    assert ident.matches(m)
    a = np.arange(3_000_001, dtype=np.int32)             # several 1 MiB blocks + a ragged tail
    h = _lib.content_hash(a)
    assert h == _lib.content_hash(a.copy()) and h != _lib.content_hash(a[::-1]) and _lib.content_hash(a[:0]) != h
    a[2_999_999] ^= 1
    assert _lib.content_hash(a) != h



def test_host_result_pool_recycles_only_unreferenced_buffers():
    """sgl_amd.hostpool (the destinations of host_output=True): a buffer is handed out again only when no tensor, view or numpy
    array derived from it is alive -- results of earlier calls are never overwritten behind the caller's back"""
    from sgl_amd import hostpool
    before = dict(hostpool.stats)
    a = hostpool.take((5000, 64), pinned=False)
    a.fill_(3.0)
    p = a.data_ptr()
    b = hostpool.take((5000, 64), pinned=False)
    assert b.data_ptr() != p and b.shape == (5000, 64) and b.dtype == torch.float32 and b.is_contiguous()
    arr = a[100:200].numpy()                      # a numpy view of a view keeps the buffer out of circulation
    del a
    c = hostpool.take((5000, 64), pinned=False)
    assert c.data_ptr() != p and float(arr[0, 0]) == 3.0
    del arr
    d = hostpool.take((5000, 64), pinned=False)
    assert d.data_ptr() == p                      # nothing references it any more: recycled
    assert hostpool.stats["reused"] == before["reused"] + 1 and hostpool.stats["allocated"] == before["allocated"] + 3
    e = hostpool.take((5000, 60), pinned=False)   # another shape in the same size class may take a free buffer too
    assert e is not None and e.shape == (5000, 60)
    del b, c, d, e
    hostpool.trim()


def test_community_order_on_cpu_tensors():
    """sgl_amd.reorder.community_order_reference (the tensor-code statement of sgl_reorder_community): on a small
    planted-partition graph with shuffled ids it returns a permutation under which most edges join nodes of the same (now
    contiguous) community"""
    import scipy.sparse as sp
    from sgl_amd.reorder import community_order_reference as community_order
    n, bs = 1200, 60
    rng = np.random.default_rng(3)
    a = np.repeat(np.arange(n), 10)
    near = (a // bs) * bs + rng.integers(0, bs, a.size)
    far = rng.integers(0, n, a.size)
    b = np.where(rng.random(a.size) < 0.9, near, far)
    keep = a != b
    m = sp.coo_matrix((np.ones(keep.sum(), np.float32), (a[keep], b[keep])), shape=(n, n)).tocsr()
    m = ((m + m.T) > 0).astype(np.float32).tocsr()
    shuffle = rng.permutation(n)
    P = sp.coo_matrix((np.ones(n, np.float32), (shuffle, np.arange(n))), shape=(n, n)).tocsr()
    adj = (P @ m @ P.T).tocsr()
    adj.sort_indices()
    order, info = community_order(torch.from_numpy(adj.indptr.astype(np.int64)), torch.from_numpy(adj.indices.astype(np.int32)), n)
    o = order.numpy()
    assert np.array_equal(np.sort(o), np.arange(n)), info
    coo = adj.tocoo()
    before = np.mean(np.abs(coo.row - coo.col) < 2 * bs)
    after = np.mean(np.abs(o[coo.row] - o[coo.col]) < 2 * bs)
    assert before < 0.3 and after > 0.8, (before, after, info)


def test_hop_cache_digest_separates_moves_and_swaps():
    """the device-tensor fingerprint of the on-disk hop cache (sgl_amd/hopcache.py): a collision is a cache hit that returns another
    input's hop matrices, so single-element moves, swaps of two values and reorderings must change the key (the round-3 digest was a
    plain position-weighted sum and collided in 22 of 3000 single-element moves: ADVICE r3)"""
    import torch
    from sgl_amd.hopcache import _bits_digest, _tensor_digest
    g = torch.Generator().manual_seed(0)
    base = torch.zeros(4096)
    digests, positions = set(), set()
    for _ in range(3000):
        p = int(torch.randint(0, 4096, (1,), generator=g))
        x = base.clone()
        x[p] = 1.0
        positions.add(p)
        digests.add(_tensor_digest(x))
    assert len(digests) == len(positions)                      # one digest per position of the 1.0: no two one-hot tensors collide
    x = torch.randn(4096, generator=g)
    d0 = _tensor_digest(x)
    for _ in range(3000):
        i, j = torch.randint(0, 4096, (2,), generator=g).tolist()
        if x[i] != x[j]:
            y = x.clone()
            y[i], y[j] = x[j], x[i]
            assert _tensor_digest(y) != d0
    # +a at one place and -a at another (what cancels in a linear digest), a reversal, a changed length
    y = x.clone()
    y[10] += 0.5
    y[2000] -= 0.5
    assert _tensor_digest(y) != d0 and _tensor_digest(x.flip(0)) != d0 and _tensor_digest(x[:-1]) != d0
    assert _tensor_digest(x.clone()) == d0 and d0.bit_length() > 64
    # chunked evaluation gives the same value as one pass (the sum is order-free)
    import sgl_amd.hopcache as hc
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (70_000,), dtype=torch.int64, generator=g).to(torch.int32)
    whole = _bits_digest(bits)
    total = torch.zeros(2, dtype=torch.int64)
    for s in range(0, bits.numel(), 9_999):
        part = bits[s:s + 9_999].to(torch.int64) & 0xFFFFFFFF
        keyed = part + torch.arange(s, s + part.numel(), dtype=torch.int64) * hc._s64(0x9E3779B97F4A7C15)
        total[0] += hc._mix64(keyed).sum()
        total[1] += hc._mix64(keyed ^ hc._s64(0xD6E8FEB86659FD93)).sum()
    lo, hi = (int(v) & 0xFFFFFFFFFFFFFFFF for v in total.tolist())
    assert whole == (hi << 64) | lo


def test_recursive_gate_on_per_hop_scalars_is_the_step_by_step_loop():
    """device.recursive_weights (the [n, H] recursion the GPU kernels implement, plain torch) against the reference's loop as written
    (iterate_learnable_weighted_message_op.py:28-51) in float64: final weights, output and gradients of both score matrices"""
    from sgl_amd import device as dev
    g = torch.Generator().manual_seed(5)
    n, d, H = 64, 9, 7
    feats = [torch.randn(n, d, generator=g, dtype=torch.float64) * (1.0 - 0.05 * h) for h in range(H)]
    weight = torch.randn(1, 2 * d, generator=g, dtype=torch.float64) * 0.4
    bias = torch.randn(1, generator=g, dtype=torch.float64)
    acc, weights = feats[0], None
    for i in range(H):                                                     # the reference's loop
        score = torch.sigmoid(torch.hstack((feats[i], acc)) @ weight.view(-1, 1) + bias)
        weights = score if weights is None else torch.hstack((weights, score))
        weights = torch.softmax(weights, dim=1)
        acc = sum(weights[:, j:j + 1] * feats[j] for j in range(i + 1))
    a = torch.stack([f @ weight[0, :d] for f in feats], dim=1).requires_grad_(True)
    c = torch.stack([f @ weight[0, d:] for f in feats], dim=1).requires_grad_(True)
    w = dev.recursive_weights(a, c, bias)
    assert torch.allclose(w, weights, rtol=1e-12, atol=1e-13)
    assert torch.allclose(sum(w[:, j:j + 1] * feats[j] for j in range(H)), acc, rtol=1e-12, atol=1e-12)
    w.sum(dim=0)[1:4].sum().backward()
    assert float(a.grad[:, 0].abs().max()) == 0.0                          # step 0 is the soft-max of ONE score: a_0 is never seen
    assert float(c.grad[:, 0].abs().max()) > 0.0 and float(a.grad[:, 1:].abs().max()) > 0.0


def test_shared_slope_prelu_is_nn_prelu_with_another_backward():
    """models/simple_models.py keeps nn.PReLU's parameter, key and forward; its backward (three element-wise passes and a sum) gives
    torch's gradients, also at x == 0 and with the slope frozen"""
    from sgl_amd.models.simple_models import MultiLayerPerceptron, _SharedSlopePReLU
    g = torch.Generator().manual_seed(1)
    ours, ref = _SharedSlopePReLU(), torch.nn.PReLU()
    x = torch.randn(50, 13, generator=g)
    x[0, :3] = 0.0
    go = torch.randn(50, 13, generator=g)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = ours(xa), ref(xb)
    assert torch.equal(ya, yb)
    (ya * go).sum().backward()
    (yb * go).sum().backward()
    assert torch.equal(xa.grad, xb.grad) and torch.allclose(ours.weight.grad, ref.weight.grad, rtol=1e-6, atol=1e-7)
    ours.weight.requires_grad_(False)
    xa2 = x.clone().requires_grad_(True)
    (ours(xa2) * go).sum().backward()
    assert torch.equal(xa2.grad, xb.grad)
    with torch.no_grad():
        assert torch.equal(ours(x), ref(x))
    assert "_MultiLayerPerceptron__prelu.weight" in MultiLayerPerceptron(4, 8, 2, 3).state_dict()


# ---- `sgl` namespace alias (sgl_amd/compat.py): the reference's model files consume the path unchanged --------------------------
REFERENCE_ROOT = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE_ROOT, "sgl", "models", "homo")),
                    reason="needs the reference checkout (build container only; nothing of it is copied or shipped)")
def test_reference_model_files_run_unchanged_on_the_sgl_alias(goldens, monkeypatch):
    """sgl_amd.compat.install(reference_root) registers sgl.operators[.graph_op / .message_op / .base_op / .utils],
    sgl.models.base_model and sgl.models.simple_models as aliases of the sgl_amd modules; the ten files of the reference's
    sgl/models/homo/ are then imported AS THEY ARE (their four import lines, e.g. sgc.py:1-4, resolve to sgl_amd) and the classes
    they define -- the reference's own code objects -- are built on sgl_amd operators: same state_dict keys as the golden record G4
    (recorded from the reference's own classes), G4's parameters load, and with the oracle's pre-propagation result in place of the
    GPU's the logits equal G4's (models whose aggregator is not learnable; the learnable ones aggregate per batch with the HIP
    kernels and fail loudly here, where there is no GPU)."""
    import inspect
    import sys
    import oracle
    from inputs import hash_matrix
    from sgl_amd import compat
    monkeypatch.setattr(sys, "dont_write_bytecode", True)            # the reference tree is read-only
    assert not any(k == "sgl" or k.startswith("sgl.") for k in sys.modules)
    try:
        names = compat.install(REFERENCE_ROOT)
        assert set(names) == {"sgl.operators", "sgl.operators.base_op", "sgl.operators.utils", "sgl.operators.graph_op",
                              "sgl.operators.message_op", "sgl.models.base_model", "sgl.models.simple_models"}
        import sgl.operators.graph_op as ref_named
        import sgl_amd.operators.graph_op as ours
        assert ref_named is ours
        from sgl.operators.message_op import LearnableWeightedMessageOp as L1
        from sgl_amd.operators.message_op import LearnableWeightedMessageOp as L2
        assert L1 is L2
        g4 = goldens.npz("g4_models")
        g = goldens.graph("pl2000")
        n, d, C, K = 2000, 16, 5, 3
        x = hash_matrix(n, d, seed=4242)
        hops = oracle.propagate(oracle.laplacian_adj(g.indptr, g.indices, g.data, n, 0.5), x, K)
        folded = {"SGC": oracle.agg_last(hops), "SSGC": oracle.agg_mean(hops, 0, K + 1), "SIGN": oracle.agg_concat(hops, 0, K + 1),
                  "GBP": oracle.agg_simple_weighted(hops, 0, K + 1, "alpha", 0.85), "NAFS": oracle.agg_over_smooth_distance(hops)}
        ctor = {"SGC": (K, d, C), "SSGC": (K, d, C), "SIGN": (K, d, C, 32, 2), "GBP": (K, d, C, 32, 2),
                "GAMLP": (K, d, C, 32, 2), "GAMLPRecursive": (K, d, C, 32, 2), "NAFS": (K, d, C),
                "PASCA_V1": (K, d, C, 32, 3), "PASCA_V2": (K, d, C, 32, 3), "PASCA_V3": (K, 2, d, C, 32, 3)}
        idx = g4["idx"]
        for name, args in ctor.items():
            cls = compat.load_reference_model(name, REFERENCE_ROOT)
            assert inspect.getsourcefile(cls).startswith(REFERENCE_ROOT + "/sgl/models/homo/"), name      # the reference's file, unchanged
            assert cls.__mro__[1].__module__ == "sgl_amd.models.base_model", name
            model = cls(*args)
            assert type(model._pre_graph_op).__module__.startswith("sgl_amd.operators.graph_op"), name
            assert type(model._pre_msg_op).__module__.startswith("sgl_amd.operators.message_op"), name
            keys = sorted(k.split("|param|")[1] for k in g4 if k.startswith(name + "|param|"))
            assert sorted(model.state_dict().keys()) == keys, name
            model.load_state_dict({k.split("|param|")[1]: torch.from_numpy(v) for k, v in g4.items() if k.startswith(name + "|param|")})
            model.eval()
            if name in folded:
                model._pre_msg_learnable = False
                model._processed_feature = torch.from_numpy(np.ascontiguousarray(folded[name]))
                with torch.no_grad():
                    y = model.model_forward(idx, torch.device("cpu"))
                assert oracle.parity_ok(y.numpy(), g4[f"{name}|out"], 1e-5, rowwise=False), name
            elif not has_gpu():
                model._pre_msg_learnable = True
                model._processed_feat_list = [torch.from_numpy(h) for h in hops]
                with pytest.raises(_lib.SglHipError):
                    model.model_forward(idx, torch.device("cpu"))
        with pytest.raises(FileNotFoundError):
            compat.install("/nonexistent/checkout")
    finally:
        compat.uninstall()
    assert not any(k == "sgl" or k.startswith("sgl.") for k in sys.modules)


def test_sgl_alias_without_a_reference_checkout():
    """install() with no reference_root: the operator / base-model modules under their reference names, nothing else; a model written
    against the reference's API (the body of sgl/models/homo/sgc.py:7-13, typed here) builds on them; a real `sgl` that is already
    imported is not silently shadowed"""
    import sys
    import types
    from sgl_amd import compat
    try:
        compat.install()
        from sgl.models.base_model import BaseSGAPModel
        from sgl.models.simple_models import LogisticRegression
        from sgl.operators.graph_op import LaplacianGraphOp
        from sgl.operators.message_op import LastMessageOp

        class MySGC(BaseSGAPModel):
            def __init__(self, prop_steps, feat_dim, output_dim):
                super(MySGC, self).__init__(prop_steps, feat_dim, output_dim)
                self._pre_graph_op = LaplacianGraphOp(prop_steps, r=0.5)
                self._pre_msg_op = LastMessageOp()
                self._base_model = LogisticRegression(feat_dim, output_dim)
        m = MySGC(2, 8, 3)
        assert type(m._pre_graph_op).__module__.startswith("sgl_amd.") and sys.modules["sgl"].__sgl_amd_shell__
        with pytest.raises(ModuleNotFoundError):
            import sgl.models.homo.sgc  # noqa: F401  (no checkout was named: the shells have an empty path)
    finally:
        compat.uninstall()
    fake = types.ModuleType("sgl")
    sys.modules["sgl"] = fake
    try:
        with pytest.raises(RuntimeError):
            compat.install()
        compat.install(force=True)
        assert sys.modules["sgl.operators.graph_op"].__name__ == "sgl_amd.operators.graph_op" and sys.modules["sgl"] is fake
    finally:
        compat.uninstall()
        sys.modules.pop("sgl", None)
        for k in [k for k in sys.modules if k.startswith("sgl.")]:
            del sys.modules[k]
