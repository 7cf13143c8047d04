#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); nothing of the reference travels:
the fixtures hold inputs/outputs (data) only.  Re-run with

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

Fixture families:
  graphs.npz   the small test graphs (scipy CSR arrays)
  g1_norm.npz  normalised adjacency values (fp64) of LaplacianGraphOp / PprGraphOp._construct_adj
  g2_prop.npz  GraphOp.propagate outputs (through the reference's ctypes -> libmatmul.so path)
  g3_agg.npz   every MessageOp.aggregate output (+ parameter / input gradients for learnable ops)
  g4_models.npz  SGC / GAMLP / NAFS / ... preprocess + model_forward outputs with saved params
  g5_errors.json the exception contract of propagate / aggregate
  g12_fp64_truth.npz  float64 truth (reference modules in .double()) for the learnable ops of g3 / g9 and the g4 logits
  g11            the NAFS task's hop sweep (hops in {0, 1, 3, 6} x every ensemble method)
  g6 / g7 / g8   consumers of the SpMM (label propagation, C&S, NAFS task), ingest, hop-range quirks
  g10_label_reuse.npz BASELINE config 3: the label use / reuse loop around preprocess, through the reference's task code
  g9_config5.npz BASELINE config 5 at its own hop count: PPR / Laplacian k = 10, every MessageOp over H = 11 hops (d = 16, 128)
Dense inputs are regenerated from tests/golden/inputs.py (integer hash), not stored.
"""
import importlib.util
import json
import os
import sys
import types
from unittest.mock import MagicMock

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import scipy.sparse as sp
import torch

from inputs import hash_matrix, hash_positive  # noqa: E402

from sgl.operators.graph_op import LaplacianGraphOp, PprGraphOp  # noqa: E402
from sgl.operators.message_op import (  # noqa: E402
    ConcatMessageOp, IterateLearnableWeightedMessageOp, LastMessageOp, LearnableWeightedMessageOp,
    MaxMessageOp, MeanMessageOp, MinMessageOp, OverSmoothDistanceWeightedOp, ProjectedConcatMessageOp,
    SimpleWeightedMessageOp, SumMessageOp)


# ------------------------------------------------------------------------------------------
# graphs
# ------------------------------------------------------------------------------------------
def g_sym_binary(n, p, seed):
    rng = np.random.default_rng(seed)
    m = rng.random((n, n)) < p
    m = np.triu(m, 1)
    m = m | m.T
    return sp.csr_matrix(m.astype(np.float32))


def g_dir_weighted(n, m, seed):
    """directed, weighted, with duplicate (r,c) pairs (summed by csr_matrix, as data/base_data.py:29
    does), a few self loops, and node n-1 isolated"""
    rng = np.random.default_rng(seed)
    r = rng.integers(0, n - 1, m)
    c = rng.integers(0, n - 1, m)
    w = rng.uniform(0.25, 3.0, m).astype(np.float32)
    r[:4], c[:4] = [1, 5, 7, 7], [1, 5, 9, 9]          # self loops + an explicit duplicate pair
    return sp.csr_matrix((w, (r, c)), shape=(n, n), dtype=np.float32)


def g_power_law(n, avg_deg, max_deg, seed, weight=1.0):
    rng = np.random.default_rng(seed)
    w = np.clip(rng.lognormal(0.0, 1.2, n), 0.05, None)
    w = np.minimum(w / w.sum() * n * avg_deg, max_deg)
    p = w / w.sum()
    m = int(n * avg_deg / 2)
    a = rng.choice(n, m, p=p)
    b = rng.choice(n, m, p=p)
    keep = a != b
    a, b = a[keep], b[keep]
    adj = sp.coo_matrix((np.ones(len(a), np.float32), (a, b)), shape=(n, n)).tocsr()
    adj = adj + adj.T
    adj.data[:] = weight
    adj = adj.tocsr().astype(np.float32)
    adj.sort_indices()
    return adj


GRAPHS = {
    "sym64": g_sym_binary(64, 0.08, 1),
    "dir40": g_dir_weighted(40, 160, 2),
    "pl2000": g_power_law(2000, 7.0, 300, 3),
    "pl256": g_power_law(256, 8.0, 80, 4),
    "pl256w2": g_power_law(256, 8.0, 80, 4, weight=2.0),
}


def save_graphs():
    out = {}
    for name, g in GRAPHS.items():
        assert g.has_canonical_format
        out[name + "|indptr"] = g.indptr.astype(np.int32)
        out[name + "|indices"] = g.indices.astype(np.int32)
        out[name + "|data"] = g.data.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "graphs.npz"), **out)


# ------------------------------------------------------------------------------------------
# G1: normalisation
# ------------------------------------------------------------------------------------------
def gen_g1():
    out = {}
    for name in ("sym64", "dir40", "pl2000"):
        g = GRAPHS[name]
        variants = [("lap", r, None) for r in (0.0, 0.3, 0.5, 1.0)]
        variants += [("ppr", 0.5, a) for a in (0.1, 0.15, 0.2, 0.3)] + [("ppr", 0.3, 0.15)]
        struct = None
        for kind, r, a in variants:
            op = LaplacianGraphOp(1, r=r) if kind == "lap" else PprGraphOp(1, r=r, alpha=a)
            adj = op._construct_adj(g)
            assert sp.isspmatrix_csr(adj) and adj.dtype == np.float64
            adj.sort_indices()   # .tocsr() from CSC already yields sorted rows; make it explicit
            key = f"{name}|{kind}|{r}" + ("" if a is None else f"|{a}")
            if struct is None:
                struct = (adj.indptr.copy(), adj.indices.copy())
                out[name + "|indptr"] = adj.indptr.astype(np.int32)
                out[name + "|indices"] = adj.indices.astype(np.int32)
            else:
                assert np.array_equal(struct[0], adj.indptr) and np.array_equal(struct[1], adj.indices), key
            out[key] = adj.data.astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "g1_norm.npz"), **out)


# ------------------------------------------------------------------------------------------
# G2: propagate (reference ctypes -> libmatmul.so path, Linux branch base_op.py:31-32)
# ------------------------------------------------------------------------------------------
G2_CONFIGS = [
    # key, graph, op kind, r, alpha, d, K, order, keep ("all" hops or "last")
    ("d1_k10", "pl256", "lap", 0.5, None, 1, 10, "C", "all"),
    ("d3_k5", "pl256", "lap", 0.5, None, 3, 5, "C", "all"),
    ("d7_k3", "pl256", "lap", 0.3, None, 7, 3, "C", "all"),
    ("d16_k10", "pl256", "lap", 0.5, None, 16, 10, "C", "all"),
    ("d16_k3_F", "pl256", "lap", 0.5, None, 16, 3, "F", "all"),
    ("d16_k3_w2", "pl256w2", "lap", 0.5, None, 16, 3, "C", "all"),
    ("d16_k3_ppr", "pl256", "ppr", 0.5, 0.15, 16, 3, "C", "all"),
    ("d47_k2_dir", "dir40", "ppr", 0.5, 0.3, 47, 2, "C", "all"),
    ("d100_k3", "pl256", "lap", 0.5, None, 100, 3, "C", "last"),
    ("d128_k3", "pl256", "lap", 0.5, None, 128, 3, "C", "last"),
    ("d147_k5", "pl256", "lap", 0.5, None, 147, 5, "C", "last"),
    ("d500_k3", "sym64", "lap", 0.5, None, 500, 3, "C", "last"),
    ("d32_k3_pl2000", "pl2000", "lap", 0.5, None, 32, 3, "C", "last"),
]


def gen_g2():
    out = {}
    meta = {}
    for key, gname, kind, r, a, d, K, order, keep in G2_CONFIGS:
        g = GRAPHS[gname]
        x = hash_matrix(g.shape[0], d, seed=len(key) + d, order=order)
        op = LaplacianGraphOp(K, r=r) if kind == "lap" else PprGraphOp(K, r=r, alpha=a)
        feats = op.propagate(g, x)
        assert len(feats) == K + 1 and all(f.dtype == torch.float32 for f in feats)
        if keep == "all":
            for h in range(1, K + 1):
                out[f"{key}|h{h}"] = feats[h].numpy().copy()
        else:
            out[f"{key}|h{K}"] = feats[K].numpy().copy()
        # cheap pin for the hops that are not stored: fp64 sum of every hop
        out[f"{key}|sums"] = np.array([f.numpy().astype(np.float64).sum() for f in feats])
        meta[key] = dict(graph=gname, kind=kind, r=r, alpha=a, d=d, K=K, order=order, keep=keep,
                         seed=len(key) + d)
    np.savez_compressed(os.path.join(HERE, "g2_prop.npz"), **out)
    with open(os.path.join(HERE, "g2_prop.json"), "w") as f:
        json.dump(meta, f, indent=1)


# ------------------------------------------------------------------------------------------
# G3: aggregators
# ------------------------------------------------------------------------------------------
AGG_N, AGG_D, AGG_K = 96, 12, 4


def agg_feats(requires_grad=False):
    feats = [torch.from_numpy(hash_matrix(AGG_N, AGG_D, seed=100 + h).copy()) for h in range(AGG_K + 1)]
    # make hop h look a bit "smoother" than hop 0 so NAFS weights are not degenerate
    feats = [(feats[0] * (1.0 - 0.15 * h) + feats[h] * (0.15 * h)).contiguous() for h in range(AGG_K + 1)]
    if requires_grad:
        feats = [f.clone().requires_grad_(True) for f in feats]
    return feats


def state_arrays(module):
    return {k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def gen_g3():
    out = {}
    H = AGG_K + 1
    feats = agg_feats()
    for j, f in enumerate(feats):
        out[f"feat{j}"] = f.numpy().copy()     # stored: they are derived (blend) values
    out["last"] = LastMessageOp().aggregate(feats).numpy().copy()
    for (s, e) in ((0, H), (1, H - 1)):
        tag = f"{s}_{e}"
        out[f"concat|{tag}"] = ConcatMessageOp(s, e).aggregate(feats).numpy().copy()
        out[f"mean|{tag}"] = MeanMessageOp(s, e).aggregate(feats).numpy().copy()
        out[f"sum|{tag}"] = SumMessageOp(s, e).aggregate(feats).numpy().copy()
        out[f"max|{tag}"] = MaxMessageOp(s, e).aggregate(feats).numpy().copy()
        out[f"min|{tag}"] = MinMessageOp(s, e).aggregate(feats).numpy().copy()
    for (s, e) in ((0, H), (1, H)):
        out[f"simple_weighted|alpha0.85|{s}_{e}"] = SimpleWeightedMessageOp(s, e, "alpha", 0.85).aggregate(feats).numpy().copy()
    hc = [0.5, 0.2, 0.15, 0.1, 0.05]
    out["simple_weighted|hand_crafted|0_5"] = SimpleWeightedMessageOp(0, H, "hand_crafted", hc).aggregate(feats).numpy().copy()
    out["simple_weighted|hand_crafted|w"] = np.asarray(hc, dtype=np.float32)
    out["over_smooth"] = OverSmoothDistanceWeightedOp().aggregate(feats).numpy().copy()

    gout = torch.from_numpy(hash_matrix(AGG_N, AGG_D, seed=777).copy())
    for kind, args in (("simple", (AGG_K,)), ("simple_allow_neg", (AGG_K,)), ("gate", (AGG_D,)),
                       ("ori_ref", (AGG_D,)), ("jk", (AGG_K, AGG_D))):
        for (s, e) in ((0, H), (1, H)):
            torch.manual_seed(1234 + len(kind) + s)
            op = LearnableWeightedMessageOp(s, e, kind, *args)
            fg = agg_feats(requires_grad=True)
            y = op.aggregate(fg)
            (y * gout).sum().backward()
            tag = f"learnable|{kind}|{s}_{e}"
            out[tag + "|out"] = y.detach().numpy().copy()
            for k, v in state_arrays(op).items():
                out[tag + "|param|" + k] = v
            for k, p in op.named_parameters():
                out[tag + "|grad|" + k] = p.grad.numpy().copy()
            for j, f in enumerate(fg):
                out[tag + f"|dfeat{j}"] = (f.grad if f.grad is not None else torch.zeros_like(f)).numpy().copy()

    torch.manual_seed(99)
    op = IterateLearnableWeightedMessageOp(0, H, "recursive", AGG_D)
    fg = agg_feats(requires_grad=True)
    y = op.aggregate(fg)
    (y * gout).sum().backward()
    out["iterate|0_5|out"] = y.detach().numpy().copy()
    for k, v in state_arrays(op).items():
        out["iterate|0_5|param|" + k] = v
    for k, p in op.named_parameters():
        out["iterate|0_5|grad|" + k] = p.grad.numpy().copy()
    for j, f in enumerate(fg):
        out[f"iterate|0_5|dfeat{j}"] = f.grad.numpy().copy()

    # proj_concat needs sgl.models.simple_models -> imported lazily below (after stubbing)
    return out


# ------------------------------------------------------------------------------------------
# model import recipe
# ------------------------------------------------------------------------------------------
def import_models():
    for m in ["torch_geometric", "torch_geometric.data", "torch_geometric.datasets", "torch_geometric.io",
              "torch_geometric.utils", "torch_sparse", "ogb", "ogb.nodeproppred", "munkres", "gensim",
              "gensim.models", "openbox"]:
        sys.modules.setdefault(m, MagicMock())
    import sgl.dataset  # noqa: F401  (must come first: circular import)
    import sgl.models.base_model  # noqa: F401
    pkg = types.ModuleType("sgl.models.homo")
    pkg.__path__ = [REF + "/sgl/models/homo"]
    sys.modules["sgl.models.homo"] = pkg

    def load(n):
        spec = importlib.util.spec_from_file_location("sgl.models.homo." + n, f"{REF}/sgl/models/homo/{n}.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    return {n: load(n) for n in ("sgc", "ssgc", "sign", "gbp", "gamlp", "gamlp_recursive", "nafs", "pasca_v1", "pasca_v2",
                                 "pasca_v3")}


def gen_g3_proj(out):
    torch.manual_seed(5)
    H = AGG_K + 1
    op = ProjectedConcatMessageOp(0, H, AGG_D, 8, 2)
    op.eval()
    feats = agg_feats()
    with torch.no_grad():
        y = op.aggregate(feats)
    out["proj_concat|0_5|out"] = y.numpy().copy()
    for k, v in state_arrays(op).items():
        out["proj_concat|0_5|param|" + k] = v


def gen_g4(mods):
    out = {}
    g = GRAPHS["pl2000"]
    n, d, C, K = g.shape[0], 16, 5, 3
    x = hash_matrix(n, d, seed=4242)
    idx = np.arange(0, n, 10)
    out["idx"] = idx
    specs = {
        "SGC": (mods["sgc"].SGC, (K, d, C)),
        "SSGC": (mods["ssgc"].SSGC, (K, d, C)),
        "SIGN": (mods["sign"].SIGN, (K, d, C, 32, 2)),
        "GBP": (mods["gbp"].GBP, (K, d, C, 32, 2)),
        "GAMLP": (mods["gamlp"].GAMLP, (K, d, C, 32, 2)),
        "GAMLPRecursive": (mods["gamlp_recursive"].GAMLPRecursive, (K, d, C, 32, 2)),
        "NAFS": (mods["nafs"].NAFS, (K, d, C)),
        "PASCA_V1": (mods["pasca_v1"].PASCA_V1, (K, d, C, 32, 3)),
        "PASCA_V2": (mods["pasca_v2"].PASCA_V2, (K, d, C, 32, 3)),
        "PASCA_V3": (mods["pasca_v3"].PASCA_V3, (K, 2, d, C, 32, 3)),
    }
    for name, (cls, args) in specs.items():
        torch.manual_seed(7)
        model = cls(*args)
        model.eval()
        model.preprocess(g, x)
        with torch.no_grad():
            y = model.model_forward(idx, torch.device("cpu"))
        out[f"{name}|out"] = y.numpy().copy()
        for k, v in state_arrays(model).items():
            out[f"{name}|param|{k}"] = v
        if name == "PASCA_V3":
            with torch.no_grad():
                full = model.model_forward(range(n), torch.device("cpu"))
                post = model.postprocess(g, full)
            out[f"{name}|post"] = post.numpy().copy()[idx]
    np.savez_compressed(os.path.join(HERE, "g4_models.npz"), **out)


# ------------------------------------------------------------------------------------------
# G5: error contract
# ------------------------------------------------------------------------------------------
def gen_g5():
    g = GRAPHS["sym64"]
    x = hash_matrix(64, 4, seed=1)
    cases = {}

    def record(name, fn):
        try:
            r = fn()
            cases[name] = {"raised": None, "returned_type": type(r).__name__,
                           "returned_msg": str(r) if isinstance(r, Exception) else None}
        except Exception as e:  # noqa: BLE001
            cases[name] = {"raised": type(e).__name__, "msg": str(e)}

    record("propagate_tensor_feature", lambda: LaplacianGraphOp(2).propagate(g, torch.from_numpy(x)))
    record("propagate_coo_adj", lambda: LaplacianGraphOp(2).propagate(g.tocoo(), x))
    record("propagate_dense_adj", lambda: LaplacianGraphOp(2).propagate(g.toarray(), x))
    record("propagate_shape_mismatch", lambda: LaplacianGraphOp(2).propagate(g, x[:10]))
    record("ppr_dense_adj", lambda: PprGraphOp(2).propagate(np.zeros((3, 3)), x))
    record("aggregate_not_list", lambda: MeanMessageOp(0, 2).aggregate(torch.zeros(2, 2)))
    record("aggregate_not_tensor", lambda: MeanMessageOp(0, 2).aggregate([np.zeros((2, 2)), np.zeros((2, 2))]))
    record("simple_weighted_bad_type", lambda: SimpleWeightedMessageOp(0, 2, "nope", 0.5))
    record("simple_weighted_alpha_int", lambda: SimpleWeightedMessageOp(0, 2, "alpha", 1))
    record("simple_weighted_alpha_range", lambda: SimpleWeightedMessageOp(0, 2, "alpha", 1.5))
    record("simple_weighted_nargs", lambda: SimpleWeightedMessageOp(0, 2, "alpha"))
    record("simple_weighted_hand_bad", lambda: SimpleWeightedMessageOp(0, 2, "hand_crafted", 3))
    record("learnable_bad_type", lambda: LearnableWeightedMessageOp(0, 2, "nope", 1))
    record("learnable_simple_nargs", lambda: LearnableWeightedMessageOp(0, 2, "simple"))
    record("learnable_jk_nargs", lambda: LearnableWeightedMessageOp(0, 2, "jk", 3))
    record("iterate_bad_type", lambda: IterateLearnableWeightedMessageOp(0, 2, "nope", 4))
    record("iterate_nargs", lambda: IterateLearnableWeightedMessageOp(0, 2, "recursive"))
    with open(os.path.join(HERE, "g5_errors.json"), "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)


# ------------------------------------------------------------------------------------------
# G6: SURVEY 8(f) rank 2 consumers of the same SpMM: label propagation / Correct&Smooth (sgl/tricks) and the
#     NAFS multi-r feature-smoothing pipeline of the NAFS tasks (tasks/node_clustering.py:205-258)
# ------------------------------------------------------------------------------------------
def gen_g6():
    from sgl.tricks.utils import adj_to_symmetric_norm as tricks_norm, label_propagation
    from sgl.tricks.correct_and_smooth import CorrectAndSmooth
    out = {}
    g = GRAPHS["pl2000"]
    n, C = g.shape[0], 5
    adj = tricks_norm(g, 0.5)
    lab = torch.from_numpy((hash_matrix(n, 1, seed=31)[:, 0] * 1000).astype(np.int64) % C)
    lab[:C] = torch.arange(C)                       # every class present (F.one_hot infers C from max)
    mask = np.arange(0, n, 3)
    out["lp|labels"] = lab.numpy()
    out["lp|mask"] = mask
    out["lp|long_masked"] = label_propagation(lab, adj, 5, 0.8, mask=torch.from_numpy(mask)).numpy().copy()
    out["lp|long_nomask"] = label_propagation(lab, adj, 3, 0.5).numpy().copy()
    soft = torch.softmax(torch.from_numpy(hash_matrix(n, C, seed=32)) * 3, 1)
    out["lp|float_clamp11"] = label_propagation(soft - 0.3, adj, 4, 0.9,
                                                post_process=lambda x: x.clamp_(-1., 1.)).numpy().copy()
    for autoscale in (True, False):
        cs = CorrectAndSmooth(4, 0.9, 3, 0.7, autoscale=autoscale, scale=1.5)
        y1 = cs.correct(soft.clone(), lab, mask, adj)
        y2 = cs.smooth(y1.clone(), lab, mask, adj)
        out[f"cs|autoscale{int(autoscale)}|correct"] = y1.numpy().copy()
        out[f"cs|autoscale{int(autoscale)}|smooth"] = y2.numpy().copy()

    # NAFS task pipeline: run the reference's own _k_hop_cluster and capture what it hands to KMeans
    for m in ["matplotlib", "matplotlib.pyplot", "munkres"]:
        try:
            importlib.import_module(m)
        except ImportError:
            sys.modules[m] = MagicMock()
    pkg = types.ModuleType("sgl.tasks")
    pkg.__path__ = [REF + "/sgl/tasks"]
    sys.modules["sgl.tasks"] = pkg
    import sgl.tasks.node_clustering as nc

    class Captured(Exception):
        pass

    class FakeKMeans:
        def __init__(self, *a, **k):
            pass

        def fit_predict(self, x):
            FakeKMeans.seen = np.array(x, copy=True)
            raise Captured()

    nc.KMeans = FakeKMeans
    g8 = GRAPHS["pl256"]
    x8 = hash_positive(256, 8, seed=33)

    class DS:
        x = x8
        adj = g8
        num_node = 256

    for method in ("mean", "max", "concat", "simple"):
        task = object.__new__(nc.NodeClusteringNAFS)
        task._NodeClusteringNAFS__dataset = DS
        task._NodeClusteringNAFS__r_list = [0.5, 0.4, 0.3, 0.2, 0.1, 0]
        task._NodeClusteringNAFS__method = method
        task._NodeClusteringNAFS__n_clusters = 3
        task._NodeClusteringNAFS__n_init = 1
        task._NodeClusteringNAFS__seed = 0
        try:
            task._k_hop_cluster(3)
        except Captured:
            pass
        out[f"nafs_task|{method}|hops3"] = FakeKMeans.seen.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g6_consumers.npz"), **out)


# ------------------------------------------------------------------------------------------
# G11: the NAFS task's hop SWEEP (tasks/node_clustering.py:139,176-178: _k_hop_cluster(hop) for every hop of `hops`): what the
#      reference hands to KMeans for hops in {0, 1, 3, 6}, every ensemble method -- the pin of sgl_nafs_prefix_f32 / nafs_ensemble_sweep
# ------------------------------------------------------------------------------------------
def gen_g11():
    for m in ["matplotlib", "matplotlib.pyplot", "munkres"]:
        try:
            importlib.import_module(m)
        except ImportError:
            sys.modules[m] = MagicMock()
    if "sgl.tasks" not in sys.modules:
        pkg = types.ModuleType("sgl.tasks")
        pkg.__path__ = [REF + "/sgl/tasks"]
        sys.modules["sgl.tasks"] = pkg
    import sgl.tasks.node_clustering as nc

    class Captured(Exception):
        pass

    class FakeKMeans:
        def __init__(self, *a, **k):
            pass

        def fit_predict(self, x):
            FakeKMeans.seen = np.array(x, copy=True)
            raise Captured()

    nc.KMeans = FakeKMeans
    g8 = GRAPHS["pl256"]
    x8 = hash_positive(256, 12, seed=41)
    x8[5] = 0.0                                      # an all-zero feature row: cosine 0 for every hop (the + 1e-10 quirk)

    class DS:
        x = x8
        adj = g8
        num_node = 256

    out = {"x": x8, "hops": np.array([0, 1, 3, 6]), "r_list": np.array([0.5, 0.3, 0.0])}
    for method in ("mean", "max", "concat", "simple"):
        for hops in (0, 1, 3, 6):
            task = object.__new__(nc.NodeClusteringNAFS)
            task._NodeClusteringNAFS__dataset = DS
            task._NodeClusteringNAFS__r_list = [0.5, 0.3, 0]
            task._NodeClusteringNAFS__method = method
            task._NodeClusteringNAFS__n_clusters = 3
            task._NodeClusteringNAFS__n_init = 1
            task._NodeClusteringNAFS__seed = 0
            try:
                task._k_hop_cluster(hops)
            except Captured:
                pass
            out[f"nafs_sweep|{method}|hops{hops}"] = FakeKMeans.seen.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g11_nafs_sweep.npz"), **out)


# ------------------------------------------------------------------------------------------
# G7: ingest -- the reference's Edge (sgl/data/base_data.py:8-30) turning COO arrays into the CSR adjacency
# ------------------------------------------------------------------------------------------
def gen_g7():
    import sgl.dataset  # noqa: F401  (circular import: must precede sgl.data)
    from sgl.data.base_data import Edge
    rng = np.random.default_rng(11)
    n, m = 500, 6000
    row = rng.integers(0, n, m)
    col = rng.integers(0, n, m)
    row[:300], col[:300] = row[300:600], col[300:600]          # 300 duplicated pairs
    row[600:620], col[600:620] = 7, 9                            # one pair repeated 20 times
    w = rng.uniform(0.1, 2.0, m).astype(np.float32)
    e = Edge(row, col, w, "n__to__n", n)
    a = e.sparse_matrix
    a.sum_duplicates()
    a.sort_indices()
    np.savez_compressed(os.path.join(HERE, "g7_ingest.npz"), row=row, col=col, data=w, n=n,
                        indptr=a.indptr.astype(np.int64), indices=a.indices.astype(np.int32), values=a.data.astype(np.float32))


def gen_g8():
    """hop-range quirks of the stateless aggregators, straight and on top of a real propagation (the fused
    GraphOp.propagate_reduce path of sgl_amd must reproduce aggregate(propagate(...)) of the reference):
    Mean divides by (end - start) whatever the slice held (mean_message_op.py:10), partial ranges, a single hop."""
    out = {}
    feats = agg_feats()
    H = AGG_K + 1
    for (s_, e_) in ((0, 10), (4, 5), (2, 9)):
        out[f"mean|{s_}_{e_}"] = MeanMessageOp(s_, e_).aggregate(feats).numpy().copy()
    for (s_, e_) in ((2, 3), (3, 10)):
        out[f"sum|{s_}_{e_}"] = SumMessageOp(s_, e_).aggregate(feats).numpy().copy()
    g = GRAPHS["pl256"]
    x = hash_matrix(256, 20, seed=808)
    K = 4
    hops = LaplacianGraphOp(K, r=0.5).propagate(g, x)
    hops_ppr = PprGraphOp(K, r=0.3, alpha=0.2).propagate(g, x)
    for name, hp in (("lap", hops), ("ppr", hops_ppr)):
        out[f"prop|{name}|last"] = LastMessageOp().aggregate(hp).numpy().copy()
        out[f"prop|{name}|sum|0_{K + 1}"] = SumMessageOp(0, K + 1).aggregate(hp).numpy().copy()
        out[f"prop|{name}|sum|1_3"] = SumMessageOp(1, 3).aggregate(hp).numpy().copy()
        out[f"prop|{name}|mean|0_{K + 1}"] = MeanMessageOp(0, K + 1).aggregate(hp).numpy().copy()
        out[f"prop|{name}|mean|0_10"] = MeanMessageOp(0, 10).aggregate(hp).numpy().copy()
        out[f"prop|{name}|mean|2_4"] = MeanMessageOp(2, 4).aggregate(hp).numpy().copy()
        out[f"prop|{name}|simple_weighted|alpha0.85|0_{K + 1}"] = \
            SimpleWeightedMessageOp(0, K + 1, "alpha", 0.85).aggregate(hp).numpy().copy()
        out[f"prop|{name}|simple_weighted|alpha0.3|1_{K + 1}"] = \
            SimpleWeightedMessageOp(1, K + 1, "alpha", 0.3).aggregate(hp).numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g8_ranges.npz"), **out)


# ------------------------------------------------------------------------------------------
# G9: BASELINE config 5 (PaSca op sweep: Laplacian + PPR, k = 10, all MessageOps -- search/search_models.py:19-46) at ITS hop
#     count: k = 10 propagation of both graph ops, and every MessageOp over H = 11 hop matrices (jk's parameter shape depends on
#     K: learnable_weighted_messahe_op.py:56-57), d = 16 and d = 128
# ------------------------------------------------------------------------------------------
G9_PROP = [
    # key, graph, kind, r, alpha, d, K, hops stored
    ("ppr0.1_k10_pl256", "pl256", "ppr", 0.5, 0.1, 16, 10, (1, 2, 5, 10)),
    ("ppr0.2_k10_pl256", "pl256", "ppr", 0.5, 0.2, 16, 10, (1, 2, 5, 10)),
    ("ppr0.3_k10_pl256", "pl256", "ppr", 0.5, 0.3, 16, 10, (1, 2, 5, 10)),
    ("ppr0.15_k1_pl256", "pl256", "ppr", 0.5, 0.15, 16, 1, (1,)),
    ("ppr0.2_k10_d128_pl256", "pl256", "ppr", 0.5, 0.2, 128, 10, (10,)),
    ("ppr0.1_k10_pl2000", "pl2000", "ppr", 0.5, 0.1, 8, 10, (10,)),
    ("ppr0.2_k10_pl2000", "pl2000", "ppr", 0.5, 0.2, 8, 10, (10,)),
    ("ppr0.3_k10_pl2000", "pl2000", "ppr", 0.5, 0.3, 8, 10, (10,)),
    ("lap_k10_pl2000", "pl2000", "lap", 0.5, None, 8, 10, (10,)),
    ("ppr0.2_k10_dir40", "dir40", "ppr", 0.3, 0.2, 5, 10, (1, 10)),
]
G9_AGG = {16: 64, 128: 16}          # feat_dim -> rows
G9_K = 10
G9_DFEAT = (0, 5, 10)               # input gradients stored in full for these hops; fp64 sums for all


def g9_feats(d, requires_grad=False):
    n = G9_AGG[d]
    feats = [torch.from_numpy(hash_matrix(n, d, seed=300 + h).copy()) for h in range(G9_K + 1)]
    feats = [(feats[0] * (1.0 - 0.08 * h) + feats[h] * (0.08 * h)).contiguous() for h in range(G9_K + 1)]
    if requires_grad:
        feats = [f.clone().requires_grad_(True) for f in feats]
    return feats


def gen_g9():
    out, meta = {}, {"prop": {}, "agg": {"dims": {str(d): n for d, n in G9_AGG.items()}, "K": G9_K, "dfeat_stored": list(G9_DFEAT)}}
    for key, gname, kind, r, a, d, K, keep in G9_PROP:
        g = GRAPHS[gname]
        seed = 900 + len(key) + d
        x = hash_matrix(g.shape[0], d, seed=seed)
        op = LaplacianGraphOp(K, r=r) if kind == "lap" else PprGraphOp(K, r=r, alpha=a)
        feats = op.propagate(g, x)
        assert len(feats) == K + 1
        for h in keep:
            out[f"prop|{key}|h{h}"] = feats[h].numpy().copy()
        out[f"prop|{key}|sums"] = np.array([f.numpy().astype(np.float64).sum() for f in feats])
        meta["prop"][key] = dict(graph=gname, kind=kind, r=r, alpha=a, d=d, K=K, keep=list(keep), seed=seed)
    H = G9_K + 1
    import_models()                                    # ProjectedConcatMessageOp needs sgl.models.simple_models
    for d, n in G9_AGG.items():
        P = f"agg|d{d}|"
        feats = g9_feats(d)
        for j, f in enumerate(feats):
            out[P + f"feat{j}"] = f.numpy().copy()
        out[P + "last"] = LastMessageOp().aggregate(feats).numpy().copy()
        for (s, e) in ((0, H), (1, H - 1)):
            tag = f"{s}_{e}"
            out[P + f"concat|{tag}"] = ConcatMessageOp(s, e).aggregate(feats).numpy().copy()
            out[P + f"mean|{tag}"] = MeanMessageOp(s, e).aggregate(feats).numpy().copy()
            out[P + f"sum|{tag}"] = SumMessageOp(s, e).aggregate(feats).numpy().copy()
            out[P + f"max|{tag}"] = MaxMessageOp(s, e).aggregate(feats).numpy().copy()
            out[P + f"min|{tag}"] = MinMessageOp(s, e).aggregate(feats).numpy().copy()
        for (s, e) in ((0, H), (1, H)):
            out[P + f"simple_weighted|alpha0.85|{s}_{e}"] = SimpleWeightedMessageOp(s, e, "alpha", 0.85).aggregate(feats).numpy().copy()
        out[P + "over_smooth"] = OverSmoothDistanceWeightedOp().aggregate(feats).numpy().copy()
        gout = torch.from_numpy(hash_matrix(n, d, seed=778).copy())

        def record(tag, op):
            fg = g9_feats(d, requires_grad=True)
            y = op.aggregate(fg)
            (y * gout).sum().backward()
            out[tag + "|out"] = y.detach().numpy().copy()
            for k, v in state_arrays(op).items():
                out[tag + "|param|" + k] = v
            for k, p_ in op.named_parameters():
                out[tag + "|grad|" + k] = p_.grad.numpy().copy()
            grads = [(f.grad if f.grad is not None else torch.zeros_like(f)).numpy() for f in fg]
            for j in G9_DFEAT:
                out[tag + f"|dfeat{j}"] = grads[j].copy()
            out[tag + "|dfeat_sums"] = np.array([gr.astype(np.float64).sum() for gr in grads])
            out[tag + "|dfeat_abs_sums"] = np.array([np.abs(gr.astype(np.float64)).sum() for gr in grads])

        for kind, args in (("simple", (G9_K,)), ("simple_allow_neg", (G9_K,)), ("gate", (d,)), ("ori_ref", (d,)), ("jk", (G9_K, d))):
            for (s, e) in ((0, H), (1, H)):
                torch.manual_seed(4321 + len(kind) + s + d)
                record(P + f"learnable|{kind}|{s}_{e}", LearnableWeightedMessageOp(s, e, kind, *args))
        torch.manual_seed(98 + d)
        record(P + f"iterate|0_{H}", IterateLearnableWeightedMessageOp(0, H, "recursive", d))
        torch.manual_seed(6 + d)
        pc = ProjectedConcatMessageOp(0, H, d, 8, 2)
        pc.eval()
        with torch.no_grad():
            y = pc.aggregate(g9_feats(d))
        out[P + f"proj_concat|0_{H}|out"] = y.numpy().copy()
        for k, v in state_arrays(pc).items():
            out[P + f"proj_concat|0_{H}|param|" + k] = v
    np.savez_compressed(os.path.join(HERE, "g9_config5.npz"), **out)
    with open(os.path.join(HERE, "g9_config5.json"), "w") as f:
        json.dump(meta, f, indent=1)


# ------------------------------------------------------------------------------------------
# G10: BASELINE config 3 -- the label use / label reuse loop around model.preprocess (GAMLP, d + C = 147 columns, K = 5), run
#      through the REFERENCE'S OWN task code (tasks/node_classification_with_label_use.py:58-137) with training and evaluation
#      stubbed out (they are dense-head work, outside SURVEY section 8) and ONE adaptation without which the reference loop cannot
#      run on Linux at all: its add_labels (tasks/utils.py:33-36) returns float64 and csr_sparse_dense_matmul's ctypes signature
#      accepts float32 only (operators/utils.py:16-26) -> ArgumentError on the first preprocess.  The adaptation casts to float32.
# ------------------------------------------------------------------------------------------
def _run_label_reuse(fp64):
    """the reference's label use / reuse task loop; fp64=True: the same loop in float64 (features as add_labels returns them, the model in
    .double(), propagate through _construct_adj + scipy's dot instead of the float32-only C kernel) -- the truth for G12"""
    mods = import_models()
    pkg = types.ModuleType("sgl.tasks")
    pkg.__path__ = [REF + "/sgl/tasks"]
    sys.modules.setdefault("sgl.tasks", pkg)
    for m in ["matplotlib", "matplotlib.pyplot", "munkres"]:
        try:
            importlib.import_module(m)
        except ImportError:
            sys.modules[m] = MagicMock()
    import sgl.tasks.node_classification_with_label_use as lu
    orig_add = lu.add_labels
    lu.add_labels = (lambda f, y, idx, C: orig_add(f, y, idx, C)) if fp64 else (lambda f, y, idx, C: orig_add(f, y, idx, C).astype(np.float32))
    lu.train = lambda *a, **k: (0.0, 0.0)
    lu.evaluate = lambda *a, **k: (0.0, 0.0)
    lu.accuracy = lambda *a, **k: 0.0

    g = GRAPHS["pl2000"]
    n, d, C, K = g.shape[0], 100, 47, 5
    rng = np.random.default_rng(77)
    perm = rng.permutation(n)
    labels = torch.from_numpy((rng.integers(0, C, n)).astype(np.int64))

    class Data:
        num_node = n

    class DS:
        x = hash_matrix(n, d, seed=1010)
        y = labels
        adj = g
        num_node = n
        num_classes = C
        data = Data
        train_idx = perm[: int(0.3 * n)].tolist()
        val_idx = perm[int(0.3 * n): int(0.5 * n)].tolist()
        test_idx = perm[int(0.5 * n):].tolist()

    torch.manual_seed(11)
    model = mods["gamlp"].GAMLP(K, d + C, C, 64, 2)
    model.eval()                      # the state label reuse runs in: the previous epoch's evaluate() left the model in eval mode
    if fp64:
        model = model.double()
        gop = model._pre_graph_op

        def propagate64(adj, feature):
            a64 = gop._construct_adj(adj)
            hops = [np.asarray(feature, dtype=np.float64)]
            for _ in range(gop._prop_steps):
                hops.append(a64.dot(hops[-1]))
            return [torch.from_numpy(h) for h in hops]
        gop.propagate = propagate64
    out = {"n": n, "d": d, "C": C, "K": K, "train_idx": np.asarray(DS.train_idx), "val_idx": np.asarray(DS.val_idx),
           "test_idx": np.asarray(DS.test_idx), "labels": labels.numpy()}
    for k, v in state_arrays(model).items():
        out["param|" + k] = v
    sub = np.arange(0, n, 20)
    out["sub_rows"] = sub
    calls = []
    real_pre = model.preprocess

    def spy(adj, features):
        assert features.dtype == (np.float64 if fp64 else np.float32) and features.shape == (n, d + C)
        i = len(calls)
        out[f"call{i}|label_cols_sub"] = features[sub, d:].copy()
        out[f"call{i}|feature_colsum"] = features.astype(np.float64).sum(0)
        real_pre(adj, features)
        out[f"call{i}|hop_sums"] = np.array([f.numpy().astype(np.float64).sum() for f in model._processed_feat_list])
        calls.append(i)
    model.preprocess = spy
    masks = []
    real_add = lu.add_labels

    def spy_add(f, y, idx, C_):
        masks.append(np.asarray(idx).copy())
        return real_add(f, y, idx, C_)
    lu.add_labels = spy_add

    task = object.__new__(lu.NodeClassificationWithLabelUse)
    P = "_NodeClassificationWithLabelUse__"
    for k, v in dict(dataset=DS, labels=labels, model=model, optimizer=None, epochs=3, loss_fn=None, device=torch.device("cpu"),
                     seed=42, mask_rate=0.5, use_labels=True, reuse_start_epoch=0, label_iters=2, label_reuse_batch_size=700,
                     mini_batch=False).items():
        setattr(task, P + k, v)
    task._execute()
    assert len(calls) == 3 + 2 * 2 and len(masks) == 3
    for e, mk in enumerate(masks):
        out[f"epoch{e}|train_labels_idx"] = mk
    hops = model._processed_feat_list
    for h in (1, 3, 5):
        out[f"final|hop{h}_sub"] = hops[h].numpy()[sub].copy()
    with torch.no_grad():
        logits = model.model_forward(range(n), torch.device("cpu"))
    out["final|logits_sub"] = logits.numpy()[sub].copy()
    out["final|logits_colsum"] = logits.numpy().astype(np.float64).sum(0)
    return out


def gen_g10():
    out = _run_label_reuse(False)
    np.savez_compressed(os.path.join(HERE, "g10_label_reuse.npz"), **out)


# ------------------------------------------------------------------------------------------
# G12: float64 TRUTH for every floating-point case whose test tolerance would otherwise be chosen by hand: the learnable MessageOps
#      of G3 / G9 (outputs, parameter gradients, input gradients) and the model logits of G4.  The reference's own modules, built
#      with the same seeds (hence the same float32 parameters), cast to .double() and fed the same inputs in float64; the hops of
#      G4 come from the reference's _construct_adj (float64) and scipy's dot.  The tests then require
#      err(HIP, truth) <= max(2 * err(reference float32, truth), floor) instead of a flat tolerance.
# ------------------------------------------------------------------------------------------
def gen_g12():
    out = {}

    def record(prefix, op, feats32, gout32, stored):
        op = op.double()
        fg = [f.detach().double().clone().requires_grad_(True) for f in feats32]
        # condition magnitudes of the Linear gradients: weight.grad[k] = sum_rows dS x[:, k] and bias.grad = sum_rows dS are sums of
        # cancelling terms; recorded next to the truth: the sums of the ABSOLUTE terms (over every call of the module)
        cond = {}
        hooks = []
        for mname, m in op.named_modules():
            if isinstance(m, torch.nn.Linear):
                def fwd(mod, inp, outp, mname=mname):
                    xin = inp[0].detach()

                    def bwd(gr):
                        cw = gr.abs().t() @ xin.abs()
                        cb = gr.abs().sum(0)
                        cond[mname + ".weight"] = cond.get(mname + ".weight", 0) + cw
                        cond[mname + ".bias"] = cond.get(mname + ".bias", 0) + cb
                    outp.register_hook(bwd)
                hooks.append(m.register_forward_hook(fwd))
        y = op.aggregate(fg)
        (y * gout32.double()).sum().backward()
        for h_ in hooks:
            h_.remove()
        for k, v in cond.items():
            out[prefix + "|gradcond|" + k] = v.numpy().copy()
        out[prefix + "|out"] = y.detach().numpy().copy()
        for k, p_ in op.named_parameters():
            out[prefix + "|grad|" + k] = p_.grad.numpy().copy()
        for j in stored:
            out[prefix + f"|dfeat{j}"] = (fg[j].grad if fg[j].grad is not None else torch.zeros_like(fg[j])).numpy().copy()

    H = AGG_K + 1
    gout = torch.from_numpy(hash_matrix(AGG_N, AGG_D, seed=777).copy())
    for kind, args in (("simple", (AGG_K,)), ("simple_allow_neg", (AGG_K,)), ("gate", (AGG_D,)), ("ori_ref", (AGG_D,)), ("jk", (AGG_K, AGG_D))):
        for (s, e) in ((0, H), (1, H)):
            torch.manual_seed(1234 + len(kind) + s)
            record(f"g3|learnable|{kind}|{s}_{e}", LearnableWeightedMessageOp(s, e, kind, *args), agg_feats(), gout, range(H))
    torch.manual_seed(99)
    record("g3|iterate|0_5", IterateLearnableWeightedMessageOp(0, H, "recursive", AGG_D), agg_feats(), gout, range(H))
    H = G9_K + 1
    for d, n in G9_AGG.items():
        P = f"g9|agg|d{d}|"
        gout = torch.from_numpy(hash_matrix(n, d, seed=778).copy())
        for kind, args in (("simple", (G9_K,)), ("simple_allow_neg", (G9_K,)), ("gate", (d,)), ("ori_ref", (d,)), ("jk", (G9_K, d))):
            for (s, e) in ((0, H), (1, H)):
                torch.manual_seed(4321 + len(kind) + s + d)
                record(P + f"learnable|{kind}|{s}_{e}", LearnableWeightedMessageOp(s, e, kind, *args), g9_feats(d), gout, G9_DFEAT)
        torch.manual_seed(98 + d)
        record(P + f"iterate|0_{H}", IterateLearnableWeightedMessageOp(0, H, "recursive", d), g9_feats(d), gout, G9_DFEAT)

    # G4 logits: same seeds as gen_g4, hops in float64 (reference _construct_adj + scipy dot), modules in double
    mods = import_models()
    g = GRAPHS["pl2000"]
    n, d, C, K = g.shape[0], 16, 5, 3
    x = hash_matrix(n, d, seed=4242).astype(np.float64)
    idx = np.arange(0, n, 10)
    specs = {
        "SGC": (mods["sgc"].SGC, (K, d, C)), "SSGC": (mods["ssgc"].SSGC, (K, d, C)), "SIGN": (mods["sign"].SIGN, (K, d, C, 32, 2)),
        "GBP": (mods["gbp"].GBP, (K, d, C, 32, 2)), "GAMLP": (mods["gamlp"].GAMLP, (K, d, C, 32, 2)),
        "GAMLPRecursive": (mods["gamlp_recursive"].GAMLPRecursive, (K, d, C, 32, 2)), "NAFS": (mods["nafs"].NAFS, (K, d, C)),
        "PASCA_V1": (mods["pasca_v1"].PASCA_V1, (K, d, C, 32, 3)), "PASCA_V2": (mods["pasca_v2"].PASCA_V2, (K, d, C, 32, 3)),
        "PASCA_V3": (mods["pasca_v3"].PASCA_V3, (K, 2, d, C, 32, 3)),
    }
    for name, (cls, args) in specs.items():
        torch.manual_seed(7)
        model = cls(*args)
        model.eval()
        adj = model._pre_graph_op._construct_adj(g)                     # float64 scipy CSR
        hops = [x]
        for _ in range(model._pre_graph_op._prop_steps):
            hops.append(adj.dot(hops[-1]))
        model = model.double()
        model._processed_feat_list = [torch.from_numpy(h) for h in hops]
        if model._pre_msg_op.aggr_type in ["proj_concat", "learnable_weighted", "iterate_learnable_weighted"]:
            model._pre_msg_learnable = True
        else:
            model._pre_msg_learnable = False
            model._processed_feature = model._pre_msg_op.aggregate(model._processed_feat_list)
        with torch.no_grad():
            y = model.model_forward(idx, torch.device("cpu"))
        assert y.dtype == torch.float64, (name, y.dtype)
        out[f"g4|{name}|out"] = y.numpy().copy()
        if name == "PASCA_V3":                                          # post-propagation of the softmaxed logits, all in float64
            with torch.no_grad():
                full = torch.softmax(model.model_forward(range(n), torch.device("cpu")), dim=1).numpy()
            padj = model._post_graph_op._construct_adj(g)
            ph = [full]
            for _ in range(model._post_graph_op._prop_steps):
                ph.append(padj.dot(ph[-1]))
            post = model._post_msg_op.aggregate([torch.from_numpy(h) for h in ph])
            assert post.dtype == torch.float64
            out[f"g4|{name}|post"] = post.numpy().copy()[idx]
    # G10's loop in float64: the label columns before every preprocess call, the final hop rows and logits
    lr = _run_label_reuse(True)
    for k, v in lr.items():
        if k.endswith("|label_cols_sub") or k.startswith("final|hop") or k == "final|logits_sub":
            out["g10|" + k] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "g12_fp64_truth.npz"), **out)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--only":       # regenerate one family without touching the others
        globals()["gen_" + sys.argv[2]]()
        return
    save_graphs()
    gen_g1()
    gen_g2()
    g3 = gen_g3()
    mods = import_models()
    gen_g3_proj(g3)
    np.savez_compressed(os.path.join(HERE, "g3_agg.npz"), **g3)
    gen_g4(mods)
    gen_g5()
    gen_g6()
    gen_g7()
    gen_g8()
    gen_g9()
    gen_g10()
    gen_g11()
    gen_g12()
    tot = 0
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith((".npz", ".json")):
            sz = os.path.getsize(os.path.join(HERE, fn))
            tot += sz
            print(f"{fn:20s} {sz / 1024:9.1f} KiB")
    print(f"total {tot / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
