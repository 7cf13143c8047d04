"""The homogeneous SGAP model zoo as ONE declarative table.

Every model of sgl/models/homo/*.py is the same three-slot composition -- a pre-propagation GraphOp, a hop
aggregator MessageOp and a dense head (plus, for PaSca V3, a post-propagation pair) -- so instead of ten
near-identical files the compositions live in `_ZOO` below and the classes are stamped out from it.  Class names,
constructor signatures and therefore state_dict layouts match the reference models they are named after
(reference file:line cited per entry), which is what makes checkpoints and call sites interchangeable."""
from ..base_model import BaseSGAPModel
from ..simple_models import IdenticalMapping, LogisticRegression, MultiLayerPerceptron, ResMultiLayerPerceptron
from ...operators.graph_op import LaplacianGraphOp, PprGraphOp
from ...operators.message_op import (ConcatMessageOp, IterateLearnableWeightedMessageOp, LastMessageOp,
                                     LearnableWeightedMessageOp, MeanMessageOp, OverSmoothDistanceWeightedOp,
                                     SimpleWeightedMessageOp)

# name -> (constructor argument names after prop_steps, builder(args) -> dict of slots, reference citation)
_ZOO = {
    "SGC": (("feat_dim", "output_dim"), lambda a: dict(
        graph=LaplacianGraphOp(a.prop_steps, r=0.5), msg=LastMessageOp(),
        head=LogisticRegression(a.feat_dim, a.output_dim)), "sgc.py:7-13 (Wu et al. 2019: A_hat^K X -> logistic regression)"),
    "SSGC": (("feat_dim", "output_dim"), lambda a: dict(
        graph=LaplacianGraphOp(a.prop_steps, r=0.5), msg=MeanMessageOp(start=0, end=a.prop_steps + 1),
        head=LogisticRegression(a.feat_dim, a.output_dim)), "ssgc.py:7-13 (mean of all hops)"),
    "SIGN": (("feat_dim", "output_dim", "hidden_dim", "num_layers"), lambda a: dict(
        graph=LaplacianGraphOp(a.prop_steps, r=0.5), msg=ConcatMessageOp(0, a.prop_steps + 1),
        head=MultiLayerPerceptron((a.prop_steps + 1) * a.feat_dim, a.hidden_dim, a.num_layers, a.output_dim)),
        "sign.py:8-15 (concatenated hops -> MLP)"),
    "GBP": (("feat_dim", "output_dim", "hidden_dim", "num_layers", ("r", 0.5), ("alpha", 0.85)), lambda a: dict(
        graph=LaplacianGraphOp(a.prop_steps, r=0.5),       # the reference ignores its own `r` argument too
        msg=SimpleWeightedMessageOp(0, a.prop_steps + 1, "alpha", a.alpha),
        head=MultiLayerPerceptron(a.feat_dim, a.hidden_dim, a.num_layers, a.output_dim)), "gbp.py:7-13 (alpha-decayed hop sum)"),
    "GAMLP": (("feat_dim", "output_dim", "hidden_dim", "num_layers"), lambda a: dict(
        graph=LaplacianGraphOp(a.prop_steps, r=0.5),
        msg=LearnableWeightedMessageOp(0, a.prop_steps + 1, "jk", a.prop_steps, a.feat_dim),
        head=MultiLayerPerceptron(a.feat_dim, a.hidden_dim, a.num_layers, a.output_dim)), "gamlp.py:7-13 (JK attention)"),
    "GAMLPRecursive": (("feat_dim", "output_dim", "hidden_dim", "num_layers"), lambda a: dict(
        graph=LaplacianGraphOp(a.prop_steps, r=0.5),
        msg=IterateLearnableWeightedMessageOp(0, a.prop_steps + 1, "recursive", a.feat_dim),
        head=MultiLayerPerceptron(a.feat_dim, a.hidden_dim, a.num_layers, a.output_dim)),
        "gamlp_recursive.py:7-13 (recursive attention)"),
    "NAFS": (("feat_dim", "output_dim"), lambda a: dict(
        graph=LaplacianGraphOp(a.prop_steps, r=0.5), msg=OverSmoothDistanceWeightedOp(), head=IdenticalMapping()),
        "nafs.py:7-13 (over-smoothing-distance weighted hops, no trainable head)"),
    "PASCA_V1": (("feat_dim", "output_dim", "hidden_dim", "num_layers"), lambda a: dict(
        graph=PprGraphOp(a.prop_steps, r=0.5, alpha=0.1),
        # the reference passes feat_dim where prop_steps is expected -> weight vector of length feat_dim + 1 (kept)
        msg=LearnableWeightedMessageOp(1, a.prop_steps + 1, "simple", a.feat_dim),
        head=ResMultiLayerPerceptron(a.feat_dim, a.hidden_dim, a.num_layers, a.output_dim, 0.8)), "pasca_v1.py:7-13"),
    "PASCA_V2": (("feat_dim", "output_dim", "hidden_dim", "num_layers"), lambda a: dict(
        graph=LaplacianGraphOp(a.prop_steps, r=0.5),
        msg=LearnableWeightedMessageOp(1, a.prop_steps + 1, "gate", a.feat_dim),
        head=ResMultiLayerPerceptron(a.feat_dim, a.hidden_dim, a.num_layers, a.output_dim, 0.8)), "pasca_v2.py:7-13"),
    "PASCA_V3": (("post_steps", "feat_dim", "output_dim", "hidden_dim", "num_layers"), lambda a: dict(
        graph=LaplacianGraphOp(a.prop_steps, r=0.5),
        msg=LearnableWeightedMessageOp(1, a.prop_steps + 1, "gate", a.feat_dim),
        head=ResMultiLayerPerceptron(a.feat_dim, a.hidden_dim, a.num_layers, a.output_dim, 0.8),
        post_graph=PprGraphOp(a.post_steps, r=0.5, alpha=0.3), post_msg=LastMessageOp()),
        "pasca_v3.py:7-15 (+ PPR(alpha=0.3) post-propagation)"),
}


class _Args:
    pass


def _bind(name, spec, args, kwargs):
    """positional / keyword arguments of the reference signature -> attribute bag"""
    names = ["prop_steps"] + [s if isinstance(s, str) else s[0] for s in spec]
    defaults = {s[0]: s[1] for s in spec if not isinstance(s, str)}
    if len(args) > len(names):
        raise TypeError(f"{name}() takes {len(names)} positional arguments but {len(args)} were given")
    bag = _Args()
    given = dict(zip(names, args))
    for k, v in kwargs.items():
        if k not in names:
            raise TypeError(f"{name}() got an unexpected keyword argument '{k}'")
        if k in given:
            raise TypeError(f"{name}() got multiple values for argument '{k}'")
        given[k] = v
    for n in names:
        if n in given:
            setattr(bag, n, given[n])
        elif n in defaults:
            setattr(bag, n, defaults[n])
        else:
            raise TypeError(f"{name}() missing required argument: '{n}'")
    return bag


def _make(name):
    spec, build, cite = _ZOO[name]

    def __init__(self, *args, **kwargs):
        a = _bind(name, spec, args, kwargs)
        BaseSGAPModel.__init__(self, a.prop_steps, a.feat_dim, a.output_dim)
        slots = build(a)
        self._pre_graph_op, self._pre_msg_op, self._base_model = slots["graph"], slots["msg"], slots["head"]
        self._post_graph_op, self._post_msg_op = slots.get("post_graph"), slots.get("post_msg")

    return type(name, (BaseSGAPModel,), {"__init__": __init__, "__doc__": f"{name}: reference sgl/models/homo/{cite}",
                                         "__module__": __name__})


SGC, SSGC, SIGN, GBP, GAMLP, GAMLPRecursive, NAFS, PASCA_V1, PASCA_V2, PASCA_V3 = (
    _make(n) for n in ("SGC", "SSGC", "SIGN", "GBP", "GAMLP", "GAMLPRecursive", "NAFS", "PASCA_V1", "PASCA_V2", "PASCA_V3"))
