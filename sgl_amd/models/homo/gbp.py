"""GBP: alpha-decayed hop sum (the ctor's `r` is ignored by the reference too).  Reference: sgl/models/homo/gbp.py:7-13"""
from ..base_model import BaseSGAPModel
from ..simple_models import IdenticalMapping, LogisticRegression, MultiLayerPerceptron, ResMultiLayerPerceptron  # noqa: F401
from ...operators.graph_op import LaplacianGraphOp, PprGraphOp  # noqa: F401
from ...operators.message_op import (  # noqa: F401
    ConcatMessageOp, IterateLearnableWeightedMessageOp, LastMessageOp, LearnableWeightedMessageOp, MeanMessageOp,
    OverSmoothDistanceWeightedOp, SimpleWeightedMessageOp)


class GBP(BaseSGAPModel):
    def __init__(self, prop_steps, feat_dim, output_dim, hidden_dim, num_layers, r=0.5, alpha=0.85):
        super(GBP, self).__init__(prop_steps, feat_dim, output_dim)
        self._pre_graph_op = LaplacianGraphOp(prop_steps, r=0.5)
        self._pre_msg_op = SimpleWeightedMessageOp(0, prop_steps + 1, "alpha", alpha)
        self._base_model = MultiLayerPerceptron(feat_dim, hidden_dim, num_layers, output_dim)
