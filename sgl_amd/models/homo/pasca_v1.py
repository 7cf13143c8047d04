"""PaSca V1 (the reference passes feat_dim where prop_steps is expected -> weight vector of length feat_dim+1; kept).  Reference: sgl/models/homo/pasca_v1.py:7-13"""
from ..base_model import BaseSGAPModel
from ..simple_models import IdenticalMapping, LogisticRegression, MultiLayerPerceptron, ResMultiLayerPerceptron  # noqa: F401
from ...operators.graph_op import LaplacianGraphOp, PprGraphOp  # noqa: F401
from ...operators.message_op import (  # noqa: F401
    ConcatMessageOp, IterateLearnableWeightedMessageOp, LastMessageOp, LearnableWeightedMessageOp, MeanMessageOp,
    OverSmoothDistanceWeightedOp, SimpleWeightedMessageOp)


class PASCA_V1(BaseSGAPModel):
    def __init__(self, prop_steps, feat_dim, output_dim, hidden_dim, num_layers):
        super(PASCA_V1, self).__init__(prop_steps, feat_dim, output_dim)
        self._pre_graph_op = PprGraphOp(prop_steps, r=0.5, alpha=0.1)
        self._pre_msg_op = LearnableWeightedMessageOp(1, prop_steps + 1, "simple", feat_dim)
        self._base_model = ResMultiLayerPerceptron(feat_dim, hidden_dim, num_layers, output_dim, 0.8)
