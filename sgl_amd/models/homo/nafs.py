"""NAFS: over-smoothing-distance weighted hops, no trainable head.  Reference: sgl/models/homo/nafs.py:7-13"""
from ..base_model import BaseSGAPModel
from ..simple_models import IdenticalMapping, LogisticRegression, MultiLayerPerceptron, ResMultiLayerPerceptron  # noqa: F401
from ...operators.graph_op import LaplacianGraphOp, PprGraphOp  # noqa: F401
from ...operators.message_op import (  # noqa: F401
    ConcatMessageOp, IterateLearnableWeightedMessageOp, LastMessageOp, LearnableWeightedMessageOp, MeanMessageOp,
    OverSmoothDistanceWeightedOp, SimpleWeightedMessageOp)


class NAFS(BaseSGAPModel):
    def __init__(self, prop_steps, feat_dim, output_dim):
        super(NAFS, self).__init__(prop_steps, feat_dim, output_dim)
        self._pre_graph_op = LaplacianGraphOp(prop_steps, r=0.5)
        self._pre_msg_op = OverSmoothDistanceWeightedOp()
        self._base_model = IdenticalMapping()
