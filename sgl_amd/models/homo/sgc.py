"""SGC (Wu et al. 2019): A_hat^K X -> logistic regression.  Reference: sgl/models/homo/sgc.py:7-13"""
from ..base_model import BaseSGAPModel
from ..simple_models import IdenticalMapping, LogisticRegression, MultiLayerPerceptron, ResMultiLayerPerceptron  # noqa: F401
from ...operators.graph_op import LaplacianGraphOp, PprGraphOp  # noqa: F401
from ...operators.message_op import (  # noqa: F401
    ConcatMessageOp, IterateLearnableWeightedMessageOp, LastMessageOp, LearnableWeightedMessageOp, MeanMessageOp,
    OverSmoothDistanceWeightedOp, SimpleWeightedMessageOp)


class SGC(BaseSGAPModel):
    def __init__(self, prop_steps, feat_dim, output_dim):
        super(SGC, self).__init__(prop_steps, feat_dim, output_dim)
        self._pre_graph_op = LaplacianGraphOp(prop_steps, r=0.5)
        self._pre_msg_op = LastMessageOp()
        self._base_model = LogisticRegression(feat_dim, output_dim)
