"""PaSca V3: Laplacian pre-propagation + gate aggregation + ResMLP, PPR(alpha=0.3) post-propagation.
Reference: sgl/models/homo/pasca_v3.py:7-15"""
from ..base_model import BaseSGAPModel
from ..simple_models import ResMultiLayerPerceptron
from ...operators.graph_op import LaplacianGraphOp, PprGraphOp
from ...operators.message_op import LastMessageOp, LearnableWeightedMessageOp


class PASCA_V3(BaseSGAPModel):
    def __init__(self, prop_steps, post_steps, feat_dim, output_dim, hidden_dim, num_layers):
        super(PASCA_V3, self).__init__(prop_steps, feat_dim, output_dim)
        self._pre_graph_op = LaplacianGraphOp(prop_steps, r=0.5)
        self._pre_msg_op = LearnableWeightedMessageOp(1, prop_steps + 1, "gate", feat_dim)
        self._base_model = ResMultiLayerPerceptron(feat_dim, hidden_dim, num_layers, output_dim, 0.8)
        self._post_graph_op = PprGraphOp(post_steps, r=0.5, alpha=0.3)
        self._post_msg_op = LastMessageOp()
