"""Per-hop MLP projection then concatenation (NARS/SIGN style).  Reference: message_op/projected_concat_message_op.py:9-28.
The projections are dense GEMMs (rocBLAS through torch.nn.Linear: out of the hot path's scope); the aggregator part
-- the hstack of the projected hops (:28) -- is the library's concat kernel, carried through autograd by device.hop_concat_grad
(forward sgl_hop_concat_f32, backward the column slices of the incoming gradient)."""
import torch.nn.functional as F
from torch.nn import ModuleList

from ...models.simple_models import MultiLayerPerceptron
from ..base_op import MessageOp
from ._common import concat_hops


class ProjectedConcatMessageOp(MessageOp):
    def __init__(self, start, end, feat_dim, hidden_dim, num_layers):
        super(ProjectedConcatMessageOp, self).__init__(start, end)
        self._aggr_type = "proj_concat"
        self.__learnable_weight = ModuleList(
            [MultiLayerPerceptron(feat_dim, hidden_dim, num_layers, hidden_dim) for _ in range(end - start)])

    def _combine(self, feat_list):
        hops = feat_list[self._start:self._end]
        cols = [self.__learnable_weight[0](hops[0])]                      # hop `start` is not activated (:22)
        cols += [F.relu(mlp(x)) for mlp, x in zip(list(self.__learnable_weight)[1:], hops[1:])]
        return concat_hops(cols)
