from ... import device as dev
from ..base_op import MessageOp
from ._common import back_home, device_hops, torch_combine, wants_grad


class OverSmoothDistanceWeightedOp(MessageOp):
    """NAFS: W = softmax_h(cos(X_0[n], X_h[n])), out[n] = sum_h W[n,h] X_h[n]
    (reference: message_op/over_smooth_distance_op.py:6-33 -- its O(N*H) Python loop is one fused HIP pass here)."""

    def __init__(self):
        super(OverSmoothDistanceWeightedOp, self).__init__()
        self._aggr_type = 'over_smooth_dis_weighted'

    def _combine(self, feat_list):
        if wants_grad(feat_list):
            return torch_combine("nafs", list(feat_list))
        feats, home = device_hops(feat_list)
        return back_home(dev.nafs_aggregate(feats), home)
