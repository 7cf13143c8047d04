"""Recursive gated hop combination (GAMLP-R).  Reference: message_op/iterate_learnable_weighted_message_op.py:8-51.

Per step i the gate sees [X_i || acc] and the running soft-max over the scores so far re-weights ALL hops
0..i (the reference re-applies soft-max to already soft-maxed columns, :39 -- kept).  The O(n d) work per
step -- the re-weighted hop sum -- runs in the HIP weighted-sum kernel (with its hand-written backward);
the gate Linear(2d -> 1) is split as x . W_x + acc . W_acc + b so no [n, 2d] hstack is materialised."""
import torch
import torch.nn.functional as F
from torch.nn import Linear

from ..base_op import MessageOp
from ..utils import two_dim_weighted_add


class IterateLearnableWeightedMessageOp(MessageOp):
    # 'recursive' needs one additional parameter 'feat_dim'
    def __init__(self, start, end, combination_type, *args):
        super(IterateLearnableWeightedMessageOp, self).__init__(start, end)
        self._aggr_type = "iterate_learnable_weighted"
        if combination_type not in ("recursive",):
            raise ValueError("Invalid weighted combination type! Type must be 'recursive'.")
        if len(args) != 1:
            raise ValueError("Invalid parameter numbers for the recursive iterate weighted aggregator!")
        self.__combination_type = combination_type
        self.__learnable_weight = Linear(2 * args[0], 1)

    def _combine(self, feat_list):
        s, e = self._start, self._end
        lin = self.__learnable_weight
        d = feat_list[s].shape[1]
        w_x, w_acc = lin.weight.view(-1)[:d], lin.weight.view(-1)[d:]
        acc = feat_list[s]
        weights = None
        for i in range(s, e):
            score = torch.sigmoid(feat_list[i] @ w_x + acc @ w_acc + lin.bias).unsqueeze(1)
            weights = score if weights is None else torch.hstack((weights, score))
            weights = F.softmax(weights, dim=1)
            # hops s .. s+i (absolute i, as the reference indexes: only meaningful for start == 0, its sole use)
            acc = two_dim_weighted_add(feat_list[s:s + i + 1], weight_list=weights[:, :i + 1])
        return acc
