"""Recursive gated hop combination (GAMLP-R).  Reference: message_op/iterate_learnable_weighted_message_op.py:8-51.

Per step i the gate sees [X_i || acc] and the running soft-max over the scores so far re-weights ALL hops
0..i (the reference re-applies soft-max to already soft-maxed columns, :39 -- kept).

The recursion is ROW-LOCAL and acc is always a per-row weighted sum of the hops, acc_i[n] = sum_j W_i[n, j] X_j[n], so the
gate's view of it is a combination of per-hop scalars:

    Linear([X_i || acc_{i-1}])[n] = <X_i[n], w_x> + sum_j W_{i-1}[n, j] <X_j[n], w_acc> + b

With a[n, h] = <X_h[n], w_x> and c[n, h] = <X_h[n], w_acc> the whole loop runs on [n, H] scalars.  On the device the operator is
ONE pass over the hop matrices (device.hop_recursive -> sgl_hop_recursive_f32: the hop rows in registers, 2 H row-dots, the
recursion on the scalars, the final weighted sum) where the step-by-step form makes H (H + 3) / 2 reads of a hop matrix and writes
H intermediate accumulators; rows the register-resident kernel cannot take go through two row-dot passes, the [n, H] recursion
in torch and one weighted-sum pass.  The backward re-runs the [n, H] recursion under autograd and takes the weight gradients from
sgl_hop_colsum_f32.  The step-by-step form stays for host tensors and for start != 0 (where the reference's own indexing, kept
below, is only meaningful by accident)."""
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
        f0 = feat_list[s]
        if s == 0 and 0 < e <= len(feat_list) and lin.weight.numel() == 2 * d and torch.is_tensor(f0) and f0.is_cuda and f0.dtype == torch.float32 and f0.dim() == 2:
            from ... import device as dev
            from ..utils import _rowmajor
            return dev.hop_recursive([_rowmajor(x) for x in feat_list[:e]], lin.weight, lin.bias)
        acc = feat_list[s]
        weights = None
        for i in range(s, e):
            score = torch.sigmoid(feat_list[i] @ w_x + acc @ w_acc + lin.bias).unsqueeze(1)
            weights = score if weights is None else torch.hstack((weights, score))
            weights = F.softmax(weights, dim=1)
            # hops s .. s+i (absolute i, as the reference indexes: only meaningful for start == 0, its sole use)
            acc = two_dim_weighted_add(feat_list[s:s + i + 1], weight_list=weights[:, :i + 1])
        return acc
