"""Fixed-weight hop combination: out = sum_h w_h X_h, one streaming HIP pass over the hop matrices.

Same constructor contract and exceptions as the reference class
(sgl/operators/message_op/simple_weighted_message_op.py:8-56):
  SimpleWeightedMessageOp(start, end, "alpha", a)            w_0 = a, w_h = (1 - a) w_{h-1}
  SimpleWeightedMessageOp(start, end, "hand_crafted", w)     w given as list or tensor
"""
import torch
from torch import Tensor

from ..base_op import MessageOp
from ..utils import one_dim_weighted_add

_KINDS = ("alpha", "hand_crafted")


def _decay_weights(alpha, n_hops):
    # geometric decay evaluated in Python float64 over ALL hops and only then sliced / cast to float32,
    # exactly as the reference does (:41-47) so the float32 weights are bit-identical
    ws = [alpha]
    while len(ws) < n_hops:
        ws.append((1 - alpha) * ws[-1])
    return ws


class SimpleWeightedMessageOp(MessageOp):
    def __init__(self, start, end, combination_type, *args):
        super(SimpleWeightedMessageOp, self).__init__(start, end)
        self._aggr_type = "simple_weighted"
        if combination_type not in _KINDS:
            raise ValueError("Invalid weighted combination type! Type must be 'alpha' or 'hand_crafted'.")
        if len(args) != 1:
            raise ValueError("Invalid parameter numbers for the simple weighted aggregator!")
        self._kind = combination_type
        self._alpha = None
        self._fixed = None
        (arg,) = args
        if self._kind == "alpha":
            if not isinstance(arg, float):
                raise TypeError("The alpha must be a float!")
            if not 0 <= arg <= 1:
                raise ValueError("The alpha must be a float in [0,1]!")
            self._alpha = arg
        else:
            if isinstance(arg, list):
                arg = torch.FloatTensor(arg)
            if not isinstance(arg, Tensor):
                raise TypeError("The input weight list must be a list or a tensor!")
            self._fixed = arg

    def weights(self, n_hops):
        """float32 weight vector for feat_list[start:end] given the total number of hops"""
        if self._kind == "alpha":
            return torch.FloatTensor(_decay_weights(self._alpha, n_hops)[self._start:self._end])
        return self._fixed

    def fused_spec(self, n_hops):
        if not (isinstance(self._start, int) and isinstance(self._end, int)) or self._start < 0 or self._end <= self._start:
            return None
        w = self.weights(n_hops)
        if w.dim() != 1 or w.numel() != min(self._end, n_hops) - self._start:
            return None           # a weight list that does not match the slice: let the unfused path raise as usual
        return {"kind": "wsum", "start": self._start, "end": self._end, "weights": w}

    def _combine(self, feat_list):
        w = self.weights(len(feat_list))
        first = feat_list[0] if len(feat_list) else None
        if torch.is_tensor(first) and first.is_cuda and not w.is_cuda:
            # the fixed weights live on the device after their first use: no upload per call (and the call records into a HIP graph)
            key = (len(feat_list), first.device, w.data_ptr() if self._kind != "alpha" else None, w._version if self._kind != "alpha" else None)
            held = getattr(self, "_w_dev", None)
            if held is None or held[0] != key:
                held = (key, w.to(first.device))
                self._w_dev = held
            w = held[1]
        return one_dim_weighted_add(feat_list[self._start:self._end], weight_list=w)
