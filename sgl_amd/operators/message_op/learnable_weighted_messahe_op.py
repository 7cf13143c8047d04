"""Learnable hop weighting (GAMLP / PaSca): out = sum_h W[., h] X_h with W produced by trainable parameters.

API, parameter layout (state_dict keys) and numerics follow the reference class
(sgl/operators/message_op/learnable_weighted_messahe_op.py:10-101; the file name keeps the reference's
spelling so imports stay drop-in).  What differs is HOW it is computed on the MI355X:

  * the weighted hop sum (the only O(n d H) part) is a hand-written HIP kernel with a hand-written backward
    (sgl_hop_wsum2d_f32 / _bwd, sgl_hop_reduce_f32 WSUM / sgl_hop_wsum1d_bwd_f32);
  * 'gate' is ONE pass over the hops (sgl_hop_gate_f32: scores, sigmoid, softmax over the hops and the weighted sum while
    the H rows of a node sit in registers; every hop element is read once);
  * the 'ori_ref' / 'jk' scores never materialise the reference's [(H n), (H+1) d] `repeat`/`hstack` temporaries
    (:74-76,:81-84): Linear([ref || x_h]) = ref . W_ref + x_h . W_x + b, and ref . W_ref = sum_j x_j . W_ref[j] over the
    hops ref is stacked from -- both parts are row-dots taken in ONE pass over the hop list (sgl_hop_rowdot2_f32), no
    torch.hstack, no GEMV; the scramble below forces a second pass for the weighted sum.

Reference quirk reproduced on purpose: 'ori_ref' and 'jk' reshape the hop-major score vector with
.view(-1, H) (:78,:86), which pairs scores of DIFFERENT nodes/hops; 'gate' uses the intended
.view(H, -1).T (:71).  sgl.models.homo.GAMLP depends on the 'jk' behaviour, so parity means keeping it.
"""
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import Linear, Parameter

from ..base_op import MessageOp
from ..utils import one_dim_weighted_add, two_dim_weighted_add

_VECTOR_KINDS = ("simple", "simple_allow_neg")
_GATE_KINDS = ("gate", "ori_ref", "jk")
_NARGS = {"simple": 1, "simple_allow_neg": 1, "gate": 1, "ori_ref": 1, "jk": 2}


class LearnableWeightedMessageOp(MessageOp):
    # 'simple' / 'simple_allow_neg': (prop_steps);  'gate' / 'ori_ref': (feat_dim);  'jk': (prop_steps, feat_dim)
    def __init__(self, start, end, combination_type, *args):
        super(LearnableWeightedMessageOp, self).__init__(start, end)
        self._aggr_type = "learnable_weighted"
        if combination_type not in _NARGS:
            raise ValueError(
                "Invalid weighted combination type! Type must be 'simple', 'simple_allow_neg', 'gate', 'ori_ref' or 'jk'.")
        if len(args) != _NARGS[combination_type]:
            label = "simple" if combination_type in _VECTOR_KINDS else combination_type
            raise ValueError(f"Invalid parameter numbers for the {label} learnable weighted aggregator!")
        self.__combination_type = combination_type
        # the attribute name (and therefore the state_dict key `_LearnableWeightedMessageOp__learnable_weight...`)
        # matches the reference so checkpoints interchange
        if combination_type in _VECTOR_KINDS:
            init = torch.FloatTensor(1, args[0] + 1)
            nn.init.xavier_normal_(init)                 # same init stream as the reference (:33-35)
            self.__learnable_weight = Parameter(init.view(-1))
        elif combination_type == "gate":
            self.__learnable_weight = Linear(args[0], 1)
        elif combination_type == "ori_ref":
            self.__learnable_weight = Linear(2 * args[0], 1)
        else:  # jk
            prop_steps, feat_dim = args
            self.__learnable_weight = Linear(feat_dim + (prop_steps + 1) * feat_dim, 1)

    # ---- weights ------------------------------------------------------------------------------
    def _hop_scores(self, ref, hops):
        """score of (hop h, node b) = Linear([ref[b] || hops[h][b]]) laid out hop-major: flat[h*n + b]"""
        lin = self.__learnable_weight
        w = lin.weight.view(-1)
        d_ref = 0 if ref is None else ref.shape[1]
        shared = lin.bias if ref is None else ref @ w[:d_ref] + lin.bias           # [n] (or [1])
        w_x = w[d_ref:]
        if hops[0].is_cuda and hops[0].dtype == torch.float32:
            # all H per-hop products in ONE HIP pass over the hop matrices (sgl_hop_rowdot_f32)
            from ... import device as dev
            from ..utils import _rowmajor
            per_hop = dev.hop_scores([_rowmajor(x) for x in hops], w_x)            # [n, H]
            return (per_hop + shared.view(-1, 1)).t().reshape(-1)                  # hop-major flat [H*n]
        return torch.cat([(x @ w_x + shared) for x in hops], dim=0)                # [H*n]

    def _fused(self, feats):
        """these hops can go through the register-resident row kernels (device float32, <= 16 hops, d <= 512)"""
        f0 = feats[0]
        if not (torch.is_tensor(f0) and f0.is_cuda and f0.dtype == torch.float32 and f0.dim() == 2):
            return None
        from ... import device as dev
        from ..utils import _rowmajor
        rm = [_rowmajor(x) for x in feats]
        return rm if dev.gate_fusable(rm) else None

    def _ref_scores(self, feat_list, kind):
        """hop-major flat scores of 'ori_ref' / 'jk' from ONE pass over feat_list, or None when the fused kernel does not apply"""
        s, e = self._start, self._end
        L = len(feat_list)
        rm = self._fused(feat_list) if 0 <= s <= e <= L else None
        if rm is None:
            return None
        from ... import device as dev
        lin = self.__learnable_weight
        w = lin.weight.view(-1)
        d = rm[0].shape[1]
        d_ref = d if kind == "ori_ref" else L * d
        if w.numel() != d_ref + d:
            return None                                    # let the reference expression raise its own shape error
        if kind == "ori_ref":
            u = torch.cat([w[:d].view(1, d), torch.zeros((L - 1, d), dtype=w.dtype, device=w.device)]) if L > 1 else w[:d].view(1, d)
            mask = 1
        else:
            u, mask = w[:d_ref].view(L, d), (1 << L) - 1
        per_hop, shared = dev.hop_scores2(rm, w[d_ref:], u, mask, s, e)          # [n, H], [n]
        return (per_hop + (shared + lin.bias).view(-1, 1)).t().reshape(-1)        # hop-major flat [H*n]

    def hop_weights(self, feat_list):
        """the reference's `weight_list` (1-D [H] or 2-D [n, H])"""
        kind = self.__combination_type
        s, e = self._start, self._end
        H = e - s
        if kind == "simple":
            return F.softmax(torch.sigmoid(self.__learnable_weight[s:e]), dim=0)
        if kind == "simple_allow_neg":
            return self.__learnable_weight[s:e]
        hops = feat_list[s:e]
        if kind == "gate":
            flat = self._hop_scores(None, hops)
            return F.softmax(torch.sigmoid(flat.view(H, -1).T), dim=1)
        flat = self._ref_scores(feat_list, kind)
        if flat is None:
            ref = feat_list[0] if kind == "ori_ref" else torch.hstack(feat_list)
            flat = self._hop_scores(ref, hops)
        return F.softmax(torch.sigmoid(flat.view(-1, H)), dim=1)                   # reference's scrambled pairing

    def _combine(self, feat_list):
        hops = feat_list[self._start:self._end]
        if self.__combination_type == "gate" and len(hops) > 0 and len(hops) == self._end - self._start:
            rm = self._fused(hops)
            if rm is not None:                               # scores, sigmoid, softmax and the sum in one pass
                from ... import device as dev
                lin = self.__learnable_weight
                return dev.hop_gate(rm, lin.weight.view(-1), lin.bias)
        weight_list = self.hop_weights(feat_list)
        if self.__combination_type in _VECTOR_KINDS:
            return one_dim_weighted_add(hops, weight_list=weight_list)
        return two_dim_weighted_add(hops, weight_list=weight_list)
