"""The parameter-free hop aggregators: last / concat / sum / mean / max / min.

Reference classes: sgl/operators/message_op/{last,concat,sum,mean,max,min}_message_op.py (each a 10-line file around
one torch expression: `feat_list[-1]`, `hstack`, Python `sum()`, `sum()/H`, `stack().max(0)[0]`, `stack().min(0)[0]`).
Here the four reductions are ONE streaming HIP kernel (sgl_hop_reduce_f32: a single pass over the H hop matrices,
no [H, N, d] stack is ever materialised) and concat is a strided copy kernel; results are bit-identical to the
reference's (same left-to-right order, one true division for mean, NaN-propagating max/min)."""
import torch

from ..base_op import MessageOp
from ._common import concat_hops, reduce_hops, wants_grad


class LastMessageOp(MessageOp):
    """the deepest hop, untouched (and un-copied): ignores start / end like the reference"""

    def __init__(self):
        super(LastMessageOp, self).__init__()
        self._aggr_type = "last"

    def _combine(self, feat_list):
        return feat_list[-1]

    def fused_spec(self, n_hops):
        return {"kind": "last"}


class ConcatMessageOp(MessageOp):
    """[X_start | ... | X_{end-1}] side by side -> [n, (end-start) d]"""

    def __init__(self, start, end):
        super(ConcatMessageOp, self).__init__(start, end)
        self._aggr_type = "concat"

    @staticmethod
    def _slab_view(feats):
        """the hops side by side in one buffer already (GraphOp slab_hops layout)?  Then their concatenation is a view."""
        f0 = feats[0]
        if not (torch.is_tensor(f0) and f0.dim() == 2 and f0.shape[0] > 0):
            return None
        n, d = f0.shape
        base = f0.untyped_storage().data_ptr()
        for k, f in enumerate(feats):
            if not (torch.is_tensor(f) and f.shape == f0.shape and f.dtype == f0.dtype and f.device == f0.device
                    and f.untyped_storage().data_ptr() == base and f.stride() == f0.stride() and f.stride(1) == 1
                    and f.storage_offset() == f0.storage_offset() + k * d):
                return None
        if n > 1 and f0.stride(0) < len(feats) * d:
            return None
        return torch.as_strided(f0, (n, len(feats) * d), (f0.stride(0), 1), f0.storage_offset())

    def _combine(self, feat_list):
        hops = feat_list[self._start:self._end]
        if len(hops) > 1 and not wants_grad(hops):
            view = self._slab_view(hops)
            if view is not None:
                return view
        return concat_hops(hops)


def _reduction(kind, doc):
    class _Op(MessageOp):
        def __init__(self, start, end):
            super(_Op, self).__init__(start, end)
            self._aggr_type = kind

        def _combine(self, feat_list):
            hops = feat_list[self._start:self._end]
            if wants_grad(hops):
                return reduce_hops(kind, hops, divisor=(self._end - self._start) if kind == "mean" else None)
            if kind == "mean" and len(hops) != self._end - self._start:
                # the reference divides by (end - start) whatever the slice held (mean_message_op.py:10)
                total = reduce_hops("sum", hops)
                # a 0-dim DEVICE divisor: torch's GPU kernel turns division by a host scalar into a multiplication by
                # its reciprocal, which is not the reference's (CPU) true division in the last bit
                return total / torch.full((), float(self._end - self._start), dtype=torch.float32, device=total.device)
            return reduce_hops(kind, hops)

        def fused_spec(self, n_hops):
            if kind not in ("sum", "mean", "max", "min") or not (isinstance(self._start, int) and isinstance(self._end, int)):
                return None
            if self._start < 0 or self._end <= self._start:
                return None
            spec = {"kind": kind, "start": self._start, "end": self._end}
            if kind == "mean":
                spec["divisor"] = self._end - self._start
            return spec

    _Op.__name__ = _Op.__qualname__ = kind.capitalize() + "MessageOp"
    _Op.__doc__ = doc
    return _Op


SumMessageOp = _reduction("sum", "X_start + ... + X_{end-1}, accumulated left to right")
MeanMessageOp = _reduction("mean", "the hop sum followed by one true division by (end - start)")
MaxMessageOp = _reduction("max", "element-wise maximum over the hops (NaN propagates, like torch.max)")
MinMessageOp = _reduction("min", "element-wise minimum over the hops (NaN propagates, like torch.min)")
