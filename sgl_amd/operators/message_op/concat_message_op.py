from ... import device as dev
from ..base_op import MessageOp
from ._common import back_home, device_hops, no_grad_inputs


class ConcatMessageOp(MessageOp):
    """hstack(feat_list[start:end])  (reference: message_op/concat_message_op.py:6-12)"""

    def __init__(self, start, end):
        super(ConcatMessageOp, self).__init__(start, end)
        self._aggr_type = "concat"

    def _combine(self, feat_list):
        feats, home = device_hops(feat_list[self._start:self._end])
        no_grad_inputs(feats, "concat")
        return back_home(dev.hop_concat(feats), home)
