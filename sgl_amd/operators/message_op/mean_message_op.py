from ..base_op import MessageOp
from ._common import reduce_hops


class MeanMessageOp(MessageOp):
    """mean over feat_list[start:end]  (reference: message_op/mean_message_op.py)"""

    def __init__(self, start, end):
        super(MeanMessageOp, self).__init__(start, end)
        self._aggr_type = "mean"

    def _combine(self, feat_list):
        return reduce_hops("mean", feat_list[self._start:self._end])
