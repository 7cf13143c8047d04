from ..base_op import MessageOp
from ._common import reduce_hops


class MinMessageOp(MessageOp):
    """min over feat_list[start:end]  (reference: message_op/min_message_op.py)"""

    def __init__(self, start, end):
        super(MinMessageOp, self).__init__(start, end)
        self._aggr_type = "min"

    def _combine(self, feat_list):
        return reduce_hops("min", feat_list[self._start:self._end])
