from ..base_op import MessageOp
from ._common import reduce_hops


class MaxMessageOp(MessageOp):
    """max over feat_list[start:end]  (reference: message_op/max_message_op.py)"""

    def __init__(self, start, end):
        super(MaxMessageOp, self).__init__(start, end)
        self._aggr_type = "max"

    def _combine(self, feat_list):
        return reduce_hops("max", feat_list[self._start:self._end])
