from ..base_op import MessageOp
from ._common import reduce_hops


class SumMessageOp(MessageOp):
    """sum over feat_list[start:end]  (reference: message_op/sum_message_op.py)"""

    def __init__(self, start, end):
        super(SumMessageOp, self).__init__(start, end)
        self._aggr_type = "sum"

    def _combine(self, feat_list):
        return reduce_hops("sum", feat_list[self._start:self._end])
