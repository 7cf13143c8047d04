from ..base_op import MessageOp


class LastMessageOp(MessageOp):
    """feat_list[-1]  (reference: message_op/last_message_op.py:4-10)"""

    def __init__(self):
        super(LastMessageOp, self).__init__()
        self._aggr_type = "last"

    def _combine(self, feat_list):
        return feat_list[-1]
