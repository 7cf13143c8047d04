"""MessageOp plugins: per-hop aggregators over the list [X, A_hat X, ..., A_hat^k X] (API of sgl.operators.message_op)."""
from .stateless_ops import (ConcatMessageOp, LastMessageOp, MaxMessageOp, MeanMessageOp, MinMessageOp,
                            SumMessageOp)  # isort: skip
from .simple_weighted_message_op import SimpleWeightedMessageOp
from .learnable_weighted_messahe_op import LearnableWeightedMessageOp
from .iterate_learnable_weighted_message_op import IterateLearnableWeightedMessageOp
from .projected_concat_message_op import ProjectedConcatMessageOp
from .over_smooth_distance_op import OverSmoothDistanceWeightedOp

__all__ = sorted([
    "LastMessageOp", "ConcatMessageOp", "SumMessageOp", "MeanMessageOp", "MaxMessageOp", "MinMessageOp",
    "SimpleWeightedMessageOp", "LearnableWeightedMessageOp", "IterateLearnableWeightedMessageOp",
    "ProjectedConcatMessageOp", "OverSmoothDistanceWeightedOp",
])
