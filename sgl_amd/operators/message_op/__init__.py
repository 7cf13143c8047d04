from .concat_message_op import ConcatMessageOp
from .iterate_learnable_weighted_message_op import IterateLearnableWeightedMessageOp
from .last_message_op import LastMessageOp
from .learnable_weighted_messahe_op import LearnableWeightedMessageOp
from .max_message_op import MaxMessageOp
from .mean_message_op import MeanMessageOp
from .min_message_op import MinMessageOp
from .over_smooth_distance_op import OverSmoothDistanceWeightedOp
from .projected_concat_message_op import ProjectedConcatMessageOp
from .simple_weighted_message_op import SimpleWeightedMessageOp
from .sum_message_op import SumMessageOp

__all__ = [
    "ConcatMessageOp",
    "IterateLearnableWeightedMessageOp",
    "LastMessageOp",
    "LearnableWeightedMessageOp",
    "MaxMessageOp",
    "MeanMessageOp",
    "MinMessageOp",
    "ProjectedConcatMessageOp",
    "SimpleWeightedMessageOp",
    "SumMessageOp",
    "OverSmoothDistanceWeightedOp",
]
