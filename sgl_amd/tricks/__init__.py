"""Consumers of the same device SpMM beyond GraphOp.propagate:
label propagation / Correct&Smooth (reference: sgl/tricks) and the NAFS feature-smoothing pipeline of the
NAFS clustering / link-prediction tasks (reference: sgl/tasks/node_clustering.py:205-258); and the label use / label reuse
loop around `model.preprocess` (reference: sgl/tasks/node_classification_with_label_use.py:58-137)."""
from .correct_and_smooth import CorrectAndSmooth
from .label_reuse import add_labels, label_reuse, predict_all
from .nafs_features import nafs_ensemble_features, nafs_ensemble_sweep
from .utils import label_propagation

__all__ = ["CorrectAndSmooth", "label_propagation", "nafs_ensemble_features", "nafs_ensemble_sweep", "add_labels", "label_reuse", "predict_all"]
