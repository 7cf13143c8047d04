from .base_model import BaseSGAPModel
from .simple_models import IdenticalMapping, LogisticRegression, MultiLayerPerceptron, ResMultiLayerPerceptron

__all__ = ["BaseSGAPModel", "IdenticalMapping", "LogisticRegression", "MultiLayerPerceptron", "ResMultiLayerPerceptron"]
