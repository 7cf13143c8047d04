"""Dense heads used by the SGAP models (LogisticRegression / MLP / ResMLP / identity).

These are plain torch.nn modules: dense GEMMs belong to rocBLAS/hipBLASLt through PyTorch-ROCm and are out of
the hot-path scope.  Class names, constructor signatures, forward semantics and the
private attribute names (hence state_dict keys such as `_MultiLayerPerceptron__fcs.0.weight`) match
sgl/models/simple_models.py:86-184 so checkpoints of the reference load unchanged."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _linear_stack(sizes):
    return nn.ModuleList([nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:])])


def _norm_stack(bn, hidden_dim, count):
    return nn.ModuleList([nn.BatchNorm1d(hidden_dim) for _ in range(count)]) if bn else None


class _SharedSlopePReLUFn(torch.autograd.Function):
    """F.prelu with ONE shared slope; the backward as three element-wise passes and a sum.  torch's `_prelu_kernel_backward`
    broadcasts the slope through its generic un-vectorised element-wise kernel: 0.48 ms per call on a [50 000, 256] activation,
    0.96 of the 3.0 ms GAMLP training step at the products shape (profiles/r04_train_step.log) -- against ~0.15 ms this way."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return F.prelu(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.where(x > 0, g, g * w)
        if ctx.needs_input_grad[1]:
            dw = (g * x.clamp(max=0)).sum().reshape(w.shape)         # d/dw = x where x <= 0, else 0
        return dx, dw


class _SharedSlopePReLU(nn.PReLU):
    """nn.PReLU() (same parameter, same state_dict key, same forward) with the cheaper backward above"""

    def forward(self, x):
        if self.weight.numel() != 1 or not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
            return F.prelu(x, self.weight)
        return _SharedSlopePReLUFn.apply(x, self.weight)


class IdenticalMapping(nn.Module):
    def forward(self, feature):
        return feature


class LogisticRegression(nn.Module):
    def __init__(self, feat_dim, output_dim):
        super(LogisticRegression, self).__init__()
        self.__fc = nn.Linear(feat_dim, output_dim)

    def forward(self, feature):
        return self.__fc(feature)


class MultiLayerPerceptron(nn.Module):
    """Linear -> [BN] -> PReLU (one shared slope) -> Dropout, repeated; last layer linear.
    Xavier-uniform weights with the ReLU gain, zero biases."""

    def __init__(self, feat_dim, hidden_dim, num_layers, output_dim, dropout=0.5, bn=False):
        super(MultiLayerPerceptron, self).__init__()
        if num_layers < 2:
            raise ValueError("MLP must have at least two layers!")
        self.__num_layers = num_layers
        self.__fcs = _linear_stack([feat_dim] + [hidden_dim] * (num_layers - 1) + [output_dim])
        self.__bn = bn
        if bn is True:
            self.__bns = _norm_stack(True, hidden_dim, num_layers - 1)
        self.__dropout = nn.Dropout(dropout)
        self.__prelu = _SharedSlopePReLU()
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        for fc in self.__fcs:
            nn.init.xavier_uniform_(fc.weight, gain=gain)
            nn.init.zeros_(fc.bias)

    def forward(self, feature):
        hidden = list(self.__fcs)[:-1]
        for i, fc in enumerate(hidden):
            feature = fc(feature)
            if self.__bn is True:
                feature = self.__bns[i](feature)
            feature = self.__dropout(self.__prelu(feature))
        return self.__fcs[-1](feature)


class ResMultiLayerPerceptron(nn.Module):
    """Dropout-first MLP whose hidden layers add the PREVIOUS layer's activation (not the running sum)."""

    def __init__(self, feat_dim, hidden_dim, num_layers, output_dim, dropout=0.8, bn=False):
        super(ResMultiLayerPerceptron, self).__init__()
        if num_layers < 2:
            raise ValueError("ResMLP must have at least two layers!")
        self.__num_layers = num_layers
        self.__fcs = _linear_stack([feat_dim] + [hidden_dim] * (num_layers - 1) + [output_dim])
        self.__bn = bn
        if bn is True:
            self.__bns = _norm_stack(True, hidden_dim, num_layers - 1)
        self.__dropout = nn.Dropout(dropout)
        self.__relu = nn.ReLU()

    def _block(self, i, feature):
        feature = self.__fcs[i](self.__dropout(feature))
        if self.__bn is True:
            feature = self.__bns[i](feature)
        return self.__relu(feature)

    def forward(self, feature):
        feature = self._block(0, feature)
        residual = feature
        for i in range(1, self.__num_layers - 1):
            act = self._block(i, feature)
            feature, residual = act + residual, act
        return self.__fcs[-1](self.__dropout(feature))
