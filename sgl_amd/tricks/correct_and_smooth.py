"""Correct & Smooth on the MI355X.  Reference: sgl/tricks/correct_and_smooth.py:6-62 (same constructor and
`correct` / `smooth` signatures; y_true holds ALL labels, `mask` selects the training nodes).  Both stages are
label_propagation runs, i.e. fused SpMM+epilogue kernels; the scaling arithmetic in between is a handful of
row-wise torch ops on [N, C] matrices that already live in HBM."""
import numpy as np
import torch
import torch.nn.functional as F

from .utils import label_propagation


class CorrectAndSmooth:
    def __init__(self, num_correct_layers, correct_alpha, num_smooth_layers, smooth_alpha, autoscale=True, scale=1.0,
                 device="cuda"):
        self._num_correct_layers = num_correct_layers
        self._correct_alpha = correct_alpha
        self._num_smooth_layers = num_smooth_layers
        self._smooth_alpha = smooth_alpha
        self._autoscale = autoscale
        self._scale = scale
        self._device = torch.device(device)

    def _prep(self, y_soft, y_true, mask):
        y_soft = y_soft.detach().to(self._device, torch.float32)
        y_true = y_true.to(self._device)
        if y_true.dtype == torch.long:
            y_true = F.one_hot(y_true.view(-1), y_soft.size(-1)).to(y_soft.dtype)
        mask = torch.as_tensor(np.asarray(mask) if not torch.is_tensor(mask) else mask).to(self._device)
        return y_soft, y_true, mask

    @torch.no_grad()
    def correct(self, y_soft, y_true, mask, adj):
        y_soft, y_true, mask = self._prep(y_soft, y_true, mask)
        error = torch.zeros_like(y_soft)
        error[mask] = y_true[mask] - y_soft[mask]
        num_true = mask.shape[0] if mask.dtype == torch.long else int(mask.sum())
        if self._autoscale:
            smoothed = label_propagation(error, adj, self._num_correct_layers, self._correct_alpha,
                                         post_process=(-1., 1.), device=self._device)
            sigma = error[mask].abs().sum() / num_true
            scale = sigma / smoothed.abs().sum(dim=1, keepdim=True)
            scale[scale.isinf() | (scale > 1000)] = 1.0
            return y_soft + smoothed * scale

        def fix_input(x):
            x[mask] = error[mask]
            return x

        smoothed = label_propagation(error, adj, self._num_correct_layers, self._correct_alpha,
                                     post_process=fix_input, device=self._device)
        return y_soft + smoothed * self._scale

    @torch.no_grad()
    def smooth(self, y_soft, y_true, mask, adj):
        y_soft, y_true, mask = self._prep(y_soft, y_true, mask)
        y_soft = y_soft.clone()
        y_soft[mask] = y_true[mask]
        return label_propagation(y_soft, adj, self._num_smooth_layers, self._smooth_alpha, device=self._device)
