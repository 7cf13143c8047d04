"""Shared plumbing for the MessageOp kernels: move the hop list to the GPU in the layout the HIP kernels need."""
import torch

from ... import _lib
from ... import device as dev
from ..utils import _as_device_list, _rowmajor


def device_hops(feat_list):
    """-> (list of row-major float32 CUDA tensors, original device or None when already on the GPU)"""
    if len(feat_list) == 0:
        raise ValueError("empty feature list")
    feats, home = _as_device_list(feat_list)
    return [_rowmajor(f) for f in feats], home


def back_home(t, home):
    return t if home is None else t.to(home)


def no_grad_inputs(feats, what):
    if torch.is_grad_enabled() and any(f.requires_grad for f in feats):
        raise NotImplementedError(
            f"{what}: the HIP kernel of this non-learnable aggregator is forward-only; detach the hop features "
            "(SGAP pre-propagation never needs their gradient)")


REDUCE = {"sum": _lib.SGL_REDUCE_SUM, "mean": _lib.SGL_REDUCE_MEAN, "max": _lib.SGL_REDUCE_MAX, "min": _lib.SGL_REDUCE_MIN}


def reduce_hops(kind, feat_list):
    feats, home = device_hops(feat_list)
    no_grad_inputs(feats, kind)
    return back_home(dev.hop_reduce(REDUCE[kind], feats), home)
