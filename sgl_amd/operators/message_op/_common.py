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


def wants_grad(feats):
    """do gradients have to flow through these hop matrices?  (e.g. when the outputs of a ProjectedConcat / MLP are fed into Concat
    or Mean; SGAP pre-propagation itself never needs it.)  The parameter-free aggregators then run the same HIP kernels inside
    autograd Functions whose backward is the broadcast / slice / selection the reference's torch expression has
    (device.hop_reduce_grad / hop_concat_grad); only the NAFS op still evaluates its differentiable torch expression."""
    return torch.is_grad_enabled() and any(torch.is_tensor(f) and f.requires_grad for f in feats)


def torch_combine(kind, feats, divisor=None):
    """the reference's expressions, differentiable (message_op/{sum,mean,max,min,concat}_message_op.py)"""
    if kind == "sum":
        return sum(feats)
    if kind == "mean":
        return sum(feats) / (len(feats) if divisor is None else divisor)
    if kind == "max":
        return torch.stack(feats, dim=0).max(dim=0)[0]
    if kind == "min":
        return torch.stack(feats, dim=0).min(dim=0)[0]
    if kind == "concat":
        return torch.hstack(feats)
    if kind == "nafs":
        # over_smooth_distance_op.py:11-33 with its per-node Python loop written as one weighted sum (same values)
        x0 = feats[0]
        n0 = torch.norm(x0, 2, 1).add(1e-10)
        scores = [torch.div(torch.div((x0 * f).sum(1), torch.norm(f, 2, 1).add(1e-10)), n0).unsqueeze(-1) for f in feats]
        w = torch.softmax(torch.cat(scores, dim=1), dim=1)
        out = 0.
        for h, f in enumerate(feats):
            out = out + w[:, h:h + 1] * f
        return out
    raise ValueError(kind)


REDUCE = {"sum": _lib.SGL_REDUCE_SUM, "mean": _lib.SGL_REDUCE_MEAN, "max": _lib.SGL_REDUCE_MAX, "min": _lib.SGL_REDUCE_MIN}


def reduce_hops(kind, feat_list, divisor=None):
    feats, home = device_hops(feat_list)
    if wants_grad(feat_list):
        return back_home(dev.hop_reduce_grad(REDUCE[kind], feats, divisor=divisor), home)
    return back_home(dev.hop_reduce(REDUCE[kind], feats), home)


def concat_hops(feat_list):
    feats, home = device_hops(feat_list)
    if wants_grad(feat_list):
        return back_home(dev.hop_concat_grad(feats), home)
    return back_home(dev.hop_concat(feats), home)
