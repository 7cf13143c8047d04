"""SGAP container: pre-propagation (GraphOp) -> hop aggregation (MessageOp) -> dense head.

Public behaviour of sgl/models/base_model.py:8-66 (BaseSGAPModel) with the hot path on the MI355X:
  * preprocess(): the K SpMMs and the non-learnable aggregation are HIP kernels; hop matrices stay in HBM;
  * forward(): the per-step row gather `feat[idx]` runs on the GPU (sgl_gather_rows_f32) instead of a CPU
    fancy-index + PCIe copy per hop (reference :58,:60);
  * postprocess(): softmax -> propagate -> aggregate without leaving the device (reference :44-47 goes through numpy).
The de-facto attribute interface used by the reference's tasks (`_pre_graph_op`, `_pre_msg_op`, `_base_model`,
`_processed_feat_list`, `_processed_feature`, `_pre_msg_learnable`, ...) is kept."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import config
from .. import device as dev
from .simple_models import IdenticalMapping

_LEARNABLE = ("proj_concat", "learnable_weighted", "iterate_learnable_weighted")


def take_rows(feat, idx, device):
    """feat[idx].to(device) -- on the GPU when the hop matrix lives there.  A contiguous `range` of rows of a device matrix (a full
    prediction pass: tricks.label_reuse.predict_all) is a VIEW of it, not a copy: what follows only reads it (the aggregator
    kernels and the head's GEMM take a row pitch), and a copy would move every hop matrix once more (2 x 1.6 GB per hop at the
    products shape)."""
    if torch.is_tensor(feat) and feat.is_cuda:
        if isinstance(idx, range) and idx.step == 1 and 0 <= idx.start <= idx.stop <= feat.shape[0]:
            return feat[idx.start:idx.stop].to(device)
        return dev.gather_rows(feat, idx).to(device)
    if isinstance(idx, range):
        idx = list(idx)
    return feat[idx].to(device)


def take_hop_rows(feats, idx, device):
    """[take_rows(feat, idx, device) for feat in feats] (reference: models/base_model.py:58-60) -- for device-resident hop matrices
    the indices are validated and uploaded once for all of them (device.gather_hops)"""
    feats = list(feats)
    same = len(feats) > 1 and all(torch.is_tensor(f) and f.is_cuda and f.dtype == torch.float32 and f.dim() == 2
                                  and f.shape[0] == feats[0].shape[0] and f.device == feats[0].device for f in feats)
    contiguous_range = isinstance(idx, range) and idx.step == 1
    if not same or contiguous_range:
        return [take_rows(feat, idx, device) for feat in feats]
    return [t.to(device) for t in dev.gather_hops(feats, idx)]


class BaseSGAPModel(nn.Module):
    def __init__(self, prop_steps, feat_dim, output_dim):
        super(BaseSGAPModel, self).__init__()
        self._prop_steps = prop_steps
        self._feat_dim = feat_dim
        self._output_dim = output_dim

        self._pre_graph_op, self._pre_msg_op = None, None
        self._post_graph_op, self._post_msg_op = None, None
        self._base_model = None

        self._processed_feat_list = None
        self._processed_feature = None
        self._pre_msg_learnable = False

    # `_processed_feat_list` is part of the de-facto interface (the reference's distributed tasks and search code read it:
    # sgl/tasks/node_classification_dist.py:69, sgl/search/auto_search_dist.py:80).  When preprocess() folded the aggregation into
    # the SpMM epilogue no hop list was kept; it is then produced ON DEMAND, the first time somebody asks for it.
    # A read must not silently trigger K SpMMs over matrices that were folded precisely because the K+1 hop matrices would not fit:
    # the lazy path only runs when they take at most half of the free device memory.  Otherwise the read gives None -- what the
    # reference's plain attribute holds before preprocess(), so hasattr(), inspect.getmembers() and `is None` probes behave -- and
    # says so ONCE in a warning that names the way out; materialize_hops() is the explicit request and raises instead, and
    # hops_available() is the query that never computes anything.
    @property
    def _processed_feat_list(self):
        hops = self.__dict__.get("_hop_list")
        if hops is None and self.__dict__.get("_hop_source") is not None:
            if not self._hops_fit():
                if not self.__dict__.get("_hop_warned"):
                    import warnings
                    self.__dict__["_hop_warned"] = True
                    warnings.warn("_processed_feat_list: the aggregation of the last preprocess() was folded into the propagation and the "
                                  "K+1 hop matrices would take more than half of the free device memory (or the device could not report "
                                  "it), so none were produced and the attribute reads as None; call model.materialize_hops(force=True) "
                                  "to propagate them anyway, or set sgl_amd.config.fuse_aggregate = False before preprocess()",
                                  RuntimeWarning, stacklevel=2)
                return None
            hops = self.materialize_hops()
        return hops

    def hops_available(self):
        """'kept' (the hop list exists), 'lazy' (folded preprocess: a read of _processed_feat_list would propagate them now and
        they fit), 'too_large' (folded, and they would not fit: the attribute reads as None, materialize_hops(force=True)
        overrides) or 'none' (preprocess() has not run / no graph operator).  Never computes anything."""
        if self.__dict__.get("_hop_list") is not None:
            return "kept"
        if self.__dict__.get("_hop_source") is None:
            return "none"
        return "lazy" if self._hops_fit() else "too_large"

    def materialize_hops(self, force=False):
        """The hop list [X, A_hat X, ..., A_hat^K X] of the last preprocess() when its aggregation was folded into the SpMM
        epilogue (no list was kept): propagated now, once, and kept.  Raises if the K+1 matrices would take more than half of
        the free device memory unless force=True.  Returns the list (None if preprocess() has not run)."""
        hops = self.__dict__.get("_hop_list")
        src = self.__dict__.get("_hop_source")
        if hops is None and src is not None:
            if not force and not self._hops_fit():
                raise RuntimeError("the K+1 hop matrices of the folded pre-propagation (_processed_feat_list) would take more than half "
                                   "of the free device memory (or the device could not report it): call "
                                   "model.materialize_hops(force=True) to propagate them anyway, or set sgl_amd.config.fuse_aggregate "
                                   "= False before preprocess() to keep the hop list")
            hops = self._pre_graph_op.propagate(*src)
            self.__dict__["_hop_list"], self.__dict__["_hop_source"] = hops, None
        return hops

    def _hops_fit(self):
        src = self.__dict__.get("_hop_source")
        try:
            n, d = src[1].shape
            need = (self._pre_graph_op._prop_steps + 1) * n * dev.row_pitch(d) * 4
            free, _ = torch.cuda.mem_get_info(torch.device(self._pre_graph_op._opt("device")))
            return need <= free // 2
        except (ValueError, TypeError, AttributeError):
            return True                                   # no shape to ask about (nothing folded on a device): nothing to refuse
        except (RuntimeError, AssertionError):
            return False                                  # the device cannot say how much memory is free: refuse, explicitly

    # the inputs of a folded preprocess() (adjacency, features) are kept only to serve the lazy hop list: they are not part of the
    # model and do not travel with torch.save(model) / copy.deepcopy(model) (the reference's search code pickles whole models)
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_hop_source"] = None
        return state

    @_processed_feat_list.setter
    def _processed_feat_list(self, hops):
        self.__dict__["_hop_list"], self.__dict__["_hop_source"] = hops, None

    def preprocess(self, adj, feature):
        if self._pre_graph_op is None:
            self._pre_msg_learnable = False
            self._processed_feature = feature
            return
        self._pre_msg_learnable = self._pre_msg_op.aggr_type in _LEARNABLE
        gop = self._pre_graph_op
        if (config.fuse_aggregate is not False and not self._pre_msg_learnable and hasattr(gop, "propagate_reduce")
                and hasattr(self._pre_msg_op, "fused_spec") and not gop._opt("host_output")):
            # last / sum / mean / max / min / simple_weighted: accumulated in the SpMM epilogue, the K+1 hop matrices never coexist
            spec = self._pre_msg_op.fused_spec(gop._prop_steps + 1)
            if spec is not None and (spec["kind"] == "last" or config.fuse_aggregate is True or self._hops_are_heavy(feature)):
                with torch.no_grad():
                    fused = gop.propagate_reduce(adj, feature, **spec)
                if fused is not None:
                    self._processed_feat_list = None
                    self.__dict__["_hop_source"] = (adj, feature)     # the hop list stays available, lazily
                    self._processed_feature = fused
                    return
        self._processed_feat_list = self._pre_graph_op.propagate(adj, feature)
        if not self._pre_msg_learnable:
            with torch.no_grad():
                self._processed_feature = self._pre_msg_op.aggregate(self._processed_feat_list)

    def _hops_are_heavy(self, feature):
        """fuse_aggregate = "auto": would the K+1 hop matrices take more than a quarter of the free device memory?"""
        try:
            n, d = feature.shape
            need = (self._pre_graph_op._prop_steps + 1) * n * dev.row_pitch(d) * 4
            free, _ = torch.cuda.mem_get_info(torch.device(self._pre_graph_op._opt("device")))
            return need > free // 4
        except (RuntimeError, AssertionError, ValueError):   # no usable device to ask: keep the hop list (the small-job default)
            return False

    def postprocess(self, adj, output):
        if self._post_graph_op is None:
            return output
        if self._post_msg_op.aggr_type in _LEARNABLE:
            raise ValueError(
                "Learnable weighted message operator is not supported in the post-processing phase!")
        probs = F.softmax(output, dim=1).detach()
        strict = self._post_graph_op._opt("strict_types") if hasattr(self._post_graph_op, "_opt") else config.strict_types
        hops = self._post_graph_op.propagate(adj, probs.cpu().numpy() if strict else probs)
        with torch.no_grad():
            return self._post_msg_op.aggregate(hops)

    # a wrapper of the forward function
    def model_forward(self, idx, device):
        return self.forward(idx, device)

    def forward(self, idx, device):
        if self._pre_msg_learnable:
            hop_rows = take_hop_rows(self._processed_feat_list, idx, device)
            processed_feature = self._pre_msg_op.aggregate(hop_rows)
        else:
            feat = self._processed_feature
            if isinstance(feat, np.ndarray):  # no pre-graph-op: the raw feature matrix
                feat = torch.from_numpy(feat)
            processed_feature = take_rows(feat, idx, device)
            if isinstance(self._base_model, IdenticalMapping) and torch.is_tensor(feat) and \
                    processed_feature.untyped_storage().data_ptr() == feat.untyped_storage().data_ptr():
                processed_feature = processed_feature.clone()     # no head (NAFS): never hand out a view of the stored matrix
        return self._base_model(processed_feature)
