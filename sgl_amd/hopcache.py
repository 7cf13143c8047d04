"""On-disk cache of propagated hop matrices ("checkpoint / resume": optional).

The reference never caches its propagated features: every run of a task, every seed of a sweep and every trial of a search
repeats the same k SpMMs on the same graph and features (sgl/tasks/node_classification.py:34-38,
sgl/tasks/node_classification_with_label_use.py:79,104).  `GraphOp(hop_cache_dir=...)` (or SGL_AMD_HOP_CACHE) keeps the result
of propagate() under a key made of the CONTENT of the adjacency and of the features (full hashes: sgl_content_hash for host
arrays, a position-weighted wrapping sum of the raw bits for device tensors) and of everything that shapes the result
(operator class, r, alpha, prop_steps, strict_order, library version).  A hit loads the hop matrices straight to the device;
anything unexpected (partial directory, shape mismatch) is a miss.  Files: <dir>/<key>/hop_<k>.npy + meta.json (written last)."""
import json
import os
import uuid

import numpy as np
import torch

from . import _lib


def _tensor_digest(t):
    """order-sensitive 64-bit digest of a float tensor's raw bits"""
    flat = t.contiguous().view(-1)
    if flat.dtype != torch.float32:
        flat = flat.float()
    return _bits_digest(flat.view(torch.int32))


def _bits_digest(bits):
    """order-sensitive 64-bit digest of an int32 tensor (wrapping int64 arithmetic on the device, chunked)"""
    bits = bits.contiguous().view(-1)
    total = torch.zeros((), dtype=torch.int64, device=bits.device)
    step = 1 << 26
    for s in range(0, bits.numel(), step):
        part = bits[s:s + step].to(torch.int64)
        pos = torch.arange(s, s + part.numel(), dtype=torch.int64, device=bits.device)
        total += ((part ^ (pos * -7046029254386353131)) * 1099511628211).sum()
    return int(total.item()) & 0xFFFFFFFFFFFFFFFF


def content_key(obj):
    """hashable description of the CONTENT of an adjacency / feature argument of propagate()"""
    import scipy.sparse as sp
    if sp.issparse(obj):
        parts = [(nm, str(np.asarray(getattr(obj, nm)).dtype), _lib.content_hash(np.asarray(getattr(obj, nm))))
                 for nm in ("indptr", "indices", "data", "row", "col", "offsets") if hasattr(obj, nm)]
        return ("scipy", obj.format, tuple(obj.shape), tuple(parts))
    if isinstance(obj, np.ndarray):
        return ("ndarray", tuple(obj.shape), str(obj.dtype), _lib.content_hash(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            return ("tensor", tuple(obj.shape), str(obj.dtype), _tensor_digest(obj))
        return ("tensor", tuple(obj.shape), str(obj.dtype), _lib.content_hash(obj.detach().contiguous().numpy()))
    if hasattr(obj, "rowptr") and hasattr(obj, "col") and hasattr(obj, "val"):     # DeviceAdjacency
        rp32 = obj.rowptr.to(torch.int64).contiguous().view(torch.int32)           # raw bits, two words per pointer
        return ("device_adj", tuple(obj.shape), _bits_digest(rp32), _bits_digest(obj.col.to(torch.int32)), _tensor_digest(obj.val))
    raise TypeError(f"cannot fingerprint {type(obj).__name__} for the hop cache")


class HopCache:
    def __init__(self, directory):
        self.dir = str(directory)
        self.hits = self.misses = 0

    def key(self, op_desc, adj, feature):
        import hashlib
        text = repr((op_desc, _lib.lib().sgl_version(), content_key(adj), content_key(feature)))
        return hashlib.blake2b(text.encode(), digest_size=16).hexdigest()

    def load(self, key, n_hops, device):
        d = os.path.join(self.dir, key)
        try:
            with open(os.path.join(d, "meta.json")) as f:
                meta = json.load(f)
            if meta.get("n_hops") != n_hops:
                raise ValueError("other hop count")
            from . import device as dev
            hops = []
            for k in range(n_hops):
                a = np.load(os.path.join(d, f"hop_{k}.npy"))
                if list(a.shape) != meta["shape"] or a.dtype != np.float32:
                    raise ValueError("unexpected array")
                hops.append(dev.upload_rows(a, device))
        except (OSError, ValueError, KeyError, json.JSONDecodeError):
            self.misses += 1
            return None
        self.hits += 1
        return hops

    def save(self, key, hops):
        d = os.path.join(self.dir, key)
        os.makedirs(d, exist_ok=True)
        from . import device as dev
        tag = f"{os.getpid()}.{uuid.uuid4().hex[:8]}"
        for k, h in enumerate(hops):
            a = (dev.download_rows(h) if h.is_cuda else h.detach()).contiguous().numpy()
            tmp = os.path.join(d, f".hop_{k}.{tag}.tmp.npy")     # per-writer names: ranks of one job may save the same key at once
            np.save(tmp, a)
            os.replace(tmp, os.path.join(d, f"hop_{k}.npy"))
        tmp = os.path.join(d, f".meta.{tag}.tmp")
        with open(tmp, "w") as f:
            json.dump({"n_hops": len(hops), "shape": list(hops[0].shape)}, f)
        os.replace(tmp, os.path.join(d, "meta.json"))        # written last: a directory without it is a miss
