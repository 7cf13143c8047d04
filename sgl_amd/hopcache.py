"""On-disk cache of propagated hop matrices ("checkpoint / resume": optional).

The reference never caches its propagated features: every run of a task, every seed of a sweep and every trial of a search
repeats the same k SpMMs on the same graph and features (sgl/tasks/node_classification.py:34-38,
sgl/tasks/node_classification_with_label_use.py:79,104).  `GraphOp(hop_cache_dir=...)` (or SGL_AMD_HOP_CACHE) keeps the result
of propagate() under a key made of the CONTENT of the adjacency and of the features (full hashes: sgl_content_hash for host
arrays, a sum of non-linearly mixed (word, position) pairs of the raw bits for device tensors: _bits_digest) and of everything that shapes the result
(operator class, r, alpha, prop_steps, strict_order, library version).  A hit loads the hop matrices straight to the device;
anything unexpected (partial directory, shape mismatch) is a miss.  Files: <dir>/<key>/hop_<k>.npy + meta.json (written last)."""
import json
import os
import uuid

import numpy as np
import torch

from . import _lib


def _tensor_digest(t):
    """order-sensitive 128-bit digest of a float tensor's raw bits"""
    flat = t.contiguous().view(-1)
    if flat.dtype != torch.float32:
        flat = flat.float()
    return _bits_digest(flat.view(torch.int32))


def _s64(v):
    """a 64-bit constant as the signed value torch's int64 arithmetic (which wraps) takes"""
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x, k):
    """logical right shift of an int64 tensor (torch's >> is arithmetic)"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix64(x):
    """splitmix64 finaliser: every input bit reaches every output bit, NOT linear in x (multiply / xor-shift rounds)"""
    x = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _s64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def _bits_digest(bits):
    """Order-sensitive 128-bit digest of an int32 tensor: every (word, position) pair goes through a non-linear 64-bit mixer
    before the (wrapping, order-free, chunkable) sum, twice with independent keys.  A plain position-weighted sum -- what this
    was until round 3 -- is linear: edits at two positions cancel easily (moving a single 1.0 inside a one-hot tensor collided
    in 22 of 3000 trials), and a collision here is a cache HIT that returns another input's hop matrices."""
    bits = bits.contiguous().view(-1)
    total = torch.zeros(2, dtype=torch.int64, device=bits.device)
    step = 1 << 25
    for s in range(0, bits.numel(), step):
        part = bits[s:s + step].to(torch.int64) & 0xFFFFFFFF
        pos = torch.arange(s, s + part.numel(), dtype=torch.int64, device=bits.device)
        keyed = part + pos * _s64(0x9E3779B97F4A7C15)
        total[0] += _mix64(keyed).sum()
        total[1] += _mix64(keyed ^ _s64(0xD6E8FEB86659FD93)).sum()
    lo, hi = (int(v) & 0xFFFFFFFFFFFFFFFF for v in total.tolist())
    return (hi << 64) | lo


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
