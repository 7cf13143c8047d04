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


class SharedHops:
    """Device-resident hop matrices shared between operators, process-wide (sgl_amd.config.share_hops; off by default).

    A PaSca-style search builds a fresh model per trial -- 4 graph operators x 11 message operators, every trial re-propagating the
    same graph and features (sgl/search/search_models.py:19-46, search_config.py:14-15).  With the store on, GraphOp.propagate
    (i) returns the hop list a previous operator with the same parameters produced for the same CONTENT of adjacency and features
    (content_key: full hashes, so an edited matrix or feature array is another key), and (ii) serves a PprGraphOp from the
    LaplacianGraphOp chain of the same r by the triangular mix of sgl_amd.operators.graph_op.ppr_hops_from_laplacian -- propagating
    that Laplacian chain first when nobody has (GraphOp._propagate_or_cache), so that a search pays k SpMMs per (graph, features, r)
    and one mixing pass per alpha (not under strict_order: the mix equals the chain to float32 rounding, not bit for bit).  The lists are SHARED: callers must treat hop
    matrices as read-only (the library's aggregators do).  Least recently used entries leave when the byte budget
    (config.share_hops_gb) is exceeded."""

    def __init__(self):
        self.entries = {}          # (graph_feat_key, cls, r, alpha, strict) -> [hops, last_use]
        self.clock = 0
        self._memo = {}            # id(device object) -> (weakref, buffer marks, content key)
        self.stats = {"hits": 0, "derived": 0, "misses": 0, "evicted": 0}

    def _content(self, obj):
        """content_key, remembered for DEVICE objects while they are alive and untouched (same buffers, same torch version counters:
        every in-place torch write bumps them) -- the digest of a gigabyte on the device costs about as much as a hop"""
        import weakref
        if torch.is_tensor(obj) and obj.is_cuda:
            mark = (obj.data_ptr(), obj._version, tuple(obj.shape), tuple(obj.stride()), str(obj.dtype))
        elif hasattr(obj, "rowptr") and hasattr(obj, "val") and torch.is_tensor(obj.val):
            mark = tuple((t.data_ptr(), t._version, t.numel()) for t in (obj.rowptr, obj.col, obj.val)) + (tuple(obj.shape),)
        else:
            return content_key(obj)                                   # host arrays: hashed in full every time
        held = self._memo.get(id(obj))
        if held is not None and held[0]() is obj and held[1] == mark:
            return held[2]
        key = content_key(obj)
        try:
            self._memo[id(obj)] = (weakref.ref(obj), mark, key)
        except TypeError:
            return key
        if len(self._memo) > 64:
            for q in [q for q, v in self._memo.items() if v[0]() is None]:
                del self._memo[q]
        return key

    def data_key(self, adj, feature, device=None):
        """key of (adjacency content, feature content, target device): hop lists live on ONE device and are handed out as they are,
        so an operator on cuda:1 must not be served a chain that lives on cuda:0"""
        import hashlib
        return hashlib.blake2b(repr((_lib.lib().sgl_version(), self._content(adj), self._content(feature), str(device))).encode(),
                               digest_size=16).hexdigest()

    def _touch(self, k):
        self.clock += 1
        self.entries[k][1] = self.clock
        return list(self.entries[k][0])

    def lookup(self, dkey, cls, r, alpha, K, strict):
        """exact entry with at least K hops, else (PPR, not strict) a Laplacian chain of the same r to mix from, else None"""
        for s_ in ((True,) if strict else (False, True)):          # a strict-order chain also answers a relaxed request
            k = (dkey, cls, float(r), None if alpha is None else float(alpha), s_)
            if k in self.entries and len(self.entries[k][0]) >= K + 1:
                self.stats["hits"] += 1
                return self._touch(k)[:K + 1]
        if alpha is not None and not strict:
            for s_ in (False, True):
                lk = (dkey, "LaplacianGraphOp", float(r), None, s_)
                if lk in self.entries and len(self.entries[lk][0]) >= K + 1:
                    from .operators.graph_op import ppr_hops_from_laplacian
                    hops = ppr_hops_from_laplacian(self._touch(lk)[:K + 1], alpha)
                    self.stats["derived"] += 1
                    self.store(dkey, cls, r, alpha, strict, hops)
                    return list(hops)
        self.stats["misses"] += 1
        return None

    def store(self, dkey, cls, r, alpha, strict, hops):
        from . import config
        k = (dkey, cls, float(r), None if alpha is None else float(alpha), bool(strict))
        self.clock += 1
        self.entries[k] = [list(hops), self.clock]
        budget = float(getattr(config, "share_hops_gb", 64.0)) * (1 << 30)

        def nbytes(e):
            return sum(h.numel() * h.element_size() for h in e[0][1:])
        while len(self.entries) > 1 and sum(nbytes(e) for e in self.entries.values()) > budget:
            old = min(self.entries, key=lambda q: self.entries[q][1])
            if old == k:
                break
            del self.entries[old]
            self.stats["evicted"] += 1

    def clear(self):
        self.entries.clear()
        self._memo.clear()


SHARED = SharedHops()
