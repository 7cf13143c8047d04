"""Pinned, pre-faulted host buffers for results the reference contract wants on the CPU (GraphOp.propagate with
host_output=True returns CPU FloatTensors like sgl/operators/base_op.py:36).

A fresh 1 GB destination costs more to fault in than to fill over PCIe (profiles/r02_e2e_host_propagate.log: 0.24 s for the
three hop matrices of the products-sized job, of which ~0.05 s is the transfer).  The pool keeps page-locked buffers across
calls and hands one out again ONLY when nothing derived from it is alive any more -- no tensor, view or numpy array: the
storage's reference count says so -- so every call still returns tensors that nobody else writes to, exactly like freshly
allocated ones.  A caller that keeps the results of earlier calls simply makes the pool grow (up to its cap); beyond the cap,
and wherever the count cannot be read, plain pageable tensors are returned through the staged sgl_download path."""
import os
import threading

import torch

# Page-locked memory is a machine-wide resource: the default cap is 4 GB (the three hop matrices of a products-sized k = 3 call are
# 2.9 GB); a job that returns more per call raises it with SGL_HOST_POOL_GB.  Buffers nobody references any more that did not fit
# the last request are released by trim(), which GraphOp.propagate calls when a host-output call ends.
_CAP_BYTES = int(float(os.environ.get("SGL_HOST_POOL_GB", "4")) * (1 << 30))
_ROUND = 2 << 20

_lock = threading.Lock()
_buckets = {}          # rounded byte size -> [UntypedStorage, ...]
_pooled_bytes = 0
stats = {"reused": 0, "allocated": 0, "declined": 0}


_USE_COUNT = getattr(torch._C, "_storage_Use_Count", None)      # private API, looked up once: without it the pool declines everything


def _use_count(storage):
    return None if _USE_COUNT is None else int(_USE_COUNT(storage._cdata))


def take(shape, dtype=torch.float32, pinned=True):
    """A (pinned) CPU tensor of `shape` that no one else references, or None (pool disabled / full / no pinned memory).
    pinned=False pools ordinary pre-faulted memory (what the CPU tests exercise the recycling rule with)."""
    global _pooled_bytes
    if _CAP_BYTES <= 0 or (pinned and not torch.cuda.is_available()):
        return None
    n = 1
    for s in shape:
        n *= int(s)
    nbytes = n * torch.empty(0, dtype=dtype).element_size()
    if nbytes == 0:
        return None
    size = (nbytes + _ROUND - 1) // _ROUND * _ROUND + (0 if pinned else 1)      # pinned and plain buffers never share a bucket
    with _lock:
        for st in _buckets.get(size, ()):
            c = _use_count(st)
            if c is None:
                stats["declined"] += 1
                return None
            if c == 1:                                   # only the pool's own handle: nothing derived from it is alive
                stats["reused"] += 1
                return torch.empty(0, dtype=dtype).set_(st, 0, tuple(int(s) for s in shape))
        if _pooled_bytes + size > _CAP_BYTES or _use_count(torch.empty(1).untyped_storage()) is None:
            stats["declined"] += 1
            return None
        try:
            st = torch.empty(size, dtype=torch.uint8, pin_memory=pinned).untyped_storage()
        except RuntimeError:
            stats["declined"] += 1
            return None
        _buckets.setdefault(size, []).append(st)
        _pooled_bytes += size
        stats["allocated"] += 1
        return torch.empty(0, dtype=dtype).set_(st, 0, tuple(int(s) for s in shape))


def trim(keep_sizes=(), only_above=0):
    """release every pooled buffer that is not in use (except buckets whose rounded size is in keep_sizes: what the call that just
    ended used -- the next call of the same shape finds them warm).  only_above: do nothing while the pool holds no more than
    this many bytes (a caller alternating between two shapes should not re-pin its buffers on every call)"""
    global _pooled_bytes
    with _lock:
        if _pooled_bytes <= only_above:
            return
        for size, lst in list(_buckets.items()):
            if size in keep_sizes:
                continue
            keep = [st for st in lst if (_use_count(st) or 2) > 1]
            _pooled_bytes -= size * (len(lst) - len(keep))
            _buckets[size] = keep


def bucket_size(shape, dtype=torch.float32, pinned=True):
    """the bucket a request of this shape falls into (for trim(keep_sizes=...))"""
    n = 1
    for s in shape:
        n *= int(s)
    nbytes = n * torch.empty(0, dtype=dtype).element_size()
    return (nbytes + _ROUND - 1) // _ROUND * _ROUND + (0 if pinned else 1)
