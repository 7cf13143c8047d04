"""GraphOp / MessageOp plugin base classes -- same public API as sgl/operators/base_op.py:11-60.

GraphOp.propagate(adj, feature) -> [X, A_hat X, ..., A_hat^K X]; the K SpMMs run as hand-written HIP kernels on
the MI355X with the normalised adjacency and every hop matrix resident in HBM (the reference re-creates
its buffers and, in its cuSPARSE twin, re-uploads them on every hop: cudamatmul.c:57-74,129)."""
import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn
from torch import Tensor

from .. import config
from .. import device as dev


class AdjIdentity:
    """Identity of an adjacency for the normalised-adjacency caches: the cache hits only for the SAME object (held by
    weak reference and compared with `is`, so a recycled id() of a freed temporary can never match) whose contents
    are still the same -- FULL hashes of the row pointers, indices and values (sgl_content_hash: the library's team of host
    threads, about a gigabyte in a few tens of milliseconds, no optional dependency), buffer addresses -- so any in-place edit
    of a cached matrix is noticed."""

    def __init__(self, adj, on_death=None):
        import weakref
        try:
            self._ref = weakref.ref(adj, on_death) if on_death is not None else weakref.ref(adj)
        except TypeError:
            self._ref = None
        self._strong = adj if self._ref is None else None
        self._print = self.fingerprint(adj)

    @classmethod
    def fingerprint(cls, adj):
        if sp.issparse(adj):                # any scipy format (the reference normalises coo / csc input too before it
            from .._lib import content_hash  # rejects it, base_op.py:20-23)

            def whole(a):
                a = np.asarray(a)
                return (a.ctypes.data, a.size, str(a.dtype), content_hash(a))
            parts = [whole(getattr(adj, nm)) for nm in ("indices", "data", "row", "col", "offsets", "indptr") if hasattr(adj, nm)]
            return ("scipy", adj.format, adj.shape, int(adj.nnz), tuple(parts))
        # sgl_amd.io.DeviceAdjacency: device buffers; torch bumps _version on every in-place write
        return ("device", tuple(adj.shape), int(adj.nnz), adj.rowptr.data_ptr(), adj.col.data_ptr(), adj.val.data_ptr(),
                adj.rowptr._version, adj.col._version, adj.val._version)

    def holds(self, adj):
        return (self._ref() if self._ref is not None else self._strong) is adj

    def matches(self, adj):
        held = self._ref() if self._ref is not None else self._strong
        return held is adj and self._print == self.fingerprint(adj)


_GRAPHS = []            # (AdjIdentity, device string, PreparedAdjacency), least recently used first


def _prune_dead(_ref=None):
    """weak-reference callback of a cached matrix: its preparation goes the moment the matrix does"""
    _GRAPHS[:] = [e for e in _GRAPHS if e[0]._ref is None or e[0]._ref() is not None]


def _cache_bytes():
    return sum(e[2].nbytes() for e in _GRAPHS)


def _enforce_budget(keep=None):
    """stay under config.cache_prepared_gb: cached fp64 Laplacians go first, then whole entries, least recently used first"""
    budget = config.cache_prepared_gb * 2 ** 30
    if _cache_bytes() <= budget:
        return
    for e in _GRAPHS:
        e[2].drop_values()
    while len(_GRAPHS) > 0 and _cache_bytes() > budget:
        victim = next((i for i, e in enumerate(_GRAPHS) if e[2] is not keep), None)
        if victim is None:
            break
        del _GRAPHS[victim]


def prepared_graph(adj, device=None):
    """device.PreparedAdjacency of `adj` (scipy sparse matrix or sgl_amd.io.DeviceAdjacency), shared process-wide between all
    operators over the same matrix object with unchanged contents (AdjIdentity: object identity + full content hashes / device
    buffer versions).  A cached entry holds A + I in fp64 and the degrees (12 bytes per non-zero).  The cache is bounded by
    config.cache_prepared_gb: a preparation that alone would exceed it is returned WITHOUT being kept (`prep.cached` False: the
    caller's reference is the only one, so it is freed before the hop matrices are allocated), older entries are evicted to make
    room, and an entry dies with its matrix.  clear_graph_cache() releases everything, config.cache_prepared = False turns the
    sharing off."""
    from .. import _lib
    from ..io import DeviceAdjacency
    from .utils import canonical_csr
    _lib.require_gpu()                                   # no GPU: the library's own error, loudly (there is no CPU path)
    device = torch.device(device or "cuda")
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())     # "cuda" and "cuda:0" are the same place
    key = str(device)
    _prune_dead()
    for i, (ident, dkey, prep) in enumerate(_GRAPHS):
        if dkey == key and ident.matches(adj):
            _GRAPHS.append(_GRAPHS.pop(i))
            return prep
    if isinstance(adj, DeviceAdjacency):
        rowptr, col, val, n = adj.rowptr, adj.col, adj.val, adj.shape[0]
    else:
        csr = canonical_csr(adj)
        if csr.shape[0] != csr.shape[1]:
            raise ValueError("the adjacency matrix must be square")
        n = csr.shape[0]
        rowptr = torch.from_numpy(csr.indptr.astype(np.int64)).to(device)
        col = torch.from_numpy(csr.indices.astype(np.int32)).to(device)
        val = torch.from_numpy(csr.data.astype(np.float32)).to(device)
    # the same object with other contents (edited in place since it was prepared): its old preparation can never be served again
    _GRAPHS[:] = [e for e in _GRAPHS if not (e[1] == key and e[0].holds(adj))]
    prep = dev.PreparedAdjacency(rowptr, col, val, n)
    prep.cached = prep.nbytes() <= config.cache_prepared_gb * 2 ** 30
    if prep.cached:
        _GRAPHS.append((AdjIdentity(adj, on_death=_prune_dead), key, prep))
        _enforce_budget(keep=prep)
    return prep


def clear_graph_cache():
    """release the device copies prepared_graph() keeps"""
    _GRAPHS.clear()
    dev.clear_power_cache()


def _lib_reduce(kind):
    from .. import _lib
    return {"sum": _lib.SGL_REDUCE_SUM, "mean": _lib.SGL_REDUCE_MEAN, "wsum": _lib.SGL_REDUCE_WSUM,
            "max": _lib.SGL_REDUCE_MAX, "min": _lib.SGL_REDUCE_MIN}[kind]


class GraphOp:
    def __init__(self, prop_steps, device=None, host_output=None, strict_types=None, strict_order=None, cache_adj=None,
                 slab_hops=None, reorder=None, hop_cache_dir=None):
        self._prop_steps = prop_steps
        self._reorder = reorder
        self._adj = None
        self._device = device
        self._host_output = host_output
        self._strict_types = strict_types
        self._strict_order = strict_order
        self._cache_adj = cache_adj
        self._slab_hops = slab_hops
        self._hop_cache_dir = hop_cache_dir
        self._hop_cache = None
        self._adj_key = None

    # The cached device adjacency (a library handle + device arrays), its identity record, the side stream and the on-disk cache
    # object belong to THIS process: a pickled / deep-copied operator (torch.save(model), the reference's search code) carries the
    # settings only and rebuilds them on its next propagate().
    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("_adj", "_adj_key", "_hop_cache", "_download_stream", "_prepared", "last_trace", "_delta", "delta_info"):
            if k in state:
                state[k] = None
        return state

    # ---- effective settings (ctor kwarg, else sgl_amd.config) ------------------------------------
    def _opt(self, name):
        v = getattr(self, "_" + name)
        return getattr(config, name) if v is None else v

    def _construct_adj(self, adj):
        raise NotImplementedError

    def _norm_params(self):
        """(r, alpha) of A_hat = D^{r-1}(A+I)^T D^{-r} [PPR: (1-alpha) A_hat + alpha I]; subclasses override"""
        raise NotImplementedError

    def _device_adj(self, adj):
        """normalise `adj` on the GPU (cached per matrix) -> DeviceCSR of A_hat"""
        from ..io import DeviceAdjacency
        from .utils import adj_to_symmetric_norm_device
        r, alpha = self._norm_params()
        reorder = self._opt("reorder") or None
        if reorder not in (None, "community", "auto"):
            raise ValueError("reorder must be None, 'community' or 'auto'")
        params = (r, alpha, bool(self._opt("strict_order")), str(self._opt("device")), reorder)
        if self._opt("cache_adj") and self._adj is not None and self._adj_key is not None:
            ident, cached_params = self._adj_key
            if cached_params == params and ident.matches(adj):
                return self._adj
        if config.cache_prepared:
            # the (r, alpha)-independent part of the normalisation -- the device copy of A, A + I in fp64, the degrees, the symmetry
            # check -- is shared by EVERY operator over the same matrix (prepared_graph below): a PaSca-style search builds a fresh
            # GraphOp per trial (sgl/search/search_models.py:19-46) and pays one scaling pass per trial instead of upload + preparation
            prep = prepared_graph(adj, self._opt("device"))
            rowptr, col, val = prep.normalize(r, alpha)
            if not config.keep_sweep_values:
                prep.drop_values()                 # the fp64 Laplacian a PPR request leaves behind (8 bytes per non-zero)
            elif prep.cached:
                _enforce_budget(keep=prep)
            del prep
        elif isinstance(adj, DeviceAdjacency):   # already on the device (sgl_amd.io ingest): nothing touches the host
            rowptr, col, val = dev.normalize_adj(adj.rowptr, adj.col, adj.val, adj.shape[0], r, alpha)
        else:
            rowptr, col, val = adj_to_symmetric_norm_device(adj, r, alpha, device=self._opt("device"))
        rowmap = None
        if reorder:
            # plan-time locality ordering (sgl_amd/reorder.py): the rows of A_hat are STORED in an order that keeps communities
            # together and processed in that order, so a gathered row of X is re-used while it is still in L2 / the Infinity
            # Cache.  Column ids, X, Y and the order of every row's terms stay the caller's: results are bit-identical.
            # "auto" keeps the order only when it makes the graph measurably more local than its own ids do.
            from ..reorder import plan_rowmap
            rowmap, self.reorder_info = plan_rowmap(rowptr, col, adj.shape[0], reorder)   # rowmap[k] = node processed k-th
            if rowmap is not None:
                rowptr, col, val = dev.permute_rows(rowptr, col, val, rowmap)
        csr = dev.DeviceCSR(rowptr, col, val, adj.shape, strict=bool(self._opt("strict_order")))
        if rowmap is not None:
            csr.set_rowmap(rowmap)
        self._adj_key = (AdjIdentity(adj), params) if self._opt("cache_adj") else None
        return csr

    def _checked(self, adj, feature):
        """_construct_adj BEFORE validation (reference order, base_op.py:20-27), then the reference's exceptions"""
        self._adj = self._construct_adj(adj)
        self._validate_inputs(adj, feature, self._adj.shape[1])

    def _validate_inputs(self, adj, feature, n_cols):
        """the reference's type / shape exceptions (base_op.py:22-27) -- needs no normalised adjacency, only its column count"""
        from ..io import DeviceAdjacency
        if not isinstance(adj, (sp.csr_matrix, DeviceAdjacency)):
            raise TypeError("The adjacency matrix must be a scipy csr sparse matrix!")
        elif not isinstance(feature, np.ndarray) and not (isinstance(feature, Tensor) and not self._opt("strict_types")):
            raise TypeError("The feature matrix must be a numpy.ndarray!")
        elif n_cols != feature.shape[0]:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        if self._opt("strict_types") and feature.dtype != np.float32:
            # the reference's ctypes ndpointer(float32) rejects anything else (operators/utils.py:22-26)
            raise TypeError("The feature matrix must be a float32 numpy.ndarray!")
        if feature.ndim != 2:
            raise ValueError("The feature matrix must be two-dimensional!")

    def _target_device(self):
        dv = torch.device(self._opt("device") or "cuda")
        if dv.type == "cuda" and dv.index is None:
            dv = torch.device("cuda", torch.cuda.current_device())
        return dv

    def _device_features(self, feature):
        """the input features as a [n, d] view of a 16-byte aligned, line-aware-pitch device buffer"""
        device = self._adj.device
        x0 = feature if (isinstance(feature, Tensor) and feature.is_cuda and feature.dtype == torch.float32) else None
        cur = dev.upload_rows(feature, device) if x0 is None else x0
        if (cur.shape[1] > 1 and cur.stride(1) != 1) or cur.data_ptr() % 16 != 0 or \
                (cur.shape[0] > 1 and cur.stride(0) != dev.row_pitch(cur.shape[1])):
            cur = dev.upload_rows(cur, device)  # re-pack into an aligned buffer with the line-aware row pitch
        return cur

    def propagate_reduce(self, adj, feature, kind, start=0, end=None, weights=None, divisor=None):
        """The hop aggregate WITHOUT the hops: `last`, `sum`, `mean`, `max`, `min` or `wsum` (fixed weights) of hops start..end-1 of
        [X, A_hat X, ..., A_hat^K X], accumulated in the SpMM epilogue where each row is produced
        (sgl_spmm_acc_f32).  Same arithmetic and order as aggregate(propagate(...)) with the corresponding MessageOp
        (bit-identical for last / sum / mean / max / min; wsum identical to the HIP aggregator), but no pass over the hop matrices
        and only two hop buffers alive at any time instead of K + 1.  Returns the [n, d] device tensor, or None when the
        hop range is not one this path handles (the caller then uses propagate + aggregate)."""
        K = self._prop_steps
        n_hops = K + 1
        s = 0 if start is None else start
        e = n_hops if end is None else min(end, n_hops)
        if kind == "last":
            s, e = K, n_hops
        if not (isinstance(s, int) and isinstance(e, int) and 0 <= s < e):
            return None
        if kind == "wsum":
            w = [float(v) for v in torch.as_tensor(weights, dtype=torch.float32).reshape(-1)]
            if len(w) != e - s:
                return None
        self._checked(adj, feature)
        cur = self._device_features(feature)
        d = cur.shape[1]
        src = dev.padded_parent(cur) if cur.stride(0) % 4 == 0 else cur
        n = self._adj.shape[0]
        bufs = [dev.padded_parent(dev.alloc_rows(n, d, src.device)) for _ in range(min(2, e - 1))]

        def begin(x_s):           # the aggregate's first term, with the aggregator kernel's own arithmetic
            if kind == "wsum":
                return dev.padded_parent(dev.hop_reduce(_lib_reduce("wsum"), [x_s[:, :d]], torch.tensor(w[:1])))
            return dev.padded_parent(dev.hop_reduce(_lib_reduce(kind if kind in ("max", "min") else "sum"), [x_s[:, :d]]))

        acc = begin(src) if (s == 0 and kind != "last") else None
        x = src
        for h in range(1, e):
            y = bufs[(h - 1) % len(bufs)]
            if acc is not None:
                last = h == e - 1
                div = float(divisor if divisor is not None else (e - s)) if (kind == "mean" and last) else 1.0
                self._adj.spmm_acc(x, y, acc, w=w[h - s] if kind == "wsum" else 1.0, divisor=div,
                                   mode=kind if kind in ("wsum", "max", "min") else "sum")
            else:
                self._adj.spmm(x, out=y)
                if h == s and kind != "last":
                    acc = begin(y)
            x = y
        if kind == "last":
            return x[:, :d] if x.shape[1] != d else x
        if kind == "mean" and e - s == 1:      # a single hop in range: the division has no SpMM to ride on
            acc = acc / torch.tensor(float(divisor if divisor is not None else 1), device=acc.device)   # true division
        return acc[:, :d] if acc.shape[1] != d else acc

    def propagate(self, adj, feature):
        if not config.trace:
            return self._propagate_or_cache(adj, feature)
        # SGL_AMD_TRACE: wall time per phase (the phases mark themselves through self._mark; synchronised at their ends)
        import sys
        import time
        marks = [("start", time.perf_counter())]

        def mark(name):
            torch.cuda.synchronize()
            marks.append((name, time.perf_counter()))
        self._mark = mark
        try:
            out = self._propagate_or_cache(adj, feature)
            mark("output")
        finally:
            self._mark = None
        self.last_trace = {b[0] + "_s": round(b[1] - a[1], 6) for a, b in zip(marks, marks[1:])}
        self.last_trace["total_s"] = round(marks[-1][1] - marks[0][1], 6)
        sys.stderr.write(f"[sgl_amd trace] {type(self).__name__}.propagate: " +
                         " ".join(f"{k}={v * 1e3:.2f}ms" for k, v in self.last_trace.items()) + "\n")
        return out

    _mark = None

    def _phase_done(self, name):
        if self._mark is not None:
            self._mark(name)

    def _propagate_or_cache(self, adj, feature):
        cache_dir = self._opt("hop_cache_dir")
        if config.share_hops and not cache_dir and not self._opt("host_output") and not self._opt("slab_hops"):
            # process-wide store of device-resident hop lists (hopcache.SharedHops): the reference's exceptions first, then the lookup.
            # The normalised adjacency and its SpMM plan are NOT built here: a hit (or a PPR chain mixed from the Laplacian's) never
            # needs them; the real-miss branch below builds them through _propagate
            from .. import _lib
            _lib.require_gpu()
            if not (sp.issparse(adj) or hasattr(adj, "rowptr")):
                self._checked(adj, feature)                         # not a matrix at all: let the reference's own order of errors apply
            else:
                self._validate_inputs(adj, feature, adj.shape[1])
            from ..hopcache import SHARED
            r, alpha = self._norm_params()
            dkey = SHARED.data_key(adj, feature, self._target_device())
            strict = bool(self._opt("strict_order"))
            hops = SHARED.lookup(dkey, type(self).__name__, r, alpha, self._prop_steps, strict)
            if hops is None and alpha is not None and not strict:
                # a PPR chain nobody has yet: propagate the LAPLACIAN chain of this r once (same k SpMMs) and mix -- every other
                # alpha of the search then costs the mixing pass only (hopcache.SharedHops.lookup, ppr_hops_from_laplacian)
                from .graph_op import LaplacianGraphOp
                LaplacianGraphOp(self._prop_steps, r=r, device=self._device, strict_types=self._strict_types, strict_order=False,
                                 cache_adj=self._cache_adj, reorder=self._reorder).propagate(adj, feature)
                hops = SHARED.lookup(dkey, type(self).__name__, r, alpha, self._prop_steps, strict)
            if hops is None:
                hops = self._propagate(adj, feature)                # the real miss: normalise, plan, k SpMMs
                if torch.is_tensor(feature) and hops[0].data_ptr() == feature.data_ptr():
                    hops[0] = hops[0].clone()                  # the caller may edit its tensor later; the stored hop 0 must not follow
                SHARED.store(dkey, type(self).__name__, r, alpha, strict, hops)
            return hops
        if not cache_dir or self._opt("host_output") or self._opt("slab_hops"):
            return self._propagate(adj, feature)
        # on-disk hop cache (sgl_amd/hopcache.py): the reference's exceptions first, then content-keyed lookup
        self._checked(adj, feature)
        from ..hopcache import HopCache
        if self._hop_cache is None or self._hop_cache.dir != str(cache_dir):
            self._hop_cache = HopCache(cache_dir)
        desc = (type(self).__name__, self._norm_params(), self._prop_steps, bool(self._opt("strict_order")))
        key = self._hop_cache.key(desc, adj, feature)
        hops = self._hop_cache.load(key, self._prop_steps + 1, self._adj.device)
        if hops is None:
            hops = self._propagate(adj, feature, checked=True)
            self._hop_cache.save(key, hops)
        return hops

    def _propagate(self, adj, feature, checked=False):
        if not checked:
            self._checked(adj, feature)
        self._phase_done("adjacency")
        device = self._adj.device
        cur = self._device_features(feature)
        self._phase_done("features")
        # the k hops run inside one library call, over the padded width so every d gets 16-byte lanes (pad columns
        # are zeros and stay zeros under propagation)
        d = cur.shape[1]
        K = self._prop_steps
        if self._opt("slab_hops") and d % 4 == 0 and K >= 1 and not self._opt("host_output"):
            # concat-as-layout: hop k is produced directly in column slice k of one [n, (K+1) d] slab (the kernel takes
            # leading dimensions), so ConcatMessageOp over consecutive hops is a view of it -- no copy of any hop
            slab = torch.empty((cur.shape[0], (K + 1) * d), dtype=torch.float32, device=device)
            views = [slab[:, k * d:(k + 1) * d] for k in range(K + 1)]
            views[0].copy_(cur)
            self._adj.spmm_chain(views[0], K, outs=views[1:])
            return views
        src = dev.padded_parent(cur) if cur.stride(0) % 4 == 0 else cur
        if self._opt("host_output"):
            pooled = self._propagate_to_pooled_host(feature, cur, src, d, K)
            if pooled is not None:
                return pooled
        prop_feat_list, sig = (None, None) if self._opt("host_output") else self._delta_propagate(cur, src, d, K)
        if prop_feat_list is None:
            prop_feat_list = [cur] + [y[:, :d] if y.shape[1] != d else y for y in self._adj.spmm_chain(src, self._prop_steps)]
        if sig is not None:
            self._delta_remember(sig, prop_feat_list, src.shape[1])
        self._phase_done("hops")

        if self._opt("host_output"):
            # reference contract: CPU FloatTensors (ordinary pageable memory, like the reference's).  The download runs through
            # the library's pinned staging team (sgl_download): ~43 GB/s instead of ~6 GB/s for a pageable .cpu()
            alias0 = isinstance(feature, np.ndarray) and feature.dtype == np.float32
            out = []
            for i, f in enumerate(prop_feat_list):
                if i == 0 and alias0:
                    out.append(torch.from_numpy(feature))  # the reference's first element aliases the caller's array
                    continue
                out.append(dev.download_rows(f))
            return out
        return prop_feat_list


    # ---- only what changed (config.delta_propagate) -----------------------------------------------------------------------------
    # A_hat . X is separable by columns.  The label-reuse loop (sgl/tasks/node_classification_with_label_use.py:88-104) calls
    # preprocess() again and again on a matrix of which only the last C columns were rewritten; the reference propagates all d + C
    # columns every time.  Here a call leaves behind the per-column content signature of its X and weak references to the hop
    # matrices it returned; the next call re-propagates only the 4-aligned column range covering the columns whose signature moved
    # and copies the others from the previous hop matrices into FRESH ones (nothing the caller holds is ever written).
    delta_info = None

    def _delta_propagate(self, cur, src, d, K):
        """(hop list or None, signature of X or None)"""
        self.delta_info = None
        st, self._delta = self.__dict__.get("_delta"), None
        n, ld = cur.shape[0], src.shape[1]
        if not config.delta_propagate or K < 1 or n * ld * 4 < config.delta_propagate_min_mb * 2 ** 20:
            return None, None
        sig = dev.column_signature(cur)
        if sig is None or st is None:
            return None, sig
        if st["adj"]() is not self._adj or st["shape"] != (n, d, K, ld) or st["sig"].device != sig.device or st["val"] != self._val_token():
            return None, sig                      # another matrix (or the same handle re-weighted), another shape
        old = [w() for w in st["hops"]]
        if any(o is None for o in old) or any(o._version != v for o, v in zip(old, st["versions"])):
            return None, sig                      # the previous hop matrices are gone, or somebody wrote into them
        dw = sig.numel()
        ch = torch.nonzero(sig != st["sig"]).flatten().cpu().numpy()
        c0, c1 = (0, 0) if ch.size == 0 else (int(ch.min()) // 4 * 4, min(dw, (int(ch.max()) // 4 + 1) * 4))
        if c1 - c0 > config.delta_propagate_max_fraction * dw:
            return None, sig
        outs, prev = [], src
        for k in range(K):
            # one contiguous copy of the whole pitched matrix (whole lines, pad columns included), then the slice is overwritten
            y = dev.padded_parent(old[k]).clone(memory_format=torch.contiguous_format)
            if c1 > c0:
                self._adj.spmm(prev[:, c0:c1], out=y[:, c0:c1])
            outs.append(y)
            prev = y
        self.delta_info = {"columns_propagated": (c0, c1), "columns_changed": int(ch.size), "of": d}
        return [cur] + [y[:, :d] if y.shape[1] != d else y for y in outs], sig

    def _val_token(self):
        v = getattr(self._adj, "val", None)
        return (v.data_ptr(), v._version) if torch.is_tensor(v) else None

    def _delta_remember(self, sig, hops, ld):
        import weakref
        try:
            self._delta = {"adj": weakref.ref(self._adj), "val": self._val_token(),
                           "shape": (hops[0].shape[0], hops[0].shape[1], len(hops) - 1, ld), "sig": sig,
                           "hops": [weakref.ref(h) for h in hops[1:]], "versions": [h._version for h in hops[1:]]}
        except TypeError:
            self._delta = None

    def _propagate_to_pooled_host(self, feature, cur, src, d, K):
        """host_output=True with destinations from the pinned pool (sgl_amd/hostpool.py): the hops are launched one by one and
        every finished hop travels to its page-locked destination on a side stream WHILE the next hop is being computed -- the
        call costs about the PCIe time of the hop matrices instead of PCIe + first-touch page faults + compute.  Same kernels,
        same bits as the chained path.  Returns None when the pool has nothing to offer (then the staged path runs)."""
        from .. import hostpool
        n = cur.shape[0]
        alias0 = isinstance(feature, np.ndarray) and feature.dtype == np.float32
        need = K + (0 if alias0 else 1)
        hosts = []
        for _ in range(need):
            h_ = hostpool.take((n, d))
            if h_ is None:
                return None
            hosts.append(h_)
        device = cur.device
        main = torch.cuda.current_stream(device)
        side = getattr(self, "_download_stream", None)
        if side is None or side.device != device:
            side = self._download_stream = torch.cuda.Stream(device=device)
        out, keep = [], []

        def send(t, host):
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                host.copy_(t, non_blocking=True)

        if alias0:
            out.append(torch.from_numpy(feature))        # the reference's first element aliases the caller's array
        else:
            send(cur, hosts[0])
            out.append(hosts[0])
        prev = src
        for h in range(K):
            ypad = torch.empty((n, src.shape[1]), dtype=torch.float32, device=device)
            if src.shape[1] != d:
                ypad[:, d:].zero_()
            self._adj.spmm(prev, out=ypad)
            host = hosts[h + (0 if alias0 else 1)]
            send(ypad[:, :d] if src.shape[1] != d else ypad, host)
            out.append(host)
            keep.append(ypad)
            prev = ypad
        side.synchronize()                                # the results are complete when the call returns (reference contract)
        self._phase_done("hops")                          # (the downloads overlap the hops on this path: one phase)
        # once the pool is more than half full, buffers of other shapes that nobody references go back to the system (this call's
        # bucket stays warm)
        hostpool.trim(keep_sizes=(hostpool.bucket_size((n, d)),), only_above=hostpool._CAP_BYTES // 2)
        return out


class MessageOp(nn.Module):
    def __init__(self, start=None, end=None):
        super(MessageOp, self).__init__()
        self._aggr_type = None
        self._start, self._end = start, end

    @property
    def aggr_type(self):
        return self._aggr_type

    def _combine(self, feat_list):
        return NotImplementedError

    def fused_spec(self, n_hops):
        """keyword arguments for GraphOp.propagate_reduce when this aggregator can be folded into the SpMM epilogue
        (None = it cannot; subclasses override)"""
        return None

    def aggregate(self, feat_list):
        if not isinstance(feat_list, list):
            # reference quirk kept on purpose (base_op.py:55 RETURNS the exception instead of raising it)
            return TypeError("The input must be a list consists of feature matrices!")
        for feat in feat_list:
            if not isinstance(feat, Tensor):
                raise TypeError("The feature matrices must be tensors!")

        return self._combine(feat_list)
