"""NAFS feature smoothing for the NAFS clustering / link-prediction tasks, on the MI355X.

Reference: NodeClusteringNAFS._k_hop_cluster (sgl/tasks/node_clustering.py:205-258) and its twin in
tasks/link_prediction.py:233-284: for every r in r_list normalise the adjacency (D^{r-1}(A+I)^T D^{-r}), propagate
`hops` times with torch.spmm, weight the hops per node by softmax(cosine similarity to hop 0) -- an O(N * hops)
Python loop in the reference -- and finally ensemble the per-r results (mean / max / concat; 'simple' = plain
hops-step propagation with the first r).  Here the raw adjacency is uploaded ONCE and prepared ONCE (A + I, degrees,
symmetry: device.PreparedAdjacency); every r then costs the degree powers of the distinct degrees plus one scaling pass
(symmetric graphs: no transposition at all), re-using one SpMM plan (the sparsity structure does not depend on r:
sgl_csr_set_values) and one set of hop buffers; the fused NAFS kernel reads each hop of an r exactly once, and one
streaming reduction kernel forms the ensemble.  Nothing but the first upload crosses PCIe."""
import scipy.sparse as sp
import torch

from .. import _lib
from .. import device as dev
from ..io import DeviceAdjacency

_METHODS = ("mean", "max", "concat", "simple")


@torch.no_grad()
def nafs_ensemble_features(adj, x, hops, r_list=(0.5, 0.4, 0.3, 0.2, 0.1, 0), method="mean", device="cuda",
                           strict_order=False, reorder=None):
    """adj: scipy sparse adjacency (un-normalised); x: [N, d] ndarray / tensor.  Returns a CUDA tensor:
    [N, d] for mean / max / simple, [N, len(r_list) * d] for concat.
    reorder="community": the rows of every A_hat are processed in a plan-time locality order (sgl_amd/reorder.py; found once,
    the structure does not depend on r) -- same results, fewer cache misses on graphs with communities."""
    method = method.lower()
    if method not in _METHODS:
        raise ValueError("Method not Suppoted! Choose 'mean', 'max' or 'concat' !")
    if not (sp.issparse(adj) or isinstance(adj, DeviceAdjacency)):
        raise TypeError("adj must be a scipy sparse matrix (or a DeviceAdjacency already on the GPU)")
    _lib.require_gpu()
    device = torch.device(device)
    dadj = adj if isinstance(adj, DeviceAdjacency) else DeviceAdjacency.from_scipy(adj, device=device)   # the only H2D
    n = dadj.shape[0]
    x0 = dev.upload_rows(x, device)
    d = x0.shape[1]
    hop_bufs = [dev.alloc_rows(n, d, device) for _ in range(hops)]                  # shared by all r
    if reorder not in (None, "community"):
        raise ValueError("reorder must be None or 'community'")
    csr = None
    rowmap = None
    per_r = []
    prep = dev.PreparedAdjacency(dadj.rowptr, dadj.col, dadj.val, n)               # r-independent part, once
    for r in r_list:
        rowptr, col, val = prep.normalize(r, None)
        if reorder and rowmap is None:
            from ..reorder import community_order
            order, _ = community_order(rowptr, col, n)
            rowmap = torch.argsort(order).to(torch.int32)
        if rowmap is not None:
            rowptr, col, val = dev.permute_rows(rowptr, col, val, rowmap)          # rows stored in processing order
        if csr is None:
            csr = dev.DeviceCSR(rowptr, col, val, dadj.shape, strict=strict_order)  # one plan: the structure is r-independent
            if rowmap is not None:
                csr.set_rowmap(rowmap)
        else:
            csr.set_values(val)
        feats = [x0]
        for h in range(hops):
            csr.spmm(dev.padded_parent(feats[-1]), out=dev.padded_parent(hop_bufs[h]))
            feats.append(hop_bufs[h])
        if method == "simple":
            return feats[-1]
        per_r.append(dev.nafs_aggregate(feats))
    if method == "mean":
        return dev.hop_reduce(_lib.SGL_REDUCE_MEAN, per_r)
    if method == "max":
        return dev.hop_reduce(_lib.SGL_REDUCE_MAX, per_r)
    return dev.hop_concat(per_r)


@torch.no_grad()
def nafs_ensemble_sweep(adj, x, hops_list, r_list=(0.5, 0.4, 0.3, 0.2, 0.1, 0), method="mean", device="cuda", strict_order=False,
                        reorder=None, consume=None):
    """The feature matrices NodeClusteringNAFS hands to KMeans for EVERY hop count of a sweep (the task's `hops` argument: an int
    = range(hops), or a list; tasks/node_clustering.py:124-139,176-178 -- and LinkPredictionNAFS likewise, link_prediction.py:
    233-284), from ONE propagation per r.

    The reference calls _k_hop_cluster(hop) per hop count, each re-normalising and re-propagating from X_0 for every r:
    6 x (0 + 1 + ... + 19) = 1 140 SpMMs for hops = 20.  Here every r propagates max(hops_list) steps once (6 x 19 SpMMs) and one
    kernel (sgl_nafs_prefix_f32) emits the aggregate of every requested prefix while streaming the hop matrices once, combining it
    straight into the ensemble over r (mean / max; concat keeps one slab per (hop count, r)).

    Returns {hop count: CUDA tensor} ([N, d]; [N, len(r_list) * d] for concat) -- or, with consume=callable, calls
    consume(hop count, features) in increasing hop order once the ensemble is complete and returns {hop count: its result}, so that
    only one consumer output needs to outlive its feature matrix.  Memory: the max(hops_list) + 1 hop matrices of one r plus one
    matrix per requested hop count (len(r_list) for concat)."""
    method = method.lower()
    if method not in _METHODS:
        raise ValueError("Method not Suppoted! Choose 'mean', 'max' or 'concat' !")
    if not (sp.issparse(adj) or isinstance(adj, DeviceAdjacency)):
        raise TypeError("adj must be a scipy sparse matrix (or a DeviceAdjacency already on the GPU)")
    hops_list = sorted({int(h) for h in (range(hops_list) if isinstance(hops_list, int) else hops_list)})
    if not hops_list or hops_list[0] < 0 or hops_list[-1] >= _lib.SGL_MAX_HOPS:
        raise ValueError(f"hop counts must lie in [0, {_lib.SGL_MAX_HOPS})")
    r_list = list(r_list)
    _lib.require_gpu()
    device = torch.device(device)
    dadj = adj if isinstance(adj, DeviceAdjacency) else DeviceAdjacency.from_scipy(adj, device=device)   # the only H2D
    n = dadj.shape[0]
    x0 = dev.upload_rows(x, device)
    d = x0.shape[1]
    kmax = hops_list[-1]
    hop_bufs = [dev.alloc_rows(n, d, device) for _ in range(kmax)]                  # shared by all r
    if reorder not in (None, "community"):
        raise ValueError("reorder must be None or 'community'")
    csr = rowmap = None
    prep = dev.PreparedAdjacency(dadj.rowptr, dadj.col, dadj.val, n)               # r-independent part, once
    ens = None                                                                     # mean / max: one accumulator per hop count
    slabs = {h: [] for h in hops_list}                                             # concat: per hop count, one matrix per r
    for ri, r in enumerate(r_list):
        rowptr, col, val = prep.normalize(r, None)
        if reorder and rowmap is None:
            from ..reorder import community_order
            order, _ = community_order(rowptr, col, n)
            rowmap = torch.argsort(order).to(torch.int32)
        if rowmap is not None:
            rowptr, col, val = dev.permute_rows(rowptr, col, val, rowmap)
        if csr is None:
            csr = dev.DeviceCSR(rowptr, col, val, dadj.shape, strict=strict_order)  # one plan: the structure is r-independent
            if rowmap is not None:
                csr.set_rowmap(rowmap)
        else:
            csr.set_values(val)
        feats = [x0]
        for h in range(kmax):
            csr.spmm(dev.padded_parent(feats[-1]), out=dev.padded_parent(hop_bufs[h]))
            feats.append(hop_bufs[h])
        if method == "simple":                                                     # plain propagation with the first r only
            ens = [feats[h] for h in hops_list]
            break
        if d > 512:
            # rows beyond the sweep kernel's register layout: the per-prefix kernel, one call per hop count (still one propagation)
            outs = [dev.nafs_aggregate(feats[:h + 1]) for h in hops_list]
            if method == "concat":
                for h, o in zip(hops_list, outs):
                    slabs[h].append(o)
            elif ens is None:
                ens = outs
            else:
                kind = _lib.SGL_REDUCE_SUM if method == "mean" else _lib.SGL_REDUCE_MAX
                ens = [dev.hop_reduce(kind, [e, o]) for e, o in zip(ens, outs)]
                if method == "mean" and ri == len(r_list) - 1:
                    ens = [e / float(len(r_list)) for e in ens]
            continue
        if method == "concat" and d % 4 == 0:
            # one [N, R d] slab per hop count; the kernel writes r's columns of every slab directly (16-byte aligned slices)
            if ens is None:
                ens = [torch.empty((n, len(r_list) * d), dtype=torch.float32, device=device) for _ in hops_list]
            dev.nafs_prefix(feats, hops_list, outs=[e[:, ri * d:(ri + 1) * d] for e in ens])
        elif method == "concat":
            for h, o in zip(hops_list, dev.nafs_prefix(feats, hops_list)):
                slabs[h].append(o)
        elif ens is None:
            ens = dev.nafs_prefix(feats, hops_list)                                # first r: 0 + f = f (Python's sum() starts at 0)
            if method == "mean" and len(r_list) == 1:
                pass                                                               # f / 1
        else:
            last = ri == len(r_list) - 1
            combine = dev.NAFS_MAX if method == "max" else (dev.NAFS_ADD_DIV if last else dev.NAFS_ADD)
            dev.nafs_prefix(feats, hops_list, outs=ens, combine=combine, divisor=float(len(r_list)), outs_padded=True)
    results = {}
    for k, h in enumerate(hops_list):
        f = dev.hop_concat(slabs.pop(h)) if (method == "concat" and ens is None) else ens[k]
        results[h] = consume(h, f) if consume is not None else f
        if consume is not None and ens is not None:
            ens[k] = None
    return results
