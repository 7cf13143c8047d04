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
