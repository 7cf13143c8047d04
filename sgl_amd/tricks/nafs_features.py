"""NAFS feature smoothing for the NAFS clustering / link-prediction tasks, on the MI355X.

Reference: NodeClusteringNAFS._k_hop_cluster (sgl/tasks/node_clustering.py:205-258) and its twin in
tasks/link_prediction.py:233-284: for every r in r_list normalise the adjacency (D^{r-1}(A+I)^T D^{-r}), propagate
`hops` times with torch.spmm, weight the hops per node by softmax(cosine similarity to hop 0) -- an O(N * hops)
Python loop in the reference -- and finally ensemble the per-r results (mean / max / concat; 'simple' = plain
hops-step propagation with the first r).  Here: device normalisation + k HIP SpMMs + the fused NAFS kernel per r,
and one streaming reduction kernel for the ensemble; nothing leaves HBM."""
import scipy.sparse as sp
import torch

from .. import _lib
from .. import device as dev
from ..operators.utils import adj_to_symmetric_norm_device

_METHODS = ("mean", "max", "concat", "simple")


@torch.no_grad()
def nafs_ensemble_features(adj, x, hops, r_list=(0.5, 0.4, 0.3, 0.2, 0.1, 0), method="mean", device="cuda",
                           strict_order=False):
    """adj: scipy sparse adjacency (un-normalised); x: [N, d] ndarray / tensor.  Returns a CUDA tensor:
    [N, d] for mean / max / simple, [N, len(r_list) * d] for concat."""
    method = method.lower()
    if method not in _METHODS:
        raise ValueError("Method not Suppoted! Choose 'mean', 'max' or 'concat' !")
    if not sp.issparse(adj):
        raise TypeError("adj must be a scipy sparse matrix")
    _lib.require_gpu()
    device = torch.device(device)
    x0 = dev.upload_rows(x, device)
    per_r = []
    for r in r_list:
        rowptr, col, val = adj_to_symmetric_norm_device(adj, r, None, device=device)
        csr = dev.DeviceCSR(rowptr, col, val, adj.shape, strict=strict_order)
        feats = [x0]
        for _ in range(hops):
            y = csr.spmm(dev.padded_parent(feats[-1]))
            feats.append(y[:, :x0.shape[1]] if y.shape[1] != x0.shape[1] else y)
        if method == "simple":
            return feats[-1]
        per_r.append(dev.nafs_aggregate(feats))
        del csr, feats
    if method == "mean":
        return dev.hop_reduce(_lib.SGL_REDUCE_MEAN, per_r)
    if method == "max":
        return dev.hop_reduce(_lib.SGL_REDUCE_MAX, per_r)
    return dev.hop_concat(per_r)
