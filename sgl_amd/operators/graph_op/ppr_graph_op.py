"""PprGraphOp: (1 - alpha) * A_hat + alpha * I, applied k times -- a lazy random walk, not an iterative PPR solve
(reference: sgl/operators/graph_op/ppr_graph_op.py:7-21)"""
from math import comb

import numpy as np
import scipy.sparse as sp
import torch

from ..base_op import GraphOp


def ppr_hops_from_laplacian(lap_hops, alpha):
    """The K + 1 hop matrices PprGraphOp(K, r, alpha).propagate would return, from the hop matrices [X, A_hat X, ..., A_hat^K X] of
    LaplacianGraphOp(K, r) over the same graph and features -- WITHOUT another propagation.

    ((1 - alpha) A_hat + alpha I)^k X = sum_j C(k, j) (1 - alpha)^j alpha^(k - j) A_hat^j X  (ppr_graph_op.py:20 applied k times,
    base_op.py:29-35), so every alpha of a PaSca-style sweep over graph operators (sgl/search/search_config.py:14-15) follows from
    ONE propagation chain by a triangular mix of its hop matrices: K + 1 matrix streams read, K written (sgl_hop_lincomb_f32), instead
    of K more SpMMs per alpha.  All weights are positive and sum to 1: the result differs from the reference's own chain by float32
    rounding only (4e-7 ... 6e-7 against the goldens recorded from the reference, tolerance 1e-5); it is not bit-identical, so a
    strict_order caller propagates instead.  Hop 0 is returned as it is (the same object)."""
    from ... import _lib
    from ... import device as dev
    K = len(lap_hops) - 1
    if K < 0:
        raise ValueError("empty hop list")
    if not (0.0 <= float(alpha) <= 1.0):
        raise ValueError("alpha must lie in [0, 1]")
    a = float(alpha)
    hops = [lap_hops[0]]
    if K == 0:
        return hops
    feats = [h if h.is_cuda else dev.upload_rows(h, "cuda") for h in lap_hops]
    w = np.zeros((K, K + 1), dtype=np.float64)
    for k in range(1, K + 1):
        for j in range(k + 1):
            w[k - 1, j] = comb(k, j) * (1.0 - a) ** j * a ** (k - j)
    first = min(K, 15)                                   # outputs whose inputs fit one pass (16 matrices)
    hops += dev.hop_lincomb(feats[:first + 1], w[:first, :first + 1])
    for k in range(first + 1, K + 1):                    # deeper hops: one weighted sum each over k + 1 matrices
        hops.append(dev.hop_reduce(_lib.SGL_REDUCE_WSUM, feats[:k + 1], torch.from_numpy(w[k - 1, :k + 1].astype(np.float32))))
    return hops


class PprGraphOp(GraphOp):
    def __init__(self, prop_steps, r=0.5, alpha=0.15, **kwargs):
        super(PprGraphOp, self).__init__(prop_steps, **kwargs)
        self.__r = r
        self.__alpha = alpha

    def _norm_params(self):
        return self.__r, self.__alpha

    def _construct_adj(self, adj):
        from ...io import DeviceAdjacency
        if not isinstance(adj, (sp.csr_matrix, sp.coo_matrix, DeviceAdjacency)):
            raise TypeError("The adjacency matrix must be a scipy.sparse.coo_matrix/csr_matrix!")
        return self._device_adj(adj)

    def propagate_from_laplacian(self, lap_hops):
        """this operator's hop matrices from those of LaplacianGraphOp(prop_steps, r) over the same graph and features
        (ppr_hops_from_laplacian): no propagation, one mixing pass.  Refused under strict_order (not bit-identical to the chain)."""
        if bool(self._opt("strict_order")):
            raise ValueError("propagate_from_laplacian is not bit-identical to the propagation chain: not with strict_order")
        if len(lap_hops) != self._prop_steps + 1:
            raise ValueError("the Laplacian hop list must hold prop_steps + 1 matrices")
        return ppr_hops_from_laplacian(lap_hops, self.__alpha)
