"""PprGraphOp: (1 - alpha) * A_hat + alpha * I, applied k times -- a lazy random walk, not an iterative PPR solve
(reference: sgl/operators/graph_op/ppr_graph_op.py:7-21)"""
import scipy.sparse as sp

from ..base_op import GraphOp


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
