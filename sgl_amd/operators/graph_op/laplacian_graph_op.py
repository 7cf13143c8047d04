"""LaplacianGraphOp: A_hat = D^{r-1} (A + I)^T D^{-r}   (reference: sgl/operators/graph_op/laplacian_graph_op.py:7-19)"""
import scipy.sparse as sp

from ..base_op import GraphOp


class LaplacianGraphOp(GraphOp):
    def __init__(self, prop_steps, r=0.5, **kwargs):
        super(LaplacianGraphOp, self).__init__(prop_steps, **kwargs)
        self.__r = r

    def _norm_params(self):
        return self.__r, None

    def _construct_adj(self, adj):
        from ...io import DeviceAdjacency
        if not isinstance(adj, (sp.csr_matrix, sp.coo_matrix, DeviceAdjacency)):
            raise TypeError("The adjacency matrix must be a scipy.sparse.coo_matrix/csr_matrix!")
        return self._device_adj(adj)
