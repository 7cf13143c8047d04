"""GraphOp plugins (API of sgl.operators.graph_op): the two normalised-adjacency propagators of SGL, on device."""
from .ppr_graph_op import PprGraphOp, ppr_hops_from_laplacian  # (1 - alpha) A_hat + alpha I, applied k times
from .laplacian_graph_op import LaplacianGraphOp  # A_hat = D^{r-1} (A + I)^T D^{-r}

__all__ = ("LaplacianGraphOp", "PprGraphOp", "ppr_hops_from_laplacian")
