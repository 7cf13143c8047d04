from .laplacian_graph_op import LaplacianGraphOp
from .ppr_graph_op import PprGraphOp

__all__ = [
    "LaplacianGraphOp",
    "PprGraphOp",
]
