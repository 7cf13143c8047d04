"""SGL's homogeneous SGAP models on the MI355X path (same class names / signatures as sgl.models.homo)."""
from .zoo import GAMLP, GBP, NAFS, PASCA_V1, PASCA_V2, PASCA_V3, SGC, SIGN, SSGC, GAMLPRecursive

__all__ = ["GAMLP", "GAMLPRecursive", "GBP", "NAFS", "PASCA_V1", "PASCA_V2", "PASCA_V3", "SGC", "SIGN", "SSGC"]
