from .gamlp import GAMLP
from .gamlp_recursive import GAMLPRecursive
from .gbp import GBP
from .nafs import NAFS
from .pasca_v1 import PASCA_V1
from .pasca_v2 import PASCA_V2
from .pasca_v3 import PASCA_V3
from .sgc import SGC
from .sign import SIGN
from .ssgc import SSGC

__all__ = ["SGC", "SSGC", "SIGN", "GBP", "GAMLP", "GAMLPRecursive", "NAFS", "PASCA_V1", "PASCA_V2", "PASCA_V3"]
