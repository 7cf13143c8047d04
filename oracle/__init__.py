"""CPU oracle for the SGAP pre-propagation path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``sgl_amd/`` imports this package.  Allowed users: ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Contents
--------
spmm_ref.c / liboracle_spmm.so   C restatement of FloatCSRMulDenseOMP (matmul.c:23-40)
_ref/libmatmul.so                the reference's own matmul.c compiled in place (git-ignored)
ref_ops.py                       numpy restatement of the normalisation (operators/utils.py:76-88),
                                 GraphOp.propagate (base_op.py:19-36) and every MessageOp._combine
                                 (operators/message_op/*.py)

Pin status: every function here is checked against golden vectors produced by
importing the reference itself (tests/golden/make_goldens.py -> tests/golden/*.npz)
and, for the SpMM, bit-for-bit against ``_ref/libmatmul.so``.
"""
from .ref_ops import *  # noqa: F401,F403
