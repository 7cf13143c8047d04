"""ctypes binding of libsgl_hip.so (C ABI declared in include/sgl_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, or a device function is
called without a GPU, this module raises.  torch is imported first so that the HIP runtime
the library binds to (SONAME libamdhip64.so.7) is the one torch already loaded -- device
pointers from torch tensors are then valid inside the library."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import torch  # noqa: F401  (must precede the CDLL: shares torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGL_HIP_LIB: load another build of the library (A/B timing of a kernel change on one GPU box); default: the in-tree build
LIB_PATH = os.environ.get("SGL_HIP_LIB") or os.path.join(_HERE, "csrc", "libsgl_hip.so")

SGL_CSR_STRICT_ORDER = 0x1
SGL_CSR_NO_XCD_REMAP = 0x2
SGL_REDUCE_SUM, SGL_REDUCE_MEAN, SGL_REDUCE_MAX, SGL_REDUCE_MIN, SGL_REDUCE_WSUM = 0, 1, 2, 3, 4
SGL_MAX_HOPS = 64
SGL_MEM_DEFAULT, SGL_MEM_CONTIGUOUS, SGL_MEM_VMM = 0, 1, 2

_lib = None

# name -> (restype, argtypes); every symbol of include/sgl_hip.h
PROTOTYPES = {
    "sgl_version": (c_int, []),
    "sgl_last_error": (c_char_p, []),
    "sgl_device_count": (c_int, [POINTER(c_int)]),
    "sgl_set_tuning": (c_int, [c_char_p, c_int64]),
    "sgl_get_tuning": (c_int, [c_char_p, POINTER(c_int64)]),
    "sgl_plan_build": (c_int, [POINTER(c_void_p), c_void_p, c_int64, c_int32, c_int32]),
    "sgl_plan_counts": (c_int, [c_void_p, POINTER(c_int64)]),
    "sgl_plan_export": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sgl_plan_destroy": (None, [c_void_p]),
    "sgl_csr_create": (c_int, [POINTER(c_void_p), c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_uint32,
                               c_int32, c_int32, c_void_p]),
    "sgl_csr_destroy": (c_int, [c_void_p]),
    "sgl_csr_set_values": (c_int, [c_void_p, c_void_p]),
    "sgl_csr_info": (c_int, [c_void_p, POINTER(c_int64)]),
    "sgl_spmm_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "sgl_spmm_multi_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "sgl_spmm_chain_f32": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "sgl_chain_graph_create": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64]),
    "sgl_chain_graph_launch": (c_int, [c_void_p, c_void_p]),
    "sgl_chain_graph_destroy": (c_int, [c_void_p]),
    "sgl_spmm_axpb_clamp_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_float, c_void_p, c_int64,
                                        c_float, c_float, c_void_p]),
    "sgl_spmm_acc_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_float, c_int,
                                 c_float, c_void_p]),
    "sgl_allgather_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "sgl_exchange_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sgl_exchange_backend": (c_char_p, []),
    "sgl_exchange_selftest": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "FloatCSRMulDenseOMP": (None, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "FloatCSRMulDense": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "sgl_shim_cache_stats": (c_int, [POINTER(c_int64), POINTER(c_int64)]),
    "sgl_norm_prepare": (c_int, [c_int64, c_int64, c_void_p, c_void_p, POINTER(c_int64), c_void_p]),
    "sgl_norm_execute": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_double, c_int, c_double, c_int64,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sgl_norm_degrees": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sgl_norm_execute_lr": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double,
                                    c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sgl_norm_block_prepare": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, POINTER(c_int64), c_void_p]),
    "sgl_norm_block_build": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "sgl_norm_build_symcheck": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "sgl_norm_block_colsum": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sgl_norm_block_scale": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double,
                                     c_void_p, c_void_p, c_void_p]),
    "sgl_norm_block_mix": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "sgl_norm_block_diag_positions": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sgl_norm_block_mix_at": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "sgl_norm_degree_powers": (c_int, [c_int64, c_void_p, c_double, c_void_p, c_void_p, c_void_p]),
    "sgl_coo_to_csr": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               POINTER(c_int64), c_void_p]),
    "sgl_download": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "sgl_csr_permute_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sgl_csr_set_rowmap": (c_int, [c_void_p, c_void_p, c_void_p]),
    "sgl_reorder_lpa_round": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sgl_reorder_community": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, POINTER(c_int64), c_void_p]),
    "sgl_hop_reduce_f32": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                   c_void_p]),
    "sgl_hop_lincomb_f32": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "sgl_hop_select_bwd_f32": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                       c_void_p]),
    "sgl_hop_wsum2d_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                   c_void_p]),
    "sgl_hop_wsum2d_bwd_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                       c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "sgl_hop_rowdot_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "sgl_hop_gate_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                 c_int64, c_int64, c_int64, c_void_p]),
    "sgl_hop_rowdot2_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_uint64, c_void_p, c_int, c_int, c_void_p,
                                    c_int64, c_void_p, c_int64, c_int64, c_void_p]),
    "sgl_hop_colsum_scratch": (c_int64, [c_int, c_int64, c_int64]),
    "sgl_hop_colsum_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                   c_void_p]),
    "sgl_hop_wsum1d_bwd_scratch": (c_int64, [c_int]),
    "sgl_hop_wsum1d_bwd_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                       c_int64, c_void_p]),
    "sgl_hop_concat_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "sgl_hop_concat_padded_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "sgl_nafs_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                             c_void_p]),
    "sgl_nafs_padded_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                    c_void_p]),
    "sgl_nafs_prefix_f32": (c_int, [c_int, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_int64, c_int, c_float, c_int64, c_int64,
                                    c_void_p]),
    "sgl_hop_gate_padded_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                        c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "sgl_hop_recursive_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                      c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "sgl_hop_recursive_bwd_f32": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_int64, c_void_p,
                                          c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "sgl_gather_rows_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                    c_void_p]),
    "sgl_gather_rows_padded_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                           c_void_p]),
    "sgl_gather_hops_padded_f32": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                           c_void_p]),
    "sgl_scatter_rows_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                     c_int64, c_void_p]),
    "sgl_col_signature_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "sgl_content_hash": (c_int, [c_void_p, c_int64, POINTER(c_uint64)]),
}


# libsgl_probe.so (include/sgl_probe.h): measurement and test support -- memory probes, placed allocations, synthetic workloads
PROBE_LIB_PATH = os.environ.get("SGL_PROBE_LIB") or os.path.join(_HERE, "csrc", "libsgl_probe.so")
PROBE_PROTOTYPES = {
    "sgl_probe_last_error": (c_char_p, []),
    "sgl_synth_degrees": (c_int, [c_uint64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "sgl_synth_fill": (c_int, [c_uint64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sgl_synth_features": (c_int, [c_uint64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "sgl_mem_alloc": (c_int, [POINTER(c_void_p), c_int64, c_int, c_int64]),
    "sgl_mem_free": (c_int, [c_void_p]),
    "sgl_probe_stream_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "sgl_probe_gather_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
}
_probe = None


class SglHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SglHipError(
                f"{LIB_PATH} is missing: build it with `python -m sgl_amd.csrc.build` (needs hipcc). "
                "sgl_amd has no CPU fallback for the propagation hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def probe_lib():
    """libsgl_probe.so: what benchmarks and tests need next to the product library (never the propagation path itself)"""
    global _probe
    if _probe is None:
        if not os.path.exists(PROBE_LIB_PATH):
            raise SglHipError(f"{PROBE_LIB_PATH} is missing: build it with `python -m sgl_amd.csrc.build` (needs hipcc)")
        handle = ctypes.CDLL(PROBE_LIB_PATH)
        for name, (res, args) in PROBE_PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _probe = handle
    return _probe


def check_probe(rc, what=""):
    if rc != 0:
        raise SglHipError(f"{what or 'libsgl_probe'} failed (code {rc}): {probe_lib().sgl_probe_last_error().decode('utf-8', 'replace')}")


def last_error():
    return lib().sgl_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise SglHipError(f"{what or 'libsgl_hip'} failed (code {rc}): {last_error()}")


def device_count():
    n = c_int(0)
    check(lib().sgl_device_count(ctypes.byref(n)), "sgl_device_count")
    return n.value


def require_gpu():
    if not torch.cuda.is_available() or device_count() < 1:
        raise SglHipError("no MI355X/HIP device visible: the sgl_amd propagation path is GPU-only (no CPU fallback)")


def set_tuning(key, value):
    check(lib().sgl_set_tuning(key.encode(), int(value)), f"sgl_set_tuning({key})")


def get_tuning(key):
    v = c_int64(0)
    check(lib().sgl_get_tuning(key.encode(), ctypes.byref(v)), f"sgl_get_tuning({key})")
    return v.value


def current_stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def hop_arrays(tensors):
    """(host array of device pointers, host array of leading dimensions) for a list of 2-D row-major
    float32 CUDA tensors (column stride 1)."""
    n = len(tensors)
    ptrs = (c_void_p * n)(*[t.data_ptr() for t in tensors])
    lds = (c_int64 * n)(*[t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)) for t in tensors])
    return ptrs, lds


def content_hash(arr):
    """64-bit content hash of a C-contiguous numpy array, computed by the library's team of host threads (no GPU needed)"""
    import numpy as np
    a = np.ascontiguousarray(arr)
    h = c_uint64(0)
    check(lib().sgl_content_hash(c_void_p(a.ctypes.data), a.nbytes, ctypes.byref(h)), "sgl_content_hash")
    return int(h.value)
