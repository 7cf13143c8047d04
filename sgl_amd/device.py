"""Thin Python layer over the C ABI (include/sgl_hip.h): device CSR handle, SpMM, aggregators.

torch is used for device memory, streams and autograd bookkeeping only; every computation
below is a hand-written HIP kernel in libsgl_hip.so.  There is no CPU fallback."""
import ctypes
import os
from ctypes import c_int64, c_void_p

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check, current_stream_ptr, lib, ptr

__all__ = [
    "DeviceCSR", "default_long_row_nnz", "ChainGraph", "round_up", "row_pitch", "expected_lines", "alloc_rows", "upload_rows", "normalize_adj",
    "normalize_block", "degree_powers", "PreparedAdjacency", "PreparedBlock",
    "placed_empty", "MEM_MODES",
    "hop_reduce", "hop_concat", "hop_reduce_grad", "hop_concat_grad", "hop_lincomb", "hop_wsum1d", "hop_wsum2d", "hop_scores", "hop_scores2", "hop_gate", "gate_fusable", "nafs_aggregate", "nafs_prefix",
    "gather_rows",
]


def round_up(x, m):
    return (x + m - 1) // m * m


def expected_lines(ld, d):
    """average number of 128-byte cache lines a d-float row touches when rows are laid out at a pitch of ld floats"""
    import math
    step = math.gcd(ld * 4, 128)
    offs = range(0, 128, step)
    return sum((o + d * 4 + 127) // 128 for o in offs) / len(offs)


def row_pitch(d, growth=1.25):
    """Leading dimension (floats) for a hop matrix of width d.

    Always a multiple of 4 floats (16-byte aligned rows -> 16-byte lane accesses).  The SpMM is bound by the number of
    128-byte lines it pulls through the fabric, so when rounding the pitch up to a whole number of lines
    (or, for narrow rows, to a power of two that never straddles a line) makes every gathered row touch measurably
    fewer lines, take it: d = 147 -> 160 floats (5 lines instead of 5.5 on average, +8 % memory), d = 500 -> 512,
    d = 12 -> 16; d = 100 stays 100 (always exactly 4 lines either way).  `growth` caps the memory overhead."""
    d = max(int(d), 1)
    ld4, ld32 = round_up(d, 4), round_up(d, 32)
    cands = [(ld32, growth)]
    if d < 32:
        p2 = 4
        while p2 < d:
            p2 *= 2
        cands.insert(0, (p2, max(growth, 1.34)))
    best, best_lines = ld4, expected_lines(ld4, d)
    for ld, cap in cands:
        if ld != ld4 and ld <= cap * ld4 and expected_lines(ld, d) <= 0.97 * best_lines:
            best, best_lines = ld, expected_lines(ld, d)
    return best


def alloc_rows(n, d, device, zero_pad=True):
    """[n, d] float32 view of a row-padded buffer (pitch = row_pitch(d)): every row starts 16-byte aligned, so the
    kernels use 16-byte lane accesses for any d, and rows are placed to touch as few cache lines as possible."""
    ld = row_pitch(d)
    buf = torch.empty((n, ld), dtype=torch.float32, device=device)
    if zero_pad and ld != d:
        buf[:, d:].zero_()
    return buf[:, :d] if ld != d else buf


def own_pad(t):
    """pad columns of an output WE allocated with alloc_rows (the tail of its pitch): what the *_padded_f32 entry points may write
    as zeros so that every line of a row is written whole; 0 for anything that is not a padded [n, d] view with 16-byte rows"""
    n, d = t.shape
    if n <= 1:
        return 0
    ld = t.stride(0)
    return ld - d if (ld > d and ld % 4 == 0 and t.stride(1) == 1 and t.data_ptr() % 16 == 0) else 0


def padded_parent(t):
    """For a [n, d] view created by alloc_rows return the [n, ld] parent view (pad columns included)."""
    n, d = t.shape
    ld = t.stride(0) if n > 1 else row_pitch(d)
    if ld == d:
        return t
    return torch.as_strided(t, (n, ld), (ld, 1), t.storage_offset())


MEM_MODES = {"default": _lib.SGL_MEM_DEFAULT, "contiguous": _lib.SGL_MEM_CONTIGUOUS, "vmm": _lib.SGL_MEM_VMM}


class _PlacedBlock:
    """device memory from sgl_mem_alloc, exported to torch through __cuda_array_interface__; freed with the last tensor"""

    def __init__(self, n_floats, mode, chunk_bytes, device):
        self.device = torch.device(device)
        self.n = int(n_floats)
        p = c_void_p()
        with torch.cuda.device(self.device):
            _lib.check_probe(_lib.probe_lib().sgl_mem_alloc(ctypes.byref(p), self.n * 4, MEM_MODES[mode], int(chunk_bytes)), f"sgl_mem_alloc({mode})")
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": (self.n,), "typestr": "<f4", "data": (self.ptr, False), "version": 2}

    def __del__(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p:
            try:
                with torch.cuda.device(self.device):
                    _lib.probe_lib().sgl_mem_free(c_void_p(p))
            except Exception:
                pass


def placed_empty(shape, device, mode="contiguous", chunk_bytes=0):
    """float32 tensor of `shape` whose memory comes from sgl_mem_alloc with a stated physical placement (MEM_MODES).  For the
    multi-GB feature replicas the SpMM gathers from: a physically contiguous table is translated through far fewer TLB entries
    than whatever hipMalloc found free (profiles/r03_papers_tlb.md).  The block is released when the last view dies."""
    n = 1
    for s_ in shape:
        n *= int(s_)
    blk = _PlacedBlock(max(n, 1), mode, chunk_bytes, device)
    flat = torch.as_tensor(blk, device=blk.device)
    return flat[:n].view(*shape)


_STAGED_MIN_BYTES = 8 << 20     # below this a plain copy is as fast as the staging team


def upload_rows(x, device):
    """host ndarray / tensor [n, d] (any float dtype / order) -> padded device buffer view [n, d] float32"""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    x = x.detach()
    n, d = x.shape
    out = alloc_rows(n, d, device)
    out.copy_(x.to(dtype=torch.float32), non_blocking=False)
    return out


def permute_rows(rowptr, col, val, perm):
    """the rows of a device CSR in the order `perm` (storage row k = input row perm[k]); column ids and every row's entries
    untouched (sgl_csr_permute_rows).  perm: int32 permutation on the device."""
    n = rowptr.numel() - 1
    perm = perm.to(torch.int32).contiguous()
    out_ptr = torch.empty(n + 1, dtype=torch.int64, device=rowptr.device)
    out_col = torch.empty_like(col)
    out_val = torch.empty_like(val)
    with torch.cuda.device(rowptr.device):
        check(lib().sgl_csr_permute_rows(ptr(rowptr), ptr(col), ptr(val), n, ptr(perm), ptr(out_ptr), ptr(out_col), ptr(out_val),
                                         current_stream_ptr()), "sgl_csr_permute_rows")
    return out_ptr, out_col, out_val


def download_rows(t):
    """device [n, d] float32 (any row pitch) -> contiguous pageable CPU tensor, at link rate (sgl_download)"""
    src = t if t.is_contiguous() else t.contiguous()
    host = torch.empty(src.shape, dtype=torch.float32)
    if src.numel() * 4 < _STAGED_MIN_BYTES:
        host.copy_(src)
        return host
    with torch.cuda.device(src.device):
        check(lib().sgl_download(host.data_ptr(), ptr(src), src.numel() * 4, current_stream_ptr()), "sgl_download")
    return host


def _wrote(*tensors):
    """The library wrote into these caller-visible tensors through raw pointers: bump their torch version counters, as any in-place
    torch write would.  Whatever keys on (data_ptr, _version) -- autograd's saved-tensor checks, the content memo of
    hopcache.SharedHops, degree_powers' cache -- then sees the write."""
    for t in tensors:
        if torch.is_tensor(t):
            torch.autograd.graph.increment_version(t)


def _check_mat(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
        raise TypeError(f"{name} must be a 2-D float32 CUDA tensor")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError(f"{name} must be row-major (column stride 1)")
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        raise ValueError(f"{name}: row stride smaller than the row length")


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def default_long_row_nnz(nnz):
    """where sgl_csr_create cuts long rows when it is not told (csrc/sgl_spmm.hip: a function of the matrix's nnz only).  A row
    block of a sharded matrix has fewer non-zeros than the whole and would fall into another bracket: the distributed paths pass
    default_long_row_nnz(GLOBAL nnz) explicitly, so that the same rows are cut at the same places whatever the world size and the
    default-order hops of a sharded run stay bit-identical to the single-GPU run."""
    nnz = int(nnz)
    return 32 if nnz < (1 << 18) else 128 if nnz < (1 << 20) else 512 if nnz < (1 << 22) else 2048


class DeviceCSR:
    """A CSR matrix resident on one GPU plus its SpMM execution plan (sgl_csr_create).

    rowptr int64 [n_rows+1], col int32 [nnz], val float32 [nnz] -- CUDA tensors, kept alive here."""

    def __init__(self, rowptr, col, val, shape, strict=False, item_nnz=0, long_row_nnz=0, xcd_remap=True):
        _lib.require_gpu()
        for t, dt, nm in ((rowptr, torch.int64, "rowptr"), (col, torch.int32, "col"), (val, torch.float32, "val")):
            if not (t.is_cuda and t.dtype == dt and t.dim() == 1 and t.is_contiguous()):
                raise TypeError(f"{nm} must be a contiguous 1-D CUDA tensor of dtype {dt}")
        self.rowptr, self.col, self.val = rowptr, col, val
        self.shape = (int(shape[0]), int(shape[1]))
        self.nnz = int(col.numel())
        self.device = rowptr.device
        self.strict = bool(strict)
        flags = (_lib.SGL_CSR_STRICT_ORDER if strict else 0) | (0 if xcd_remap else _lib.SGL_CSR_NO_XCD_REMAP)
        h = c_void_p()
        with torch.cuda.device(self.device):
            check(lib().sgl_csr_create(ctypes.byref(h), self.shape[0], self.shape[1], self.nnz, ptr(rowptr), ptr(col),
                                       ptr(val), flags, int(item_nnz), int(long_row_nnz), current_stream_ptr()),
                  "sgl_csr_create")
        self._h = h

    @classmethod
    def from_scipy(cls, adj, device="cuda", **kw):
        """scipy CSR (any float dtype; values rounded to float32 here, where the reference rounds them,
        operators/utils.py:32) -> device."""
        import scipy.sparse as sp
        if not sp.isspmatrix_csr(adj):
            raise TypeError("expected a scipy.sparse.csr_matrix")
        rowptr = torch.from_numpy(adj.indptr.astype(np.int64)).to(device)
        col = torch.from_numpy(adj.indices.astype(np.int32)).to(device)
        val = torch.from_numpy(adj.data.astype(np.float32)).to(device)
        return cls(rowptr, col, val, adj.shape, **kw)

    def set_values(self, val):
        """same sparsity structure, new values (another normalisation of the same graph): the execution plan is kept"""
        if not (val.is_cuda and val.dtype == torch.float32 and val.dim() == 1 and val.is_contiguous() and val.numel() == self.nnz):
            raise TypeError("val must be a contiguous float32 CUDA tensor with one entry per non-zero")
        check(lib().sgl_csr_set_values(self._h, ptr(val)), "sgl_csr_set_values")
        self.val = val
        return self

    rowmap = None

    def set_rowmap(self, rowmap):
        """the handle's rows are stored in processing order: storage row i is output row rowmap[i] (int32 permutation on the
        device, kept alive here; None removes it).  sgl_csr_set_rowmap."""
        if rowmap is not None and not (rowmap.is_cuda and rowmap.dtype == torch.int32 and rowmap.dim() == 1
                                       and rowmap.is_contiguous() and rowmap.numel() == self.shape[0]):
            raise TypeError("rowmap must be a contiguous int32 CUDA tensor with one entry per row")
        with torch.cuda.device(self.device):
            check(lib().sgl_csr_set_rowmap(self._h, ptr(rowmap) if rowmap is not None else None, current_stream_ptr()),
                  "sgl_csr_set_rowmap")
        self.rowmap = rowmap
        return self

    def info(self):
        a = (c_int64 * 8)()
        check(lib().sgl_csr_info(self._h, a), "sgl_csr_info")
        keys = ("n_rows", "n_cols", "nnz", "n_items", "n_pieces", "n_long_rows", "flags", "workspace_bytes")
        return dict(zip(keys, list(a)))

    def spmm(self, x, out=None, accumulate=False):
        """out = A @ x (+ out when accumulate).  x: [n_cols, d] float32 CUDA row-major; returns [n_rows, d]."""
        _check_mat(x, "x")
        if x.shape[0] != self.shape[1]:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        d = x.shape[1]
        if out is None:
            if accumulate:
                raise ValueError("accumulate=True needs an `out` tensor")
            out = alloc_rows(self.shape[0], d, x.device, zero_pad=True)
        else:
            _check_mat(out, "out")
            if out.shape != (self.shape[0], d):
                raise ValueError("out has the wrong shape")
        with torch.cuda.device(self.device):
            check(lib().sgl_spmm_f32(self._h, ptr(x), _ld(x), ptr(out), _ld(out), d, int(bool(accumulate)),
                                     current_stream_ptr()), "sgl_spmm_f32")
        _wrote(out)
        return out

    ACC_MODES = {"sum": 0, "wsum": 1, "max": 2, "min": 3}

    def spmm_acc(self, x, out, acc, w=1.0, weighted=False, divisor=1.0, mode=None):
        """out = A @ x and, in the same kernel, acc <- acc + out (weighted: acc + w * out), then acc / divisor if
        divisor != 1, or acc <- max / min(acc, out) (sgl_spmm_acc_f32): the running hop aggregate of Sum / Mean /
        SimpleWeighted / Max / Min"""
        mode = self.ACC_MODES[mode] if mode is not None else int(bool(weighted))
        _check_mat(x, "x")
        _check_mat(out, "out")
        _check_mat(acc, "acc")
        if x.shape[0] != self.shape[1] or out.shape != (self.shape[0], x.shape[1]) or acc.shape != out.shape:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        with torch.cuda.device(self.device):
            check(lib().sgl_spmm_acc_f32(self._h, ptr(x), _ld(x), ptr(out), _ld(out), x.shape[1], ptr(acc), _ld(acc),
                                         float(w), mode, float(divisor), current_stream_ptr()),
                  "sgl_spmm_acc_f32")
        _wrote(out, acc)
        return out

    def spmm_multi(self, x, out_ptrs, ld, row_mask=None):
        """A @ x stored into several [n_rows, d] matrices given as RAW device addresses (the first is normally local,
        the others peer-GPU replicas mapped through IPC) with leading dimension `ld`.  row_mask: optional uint8 CUDA
        tensor [n_rows]; bit q set = destination q+1 gets that row."""
        _check_mat(x, "x")
        if x.shape[0] != self.shape[1]:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        n_out = len(out_ptrs)
        if not 1 <= n_out <= 8:
            raise ValueError("between 1 and 8 output replicas are supported")
        arr = (c_void_p * n_out)(*[int(p) for p in out_ptrs])
        with torch.cuda.device(self.device):
            if row_mask is not None and not (row_mask.is_cuda and row_mask.dtype == torch.uint8 and row_mask.is_contiguous()
                                             and row_mask.numel() == self.shape[0]):
                raise ValueError("row_mask must be a contiguous uint8 CUDA tensor with one entry per row")
            check(lib().sgl_spmm_multi_f32(self._h, ptr(x), _ld(x), n_out, arr, int(ld), x.shape[1],
                                           ptr(row_mask) if row_mask is not None else None, current_stream_ptr()),
                  "sgl_spmm_multi_f32")

    def spmm_chain(self, x, n_hops, outs=None):
        """[A x, A^2 x, ..., A^k x] with ONE library call (the hop loop runs in C).  x: [n, d] row-major CUDA; the
        results are row-padded buffers of the same width as x (or the caller's `outs`)."""
        _check_mat(x, "x")
        if x.shape[0] != self.shape[1] or self.shape[0] != self.shape[1]:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        d = x.shape[1]
        if outs is None:
            outs = [alloc_rows(self.shape[0], d, x.device, zero_pad=True) for _ in range(n_hops)]
        else:
            if len(outs) != n_hops:
                raise ValueError("need one output matrix per hop")
            for o in outs:
                _check_mat(o, "outs[*]")
                if o.shape != (self.shape[0], d):
                    raise ValueError("an output matrix has the wrong shape")
        if n_hops:
            ptrs = (c_void_p * n_hops)(*[o.data_ptr() for o in outs])
            lds = (c_int64 * n_hops)(*[_ld(o) for o in outs])
            with torch.cuda.device(self.device):
                check(lib().sgl_spmm_chain_f32(self._h, n_hops, ptr(x), _ld(x), ptrs, lds, d, current_stream_ptr()),
                      "sgl_spmm_chain_f32")
            _wrote(*outs)
        return outs

    def capture_chain(self, x, outs):
        """hipGraph of `spmm_chain(x, len(outs), outs)`: returns a ChainGraph whose replay() re-runs the k hops on the
        current stream with one launch (x and outs are baked in: refill x in place between replays)."""
        _check_mat(x, "x")
        for o in outs:
            _check_mat(o, "outs[*]")
            if o.shape != (self.shape[0], x.shape[1]):
                raise ValueError("an output matrix has the wrong shape")
        n_hops = len(outs)
        ptrs = (c_void_p * n_hops)(*[o.data_ptr() for o in outs])
        lds = (c_int64 * n_hops)(*[_ld(o) for o in outs])
        g = c_void_p()
        torch.cuda.synchronize(self.device)
        with torch.cuda.device(self.device):
            check(lib().sgl_chain_graph_create(ctypes.byref(g), self._h, n_hops, ptr(x), _ld(x), ptrs, lds, x.shape[1]),
                  "sgl_chain_graph_create")
        return ChainGraph(g, self, x, list(outs))

    def spmm_axpb_clamp(self, x, alpha, res=None, lo=float("-inf"), hi=float("inf"), out=None):
        """out = clamp(alpha * (A @ x) + res, lo, hi) in one kernel (the label-propagation step)"""
        _check_mat(x, "x")
        if x.shape[0] != self.shape[1]:
            raise ValueError("Dimension mismatch detected for the adjacency and the feature matrix!")
        d = x.shape[1]
        if out is None:
            out = alloc_rows(self.shape[0], d, x.device, zero_pad=True)
        else:
            _check_mat(out, "out")
        if res is not None:
            _check_mat(res, "res")
            if res.shape != (self.shape[0], d):
                raise ValueError("res has the wrong shape")
        with torch.cuda.device(self.device):
            check(lib().sgl_spmm_axpb_clamp_f32(self._h, ptr(x), _ld(x), ptr(out), _ld(out), d, float(alpha),
                                                ptr(res) if res is not None else None, _ld(res) if res is not None else 0,
                                                float(lo), float(hi), current_stream_ptr()), "sgl_spmm_axpb_clamp_f32")
        _wrote(out)
        return out

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().sgl_csr_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class ChainGraph:
    """a captured k-hop propagation (DeviceCSR.capture_chain); keeps the matrix and the buffers alive"""

    def __init__(self, handle, csr, x, outs):
        self._g, self.csr, self.x, self.outs = handle, csr, x, outs

    def replay(self):
        with torch.cuda.device(self.csr.device):
            check(lib().sgl_chain_graph_launch(self._g, current_stream_ptr()), "sgl_chain_graph_launch")
        _wrote(*self.outs)
        return self.outs

    def close(self):
        g, self._g = getattr(self, "_g", None), None
        if g:
            lib().sgl_chain_graph_destroy(g)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_POW_CACHE = []          # at most one entry: (deg tensor, its version counter, r, host_pow, left, right)
_POW_THREADS = max(1, min(64, (os.cpu_count() or 2) // 2))
pow_stats = {"cache_hits": 0, "host_unique": 0, "host_full": 0, "device": 0}


def _host_powers(d, r):
    """numpy's deg^(r-1), deg^(-r) with inf -> 0 (utils.py:79-84) on a 1-D float64 array, written into two new arrays; long
    vectors are cut into chunks evaluated by a team of threads (the ufunc loop releases the GIL; the values do not depend on the cut)"""
    left, right = np.empty_like(d), np.empty_like(d)

    def run(a, b):
        with np.errstate(divide="ignore", invalid="ignore"):
            np.power(d[a:b], r - 1, out=left[a:b])
            left[a:b][np.isinf(left[a:b])] = 0.
            np.power(d[a:b], -r, out=right[a:b])
            right[a:b][np.isinf(right[a:b])] = 0.
    n = d.shape[0]
    chunk = 1 << 20
    if n <= 2 * chunk or _POW_THREADS == 1:
        run(0, n)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(_POW_THREADS) as ex:
            list(ex.map(lambda a: run(a, min(a + chunk, n)), range(0, n, chunk)))
    return left, right


def _few_distinct(deg, sample=4096):
    """cheap guess BEFORE sorting the whole vector: do evenly spaced samples repeat each other?  (unit-weight graphs: a few
    hundred distinct degrees among millions of nodes; real-valued weights: every degree its own)"""
    n = deg.numel()
    if n <= 4 * sample:
        return True
    s = deg[:: n // sample][:sample]
    return torch.unique(s).numel() * 2 <= s.numel()


def degree_powers(deg, r, host_pow=True):
    """deg^(r-1), deg^(-r) with inf -> 0 (operators/utils.py:79-84) as two fp64 tensors on deg's device.

    host_pow=True: evaluated on the HOST by numpy exactly as the reference does: the same routine, hence bit-identical
    degree factors and a bit-identical A_hat.  Only the DISTINCT degree values travel when there are few of them (a unit-weight
    graph has far fewer distinct degrees than nodes: 2.4 M nodes -> a few thousand values; a strided sample decides, so a
    vector of all-distinct real-valued degrees is never sorted); otherwise the vector goes through pinned buffers and a team of
    host threads.  host_pow=False: the device's pow() (sgl_norm_degree_powers), within 1 ulp(fp64) of the host's -- A_hat then
    differs from the reference's in the last fp32 bit of a handful of entries (tests: <= 1 ulp), far inside the 1e-5 contract.
    host_pow="auto": the host route when the degrees have few distinct values (it then costs microseconds), the device otherwise.
    The last result is cached per (degree vector, r, route): an alpha sweep, or re-normalising the same graph, pays once."""
    r = float(r)
    for ent in _POW_CACHE:
        if ent[0] is deg and ent[1] == deg._version and ent[2] == r and ent[3] == host_pow:
            pow_stats["cache_hits"] += 1
            return ent[4], ent[5]
    asked = host_pow
    few = None
    if host_pow == "auto":
        # host route (bit-identical to the reference) whenever it is cheap -- few distinct degrees: every unit-weight graph --
        # the device's pow() when every node has its own real-valued degree
        few = (not deg.is_cuda) or deg.numel() <= 4096 or _few_distinct(deg)
        host_pow = few
    if deg.is_cuda and not host_pow:
        left, right = torch.empty_like(deg), torch.empty_like(deg)
        with torch.cuda.device(deg.device):
            check(lib().sgl_norm_degree_powers(deg.numel(), ptr(deg), r, ptr(left), ptr(right), current_stream_ptr()),
                  "sgl_norm_degree_powers")
        pow_stats["device"] += 1
    elif deg.is_cuda and deg.numel() > 4096 and (few if few is not None else _few_distinct(deg)):
        uniq, inv = torch.unique(deg, return_inverse=True)
        lu, ru = _host_powers(uniq.cpu().numpy(), r)
        left, right = torch.from_numpy(lu).to(deg.device)[inv], torch.from_numpy(ru).to(deg.device)[inv]
        pow_stats["host_unique"] += 1
    elif deg.is_cuda:
        from . import hostpool
        n = deg.numel()
        bufs = [hostpool.take((n,), torch.float64) if n >= (1 << 20) else None for _ in range(3)]
        h = bufs[0] if bufs[0] is not None else torch.empty(n, dtype=torch.float64)
        h.copy_(deg)
        hl, hr = _host_powers(h.numpy(), r)
        outs = []
        for src, buf in ((hl, bufs[1]), (hr, bufs[2])):
            t = torch.from_numpy(src)
            if buf is not None:
                buf.copy_(t)
                t = buf
            outs.append(t.to(deg.device, non_blocking=buf is not None))
        torch.cuda.current_stream(deg.device).synchronize()      # the pinned buffers go back to the pool
        left, right = outs
        pow_stats["host_full"] += 1
    else:
        hl, hr = _host_powers(deg.detach().numpy().astype(np.float64, copy=False), r)
        left, right = torch.from_numpy(hl), torch.from_numpy(hr)
    _POW_CACHE[:] = [(deg, deg._version, r, asked, left, right)]
    return left, right


def clear_power_cache():
    _POW_CACHE.clear()


class PreparedAdjacency:
    """A + I of a whole matrix on the device (CSR, fp64 values), its row sums = degrees, and whether A is symmetric --
    everything about a graph that does not depend on r / alpha (sgl_norm_build_symcheck).  One preparation serves every
    normalisation of the same graph: the r-sweep of the NAFS task, the alpha-sweep of a PaSca search."""

    def __init__(self, rowptr, col, val, n):
        _lib.require_gpu()
        dev = rowptr.device
        nnz = int(col.numel())
        self.n, self.src = int(n), (rowptr, col, val)
        nnz_out = c_int64(0)
        with torch.cuda.device(dev):
            check(lib().sgl_norm_block_prepare(n, 0, nnz, ptr(rowptr), ptr(col), ctypes.byref(nnz_out), current_stream_ptr()),
                  "sgl_norm_block_prepare")
            m = nnz_out.value
            self.rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
            self.col = torch.empty(m, dtype=torch.int32, device=dev)
            self.t64 = torch.empty(m, dtype=torch.float64, device=dev)
            self.deg = torch.empty(n, dtype=torch.float64, device=dev)
            fp = torch.zeros(1, dtype=torch.int64, device=dev)
            check(lib().sgl_norm_build_symcheck(n, nnz, ptr(rowptr), ptr(col), ptr(val), m, ptr(self.rowptr), ptr(self.col),
                                                ptr(self.t64), ptr(self.deg), ptr(fp), current_stream_ptr()),
                  "sgl_norm_build_symcheck")
            self.symmetric = int(fp.item()) == 0
        self.nnz_out = m
        self.device = dev
        if self.symmetric:
            self.src = None            # the raw matrix is only needed again for the one-off transposition of a directed graph

    def nbytes(self):
        """device bytes this object keeps alive beyond the caller's own matrix (A + I in fp64, degrees, cached Laplacian, ...)"""
        seen, total = set(), 0
        objs = [self] + ([self.__dict__["_tblock"]] if self.__dict__.get("_tblock") is not None else [])
        for o in objs:
            for name, v in o.__dict__.items():
                if name == "src":
                    continue
                if isinstance(v, tuple) and len(v) == 2 and torch.is_tensor(v[1]):
                    v = v[1]                                         # _hat64 = (key, tensor)
                if torch.is_tensor(v) and v.is_cuda and v.data_ptr() not in seen:
                    seen.add(v.data_ptr())
                    total += v.numel() * v.element_size()
        return total

    def drop_values(self):
        """release the cached fp64 Laplacian of the last PPR request (8 bytes per non-zero)"""
        self.__dict__["_hat64"] = None
        tb = self.__dict__.get("_tblock")
        if tb is not None:
            tb.drop_values()

    def normalize(self, r, alpha=None, return_fp64=False, host_pow=True):
        """A_hat for this (r, alpha).  Symmetric A: one scaling pass over A + I, A_hat[j,i] = (A'[j,i] L[j]) R[i] -- no
        transposition; otherwise the transpose is built once (one stable sort) and every (r, alpha) is the same single pass over it.
        Both are bit-identical to the reference's scipy result (host_pow=True; see degree_powers).  PPR requests keep the fp64
        Laplacian of their r: the next alpha of a sweep is one stream over it (sgl_norm_block_mix), bit-identical to the one-pass form."""
        dev = self.device
        with torch.cuda.device(dev):
            if self.symmetric:
                vals = _scaled_values(self, self.n, 0, self.rowptr, self.col, self.t64, self.deg, r, alpha, return_fp64, host_pow)
                return (self.rowptr, self.col) + vals
            # directed / value-asymmetric A: A_hat[j, i] = (T'[j, i] L[j]) R[i] with T = A^T.  The transposition (one stable sort:
            # sgl_coo_to_csr with rows and columns swapped) does not depend on (r, alpha) either: done ONCE, after which every
            # (r, alpha) is the same single pass as in the symmetric case -- 2 ms instead of the 21 ms of the general pipeline
            # (sgl_norm_execute_lr re-sorts per call) at the products shape.  The degrees stay the sequential row sums of A + I
            # computed above (scipy's order), so the result is bit-identical to the general pipeline and to the reference.
            tb = self.__dict__.get("_tblock")
            if tb is None:
                from .io import coo_to_csr_device
                rowptr, col, val = self.src
                rows = torch.repeat_interleave(torch.arange(self.n, dtype=torch.int64, device=dev), rowptr[1:] - rowptr[:-1])
                t = coo_to_csr_device(col.to(torch.int64), rows, val, self.n, device=dev)
                tb = self._tblock = PreparedBlock(t.rowptr, t.col, t.val, 0, self.n, symmetric=False, deg=self.deg)
                self.rowptr = self.col = self.t64 = self.src = None   # neither A + I nor the raw matrix is needed any more
            return tb.normalize(r, alpha, return_fp64=return_fp64, host_pow=host_pow)


def _scaled_values(owner, n_loc, row0, rowptr, col, t64, deg, r, alpha, return_fp64, host_pow, keep_hat64=None):
    """the (r, alpha)-dependent pass over a prepared T' = T + I (rows [row0, row0 + n_loc), global column ids; `deg` = the global
    degree vector): returns (val32,) or (val32, val64).  `owner` carries the one-entry cache of the fp64 Laplacian values
    (`_hat64` = ((r, host_pow), tensor)): kept when a PPR matrix is asked for (or keep_hat64=True), so that the other alphas of a
    sweep at the same r skip the degree factors and the R gather altogether."""
    dev = rowptr.device
    m = int(col.numel())
    key = (float(r), host_pow)
    keep = (alpha is not None) if keep_hat64 is None else bool(keep_hat64)
    cached = getattr(owner, "_hat64", None)
    hat64 = cached[1] if (cached is not None and cached[0] == key) else None
    o_val = torch.empty(m, dtype=torch.float32, device=dev)
    o_v64 = torch.empty(m, dtype=torch.float64, device=dev) if return_fp64 else None
    if hat64 is None:
        left, right = degree_powers(deg, r, host_pow)
        left_loc = left if (row0 == 0 and n_loc == left.numel()) else left[row0:row0 + n_loc].contiguous()
        if alpha is None or not keep:
            if keep:                                 # Laplacian asked for, values to be kept: the fp64 output IS the cache entry
                o_v64 = o_v64 if o_v64 is not None else torch.empty(m, dtype=torch.float64, device=dev)
            check(lib().sgl_norm_block_scale(n_loc, row0, ptr(rowptr), ptr(col), ptr(t64), ptr(left_loc), ptr(right),
                                             int(alpha is not None), float(alpha if alpha is not None else 0.0), ptr(o_val),
                                             ptr(o_v64) if o_v64 is not None else None, current_stream_ptr()), "sgl_norm_block_scale")
            if keep:
                owner._hat64 = (key, o_v64)
                if return_fp64:
                    o_v64 = o_v64.clone()            # the caller may edit what it gets; the cache entry stays private
            return (o_val, o_v64) if return_fp64 else (o_val,)
        # PPR with the cache empty: the Laplacian in fp64 first (o_val is scratch for its fp32 rounding), then the mix below
        hat64 = torch.empty(m, dtype=torch.float64, device=dev)
        check(lib().sgl_norm_block_scale(n_loc, row0, ptr(rowptr), ptr(col), ptr(t64), ptr(left_loc), ptr(right), 0, 0.0,
                                         ptr(o_val), ptr(hat64), current_stream_ptr()), "sgl_norm_block_scale")
        owner._hat64 = (key, hat64)
    if alpha is None:
        # Laplacian again at a cached r: alpha = 0 would multiply by 1.0 and add 0.0 to the diagonal -- exact, but a plain rounding
        # pass says what it does
        o_val.copy_(hat64)
        return (o_val, hat64.clone()) if return_fp64 else (o_val,)
    diag = getattr(owner, "_diag", None)
    if diag is None:                                  # where each row's diagonal entry sits: once per prepared block
        diag = torch.empty(n_loc, dtype=torch.int64, device=dev)
        check(lib().sgl_norm_block_diag_positions(n_loc, row0, ptr(rowptr), ptr(col), ptr(diag), current_stream_ptr()),
              "sgl_norm_block_diag_positions")
        owner._diag = diag
    check(lib().sgl_norm_block_mix_at(m, n_loc, ptr(hat64), ptr(diag), float(alpha), ptr(o_val), ptr(o_v64) if return_fp64 else None,
                                      current_stream_ptr()), "sgl_norm_block_mix_at")
    return (o_val, o_v64) if return_fp64 else (o_val,)


def normalize_adj(rowptr, col, val, n, r, alpha=None, return_fp64=False, host_pow=True, prepared=None):
    """Device adj_to_symmetric_norm (+ optional PPR mix): canonical CSR of A on device -> CSR of A_hat.
    rowptr int64 [n+1], col int32, val float32 (CUDA).  Returns (rowptr, col, val[, val64]).
    host_pow (default): the two degree powers are evaluated by the host's numpy like the reference's, everything per
    non-zero stays on the GPU; the rounded A_hat is then bit-identical to scipy's.  `prepared`: a PreparedAdjacency of the
    same matrix (re-used across r / alpha); host_pow=False is the all-device pipeline with the GPU's pow()."""
    _lib.require_gpu()
    if host_pow or prepared is not None:
        prep = prepared if prepared is not None else PreparedAdjacency(rowptr, col, val, n)
        return prep.normalize(r, alpha, return_fp64=return_fp64, host_pow=host_pow)
    nnz = int(col.numel())
    dev = rowptr.device
    nnz_out = c_int64(0)
    with torch.cuda.device(dev):
        check(lib().sgl_norm_prepare(n, nnz, ptr(rowptr), ptr(col), ctypes.byref(nnz_out), current_stream_ptr()),
              "sgl_norm_prepare")
        m = nnz_out.value
        o_ptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
        o_col = torch.empty(m, dtype=torch.int32, device=dev)
        o_val = torch.empty(m, dtype=torch.float32, device=dev)
        o_v64 = torch.empty(m, dtype=torch.float64, device=dev) if return_fp64 else None
        check(lib().sgl_norm_execute(n, nnz, ptr(rowptr), ptr(col), ptr(val), float(r), int(alpha is not None),
                                     float(alpha if alpha is not None else 0.0), m, ptr(o_ptr), ptr(o_col), ptr(o_val),
                                     ptr(o_v64) if return_fp64 else None, current_stream_ptr()), "sgl_norm_execute")
    return (o_ptr, o_col, o_val, o_v64) if return_fp64 else (o_ptr, o_col, o_val)


class PreparedBlock:
    """Everything about ONE RANK'S row block that does not depend on (r, alpha): T' = T + I for rows [row0, row0 + n_local) of
    T = A^T (symmetric=True: of A itself) as CSR with fp64 values (sgl_norm_block_build), and the GLOBAL degree vector of A + I
    (`deg`, fp64 [n_cols]).  The only communication is that vector: the blocks' row sums when A is symmetric, an all-reduce of
    the column sums otherwise; a caller that already holds it passes `deg=` and nothing is communicated.  One preparation then
    serves every normalize(r, alpha) of the block: the degree factors are cached per r (degree_powers), a PPR sweep keeps the
    fp64 Laplacian of its r, so (r, alpha) costs one pass over the block -- what a PaSca sweep over graph operators pays per
    candidate (sgl/search/search_config.py:14-15)."""

    def __init__(self, rowptr, col, val, row0, n_cols, symmetric=True, group=None, deg=None):
        import torch.distributed as dist
        _lib.require_gpu()
        dev = rowptr.device
        n_loc = int(rowptr.numel()) - 1
        nnz = int(col.numel())
        nnz_out = c_int64(0)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.n_local, self.row0, self.n_cols, self.symmetric = n_loc, int(row0), int(n_cols), bool(symmetric)
        with torch.cuda.device(dev):
            check(lib().sgl_norm_block_prepare(n_loc, row0, nnz, ptr(rowptr), ptr(col), ctypes.byref(nnz_out),
                                               current_stream_ptr()), "sgl_norm_block_prepare")
            m = nnz_out.value
            self.rowptr = torch.empty(n_loc + 1, dtype=torch.int64, device=dev)
            self.col = torch.empty(m, dtype=torch.int32, device=dev)
            self.t64 = torch.empty(m, dtype=torch.float64, device=dev)
            rowsum = torch.empty(n_loc, dtype=torch.float64, device=dev)
            check(lib().sgl_norm_block_build(n_loc, row0, nnz, ptr(rowptr), ptr(col), ptr(val), m, ptr(self.rowptr), ptr(self.col),
                                             ptr(self.t64), ptr(rowsum), current_stream_ptr()), "sgl_norm_block_build")
            if deg is not None:
                if deg.numel() != n_cols:
                    raise ValueError("deg must hold one entry per column")
                deg = deg.to(dev)
            elif symmetric:
                if multi:
                    deg = torch.zeros(n_cols, dtype=torch.float64, device=dev)
                    deg[row0:row0 + n_loc] = rowsum
                    dist.all_reduce(deg, group=group)      # blocks are disjoint: the sum IS the concatenation (exact)
                else:
                    if n_loc != n_cols or row0 != 0:
                        raise ValueError("a single process must hold the whole matrix")
                    deg = rowsum
            else:
                deg = torch.zeros(n_cols, dtype=torch.float64, device=dev)
                check(lib().sgl_norm_block_colsum(n_cols, m, ptr(self.col), ptr(self.t64), ptr(deg), current_stream_ptr()),
                      "sgl_norm_block_colsum")
                if multi:
                    dist.all_reduce(deg, group=group)
        self.deg = deg
        self.nnz_out = m
        self._hat64 = None

    def normalize(self, r, alpha=None, return_fp64=False, host_pow=True, keep_hat64=None):
        """(rowptr, col, val[, val64]) of this block of A_hat (local row pointers, global column ids)"""
        with torch.cuda.device(self.rowptr.device):
            vals = _scaled_values(self, self.n_local, self.row0, self.rowptr, self.col, self.t64, self.deg, r, alpha, return_fp64,
                                  host_pow, keep_hat64)
        return (self.rowptr, self.col) + vals

    def drop_values(self):
        """release the cached fp64 Laplacian (8 bytes per non-zero)"""
        self._hat64 = None


def normalize_block(rowptr, col, val, row0, n_cols, r, alpha=None, symmetric=True, group=None, return_fp64=False, deg=None,
                    host_pow=True, prepared=None):
    """Rows [row0, row0 + n_local) of A_hat from the same rows of T = A^T (symmetric=True: of A itself), each rank of a
    row-sharded job on its own block (sgl_norm_block_*).  The only communication is the degree vector: an all-gather of
    the blocks' row sums when A is symmetric, an all-reduce of the column sums otherwise.  Without an initialised
    process group (or world size 1) the block must be the whole matrix.  Returns (rowptr, col, val[, val64]) of the block
    (local row pointers, global column ids).  `deg` (fp64 [n_cols], any device): the global degree vector of A + I if
    the caller already has it -- then nothing is communicated at all.  `prepared`: a PreparedBlock of the same block (the
    (r, alpha)-independent part, re-used across a sweep); host_pow: see degree_powers (True = bit-identical to the reference)."""
    prep = prepared if prepared is not None else PreparedBlock(rowptr, col, val, row0, n_cols, symmetric=symmetric, group=group, deg=deg)
    return prep.normalize(r, alpha, return_fp64=return_fp64, host_pow=host_pow, keep_hat64=None if prepared is not None else False)


# ------------------------------------------------------------------------------------------------
# aggregators
# ------------------------------------------------------------------------------------------------
def _check_hops(feats):
    if len(feats) < 1 or len(feats) > _lib.SGL_MAX_HOPS:
        raise ValueError(f"between 1 and {_lib.SGL_MAX_HOPS} hop matrices are supported")
    for i, f in enumerate(feats):
        _check_mat(f, f"feat_list[{i}]")
        if f.shape != feats[0].shape or f.device != feats[0].device:
            raise ValueError("all hop matrices must have the same shape and device")


def _widened(feats, out):
    """For d % 4 != 0: when every hop matrix and the (freshly allocated) output are row-padded to a multiple of
    4 floats, run element-wise kernels over the padded width so they use 16-byte lanes.  Inputs' pad columns are
    only read, the output's pad columns belong to us."""
    n, d = out.shape
    strides = {t.stride(0) for t in list(feats) + [out]} if n > 1 else set()
    if len(strides) == 1 and next(iter(strides)) % 4 == 0 and next(iter(strides)) > d:
        dp = next(iter(strides))      # every buffer has the same pitch: stream whole rows, pad included (contiguous)
    else:
        dp = round_up(d, 4)
    if d == dp or n <= 1:
        return feats, out, d
    ok = all(f.stride(0) % 4 == 0 and f.stride(0) >= dp and f.data_ptr() % 16 == 0 for f in list(feats) + [out])
    if not ok:
        return feats, out, d
    wide = [torch.as_strided(f, (n, dp), (f.stride(0), 1), f.storage_offset()) for f in feats]
    return wide, torch.as_strided(out, (n, dp), (out.stride(0), 1), out.storage_offset()), dp


def hop_reduce(op, feats, weights=None):
    """sum / mean / max / min / 1-D weighted sum over the hop list -> new [n, d] tensor"""
    _check_hops(feats)
    n, d = feats[0].shape
    result = alloc_rows(n, d, feats[0].device)
    feats, out, d = _widened(feats, result)
    ptrs, lds = _lib.hop_arrays(feats)
    w = None
    if op == _lib.SGL_REDUCE_WSUM:
        w = weights.detach().to(device=feats[0].device, dtype=torch.float32).contiguous()
        if w.numel() != len(feats):
            raise ValueError("The feature list and the weight list have different lengths!")
    with torch.cuda.device(feats[0].device):
        check(lib().sgl_hop_reduce_f32(op, len(feats), ptrs, lds, ptr(w) if w is not None else None, ptr(out), _ld(out),
                                       n, d, current_stream_ptr()), "sgl_hop_reduce_f32")
    return result


def hop_concat(feats):
    _check_hops(feats)
    n, d = feats[0].shape
    H = len(feats)
    out = alloc_rows(n, H * d, feats[0].device)
    ptrs, lds = _lib.hop_arrays(feats)
    with torch.cuda.device(feats[0].device):
        check(lib().sgl_hop_concat_padded_f32(H, ptrs, lds, ptr(out), _ld(out), own_pad(out), n, d, current_stream_ptr()),
              "sgl_hop_concat_padded_f32")
    return out


def hop_lincomb(feats, weights, outs=None):
    """out_k = sum_j weights[k, j] * feats[j]: a small dense matrix applied across the hop dimension, every input element read once
    for all outputs (sgl_hop_lincomb_f32; zero weights skipped, one fma chain in j order per output).  feats: up to 16 [n, d] hop
    matrices; weights: [n_out, len(feats)] (host or device); outs: optional list of n_out [n, d] matrices (not aliasing the inputs).
    Returns the list of outputs."""
    _check_hops(feats)
    n, d = feats[0].shape
    w = torch.as_tensor(weights, dtype=torch.float32).to(feats[0].device).contiguous()
    if w.dim() != 2 or w.shape[1] != len(feats):
        raise ValueError("weights must be [n_out, len(feats)]")
    n_out = int(w.shape[0])
    if len(feats) > 16:
        raise ValueError("hop_lincomb takes at most 16 input matrices per call")
    if outs is None:
        outs = [alloc_rows(n, d, feats[0].device) for _ in range(n_out)]
    if len(outs) != n_out or any(tuple(o.shape) != (n, d) for o in outs):
        raise ValueError("one [n, d] output per row of weights")
    # stream whole pitches when every matrix shares one (pad columns: zeros in, zeros out)
    wide_in, _, dw = _widened(list(feats) + list(outs), outs[0])
    if dw != d:
        f_w, o_w = wide_in[:len(feats)], wide_in[len(feats):]
    else:
        f_w, o_w = list(feats), list(outs)
    ptrs, lds = _lib.hop_arrays(f_w)
    optrs, olds = _lib.hop_arrays(o_w)
    with torch.cuda.device(feats[0].device):
        check(lib().sgl_hop_lincomb_f32(len(feats), ptrs, lds, n_out, optrs, olds, ptr(w), int(w.stride(0)), n, dw, current_stream_ptr()),
              "sgl_hop_lincomb_f32")
    _wrote(*outs)
    return outs


class _ReduceGrad(torch.autograd.Function):
    """sum / mean / max / min over the hop list WITH gradients (sum_message_op.py:10, mean_message_op.py:10, max_message_op.py:12,
    min_message_op.py:12): forward = the streaming reduction kernel; backward of sum = the incoming gradient broadcast to every hop,
    of mean = one true division then the broadcast, of max / min = the gradient where torch would select the hop (first NaN, else the
    first extremum), zeros elsewhere (sgl_hop_select_bwd_f32)."""

    @staticmethod
    def forward(ctx, op, divisor, *feats):
        feats_d = [f.detach() for f in feats]
        ctx.op, ctx.divisor = op, divisor
        if op == _lib.SGL_REDUCE_MEAN and divisor is not None and divisor != len(feats_d):
            # the reference divides by (end - start) whatever the slice held (mean_message_op.py:10)
            out = hop_reduce(_lib.SGL_REDUCE_SUM, feats_d)
            out = out / torch.full((), float(divisor), dtype=torch.float32, device=out.device)   # (a fill kernel, not an upload: capturable)
        else:
            out = hop_reduce(op, feats_d)
        if op in (_lib.SGL_REDUCE_MAX, _lib.SGL_REDUCE_MIN):
            ctx.save_for_backward(*feats_d)
        ctx.n_hops = len(feats_d)
        return out

    @staticmethod
    def backward(ctx, gout):
        H = ctx.n_hops
        need = [ctx.needs_input_grad[2 + h] for h in range(H)]
        if ctx.op == _lib.SGL_REDUCE_SUM:
            return (None, None, *[gout if nd else None for nd in need])
        if ctx.op == _lib.SGL_REDUCE_MEAN:
            div = float(ctx.divisor if ctx.divisor is not None else H)
            g = gout / torch.full((), div, dtype=torch.float32, device=gout.device)         # DivBackward: a true division (0-dim device divisor)
            return (None, None, *[g if nd else None for nd in need])
        feats = list(ctx.saved_tensors)
        n, d = feats[0].shape
        g = gout.detach().to(torch.float32)
        if g.stride(1) != 1 or (n > 1 and g.stride(0) < d):
            g = g.contiguous()
        dxs = [alloc_rows(n, d, g.device) if nd else None for nd in need]
        ptrs, lds = _lib.hop_arrays(feats)
        dx_ptrs = (c_void_p * H)(*[(t.data_ptr() if t is not None else None) for t in dxs])
        dx_lds = (c_int64 * H)(*[(_ld(t) if t is not None else 0) for t in dxs])
        with torch.cuda.device(g.device):
            check(lib().sgl_hop_select_bwd_f32(ctx.op, H, ptrs, lds, ptr(g), _ld(g), dx_ptrs, dx_lds, n, d, current_stream_ptr()),
                  "sgl_hop_select_bwd_f32")
        return (None, None, *dxs)


def hop_reduce_grad(op, feats, divisor=None):
    """hop_reduce for hop matrices that carry gradients (op: SGL_REDUCE_SUM / MEAN / MAX / MIN)"""
    return _ReduceGrad.apply(op, divisor, *feats)


class _ConcatGrad(torch.autograd.Function):
    """hstack of the hop list WITH gradients (concat_message_op.py:12; the tail of ProjectedConcatMessageOp._combine,
    projected_concat_message_op.py:28): forward = sgl_hop_concat_f32, backward = the column slices of the incoming gradient (views)."""

    @staticmethod
    def forward(ctx, *feats):
        feats_d = [f.detach() for f in feats]
        ctx.d = feats_d[0].shape[1]
        return hop_concat(feats_d)

    @staticmethod
    def backward(ctx, gout):
        d = ctx.d
        return tuple(gout[:, h * d:(h + 1) * d] if nd else None for h, nd in enumerate(ctx.needs_input_grad))


def hop_concat_grad(feats):
    return _ConcatGrad.apply(*feats)


class _WSum2D(torch.autograd.Function):
    """out[n,:] = sum_h W[n,h] X_h[n,:]  (two_dim_weighted_add, operators/utils.py:105-116)"""

    @staticmethod
    def forward(ctx, w, *feats):
        feats_d = [f.detach() for f in feats]
        _check_hops(feats_d)
        n, d = feats_d[0].shape
        wd = w.detach().to(torch.float32).contiguous()
        if wd.shape != (n, len(feats_d)):
            raise ValueError("The feature list and the weight list have different lengths!")
        result = alloc_rows(n, d, feats_d[0].device)
        wide, out, dw_ = _widened(feats_d, result)
        ptrs, lds = _lib.hop_arrays(wide)
        with torch.cuda.device(out.device):
            check(lib().sgl_hop_wsum2d_f32(len(feats_d), ptrs, lds, ptr(wd), _ld(wd), ptr(out), _ld(out), n, dw_,
                                           current_stream_ptr()), "sgl_hop_wsum2d_f32")
        ctx.save_for_backward(wd, *feats_d)
        return result

    @staticmethod
    def backward(ctx, gout):
        wd, *feats = ctx.saved_tensors
        H = len(feats)
        dw, dxs = _wsum2d_backward(wd, feats, gout, ctx.needs_input_grad[0], [ctx.needs_input_grad[1 + h] for h in range(H)])
        return (dw, *dxs)


def _wsum2d_backward(wd, feats, gout, need_w, need_x):
    """gradients of out = sum_h W[:, h] X_h: dW[n, h] = <dOut[n], X_h[n]> (register row-dot kernel) and dX_h = W[:, h] dOut"""
    n, d = feats[0].shape
    H = len(feats)
    g = gout.detach().to(torch.float32)
    if g.stride(1) != 1 or (n > 1 and g.stride(0) < d):
        g = g.contiguous()
    if n > 1 and d > 4 and (g.stride(0) % 4 != 0 or g.data_ptr() % 16 != 0) and (any(need_x) or H > 16 or d > 512):
        # autograd hands over a dense [n, d] gradient: for d % 4 != 0 its rows are not 16-byte aligned.  The dW row-dot
        # reads such a gradient directly (dword-aligned vector loads, sgl_hop_wsum2d_bwd_f32); the element-wise dX
        # kernel and the wide / many-hop fallback would drop to 4-byte lanes (0.36 of peak at d = 147), so for those
        # one copy into a padded buffer restores the 16-byte path
        gp = alloc_rows(n, d, g.device)
        gp.copy_(g)
        g = gp
    dw = torch.empty((n, H), dtype=torch.float32, device=g.device) if need_w else None
    dxs = [alloc_rows(n, d, g.device) if need_x[h] else None for h in range(H)]
    ptrs, lds = _lib.hop_arrays(feats)
    dx_ptrs = dx_lds = None
    if any(need_x):
        dx_ptrs = (c_void_p * H)(*[(t.data_ptr() if t is not None else None) for t in dxs])
        dx_lds = (c_int64 * H)(*[(_ld(t) if t is not None else 0) for t in dxs])
    with torch.cuda.device(g.device):
        check(lib().sgl_hop_wsum2d_bwd_f32(H, ptrs, lds, ptr(wd), _ld(wd), ptr(g), _ld(g),
                                           ptr(dw) if need_w else None, H, dx_ptrs, dx_lds, n, d,
                                           current_stream_ptr()), "sgl_hop_wsum2d_bwd_f32")
    return dw, dxs


def hop_wsum2d(feats, w):
    return _WSum2D.apply(w, *feats)


class _WSum1D(torch.autograd.Function):
    """out = sum_h w[h] X_h  (one_dim_weighted_add, operators/utils.py:91-102)"""

    @staticmethod
    def forward(ctx, w, *feats):
        feats_d = [f.detach() for f in feats]
        out = hop_reduce(_lib.SGL_REDUCE_WSUM, feats_d, w)
        ctx.save_for_backward(w.detach().to(torch.float32), *feats_d)
        return out

    @staticmethod
    def backward(ctx, gout):
        wd, *feats = ctx.saved_tensors
        n, d = feats[0].shape
        H = len(feats)
        g = gout.detach().to(torch.float32)
        if g.stride(1) != 1 or (n > 1 and g.stride(0) < d):
            g = g.contiguous()
        dw = None
        if ctx.needs_input_grad[0]:
            dw = torch.empty(H, dtype=torch.float32, device=g.device)
            scratch = torch.empty(int(lib().sgl_hop_wsum1d_bwd_scratch(H)), dtype=torch.float32, device=g.device)
            ptrs, lds = _lib.hop_arrays(feats)
            with torch.cuda.device(g.device):
                check(lib().sgl_hop_wsum1d_bwd_f32(H, ptrs, lds, ptr(g), _ld(g), ptr(dw), ptr(scratch), n, d,
                                                   current_stream_ptr()), "sgl_hop_wsum1d_bwd_f32")
        dxs = []
        for h in range(H):
            dxs.append(g * wd[h] if ctx.needs_input_grad[1 + h] else None)  # rarely needed: features carry no grad
        return (dw, *dxs)


def hop_wsum1d(feats, w):
    return _WSum1D.apply(w, *feats)


def hop_colsum(feats, w, shared=False):
    """[H, d] matrix of  sum_n w[n, h] * X_h[n, :]  (shared=True: w is [n], one weight per row for every hop) -- the weight
    gradient of a row-dot, X_h^T g[:, h], for all hops in ONE streaming pass (sgl_hop_colsum_f32) instead of one transposed GEMV per
    hop (rocBLAS gemvn on a padded mini-batch slice: 0.46 ms each at [50 000, 147]).  Falls back to torch where the kernel does
    not apply (rows that are not 16-byte aligned, more than 16 hops, d > 1024, CPU tensors)."""
    feats = [f.detach() for f in feats]
    H = len(feats)
    n, d = feats[0].shape
    w = w.detach().to(torch.float32)
    if (H <= 16 and 0 < d <= 1024 and n > 0 and all(f.is_cuda and f.dtype == torch.float32 and f.stride(1) == 1 and f.data_ptr() % 16 == 0
                                                    and (n == 1 or f.stride(0) % 4 == 0) for f in feats)):
        w = w.contiguous()
        out = torch.empty((H, d), dtype=torch.float32, device=feats[0].device)
        scratch = torch.empty(int(lib().sgl_hop_colsum_scratch(H, n, d)), dtype=torch.float32, device=out.device)
        ptrs, lds = _lib.hop_arrays(feats)
        with torch.cuda.device(out.device):
            check(lib().sgl_hop_colsum_f32(H, ptrs, lds, ptr(w), 1 if shared else H, 0 if shared else 1, ptr(out), d, ptr(scratch),
                                           n, d, current_stream_ptr()), "sgl_hop_colsum_f32")
        return out
    if shared:
        return torch.stack([f.t() @ w.view(-1) for f in feats])
    return torch.stack([f.t() @ w[:, h] for h, f in enumerate(feats)])


class _HopScores(torch.autograd.Function):
    """scores[n, h] = <X_h[n, :], v>: forward is one HIP pass over all hops; backward (mini-batch sized in training) is
    dv = sum_h X_h^T g[:, h] and dX_h = g[:, h] v^T in plain torch."""

    @staticmethod
    def forward(ctx, v, *feats):
        feats_d = [f.detach() for f in feats]
        _check_hops(feats_d)
        n, d = feats_d[0].shape
        H = len(feats_d)
        vp = torch.zeros(round_up(d, 4), dtype=torch.float32, device=feats_d[0].device)
        vp[:d] = v.detach().to(torch.float32).view(-1)
        out = torch.empty((n, H), dtype=torch.float32, device=vp.device)
        ptrs, lds = _lib.hop_arrays(feats_d)
        with torch.cuda.device(vp.device):
            check(lib().sgl_hop_rowdot_f32(H, ptrs, lds, ptr(vp), ptr(out), H, n, d, current_stream_ptr()), "sgl_hop_rowdot_f32")
        ctx.save_for_backward(v.detach(), *feats_d)
        return out

    @staticmethod
    def backward(ctx, g):
        v, *feats = ctx.saved_tensors
        dv = None
        if ctx.needs_input_grad[0]:
            dv = hop_colsum(feats, g).sum(0).view_as(v)            # sum_h X_h^T g[:, h]: one pass over the hops
        dxs = [(g[:, h:h + 1] * v.view(1, -1)) if ctx.needs_input_grad[1 + h] else None for h in range(len(feats))]
        return (dv, *dxs)


def hop_scores(feats, v):
    """[n, H] matrix of <X_h[n, :], v> (differentiable w.r.t. v and the hops)"""
    return _HopScores.apply(v, *feats)


def _padded_vec(v, d, device, tail=None):
    """v zero-padded to whole 16-byte vectors; `tail`: one more float stored right after the padding (the gate's bias: the kernel
    reads it from there, so a bias that is a device tensor never travels to the host)"""
    dp = round_up(max(d, 1), 4)
    vp = torch.zeros(dp + (4 if tail is not None else 0), dtype=torch.float32, device=device)
    vp[:d] = v.detach().to(torch.float32).view(-1)
    if tail is not None:
        vp[dp:dp + 1] = tail.detach().to(device=device, dtype=torch.float32).reshape(-1)[:1]
    return vp


def _gate_launch(v, b, feats_d, outputs):
    """sgl_hop_gate_padded_f32 over detached hops; outputs: also write W and G [n, H] (training / return_weights) -> (out, W, G)"""
    _check_hops(feats_d)
    n, d = feats_d[0].shape
    H = len(feats_d)
    dev_ = feats_d[0].device
    vp = _padded_vec(v, d, dev_, tail=b)                  # [v | 0-pad | bias]: bias = NaN below = "read it from the device"
    result = alloc_rows(n, d, dev_)
    w, g = ((torch.empty((n, H), dtype=torch.float32, device=dev_) for _ in range(2)) if outputs else (None, None))
    ptrs, lds = _lib.hop_arrays(feats_d)
    with torch.cuda.device(dev_):
        check(lib().sgl_hop_gate_padded_f32(H, ptrs, lds, ptr(vp), float("nan"), ptr(result), _ld(result), own_pad(result),
                                            ptr(w) if outputs else None, H, ptr(g) if outputs else None, H, n, d,
                                            current_stream_ptr()), "sgl_hop_gate_padded_f32")
    return result, w, g


class _GateFused(torch.autograd.Function):
    """out = sum_h softmax_h(sigmoid(<X_h, v> + b)) X_h in ONE pass over the hops (sgl_hop_gate_f32); the backward re-uses the
    dW row-dot kernel and finishes the [n, H]-sized softmax / sigmoid chain in torch."""

    @staticmethod
    def forward(ctx, v, b, *feats):
        feats_d = [f.detach() for f in feats]
        result, w, g = _gate_launch(v, b, feats_d, True)
        ctx.save_for_backward(v.detach(), w, g, *feats_d)
        ctx.b_shape = tuple(b.shape)
        ctx.mark_non_differentiable(w)
        return result, w

    @staticmethod
    def backward(ctx, gout, _gw):
        v, w, g, *feats = ctx.saved_tensors
        H = len(feats)
        need_x = [ctx.needs_input_grad[2 + h] for h in range(H)]
        dwt, dxs = _wsum2d_backward(w, feats, gout, True, need_x)                 # dL/dW and the W-part of dL/dX_h
        dg = w * (dwt - (w * dwt).sum(dim=1, keepdim=True))                       # softmax
        ds = dg * g * (1.0 - g)                                                   # sigmoid
        dv = db = None
        if ctx.needs_input_grad[0]:
            dv = hop_colsum(feats, ds).sum(0).view_as(v)           # sum_h X_h^T ds[:, h]: one pass over the hops
        if ctx.needs_input_grad[1]:
            db = ds.sum().reshape(ctx.b_shape)
        for h in range(H):
            if need_x[h]:
                dxs[h] = dxs[h] + ds[:, h:h + 1] * v.view(1, -1)
        return (dv, db, *dxs)


def gate_fusable(feats):
    """can sgl_hop_gate_f32 / sgl_hop_rowdot2_f32 take these hops? (register-resident rows: <= 16 hops, d <= 512, 16-byte rows)"""
    f0 = feats[0]
    return (len(feats) <= 16 and f0.is_cuda and f0.dtype == torch.float32 and 0 < f0.shape[1] <= 512 and f0.shape[0] > 0 and
            all(f.stride(1) == 1 and f.data_ptr() % 16 == 0 and (f.shape[0] == 1 or f.stride(0) % 4 == 0) for f in feats))


def hop_gate(feats, v, b, return_weights=False):
    """LearnableWeightedMessageOp 'gate': (out, W) with W = softmax_h(sigmoid(Linear(X_h))) [n, H].  One pass over the hops
    (sgl_hop_gate_f32) when their rows fit the register-resident kernel (gate_fusable); otherwise the two-pass route -- one
    row-dot pass for the scores, the [n, H] sigmoid / softmax in torch, one weighted-sum pass."""
    if gate_fusable(feats):
        if torch.is_grad_enabled() and any(t.requires_grad for t in (v, b, *feats)):
            out, w = _GateFused.apply(v, b, *feats)
        else:                                             # inference: the [n, H] matrices are written only when asked for
            out, w, _ = _gate_launch(v, b, [f.detach() for f in feats], return_weights)
    else:
        w = torch.softmax(torch.sigmoid(hop_scores(feats, v) + b.reshape(-1)[0]), dim=1)
        out = hop_wsum2d(feats, w)
    return (out, w) if return_weights else out


def recursive_weights(a, c, b):
    """The recursion of IterateLearnableWeightedMessageOp (iterate_learnable_weighted_message_op.py:35-41) on per-hop scalars:
    a[n, h] = <X_h, w_x>, c[n, h] = <X_h, w_acc>, b the Linear's bias -> the final soft-max weights [n, H].  acc is always a
    per-row weighted sum of the hops, so <acc_{i-1}, w_acc> = sum_{j<i} W[:, j] c[:, j]; plain torch, differentiable."""
    H = a.shape[1]
    dot_acc = c[:, 0]                                    # acc starts as X_0
    weights = None
    for i in range(H):
        score = torch.sigmoid(a[:, i] + dot_acc + b).unsqueeze(1)
        weights = score if weights is None else torch.hstack((weights, score))
        weights = F.softmax(weights, dim=1)              # the earlier columns are soft-maxed again (reference :39 -- kept)
        if i + 1 < H:
            dot_acc = (weights * c[:, :i + 1]).sum(dim=1)
    return weights


def _recursive_launch(weight, b, feats_d, outputs):
    """sgl_hop_recursive_f32 over detached hops; outputs: also write W, A, C [n, H] (training / return_weights) -> (out, vp, W, A, C)"""
    _check_hops(feats_d)
    n, d = feats_d[0].shape
    H = len(feats_d)
    dev_ = feats_d[0].device
    dp = round_up(d, 4)
    wt = weight.detach().to(device=dev_, dtype=torch.float32).reshape(-1)
    if wt.numel() != 2 * d:
        raise ValueError(f"the recursive gate's Linear has {wt.numel()} weights, the hops need 2 * {d}")
    vp = torch.zeros(2 * dp + 4, dtype=torch.float32, device=dev_)          # [w_x | pad | w_acc | pad | bias]
    vp[:d] = wt[:d]
    vp[dp:dp + d] = wt[d:]
    vp[2 * dp:2 * dp + 1] = b.detach().to(device=dev_, dtype=torch.float32).reshape(-1)[:1]
    result = alloc_rows(n, d, dev_)
    w, a, c = ((torch.empty((n, H), dtype=torch.float32, device=dev_) for _ in range(3)) if outputs else (None, None, None))
    ptrs, lds = _lib.hop_arrays(feats_d)
    with torch.cuda.device(dev_):
        # bias = NaN: "read it from the device, after the two padded vectors" (no host synchronisation on a parameter)
        check(lib().sgl_hop_recursive_f32(H, ptrs, lds, ptr(vp), float("nan"), ptr(result), _ld(result), own_pad(result),
                                          ptr(w) if outputs else None, H, ptr(a) if outputs else None, H,
                                          ptr(c) if outputs else None, H, n, d, current_stream_ptr()), "sgl_hop_recursive_f32")
    return result, vp, w, a, c


class _RecursiveFused(torch.autograd.Function):
    """GAMLP-R's recursive gate in ONE pass over the hops (sgl_hop_recursive_f32): out, final weights W [n, H] and the score
    matrices A, C.  Backward: dL/dW by the row-dot kernel, the [n, H] recursion backwards in one thread-per-row kernel
    (sgl_hop_recursive_bwd_f32), the weight gradients by the column-sum kernel."""

    @staticmethod
    def forward(ctx, weight, b, *feats):
        feats_d = [f.detach() for f in feats]
        result, vp, w, a, c = _recursive_launch(weight, b, feats_d, True)
        ctx.save_for_backward(vp, w, a, c, *feats_d)
        ctx.shapes = (tuple(weight.shape), tuple(b.shape))
        ctx.mark_non_differentiable(w)
        return result, w

    @staticmethod
    def backward(ctx, gout, _gw):
        vp, w, a, c, *feats = ctx.saved_tensors
        H = len(feats)
        n, d = feats[0].shape
        dp = round_up(d, 4)
        need_x = [ctx.needs_input_grad[2 + h] for h in range(H)]
        dwt, dxs = _wsum2d_backward(w, feats, gout, True, need_x)                 # dL/dW and the W-part of dL/dX_h
        dwt = dwt.contiguous()
        da, dc = torch.empty_like(a), torch.empty_like(c)
        dbr = torch.empty(n, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            check(lib().sgl_hop_recursive_bwd_f32(H, ptr(a), H, ptr(c), H, float("nan"), ptr(vp[2 * dp:]), ptr(dwt), H, ptr(da), H,
                                                  ptr(dc), H, ptr(dbr), n, current_stream_ptr()), "sgl_hop_recursive_bwd_f32")
        dweight = dbias = None
        if ctx.needs_input_grad[0]:
            dweight = torch.cat([hop_colsum(feats, da).sum(0), hop_colsum(feats, dc).sum(0)]).reshape(ctx.shapes[0])
        if ctx.needs_input_grad[1]:
            dbias = dbr.sum().reshape(ctx.shapes[1])
        v_x, v_acc = vp[:d].view(1, -1), vp[dp:dp + d].view(1, -1)
        for h in range(H):
            if need_x[h]:
                dxs[h] = dxs[h] + da[:, h:h + 1] * v_x + dc[:, h:h + 1] * v_acc
        return (dweight, dbias, *dxs)


def hop_recursive(feats, weight, b, return_weights=False):
    """IterateLearnableWeightedMessageOp 'recursive' over hops 0 .. H-1 with Linear(2 d -> 1) parameters (weight [1, 2 d] or
    [2 d]: [w_x | w_acc], bias b).  One pass over the hops when the rows fit the register-resident kernel (gate_fusable);
    otherwise two row-dot passes for the scores, the [n, H] recursion in torch and one weighted-sum pass."""
    if gate_fusable(feats):
        if torch.is_grad_enabled() and any(t.requires_grad for t in (weight, b, *feats)):
            out, w = _RecursiveFused.apply(weight, b, *feats)
        else:                                             # inference: the [n, H] matrices are written only when asked for
            out, _, w, _, _ = _recursive_launch(weight, b, [f.detach() for f in feats], return_weights)
    else:
        d = feats[0].shape[1]
        wt = weight.reshape(-1)
        w = recursive_weights(hop_scores(feats, wt[:d]), hop_scores(feats, wt[d:]), b.reshape(-1)[:1])
        out = hop_wsum2d(feats, w)
    return (out, w) if return_weights else out


class _HopScores2(torch.autograd.Function):
    """P[n, h - h0] = <X_h[n], v> for h in [h0, h1) and A[n] = sum_{j in ref} <X_j[n], U[j]>: the two parts of the 'ori_ref' /
    'jk' Linear([ref || x_h]) in one pass over the hop list (sgl_hop_rowdot2_f32).  Backward in plain torch (mini-batch sized)."""

    @staticmethod
    def forward(ctx, v, u, mask, h0, h1, *feats):
        feats_d = [f.detach() for f in feats]
        _check_hops(feats_d)
        n, d = feats_d[0].shape
        L = len(feats_d)
        dev_ = feats_d[0].device
        vp = _padded_vec(v, d, dev_)
        ldu = round_up(d, 4)
        up = torch.zeros((L, ldu), dtype=torch.float32, device=dev_)
        up[:, :d] = u.detach().to(torch.float32).view(L, d)
        p = torch.empty((n, h1 - h0), dtype=torch.float32, device=dev_)
        a = torch.empty(n, dtype=torch.float32, device=dev_)
        ptrs, lds = _lib.hop_arrays(feats_d)
        with torch.cuda.device(dev_):
            check(lib().sgl_hop_rowdot2_f32(L, ptrs, lds, ptr(up), ldu, ctypes.c_uint64(mask), ptr(vp), h0, h1, ptr(p), h1 - h0,
                                            ptr(a), n, d, current_stream_ptr()), "sgl_hop_rowdot2_f32")
        ctx.save_for_backward(v.detach(), u.detach(), *feats_d)
        ctx.meta = (mask, h0, h1)
        return p, a

    @staticmethod
    def backward(ctx, gp, ga):
        v, u, *feats = ctx.saved_tensors
        mask, h0, h1 = ctx.meta
        L = len(feats)
        d = feats[0].shape[1]
        dv = du = None
        if ctx.needs_input_grad[0]:
            dv = (hop_colsum(feats[h0:h1], gp).sum(0) if h1 > h0 else torch.zeros(d, dtype=torch.float32, device=gp.device)).view_as(v)
        if ctx.needs_input_grad[1]:
            du = torch.zeros((L, d), dtype=torch.float32, device=gp.device)
            ref = [j for j in range(L) if (mask >> j) & 1]
            if ref:                                                    # X_j^T ga for every hop of the reference part: one pass
                part = hop_colsum([feats[j] for j in ref], ga, shared=True)
                for k_, j in enumerate(ref):                           # (row by row: a list index would be uploaded, which a
                    du[j] = part[k_]                                   # HIP-graph capture of the training step does not allow)
            du = du.view_as(u)
        dxs = []
        uu = u.view(L, d)
        for j in range(L):
            if not ctx.needs_input_grad[5 + j]:
                dxs.append(None)
                continue
            gx = torch.zeros_like(feats[j])
            if h0 <= j < h1:
                gx = gx + gp[:, j - h0:j - h0 + 1] * v.view(1, -1)
            if (mask >> j) & 1:
                gx = gx + ga.view(-1, 1) * uu[j].view(1, -1)
            dxs.append(gx)
        return (dv, du, None, None, None, *dxs)


def hop_scores2(feats, v, u, mask, h0, h1):
    """(P [n, h1 - h0], A [n]): per-hop scores <X_h, v> of the adopted hops and the shared reference term sum_j <X_j, U[j]>"""
    return _HopScores2.apply(v, u, int(mask), int(h0), int(h1), *feats)


def nafs_aggregate(feats, return_weights=False):
    """OverSmoothDistanceWeightedOp._combine (over_smooth_distance_op.py:11-33) on device"""
    _check_hops(feats)
    n, d = feats[0].shape
    H = len(feats)
    out = alloc_rows(n, d, feats[0].device)
    w = torch.empty((n, H), dtype=torch.float32, device=feats[0].device)
    ptrs, lds = _lib.hop_arrays(feats)
    with torch.cuda.device(out.device):
        check(lib().sgl_nafs_padded_f32(H, ptrs, lds, ptr(out), _ld(out), own_pad(out), ptr(w), H, n, d, current_stream_ptr()),
              "sgl_nafs_padded_f32")
    return (out, w) if return_weights else out


NAFS_STORE, NAFS_ADD, NAFS_ADD_DIV, NAFS_MAX = 0, 1, 2, 3


def nafs_prefix(feats, emit_hops, outs=None, combine=NAFS_STORE, divisor=1.0, outs_padded=False):
    """The over-smoothing-distance aggregate (OverSmoothDistanceWeightedOp / node_clustering.py:218-241) of EVERY requested prefix
    X_0..X_h of the hop list in one pass over the hop matrices (sgl_nafs_prefix_f32): out[k] = NAFS(feats[:emit_hops[k] + 1]).
    emit_hops: increasing hop indices < len(feats).  outs: matrices from alloc_rows to write / combine into (the multi-r
    ensemble: NAFS_ADD, NAFS_ADD_DIV with `divisor`, NAFS_MAX), allocated when None; any 16-byte aligned [n, d] views whose pitch
    is a multiple of 4 floats -- e.g. column slices of one wide slab (the 'concat' ensemble).  outs_padded=True declares that the
    tail of every output's pitch is its own padding (matrices from alloc_rows): it is then written as zeros so that every line of a
    row is written whole.  Returns the list of outputs."""
    _check_hops(feats)
    n, d = feats[0].shape
    emit_hops = [int(h) for h in emit_hops]
    if not emit_hops or any(b <= a for a, b in zip(emit_hops, emit_hops[1:])) or emit_hops[0] < 0 or emit_hops[-1] >= len(feats):
        raise ValueError("emit_hops must be increasing hop indices below len(feats)")
    feats = feats[:emit_hops[-1] + 1]
    if outs is None:
        if combine != NAFS_STORE:
            raise ValueError("combining needs the matrices to combine with")
        outs = [alloc_rows(n, d, feats[0].device) for _ in emit_hops]
        outs_padded = True
    if len(outs) != len(emit_hops) or any(tuple(o.shape) != (n, d) or o.dtype != torch.float32 or o.device != feats[0].device for o in outs):
        raise ValueError("one [n, d] float32 output per requested prefix")
    pads = {own_pad(o) for o in outs} if outs_padded else {0}
    pad = pads.pop() if len(pads) == 1 else 0
    mask = 0
    for h in emit_hops:
        mask |= 1 << h
    ptrs, lds = _lib.hop_arrays(feats)
    optrs, olds = _lib.hop_arrays(outs)
    with torch.cuda.device(feats[0].device):
        check(lib().sgl_nafs_prefix_f32(len(feats), ptrs, lds, mask, optrs, olds, pad, int(combine), float(divisor), n, d,
                                        current_stream_ptr()), "sgl_nafs_prefix_f32")
    _wrote(*outs)
    return outs


def scatter_rows(x, src, dst, out):
    """out[dst[i]] = x[src[i]] on device (sgl_scatter_rows_f32; src, dst: int64 CUDA tensors of one length, dst entries distinct).
    The pack step of the need-aware exchange: pairs sorted by source row, so a row several peers gather is read once."""
    _check_mat(x, "x")
    _check_mat(out, "out")
    if x.shape[1] != out.shape[1]:
        raise ValueError("scatter_rows: x and out must have the same row length")
    for t, nm in ((src, "src"), (dst, "dst")):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int64 and t.dim() == 1 and t.is_contiguous()):
            raise TypeError(f"scatter_rows: {nm} must be a contiguous 1-D int64 CUDA tensor")
    if src.numel() != dst.numel():
        raise ValueError("scatter_rows: src and dst must have the same length")
    if src.numel() == 0:
        return out
    with torch.cuda.device(x.device):
        check(lib().sgl_scatter_rows_f32(ptr(x), _ld(x), x.shape[0], ptr(src), ptr(dst), src.numel(), ptr(out), _ld(out),
                                         out.shape[0], x.shape[1], current_stream_ptr()), "sgl_scatter_rows_f32")
    _wrote(out)
    return out


def _device_index(idx, n_rows, device):
    """row indices of any kind as a flat int64 tensor on `device`"""
    if not (torch.is_tensor(idx) and idx.is_cuda):
        # host indices (range / list / ndarray / CPU tensor): validate here, like torch's CPU indexing does
        if isinstance(idx, range):
            idx = np.arange(idx.start, idx.stop, idx.step, dtype=np.int64)
        host = idx.numpy() if torch.is_tensor(idx) else np.asarray(idx)
        if host.dtype == np.bool_:
            host = np.nonzero(host)[0]
        host = host.astype(np.int64, copy=False).reshape(-1)
        if host.size and (host.min() < -n_rows or host.max() >= n_rows):
            raise IndexError("index out of range in row gather")
        idx = torch.from_numpy(np.ascontiguousarray(host))
    # device indices are range-checked inside the kernel (it traps on a bad index): no host round trip
    return idx.to(device=device, dtype=torch.int64).contiguous().view(-1)


def column_signature(x):
    """per-column content signatures of a [n, d] float32 device matrix with 16-byte rows (sgl_col_signature_f32): an int64 device
    tensor of round_up(d, 4) entries (the bits of the 64-bit sums), or None when the rows are not 16-byte vectors of a pitch that
    covers round_up(d, 4) columns.  Two matrices of one shape differ in column c (almost surely) iff their signatures differ at c."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return None
    n, d = x.shape
    dw = round_up(d, 4)
    if n == 0 or d == 0:
        return None
    ld = x.stride(0) if n > 1 else dw
    if x.stride(1) != 1 or ld % 4 or ld < dw or x.data_ptr() % 16:
        return None
    if x.untyped_storage().nbytes() // 4 - x.storage_offset() < (n - 1) * ld + dw:
        return None
    sig = torch.empty(dw, dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().sgl_col_signature_f32(ptr(x), ld, n, d, ptr(sig), current_stream_ptr()), "sgl_col_signature_f32")
    return sig


def gather_hops(feats, idx, one_launch=None):
    """[x[idx] for x in feats]: the training feed of the learnable aggregators, `[feat[idx].to(device) for feat in
    self._processed_feat_list]` (sgl/models/base_model.py:58-60), with the indices validated and uploaded ONCE for all hop
    matrices.  Host indices (what the reference's tasks pass) are where the time of the hop-by-hop form goes: 200 000 rows of 4 /
    6 / 11 hop matrices 0.46 / 0.78 / 1.25 ms -> 0.24 / 0.42 / 0.55 ms; with device indices the H queued launches already run back
    to back, and one launch over all hops (measured: profiles/r05_gather_hops.txt) is no faster."""
    feats = list(feats)
    if not feats:
        return []
    _check_mat(feats[0], "feat_list[0]")
    same_rows = all(torch.is_tensor(f) and f.is_cuda and f.dim() == 2 and f.shape[0] == feats[0].shape[0] and f.device == feats[0].device
                    for f in feats)
    if same_rows:
        idx = _device_index(idx, feats[0].shape[0], feats[0].device)
    one = _gather_hops_one_launch(feats, idx) if (same_rows and one_launch is not False and len(feats) > 1) else None
    if one is not None:
        return one
    return [gather_rows(x, idx) for x in feats]


def _gather_hops_one_launch(feats, idx):
    """sgl_gather_hops_padded_f32 when every hop matrix is a float32 [n, d] matrix of 16-byte rows (what propagate() returns), else None"""
    n_rows, d = feats[0].shape
    m = int(idx.numel())
    dp = round_up(d, 4)
    if m < 2 or n_rows < 2 or len(feats) > 16:
        return None
    for x in feats:
        if not (x.dtype == torch.float32 and x.shape == (n_rows, d) and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.stride(0) >= dp
                and x.data_ptr() % 16 == 0 and x.untyped_storage().nbytes() // 4 - x.storage_offset() >= (n_rows - 1) * x.stride(0) + dp):
            return None
    outs = [alloc_rows(m, d, feats[0].device, zero_pad=False) for _ in feats]
    ldo = outs[0].stride(0)
    if ldo % 4 or ldo < dp or any(o.data_ptr() % 16 for o in outs):
        return None
    pad = (ldo - d) if ldo - d < 32 else (dp - d)
    if pad != ldo - d:
        for o in outs:
            padded_parent(o)[:, d + pad:].zero_()
    ptrs, lds = _lib.hop_arrays(feats)
    optrs, olds = _lib.hop_arrays(outs)
    with torch.cuda.device(feats[0].device):
        check(lib().sgl_gather_hops_padded_f32(len(feats), ptrs, lds, n_rows, ptr(idx), m, optrs, olds, d, pad, current_stream_ptr()),
              "sgl_gather_hops_padded_f32")
    return outs


def gather_rows(x, idx, out=None):
    """x[idx] on device (BaseSGAPModel.forward's per-step row gather, models/base_model.py:58,60).  `out`: optional
    preallocated [len(idx), d] destination (the pack step of the need-aware exchange re-uses one send buffer per hop)."""
    _check_mat(x, "x")
    n_rows, d = x.shape
    idx = _device_index(idx, n_rows, x.device)
    own_out = out is None
    if out is None:
        # (the pad columns of our own output are written by the kernel below whenever the vector path applies: no separate zero fill)
        fast = (n_rows > 1 and idx.numel() > 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and 0 < row_pitch(d) - d < 32
                and x.stride(0) >= round_up(d, 4)
                and x.untyped_storage().nbytes() // 4 - x.storage_offset() >= (n_rows - 1) * x.stride(0) + round_up(d, 4))
        out = alloc_rows(idx.numel(), d, x.device, zero_pad=not fast)
    else:
        _check_mat(out, "out")
        if out.shape != (idx.numel(), d):
            raise ValueError("gather_rows: `out` must be [len(idx), d]")
    if idx.numel() == 0:
        return out
    # 16-byte lanes for any d: the vector that straddles column d is READ from the source (its pitch is a multiple of 4 floats, and the
    # storage must hold it for the last row too) but only the d data columns reach the result; the columns of the destination's
    # padding that get written are written as zeros (sgl_gather_rows_padded_f32) -- never the source's tail, which may be real
    # data when x is a column view of a wider matrix.  Our own output: its whole pitch is padding we own, so every line of a row is
    # written whole (a row of 147 floats on a 160-float pitch would otherwise end in a partly written line: a read-modify-write in
    # HBM); a caller's output: up to the next multiple of 4, as before.
    dp = round_up(d, 4)
    ldx, ldo = x.stride(0) if n_rows > 1 else max(x.stride(0), d), out.stride(0) if idx.numel() > 1 else max(out.stride(0), d)
    room = x.untyped_storage().nbytes() // 4 - x.storage_offset() >= (n_rows - 1) * ldx + dp
    vec_ok = (n_rows > 1 and idx.numel() > 1 and ldx % 4 == 0 and ldx >= dp and ldo % 4 == 0 and ldo >= dp and room
              and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0)
    pad = 0
    if vec_ok:
        pad = (ldo - d) if (own_out and ldo - d < 32) else (dp - d)
    with torch.cuda.device(x.device):
        check(lib().sgl_gather_rows_padded_f32(ptr(x), _ld(x), n_rows, ptr(idx), idx.numel(), ptr(out), _ld(out), d, pad,
                                               current_stream_ptr()), "sgl_gather_rows_padded_f32")
    _wrote(out)
    return out
