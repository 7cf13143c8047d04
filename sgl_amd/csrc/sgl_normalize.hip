// adj_to_symmetric_norm on device (reference: sgl/operators/utils.py:76-88 and the Laplacian / PPR wrappers
// graph_op/laplacian_graph_op.py:12-19, graph_op/ppr_graph_op.py:13-21).
//
//   A' = A + I ; deg = rowsum(A') ; L = deg^(r-1), R = deg^(-r) (inf -> 0)
//   A_hat = (A' diag(L))^T diag(R)          =>  A_hat[j,i] = (A'[i,j] * L[j]) * R[i]      (fp64, this order)
//   PPR:  (1-alpha) * A_hat + alpha * I
//   result as canonical CSR (rows sorted by column), values rounded to fp32 at the very end -- the place the
//   reference rounds (utils.py:32).
//
// Integer work (structure) is exact; the fp64 values follow the reference's operation order.  The transpose is a
// stable LSD radix sort by output row (rocPRIM), so each output row keeps ascending column order without atomics.
// This is a once-per-graph setup step (HBM-streaming + one sort), not the per-hop hot loop.
#include "sgl_common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

__global__ __launch_bounds__(256) void diag_missing_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                           int64_t n, int64_t *__restrict__ miss, unsigned long long *total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int m = 0;
    if (i < n) {
        int64_t lo = rowptr[i], hi = rowptr[i + 1];
        while (lo < hi) {  // lower_bound(col[row], i)
            const int64_t mid = (lo + hi) >> 1;
            if (col[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
        }
        m = !(lo < rowptr[i + 1] && col[lo] == (int32_t)i);
        miss[i] = m;
    } else if (i == n) {
        miss[i] = 0;
    }
    if (total) {
        // wave-level count, one atomic per wave
        const unsigned long long b = __ballot(m != 0);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(total, (unsigned long long)__popcll(b));
    }
}

// rows of A' = A + I in row-major order: key = column (the OUTPUT row after transposition), row = i, value fp64
__global__ __launch_bounds__(256) void build_aprime_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                           const float *__restrict__ val, const int64_t *__restrict__ shift,
                                                           int64_t n, uint32_t *__restrict__ t_key, int32_t *__restrict__ t_row,
                                                           double *__restrict__ a64, double *__restrict__ deg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = rowptr[i], e = rowptr[i + 1];
    int64_t o = b + shift[i];
    double s = 0.0;
    bool placed = false;
    for (int64_t p = b; p < e; ++p) {
        const int32_t c = col[p];
        double v = (double)val[p];
        if (!placed && c >= (int32_t)i) {
            placed = true;
            if (c == (int32_t)i) {
                v += 1.0;  // a_ii + 1
            } else {       // insert the missing diagonal before the first larger column
                t_key[o] = (uint32_t)i;
                t_row[o] = (int32_t)i;
                a64[o] = 1.0;
                s += 1.0;
                ++o;
            }
        }
        t_key[o] = (uint32_t)c;
        t_row[o] = (int32_t)i;
        a64[o] = v;
        s += v;
        ++o;
    }
    if (!placed) {
        t_key[o] = (uint32_t)i;
        t_row[o] = (int32_t)i;
        a64[o] = 1.0;
        s += 1.0;
    }
    deg[i] = s;
}

__global__ __launch_bounds__(256) void degree_scale_kernel(const double *__restrict__ deg, int64_t n, double r,
                                                           double *__restrict__ left, double *__restrict__ right) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dg = deg[i];
    double l = pow(dg, r - 1.0);   // utils.py:79
    if (isinf(l)) l = 0.0;         // utils.py:80
    double rr = pow(dg, -r);       // utils.py:83
    if (isinf(rr)) rr = 0.0;       // utils.py:84
    left[i] = l;
    right[i] = rr;
}

__global__ __launch_bounds__(256) void scale_values_kernel(const uint32_t *__restrict__ t_key, const int32_t *__restrict__ t_row,
                                                           const double *__restrict__ left, const double *__restrict__ right,
                                                           int64_t m, double *__restrict__ a64, uint32_t *__restrict__ iota) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    a64[p] = __dmul_rn(__dmul_rn(a64[p], left[t_key[p]]), right[t_row[p]]);  // (A'[i,j] * L[j]) * R[i]
    iota[p] = (uint32_t)p;
}

__global__ __launch_bounds__(256) void rowptr_from_sorted_kernel(const uint32_t *__restrict__ keys, int64_t m, int64_t n,
                                                                 int64_t *__restrict__ out_rowptr) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    int64_t lo = 0, hi = m;
    while (lo < hi) {  // first q with keys[q] >= j
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)keys[mid] < j) lo = mid + 1; else hi = mid;
    }
    out_rowptr[j] = lo;
}

__global__ __launch_bounds__(256) void finalize_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ perm,
                                                       const int32_t *__restrict__ t_row, const double *__restrict__ a64,
                                                       int64_t m, int use_alpha, double one_minus_alpha, double alpha,
                                                       int32_t *__restrict__ out_col, float *__restrict__ out_val,
                                                       double *__restrict__ out_val64) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const uint32_t p = perm[q];
    const int32_t i = t_row[p];
    double v = a64[p];
    if (use_alpha) {                       // ppr_graph_op.py:20
        v = __dmul_rn(one_minus_alpha, v);
        if ((uint32_t)i == keys[q]) v = __dadd_rn(v, alpha);
    }
    out_col[q] = i;
    out_val[q] = (float)v;                 // operators/utils.py:32
    if (out_val64) out_val64[q] = v;
}

// weighted degrees of A + I only (same summation order as build_aprime_kernel): lets the host evaluate the two degree
// powers with the very libm the reference's numpy calls, so the rounded A_hat is bit-identical to scipy's
__global__ __launch_bounds__(256) void degrees_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      const float *__restrict__ val, int64_t n, int64_t row0,
                                                      double *__restrict__ deg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t me = (int32_t)(row0 + i);
    double s = 0.0;
    bool placed = false;
    for (int64_t p = rowptr[i], e = rowptr[i + 1]; p < e; ++p) {
        const int32_t c = col[p];
        double v = (double)val[p];
        if (!placed && c >= me) {
            placed = true;
            if (c == me) v += 1.0;
            else s += 1.0;
        }
        s += v;
    }
    if (!placed) s += 1.0;
    deg[i] = s;
}

// ---- row-block normalisation (multi-GPU: every rank owns rows [row0, row0 + n) of A_hat) ---------------------------------
// rows of T' = T + I for a block of rows of T (global column ids), as CSR with the diagonal merged / inserted in sorted
// position; fp64 values; row sums
__device__ __forceinline__ uint64_t sym_mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// T' = T + I for rows [row0, row0 + n): the diagonal is merged into / inserted at its sorted position.  A block of 256 threads
// owns 256 consecutive rows: phase 1 streams their non-zeros in order (coalesced reads and writes; the row of an element by
// binary search over the row pointers in LDS), phase 2 has thread t place row t's missing diagonal and add up row t
// SEQUENTIALLY in column order (the fp64 sum the reference's scipy forms) from the values just written (L2-resident).
constexpr int kBuildRows = 256;

__global__ __launch_bounds__(256) void block_build_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          const float *__restrict__ val, const int64_t *__restrict__ shift,
                                                          int64_t n, int64_t row0, int64_t *__restrict__ o_rowptr,
                                                          int32_t *__restrict__ o_col, double *__restrict__ o_val,
                                                          double *__restrict__ rowsum, unsigned long long *sym_hash) {
    // row pointers and diagonal shifts of the block's rows, RELATIVE to its first row, in LDS (32-bit: a block of 256 rows of a
    // matrix with < 2^32 non-zeros per row block)
    __shared__ uint32_t rp[kBuildRows + 1], sh[kBuildRows + 1];
    const int64_t r_begin = (int64_t)blockIdx.x * kBuildRows;
    const int rows = (int)min((int64_t)kBuildRows, n - r_begin);
    const int64_t p0 = rowptr[r_begin], s0 = shift[r_begin];
    for (int t = threadIdx.x; t <= rows; t += 256) {
        rp[t] = (uint32_t)(rowptr[r_begin + t] - p0);
        sh[t] = (uint32_t)(shift[r_begin + t] - s0);
    }
    __syncthreads();
    // symmetry fingerprint (whole matrices only): every off-diagonal entry (i, j, v) adds +h(min, max, v) if i < j and
    // -h(...) if i > j, in wrapping 64-bit arithmetic: the sum over a matrix with A[i,j] == A[j,i] bit for bit is 0, and
    // it is non-zero for any other matrix except with probability 2^-64
    uint64_t fp = 0;
    const uint32_t cnt = rp[rows];
    const int32_t *colb = col + p0;
    const float *valb = val + p0;
    int32_t *ocol = o_col + p0 + s0;
    double *oval = o_val + p0 + s0;
    constexpr int U = 4;                                         // independent element loads in flight per thread
    for (uint32_t q0 = threadIdx.x; q0 < cnt; q0 += 256 * U) {
        int32_t c[U];
        float vf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + 256u * u;
            c[u] = q < cnt ? __builtin_nontemporal_load(colb + q) : 0;
            vf[u] = q < cnt ? __builtin_nontemporal_load(valb + q) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + 256u * u;
            if (q >= cnt) continue;
            int lo = 0, hi = rows;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (rp[mid] <= q) lo = mid; else hi = mid;
            }
            const int32_t me = (int32_t)(row0 + r_begin + lo);
            double v = (double)vf[u];
            const bool miss = sh[lo + 1] != sh[lo];             // this row gets a diagonal entry inserted
            const uint32_t o = q + sh[lo] + ((miss && c[u] > me) ? 1u : 0u);
            if (c[u] == me) v += 1.0;                            // a_ii + 1
            ocol[o] = c[u];
            oval[o] = v;
            if (sym_hash && c[u] != me) {
                const uint64_t lo_ = (uint64_t)(c[u] < me ? c[u] : me), hi_ = (uint64_t)(c[u] < me ? me : c[u]);
                const uint64_t h = sym_mix(sym_mix(lo_ * 0x100000001B3ull + hi_) ^ (uint64_t)__float_as_uint(vf[u]));
                fp += (me < c[u]) ? h : (0ull - h);
            }
        }
    }
    if (sym_hash && fp) atomicAdd(sym_hash, (unsigned long long)fp);
    const int t = threadIdx.x;
    if (t < rows) {
        const int32_t me = (int32_t)(row0 + r_begin + t);
        o_rowptr[r_begin + t] = p0 + s0 + rp[t] + sh[t];
        if (sh[t + 1] != sh[t]) {                                // insert the diagonal before the first larger column
            uint32_t lo = rp[t], hi = rp[t + 1];
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (colb[mid] < me) lo = mid + 1; else hi = mid;
            }
            ocol[lo + sh[t]] = me;
            oval[lo + sh[t]] = 1.0;
        }
    }
    if (r_begin + rows == n && t == 0) o_rowptr[n] = p0 + s0 + rp[rows] + sh[rows];
    __syncthreads();                                             // the block's rows of T' are complete (same CU wrote them)
    // Row sums, SEQUENTIAL in column order (the fp64 sum scipy forms).  A thread walking its row alone pays one dependent L2 round
    // trip per element: fine for the usual few dozen, but the launch then lasts as long as its LONGEST row -- 17 481 elements x
    // ~0.2 us = 3.4 ms at the products shape, the whole kernel (profiles/r05_all_kernels_stats.csv).  Rows beyond 64 elements are
    // therefore summed by a whole wavefront: 64 consecutive values in one coalesced load, then the same sequential chain of
    // additions fed from registers (v_readlane) instead of from memory -- the same operations in the same order.
    constexpr uint32_t kCoopRow = 64;
    if (t < rows) {
        const uint32_t ob = rp[t] + sh[t], oe = rp[t + 1] + sh[t + 1];
        if (oe - ob <= kCoopRow) {
            double s = 0.0;
            for (uint32_t q = ob; q < oe; ++q) s += __builtin_nontemporal_load(oval + q);
            rowsum[r_begin + t] = s;
        }
    }
    const int lane = t & 63;
    for (int r = t >> 6; r < rows; r += 4) {                     // uniform per wavefront
        const uint32_t ob = rp[r] + sh[r], oe = rp[r + 1] + sh[r + 1];
        if (oe - ob <= kCoopRow) continue;
        double s = 0.0;
        for (uint32_t q = ob; q < oe; q += 64) {
            const double x = (q + lane < oe) ? oval[q + lane] : 0.0;
            const int m = (int)min(64u, oe - q);
            const int xl = __double2loint(x), xh = __double2hiint(x);
            for (int j = 0; j < m; ++j)
                s += __hiloint2double(__builtin_amdgcn_readlane(xh, j), __builtin_amdgcn_readlane(xl, j));
        }
        if (lane == 0) rowsum[r_begin + r] = s;
    }
}

__global__ __launch_bounds__(256) void block_diag_missing_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                 int64_t n, int64_t row0, int64_t *__restrict__ miss,
                                                                 unsigned long long *total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int m = 0;
    if (i < n) {
        const int32_t me = (int32_t)(row0 + i);
        int64_t lo = rowptr[i], hi = rowptr[i + 1];
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (col[mid] < me) lo = mid + 1; else hi = mid;
        }
        m = !(lo < rowptr[i + 1] && col[lo] == me);
        miss[i] = m;
    } else if (i == n) {
        miss[i] = 0;
    }
    if (total) {
        const unsigned long long b = __ballot(m != 0);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(total, (unsigned long long)__popcll(b));
    }
}

// column sums of a CSR block (fp64 atomics: exact, hence order-independent, for integer weights; otherwise equal up to
// the last bit of the fp64 sums)
__global__ __launch_bounds__(256) void block_colsum_kernel(const int32_t *__restrict__ col, const double *__restrict__ val,
                                                           int64_t m, double *__restrict__ colsum) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < m) atomicAdd(colsum + col[p], val[p]);
}

// A_hat[j, i] = (T'[j, i] * L[j]) * R[i]  (+ PPR mix), rounded to fp32 where the reference rounds.
// A block of 256 threads owns kScaleRows consecutive rows: their row pointers go to LDS (32-bit, relative to the block's first
// element), then the threads stream the block's non-zeros in order (coalesced reads of col / val, coalesced writes) and find each
// element's row by a binary search in LDS -- one thread per ROW walks its row alone and touches memory 16 bytes at a time
// (measured 4x slower).  What bounds the pass is the random 8-byte gather R[col]: four elements per thread are in flight at once
// (four independent col loads, then four independent gathers), and the streams bypass the caches (non-temporal) so that the R table
// keeps L2 / the Infinity Cache to itself.
// kScaled: `val` already holds A_hat in fp64 (the cached Laplacian of an alpha sweep): no L / R factors, only the PPR mix.
// Algorithmic bytes per non-zero: 4 (col) + 8 (T' or A_hat64) + 8 (R gather; absent when kScaled) + 4 (fp32 out) [+ 8 fp64 out].
constexpr int kScaleRows = 512;
constexpr int kScaleUnroll = 4;

template <bool kScaled>
__global__ __launch_bounds__(256) void block_scale_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          const double *__restrict__ val, const double *__restrict__ left_local,
                                                          const double *__restrict__ right_global, int64_t n, int64_t row0,
                                                          int use_alpha, double one_minus_alpha, double alpha,
                                                          float *__restrict__ o_val, double *__restrict__ o_val64) {
    __shared__ uint32_t rp[kScaleRows + 1];
    const int64_t r_begin = (int64_t)blockIdx.x * kScaleRows;
    const int rows = (int)min((int64_t)kScaleRows, n - r_begin);
    const int64_t p0 = rowptr[r_begin];
    for (int t = threadIdx.x; t <= rows; t += 256) rp[t] = (uint32_t)(rowptr[r_begin + t] - p0);
    __syncthreads();
    const uint32_t cnt = rp[rows];
    col += p0;
    val += p0;
    o_val += p0;
    if (o_val64) o_val64 += p0;
    for (uint32_t q0 = threadIdx.x; q0 < cnt; q0 += 256 * kScaleUnroll) {
        int32_t c[kScaleUnroll];
        double v[kScaleUnroll], rg[kScaleUnroll], lf[kScaleUnroll];
        int row[kScaleUnroll];
#pragma unroll
        for (int u = 0; u < kScaleUnroll; ++u) {
            const uint32_t q = q0 + 256u * u;
            c[u] = q < cnt ? __builtin_nontemporal_load(col + q) : 0;
            v[u] = q < cnt ? __builtin_nontemporal_load(val + q) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kScaleUnroll; ++u) {
            const uint32_t q = q0 + 256u * u;
            int lo = 0, hi = rows;                      // last row whose first element is <= q
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (rp[mid] <= q) lo = mid; else hi = mid;
            }
            row[u] = lo;
            if (!kScaled) {
                rg[u] = q < cnt ? right_global[c[u]] : 0.0;
                lf[u] = left_local[r_begin + lo];
            }
        }
#pragma unroll
        for (int u = 0; u < kScaleUnroll; ++u) {
            const uint32_t q = q0 + 256u * u;
            if (q >= cnt) continue;
            double x = kScaled ? v[u] : __dmul_rn(__dmul_rn(v[u], lf[u]), rg[u]);
            if (use_alpha) {
                x = __dmul_rn(one_minus_alpha, x);
                if (c[u] == (int32_t)(row0 + r_begin + row[u])) x = __dadd_rn(x, alpha);
            }
            __builtin_nontemporal_store((float)x, o_val + q);
            if (o_val64) __builtin_nontemporal_store(x, o_val64 + q);
        }
    }
}

// Position of the diagonal entry of every row of T' = T + I (it always exists): one binary search per row, once per block.  With it the
// PPR mix of an alpha sweep needs no row lookup at all: a flat stream (1 - alpha) * A_hat over all non-zeros, then alpha added at the
// n diagonal positions -- the same two roundings per element as block_scale_kernel<true>, bit-identical.
__global__ __launch_bounds__(256) void block_diagpos_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                            const int64_t n, const int64_t row0, int64_t *__restrict__ diag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t me = (int32_t)(row0 + i);
    int64_t lo = rowptr[i], hi = rowptr[i + 1];
    const int64_t end = hi;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (col[mid] < me) lo = mid + 1; else hi = mid;
    }
    diag[i] = (lo < end && col[lo] == me) ? lo : -1;
}

__global__ __launch_bounds__(256) void mix_flat_kernel(const double *__restrict__ hat, const int64_t m, const double one_minus_alpha,
                                                       float *__restrict__ o_val, double *__restrict__ o_val64) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    for (int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; p < m; p += stride) {
        if (p + 1 < m) {                                  // two doubles = one 16-byte load per lane (hat is 16-byte aligned)
            using d2 = double __attribute__((ext_vector_type(2)));
            const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(hat + p));
            const double a = __dmul_rn(one_minus_alpha, v[0]), b = __dmul_rn(one_minus_alpha, v[1]);
            using f2 = float __attribute__((ext_vector_type(2)));
            __builtin_nontemporal_store((f2){(float)a, (float)b}, reinterpret_cast<f2 *>(o_val + p));
            if (o_val64) __builtin_nontemporal_store((d2){a, b}, reinterpret_cast<d2 *>(o_val64 + p));
        } else {
            const double a = __dmul_rn(one_minus_alpha, hat[p]);
            o_val[p] = (float)a;
            if (o_val64) o_val64[p] = a;
        }
    }
}

__global__ __launch_bounds__(256) void mix_diag_kernel(const double *__restrict__ hat, const int64_t *__restrict__ diag, const int64_t n,
                                                       const double one_minus_alpha, const double alpha, float *__restrict__ o_val,
                                                       double *__restrict__ o_val64) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t p = diag[i];
    if (p < 0) return;
    const double x = __dadd_rn(__dmul_rn(one_minus_alpha, hat[p]), alpha);
    o_val[p] = (float)x;
    if (o_val64) o_val64[p] = x;
}

struct Tmp {
    std::vector<void *> ptrs;
    ~Tmp() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    int alloc(T **out, size_t count) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) return sgl::fail((int)e, "sgl_norm: hipMalloc failed: %s", hipGetErrorString(e));
        ptrs.push_back(p);
        *out = reinterpret_cast<T *>(p);
        return SGL_OK;
    }
};

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }
inline bool aligned_to16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

SGL_EXPORT int sgl_norm_prepare(int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col, int64_t *nnz_out,
                                void *stream) {
    SGL_REQUIRE(nnz_out != nullptr, "sgl_norm_prepare: NULL nnz_out");
    SGL_REQUIRE(n >= 0 && nnz >= 0 && n < INT32_MAX, "sgl_norm_prepare: bad sizes");
    if (n == 0) {
        *nnz_out = 0;
        return SGL_OK;
    }
    SGL_REQUIRE(d_rowptr && (nnz == 0 || d_col), "sgl_norm_prepare: NULL arrays");
    hipStream_t st = sgl::as_stream(stream);
    Tmp tmp;
    int64_t *miss = nullptr;
    unsigned long long *total = nullptr;
    int rc;
    if ((rc = tmp.alloc(&miss, (size_t)n + 1)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&total, 1)) != SGL_OK) return rc;
    SGL_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(diag_missing_kernel, dim3(blocks_for(n + 1)), dim3(256), 0, st, d_rowptr, d_col, n, miss, total);
    SGL_HIP_CHECK(hipGetLastError());
    unsigned long long h_total = 0;
    SGL_HIP_CHECK(hipMemcpyAsync(&h_total, total, sizeof(h_total), hipMemcpyDeviceToHost, st));
    SGL_HIP_CHECK(hipStreamSynchronize(st));
    *nnz_out = nnz + (int64_t)h_total;
    return SGL_OK;
}

static int norm_execute_impl(int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col, const float *d_val,
                             double r, int use_alpha, double alpha, int64_t nnz_out, int64_t *d_out_rowptr,
                             int32_t *d_out_col, float *d_out_val, double *d_out_val64, const double *d_left_in,
                             const double *d_right_in, void *stream) {
    SGL_REQUIRE(n >= 0 && nnz >= 0 && n < INT32_MAX, "sgl_norm_execute: bad sizes");
    SGL_REQUIRE(nnz_out >= nnz && nnz_out <= nnz + n, "sgl_norm_execute: nnz_out inconsistent (call sgl_norm_prepare)");
    SGL_REQUIRE(nnz_out < (int64_t)UINT32_MAX, "sgl_norm_execute: nnz_out >= 2^32 not supported on one device");
    SGL_REQUIRE(d_out_rowptr != nullptr, "sgl_norm_execute: NULL output row pointers");
    hipStream_t st = sgl::as_stream(stream);
    if (n == 0) {
        SGL_HIP_CHECK(hipMemsetAsync(d_out_rowptr, 0, sizeof(int64_t), st));
        return SGL_OK;
    }
    SGL_REQUIRE(d_rowptr && (nnz == 0 || (d_col && d_val)) && d_out_col && d_out_val, "sgl_norm_execute: NULL arrays");
    const int64_t m = nnz_out;
    Tmp tmp;
    int rc;
    int64_t *miss = nullptr, *shift = nullptr;
    uint32_t *t_key = nullptr, *keys_sorted = nullptr, *iota = nullptr, *perm = nullptr;
    int32_t *t_row = nullptr;
    double *a64 = nullptr, *deg = nullptr, *left = nullptr, *right = nullptr;
    if ((rc = tmp.alloc(&miss, (size_t)n + 1)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&shift, (size_t)n + 1)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&t_key, (size_t)m)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&keys_sorted, (size_t)m)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&iota, (size_t)m)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&perm, (size_t)m)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&t_row, (size_t)m)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&a64, (size_t)m)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&deg, (size_t)n)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&left, (size_t)n)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&right, (size_t)n)) != SGL_OK) return rc;

    // 1. which rows lack a diagonal entry; exclusive scan -> how far each row of A' is shifted
    hipLaunchKernelGGL(diag_missing_kernel, dim3(blocks_for(n + 1)), dim3(256), 0, st, d_rowptr, d_col, n, miss,
                       (unsigned long long *)nullptr);
    SGL_HIP_CHECK(hipGetLastError());
    {
        size_t bytes = 0;
        SGL_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, miss, shift, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st));
        char *scratch = nullptr;
        if ((rc = tmp.alloc(&scratch, bytes)) != SGL_OK) return rc;
        SGL_HIP_CHECK(rocprim::exclusive_scan(scratch, bytes, miss, shift, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st));
    }
    // 2. A' = A + I as (key = column, row, fp64 value) triplets in row-major order; weighted degrees
    hipLaunchKernelGGL(build_aprime_kernel, dim3(blocks_for(n)), dim3(256), 0, st, d_rowptr, d_col, d_val, shift, n, t_key,
                       t_row, a64, deg);
    SGL_HIP_CHECK(hipGetLastError());
    // 3. deg^(r-1), deg^(-r): on device, or supplied by the caller (host libm: bit-identical to the reference's numpy)
    if (d_left_in && d_right_in) {
        left = const_cast<double *>(d_left_in);
        right = const_cast<double *>(d_right_in);
    } else {
        hipLaunchKernelGGL(degree_scale_kernel, dim3(blocks_for(n)), dim3(256), 0, st, deg, n, r, left, right);
        SGL_HIP_CHECK(hipGetLastError());
    }
    // 4. values of the transposed, scaled matrix (still in A' order) + identity permutation
    hipLaunchKernelGGL(scale_values_kernel, dim3(blocks_for(m)), dim3(256), 0, st, t_key, t_row, left, right, m, a64, iota);
    SGL_HIP_CHECK(hipGetLastError());
    // 5. transpose = stable sort by output row
    {
        int end_bit = 1;
        while (end_bit < 32 && ((uint64_t)1 << end_bit) < (uint64_t)n) ++end_bit;
        size_t bytes = 0;
        SGL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, t_key, keys_sorted, iota, perm, (size_t)m, 0, end_bit, st));
        char *scratch = nullptr;
        if ((rc = tmp.alloc(&scratch, bytes)) != SGL_OK) return rc;
        SGL_HIP_CHECK(rocprim::radix_sort_pairs(scratch, bytes, t_key, keys_sorted, iota, perm, (size_t)m, 0, end_bit, st));
    }
    // 6. output row pointers, 7. gather + PPR + fp32 rounding
    hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3(blocks_for(n + 1)), dim3(256), 0, st, keys_sorted, m, n, d_out_rowptr);
    SGL_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks_for(m)), dim3(256), 0, st, keys_sorted, perm, t_row, a64, m, use_alpha,
                       1.0 - alpha, alpha, d_out_col, d_out_val, d_out_val64);
    SGL_HIP_CHECK(hipGetLastError());
    SGL_HIP_CHECK(hipStreamSynchronize(st));  // temporaries are freed on return
    return SGL_OK;
}

SGL_EXPORT int sgl_norm_execute(int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col, const float *d_val,
                                double r, int use_alpha, double alpha, int64_t nnz_out, int64_t *d_out_rowptr,
                                int32_t *d_out_col, float *d_out_val, double *d_out_val64, void *stream) {
    return norm_execute_impl(n, nnz, d_rowptr, d_col, d_val, r, use_alpha, alpha, nnz_out, d_out_rowptr, d_out_col, d_out_val,
                             d_out_val64, nullptr, nullptr, stream);
}

SGL_EXPORT int sgl_norm_execute_lr(int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col, const float *d_val,
                                   const double *d_left, const double *d_right, int use_alpha, double alpha, int64_t nnz_out,
                                   int64_t *d_out_rowptr, int32_t *d_out_col, float *d_out_val, double *d_out_val64,
                                   void *stream) {
    SGL_REQUIRE(d_left && d_right, "sgl_norm_execute_lr: NULL degree powers");
    return norm_execute_impl(n, nnz, d_rowptr, d_col, d_val, 0.0, use_alpha, alpha, nnz_out, d_out_rowptr, d_out_col, d_out_val,
                             d_out_val64, d_left, d_right, stream);
}

SGL_EXPORT int sgl_norm_degrees(int64_t n, int64_t row0, const int64_t *d_rowptr, const int32_t *d_col, const float *d_val,
                                double *d_deg, void *stream) {
    SGL_REQUIRE(n >= 0 && row0 >= 0 && row0 + n < INT32_MAX, "sgl_norm_degrees: bad sizes");
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_rowptr && d_deg, "sgl_norm_degrees: NULL arrays");
    hipLaunchKernelGGL(degrees_kernel, dim3(blocks_for(n)), dim3(256), 0, sgl::as_stream(stream), d_rowptr, d_col, d_val, n, row0,
                       d_deg);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}

// ---- row-block normalisation ---------------------------------------------------------------------------------------------
SGL_EXPORT int sgl_norm_block_prepare(int64_t n, int64_t row0, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col,
                                      int64_t *nnz_out, void *stream) {
    SGL_REQUIRE(nnz_out != nullptr, "sgl_norm_block_prepare: NULL nnz_out");
    SGL_REQUIRE(n >= 0 && nnz >= 0 && row0 >= 0 && row0 + n < INT32_MAX, "sgl_norm_block_prepare: bad sizes");
    if (n == 0) {
        *nnz_out = 0;
        return SGL_OK;
    }
    SGL_REQUIRE(d_rowptr && (nnz == 0 || d_col), "sgl_norm_block_prepare: NULL arrays");
    hipStream_t st = sgl::as_stream(stream);
    Tmp tmp;
    int64_t *miss = nullptr;
    unsigned long long *total = nullptr;
    int rc;
    if ((rc = tmp.alloc(&miss, (size_t)n + 1)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&total, 1)) != SGL_OK) return rc;
    SGL_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(block_diag_missing_kernel, dim3(blocks_for(n + 1)), dim3(256), 0, st, d_rowptr, d_col, n, row0, miss, total);
    SGL_HIP_CHECK(hipGetLastError());
    unsigned long long h_total = 0;
    SGL_HIP_CHECK(hipMemcpyAsync(&h_total, total, sizeof(h_total), hipMemcpyDeviceToHost, st));
    SGL_HIP_CHECK(hipStreamSynchronize(st));
    *nnz_out = nnz + (int64_t)h_total;
    return SGL_OK;
}

static int norm_block_build_impl(int64_t n, int64_t row0, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col,
                                 const float *d_val, int64_t nnz_out, int64_t *d_out_rowptr, int32_t *d_out_col,
                                 double *d_out_val64, double *d_rowsum, unsigned long long *d_sym_hash, void *stream) {
    SGL_REQUIRE(n >= 0 && nnz >= 0 && row0 >= 0 && row0 + n < INT32_MAX, "sgl_norm_block_build: bad sizes");
    SGL_REQUIRE(nnz_out >= nnz && nnz_out <= nnz + n, "sgl_norm_block_build: nnz_out inconsistent (call sgl_norm_block_prepare)");
    // (the build / scale kernels index a group of rows with 32-bit offsets and step them by 1 024: stay clear of the wrap)
    SGL_REQUIRE(nnz_out < (int64_t)UINT32_MAX - 4096, "sgl_norm_block_build: nnz >= 2^32 per block not supported (use more row blocks)");
    SGL_REQUIRE(d_out_rowptr != nullptr, "sgl_norm_block_build: NULL output row pointers");
    hipStream_t st = sgl::as_stream(stream);
    if (n == 0) {
        SGL_HIP_CHECK(hipMemsetAsync(d_out_rowptr, 0, sizeof(int64_t), st));
        return SGL_OK;
    }
    SGL_REQUIRE(d_rowptr && (nnz == 0 || (d_col && d_val)) && d_out_col && d_out_val64 && d_rowsum, "sgl_norm_block_build: NULL arrays");
    Tmp tmp;
    int rc;
    int64_t *miss = nullptr, *shift = nullptr;
    if ((rc = tmp.alloc(&miss, (size_t)n + 1)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&shift, (size_t)n + 1)) != SGL_OK) return rc;
    hipLaunchKernelGGL(block_diag_missing_kernel, dim3(blocks_for(n + 1)), dim3(256), 0, st, d_rowptr, d_col, n, row0, miss,
                       (unsigned long long *)nullptr);
    SGL_HIP_CHECK(hipGetLastError());
    size_t bytes = 0;
    SGL_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, miss, shift, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st));
    char *scratch = nullptr;
    if ((rc = tmp.alloc(&scratch, bytes)) != SGL_OK) return rc;
    SGL_HIP_CHECK(rocprim::exclusive_scan(scratch, bytes, miss, shift, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st));
    hipLaunchKernelGGL(block_build_kernel, dim3((unsigned)((n + kBuildRows - 1) / kBuildRows)), dim3(256), 0, st, d_rowptr, d_col,
                       d_val, shift, n, row0, d_out_rowptr, d_out_col, d_out_val64, d_rowsum, d_sym_hash);
    SGL_HIP_CHECK(hipGetLastError());
    SGL_HIP_CHECK(hipStreamSynchronize(st));  // temporaries are freed on return
    return SGL_OK;
}

SGL_EXPORT int sgl_norm_block_build(int64_t n, int64_t row0, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col,
                                    const float *d_val, int64_t nnz_out, int64_t *d_out_rowptr, int32_t *d_out_col,
                                    double *d_out_val64, double *d_rowsum, void *stream) {
    return norm_block_build_impl(n, row0, nnz, d_rowptr, d_col, d_val, nnz_out, d_out_rowptr, d_out_col, d_out_val64, d_rowsum,
                                 nullptr, stream);
}

// The same for a WHOLE matrix (row0 = 0, n rows and columns), additionally answering "is A symmetric, values included?":
// *d_sym_hash (one zero-initialised 64-bit word on the device) ends up 0 iff it is (up to a 2^-64 collision).  A symmetric A
// needs no transposition: A_hat[j,i] = (A'[j,i] * L[j]) * R[i] row by row (sgl_norm_block_scale) -- no sort, no permutation.
SGL_EXPORT int sgl_norm_build_symcheck(int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col, const float *d_val,
                                       int64_t nnz_out, int64_t *d_out_rowptr, int32_t *d_out_col, double *d_out_val64,
                                       double *d_rowsum, uint64_t *d_sym_hash, void *stream) {
    SGL_REQUIRE(d_sym_hash != nullptr, "sgl_norm_build_symcheck: NULL fingerprint word");
    return norm_block_build_impl(n, 0, nnz, d_rowptr, d_col, d_val, nnz_out, d_out_rowptr, d_out_col, d_out_val64, d_rowsum,
                                 reinterpret_cast<unsigned long long *>(d_sym_hash), stream);
}

SGL_EXPORT int sgl_norm_block_colsum(int64_t n_cols, int64_t nnz, const int32_t *d_col, const double *d_val64, double *d_colsum,
                                     void *stream) {
    SGL_REQUIRE(n_cols >= 0 && nnz >= 0 && nnz < (int64_t)UINT32_MAX, "sgl_norm_block_colsum: bad sizes (nnz must be < 2^32 per block)");
    if (nnz == 0) return SGL_OK;
    SGL_REQUIRE(d_col && d_val64 && d_colsum, "sgl_norm_block_colsum: NULL arrays");
    hipLaunchKernelGGL(block_colsum_kernel, dim3(blocks_for(nnz)), dim3(256), 0, sgl::as_stream(stream), d_col, d_val64, nnz, d_colsum);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}

SGL_EXPORT int sgl_norm_block_scale(int64_t n, int64_t row0, const int64_t *d_rowptr, const int32_t *d_col, const double *d_val64,
                                    const double *d_left_local, const double *d_right_global, int use_alpha, double alpha,
                                    float *d_out_val, double *d_out_val64, void *stream) {
    SGL_REQUIRE(n >= 0 && row0 >= 0 && row0 + n < INT32_MAX, "sgl_norm_block_scale: bad sizes");
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_rowptr && d_col && d_val64 && d_left_local && d_right_global && d_out_val, "sgl_norm_block_scale: NULL arrays");
    hipLaunchKernelGGL(block_scale_kernel<false>, dim3((unsigned)((n + kScaleRows - 1) / kScaleRows)), dim3(256), 0, sgl::as_stream(stream), d_rowptr, d_col, d_val64,
                       d_left_local, d_right_global, n, row0, use_alpha, 1.0 - alpha, alpha, d_out_val, d_out_val64);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}

// (1 - alpha) A_hat + alpha I from the fp64 A_hat of the same (graph, r) -- what PprGraphOp._construct_adj adds to the Laplacian
// (ppr_graph_op.py:20).  An alpha sweep at a fixed r pays the R gather once and this pure stream per alpha; the arithmetic is the
// tail of sgl_norm_block_scale, so the result is bit-identical to the one-pass form.
SGL_EXPORT int sgl_norm_block_mix(int64_t n, int64_t row0, const int64_t *d_rowptr, const int32_t *d_col, const double *d_hat64,
                                  double alpha, float *d_out_val, double *d_out_val64, void *stream) {
    SGL_REQUIRE(n >= 0 && row0 >= 0 && row0 + n < INT32_MAX, "sgl_norm_block_mix: bad sizes");
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_rowptr && d_col && d_hat64 && d_out_val, "sgl_norm_block_mix: NULL arrays");
    hipLaunchKernelGGL(block_scale_kernel<true>, dim3((unsigned)((n + kScaleRows - 1) / kScaleRows)), dim3(256), 0, sgl::as_stream(stream), d_rowptr, d_col, d_hat64,
                       (const double *)nullptr, (const double *)nullptr, n, row0, 1, 1.0 - alpha, alpha, d_out_val, d_out_val64);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}

SGL_EXPORT int sgl_norm_block_diag_positions(int64_t n, int64_t row0, const int64_t *d_rowptr, const int32_t *d_col, int64_t *d_diag,
                                             void *stream) {
    SGL_REQUIRE(n >= 0 && row0 >= 0 && row0 + n < INT32_MAX, "sgl_norm_block_diag_positions: bad sizes");
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_rowptr && d_col && d_diag, "sgl_norm_block_diag_positions: NULL arrays");
    hipLaunchKernelGGL(block_diagpos_kernel, dim3(blocks_for(n)), dim3(256), 0, sgl::as_stream(stream), d_rowptr, d_col, n, row0, d_diag);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}

// sgl_norm_block_mix with the diagonal positions known (sgl_norm_block_diag_positions, once per block): a flat stream + n fix-ups
SGL_EXPORT int sgl_norm_block_mix_at(int64_t nnz, int64_t n, const double *d_hat64, const int64_t *d_diag, double alpha, float *d_out_val,
                                     double *d_out_val64, void *stream) {
    SGL_REQUIRE(nnz >= 0 && n >= 0, "sgl_norm_block_mix_at: bad sizes");
    if (nnz == 0) return SGL_OK;
    SGL_REQUIRE(d_hat64 && d_diag && d_out_val, "sgl_norm_block_mix_at: NULL arrays");
    SGL_REQUIRE(aligned_to16(d_hat64) && (reinterpret_cast<uintptr_t>(d_out_val) & 7u) == 0 && (!d_out_val64 || aligned_to16(d_out_val64)),
                "sgl_norm_block_mix_at: arrays must be 16-byte (fp64) / 8-byte (fp32) aligned");
    hipStream_t st = sgl::as_stream(stream);
    const int64_t pairs = (nnz + 1) / 2;
    const unsigned grid = (unsigned)std::min<int64_t>((pairs + 255) / 256, (int64_t)1 << 22);
    hipLaunchKernelGGL(mix_flat_kernel, dim3(grid), dim3(256), 0, st, d_hat64, nnz, 1.0 - alpha, d_out_val, d_out_val64);
    SGL_HIP_CHECK(hipGetLastError());
    if (n > 0) {
        hipLaunchKernelGGL(mix_diag_kernel, dim3(blocks_for(n)), dim3(256), 0, st, d_hat64, d_diag, n, 1.0 - alpha, alpha, d_out_val,
                           d_out_val64);
        SGL_HIP_CHECK(hipGetLastError());
    }
    return SGL_OK;
}

// deg^(r-1), deg^(-r) with inf -> 0 (utils.py:79-84) by the DEVICE's pow(): within 1 ulp(fp64) of the host libm the reference calls,
// i.e. A_hat within 1 ulp(fp32) in a handful of entries -- the route of every caller that does not ask for bit-identity.
SGL_EXPORT int sgl_norm_degree_powers(int64_t n, const double *d_deg, double r, double *d_left, double *d_right, void *stream) {
    SGL_REQUIRE(n >= 0, "sgl_norm_degree_powers: bad size");
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_deg && d_left && d_right, "sgl_norm_degree_powers: NULL arrays");
    hipLaunchKernelGGL(degree_scale_kernel, dim3(blocks_for(n)), dim3(256), 0, sgl::as_stream(stream), d_deg, n, r, d_left, d_right);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}
