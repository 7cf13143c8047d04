// EXPERIMENT (round 2; REJECTED, kept for the record -- not compiled into libsgl_hip.so; to re-run it add this file to
// SOURCES in sgl_amd/csrc/build.py and use tailpass_experiment.py): the 4-column tail of a d = 100 propagation as a separate pass over
// column blocks of A_hat sized so that the packed tail slice of a block (16 bytes per node) stays in one XCD's L2.
// The main pass then gathers 3 lines per non-zero instead of 4 (sgl_spmm_f32 on the first 96 columns); this pass gathers
// the 16-byte tails from L2 instead of pulling a fourth line through the fabric.  A row's entries of one column block are
// contiguous in the CSR (columns are sorted), so the pass needs only an offset table off[b][r] = number of entries of row r
// with column < b * cols_per_block.  tools/exp_tailpass.py measures it; DESIGN.md section 8 discusses it.
#include <algorithm>

#include "../../sgl_amd/csrc/sgl_common.h"

namespace {

using F4 = float __attribute__((ext_vector_type(4)));

// off[b * n_rows + r] for b = 0..n_blocks (off[0][r] = 0, off[n_blocks][r] = row length)
__global__ __launch_bounds__(256) void tp_offsets_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         const int64_t n_rows, const int n_blocks, const int cols_per_block,
                                                         int32_t *__restrict__ off) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * (n_blocks + 1)) return;
    const int b = (int)(i / n_rows);
    const int64_t r = i - (int64_t)b * n_rows;
    const int64_t p0 = rowptr[r], p1 = rowptr[r + 1];
    const int64_t target = (int64_t)b * cols_per_block;
    int64_t lo = p0, hi = p1;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (col[mid] < target) lo = mid + 1; else hi = mid;
    }
    off[i] = (int32_t)(lo - p0);
}

// one lane per row, the 4 tail columns of the row as one 16-byte vector per lane; U entries in flight per lane.
// Segments longer than `long_len` are left to tp_long_kernel (their rows are listed at plan time).
template <int U>
__global__ __launch_bounds__(256) void tp_rows_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      const float *__restrict__ val, const int32_t *__restrict__ off_b,
                                                      const int32_t *__restrict__ off_b1, const int64_t n_rows,
                                                      const F4 *__restrict__ xtail, F4 *__restrict__ ytail, const int first,
                                                      const int long_len, float *__restrict__ y_main, const int64_t ldy,
                                                      const int col0, const int last) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int o0 = off_b[r], o1 = off_b1[r];
    F4 acc = first ? (F4){0.f, 0.f, 0.f, 0.f} : ytail[r];
    if (o1 - o0 <= long_len) {
        const int64_t base = rowptr[r];
        for (int64_t p = base + o0; p < base + o1; p += U) {
            int32_t c[U];
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool on = p + u < base + o1;
                c[u] = on ? col[p + u] : 0;
                v[u] = on ? val[p + u] : 0.f;
            }
            F4 x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = xtail[c[u]];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (p + u < base + o1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(v[u], x[u][e], acc[e]);
                }
        }
    }
    if (!first || o1 - o0 <= long_len || true) ytail[r] = acc;
    if (last && y_main) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y_main[r * ldy + col0 + e] = acc[e];
    }
}

// one wavefront per long segment: lanes take strided entries, fixed butterfly, added onto ytail
__global__ __launch_bounds__(256) void tp_long_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      const float *__restrict__ val, const int32_t *__restrict__ off_b,
                                                      const int32_t *__restrict__ off_b1, const int32_t *__restrict__ long_rows,
                                                      const int n_long, const F4 *__restrict__ xtail, F4 *__restrict__ ytail,
                                                      const int long_len, float *__restrict__ y_main, const int64_t ldy,
                                                      const int col0, const int last) {
    const int w = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6);
    if (w >= n_long) return;
    const int lane = threadIdx.x & 63;
    const int64_t r = long_rows[w];
    const int o0 = off_b[r], o1 = off_b1[r];
    if (o1 - o0 <= long_len) return;                // short in this block: the row kernel did it
    const int64_t base = rowptr[r];
    F4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t p = base + o0 + lane; p < base + o1; p += 64) {
        const F4 x = xtail[col[p]];
        const float v = val[p];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(v, x[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = acc[e];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) a += __shfl_xor(a, s, 64);
        acc[e] = a;
    }
    if (lane == 0) {
        F4 y = ytail[r];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] += acc[e];
        ytail[r] = y;
        if (last && y_main) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y_main[r * ldy + col0 + e] = y[e];
        }
    }
}

}  // namespace

SGL_EXPORT int sgl_exp_tailpass_offsets(const int64_t *d_rowptr, const int32_t *d_col, int64_t n_rows, int n_blocks,
                                        int cols_per_block, int32_t *d_off, void *stream) {
    SGL_REQUIRE(d_rowptr && d_col && d_off && n_rows > 0 && n_blocks > 0 && cols_per_block > 0, "sgl_exp_tailpass_offsets: bad arguments");
    const int64_t total = n_rows * (n_blocks + 1);
    SGL_REQUIRE(sgl::launch_fits((total + 255) / 256, 256), "sgl_exp_tailpass_offsets: too large");
    hipLaunchKernelGGL(tp_offsets_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sgl::as_stream(stream), d_rowptr, d_col,
                       n_rows, n_blocks, cols_per_block, d_off);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}

SGL_EXPORT int sgl_exp_tailpass_run(const int64_t *d_rowptr, const int32_t *d_col, const float *d_val, const int32_t *d_off,
                                    int64_t n_rows, int n_blocks, const int32_t *d_long_rows, int n_long, int long_len,
                                    const float *d_xtail, float *d_ytail, float *d_y_main, int64_t ldy, int col0, int unroll,
                                    void *stream) {
    SGL_REQUIRE(d_rowptr && d_col && d_val && d_off && d_xtail && d_ytail && n_rows > 0 && n_blocks > 0, "sgl_exp_tailpass_run: bad arguments");
    hipStream_t st = sgl::as_stream(stream);
    const unsigned grid = (unsigned)((n_rows + 255) / 256);
    const F4 *xt = reinterpret_cast<const F4 *>(d_xtail);
    F4 *yt = reinterpret_cast<F4 *>(d_ytail);
    for (int b = 0; b < n_blocks; ++b) {
        const int32_t *o0 = d_off + (int64_t)b * n_rows, *o1 = d_off + (int64_t)(b + 1) * n_rows;
        const int first = b == 0, last = b == n_blocks - 1;
        const bool long_last = last && n_long > 0;      // the long kernel of the last block finishes the row
        if (unroll >= 8)
            hipLaunchKernelGGL((tp_rows_kernel<8>), dim3(grid), dim3(256), 0, st, d_rowptr, d_col, d_val, o0, o1, n_rows, xt, yt, first,
                               long_len, d_y_main, ldy, col0, last);
        else if (unroll >= 4)
            hipLaunchKernelGGL((tp_rows_kernel<4>), dim3(grid), dim3(256), 0, st, d_rowptr, d_col, d_val, o0, o1, n_rows, xt, yt, first,
                               long_len, d_y_main, ldy, col0, last);
        else
            hipLaunchKernelGGL((tp_rows_kernel<2>), dim3(grid), dim3(256), 0, st, d_rowptr, d_col, d_val, o0, o1, n_rows, xt, yt, first,
                               long_len, d_y_main, ldy, col0, last);
        if (n_long > 0)
            hipLaunchKernelGGL(tp_long_kernel, dim3((unsigned)((n_long + 3) / 4)), dim3(256), 0, st, d_rowptr, d_col, d_val, o0, o1,
                               d_long_rows, n_long, xt, yt, long_len, d_y_main, ldy, col0, long_last ? 1 : 0);
    }
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}
