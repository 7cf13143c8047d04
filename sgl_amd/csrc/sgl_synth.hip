// Seeded synthetic inputs generated directly in HBM, keyed by (seed, row): every rank of a multi-GPU job builds ITS row
// block of an ogbn-papers100M-shaped graph (workloads S3 / S4) and its feature rows without
// ever materialising the whole graph anywhere, and a CPU mirror (sgl_amd/synthetic.py, numpy) regenerates any sampled
// row bit-for-bit for the parity tests.  Integer arithmetic only (a counter-based hash), so device and host agree
// exactly.  Measurement / test infrastructure: there are no dataset files and no network on the GPU box.
//
//   h(seed, stream, a, b) = mix(mix(mix(seed * GOLD + stream) ^ a) + b),  mix = the splitmix64 finaliser
//   degree(row)    = table[t], t = h(seed, 0, row, 0) >> 52; the top bucket (t = 4095) is refined by 12 more hash bits into
//                    table[4096 + ...]: 2 x 4096 quantiles of the degree law (body + extreme tail), computed on the host
//   col(row, j)    = P( mulhi(mulhi(u, u), n_cols) ),  u = h(seed, 1, row, j)   u^2: density ~ id^(-1/2) = hub-skewed;
//                    P = keyed Feistel permutation of [0, n_cols) (cycle walking) that scatters the hubs over the id range
//   val(row, j)    = (h(seed, 2, row, j) >> 40) * 2^-29                     in [0, 1/32), exact in fp32
//   x(row, k)      = ((h(seed, 3, row, k) >> 40) - 2^23) * 2^-23            in [-1, 1),   exact in fp32
#include "sgl_common.h"

#include "../../include/sgl_probe.h"

namespace {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ __forceinline__ uint64_t hash4(uint64_t seed, uint64_t stream, uint64_t a, uint64_t b) {
    return mix64(mix64(mix64(seed * 0x9E3779B97F4A7C15ull + stream) ^ a) + b);
}

__device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b) { return __umul64hi(a, b); }

// keyed balanced Feistel network on 2*half bits, cycle-walked into [0, n)
__device__ __forceinline__ uint64_t permute_id(uint64_t x, uint64_t n, int half, uint64_t key) {
    const uint64_t mask = (1ull << half) - 1;
    do {
        uint64_t l = x >> half, r = x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint64_t f = mix64(r + key + (uint64_t)round * 0xD6E8FEB86659FD93ull) & mask;
            const uint64_t nl = r;
            r = l ^ f;
            l = nl;
        }
        x = (l << half) | r;
    } while (x >= n);
    return x;
}

__global__ __launch_bounds__(256) void synth_degrees_kernel(uint64_t seed, int64_t row0, int64_t n_rows,
                                                            const int32_t *__restrict__ table, int64_t *__restrict__ deg) {
    // grid-stride: a HIP launch is limited to 2^32 - 1 threads per dimension, the papers100M-sized inputs need more
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = hash4(seed, 0, (uint64_t)(row0 + i), 0);
        const uint64_t t = h >> 52;
        deg[i] = (t == 4095) ? table[4096 + ((h >> 40) & 4095)] : table[t];
    }
}

// one wavefront per row: lane l writes non-zeros l, l + 64, ... (coalesced)
__global__ __launch_bounds__(256) void synth_fill_kernel(uint64_t seed, int64_t row0, int64_t n_rows, uint64_t n_cols, int half,
                                                         const int64_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                         float *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const uint64_t pkey = mix64(seed ^ 0xA5A5A5A5A5A5A5A5ull);
    const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n_rows; r += waves) {   // grid-stride over rows
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        const uint64_t row = (uint64_t)(row0 + r);
        for (int64_t j = lane; j < e - b; j += 64) {
            const uint64_t u = hash4(seed, 1, row, (uint64_t)j);
            const uint64_t skew = mulhi64(mulhi64(u, u), n_cols);
            col[b + j] = (int32_t)permute_id(skew, n_cols, half, pkey);
            val[b + j] = (float)(hash4(seed, 2, row, (uint64_t)j) >> 40) * 1.862645149230957e-09f;   // 2^-29
        }
    }
}

__global__ __launch_bounds__(256) void synth_features_kernel(uint64_t seed, int64_t row0, int64_t n_rows, int d, int64_t ld,
                                                             float *__restrict__ x) {
    const int64_t total = n_rows * ld;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / ld;
        const int k = (int)(t - i * ld);
        float v = 0.f;   // pad columns are zero
        if (k < d) {
            const int64_t q = (int64_t)(hash4(seed, 3, (uint64_t)(row0 + i), (uint64_t)k) >> 40) - (1 << 23);
            v = (float)q * 1.1920928955078125e-07f;   // 2^-23
        }
        x[t] = v;
    }
}

// at most 2^22 blocks of 256 threads (2^30 threads, far below the 2^32 - 1 launch limit): the kernels stride over the rest
inline unsigned blocks_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)1 << 22); }

inline int feistel_half_bits(int64_t n) {
    int bits = 2;
    while (bits < 62 && ((int64_t)1 << bits) < n) ++bits;
    return (bits + 1) / 2;
}

}  // namespace

SGL_EXPORT int sgl_synth_degrees(uint64_t seed, int64_t row0, int64_t n_rows, const int32_t *d_table8192, int64_t *d_deg,
                                 void *stream) {
    SGL_REQUIRE(n_rows >= 0 && row0 >= 0 && d_table8192 && (n_rows == 0 || d_deg), "sgl_synth_degrees: bad arguments");
    if (n_rows == 0) return SGL_OK;
    hipLaunchKernelGGL(synth_degrees_kernel, dim3(blocks_for(n_rows)), dim3(256), 0, sgl::as_stream(stream), seed, row0, n_rows,
                       d_table8192, d_deg);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}

SGL_EXPORT int sgl_synth_fill(uint64_t seed, int64_t row0, int64_t n_rows, int64_t n_cols, const int64_t *d_rowptr,
                              int32_t *d_col, float *d_val, void *stream) {
    SGL_REQUIRE(n_rows >= 0 && row0 >= 0 && n_cols > 0 && n_cols < INT32_MAX, "sgl_synth_fill: bad sizes");
    if (n_rows == 0) return SGL_OK;
    SGL_REQUIRE(d_rowptr && d_col && d_val, "sgl_synth_fill: NULL arrays");
    const int64_t threads = n_rows * 64;
    hipLaunchKernelGGL(synth_fill_kernel, dim3(blocks_for(threads)), dim3(256), 0, sgl::as_stream(stream), seed, row0, n_rows,
                       (uint64_t)n_cols, feistel_half_bits(n_cols), d_rowptr, d_col, d_val);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}

SGL_EXPORT int sgl_synth_features(uint64_t seed, int64_t row0, int64_t n_rows, int64_t d, int64_t ld, float *d_x, void *stream) {
    SGL_REQUIRE(n_rows >= 0 && row0 >= 0 && d >= 0 && ld >= d && d < INT32_MAX, "sgl_synth_features: bad sizes");
    if (n_rows == 0 || ld == 0) return SGL_OK;
    SGL_REQUIRE(d_x != nullptr, "sgl_synth_features: NULL output");
    hipLaunchKernelGGL(synth_features_kernel, dim3(blocks_for(n_rows * ld)), dim3(256), 0, sgl::as_stream(stream), seed, row0, n_rows,
                       (int)d, ld, d_x);
    SGL_HIP_CHECK(hipGetLastError());
    return SGL_OK;
}
