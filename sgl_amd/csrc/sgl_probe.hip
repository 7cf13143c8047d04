// Memory-system probes (measurement infrastructure, not part of the propagation path): what the MI355X delivers to the
// two access patterns the SpMM is made of, with no CSR stream, no row bookkeeping and no stores in the way --
//   * sgl_probe_stream_f32: every lane reads one 16-byte word of one large array (sequential-read ceiling);
//   * sgl_probe_gather_f32: every wavefront reads whole rows table[idx[i], 0:row_floats] for a list of row ids with
//     U independent rows in flight per lane (random-row-gather ceiling for a given row width / table size).
// bench.py / tools/mem_ceilings.py report the SpMM kernel's rate next to these ceilings.
#include <algorithm>

#include "sgl_common.h"

#include "../../include/sgl_probe.h"

namespace {

using F4 = float __attribute__((ext_vector_type(4)));

// One 16-byte word per thread and as many blocks as that takes: on this memory system the plain mapping beats every
// persistent / grid-stride variant (tools/native/stream_patterns.hip, profiles/r02_stream_patterns.log: 6.8 TB/s against
// 5.5-6.5 for grid-stride loops with 1k-16k blocks); the loop only serves arrays beyond 2^31 blocks.
__global__ __launch_bounds__(256) void probe_stream_kernel(const F4 *__restrict__ x, int64_t n_vec, float *__restrict__ sink) {
    F4 acc = {0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) acc += x[i];
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 1.2345678e-30f) sink[0] = s;   // never true for real data: keeps the loads alive without a store stream
}

template <int U>
__global__ __launch_bounds__(256) void probe_gather_kernel(const float *__restrict__ table, int64_t ld, const int32_t *__restrict__ idx,
                                                           int64_t n_idx, int lanes, int per_wave, float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t begin = wave * per_wave;
    if (begin >= n_idx) return;
    const int cnt = (int)min<int64_t>(per_wave, n_idx - begin);
    const bool on = lane < lanes;
    F4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < cnt; base += 64) {
        const int have = min(64, cnt - base);
        const int my = (lane < have) ? idx[begin + base + lane] : 0;
        int t = 0;
        for (; t + U <= have; t += U) {
            F4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = __builtin_amdgcn_readlane(my, t + u);
                v[u] = on ? *reinterpret_cast<const F4 *>(table + (int64_t)r * ld + lane * 4) : acc;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
        for (; t < have; ++t) {
            const int r = __builtin_amdgcn_readlane(my, t);
            if (on) acc += *reinterpret_cast<const F4 *>(table + (int64_t)r * ld + lane * 4);
        }
    }
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 1.2345678e-30f) sink[0] = s;
}

}  // namespace

SGL_EXPORT int sgl_probe_stream_f32(const float *d_x, int64_t n_floats, float *d_sink, void *stream) {
    SGL_REQUIRE(d_x && d_sink && n_floats >= 4 && (reinterpret_cast<uintptr_t>(d_x) % 16) == 0, "sgl_probe_stream_f32: bad arguments");
    const int64_t n_vec = n_floats / 4;
    const unsigned blocks = (unsigned)std::min<int64_t>((n_vec + 255) / 256, (int64_t)1 << 30);
    hipLaunchKernelGGL(probe_stream_kernel, dim3(blocks), dim3(256), 0, sgl::as_stream(stream),
                       reinterpret_cast<const F4 *>(d_x), n_vec, d_sink);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sgl::fail((int)e, "sgl_probe_stream_f32: launch failed: %s", hipGetErrorString(e));
    return SGL_OK;
}

SGL_EXPORT int sgl_probe_gather_f32(const float *d_table, int64_t ld, const int32_t *d_idx, int64_t n_idx, int row_floats,
                                    int in_flight, float *d_sink, void *stream) {
    SGL_REQUIRE(d_table && d_idx && d_sink && n_idx > 0, "sgl_probe_gather_f32: bad arguments");
    SGL_REQUIRE(row_floats >= 4 && row_floats <= 256 && row_floats % 4 == 0 && ld % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(d_table) % 16) == 0,
                "sgl_probe_gather_f32: rows must be 16-byte aligned, 4..256 floats");
    const int per_wave = 512;
    const int64_t waves = (n_idx + per_wave - 1) / per_wave;
    const int64_t blocks = (waves + 3) / 4;
    SGL_REQUIRE(blocks < INT32_MAX, "sgl_probe_gather_f32: too many indices");
    hipStream_t st = sgl::as_stream(stream);
    const int lanes = row_floats / 4;
    if (in_flight >= 32)
        hipLaunchKernelGGL((probe_gather_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, st, d_table, ld, d_idx, n_idx, lanes, per_wave, d_sink);
    else if (in_flight >= 16)
        hipLaunchKernelGGL((probe_gather_kernel<16>), dim3((unsigned)blocks), dim3(256), 0, st, d_table, ld, d_idx, n_idx, lanes, per_wave, d_sink);
    else
        hipLaunchKernelGGL((probe_gather_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, st, d_table, ld, d_idx, n_idx, lanes, per_wave, d_sink);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sgl::fail((int)e, "sgl_probe_gather_f32: launch failed: %s", hipGetErrorString(e));
    return SGL_OK;
}
