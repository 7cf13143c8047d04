// Experiment (not part of the product): which indexing of a purely sequential read / copy does the MI355X memory system like
// best?  The random row-gather probe reaches 6.1 TB/s of pure HBM traffic while the grid-stride stream probe reaches 5.6 --
// a sequential pattern should not lose to a random one, so the indexing must be aliasing channels.
//   hipcc --offload-arch=gfx950 -O3 tools/native/stream_patterns.hip -o tools/native/stream_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
using F4 = float __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void keep(F4 acc, float *sink) {
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 1.2345678e-30f) sink[0] = s;
}

// A: grid-stride, U loads in flight, `stride` = whole grid apart
template <int U>
__global__ __launch_bounds__(256) void rd_gridstride(const F4 *__restrict__ x, int64_t n, float *sink) {
    F4 acc = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        F4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n; i += stride) acc += x[i];
    keep(acc, sink);
}

// B: one tile of U * 4 KB contiguous per block, as many blocks as tiles (no loop)
template <int U>
__global__ __launch_bounds__(256) void rd_tile(const F4 *__restrict__ x, int64_t n, float *sink) {
    F4 acc = {0, 0, 0, 0};
    const int64_t base = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
    F4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (base + u * 256 < n) ? x[base + u * 256] : acc;
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
    keep(acc, sink);
}

// C: persistent blocks, each walks tiles of U * 4 KB, consecutive tiles spread over blocks (tile t -> block t % grid)
template <int U>
__global__ __launch_bounds__(256) void rd_tile_loop(const F4 *__restrict__ x, int64_t n, float *sink) {
    F4 acc = {0, 0, 0, 0};
    const int64_t tiles = (n + 256 * U - 1) / (256 * U);
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * (256 * U) + threadIdx.x;
        F4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (base + u * 256 < n) ? x[base + u * 256] : (F4){0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    keep(acc, sink);
}

// D: persistent blocks, each owns one contiguous chunk of the array
template <int U>
__global__ __launch_bounds__(256) void rd_chunk(const F4 *__restrict__ x, int64_t n, float *sink) {
    F4 acc = {0, 0, 0, 0};
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (int64_t b = lo + threadIdx.x; b < hi; b += 256 * U) {
        F4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (b + u * 256 < hi) ? x[b + u * 256] : (F4){0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    keep(acc, sink);
}

// copy variants (read + write): grid-stride vs tiles
template <int U>
__global__ __launch_bounds__(256) void cp_gridstride(const F4 *__restrict__ x, F4 *__restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        F4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) y[i] = x[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void cp_tile(const F4 *__restrict__ x, F4 *__restrict__ y, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
    F4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(x + base + u * 256) : x[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], y + base + u * 256); else y[base + u * 256] = v[u]; }
}

// H input streams summed into one output (the shape of the hop aggregators): dynamic hop loop (loads serialised behind
// the adds) vs all H loads issued first
struct Ptrs { const F4 *p[16]; };
__global__ __launch_bounds__(256) void sum_dyn(Ptrs in, int H, F4 *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    F4 acc = in.p[0][i];
    for (int h = 1; h < H; ++h) acc += in.p[h][i];
    y[i] = acc;
}
template <int HM, bool NT>
__global__ __launch_bounds__(256) void sum_unrolled(Ptrs in, int H, F4 *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    F4 v[HM];
#pragma unroll
    for (int h = 0; h < HM; ++h) if (h < H) v[h] = NT ? __builtin_nontemporal_load(in.p[h] + i) : in.p[h][i];
    F4 acc = v[0];
#pragma unroll
    for (int h = 1; h < HM; ++h) if (h < H) acc += v[h];
    if (NT) __builtin_nontemporal_store(acc, y + i); else y[i] = acc;
}
// two elements per thread, 128 B apart? no: two consecutive 4 KB tiles per block
template <int HM>
__global__ __launch_bounds__(256) void sum_unrolled2(Ptrs in, int H, F4 *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x;
    if (i + 256 >= n) return;
    F4 a[HM], b[HM];
#pragma unroll
    for (int h = 0; h < HM; ++h) if (h < H) { a[h] = in.p[h][i]; b[h] = in.p[h][i + 256]; }
    F4 s = a[0], t = b[0];
#pragma unroll
    for (int h = 1; h < HM; ++h) if (h < H) { s += a[h]; t += b[h]; }
    y[i] = s; y[i + 256] = t;
}

template <typename F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 8.0;
    const int64_t n = (int64_t)(gb * 1e9 / 16);
    F4 *x, *y; float *sink;
    CK(hipMalloc(&x, n * 16)); CK(hipMalloc(&y, n * 16)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 1, n * 16)); CK(hipMemset(y, 0, n * 16));
    const double bytes = (double)n * 16;
#define RD(name, expr) { double ms = time_ms([&] { expr; }); printf("STREAM read  %-34s ms=%7.3f TBps=%.2f\n", name, ms, bytes / ms / 1e9); }
#define CP(name, expr) { double ms = time_ms([&] { expr; }); printf("STREAM copy  %-34s ms=%7.3f TBps=%.2f (read+write)\n", name, ms, 2 * bytes / ms / 1e9); }
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "gridstride U=4 grid=%d", g); RD(nm, (rd_gridstride<4><<<g, 256>>>(x, n, sink)));
        snprintf(nm, 64, "gridstride U=8 grid=%d", g); RD(nm, (rd_gridstride<8><<<g, 256>>>(x, n, sink)));
    }
    RD("gridstride U=1 grid=2048", (rd_gridstride<1><<<2048, 256>>>(x, n, sink)));
    RD("gridstride U=2 grid=2048", (rd_gridstride<2><<<2048, 256>>>(x, n, sink)));
    RD("tile U=1 (4 KB per block)", (rd_tile<1><<<(unsigned)((n + 255) / 256), 256>>>(x, n, sink)));
    RD("tile U=2 (8 KB per block)", (rd_tile<2><<<(unsigned)((n + 511) / 512), 256>>>(x, n, sink)));
    RD("tile U=4 (16 KB per block)", (rd_tile<4><<<(unsigned)((n + 1023) / 1024), 256>>>(x, n, sink)));
    RD("tile U=8 (32 KB per block)", (rd_tile<8><<<(unsigned)((n + 2047) / 2048), 256>>>(x, n, sink)));
    for (int g : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "tile_loop U=4 grid=%d", g); RD(nm, (rd_tile_loop<4><<<g, 256>>>(x, n, sink)));
        snprintf(nm, 64, "tile_loop U=8 grid=%d", g); RD(nm, (rd_tile_loop<8><<<g, 256>>>(x, n, sink)));
        snprintf(nm, 64, "chunk U=4 grid=%d", g); RD(nm, (rd_chunk<4><<<g, 256>>>(x, n, sink)));
        snprintf(nm, 64, "chunk U=8 grid=%d", g); RD(nm, (rd_chunk<8><<<g, 256>>>(x, n, sink)));
    }
    CP("gridstride U=4 grid=2048", (cp_gridstride<4><<<2048, 256>>>(x, y, n)));
    CP("gridstride U=4 grid=8192", (cp_gridstride<4><<<8192, 256>>>(x, y, n)));
    CP("tile U=1", (cp_tile<1, false><<<(unsigned)((n + 255) / 256), 256>>>(x, y, n)));
    CP("tile U=4", (cp_tile<4, false><<<(unsigned)((n + 1023) / 1024), 256>>>(x, y, n)));
    CP("tile U=8", (cp_tile<8, false><<<(unsigned)((n + 2047) / 2048), 256>>>(x, y, n)));
    CP("tile U=4 nontemporal", (cp_tile<4, true><<<(unsigned)((n + 1023) / 1024), 256>>>(x, y, n)));
    CP("hipMemcpyDtoD", CK(hipMemcpyAsync(y, x, n * 16, hipMemcpyDeviceToDevice, 0)));
    for (int H : {4, 6, 11}) {
        const int64_t m = n / (H + 1);          // H inputs + 1 output carved out of x / y
        Ptrs in;
        for (int h = 0; h < 16; ++h) in.p[h] = x + (int64_t)(h % H) * (n / H);
        const int64_t mm = n / H < m ? n / H : m;
        const double b2 = (double)mm * 16 * (H + 1);
        const unsigned g = (unsigned)((mm + 255) / 256);
        char nm[64];
#define SM(name, expr) { double ms = time_ms([&] { expr; }); printf("STREAM sum H=%-2d %-30s ms=%7.3f TBps=%.2f (reads+write)\n", H, name, ms, b2 / ms / 1e9); }
        SM("dynamic hop loop", (sum_dyn<<<g, 256>>>(in, H, y, mm)));
        if (H <= 4) SM("unrolled HM=4", (sum_unrolled<4, false><<<g, 256>>>(in, H, y, mm)));
        if (H <= 8) SM("unrolled HM=8", (sum_unrolled<8, false><<<g, 256>>>(in, H, y, mm)));
        if (H <= 12) SM("unrolled HM=12", (sum_unrolled<12, false><<<g, 256>>>(in, H, y, mm)));
        SM("unrolled HM=16", (sum_unrolled<16, false><<<g, 256>>>(in, H, y, mm)));
        SM("unrolled HM=16 nontemporal", (sum_unrolled<16, true><<<g, 256>>>(in, H, y, mm)));
        if (H <= 8) SM("unrolled HM=8, 2 tiles per block", (sum_unrolled2<8><<<g / 2, 256>>>(in, H, y, mm)));
        (void)nm;
    }
    return 0;
}
