// Plan-time locality ordering (sgl_reorder_community): semi-synchronous label propagation on the device, then a stable
// sort of the nodes by label.  Real co-purchase / citation graphs have communities; when their members are processed
// close together, a gathered row of X fetched for one member is still in L2 / the Infinity Cache for the next
// (DESIGN.md K1 "Locality ordering", tools/bench_reorder.py).  Not part of the per-hop path: runs once per adjacency.
//
// One round: every node looks at (a strided sample of at most 256 of) its neighbours' labels and adopts the most
// frequent one, ties to the smaller label.  One wavefront per node: the sampled labels sit in LDS, every lane counts
// the occurrences of its own labels with broadcast reads, a butterfly picks the best (count, -label).  Half of the nodes
// (a hash of node and round) may move per round, so two-coloured structures cannot oscillate; the last round moves all.
#include "sgl_common.h"

#include <algorithm>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

constexpr int kSample = 256;

__device__ __forceinline__ bool lpa_active(int64_t node, int round, int last) {
    if (last) return true;
    unsigned long long x = (unsigned long long)node ^ ((unsigned long long)round * 2654435761ull + 12345ull);
    x = (x * 0x9E3779B97F4A7C15ull) & 0x7FFFFFFFFFFFFFFFull;
    x = (x >> 29) ^ x;
    return (int)(x & 1ull) == (round & 1);
}

// rows [row0, row0 + n) of the matrix (rowptr local to the block, GLOBAL column ids); labels: one per node of the WHOLE graph;
// out: the new labels of the block's nodes (local index).  row0 = 0 and n = all nodes: the whole matrix.
__global__ __launch_bounds__(256) void lpa_round_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        const int64_t n, const int64_t row0, const int32_t *__restrict__ labels,
                                                        int32_t *__restrict__ out, const int round, const int last,
                                                        unsigned long long *__restrict__ moved) {
    __shared__ int32_t sl[4][kSample];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t node = (int64_t)blockIdx.x * 4 + w;
    if (node >= n) return;
    const int32_t mine = labels[row0 + node];
    const int64_t p0 = rowptr[node], deg = rowptr[node + 1] - p0;
    if (deg == 0 || !lpa_active(row0 + node, round, last)) {
        if (lane == 0) out[node] = mine;
        return;
    }
    const int len = (int)min<int64_t>(deg, kSample);
    const int64_t step = max<int64_t>(1, deg / kSample);
    int32_t lab[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        lab[q] = (j < len) ? labels[col[p0 + j * step]] : -1;
        if (j < len) sl[w][j] = lab[q];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the LDS writes of this wavefront are visible to its reads
    int cnt[4] = {0, 0, 0, 0};
    for (int j = 0; j < len; ++j) {
        const int32_t v = sl[w][j];       // same address for all lanes: broadcast
#pragma unroll
        for (int q = 0; q < 4; ++q) cnt[q] += (v == lab[q]) ? 1 : 0;
    }
    // best (count, smaller label): key = count * 2^32 + (2^31 - 1 - label)
    unsigned long long best = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (lab[q] >= 0) {
            const unsigned long long k = ((unsigned long long)cnt[q] << 32) | (unsigned long long)(0x7FFFFFFF - lab[q]);
            best = k > best ? k : best;
        }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(best, s, 64);
        best = o > best ? o : best;
    }
    if (lane == 0) {
        const int32_t nl = (int32_t)(0x7FFFFFFF - (int32_t)(best & 0xFFFFFFFFull));
        out[node] = nl;
        if (nl != mine) atomicAdd(moved, 1ull);
    }
}

__global__ __launch_bounds__(256) void lpa_iota_kernel(int32_t *__restrict__ a, int32_t *__restrict__ b, const int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        a[i] = (int32_t)i;
        if (b) b[i] = (int32_t)i;
    }
}

// perm[k] = node at position k (sorted by label, stable)  ->  order[node] = k ; heads of label runs are counted
__global__ __launch_bounds__(256) void lpa_finish_kernel(const int32_t *__restrict__ sorted_labels, const int32_t *__restrict__ perm,
                                                         const int64_t n, int64_t *__restrict__ order,
                                                         unsigned long long *__restrict__ n_comm) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        order[perm[k]] = k;
        if (k == 0 || sorted_labels[k] != sorted_labels[k - 1]) atomicAdd(n_comm, 1ull);
    }
}

// rows of a CSR into processing order: storage row k of the output is row perm[k] of the input, entries untouched
__global__ __launch_bounds__(256) void permute_len_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ perm,
                                                          const int64_t n, int64_t *__restrict__ len) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k <= n; k += (int64_t)gridDim.x * 256)
        len[k] = (k < n) ? rowptr[perm[k] + 1] - rowptr[perm[k]] : 0;
}

__global__ __launch_bounds__(256) void permute_copy_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                           const float *__restrict__ val, const int32_t *__restrict__ perm,
                                                           const int64_t n, const int64_t *__restrict__ out_rowptr,
                                                           int32_t *__restrict__ out_col, float *__restrict__ out_val) {
    const int lane = threadIdx.x & 63;
    const int64_t k = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;   // one wavefront per row
    if (k >= n) return;
    const int64_t src = rowptr[perm[k]], dst = out_rowptr[k], len = out_rowptr[k + 1] - dst;
    for (int64_t j = lane; j < len; j += 64) {
        out_col[dst + j] = col[src + j];
        out_val[dst + j] = val[src + j];
    }
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
        if (e != hipSuccess) return sgl::fail((int)e, "sgl_reorder_community: hipMalloc failed: %s", hipGetErrorString(e));
        ptrs.push_back(p);
        *out = reinterpret_cast<T *>(p);
        return SGL_OK;
    }
};

}  // namespace

SGL_EXPORT int sgl_reorder_community(const int64_t *d_rowptr, const int32_t *d_col, int64_t n, int rounds, int64_t *d_order,
                                     int64_t *h_info, void *stream) {
    SGL_REQUIRE(d_rowptr && d_order && n >= 0 && n < INT32_MAX, "sgl_reorder_community: bad arguments");
    SGL_REQUIRE(rounds >= 1 && rounds <= 64, "sgl_reorder_community: rounds must lie in [1, 64]");
    if (h_info) h_info[0] = h_info[1] = 0;
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_col, "sgl_reorder_community: NULL column array");
    SGL_REQUIRE(sgl::launch_fits((n + 3) / 4, 256), "sgl_reorder_community: too many nodes for one launch");
    hipStream_t st = sgl::as_stream(stream);
    Tmp tmp;
    int rc;
    int32_t *la = nullptr, *lb = nullptr, *iota = nullptr, *perm = nullptr, *sorted = nullptr;
    unsigned long long *counters = nullptr;   // [rounds] moved per round, [rounds] = communities
    if ((rc = tmp.alloc(&la, (size_t)n)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&lb, (size_t)n)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&iota, (size_t)n)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&perm, (size_t)n)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&sorted, (size_t)n)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&counters, (size_t)rounds + 1)) != SGL_OK) return rc;
    SGL_HIP_CHECK(hipMemsetAsync(counters, 0, sizeof(unsigned long long) * ((size_t)rounds + 1), st));
    const unsigned sgrid = (unsigned)std::min<int64_t>((n + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(lpa_iota_kernel, dim3(sgrid), dim3(256), 0, st, la, iota, n);
    SGL_HIP_CHECK(hipGetLastError());
    int32_t *cur = la, *nxt = lb;
    for (int it = 0; it < rounds; ++it) {
        hipLaunchKernelGGL(lpa_round_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, d_rowptr, d_col, n, (int64_t)0, cur, nxt, it,
                           it == rounds - 1 ? 1 : 0, counters + it);
        SGL_HIP_CHECK(hipGetLastError());
        std::swap(cur, nxt);
    }
    {   // stable sort of the nodes by label
        int bits = 1;
        while (bits < 31 && ((int64_t)1 << bits) < n) ++bits;
        size_t bytes = 0;
        SGL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, cur, sorted, iota, perm, (size_t)n, 0, bits, st));
        char *scratch = nullptr;
        if ((rc = tmp.alloc(&scratch, bytes)) != SGL_OK) return rc;
        SGL_HIP_CHECK(rocprim::radix_sort_pairs(scratch, bytes, cur, sorted, iota, perm, (size_t)n, 0, bits, st));
    }
    hipLaunchKernelGGL(lpa_finish_kernel, dim3(sgrid), dim3(256), 0, st, sorted, perm, n, d_order, counters + rounds);
    SGL_HIP_CHECK(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    SGL_HIP_CHECK(hipMemcpyAsync(&h[0], counters + rounds, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SGL_HIP_CHECK(hipMemcpyAsync(&h[1], counters + rounds - 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SGL_HIP_CHECK(hipStreamSynchronize(st));   // also keeps the temporaries alive until the work is done
    if (h_info) {
        h_info[0] = (int64_t)h[0];
        h_info[1] = (int64_t)h[1];
    }
    return SGL_OK;
}

// ONE round of the same label propagation for a ROW BLOCK of a matrix whose storage is row-sharded (sgl_amd/dist/redistribute.py):
// rows [row0, row0 + n_local) with local row pointers and GLOBAL column ids, the labels of ALL nodes in (int32 [n_global]), the new
// labels of the block's nodes out (int32 [n_local]); *d_moved (optional, a zeroed 64-bit word on the device) counts the nodes whose
// label changed.  The caller all-gathers the ranks' slices between rounds; `last` != 0 in the final round (every node may move).
// Exactly the per-node computation of sgl_reorder_community: running it over the blocks of a partition gives the whole-matrix labels.
SGL_EXPORT int sgl_reorder_lpa_round(const int64_t *d_rowptr, const int32_t *d_col, int64_t n_local, int64_t row0, int64_t n_global,
                                     const int32_t *d_labels, int32_t *d_out, int round, int last, uint64_t *d_moved, void *stream) {
    SGL_REQUIRE(n_local >= 0 && row0 >= 0 && row0 + n_local <= n_global && n_global < INT32_MAX, "sgl_reorder_lpa_round: bad sizes");
    SGL_REQUIRE(round >= 0 && round < 64, "sgl_reorder_lpa_round: round must lie in [0, 64)");
    if (n_local == 0) return SGL_OK;
    SGL_REQUIRE(d_rowptr && d_col && d_labels && d_out, "sgl_reorder_lpa_round: NULL arrays");
    SGL_REQUIRE(sgl::launch_fits((n_local + 3) / 4, 256), "sgl_reorder_lpa_round: too many rows for one launch");
    unsigned long long *moved = reinterpret_cast<unsigned long long *>(d_moved);
    Tmp tmp;
    hipStream_t st = sgl::as_stream(stream);
    if (!moved) {
        int rc;
        if ((rc = tmp.alloc(&moved, 1)) != SGL_OK) return rc;
        SGL_HIP_CHECK(hipMemsetAsync(moved, 0, sizeof(unsigned long long), st));
    }
    hipLaunchKernelGGL(lpa_round_kernel, dim3((unsigned)((n_local + 3) / 4)), dim3(256), 0, st, d_rowptr, d_col, n_local, row0, d_labels,
                       d_out, round, last ? 1 : 0, moved);
    SGL_HIP_CHECK(hipGetLastError());
    if (!d_moved) SGL_HIP_CHECK(hipStreamSynchronize(st));   // the scratch counter is freed on return
    return SGL_OK;
}

// out = the rows of (d_rowptr, d_col, d_val) in the order d_perm (storage row k = input row d_perm[k]); column ids and the
// order of every row's entries are unchanged.  out_rowptr [n+1], out_col / out_val [nnz] are the caller's.  With
// sgl_csr_set_rowmap(handle_of_out, d_perm) the permuted matrix computes exactly the products of the original one.
SGL_EXPORT int sgl_csr_permute_rows(const int64_t *d_rowptr, const int32_t *d_col, const float *d_val, int64_t n,
                                    const int32_t *d_perm, int64_t *d_out_rowptr, int32_t *d_out_col, float *d_out_val,
                                    void *stream) {
    SGL_REQUIRE(d_rowptr && d_perm && d_out_rowptr && n >= 0 && n < INT32_MAX, "sgl_csr_permute_rows: bad arguments");
    hipStream_t st = sgl::as_stream(stream);
    if (n == 0) {
        SGL_HIP_CHECK(hipMemsetAsync(d_out_rowptr, 0, sizeof(int64_t), st));
        return SGL_OK;
    }
    SGL_REQUIRE(sgl::launch_fits((n + 3) / 4, 256), "sgl_csr_permute_rows: too many rows for one launch");
    Tmp tmp;
    int rc;
    int64_t *len = nullptr;
    if ((rc = tmp.alloc(&len, (size_t)n + 1)) != SGL_OK) return rc;
    const unsigned sgrid = (unsigned)std::min<int64_t>((n + 256) / 256, 1 << 20);
    hipLaunchKernelGGL(permute_len_kernel, dim3(sgrid), dim3(256), 0, st, d_rowptr, d_perm, n, len);
    SGL_HIP_CHECK(hipGetLastError());
    size_t bytes = 0;
    SGL_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, len, d_out_rowptr, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st));
    char *scratch = nullptr;
    if ((rc = tmp.alloc(&scratch, bytes)) != SGL_OK) return rc;
    SGL_HIP_CHECK(rocprim::exclusive_scan(scratch, bytes, len, d_out_rowptr, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), st));
    if (d_col && d_val && d_out_col && d_out_val) {
        hipLaunchKernelGGL(permute_copy_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, d_rowptr, d_col, d_val, d_perm, n,
                           d_out_rowptr, d_out_col, d_out_val);
        SGL_HIP_CHECK(hipGetLastError());
    }
    SGL_HIP_CHECK(hipStreamSynchronize(st));   // the temporaries are freed on return
    return SGL_OK;
}
