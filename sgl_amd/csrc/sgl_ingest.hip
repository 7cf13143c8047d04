// COO -> canonical CSR on device: the ingest step in front of the hot path.
//
// Reference behaviour being reproduced: sgl/data/base_data.py:29 builds the adjacency with
//   scipy.sparse.csr_matrix((edge_weight, (row, col)), shape=(num_node, num_node))
// i.e. float32 values, duplicate (row, col) pairs SUMMED (in storage order), columns sorted inside each row --
// the only route by which a raw `adj_matrix.npz{row,col,data}` dump (dataset/custom_dataset.py:52-54) becomes the
// matrix GraphOp.propagate consumes.  Here: 64-bit (row,col) keys, a STABLE radix sort (rocPRIM), run heads by
// comparison with the predecessor, exclusive scan for output slots, and one sequential fp32 sum per run (runs are
// short; sequential = the order scipy adds them in) -- int64-safe, no atomics, deterministic.
#include "sgl_common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

__global__ __launch_bounds__(256) void make_keys_kernel(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                                        int64_t nnz, int64_t n_rows, int64_t n_cols,
                                                        unsigned long long *__restrict__ keys, uint32_t *__restrict__ iota,
                                                        int *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    const int64_t r = row[i], c = col[i];
    if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) {
        *bad = 1;
        keys[i] = ~0ull;
    } else {
        keys[i] = ((unsigned long long)r << 32) | (unsigned long long)c;
    }
    iota[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void run_heads_kernel(const unsigned long long *__restrict__ keys, int64_t nnz,
                                                        int64_t *__restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nnz) return;
    head[i] = (i < nnz && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
}

__global__ __launch_bounds__(256) void fold_runs_kernel(const unsigned long long *__restrict__ keys,
                                                        const uint32_t *__restrict__ perm, const float *__restrict__ val,
                                                        const int64_t *__restrict__ head, const int64_t *__restrict__ slot,
                                                        int64_t nnz, int32_t *__restrict__ out_col,
                                                        float *__restrict__ out_val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz || !head[i]) return;
    const unsigned long long k = keys[i];
    float acc = val[perm[i]];
    for (int64_t j = i + 1; j < nnz && keys[j] == k; ++j) acc = __fadd_rn(acc, val[perm[j]]);  // storage order
    const int64_t o = slot[i];
    out_col[o] = (int32_t)(k & 0xffffffffull);
    out_val[o] = acc;
}

// rowptr[r] = number of unique entries whose row < r  = slot of the first run head with key >= (r << 32)
__global__ __launch_bounds__(256) void rowptr_kernel(const unsigned long long *__restrict__ keys, const int64_t *__restrict__ slot,
                                                     int64_t nnz, int64_t n_rows, int64_t n_unique,
                                                     int64_t *__restrict__ rowptr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    const unsigned long long target = (unsigned long long)r << 32;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    // `lo` is the first sorted element of row >= r, necessarily a run head (or the end)
    rowptr[r] = (lo < nnz) ? slot[lo] : n_unique;
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
        if (e != hipSuccess) return sgl::fail((int)e, "sgl_coo_to_csr: hipMalloc failed: %s", hipGetErrorString(e));
        ptrs.push_back(p);
        *out = reinterpret_cast<T *>(p);
        return SGL_OK;
    }
};

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

SGL_EXPORT int sgl_coo_to_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_row, const int64_t *d_col,
                              const float *d_val, int64_t *d_out_rowptr, int32_t *d_out_col, float *d_out_val,
                              int64_t *h_nnz_out, void *stream) {
    SGL_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "sgl_coo_to_csr: negative size");
    SGL_REQUIRE(n_rows < INT32_MAX && n_cols < INT32_MAX, "sgl_coo_to_csr: n_rows / n_cols must be < 2^31");
    SGL_REQUIRE(nnz < (int64_t)UINT32_MAX, "sgl_coo_to_csr: more than 2^32-1 input entries: shard the edge list");
    SGL_REQUIRE(d_out_rowptr && h_nnz_out, "sgl_coo_to_csr: NULL outputs");
    hipStream_t st = sgl::as_stream(stream);
    if (nnz == 0) {
        SGL_HIP_CHECK(hipMemsetAsync(d_out_rowptr, 0, sizeof(int64_t) * (size_t)(n_rows + 1), st));
        SGL_HIP_CHECK(hipStreamSynchronize(st));
        *h_nnz_out = 0;
        return SGL_OK;
    }
    SGL_REQUIRE(d_row && d_col && d_val && d_out_col && d_out_val, "sgl_coo_to_csr: NULL arrays");
    Tmp tmp;
    int rc;
    unsigned long long *keys = nullptr, *keys_sorted = nullptr;
    uint32_t *iota = nullptr, *perm = nullptr;
    int64_t *head = nullptr, *slot = nullptr;
    int *bad = nullptr;
    if ((rc = tmp.alloc(&keys, (size_t)nnz)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&keys_sorted, (size_t)nnz)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&iota, (size_t)nnz)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&perm, (size_t)nnz)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&head, (size_t)nnz + 1)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&slot, (size_t)nnz + 1)) != SGL_OK) return rc;
    if ((rc = tmp.alloc(&bad, 1)) != SGL_OK) return rc;
    SGL_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), st));
    hipLaunchKernelGGL(make_keys_kernel, dim3(blocks_for(nnz)), dim3(256), 0, st, d_row, d_col, nnz, n_rows, n_cols, keys, iota, bad);
    SGL_HIP_CHECK(hipGetLastError());
    {
        int rbits = 1, cbits = 1;
        while (rbits < 31 && ((int64_t)1 << rbits) < n_rows) ++rbits;
        while (cbits < 32 && ((int64_t)1 << cbits) < n_cols) ++cbits;
        (void)cbits;  // the column bits sit at 0..31, rows at 32..32+rbits: sort all of them
        const int end_bit = 32 + rbits;
        size_t bytes = 0;
        SGL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys, keys_sorted, iota, perm, (size_t)nnz, 0, end_bit, st));
        char *scratch = nullptr;
        if ((rc = tmp.alloc(&scratch, bytes)) != SGL_OK) return rc;
        SGL_HIP_CHECK(rocprim::radix_sort_pairs(scratch, bytes, keys, keys_sorted, iota, perm, (size_t)nnz, 0, end_bit, st));
    }
    hipLaunchKernelGGL(run_heads_kernel, dim3(blocks_for(nnz + 1)), dim3(256), 0, st, keys_sorted, nnz, head);
    SGL_HIP_CHECK(hipGetLastError());
    {
        size_t bytes = 0;
        SGL_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, head, slot, (int64_t)0, (size_t)nnz + 1, rocprim::plus<int64_t>(), st));
        char *scratch = nullptr;
        if ((rc = tmp.alloc(&scratch, bytes)) != SGL_OK) return rc;
        SGL_HIP_CHECK(rocprim::exclusive_scan(scratch, bytes, head, slot, (int64_t)0, (size_t)nnz + 1, rocprim::plus<int64_t>(), st));
    }
    int64_t n_unique = 0;
    int h_bad = 0;
    SGL_HIP_CHECK(hipMemcpyAsync(&n_unique, slot + nnz, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    SGL_HIP_CHECK(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    SGL_HIP_CHECK(hipStreamSynchronize(st));
    if (h_bad) return sgl::fail(SGL_ERR_INVALID, "sgl_coo_to_csr: a row/col index lies outside [0, n)");
    hipLaunchKernelGGL(fold_runs_kernel, dim3(blocks_for(nnz)), dim3(256), 0, st, keys_sorted, perm, d_val, head, slot, nnz,
                       d_out_col, d_out_val);
    SGL_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(rowptr_kernel, dim3(blocks_for(n_rows + 1)), dim3(256), 0, st, keys_sorted, slot, nnz, n_rows, n_unique,
                       d_out_rowptr);
    SGL_HIP_CHECK(hipGetLastError());
    SGL_HIP_CHECK(hipStreamSynchronize(st));
    *h_nnz_out = n_unique;
    return SGL_OK;
}
