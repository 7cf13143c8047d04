// Reference-signature host shims: a ctypes caller written against the reference's libmatmul.so /
// libcudamatmul.so (sgl/operators/utils.py:10-73) can load libsgl_hip.so instead and run on the MI355X.
// Host pointers in, host pointers out: upload -> device SpMM -> download.  Synchronous by construction
// (the reference calls are, too).  The device-resident API (sgl_csr_create / sgl_spmm_f32) is the fast path.
#include "sgl_common.h"

namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        if (bytes == 0) bytes = 4;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return sgl::fail((int)e, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return SGL_OK;
    }
};

int host_spmm(float *answer, const float *data, const int *indices, const int *indptr, const float *mat,
              int mat_row, int mat_col, int accumulate) {
    SGL_REQUIRE(mat_row >= 0 && mat_col >= 0, "FloatCSRMulDense*: negative size");
    if (mat_row == 0 || mat_col == 0) return SGL_OK;
    SGL_REQUIRE(answer && indptr && mat, "FloatCSRMulDense*: NULL argument");
    int ndev = 0;
    sgl_device_count(&ndev);
    if (ndev <= 0) return sgl::fail(SGL_ERR_NO_DEVICE, "FloatCSRMulDense*: no HIP device available");
    // The reference passes the SAME n for rows of A, columns of A and rows of `mat` (square adjacency,
    // utils.py:36: mat_row, mat_col = feature.shape).
    const int64_t n = mat_row, d = mat_col;
    const int64_t nnz = indptr[n];
    SGL_REQUIRE(nnz >= 0 && indptr[0] == 0, "FloatCSRMulDense*: bad indptr");
    SGL_REQUIRE(nnz == 0 || (data && indices), "FloatCSRMulDense*: NULL data/indices");
    std::vector<int64_t> rp64((size_t)n + 1);
    for (int64_t i = 0; i <= n; ++i) rp64[i] = indptr[i];

    DevBuf d_rp, d_col, d_val, d_x, d_y;
    int rc;
    if ((rc = d_rp.alloc(rp64.size() * sizeof(int64_t))) != SGL_OK) return rc;
    if ((rc = d_col.alloc((size_t)nnz * sizeof(int32_t))) != SGL_OK) return rc;
    if ((rc = d_val.alloc((size_t)nnz * sizeof(float))) != SGL_OK) return rc;
    if ((rc = d_x.alloc((size_t)n * d * sizeof(float))) != SGL_OK) return rc;
    if ((rc = d_y.alloc((size_t)n * d * sizeof(float))) != SGL_OK) return rc;
    SGL_HIP_CHECK(hipMemcpy(d_rp.p, rp64.data(), rp64.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    if (nnz) {
        SGL_HIP_CHECK(hipMemcpy(d_col.p, indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        SGL_HIP_CHECK(hipMemcpy(d_val.p, data, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));
    }
    SGL_HIP_CHECK(hipMemcpy(d_x.p, mat, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice));
    if (accumulate) SGL_HIP_CHECK(hipMemcpy(d_y.p, answer, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice));

    sgl_csr_t *h = nullptr;
    // strict order: the shim promises the reference's exact per-row fmaf chain
    rc = sgl_csr_create(&h, n, n, nnz, (const int64_t *)d_rp.p, (const int32_t *)d_col.p, (const float *)d_val.p,
                        SGL_CSR_STRICT_ORDER, 0, 0, nullptr);
    if (rc != SGL_OK) return rc;
    rc = sgl_spmm_f32(h, (const float *)d_x.p, d, (float *)d_y.p, d, d, accumulate, nullptr);
    if (rc == SGL_OK) {
        hipError_t e = hipMemcpy(answer, d_y.p, (size_t)n * d * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = sgl::fail((int)e, "download failed: %s", hipGetErrorString(e));
    }
    sgl_csr_destroy(h);
    return rc;
}

}  // namespace

// matmul.h:5 / matmul.c:23-40: answer += A . mat
SGL_EXPORT void FloatCSRMulDenseOMP(float answer[], float data[], int indices[], int indptr[], float mat[], int mat_row,
                                    int mat_col) {
    sgl::set_error("%s", "");
    (void)host_spmm(answer, data, indices, indptr, mat, mat_row, mat_col, /*accumulate=*/1);
}

// cudamatmul.c:28-146: answer = A . mat (alpha = 1, beta = 0); EXIT_SUCCESS / EXIT_FAILURE
SGL_EXPORT int FloatCSRMulDense(float answer[], int data_nnz, float data[], int indices[], int indptr[], float mat[],
                                int mat_row, int mat_col) {
    sgl::set_error("%s", "");
    if (indptr && mat_row >= 0 && indptr[mat_row] != data_nnz) {
        sgl::set_error("FloatCSRMulDense: data_nnz=%d but indptr[mat_row]=%d", data_nnz, indptr[mat_row]);
        return 1;
    }
    return host_spmm(answer, data, indices, indptr, mat, mat_row, mat_col, /*accumulate=*/0) == SGL_OK ? 0 : 1;
}
