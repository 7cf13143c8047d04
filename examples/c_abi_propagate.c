/* The drop-in boundary used from plain C: no Python, no PyTorch.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_abi_propagate.c -o c_abi_propagate \
 *       -L sgl_amd/csrc -lsgl_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/sgl_amd/csrc -Wl,-rpath,/opt/rocm/lib
 *
 * Builds a ring-with-chords graph on the host, uploads its CSR with hipMemcpy, normalises it on the device
 * (sgl_norm_prepare / sgl_norm_execute = adj_to_symmetric_norm, operators/utils.py:76-88), runs the k-hop chain
 * (sgl_spmm_chain_f32 = the loop of GraphOp.propagate, base_op.py:29-35) in strict summation order and compares every
 * hop bit for bit with the reference's own loop order (matmul.c:23-40) evaluated on the host.  Prints "C-ABI OK".
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "sgl_hip.h"

#define CHECK_HIP(x)                                                              \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 2;                                                             \
        }                                                                         \
    } while (0)
#define CHECK_SGL(x)                                                              \
    do {                                                                          \
        int rc_ = (x);                                                            \
        if (rc_ != 0) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, sgl_last_error());         \
            return 3;                                                             \
        }                                                                         \
    } while (0)

int main(void) {
    int n_dev = 0;
    CHECK_SGL(sgl_device_count(&n_dev));
    if (n_dev == 0) {
        fprintf(stderr, "no GPU\n");
        return 1;
    }
    enum { N = 5000, D = 37, K = 3, DEG = 6 };
    /* undirected ring with chords i <-> i+1, i+7, i+113 (mod N): already symmetric, no self loops, sorted columns */
    static const int offs[DEG] = {1, 7, 113, N - 113, N - 7, N - 1};
    int64_t *rowptr = (int64_t *)malloc((N + 1) * sizeof(int64_t));
    int32_t *col = (int32_t *)malloc((size_t)N * DEG * sizeof(int32_t));
    float *val = (float *)malloc((size_t)N * DEG * sizeof(float));
    for (int i = 0; i <= N; ++i) rowptr[i] = (int64_t)i * DEG;
    for (int i = 0; i < N; ++i) {
        int32_t tmp[DEG];
        for (int k = 0; k < DEG; ++k) tmp[k] = (int32_t)((i + offs[k]) % N);
        for (int a = 1; a < DEG; ++a) /* insertion sort: CSR columns ascending */
            for (int b = a; b > 0 && tmp[b - 1] > tmp[b]; --b) {
                int32_t t = tmp[b];
                tmp[b] = tmp[b - 1];
                tmp[b - 1] = t;
            }
        for (int k = 0; k < DEG; ++k) {
            col[(size_t)i * DEG + k] = tmp[k];
            val[(size_t)i * DEG + k] = 1.0f;
        }
    }
    float *x = (float *)malloc((size_t)N * D * sizeof(float));
    for (size_t i = 0; i < (size_t)N * D; ++i) x[i] = (float)((double)((i * 2654435761u) % 2001u) / 1000.0 - 1.0);

    int64_t *d_rowptr;
    int32_t *d_col;
    float *d_val;
    CHECK_HIP(hipMalloc((void **)&d_rowptr, (N + 1) * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&d_col, (size_t)N * DEG * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_val, (size_t)N * DEG * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_rowptr, rowptr, (N + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_col, col, (size_t)N * DEG * sizeof(int32_t), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_val, val, (size_t)N * DEG * sizeof(float), hipMemcpyHostToDevice));

    /* A_hat = D^{r-1} (A + I)^T D^{-r}, r = 0.5: nnz grows by the missing diagonal */
    int64_t nnz_out = 0;
    CHECK_SGL(sgl_norm_prepare(N, (int64_t)N * DEG, d_rowptr, d_col, &nnz_out, NULL));
    if (nnz_out != (int64_t)N * (DEG + 1)) {
        fprintf(stderr, "unexpected nnz(A_hat) %lld\n", (long long)nnz_out);
        return 4;
    }
    int64_t *d_np;
    int32_t *d_nc;
    float *d_nv;
    CHECK_HIP(hipMalloc((void **)&d_np, (N + 1) * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&d_nc, (size_t)nnz_out * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_nv, (size_t)nnz_out * sizeof(float)));
    CHECK_SGL(sgl_norm_execute(N, (int64_t)N * DEG, d_rowptr, d_col, d_val, 0.5, 0, 0.0, nnz_out, d_np, d_nc, d_nv, NULL, NULL));

    sgl_csr_t *csr = NULL;
    CHECK_SGL(sgl_csr_create(&csr, N, N, nnz_out, d_np, d_nc, d_nv, SGL_CSR_STRICT_ORDER, 0, 0, NULL));
    float *d_x, *d_y[K];
    int64_t ldy[K];
    CHECK_HIP(hipMalloc((void **)&d_x, (size_t)N * D * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_x, x, (size_t)N * D * sizeof(float), hipMemcpyHostToDevice));
    for (int h = 0; h < K; ++h) {
        CHECK_HIP(hipMalloc((void **)&d_y[h], (size_t)N * D * sizeof(float)));
        ldy[h] = D;
    }
    CHECK_SGL(sgl_spmm_chain_f32(csr, K, d_x, D, d_y, ldy, D, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    /* host check in the reference's loop order on the SAME normalised matrix (downloaded) */
    int64_t *np = (int64_t *)malloc((N + 1) * sizeof(int64_t));
    int32_t *nc = (int32_t *)malloc((size_t)nnz_out * sizeof(int32_t));
    float *nv = (float *)malloc((size_t)nnz_out * sizeof(float));
    CHECK_HIP(hipMemcpy(np, d_np, (N + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(nc, d_nc, (size_t)nnz_out * sizeof(int32_t), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(nv, d_nv, (size_t)nnz_out * sizeof(float), hipMemcpyDeviceToHost));
    /* every row has 7 entries 1/7 here (degree 6 + self loop): a value check of the normalisation itself */
    for (int64_t j = 0; j < nnz_out; ++j)
        if (fabsf(nv[j] - 1.0f / 7.0f) > 1e-7f) {
            fprintf(stderr, "normalised value %g at %lld\n", nv[j], (long long)j);
            return 5;
        }
    float *cur = x, *nxt = (float *)malloc((size_t)N * D * sizeof(float)), *got = (float *)malloc((size_t)N * D * sizeof(float));
    float *spare = (float *)malloc((size_t)N * D * sizeof(float));
    for (int h = 0; h < K; ++h) {
        memset(nxt, 0, (size_t)N * D * sizeof(float));
        for (int i = 0; i < N; ++i)
            for (int64_t j = np[i]; j < np[i + 1]; ++j)
                for (int k = 0; k < D; ++k) nxt[(size_t)i * D + k] = fmaf(nv[j], cur[(size_t)nc[j] * D + k], nxt[(size_t)i * D + k]);
        CHECK_HIP(hipMemcpy(got, d_y[h], (size_t)N * D * sizeof(float), hipMemcpyDeviceToHost));
        if (memcmp(got, nxt, (size_t)N * D * sizeof(float)) != 0) {
            fprintf(stderr, "hop %d differs from the reference loop order\n", h + 1);
            return 6;
        }
        float *t = (cur == x) ? spare : cur;
        cur = nxt;
        nxt = t;
    }
    CHECK_SGL(sgl_csr_destroy(csr));
    printf("C-ABI OK: %d hops of a %d x %d A_hat (%lld nnz) x %d features, bit-identical to the reference loop order\n", K,
           N, N, (long long)nnz_out, D);
    return 0;
}
