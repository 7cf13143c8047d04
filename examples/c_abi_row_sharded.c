/* The contract multi-GPU layout driven from plain C: no Python, no PyTorch, no torch.distributed.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_abi_row_sharded.c -o c_abi_row_sharded \
 *       -L sgl_amd/csrc -lsgl_hip -L /opt/rocm/lib -lamdhip64 -lrccl -lm -Wl,-rpath,$PWD/sgl_amd/csrc -Wl,-rpath,/opt/rocm/lib
 *
 * One process drives every visible GPU (G >= 1; RCCL communicators from ncclCommInitAll).  GPU g owns rows [b[g], b[g+1]) of a
 * symmetric ring-with-chords graph and NOTHING else of it: it normalises its own block (sgl_norm_block_prepare / _build /
 * _scale; the only global quantity is the degree vector, assembled from the blocks' row sums), builds the SpMM plan of its rows
 * (sgl_csr_create on a rectangular block) and runs K hops: sgl_spmm_f32 writes its rows of the next feature replica,
 * sgl_allgather_rows fetches everybody else's over the caller's communicators (one grouped batch of ncclSend / ncclRecv per
 * GPU, all links at once).  Every hop is compared bit for bit with the reference's loop order (matmul.c:23-40) on the host.
 * Prints "C-ABI row-sharded OK (G GPUs)".  With G = 1 the exchange is a no-op and the flow is the single-GPU one. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

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
#define CHECK_NCCL(x)                                                             \
    do {                                                                          \
        ncclResult_t r_ = (x);                                                    \
        if (r_ != ncclSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));              \
            return 4;                                                             \
        }                                                                         \
    } while (0)

enum { N = 6000, D = 40, K = 3, DEG = 6, MAXG = 8 };

int main(void) {
    int G = 0;
    CHECK_SGL(sgl_device_count(&G));
    if (G == 0) {
        fprintf(stderr, "no GPU\n");
        return 1;
    }
    if (G > MAXG) G = MAXG;
    /* ---- the graph on the host: undirected ring with chords, unit weights, sorted columns ---- */
    static const int offs[DEG] = {1, 7, 113, N - 113, N - 7, N - 1};
    int64_t *rowptr = (int64_t *)malloc((N + 1) * sizeof(int64_t));
    int32_t *col = (int32_t *)malloc((size_t)N * DEG * sizeof(int32_t));
    float *val = (float *)malloc((size_t)N * DEG * sizeof(float));
    for (int i = 0; i <= N; ++i) rowptr[i] = (int64_t)i * DEG;
    for (int i = 0; i < N; ++i) {
        int32_t tmp[DEG];
        for (int k = 0; k < DEG; ++k) tmp[k] = (int32_t)((i + offs[k]) % N);
        for (int a = 1; a < DEG; ++a)
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
    float *x0 = (float *)malloc((size_t)N * D * sizeof(float));
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < D; ++k) x0[(size_t)i * D + k] = (float)(((i * 31 + k * 17) % 101) - 50) / 64.0f;

    int64_t b[MAXG + 1];
    for (int g = 0; g <= G; ++g) b[g] = (int64_t)N * g / G;

    /* ---- one communicator and one stream per GPU ---- */
    ncclComm_t comm[MAXG];
    hipStream_t st[MAXG];
    int devs[MAXG];
    for (int g = 0; g < G; ++g) devs[g] = g;
    CHECK_NCCL(ncclCommInitAll(comm, G, devs));

    /* ---- per GPU: my rows of A, T' = A + I for them, their row sums ---- */
    int64_t *d_rp[MAXG], *d_orp[MAXG], m[MAXG];
    int32_t *d_col[MAXG], *d_ocol[MAXG];
    float *d_val[MAXG], *d_oval[MAXG], *d_x[MAXG][2];
    double *d_t64[MAXG], *d_rs[MAXG], *d_L[MAXG], *d_R[MAXG];
    sgl_csr_t *h[MAXG];
    double *deg = (double *)malloc(N * sizeof(double));
    for (int g = 0; g < G; ++g) {
        CHECK_HIP(hipSetDevice(g));
        CHECK_HIP(hipStreamCreate(&st[g]));
        const int64_t nl = b[g + 1] - b[g], nnz = rowptr[b[g + 1]] - rowptr[b[g]];
        int64_t *rp_local = (int64_t *)malloc((nl + 1) * sizeof(int64_t));
        for (int64_t i = 0; i <= nl; ++i) rp_local[i] = rowptr[b[g] + i] - rowptr[b[g]];
        CHECK_HIP(hipMalloc((void **)&d_rp[g], (nl + 1) * sizeof(int64_t)));
        CHECK_HIP(hipMalloc((void **)&d_col[g], (size_t)(nnz + 1) * sizeof(int32_t)));
        CHECK_HIP(hipMalloc((void **)&d_val[g], (size_t)(nnz + 1) * sizeof(float)));
        CHECK_HIP(hipMemcpy(d_rp[g], rp_local, (nl + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_col[g], col + rowptr[b[g]], (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_val[g], val + rowptr[b[g]], (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));
        free(rp_local);
        CHECK_SGL(sgl_norm_block_prepare(nl, b[g], nnz, d_rp[g], d_col[g], &m[g], st[g]));
        CHECK_HIP(hipMalloc((void **)&d_orp[g], (nl + 1) * sizeof(int64_t)));
        CHECK_HIP(hipMalloc((void **)&d_ocol[g], (size_t)(m[g] + 1) * sizeof(int32_t)));
        CHECK_HIP(hipMalloc((void **)&d_oval[g], (size_t)(m[g] + 1) * sizeof(float)));
        CHECK_HIP(hipMalloc((void **)&d_t64[g], (size_t)(m[g] + 1) * sizeof(double)));
        CHECK_HIP(hipMalloc((void **)&d_rs[g], (size_t)(nl + 1) * sizeof(double)));
        CHECK_SGL(sgl_norm_block_build(nl, b[g], nnz, d_rp[g], d_col[g], d_val[g], m[g], d_orp[g], d_ocol[g], d_t64[g], d_rs[g], st[g]));
        /* symmetric graph: deg = rowsum(A + I); the blocks' row sums tile the degree vector (an all-gather in a real job) */
        CHECK_HIP(hipMemcpy(deg + b[g], d_rs[g], (size_t)nl * sizeof(double), hipMemcpyDeviceToHost));
    }
    /* ---- degree powers exactly as operators/utils.py:79-84 (r = 0.5), then the scaling and the plans ---- */
    const double r = 0.5;
    double *L = (double *)malloc(N * sizeof(double)), *R = (double *)malloc(N * sizeof(double));
    for (int i = 0; i < N; ++i) {
        L[i] = pow(deg[i], r - 1.0);
        R[i] = pow(deg[i], -r);
        if (isinf(L[i])) L[i] = 0.0;
        if (isinf(R[i])) R[i] = 0.0;
    }
    float *a_hat = (float *)malloc((size_t)(N * (DEG + 1)) * sizeof(float));     /* host copy for the check below */
    int32_t *a_col = (int32_t *)malloc((size_t)(N * (DEG + 1)) * sizeof(int32_t));
    int64_t *a_rp = (int64_t *)malloc((N + 1) * sizeof(int64_t));
    int64_t off = 0;
    a_rp[0] = 0;
    for (int g = 0; g < G; ++g) {
        CHECK_HIP(hipSetDevice(g));
        const int64_t nl = b[g + 1] - b[g];
        CHECK_HIP(hipMalloc((void **)&d_L[g], (size_t)(nl + 1) * sizeof(double)));
        CHECK_HIP(hipMalloc((void **)&d_R[g], N * sizeof(double)));
        CHECK_HIP(hipMemcpy(d_L[g], L + b[g], (size_t)nl * sizeof(double), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_R[g], R, N * sizeof(double), hipMemcpyHostToDevice));
        CHECK_SGL(sgl_norm_block_scale(nl, b[g], d_orp[g], d_ocol[g], d_t64[g], d_L[g], d_R[g], 0, 0.0, d_oval[g], NULL, st[g]));
        CHECK_SGL(sgl_csr_create(&h[g], nl, N, m[g], d_orp[g], d_ocol[g], d_oval[g], SGL_CSR_STRICT_ORDER, 0, 0, st[g]));
        int64_t *rp_h = (int64_t *)malloc((nl + 1) * sizeof(int64_t));
        CHECK_HIP(hipMemcpy(rp_h, d_orp[g], (nl + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(a_col + off, d_ocol[g], (size_t)m[g] * sizeof(int32_t), hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(a_hat + off, d_oval[g], (size_t)m[g] * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t i = 1; i <= nl; ++i) a_rp[b[g] + i] = off + rp_h[i];
        off += m[g];
        free(rp_h);
        for (int s = 0; s < 2; ++s) CHECK_HIP(hipMalloc((void **)&d_x[g][s], (size_t)N * D * sizeof(float)));
        CHECK_HIP(hipMemcpy(d_x[g][0], x0, (size_t)N * D * sizeof(float), hipMemcpyHostToDevice));
    }
    /* ---- K hops: my rows of the next replica, then everybody else's ---- */
    float *ref = (float *)malloc((size_t)N * D * sizeof(float)), *cur = (float *)malloc((size_t)N * D * sizeof(float));
    float *got = (float *)malloc((size_t)N * D * sizeof(float));
    memcpy(cur, x0, (size_t)N * D * sizeof(float));
    for (int k = 0; k < K; ++k) {
        for (int g = 0; g < G; ++g) {
            CHECK_HIP(hipSetDevice(g));
            CHECK_SGL(sgl_spmm_f32(h[g], d_x[g][k % 2], D, d_x[g][(k + 1) % 2] + b[g] * D, D, D, 0, st[g]));
        }
        CHECK_NCCL(ncclGroupStart());            /* one thread drives all GPUs: the per-GPU batches form one group */
        for (int g = 0; g < G; ++g) {
            CHECK_HIP(hipSetDevice(g));
            CHECK_SGL(sgl_allgather_rows(comm[g], g, G, b, d_x[g][(k + 1) % 2], D, st[g]));
        }
        CHECK_NCCL(ncclGroupEnd());
        /* the reference's loop order on the host (matmul.c:23-40): one fmaf chain per (row, column) in CSR order */
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < D; ++c) {
                float acc = 0.0f;
                for (int64_t j = a_rp[i]; j < a_rp[i + 1]; ++j) acc = fmaf(a_hat[j], cur[(size_t)a_col[j] * D + c], acc);
                ref[(size_t)i * D + c] = acc;
            }
        for (int g = 0; g < G; ++g) {           /* EVERY GPU must now hold the complete hop, bit for bit */
            CHECK_HIP(hipSetDevice(g));
            CHECK_HIP(hipStreamSynchronize(st[g]));
            CHECK_HIP(hipMemcpy(got, d_x[g][(k + 1) % 2], (size_t)N * D * sizeof(float), hipMemcpyDeviceToHost));
            if (memcmp(got, ref, (size_t)N * D * sizeof(float)) != 0) {
                fprintf(stderr, "hop %d on GPU %d differs from the reference order\n", k + 1, g);
                return 5;
            }
        }
        memcpy(cur, ref, (size_t)N * D * sizeof(float));
    }
    for (int g = 0; g < G; ++g) {
        CHECK_HIP(hipSetDevice(g));
        sgl_csr_destroy(h[g]);
        ncclCommDestroy(comm[g]);
    }
    printf("C-ABI row-sharded OK (%d GPU%s, exchange backend: %s)\n", G, G == 1 ? "" : "s", sgl_exchange_backend());
    return 0;
}
