/* The sweep entry points of the boundary from plain C (no Python, no PyTorch): what a PaSca-style search over graph operators and
 * the NAFS tasks' adaptive k-hop selection do with the path (sgl/search/search_config.py:14-15, sgl/tasks/node_clustering.py:139-258).
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_abi_sweeps.c -o c_abi_sweeps \
 *       -L sgl_amd/csrc -lsgl_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/sgl_amd/csrc -Wl,-rpath,/opt/rocm/lib
 *
 * 1. (r, alpha) sweep: T + I and the degrees ONCE (sgl_norm_block_prepare / sgl_norm_block_build on the whole matrix as one row
 *    block), per r the degree powers on the device (sgl_norm_degree_powers) and ONE gather pass that also keeps the fp64 Laplacian
 *    (sgl_norm_block_scale), per further alpha a pure stream (sgl_norm_block_mix = PprGraphOp's (1 - alpha) A_hat + alpha I,
 *    ppr_graph_op.py:20): every value checked against the closed form of this regular graph.
 * 2. hop sweep: K hops propagated once (sgl_spmm_chain_f32), then the NAFS aggregate of EVERY prefix X_0..X_h from one pass over the
 *    hop matrices (sgl_nafs_prefix_f32), each compared with over_smooth_distance_op.py:11-33 evaluated on the host in double.
 * Prints "C-ABI sweeps OK". */
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

enum { N = 4000, D = 37, LD = 40, K = 6, DEG = 6 };

int main(void) {
    int n_dev = 0;
    CHECK_SGL(sgl_device_count(&n_dev));
    if (n_dev == 0) {
        fprintf(stderr, "no GPU\n");
        return 1;
    }
    /* undirected ring with chords i <-> i+1, i+7, i+113 (mod N): symmetric, no self loops; columns sorted per row */
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
    const int64_t nnz = (int64_t)N * DEG;
    int64_t *d_rowptr, *d_np;
    int32_t *d_col, *d_nc;
    float *d_val, *d_lap, *d_ppr;
    double *d_t64, *d_deg, *d_left, *d_right, *d_hat64;
    CHECK_HIP(hipMalloc((void **)&d_rowptr, (N + 1) * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&d_col, (size_t)nnz * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_val, (size_t)nnz * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_rowptr, rowptr, (N + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_col, col, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_val, val, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));

    /* ---- 1. (r, alpha) sweep --------------------------------------------------------------------------------------------- */
    int64_t m = 0; /* nnz(A + I) */
    CHECK_SGL(sgl_norm_block_prepare(N, 0, nnz, d_rowptr, d_col, &m, NULL));
    if (m != (int64_t)N * (DEG + 1)) {
        fprintf(stderr, "unexpected nnz(A + I) %lld\n", (long long)m);
        return 4;
    }
    CHECK_HIP(hipMalloc((void **)&d_np, (N + 1) * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&d_nc, (size_t)m * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_t64, (size_t)m * sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&d_deg, N * sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&d_left, N * sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&d_right, N * sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&d_hat64, (size_t)m * sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&d_lap, (size_t)m * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_ppr, (size_t)m * sizeof(float)));
    CHECK_SGL(sgl_norm_block_build(N, 0, nnz, d_rowptr, d_col, d_val, m, d_np, d_nc, d_t64, d_deg, NULL)); /* once per graph */
    float *h_val = (float *)malloc((size_t)m * sizeof(float));
    int32_t *h_nc = (int32_t *)malloc((size_t)m * sizeof(int32_t));
    int64_t *h_np = (int64_t *)malloc((N + 1) * sizeof(int64_t));
    CHECK_HIP(hipMemcpy(h_nc, d_nc, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_np, d_np, (N + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    static const double rs[2] = {0.5, 0.3}, alphas[3] = {0.1, 0.2, 0.3};
    for (int ri = 0; ri < 2; ++ri) {
        /* every node has degree 7 here: A_hat entries are 7^(r-1) * 7^(-r) = 1/7 whatever r (a value check, not a tautology:
         * left and right factors come from two different arrays) */
        CHECK_SGL(sgl_norm_degree_powers(N, d_deg, rs[ri], d_left, d_right, NULL));
        CHECK_SGL(sgl_norm_block_scale(N, 0, d_np, d_nc, d_t64, d_left, d_right, 0, 0.0, d_lap, d_hat64, NULL));
        CHECK_HIP(hipMemcpy(h_val, d_lap, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t j = 0; j < m; ++j)
            if (fabs((double)h_val[j] - 1.0 / 7.0) > 2e-8) {
                fprintf(stderr, "r=%g: Laplacian value %g at %lld\n", rs[ri], h_val[j], (long long)j);
                return 5;
            }
        for (int ai = 0; ai < 3; ++ai) { /* the alpha sweep: one stream over the kept fp64 Laplacian per alpha */
            CHECK_SGL(sgl_norm_block_mix(N, 0, d_np, d_nc, d_hat64, alphas[ai], d_ppr, NULL, NULL));
            CHECK_HIP(hipMemcpy(h_val, d_ppr, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
            for (int i = 0; i < N; ++i)
                for (int64_t j = h_np[i]; j < h_np[i + 1]; ++j) {
                    const double want = (1.0 - alphas[ai]) / 7.0 + (h_nc[j] == i ? alphas[ai] : 0.0);
                    if (fabs((double)h_val[j] - want) > 6e-8 * (want > 1.0 ? want : 1.0)) {
                        fprintf(stderr, "r=%g alpha=%g: value %g, want %g at (%d, %d)\n", rs[ri], alphas[ai], h_val[j], want, i,
                                h_nc[j]);
                        return 6;
                    }
                }
        }
    }

    /* ---- 2. hop sweep ----------------------------------------------------------------------------------------------------- */
    sgl_csr_t *csr = NULL;
    CHECK_SGL(sgl_csr_create(&csr, N, N, m, d_np, d_nc, d_lap, 0, 0, 0, NULL)); /* the r = 0.3 Laplacian (same values) */
    float *x = (float *)calloc((size_t)N * LD, sizeof(float)); /* rows on a 16-byte aligned pitch, pad columns zero */
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < D; ++k) x[(size_t)i * LD + k] = (float)((double)((((size_t)i * D + k) * 2654435761u) % 2001u) / 1000.0 - 0.2);
    float *d_hop[K + 1], *d_out[K + 1];
    int64_t ld[K + 1];
    for (int h = 0; h <= K; ++h) {
        CHECK_HIP(hipMalloc((void **)&d_hop[h], (size_t)N * LD * sizeof(float)));
        CHECK_HIP(hipMalloc((void **)&d_out[h], (size_t)N * LD * sizeof(float)));
        CHECK_HIP(hipMemset(d_hop[h], 0, (size_t)N * LD * sizeof(float)));
        ld[h] = LD;
    }
    CHECK_HIP(hipMemcpy(d_hop[0], x, (size_t)N * LD * sizeof(float), hipMemcpyHostToDevice));
    CHECK_SGL(sgl_spmm_chain_f32(csr, K, d_hop[0], LD, d_hop + 1, ld + 1, D, NULL));
    /* every prefix X_0..X_h, h = 0..K, from ONE pass over the K + 1 hop matrices; the pad columns of the outputs are declared */
    CHECK_SGL(sgl_nafs_prefix_f32(K + 1, (const float *const *)d_hop, ld, ((uint64_t)1 << (K + 1)) - 1, d_out, ld, LD - D, 0, 1.0f, N, D,
                                  NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float *hops = (float *)malloc((size_t)(K + 1) * N * LD * sizeof(float)), *got = (float *)malloc((size_t)N * LD * sizeof(float));
    for (int h = 0; h <= K; ++h)
        CHECK_HIP(hipMemcpy(hops + (size_t)h * N * LD, d_hop[h], (size_t)N * LD * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int h = 0; h <= K; ++h) {
        CHECK_HIP(hipMemcpy(got, d_out[h], (size_t)N * LD * sizeof(float), hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i) {
            const float *x0 = hops + (size_t)i * LD;
            double n0 = 0.0, c[K + 1], mx = -1e300, den = 0.0;
            for (int k = 0; k < D; ++k) n0 += (double)x0[k] * x0[k];
            n0 = sqrt(n0) + 1e-10;
            for (int j = 0; j <= h; ++j) {
                const float *xj = hops + ((size_t)j * N + i) * LD;
                double dot = 0.0, nj = 0.0;
                for (int k = 0; k < D; ++k) {
                    dot += (double)x0[k] * xj[k];
                    nj += (double)xj[k] * xj[k];
                }
                c[j] = dot / (sqrt(nj) + 1e-10) / n0;
                if (c[j] > mx) mx = c[j];
            }
            for (int j = 0; j <= h; ++j) den += exp(c[j] - mx);
            for (int k = 0; k < LD; ++k) {
                double want = 0.0;
                if (k < D)
                    for (int j = 0; j <= h; ++j) want += exp(c[j] - mx) / den * hops[((size_t)j * N + i) * LD + k];
                const double err = fabs((double)got[(size_t)i * LD + k] - want);
                if (err > worst) worst = err;
                if (err > 1e-5) { /* values are O(1): 1e-5 absolute is the contract's relative 1e-5 here; pad columns must be 0 */
                    fprintf(stderr, "prefix %d row %d col %d: %g, want %g\n", h, i, k, got[(size_t)i * LD + k], want);
                    return 7;
                }
            }
        }
    }
    CHECK_SGL(sgl_csr_destroy(csr));
    printf("C-ABI sweeps OK: 2 r x (Laplacian + 3 alpha) from one preparation; %d NAFS prefixes of %d hops from one pass "
           "(max abs error %.2e)\n", K + 1, K, worst);
    return 0;
}
