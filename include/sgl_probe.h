/* sgl_probe.h -- C ABI of libsgl_probe.so: MEASUREMENT AND TEST SUPPORT, not part of the drop-in boundary.
 *
 * Everything a benchmark or a test needs next to the product library (include/sgl_hip.h, libsgl_hip.so) and nothing a user of
 * the propagation path does: memory-system probes (the bare-stream / bare-gather ceilings quoted next to the SpMM), device
 * allocations with a stated physical placement (the address-translation experiments of profiles/r03_papers_tlb.md) and the seeded
 * synthetic-workload generators (SURVEY 8(d) workloads S3 / S4, generated per row block in HBM).  Built from csrc/sgl_probe.hip,
 * sgl_synth.hip, sgl_mem.hip + sgl_probe_core.cpp by sgl_amd/csrc/build.py; loaded by sgl_amd._lib.probe_lib().
 * Conventions as in sgl_hip.h: int return codes (0 = ok), message via sgl_probe_last_error(), stream-ordered, no printing. */
#ifndef SGL_PROBE_H
#define SGL_PROBE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* text of the last error on the calling thread ("" if none) */
const char *sgl_probe_last_error(void);

/* ---- device allocations with a stated physical placement (the tables the SpMM gathers from) ------------------------ */
/* A random gather of 512-byte rows from a table of tens of GB is one address translation per row; how many of them the
 * translation caches hold depends on the size of the physically contiguous, equally aligned ranges behind the table.
 *   SGL_MEM_DEFAULT     hipMalloc
 *   SGL_MEM_CONTIGUOUS  one physically contiguous range (hipExtMallocWithFlags(hipDeviceMallocContiguous))
 *   SGL_MEM_VMM         physical chunks of chunk_bytes (0 = one chunk; rounded to the recommended granularity) mapped into one
 *                       virtual range aligned to the chunk size (hipMemCreate / hipMemAddressReserve / hipMemMap)
 * The pointer is an ordinary device pointer for every kernel and copy; release it with sgl_mem_free (which synchronises the
 * device for SGL_MEM_VMM). */
#define SGL_MEM_DEFAULT 0
#define SGL_MEM_CONTIGUOUS 1
#define SGL_MEM_VMM 2
int sgl_mem_alloc(void **d_out, int64_t bytes, int mode, int64_t chunk_bytes);
int sgl_mem_free(void *d_ptr);

/* ---- memory-system probes (measurement only: the ceilings bench.py / tools/mem_ceilings.py quote next to the SpMM) ---- */
/* sequential read of n_floats floats (16 B per lane); nothing is written (d_sink: one float, untouched in practice) */
int sgl_probe_stream_f32(const float *d_x, int64_t n_floats, float *d_sink, void *stream);
/* random row gather: every wavefront reads table[idx[i], 0:row_floats] (row_floats % 4 == 0, <= 256) for its share of
 * the n_idx row ids with `in_flight` (8 / 16 / 32) independent rows per lane; nothing is written */
int sgl_probe_gather_f32(const float *d_table, int64_t ld, const int32_t *d_idx, int64_t n_idx, int row_floats,
                         int in_flight, float *d_sink, void *stream);

/* ---- seeded synthetic inputs generated in HBM, keyed by (seed, row) (measurement / tests: SURVEY 8(d) workloads S3, S4) ---- */
/* Every rank builds its own row block [row0, row0 + n_rows) of an ogbn-papers100M-shaped directed graph and of the feature
 * matrix; integer hash arithmetic only, mirrored bit-for-bit on the host by sgl_amd/synthetic.py.
 *   degrees : d_deg[i] = d_table8192[...]: 4096 quantiles of the degree law + 4096 more refining the top bucket (the extreme
 *             tail), int32, on device; the index comes from hash(seed, row)
 *   fill    : non-zero j of row i gets a hub-skewed, permuted column id in [0, n_cols) and a value in [0, 1/32)
 *             (d_rowptr: LOCAL row pointers of the block, i.e. the exclusive scan of the degrees)
 *   features: X[i, k] in [-1, 1) for k < d, 0 for d <= k < ld */
int sgl_synth_degrees(uint64_t seed, int64_t row0, int64_t n_rows, const int32_t *d_table8192, int64_t *d_deg, void *stream);
int sgl_synth_fill(uint64_t seed, int64_t row0, int64_t n_rows, int64_t n_cols, const int64_t *d_rowptr, int32_t *d_col,
                   float *d_val, void *stream);
int sgl_synth_features(uint64_t seed, int64_t row0, int64_t n_rows, int64_t d, int64_t ld, float *d_x, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SGL_PROBE_H */
