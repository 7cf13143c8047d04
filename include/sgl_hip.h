/*
 * sgl_hip.h -- C ABI of libsgl_hip.so, the MI355X (gfx950) implementation of SGL's SGAP
 * pre-propagation hot path.  Plain C: pointers and sizes only, no torch / C++ types.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *
 *   FloatCSRMulDenseOMP        sgl/operators/csrc/matmul.h:5, matmul.c:23-40 -- the ONE native symbol the
 *                              reference resolves at run time (sgl/operators/utils.py:14,27,38).  Same name,
 *                              same argument list, same accumulate-into-`answer` semantics, host pointers.
 *   FloatCSRMulDense           sgl/operators/csrc/cudamatmul.c:28-146 (cudamatmul.h:4) -- the reference's dead
 *                              cuSPARSE twin bound by utils.py:43-73.  Same name/arguments, overwrite (beta=0)
 *                              semantics, returns 0/1 like the original.
 *   sgl_csr_create/_destroy,   device-resident successor of the two above: the CSR lives on the GPU across the
 *   sgl_spmm_f32               K hops of GraphOp.propagate (sgl/operators/base_op.py:29-35) instead of being
 *                              re-uploaded per call (cudamatmul.c:57-74,129).
 *   sgl_spmm_axpb_clamp_f32    label_propagation's update  out = alpha * spmm(adj, out) + (1-alpha) * H0 ; clamp
 *                              (sgl/tricks/utils.py:41-58, used by sgl/tricks/correct_and_smooth.py:24-62).
 *   sgl_norm_*                 adj_to_symmetric_norm (sgl/operators/utils.py:76-88) + the Laplacian / PPR
 *                              _construct_adj wrappers (graph_op/laplacian_graph_op.py:12-19,
 *                              graph_op/ppr_graph_op.py:13-21), on device.
 *   sgl_coo_to_csr             the csr_matrix((w,(row,col))) adjacency build of Edge (sgl/data/base_data.py:29) that turns a
 *                              raw adj_matrix.npz{row,col,data} dump (dataset/custom_dataset.py:52-54) into the CSR.
 *   sgl_hop_reduce_f32         Sum/Mean/Max/Min MessageOp._combine (message_op/{sum,mean,max,min}_message_op.py)
 *                              and one_dim_weighted_add (sgl/operators/utils.py:91-102).
 *   sgl_hop_wsum2d_f32(+_bwd)  two_dim_weighted_add (sgl/operators/utils.py:105-116, torch.bmm) and its autograd.
 *   sgl_hop_wsum1d_bwd_f32     autograd of one_dim_weighted_add w.r.t. the weight vector.
 *   sgl_hop_concat_f32         ConcatMessageOp._combine (message_op/concat_message_op.py:11-12, torch.hstack).
 *   sgl_nafs_f32               OverSmoothDistanceWeightedOp._combine (message_op/over_smooth_distance_op.py:11-33).
 *   sgl_gather_rows_f32        the `feat[idx]` row gather of BaseSGAPModel.forward (sgl/models/base_model.py:58,60).
 *
 * Conventions
 *   - every function returns int: 0 = success, non-zero = failure (hipError_t value or SGL_ERR_*); the message is
 *     available from sgl_last_error() (thread-local).  Nothing is printed.  (The reference's CPU symbol returns
 *     void and its GPU twin prints + returns EXIT_FAILURE, cudamatmul.c:7-25; Python never checks either.)
 *   - pointers named d_* are DEVICE pointers owned by the caller (e.g. torch tensor data_ptr()); pointers named
 *     h_* are HOST pointers.  The library owns only what it allocates inside handles.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are stream-ordered and do not
 *     synchronise the device unless documented.
 *   - matrices are row-major float32 with an explicit leading dimension (elements).  Column indices are int32,
 *     row pointers int64, all element offsets are computed in 64 bits (the reference overflows `int` at
 *     N*d >= 2^31, matmul.c:29,33).
 */
#ifndef SGL_HIP_H
#define SGL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGL_OK 0
#define SGL_ERR_INVALID 1001  /* bad argument                                  */
#define SGL_ERR_NO_DEVICE 1002 /* no usable HIP device                          */
#define SGL_ERR_UNSUPPORTED 1003
#define SGL_ERR_ALLOC 1004

#define SGL_MAX_HOPS 64 /* upper bound on the number of hop matrices one aggregator call accepts */

/* ---- library ----------------------------------------------------------------------------------------------- */
int sgl_version(void);                /* 10000*major + 100*minor + patch */
const char *sgl_last_error(void);     /* thread-local, never NULL */
int sgl_device_count(int *count);     /* 0 and *count = 0 when there is no GPU: never aborts */
/* integer tuning knobs for experiments ("spmm_unroll", "spmm_nt", "spmm_group", ...); unknown key -> SGL_ERR_INVALID */
int sgl_set_tuning(const char *key, int64_t value);
int sgl_get_tuning(const char *key, int64_t *value);

/* ---- execution plan (host-only, no GPU needed; exported so it can be unit-tested on CPU) ------------------- */
/* Partition of the rows of a CSR matrix into work items for the SpMM kernel:
 *   - "items": runs of <= 63 consecutive rows holding about `item_nnz` non-zeros, processed by one wavefront each;
 *   - rows longer than `long_row_nnz` are cut into "pieces" of <= long_row_nnz non-zeros whose partial sums are
 *     combined in storage order by a fix-up pass (never with atomics: results are deterministic).
 * long_row_nnz <= 0 disables splitting (every row is summed by one sequential fmaf chain, bit-compatible with
 * the reference's matmul.c:23-40 order).
 * The item LIST is in issue order, not row order: inside every eighth of it (the range one XCD walks) the items holding
 * >= 2 x item_nnz non-zeros come first, longest first, the rest in row order (tuning key "spmm_heavy_first" = 0 keeps plain
 * row order).  Items are whole rows, so the order never changes a result. */
typedef struct sgl_plan sgl_plan_t;
int sgl_plan_build(sgl_plan_t **out, const int64_t *h_rowptr, int64_t n_rows, int32_t item_nnz, int32_t long_row_nnz);
/* counts[0]=n_items, [1]=n_pieces, [2]=n_long_rows, [3]=max rows in an item, [4]=max nnz in an item, [5]=n_rows */
int sgl_plan_counts(const sgl_plan_t *plan, int64_t counts[8]);
/* copy-out (any pointer may be NULL): items as (row_begin,row_end) pairs [2*n_items];
 * pieces as (nnz_begin [n_pieces] int64, nnz_len [n_pieces] int32, row [n_pieces] int32);
 * long rows as (row [n_long] int32, first_piece [n_long+1] int32) */
int sgl_plan_export(const sgl_plan_t *plan, int32_t *h_items, int64_t *h_piece_begin, int32_t *h_piece_len,
                    int32_t *h_piece_row, int32_t *h_long_row, int32_t *h_long_first);
void sgl_plan_destroy(sgl_plan_t *plan);

/* ---- device CSR handle ------------------------------------------------------------------------------------- */
typedef struct sgl_csr sgl_csr_t;

#define SGL_CSR_STRICT_ORDER 0x1u /* no row splitting, one non-zero per step: bit-exact reference order */
#define SGL_CSR_NO_XCD_REMAP 0x2u /* keep the hardware's round-robin block->XCD order                  */

/* A handle carries one split-row workspace: use a handle from ONE stream at a time (create one handle per stream for
 * concurrent launches on the same matrix; they can share the caller's CSR arrays).
 * Wraps caller-owned device arrays (NOT copied; they must outlive the handle) and builds the execution plan
 * (copies the row pointers to the host once: this call synchronises `stream`).
 * item_nnz / long_row_nnz: 0 = library default (items of 512 non-zeros from 1e8 non-zeros per launch, 256 below, fewer for
 * matrices too small to fill the chip; rows above 2048 non-zeros are cut -- above 512 / 128 / 32 for matrices of fewer than
 * 2^22 / 2^20 / 2^18 non-zeros, whose longest row would otherwise be the whole launch; the threshold depends on nnz only, so
 * two handles of one matrix cut the same rows at the same places). */
int sgl_csr_create(sgl_csr_t **out, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_rowptr,
                   const int32_t *d_col, const float *d_val, uint32_t flags, int32_t item_nnz,
                   int32_t long_row_nnz, void *stream);
int sgl_csr_destroy(sgl_csr_t *csr);
/* swap the value array (same sparsity structure, hence same plan): caller-owned, must outlive the handle */
int sgl_csr_set_values(sgl_csr_t *csr, const float *d_val);
/* info[0]=n_rows [1]=n_cols [2]=nnz [3]=n_items [4]=n_pieces [5]=n_long_rows [6]=flags [7]=workspace bytes */
int sgl_csr_info(const sgl_csr_t *csr, int64_t info[8]);

/* Y[0:n_rows, 0:d] = A . X[0:n_cols, 0:d]  (+ Y if accumulate != 0), fp32, fmaf accumulation in CSR order.
 * accumulate = 0 is the cuSPARSE-twin semantics (beta = 0, cudamatmul.c:46); accumulate = 1 is the CPU kernel's
 * (matmul.c:37).  X and Y must not overlap.  ldx, ldy >= d. */
int sgl_spmm_f32(sgl_csr_t *csr, const float *d_x, int64_t ldx, float *d_y, int64_t ldy, int64_t d,
                 int accumulate, void *stream);

/* Y_0 = Y_1 = ... = A . X stored into n_out (1..8) matrices with a common leading dimension.  h_y: HOST array of
 * device pointers; entries beyond the first may point into peer GPUs' memory (IPC / symmetric memory): the kernel
 * then pushes each finished row to every replica over xGMI (row-sharded multi-GPU propagation). */
int sgl_spmm_multi_f32(sgl_csr_t *csr, const float *d_x, int64_t ldx, int n_out, float *const *h_y, int64_t ldy,
                       int64_t d, const uint8_t *d_row_mask, void *stream);
/* d_row_mask (optional, [n_rows] bytes on device): bit q set = destination q+1 receives this row; a peer whose shard
 * never references column i does not need row i of the next feature block, so its store is skipped (NULL = all). */

/* The hop loop of GraphOp.propagate (sgl/operators/base_op.py:29-35) in one call: Y_1 = A.X_0, Y_k = A.Y_{k-1}.
 * h_y / h_ldy: HOST arrays of n_hops device pointers / leading dimensions.  A must be square. */
int sgl_spmm_chain_f32(sgl_csr_t *csr, int n_hops, const float *d_x0, int64_t ldx0, float *const *h_y,
                       const int64_t *h_ldy, int64_t d, void *stream);

/* The same hop loop captured once into a hipGraph and replayed with one launch: for small graphs the k kernels of a
 * propagate() are launch-bound.  Pointers, sizes and the handle are baked in at creation (which runs the chain once,
 * eagerly, and synchronises the device); launching replays onto `stream`.  The handle must outlive the graph. */
typedef struct sgl_graph sgl_graph_t;
int sgl_chain_graph_create(sgl_graph_t **out, sgl_csr_t *csr, int n_hops, const float *d_x0, int64_t ldx0,
                           float *const *h_y, const int64_t *h_ldy, int64_t d);
int sgl_chain_graph_launch(sgl_graph_t *graph, void *stream);
int sgl_chain_graph_destroy(sgl_graph_t *graph);

/* Fused label-propagation step (sgl/tricks/utils.py:55-56, the inner loop of label_propagation and of
 * CorrectAndSmooth):  Y = clamp( alpha * (A . X) + RES, lo, hi )  with the reference's rounding order (rounded product,
 * rounded add, clamp that keeps NaN).  d_res may be NULL (no residual); lo = -INF / hi = +INF disable the clamp. */
int sgl_spmm_axpb_clamp_f32(sgl_csr_t *csr, const float *d_x, int64_t ldx, float *d_y, int64_t ldy, int64_t d,
                            float alpha, const float *d_res, int64_t ldres, float lo, float hi, void *stream);

/* Y = A . X and, in the same pass, the running aggregate over hops:
 *   SGL_ACC_SUM   ACC <- ACC + Y                    SGL_ACC_WSUM  ACC <- ACC + w * Y   (rounded product, then add)
 *   SGL_ACC_MAX   ACC <- max(ACC, Y)                SGL_ACC_MIN   ACC <- min(ACC, Y)   (a NaN in any hop wins, as in torch)
 * followed, for the two sums, by ACC <- ACC / divisor when divisor != 1 (Mean's single true division, on the last hop).
 * Same arithmetic and order as sgl_hop_reduce_f32 over the materialised hops (SUM / MEAN / WSUM / MAX / MIN): the Sum / Mean /
 * SimpleWeighted / Max / Min MessageOps (message_op/sum_message_op.py:10, mean_message_op.py:10,
 * simple_weighted_message_op.py:41-56, max_message_op.py:10, min_message_op.py:10) then cost no pass of their own and no hop
 * matrix has to be kept.  ACC is initialised by the caller (X_s, or w_s * X_s). */
enum { SGL_ACC_SUM = 0, SGL_ACC_WSUM = 1, SGL_ACC_MAX = 2, SGL_ACC_MIN = 3 };
int sgl_spmm_acc_f32(sgl_csr_t *csr, const float *d_x, int64_t ldx, float *d_y, int64_t ldy, int64_t d, float *d_acc,
                     int64_t ldacc, float w, int mode, float divisor, void *stream);

/* ---- device memory -> pageable host memory --------------------------------------------------------------------------------- */
/* Contiguous copy through a team of host threads and pinned staging buffers, huge pages requested for a freshly allocated
 * destination: the reference contract's CPU hop tensors (base_op.py:36 returns torch.FloatTensor(host array)) without a pinned
 * allocation per result.  Synchronous; waits for `stream` (the producer) first. */
int sgl_download(void *h_dst, const void *d_src, int64_t bytes, void *stream);

/* ---- plan-time locality ordering ------------------------------------------------------------------------------------------ */
/* d_order[i] = position of node i in an order that keeps communities contiguous: `rounds` (1..64, typically 8) rounds of
 * semi-synchronous label propagation on the structure (d_rowptr, d_col) of a symmetric adjacency -- every node adopts the most
 * frequent label among (a strided sample of at most 256 of) its neighbours, ties to the smaller label, half of the nodes per
 * round, all in the last -- followed by a stable sort of the nodes by label.  Deterministic.  h_info (optional, host):
 * [0] number of communities, [1] nodes that changed label in the last round.  Runs once per adjacency (tens of ms at
 * ogbn-products size); the caller relabels the problem (sgl_coo_to_csr on the relabelled COO) and permutes features / results.
 * Synchronises the stream. */
int sgl_reorder_community(const int64_t *d_rowptr, const int32_t *d_col, int64_t n, int rounds, int64_t *d_order,
                          int64_t *h_info, void *stream);
/* One round of that label propagation over a ROW BLOCK of a row-sharded matrix (rows [row0, row0 + n_local): local row pointers,
 * GLOBAL column ids): d_labels = the current labels of all n_global nodes (int32), d_out = the new labels of the block's nodes
 * (int32 [n_local]), *d_moved (optional, a zeroed 64-bit device word) += nodes whose label changed.  The caller all-gathers the
 * ranks' slices between rounds and passes last != 0 in the final round; the stable sort by label is the caller's.  Running it over
 * the blocks of a partition reproduces sgl_reorder_community's labels exactly (sgl_amd/dist/redistribute.py). */
int sgl_reorder_lpa_round(const int64_t *d_rowptr, const int32_t *d_col, int64_t n_local, int64_t row0, int64_t n_global,
                          const int32_t *d_labels, int32_t *d_out, int round, int last, uint64_t *d_moved, void *stream);

/* The rows of a CSR in processing order: storage row k of the output = input row d_perm[k]; column ids and the order of every
 * row's entries unchanged (out_rowptr [n+1], out_col / out_val [nnz] caller-allocated).  A handle created on the output and
 * given the same array as its row map computes bit for bit the products of the original matrix -- X and Y keep the caller's node
 * order, only the order in which rows are processed (and with it the reuse of gathered rows in L2 / the Infinity Cache) changes. */
int sgl_csr_permute_rows(const int64_t *d_rowptr, const int32_t *d_col, const float *d_val, int64_t n, const int32_t *d_perm,
                         int64_t *d_out_rowptr, int32_t *d_out_col, float *d_out_val, void *stream);
/* Storage row i of the handle is output row d_rowmap[i] (a permutation of 0..n_rows-1; NULL removes the map; the array must
 * outlive its use).  Applies to sgl_spmm_f32 / _chain / _acc / _axpb_clamp (outputs, residual and running aggregate are addressed
 * through the map); sgl_spmm_multi_f32 refuses a mapped handle. */
int sgl_csr_set_rowmap(sgl_csr_t *csr, const int32_t *d_rowmap, void *stream);

/* ---- multi-GPU exchange (row-sharded layout): the all-gather of the feature block between hops ---------------- */
/* Rank `rank` of `world` owns rows [h_bounds[rank], h_bounds[rank+1]) of the [n, ldx] replica d_x (row-major, whole padded
 * rows travel) and has already written them; on completion (stream-ordered) every rank's rows are in place.  One grouped
 * batch of ncclSend / ncclRecv to / from all peers on the CALLER's communicator (`nccl_comm` = ncclComm_t): direct, unequal
 * blocks, all xGMI links at once.  RCCL is resolved at run time from the host process (else librccl.so is loaded);
 * SGL_ERR_UNSUPPORTED if there is none.  world == 1 is a no-op.  sgl_exchange_backend() says which RCCL was found. */
int sgl_allgather_rows(void *nccl_comm, int rank, int world, const int64_t *h_bounds, float *d_x, int64_t ldx, void *stream);
/* Need-aware form (the plan: sgl_amd/dist/halo.py): a rank receives only the rows its block gathers, packed.  d_send holds the
 * rows of this rank that the peers gather, peer q's share at rows [h_send_off[q], h_send_off[q+1]) (packed by
 * sgl_gather_rows_f32 with the rank's send list); the rows of peer q that this rank gathers land at rows
 * [h_recv_off[q], h_recv_off[q+1]) of d_recv (the ghost range of its compact table).  Offsets are host arrays of world + 1
 * rows, a rank's own entry is empty, ld = floats per row of both buffers.  Same transport and error behaviour as above. */
int sgl_exchange_rows(void *nccl_comm, int rank, int world, const float *d_send, const int64_t *h_send_off, float *d_recv,
                      const int64_t *h_recv_off, int64_t ld, void *stream);
const char *sgl_exchange_backend(void);
/* Loop-back check of the RCCL binding on the caller's communicator: n floats travel from d_src to d_dst (disjoint device buffers)
 * as n_ops grouped ncclSend / ncclRecv pairs whose peer is `rank` itself -- posted by the same internal routine, function table
 * and data-type constant as the two exchanges above.  Works on a one-rank communicator (one GPU) and on every rank of a real job
 * before its first hop; the reference's only communicator set-up is dist.init_process_group('nccl'),
 * tasks/node_classification_dist.py:61.  Stream-ordered; SGL_ERR_UNSUPPORTED without RCCL. */
int sgl_exchange_selftest(void *nccl_comm, int rank, const float *d_src, float *d_dst, int64_t n, int n_ops, void *stream);

/* ---- reference-signature host shims (H2D -> kernel -> D2H; synchronous) ------------------------------------ */
/* matmul.h:5 -- accumulates into `answer` (caller pre-zeroes it, utils.py:31).  Errors are recorded in
 * sgl_last_error() (the reference symbol is void). */
void FloatCSRMulDenseOMP(float answer[], float data[], int indices[], int indptr[], float mat[], int mat_row,
                         int mat_col);
/* cudamatmul.c:28 -- overwrites `answer`; returns 0 on success, 1 on failure (EXIT_SUCCESS / EXIT_FAILURE). */
int FloatCSRMulDense(float answer[], int data_nnz, float data[], int indices[], int indptr[], float mat[],
                     int mat_row, int mat_col);

/* The shims keep the uploaded adjacency, its plan and all device / pinned buffers between calls and re-use them when the
 * next call passes the bit-identical CSR (the reference calls the symbol K times per propagate() with the same matrix):
 * counters of calls served from / not served from that cache. */
int sgl_shim_cache_stats(int64_t *hits, int64_t *misses);

/* ---- normalisation on device (adj_to_symmetric_norm, operators/utils.py:76-88) ----------------------------- */
/* Input: canonical CSR of A (sorted columns, no duplicates), n x n, values fp32, on device.
 * Output: canonical CSR of  A_hat = D^{r-1} (A+I)^T D^{-r}   [then (1-alpha) A_hat + alpha I when use_alpha != 0]
 * with nnz_out = nnz(A  U  diag).  Arithmetic in fp64 exactly in the reference's order, rounded to fp32 at the
 * end (where utils.py:32 rounds).  Two-step protocol so the caller owns the output buffers:
 *   1. sgl_norm_prepare(...)  -> *nnz_out   (counts missing diagonal entries; synchronises `stream`)
 *   2. sgl_norm_execute(...)  fills d_out_rowptr [n+1], d_out_col [nnz_out], d_out_val [nnz_out]
 *      (and d_out_val64 [nnz_out] when not NULL: the fp64 values before rounding, for parity tests). */
int sgl_norm_prepare(int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col, int64_t *nnz_out,
                     void *stream);
int sgl_norm_execute(int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col, const float *d_val,
                     double r, int use_alpha, double alpha, int64_t nnz_out, int64_t *d_out_rowptr,
                     int32_t *d_out_col, float *d_out_val, double *d_out_val64, void *stream);

/* The two degree powers can be supplied by the caller: sgl_norm_degrees returns the fp64 weighted degrees of A + I
 * (rows [row0, row0 + n) of a row block; row0 = 0 for a whole matrix), the host evaluates deg^(r-1), deg^(-r) with the
 * libm the reference's numpy calls (utils.py:79-84, inf -> 0) and sgl_norm_execute_lr uses them: A_hat then rounds to
 * fp32 bit-identically to the reference (device pow() differs from glibc's in the last bit of a few entries). */
int sgl_norm_degrees(int64_t n, int64_t row0, const int64_t *d_rowptr, const int32_t *d_col, const float *d_val,
                     double *d_deg, void *stream);
int sgl_norm_execute_lr(int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col, const float *d_val,
                        const double *d_left, const double *d_right, int use_alpha, double alpha, int64_t nnz_out,
                        int64_t *d_out_rowptr, int32_t *d_out_col, float *d_out_val, double *d_out_val64, void *stream);

/* Row-block normalisation for the row-sharded multi-GPU layout (every GPU holds only ITS rows of A_hat).
 * Input: rows [row0, row0 + n) of T = A^T (for a symmetric A: of A itself) as CSR with GLOBAL, sorted column ids.
 *   1. sgl_norm_block_prepare -> *nnz_out = nnz + number of local rows without a diagonal entry (synchronises)
 *   2. sgl_norm_block_build   -> T' = T + I for the block (CSR, fp64 values) and its fp64 row sums
 *   3. deg = rowsum(A + I): for a symmetric A the blocks' row sums (all-gather); otherwise the column sums of T'
 *      (sgl_norm_block_colsum accumulates a block's share into a zeroed [n_cols] vector; all-reduce over the ranks)
 *   4. sgl_norm_block_scale   -> A_hat[j, i] = (T'[j, i] * L[j]) * R[i]  [then (1-alpha) A_hat + alpha I], rounded to fp32;
 *      d_left_local = deg^(r-1) of the block's rows, d_right_global = deg^(-r) of ALL n_cols nodes.
 * Same arithmetic and rounding as sgl_norm_execute; none of it needs the other ranks' rows. */
int sgl_norm_block_prepare(int64_t n, int64_t row0, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col,
                           int64_t *nnz_out, void *stream);
int sgl_norm_block_build(int64_t n, int64_t row0, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col,
                         const float *d_val, int64_t nnz_out, int64_t *d_out_rowptr, int32_t *d_out_col,
                         double *d_out_val64, double *d_rowsum, void *stream);
/* sgl_norm_block_build for a WHOLE matrix (row0 = 0) that also answers "is A symmetric, values included?": *d_sym_hash (one
 * zero-initialised 64-bit word on the device) is 0 afterwards iff it is (every off-diagonal (i, j, v) adds +-h(min, max, v) in
 * wrapping arithmetic; a false "symmetric" has probability 2^-64).  A symmetric A needs no transposition: sgl_norm_block_scale
 * on the result IS adj_to_symmetric_norm -- no sort, no permutation, bit-identical to sgl_norm_execute. */
int sgl_norm_build_symcheck(int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_col, const float *d_val,
                            int64_t nnz_out, int64_t *d_out_rowptr, int32_t *d_out_col, double *d_out_val64, double *d_rowsum,
                            uint64_t *d_sym_hash, void *stream);
int sgl_norm_block_colsum(int64_t n_cols, int64_t nnz, const int32_t *d_col, const double *d_val64, double *d_colsum,
                          void *stream);
int sgl_norm_block_scale(int64_t n, int64_t row0, const int64_t *d_rowptr, const int32_t *d_col, const double *d_val64,
                         const double *d_left_local, const double *d_right_global, int use_alpha, double alpha,
                         float *d_out_val, double *d_out_val64, void *stream);
/* (1 - alpha) A_hat + alpha I (ppr_graph_op.py:20) from the fp64 A_hat of the same block and r (d_out_val64 of a
 * sgl_norm_block_scale call with use_alpha = 0): a pure stream, bit-identical to sgl_norm_block_scale(use_alpha = 1).  An alpha
 * sweep at one r pays the degree factors and the gather of d_right_global once. */
int sgl_norm_block_mix(int64_t n, int64_t row0, const int64_t *d_rowptr, const int32_t *d_col, const double *d_hat64,
                       double alpha, float *d_out_val, double *d_out_val64, void *stream);
/* The same mix without any row lookup: d_diag[i] = the position of row i's diagonal entry in the block's CSR (sgl_norm_block_diag_positions,
 * once per block; -1 if a row has none), then a flat stream (1 - alpha) A_hat over the nnz values and alpha added at the n diagonal
 * positions.  Same roundings as sgl_norm_block_mix: bit-identical.  d_hat64 16-byte aligned. */
int sgl_norm_block_diag_positions(int64_t n, int64_t row0, const int64_t *d_rowptr, const int32_t *d_col, int64_t *d_diag, void *stream);
int sgl_norm_block_mix_at(int64_t nnz, int64_t n, const double *d_hat64, const int64_t *d_diag, double alpha, float *d_out_val,
                          double *d_out_val64, void *stream);
/* d_left = deg^(r-1), d_right = deg^(-r), inf -> 0 (utils.py:79-84) with the device's pow(): within 1 ulp(fp64) of the host
 * libm route described above -- for callers that want 1e-5 parity, not bit-identity, and no host round trip. */
int sgl_norm_degree_powers(int64_t n, const double *d_deg, double r, double *d_left, double *d_right, void *stream);

/* ---- ingest: COO edge list -> canonical CSR on device (sgl/data/base_data.py:29, dataset/custom_dataset.py:52-54) ---- */
/* d_row / d_col: int64 [nnz] (the reference keeps them as torch.LongTensor), d_val float32 [nnz].  Duplicate (row,col)
 * pairs are summed in input order (fp32), columns come out sorted inside each row -- what scipy's
 * csr_matrix((data,(row,col))) produces.  Outputs: d_out_rowptr [n_rows+1]; d_out_col / d_out_val sized for nnz
 * entries, the first *h_nnz_out of them are valid.  Synchronises `stream`.  Fails on an out-of-range index. */
int sgl_coo_to_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_row, const int64_t *d_col,
                   const float *d_val, int64_t *d_out_rowptr, int32_t *d_out_col, float *d_out_val, int64_t *h_nnz_out,
                   void *stream);

/* ---- per-hop aggregators (MessageOp._combine) -------------------------------------------------------------- */
#define SGL_REDUCE_SUM 0  /* ((X0 + X1) + X2) + ...            sum_message_op.py:10   */
#define SGL_REDUCE_MEAN 1 /* sum, then one true division by H  mean_message_op.py:10  */
#define SGL_REDUCE_MAX 2  /* NaN-propagating                   max_message_op.py:12   */
#define SGL_REDUCE_MIN 3  /*                                   min_message_op.py:12   */
#define SGL_REDUCE_WSUM 4 /* sum_h w[h] * X_h, d_w = H floats  operators/utils.py:91-102 */

/* h_x: HOST array of n_hops DEVICE pointers; h_ldx: host array of leading dimensions (NULL = all d). */
int sgl_hop_reduce_f32(int op, int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_w,
                       float *d_out, int64_t ldo, int64_t n, int64_t d, void *stream);
/* out[n, k] = sum_h W[n, h] * X_h[n, k]   (W row-major [n, n_hops], leading dimension ldw) */
int sgl_hop_wsum2d_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_w, int64_t ldw,
                       float *d_out, int64_t ldo, int64_t n, int64_t d, void *stream);
/* backward of the above: dW[n,h] = <dOut[n,:], X_h[n,:]> (d_dw may be NULL);
 * dX_h[n,k] = W[n,h] * dOut[n,k] for every non-NULL h_dx[h] (h_dx may be NULL) */
int sgl_hop_wsum2d_bwd_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_w, int64_t ldw,
                           const float *d_dout, int64_t lddo, float *d_dw, int64_t lddw, float *const *h_dx,
                           const int64_t *h_lddx, int64_t n, int64_t d, void *stream);
/* out[n, h] = <X_h[n, :], v> for every hop in one pass: the gate scores Linear(d -> 1)(X_h) of
 * LearnableWeightedMessageOp (message_op/learnable_weighted_messahe_op.py:69-86).  d_vec: d floats on device, readable
 * up to round_up(d, 4) floats (zero padded) when 16-byte aligned. */
int sgl_hop_rowdot_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_vec, float *d_out,
                       int64_t ldo, int64_t n, int64_t d, void *stream);
/* backward of SGL_REDUCE_WSUM w.r.t. the weights: d_dw[h] = sum_{n,k} dOut[n,k] * X_h[n,k]   (d_dw: H floats,
 * overwritten; deterministic two-level reduction; d_scratch: >= sgl_hop_wsum1d_bwd_scratch(n_hops) floats) */
int64_t sgl_hop_wsum1d_bwd_scratch(int n_hops);
int sgl_hop_wsum1d_bwd_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_dout,
                           int64_t lddo, float *d_dw, float *d_scratch, int64_t n, int64_t d, void *stream);
/* out_k = sum_j W[k, j] X_j (k < n_out <= 64, j < n_in <= 16): a small dense matrix applied across the hop dimension in ONE pass --
 * every input element is read once for all outputs.  d_w: [n_out, ldw] floats on the device; zero weights are skipped (hops that do
 * not enter an output cannot contaminate it with NaN / Inf); each sum is one fma chain in j order.  Outputs must not alias inputs.
 * Use: the hop matrices of PprGraphOp(K, r, alpha) are polynomials in the Laplacian's of the same r,
 *   ((1 - alpha) A_hat + alpha I)^k X = sum_j C(k, j) (1 - alpha)^j alpha^(k - j) A_hat^j X   (ppr_graph_op.py:20, base_op.py:29-35),
 * so every alpha of a sweep follows from ONE propagation chain (sgl_amd.operators.graph_op.ppr_hops_from_laplacian). */
int sgl_hop_lincomb_f32(int n_in, const float *const *h_x, const int64_t *h_ldx, int n_out, float *const *h_out, const int64_t *h_ldo,
                        const float *d_w, int64_t ldw, int64_t n, int64_t d, void *stream);
/* Backward of MaxMessageOp / MinMessageOp (torch.stack(hops).max(0)[0], max_message_op.py:12): dX_h = dOut where hop h is the one
 * torch selects for the element -- the first NaN if there is one, otherwise the first hop attaining the extremum -- and 0 elsewhere.
 * op = SGL_REDUCE_MAX or SGL_REDUCE_MIN; h_dx[h] may be NULL for hops that need no gradient. */
int sgl_hop_select_bwd_f32(int op, int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_gout, int64_t ldg,
                           float *const *h_dx, const int64_t *h_lddx, int64_t n, int64_t d, void *stream);
/* Row outputs and padding: sgl_hop_concat_padded_f32 / sgl_nafs_padded_f32 / sgl_hop_gate_padded_f32 take `pad_cols` = the number of
 * columns after the row (after column n_hops * d resp. d) that are the row's OWN padding inside its pitch ldo.  They are WRITTEN AS
 * ZEROS, so that every 128-byte line of a row is written whole: a partly written line costs a read-modify-write in the ECC-protected
 * HBM (output stream 2.7 instead of 5.8 TB/s at d = 147 on a 160-float pitch).  width + pad_cols <= ldo, both multiples of 4.  The
 * kernels never guess: the un-suffixed entry points are pad_cols = 0 and touch nothing beyond the row (the vector that straddles
 * column d is then written element by element). */
/* out[:, h*d:(h+1)*d] = X_h */
int sgl_hop_concat_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, float *d_out, int64_t ldo,
                       int64_t n, int64_t d, void *stream);
int sgl_hop_concat_padded_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, float *d_out, int64_t ldo,
                              int64_t pad_cols, int64_t n, int64_t d, void *stream);
/* NAFS: W[n,h] = softmax_h( <X_0[n],X_h[n]> / (|X_h[n]|+1e-10) / (|X_0[n]|+1e-10) ), out = sum_h W[n,h] X_h[n]
 * d_w_out (optional, [n, n_hops] leading dimension ldw) receives W. */
int sgl_nafs_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, float *d_out, int64_t ldo,
                 float *d_w_out, int64_t ldw, int64_t n, int64_t d, void *stream);
int sgl_nafs_padded_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, float *d_out, int64_t ldo, int64_t pad_cols,
                        float *d_w_out, int64_t ldw, int64_t n, int64_t d, void *stream);
/* NAFS hop sweep (NodeClusteringNAFS._execute: _k_hop_cluster(hop) for EVERY hop of a range, tasks/node_clustering.py:139,176-178,
 * 205-241; twin tasks/link_prediction.py:233-284): the over-smoothing-distance aggregate of every PREFIX X_0..X_h of the hop list in
 * one pass -- running numerator sum e^{c_j} X_j and denominator sum e^{c_j}, c_j = cos(X_0, X_j) as in sgl_nafs_f32; each hop element
 * is read once.  Bit h of emit_mask set = emit prefix h (h < n_hops) into h_out[rank of the bit among the set bits] (pitch h_ldo[.],
 * 16-byte aligned, multiple of 4 floats; pad_cols as in the *_padded entry points).  combine says what happens to the values already
 * there (the multi-r ensemble, node_clustering.py:242-249): 0 = overwrite, 1 = add, 2 = add then divide by `divisor` (the last r of
 * 'mean'), 3 = element-wise max.  d <= 512.  Agrees with the per-prefix sgl_nafs_f32 to fp32 rounding (softmax without the max
 * subtraction: |c| <= 1), inside the 1e-5 contract. */
int sgl_nafs_prefix_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, uint64_t emit_mask, float *const *h_out,
                        const int64_t *h_ldo, int64_t pad_cols, int combine, float divisor, int64_t n, int64_t d, void *stream);

/* Learnable gate in one pass (LearnableWeightedMessageOp 'gate', message_op/learnable_weighted_messahe_op.py:67-71 followed by
 * two_dim_weighted_add, operators/utils.py:105-116):  G[n,h] = sigmoid(<X_h[n], vec> + bias),  W[n,:] = softmax_h(G[n,:]),
 * out[n] = sum_h W[n,h] X_h[n].  Every hop element is read once.  d_vec: round_up(d, 4) floats, 16-byte aligned, zero beyond d.
 * sgl_hop_gate_f32: `bias` is the scalar it says (a NaN bias gives NaN scores, and nothing beyond d_vec's round_up(d, 4) floats is
 * read).  sgl_hop_gate_padded_f32 ONLY: bias = NaN means "the bias is on the device": it is read from d_vec[round_up(d, 4)] by
 * the kernel -- d_vec then holds round_up(d, 4) + 1 floats -- (no host synchronisation, the launch can be captured in a hipGraph
 * and replayed while the parameter changes).
 * d_w_out / d_g_out (optional, [n, n_hops]) receive W and G (the backward needs both).  Register-resident rows: n_hops <= 16,
 * d <= 512, 16-byte aligned rows -- otherwise SGL_ERR_UNSUPPORTED (callers then use sgl_hop_rowdot_f32 + sgl_hop_wsum2d_f32). */
int sgl_hop_gate_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_vec, float bias, float *d_out,
                     int64_t ldo, float *d_w_out, int64_t ldw, float *d_g_out, int64_t ldg, int64_t n, int64_t d, void *stream);
int sgl_hop_gate_padded_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_vec, float bias, float *d_out,
                            int64_t ldo, int64_t pad_cols, float *d_w_out, int64_t ldw, float *d_g_out, int64_t ldg, int64_t n,
                            int64_t d, void *stream);
/* Recursive gate in one pass (IterateLearnableWeightedMessageOp 'recursive', message_op/iterate_learnable_weighted_message_op.py:28-51;
 * GAMLP-R, models/homo/gamlp_recursive.py:7-13).  The reference walks the hops: step i scores sigmoid(Linear([X_i || acc])), appends
 * the score to the weights of the steps before, soft-maxes all of them (the earlier ones again) and rebuilds acc = sum_j W[:, j] X_j.
 * acc is a per-row weighted sum of the hops, so with A[n,h] = <X_h[n], w_x> and C[n,h] = <X_h[n], w_acc>
 *     Linear([X_i || acc_{i-1}])[n] = A[n,i] + sum_{j<i} W_{i-1}[n,j] C[n,j] + bias
 * and the recursion runs on the 2 H scalars of a row while its hop rows sit in registers; out[n] = sum_h W[n,h] X_h[n] with the final
 * weights.  Every hop element is read once (step by step: H (H + 3) / 2 reads of a hop matrix, H accumulator writes).
 * d_vec: [w_x | w_acc], each zero-padded to round_up(d, 4) floats, 16-byte aligned; bias = NaN: read from d_vec[2 * round_up(d, 4)].
 * pad_cols: as in the *_padded_f32 entry points above.  d_w_out / d_a_out / d_c_out (optional, [n, n_hops]) receive W, A and C (all the
 * backward needs besides the hops).  Register-resident rows: n_hops <= 16, d <= 512, 16-byte aligned rows -- otherwise
 * SGL_ERR_UNSUPPORTED (callers then use two sgl_hop_rowdot_f32 passes, the [n, H] recursion on the host side, sgl_hop_wsum2d_f32). */
int sgl_hop_recursive_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_vec, float bias, float *d_out,
                          int64_t ldo, int64_t pad_cols, float *d_w_out, int64_t ldw, float *d_a_out, int64_t lda, float *d_c_out,
                          int64_t ldc, int64_t n, int64_t d, void *stream);
/* Backward of the [n, n_hops] recursion of sgl_hop_recursive_f32: from A, C, the bias (NaN: *d_bias on the device) and G = dL/dW to
 * dA, dC [n, n_hops] and dB [n] (per row; sum it for the Linear's bias gradient).  One thread per row; any output may be NULL.
 * The Linear's weight gradient follows as [sum_h colsum(X_h, dA[:, h]) | sum_h colsum(X_h, dC[:, h])] (sgl_hop_colsum_f32). */
int sgl_hop_recursive_bwd_f32(int n_hops, const float *d_a, int64_t lda, const float *d_c, int64_t ldc, float bias, const float *d_bias,
                              const float *d_gw, int64_t ldg, float *d_da, int64_t ldda, float *d_dc, int64_t lddc, float *d_db,
                              int64_t n, void *stream);
/* Weight gradient of the row-dots (the backward of sgl_hop_rowdot_f32 / sgl_hop_rowdot2_f32 / sgl_hop_gate_f32 w.r.t. the Linear's
 * weight; torch: one transposed GEMV `X_h.t() @ g[:, h]` per hop):  out[h, :] = sum_n W[n * ldw + h * sw] * X_h[n, :]  for every hop in
 * ONE pass (sw = 1: a weight per row and hop, sw = 0: one per row shared by all hops).  out: [n_hops, ldo] on device, d_scratch:
 * sgl_hop_colsum_scratch(n_hops, n, d) floats.  Deterministic two-level reduction, no atomics.  n_hops <= 16, d <= 1024, 16-byte
 * aligned rows, else SGL_ERR_UNSUPPORTED. */
int64_t sgl_hop_colsum_scratch(int n_hops, int64_t n, int64_t d);
int sgl_hop_colsum_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_w, int64_t ldw, int sw, float *d_out,
                       int64_t ldo, float *d_scratch, int64_t n, int64_t d, void *stream);
/* Scores of the 'ori_ref' / 'jk' gates (learnable_weighted_messahe_op.py:73-86) in one pass over the hop list:
 *   P[n, h - h0] = <X_h[n], vec>  for h in [h0, h1);   A[n] = sum over the hops j with bit j of u_mask set of <X_j[n], U[j, :]>
 * (the reference concatenates [ref || x_h] with ref = feat_list[0] or hstack(feat_list) and applies one Linear: the ref part is
 * the same for every adopted hop of a node).  U: [n_hops, ldu] on device, vec as above.  Same limits as sgl_hop_gate_f32. */
int sgl_hop_rowdot2_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_u, int64_t ldu, uint64_t u_mask,
                        const float *d_vec, int h0, int h1, float *d_p, int64_t ldp, float *d_a, int64_t n, int64_t d, void *stream);
/* out[i, :] = X[idx[i], :]   (idx: int64 on device; negative indices are NOT wrapped) */
int sgl_gather_rows_f32(const float *d_x, int64_t ldx, int64_t n_rows, const int64_t *d_idx, int64_t n_idx,
                        float *d_out, int64_t ldo, int64_t d, void *stream);
/* the same with the destination rows' own padding declared (as in the other *_padded entry points): columns [d, d + pad_cols) of every
 * output row are written as ZEROS in whole 16-byte vectors, so that every line of a row is written whole; nothing beyond column d of
 * the source is copied (the source may be a column view of a wider matrix whose tail is somebody's data).  Needs 16-byte aligned
 * rows, pitches and d + pad_cols multiples of 4. */
int sgl_gather_rows_padded_f32(const float *d_x, int64_t ldx, int64_t n_rows, const int64_t *d_idx, int64_t n_idx, float *d_out,
                               int64_t ldo, int64_t d, int64_t pad_cols, void *stream);
/* The same rows of EVERY hop matrix in ONE launch: out_h[i, :] = X_h[idx[i], :] for h < n_hops (host arrays of n_hops device pointers
 * and pitches; every X_h has n_rows rows, every out_h n_idx rows; pad_cols as above).  The training feed of the learnable aggregators,
 * `[feat[idx].to(device) for feat in self._processed_feat_list]` (sgl/models/base_model.py:58-60): one grid over (hop, block of rows),
 * no launch gaps between the hops.  16-byte aligned rows and pitches that are multiples of 4 floats on both sides, else SGL_ERR_UNSUPPORTED. */
int sgl_gather_hops_padded_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, int64_t n_rows, const int64_t *d_idx,
                               int64_t n_idx, float *const *h_out, const int64_t *h_ldo, int64_t d, int64_t pad_cols, void *stream);
/* out[dst[i], :] = X[src[i], :] for i < n_idx   (src, dst: int64 on device, dst entries distinct and < n_out_rows; a bad index
 * traps the kernel).  The pack step of the need-aware exchange (sgl_exchange_rows): the (own row, send-buffer row) pairs sorted
 * by own row, so that a row several peers gather is read from HBM once. */
int sgl_scatter_rows_f32(const float *d_x, int64_t ldx, int64_t n_rows, const int64_t *d_src, const int64_t *d_dst,
                         int64_t n_idx, float *d_out, int64_t ldo, int64_t n_out_rows, int64_t d, void *stream);

/* Per-column content signature of a DEVICE matrix: d_sig[c] = wrapping 64-bit sum over the rows r of mix(bits(X[r, c]), r), for
 * c < round_up(d, 4) (uint64 on device; rows 16-byte aligned, pitch a multiple of 4 floats covering round_up(d, 4)).  One streaming
 * read.  GraphOp.propagate (sgl/operators/base_op.py:29-35) is separable by columns: between two calls over one adjacency only the
 * columns whose signature moved need new hops -- the label-reuse loop (sgl/tasks/node_classification_with_label_use.py:88-104)
 * rewrites the last C of its d + C columns between preprocess() calls. */
int sgl_col_signature_f32(const float *d_x, int64_t ldx, int64_t n, int64_t d, uint64_t *d_sig, void *stream);

/* order-sensitive 64-bit hash of a HOST buffer, multi-threaded (what the reference-signature shims key their cached adjacency
 * on; the operator layer fingerprints scipy index / value arrays with it).  Host-only: works without a GPU. */
int sgl_content_hash(const void *h_ptr, int64_t bytes, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* SGL_HIP_H */
