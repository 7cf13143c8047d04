"""Process-wide defaults of the MI355X propagation path (overridable per operator through ctor kwargs).

device        where propagation runs and where the hop matrices stay ("cuda" = current device)
host_output   True  -> GraphOp.propagate returns CPU FloatTensors exactly like the reference
              False -> the K+1 hop matrices stay resident in HBM (`.to(device)` is then a no-op and the
                       row gathers of BaseSGAPModel.forward run on the GPU)
strict_types  True  -> reject torch.Tensor features / non-float32 input with the reference's exceptions
              False -> superset: torch tensors and any float dtype are accepted
strict_order  True  -> SpMM walks every row as ONE sequential fmaf chain (bit-exact with the reference's
                       matmul.c:23-40 order); False -> fastest lane layout (same result within 1e-5)
cache_adj     reuse the normalised device adjacency across propagate() calls on the same scipy matrix
slab_hops     GraphOp.propagate writes hop k into column slice k of ONE [N, (K+1) d] buffer (when d % 4 == 0), so that
              ConcatMessageOp over consecutive hops is a zero-copy view of it instead of a copy of every hop
fuse_aggregate  BaseSGAPModel.preprocess folds last / sum / mean / max / min / simple_weighted aggregation into the SpMM epilogue
              (GraphOp.propagate_reduce): no pass over the hop matrices, only two hop buffers alive; the K+1 hop list
              (`_processed_feat_list`) is then not kept (the reference's own consumers never read it for these ops).
              `last` costs nothing and is always folded.  sum / mean / weighted cost ~2 % more time than the separate
              pass (the running aggregate is read and written once per hop) and save K-1 hop buffers:
              "auto" (default) folds them only when the K+1 hop matrices would take more than a quarter of the free
              device memory; True / False force it
cache_prepared  keep the (r, alpha)-independent part of a normalisation (A + I in fp64, degrees: 12 bytes per non-zero) with the
              device adjacency / row block it was computed from, so that a sweep over r / alpha pays one pass per candidate.
              cache_prepared_gb (default 8) bounds what the process-wide cache of whole-matrix preparations may keep resident: a
              matrix whose preparation would exceed it is prepared transiently (freed before the hop matrices are allocated, as if
              the cache were off -- a papers100M-sized matrix on one GPU would otherwise pin ~40 GB), least recently used entries
              are evicted to stay under it, and entries die with their matrix (weak-reference callback).
              keep_sweep_values (default False): additionally keep the fp64 Laplacian of the last PPR request (8 bytes per
              non-zero) so that the next alpha at the same r is a pure stream (0.36 instead of 2.0 ms at the products shape)
reorder       None -> the rows of A_hat are processed in the caller's node order; "community" -> a plan-time locality ordering
              (sgl_amd/reorder.py -> sgl_reorder_community: label propagation on the device, ~70 ms at products size, cached
              with the adjacency) decides the order in which the rows are STORED and PROCESSED (sgl_csr_permute_rows +
              sgl_csr_set_rowmap); node ids, X, Y and every row's summation order are untouched, results are bit-identical.
              Pays on graphs that HAVE communities their ids do not show (-32 % per hop on the shuffled community graph of
              tools/bench_reorder.py), neutral on the random benchmark graph; "auto" -> run the ordering and keep it only
              when it makes the graph measurably more local than its own ids do (sgl_amd.reorder.plan_rowmap)
share_hops    True -> GraphOp.propagate consults a process-wide store of device-resident hop lists keyed on the CONTENT of adjacency +
              features + operator parameters (sgl_amd/hopcache.py: SharedHops): a fresh operator per search trial re-uses the chain an
              earlier one produced, and a PprGraphOp is served from the LaplacianGraphOp chain of the same r by a mixing pass (not under
              strict_order).  The lists are shared: treat hop matrices as read-only.  share_hops_gb bounds the store (default 64)
trace         True (SGL_AMD_TRACE=1) -> every GraphOp.propagate() records the wall time of its phases (adjacency: fingerprint /
              upload / normalise / plan; features: upload; hops: the k SpMMs; output: download or cache), synchronising at the phase
              ends, in `op.last_trace` and prints them on stderr -- the reference times its whole preprocess() with time.time() and a
              print (tasks/node_classification.py:34-38)
delta_propagate  True (default) -> GraphOp.propagate remembers a 64-bit content signature per feature COLUMN (sgl_col_signature_f32: one
              streaming read) and, weakly, the hop matrices it returned.  The product is separable by columns, so when the next call
              comes with the same adjacency, the same shape, the previous hop matrices still alive and untouched, and only some
              columns of X changed -- the label-reuse loop rewrites the last C of d + C columns between its preprocess() calls
              (sgl/tasks/node_classification_with_label_use.py:88-104) -- only the 4-aligned column range that covers the changed
              columns is propagated again; the other columns of the NEW hop matrices are copied from the old ones.  Fresh tensors
              are returned either way.  Bit-identical to a full propagation under strict_order; otherwise the narrower slice may
              run in another lane layout (same result within 1e-5).  Only for feature matrices of at least delta_propagate_min_mb
              (default 64) and when the range is at most delta_propagate_max_fraction (default 0.7) of the columns; `op.delta_info`
              says what the last call did
hop_cache_dir None -> every propagate() computes; a directory -> the hop matrices of propagate() are kept on disk under a key of
              the CONTENT of adjacency + features + operator parameters and loaded on a hit (sgl_amd/hopcache.py; the reference
              recomputes them in every run of every task)
"""
import os


def _env_bool(name, default):
    v = os.environ.get(name)
    if v is None:
        return default
    return v.strip().lower() in ("1", "true", "yes", "on")


device = os.environ.get("SGL_AMD_DEVICE", "cuda")
host_output = _env_bool("SGL_AMD_HOST_OUTPUT", False)
strict_types = _env_bool("SGL_AMD_STRICT_TYPES", False)
strict_order = _env_bool("SGL_AMD_STRICT_ORDER", False)
cache_adj = _env_bool("SGL_AMD_CACHE_ADJ", True)
cache_prepared = _env_bool("SGL_AMD_CACHE_PREPARED", True)
cache_prepared_gb = float(os.environ.get("SGL_AMD_CACHE_PREPARED_GB", "8"))
keep_sweep_values = _env_bool("SGL_AMD_KEEP_SWEEP_VALUES", False)
delta_propagate = _env_bool("SGL_AMD_DELTA_PROPAGATE", True)
delta_propagate_min_mb = float(os.environ.get("SGL_AMD_DELTA_PROPAGATE_MIN_MB", "64"))
delta_propagate_max_fraction = float(os.environ.get("SGL_AMD_DELTA_PROPAGATE_MAX_FRACTION", "0.7"))
share_hops = _env_bool("SGL_AMD_SHARE_HOPS", False)
share_hops_gb = float(os.environ.get("SGL_AMD_SHARE_HOPS_GB", "64"))
_fa = os.environ.get("SGL_AMD_FUSE_AGGREGATE", "auto").strip().lower()
fuse_aggregate = "auto" if _fa == "auto" else _fa in ("1", "true", "yes", "on")
slab_hops = _env_bool("SGL_AMD_SLAB_HOPS", False)
reorder = os.environ.get("SGL_AMD_REORDER") or None
hop_cache_dir = os.environ.get("SGL_AMD_HOP_CACHE") or None
trace = _env_bool("SGL_AMD_TRACE", False)
