// CSR x dense fp32 SpMM for gfx950 (MI355X) -- the k-step A_hat . X kernel of GraphOp.propagate
// (reference: sgl/operators/base_op.py:29-35 -> csrc/matmul.c:23-40; successor of the dead cuSPARSE twin
// csrc/cudamatmul.c:104-119).
//
// Design:
//   * the path is sparse x dense, ~0.5 flop/byte: HBM/cache bound, no MFMA;
//   * one 64-lane wavefront owns one work item = a run of <= 63 whole rows holding ~item_nnz non-zeros
//     (host-built plan, sgl_core.cpp); very long rows are cut into pieces whose partial sums are combined by a
//     deterministic fix-up pass (no float atomics);
//   * inside a wavefront, lanes are laid out as R "non-zero slots" x GROUP "feature lanes" (R*GROUP = 64):
//     every step gathers R rows of X, each read as one contiguous, 16-byte-per-lane segment (coalesced), and
//     FMAs them into per-lane fp32 accumulators; the R slot accumulators are folded with a butterfly of
//     cross-lane shuffles when the row ends (wavefront-level reduction).  R = 1 walks the row in storage order
//     with one fmaf per non-zero = the reference's exact chain (bit-exact mode);
//   * the item's (col, val) stream is read as coalesced 64-element slices that stay in registers and are
//     broadcast to the slots through the LDS crossbar (ds_bpermute / v_readlane): the next slice is prefetched
//     while the current one is consumed, so the CSR stream never stalls the gathers;
//   * U independent gathers are kept in flight per lane (memory-level parallelism), x up to 8 waves per SIMD;
//   * block -> item mapping is XCD-aware: each of the 8 XCDs walks its own contiguous range of rows, so rows that
//     share neighbours (and the CSR stream) stay in one XCD's L2.
#include "sgl_common.h"

#include <atomic>
#include <memory>

namespace {

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
    using type = float;
};
template <>
struct VecT<2> {
    using type = float __attribute__((ext_vector_type(2)));
};
template <>
struct VecT<4> {
    using type = float __attribute__((ext_vector_type(4)));
};

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vzero() {
    typename VecT<VEC>::type z;
    if constexpr (VEC == 1) {
        z = 0.f;
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) z[e] = 0.f;
    }
    return z;
}

template <int VEC>
__device__ __forceinline__ void vfma(typename VecT<VEC>::type &acc, float v, const typename VecT<VEC>::type &x) {
    if constexpr (VEC == 1) {
        acc = __builtin_fmaf(v, x, acc);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = __builtin_fmaf(v, x[e], acc[e]);
    }
}

template <int VEC>
__device__ __forceinline__ void vadd_xor(typename VecT<VEC>::type &acc, int off) {
    if constexpr (VEC == 1) {
        acc += __shfl_xor(acc, off, 64);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += __shfl_xor(acc[e], off, 64);
    }
}

template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T *p) {
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}

template <bool NT, typename T>
__device__ __forceinline__ void st_stream(T *p, const T &v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// broadcast element `idx` (0..63) of a wave-distributed register to this lane
template <int R>
__device__ __forceinline__ int bcast_i(int v, int idx) {
    if constexpr (R == 1)
        return __builtin_amdgcn_readlane(v, idx);  // idx is wave-uniform -> SGPR result
    else
        return __builtin_amdgcn_ds_bpermute(idx << 2, v);
}
template <int R>
__device__ __forceinline__ float bcast_f(float v, int idx) {
    return __int_as_float(bcast_i<R>(__float_as_int(v), idx));
}

struct SpmmArgs {
    const int32_t *items;       // (row_begin,row_end) pairs
    const sgl::Piece *pieces;   // long-row pieces
    const int64_t *rowptr;
    const int32_t *col;
    const float *val;
    const float *x;
    float *y;
    float *partial;
    int64_t ldx, ldy, ldp;
    int32_t n_items, n_pieces, d, accumulate;
    int32_t piece_blocks, item_blocks_per_xcd, xcd_remap, waves;
    // optional fused epilogue  Y = clamp(alpha * (A X) + res, lo, hi)   (label propagation / C&S step,
    // reference: sgl/tricks/utils.py:55-56); epi == 0 -> plain store
    const float *res;
    int64_t ldres;
    float epi_alpha, epi_lo, epi_hi;
    int32_t epi;
    float *acc;                 // optional running aggregate (see Epilogue)
    int64_t ldacc;
    float acc_w, acc_div;
    int32_t acc_mode;
    // optional replicas of Y: every output row is ALSO stored at the same (row, column) offset of these matrices
    // (same leading dimension as y).  Used to push a rank's new rows straight into the peers' feature replicas
    // over xGMI from the producing kernel (sgl_spmm_multi_f32); n_more == 0 otherwise.
    float *y_more[7];
    int32_t n_more;
    const uint8_t *row_mask;   // optional [n_rows]: bit q set = replica q needs this row (NULL = every replica gets it)
    const int32_t *rowmap;     // optional [n_rows]: storage row -> output row (sgl_csr_set_rowmap); NULL = identity
};

struct MultiOut {
    float *p[7];   // already offset to the item's first row
    int n;
    const uint8_t *mask;   // already offset to the item's first row (or nullptr)
};

struct Epilogue {
    const float *res;   // row pointer already applied by the caller (may be nullptr)
    float alpha, lo, hi;
    int on;
    // running aggregate over hops, updated where the row is produced (Sum / Mean / SimpleWeighted MessageOps without a
    // second pass over the hop matrices): acc_mode 1: ACC += Y, 2: ACC += w * Y (rounded product, then add: the order of
    // hop_reduce_kernel), 3: ACC = max(ACC, Y) (+8: min), +4: ACC /= acc_div afterwards (Mean's one true division, on the
    // last hop).  Y itself is stored unchanged: it is the next hop's input.
    float *acc;         // row pointer already applied (nullptr = off)
    int64_t ldacc;
    float acc_w, acc_div;
    int acc_mode;
};

__device__ __forceinline__ float acc_apply(float a, float y, const Epilogue &e) {
    if ((e.acc_mode & 3) == 3)   // running extremum with torch's NaN rule (a NaN in any hop wins), +8: min instead of max
        return (e.acc_mode & 8) ? ((y < a || y != y) ? y : a) : ((y > a || y != y) ? y : a);
    a = ((e.acc_mode & 3) == 2) ? __fadd_rn(a, __fmul_rn(y, e.acc_w)) : __fadd_rn(a, y);
    if (e.acc_mode & 4) a = __fdiv_rn(a, e.acc_div);
    return a;
}

__device__ __forceinline__ float epi_apply(float v, float r, const Epilogue &e, bool has_res) {
    float t = __fmul_rn(e.alpha, v);          // alpha * spmm(...)   (rounded product, then rounded add: torch order)
    if (has_res) t = __fadd_rn(t, r);
    return t < e.lo ? e.lo : (t > e.hi ? e.hi : t);   // clamp that keeps NaN, like torch.clamp_
}

// One wavefront walks `nrows` consecutive rows whose non-zeros are colb/valb[0 .. tot) ; lane i of `my_rel`
// holds the offset of row i's first non-zero (lane nrows holds tot).
// Row map (sgl_csr_set_rowmap): the CSR's rows are stored in PROCESSING order (a locality ordering found at plan time), row i
// of the storage is row my_map[i] of the product.  Only the output side is indirect -- Y, the residual and the running
// aggregate are addressed with the mapped index from un-offset base pointers; the gathers use the original column ids, and a
// row's terms are added in their original order, so the result is bit-identical to the unpermuted matrix's.
struct RowMap {
    int my_map = 0;      // lane i: output row of the item's row i
    bool on = false;
};

template <int VEC, int GROUP, int NCH, int U, bool NT, bool MULTI>
__device__ __forceinline__ void run_rows(const int32_t *__restrict__ colb, const float *__restrict__ valb,
                                         const int my_rel, const int nrows, const int tot,
                                         const float *__restrict__ x, const int64_t ldx, float *__restrict__ out,
                                         const int64_t ldo, const int d, const bool accumulate, const int lane,
                                         const Epilogue epi, const int64_t ldres, const MultiOut mo,
                                         const RowMap rm = RowMap()) {
    using V = typename VecT<VEC>::type;
    constexpr int R = 64 / GROUP;
    const int s = (R == 1) ? 0 : (lane / GROUP);
    const int l = lane % GROUP;
    int colofs[NCH];
    bool on[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        colofs[ch] = (ch * GROUP + l) * VEC;
        on[ch] = colofs[ch] < d;
    }
    // current / next 64-element slice of the (col,val) stream, one element per lane
    int cbr = 0;
    int my_c = 0, nx_c = 0;
    float my_v = 0.f, nx_v = 0.f;
    if (lane < tot) {
        my_c = ld_stream<NT>(colb + lane);
        my_v = ld_stream<NT>(valb + lane);
    }
    if (64 + lane < tot) {
        nx_c = ld_stream<NT>(colb + 64 + lane);
        nx_v = ld_stream<NT>(valb + 64 + lane);
    }

    for (int ri = 0; ri < nrows; ++ri) {
        const int jb = __builtin_amdgcn_readlane(my_rel, ri);
        const int je = __builtin_amdgcn_readlane(my_rel, ri + 1);
        const int64_t ro = rm.on ? (int64_t)__builtin_amdgcn_readlane(rm.my_map, ri) : (int64_t)ri;   // output row
        float *orow = out + ro * ldo;
        V acc[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) acc[ch] = vzero<VEC>();
        if (accumulate && s == 0) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                if (on[ch]) acc[ch] = *reinterpret_cast<const V *>(orow + colofs[ch]);
        }
        int j = jb;
        while (j < je) {
            const int lim = min(je, cbr + 64);
            const int o = j - cbr;
            const int cnt = lim - j;
            // slot of a non-zero = its index WITHIN ITS ROW mod R, not its index within this chunk: the chunk boundaries are
            // the 64-element slices of the item's stream, i.e. they depend on which rows share the item.  With the row-relative
            // assignment every slot adds the same terms in the same order under any plan and any processing order of the rows
            // (row maps, other item sizes): packed layouts (d <= 64) are bit-reproducible across plans like the R = 1 walk.
            const int sh = (R == 1) ? 0 : ((s - (j - jb)) & (R - 1));
            int t = 0;
            for (; t + R * U <= cnt; t += R * U) {
                int c[U];
                float v[U];
                V xv[U][NCH];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = o + t + u * R + sh;
                    c[u] = bcast_i<R>(my_c, idx);
                    v[u] = bcast_f<R>(my_v, idx);
                }
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    if (on[ch]) {
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            xv[u][ch] = *reinterpret_cast<const V *>(x + (int64_t)c[u] * ldx + colofs[ch]);
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) xv[u][ch] = vzero<VEC>();
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) vfma<VEC>(acc[ch], v[u], xv[u][ch]);
            }
            for (; t < cnt; t += R) {
                const int idx = (o + t + sh) & 63;
                const bool valid = (t + sh) < cnt;
                const int c = bcast_i<R>(my_c, idx);
                const float v = bcast_f<R>(my_v, idx);
                if (valid) {
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch)
                        if (on[ch]) {
                            const V xv = *reinterpret_cast<const V *>(x + (int64_t)c * ldx + colofs[ch]);
                            vfma<VEC>(acc[ch], v, xv);
                        }
                }
            }
            j = lim;
            if (lim == cbr + 64) {  // slice exhausted: rotate, prefetch the one after next
                cbr += 64;
                my_c = nx_c;
                my_v = nx_v;
                if (cbr + 64 + lane < tot) {
                    nx_c = ld_stream<NT>(colb + cbr + 64 + lane);
                    nx_v = ld_stream<NT>(valb + cbr + 64 + lane);
                }
            }
        }
        if constexpr (R > 1) {
#pragma unroll
            for (int off = GROUP; off < 64; off <<= 1)
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) vadd_xor<VEC>(acc[ch], off);
        }
        if (s == 0) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                if (on[ch]) {
                    V v = acc[ch];
                    if (epi.on) {
                        const bool has_res = epi.res != nullptr;
                        V r = vzero<VEC>();
                        if (has_res) r = *reinterpret_cast<const V *>(epi.res + ro * ldres + colofs[ch]);
                        if constexpr (VEC == 1) {
                            v = epi_apply(v, r, epi, has_res);
                        } else {
#pragma unroll
                            for (int e = 0; e < VEC; ++e) v[e] = epi_apply(v[e], r[e], epi, has_res);
                        }
                    }
                    if (epi.acc) {
                        float *ap = epi.acc + ro * epi.ldacc + colofs[ch];
                        V a = *reinterpret_cast<const V *>(ap);
                        if constexpr (VEC == 1) {
                            a = acc_apply(a, v, epi);
                        } else {
#pragma unroll
                            for (int e = 0; e < VEC; ++e) a[e] = acc_apply(a[e], v[e], epi);
                        }
                        *reinterpret_cast<V *>(ap) = a;   // (non-temporal hints on this stream measured slower: +4.4 vs +4.1 ms / 10 hops)
                    }
                    st_stream<NT>(reinterpret_cast<V *>(orow + colofs[ch]), v);
                    if constexpr (MULTI) {
                        const int need = mo.mask ? (int)mo.mask[ri] : 0x7f;   // wave-uniform: one byte per row
#pragma unroll
                        for (int q = 0; q < 7; ++q)  // replicas (peer memory): posted stores, nothing waits on them
                            if (q < mo.n && ((need >> q) & 1))
                                *reinterpret_cast<V *>(mo.p[q] + ro * ldo + colofs[ch]) = v;
                    }
                }
        }
    }
}

template <int VEC, int GROUP, int NCH, int U, bool NT, bool MULTI>
__global__ __launch_bounds__(256) void spmm_kernel(const SpmmArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    if (b < a.piece_blocks) {
        const int p = b * a.waves + wave;
        if (p >= a.n_pieces) return;
        const sgl::Piece pc = a.pieces[p];
        const int my_rel = (lane == 0) ? 0 : pc.len;
        Epilogue none;
        none.res = nullptr;
        none.alpha = 1.f;
        none.lo = none.hi = 0.f;
        none.on = 0;   // pieces hold partial sums: the epilogue runs in the fix-up kernel
        none.acc = nullptr;
        none.ldacc = 0;
        none.acc_w = none.acc_div = 1.f;
        none.acc_mode = 0;
        MultiOut solo;
        solo.n = 0;
        solo.mask = nullptr;
        run_rows<VEC, GROUP, NCH, U, NT, false>(a.col + pc.begin, a.val + pc.begin, my_rel, 1, pc.len, a.x, a.ldx,
                                         a.partial + (int64_t)p * a.ldp, a.ldp, a.d, false, lane, none, 0, solo);
    } else {
        int ib = b - a.piece_blocks;
        if (a.xcd_remap) ib = (ib & 7) * a.item_blocks_per_xcd + (ib >> 3);
        const int item = ib * a.waves + wave;
        if (item >= a.n_items) return;
        const int row_begin = a.items[2 * item], row_end = a.items[2 * item + 1];
        const int nrows = row_end - row_begin;
        const int64_t rp = a.rowptr[(int64_t)row_begin + min(lane, nrows)];
        const int lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)rp);
        const int hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)rp >> 32));
        const int64_t base = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
        const int my_rel = (int)(rp - base);
        const int tot = __builtin_amdgcn_readlane(my_rel, nrows);
        RowMap rm;
        rm.on = a.rowmap != nullptr;            // then the output-side pointers stay un-offset: rows are addressed through the map
        rm.my_map = rm.on ? a.rowmap[(int64_t)row_begin + max(min(lane, nrows - 1), 0)] : 0;
        const int64_t first = rm.on ? 0 : row_begin;
        Epilogue epi;
        epi.res = a.res ? a.res + first * a.ldres : nullptr;
        epi.alpha = a.epi_alpha;
        epi.lo = a.epi_lo;
        epi.hi = a.epi_hi;
        epi.on = a.epi;
        epi.acc = a.acc ? a.acc + first * a.ldacc : nullptr;
        epi.ldacc = a.ldacc;
        epi.acc_w = a.acc_w;
        epi.acc_div = a.acc_div;
        epi.acc_mode = a.acc_mode;
        MultiOut mo;
        mo.n = MULTI ? a.n_more : 0;
        mo.mask = (MULTI && a.row_mask) ? a.row_mask + row_begin : nullptr;
#pragma unroll
        for (int q = 0; q < 7; ++q) mo.p[q] = (MULTI && q < a.n_more) ? a.y_more[q] + (int64_t)row_begin * a.ldy : nullptr;
        run_rows<VEC, GROUP, NCH, U, NT, MULTI>(a.col + base, a.val + base, my_rel, nrows, tot, a.x, a.ldx,
                                         a.y + first * a.ldy, a.ldy, a.d, a.accumulate != 0, lane, epi,
                                         a.ldres, mo, rm);
    }
}

// Y[row, :] = (accumulate ? Y[row, :] : 0) + sum of the row's piece partials, in storage order.
__global__ __launch_bounds__(256) void spmm_fixup_kernel(const int32_t *__restrict__ long_row,
                                                         const int32_t *__restrict__ long_first,
                                                         const float *__restrict__ partial, int64_t ldp,
                                                         float *__restrict__ y, int64_t ldy, int d, int accumulate,
                                                         const float *__restrict__ res, int64_t ldres, Epilogue epi,
                                                         MultiOut mo) {
    const int kblocks = (d + 255) / 256;
    const int lr = blockIdx.x / kblocks;
    const int k = (blockIdx.x % kblocks) * 256 + threadIdx.x;
    if (k >= d) return;
    const int row = long_row[lr];
    const int p0 = long_first[lr], p1 = long_first[lr + 1];
    float *yp = y + (int64_t)row * ldy + k;
    float acc = accumulate ? *yp : 0.f;
    for (int p = p0; p < p1; ++p) acc += partial[(int64_t)p * ldp + k];
    if (epi.on) acc = epi_apply(acc, res ? res[(int64_t)row * ldres + k] : 0.f, epi, res != nullptr);
    if (epi.acc) {   // here epi.acc is the matrix base (rows are absolute in the fix-up)
        float *ap = epi.acc + (int64_t)row * epi.ldacc + k;
        *ap = acc_apply(*ap, acc, epi);
    }
    *yp = acc;
    const int need = mo.mask ? (int)mo.mask[row] : 0x7f;
#pragma unroll
    for (int q = 0; q < 7; ++q)
        if (q < mo.n && ((need >> q) & 1)) mo.p[q][(int64_t)row * ldy + k] = acc;
}

template <int VEC, int GROUP, int NCH, int U, bool NT>
hipError_t launch_variant(const SpmmArgs &a, int grid, hipStream_t st) {
    if (a.n_more > 0)
        hipLaunchKernelGGL((spmm_kernel<VEC, GROUP, NCH, U, NT, true>), dim3(grid), dim3(64 * a.waves), 0, st, a);
    else
        hipLaunchKernelGGL((spmm_kernel<VEC, GROUP, NCH, U, NT, false>), dim3(grid), dim3(64 * a.waves), 0, st, a);
    return hipGetLastError();
}

template <int VEC, int GROUP, int NCH, int U>
hipError_t launch_nt(const SpmmArgs &a, int grid, hipStream_t st, bool nt) {
    return nt ? launch_variant<VEC, GROUP, NCH, U, true>(a, grid, st)
              : launch_variant<VEC, GROUP, NCH, U, false>(a, grid, st);
}

template <int VEC, int GROUP, int NCH>
hipError_t launch_u(const SpmmArgs &a, int grid, hipStream_t st, bool nt, int ulevel) {
    // gathers in flight per lane, scaled down with the number of column chunks to bound registers
    constexpr int UH = (NCH == 1) ? 8 : (NCH == 2 ? 4 : 2);
    constexpr int UL = UH / 2;
    if (ulevel == 2) return launch_nt<VEC, GROUP, NCH, UH * 2 / (NCH == 1 ? 1 : 2)>(a, grid, st, nt);
    if (ulevel == 3) return launch_nt<VEC, GROUP, NCH, (NCH == 1 && GROUP == 64) ? 32 : UH>(a, grid, st, nt);
    return ulevel == 0 ? launch_nt<VEC, GROUP, NCH, UL>(a, grid, st, nt) : launch_nt<VEC, GROUP, NCH, UH>(a, grid, st, nt);
}

template <int VEC>
hipError_t launch_group(const SpmmArgs &a, int grid, hipStream_t st, bool nt, int ulevel, int group, int nch) {
    if (nch == 1) {
        switch (group) {
            case 8:
                return launch_u<VEC, 8, 1>(a, grid, st, nt, ulevel);
            case 16:
                return launch_u<VEC, 16, 1>(a, grid, st, nt, ulevel);
            case 32:
                return launch_u<VEC, 32, 1>(a, grid, st, nt, ulevel);
            default:
                return launch_u<VEC, 64, 1>(a, grid, st, nt, ulevel);
        }
    }
    if (nch == 2) return launch_u<VEC, 64, 2>(a, grid, st, nt, ulevel);
    return launch_u<VEC, 64, 4>(a, grid, st, nt, ulevel);
}

}  // namespace

struct sgl_csr {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    const int64_t *d_rowptr = nullptr;
    const int32_t *d_col = nullptr;
    const float *d_val = nullptr;
    uint32_t flags = 0;
    int64_t n_items = 0, n_pieces = 0, n_long = 0;
    int32_t *d_items = nullptr;
    sgl::Piece *d_pieces = nullptr;
    int32_t *d_long_row = nullptr;
    int32_t *d_long_first = nullptr;
    float *d_partial = nullptr;
    size_t partial_cap = 0;  // floats
    std::vector<float *> retired;   // outgrown workspaces: a captured hipGraph may still replay into them (freed at destroy)
    int device = 0;
    const int32_t *d_rowmap = nullptr;   // caller's [n_rows] storage row -> output row (sgl_csr_set_rowmap), not owned
    int32_t *d_long_out = nullptr;       // output rows of the split rows under the row map
    // bumped by sgl_csr_set_values / sgl_csr_set_rowmap, set to ~0 by sgl_csr_destroy: a captured chain graph has the value and
    // row-map pointers of its capture baked in and refuses to replay once they changed (shared: outlives the handle)
    std::shared_ptr<std::atomic<uint64_t>> epoch = std::make_shared<std::atomic<uint64_t>>(0);
};

// permutation check of a row map on the device: every entry in range, no output row named twice
__global__ __launch_bounds__(256) void rowmap_check_kernel(const int32_t *__restrict__ map, const int64_t n, unsigned *__restrict__ seen,
                                                           int *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t m = map[i];
    if (m < 0 || m >= n) {
        atomicOr(bad, 1);
        return;
    }
    const unsigned bit = 1u << (m & 31);
    if (atomicOr(&seen[m >> 5], bit) & bit) atomicOr(bad, 2);
}

// The row pointers come to the host for the plan (sgl::build_plan).  Small ones in one copy; the 10^7 ... 10^8 rows of a
// papers100M-sized block (up to 888 MB) through two page-locked 32 MB staging buffers, the copy of chunk c + 1 in flight while
// chunk c is unpacked -- a pageable destination of that size is staged by the runtime at a fraction of the link rate -- and the
// host waits on the chunks' EVENTS, not on the stream: work the caller queued behind this call on other streams is not held up.
static int fetch_rowptr(std::vector<int64_t> &h, const int64_t *d, hipStream_t st) {
    const size_t n = h.size();
    constexpr size_t kChunk = (size_t)4 << 20;                        // elements: 32 MB
    if (n <= 2 * kChunk) {
        SGL_HIP_CHECK(hipMemcpyAsync(h.data(), d, n * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        SGL_HIP_CHECK(hipStreamSynchronize(st));
        return SGL_OK;
    }
    int64_t *stage[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = SGL_OK;
    auto cleanup = [&]() {
        for (int i = 0; i < 2; ++i) {
            if (stage[i]) (void)hipHostFree(stage[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
    };
    for (int i = 0; i < 2 && rc == SGL_OK; ++i) {
        if (hipHostMalloc((void **)&stage[i], kChunk * sizeof(int64_t), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess)
            rc = sgl::fail(SGL_ERR_ALLOC, "sgl_csr_create: no page-locked staging buffer for the row pointers");
    }
    const size_t n_chunks = (n + kChunk - 1) / kChunk;
    auto issue = [&](size_t c) -> hipError_t {
        const size_t off = c * kChunk, len = std::min(kChunk, n - off);
        hipError_t e = hipMemcpyAsync(stage[c & 1], d + off, len * sizeof(int64_t), hipMemcpyDeviceToHost, st);
        return e != hipSuccess ? e : hipEventRecord(ev[c & 1], st);
    };
    if (rc == SGL_OK && issue(0) != hipSuccess) rc = sgl::fail(SGL_ERR_INVALID, "sgl_csr_create: copying the row pointers failed");
    for (size_t c = 0; c < n_chunks && rc == SGL_OK; ++c) {
        if (hipEventSynchronize(ev[c & 1]) != hipSuccess) {
            rc = sgl::fail(SGL_ERR_INVALID, "sgl_csr_create: copying the row pointers failed");
            break;
        }
        const size_t off = c * kChunk, len = std::min(kChunk, n - off);
        // chunk c sits in stage[c & 1]; chunk c + 1 goes to the other buffer, whose contents (chunk c - 1) were unpacked in the last round
        if (c + 1 < n_chunks && issue(c + 1) != hipSuccess) {
            rc = sgl::fail(SGL_ERR_INVALID, "sgl_csr_create: copying the row pointers failed");
            break;
        }
        memcpy(h.data() + off, stage[c & 1], len * sizeof(int64_t));
    }
    if (rc != SGL_OK) (void)hipStreamSynchronize(st);                   // nothing may still write into the buffers we free
    cleanup();
    return rc;
}

SGL_EXPORT int sgl_csr_create(sgl_csr_t **out, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_rowptr,
                              const int32_t *d_col, const float *d_val, uint32_t flags, int32_t item_nnz,
                              int32_t long_row_nnz, void *stream) {
    if (!out) return sgl::fail(SGL_ERR_INVALID, "sgl_csr_create: NULL out");
    *out = nullptr;
    SGL_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "sgl_csr_create: negative size");
    SGL_REQUIRE(n_rows < INT32_MAX && n_cols < INT32_MAX, "sgl_csr_create: n_rows/n_cols must be < 2^31 (int32 ids)");
    SGL_REQUIRE(d_rowptr != nullptr, "sgl_csr_create: NULL row pointers");
    SGL_REQUIRE(nnz == 0 || (d_col && d_val), "sgl_csr_create: NULL col/val with nnz > 0");
    hipStream_t st = sgl::as_stream(stream);
    std::vector<int64_t> h_rowptr((size_t)n_rows + 1);
    {
        const int rc_fetch = fetch_rowptr(h_rowptr, d_rowptr, st);
        if (rc_fetch != SGL_OK) return rc_fetch;
    }
    SGL_REQUIRE(h_rowptr[0] == 0 && h_rowptr[n_rows] == nnz, "sgl_csr_create: rowptr[0]=%lld rowptr[n]=%lld but nnz=%lld",
                (long long)h_rowptr[0], (long long)h_rowptr[n_rows], (long long)nnz);
    if (item_nnz <= 0) {
        // one wavefront per item: small matrices get smaller items so that the chip (256 CUs x 4 SIMDs x 8 waves) still
        // sees enough wavefronts to hide memory latency; large ones use the 512-nnz default.  In between -- a launch of fewer
        // than ~200 000 such items, e.g. a rank's block of a sharded job: a few "rounds" of the 8 192 resident wavefronts --
        // 256-nnz items end the launch more evenly (profiles/r03_probe_small_launch.log: -3 % at an eighth of the
        // products-sized graph, neutral on the whole of it).  Results do not depend on the item size.
        const int64_t want_items = 256 * 4 * 8;
        const int64_t cap = nnz >= sgl::kSmallLaunchNnz ? sgl::kDefaultItemNnz : sgl::kDefaultItemNnz / 2;
        item_nnz = (int32_t)std::min<int64_t>(cap, std::max<int64_t>(16, nnz / want_items));
    }
    // Where a row is cut into pieces.  A row is ONE sequential chain of gathers in one wavefront; on a small matrix the whole
    // launch is only a few such chains long, so its longest row IS the launch (Pubmed-sized S0: a 171-nnz hub row of 2 KB gathers
    // took 62 us per hop whatever the item size; cut at 32 non-zeros the hop takes 44 us, 0.52 -> 0.73 of the roofline,
    // profiles/r04_small_graph_sweep.log).  The threshold depends on the matrix (its nnz) only -- never on the plan -- so any two
    // plans of one matrix still cut the same rows at the same places and agree bit for bit; strict order never cuts.
    if (long_row_nnz == 0)
        long_row_nnz = nnz < (1 << 18) ? 32 : nnz < (1 << 20) ? 128 : nnz < (1 << 22) ? 512 : sgl::kDefaultLongRowNnz;
    if (flags & SGL_CSR_STRICT_ORDER) long_row_nnz = -1;
    sgl::Plan plan;
    int rc = sgl::build_plan(plan, h_rowptr.data(), n_rows, item_nnz, long_row_nnz);
    if (rc != SGL_OK) return rc;

    sgl_csr_t *h = new (std::nothrow) sgl_csr_t();
    if (!h) return sgl::fail(SGL_ERR_ALLOC, "sgl_csr_create: out of memory");
    h->n_rows = n_rows;
    h->n_cols = n_cols;
    h->nnz = nnz;
    h->d_rowptr = d_rowptr;
    h->d_col = d_col;
    h->d_val = d_val;
    h->flags = flags;
    h->n_items = (int64_t)plan.items.size() / 2;
    h->n_pieces = (int64_t)plan.pieces.size();
    h->n_long = (int64_t)plan.long_row.size();
    (void)hipGetDevice(&h->device);
    auto upload = [&](void **dst, const void *src, size_t bytes) -> int {
        if (bytes == 0) return SGL_OK;
        SGL_HIP_CHECK(hipMalloc(dst, bytes));
        SGL_HIP_CHECK(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st));
        return SGL_OK;
    };
    rc = upload((void **)&h->d_items, plan.items.data(), plan.items.size() * sizeof(int32_t));
    if (rc == SGL_OK) rc = upload((void **)&h->d_pieces, plan.pieces.data(), plan.pieces.size() * sizeof(sgl::Piece));
    if (rc == SGL_OK) rc = upload((void **)&h->d_long_row, plan.long_row.data(), plan.long_row.size() * sizeof(int32_t));
    if (rc == SGL_OK && h->n_long > 0)
        rc = upload((void **)&h->d_long_first, plan.long_first.data(), plan.long_first.size() * sizeof(int32_t));
    if (rc == SGL_OK) {
        hipError_t e = hipStreamSynchronize(st);  // host vectors die at return
        if (e != hipSuccess) rc = sgl::fail((int)e, "sgl_csr_create: sync failed: %s", hipGetErrorString(e));
    }
    if (rc != SGL_OK) {
        sgl_csr_destroy(h);
        return rc;
    }
    *out = h;
    return SGL_OK;
}

SGL_EXPORT int sgl_csr_destroy(sgl_csr_t *h) {
    if (!h) return SGL_OK;
    h->epoch->store(~0ull);                // a chain graph captured on this handle refuses to replay from now on
    (void)hipFree(h->d_items);
    (void)hipFree(h->d_pieces);
    (void)hipFree(h->d_long_row);
    (void)hipFree(h->d_long_first);
    (void)hipFree(h->d_partial);
    (void)hipFree(h->d_long_out);
    for (float *p : h->retired) (void)hipFree(p);
    delete h;
    return SGL_OK;
}

// The handle's rows are stored in processing order: storage row i is row d_rowmap[i] of the product (a permutation of
// 0..n_rows-1, e.g. sgl_reorder_community's order applied with sgl_csr_permute_rows).  Every product of this handle then writes
// (and, for the epilogues, reads the residual / running aggregate of) output row d_rowmap[i]; X is gathered by the ORIGINAL
// column ids and every row keeps the order of its terms, so results are bit-identical to the unpermuted matrix's.  NULL
// removes the map.  The array must stay alive as long as the handle uses it.
SGL_EXPORT int sgl_csr_set_rowmap(sgl_csr_t *h, const int32_t *d_rowmap, void *stream) {
    if (!h) return sgl::fail(SGL_ERR_INVALID, "sgl_csr_set_rowmap: NULL handle");
    h->d_rowmap = nullptr;
    h->epoch->fetch_add(1);
    if (!d_rowmap || h->n_rows == 0) return SGL_OK;
    hipStream_t st = sgl::as_stream(stream);
    {   // a map that is not a permutation would make every SpMM write rows out of bounds or leave rows unwritten: checked once
        const size_t words = (size_t)(h->n_rows + 31) / 32;
        unsigned *d_seen = nullptr;
        SGL_HIP_CHECK(hipMalloc(&d_seen, (words + 1) * sizeof(unsigned)));
        int *d_bad = reinterpret_cast<int *>(d_seen + words);
        hipError_t e = hipMemsetAsync(d_seen, 0, (words + 1) * sizeof(unsigned), st);
        int bad = 0;
        if (e == hipSuccess) {
            hipLaunchKernelGGL(rowmap_check_kernel, dim3((unsigned)((h->n_rows + 255) / 256)), dim3(256), 0, st, d_rowmap, h->n_rows, d_seen, d_bad);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_seen);
        if (e != hipSuccess) return sgl::fail((int)e, "sgl_csr_set_rowmap: validation failed: %s", hipGetErrorString(e));
        SGL_REQUIRE(!(bad & 1), "sgl_csr_set_rowmap: map entry outside [0, n_rows)");
        SGL_REQUIRE(!(bad & 2), "sgl_csr_set_rowmap: the map names an output row twice (it must be a permutation)");
    }
    if (h->n_long > 0) {   // the split rows' fix-up writes whole output rows: give it their mapped ids
        std::vector<int32_t> map((size_t)h->n_rows), rows((size_t)h->n_long);
        SGL_HIP_CHECK(hipMemcpyAsync(map.data(), d_rowmap, map.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        SGL_HIP_CHECK(hipMemcpyAsync(rows.data(), h->d_long_row, rows.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        SGL_HIP_CHECK(hipStreamSynchronize(st));
        for (auto &r : rows) {
            SGL_REQUIRE(r >= 0 && r < h->n_rows && map[r] >= 0 && map[r] < h->n_rows, "sgl_csr_set_rowmap: map entry outside [0, n_rows)");
            r = map[r];
        }
        if (!h->d_long_out) SGL_HIP_CHECK(hipMalloc(&h->d_long_out, rows.size() * sizeof(int32_t)));
        SGL_HIP_CHECK(hipMemcpyAsync(h->d_long_out, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        SGL_HIP_CHECK(hipStreamSynchronize(st));
    }
    h->d_rowmap = d_rowmap;
    return SGL_OK;
}

// Same structure, new values: the plan depends on the row pointers only, so re-weighting the matrix (another r of the
// NAFS ensemble, another alpha of a PPR sweep: sgl/tasks/node_clustering.py:205-217) needs no new plan.
SGL_EXPORT int sgl_csr_set_values(sgl_csr_t *h, const float *d_val) {
    if (!h) return sgl::fail(SGL_ERR_INVALID, "sgl_csr_set_values: NULL handle");
    SGL_REQUIRE(h->nnz == 0 || d_val, "sgl_csr_set_values: NULL values");
    if (d_val != h->d_val) h->epoch->fetch_add(1);
    h->d_val = d_val;
    return SGL_OK;
}

SGL_EXPORT int sgl_csr_info(const sgl_csr_t *h, int64_t info[8]) {
    if (!h || !info) return sgl::fail(SGL_ERR_INVALID, "sgl_csr_info: NULL");
    info[0] = h->n_rows;
    info[1] = h->n_cols;
    info[2] = h->nnz;
    info[3] = h->n_items;
    info[4] = h->n_pieces;
    info[5] = h->n_long;
    info[6] = h->flags;
    info[7] = (int64_t)(h->partial_cap * sizeof(float));
    return SGL_OK;
}

static bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

static int pick_vec(const float *d_x, int64_t ldx, const float *d_y, int64_t ldy, int64_t d) {
    // vector width from alignment: every lane reads/writes VEC consecutive floats of a row
    if (d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned_to(d_x, 16) && aligned_to(d_y, 16)) return 4;
    if (d % 2 == 0 && ldx % 2 == 0 && ldy % 2 == 0 && aligned_to(d_x, 8) && aligned_to(d_y, 8)) return 2;
    return 1;
}

struct EpiHost {
    int on = 0;
    float alpha = 1.f, lo = 0.f, hi = 0.f;
    const float *res = nullptr;
    int64_t ldres = 0;
    int n_more = 0;             // replicas of Y (sgl_spmm_multi_f32)
    float *y_more[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const uint8_t *row_mask = nullptr;
    // running aggregate (sgl_spmm_acc_f32)
    float *acc = nullptr;
    int64_t ldacc = 0;
    float acc_w = 1.f, acc_div = 1.f;
    int acc_mode = 0;
};

static int spmm_slice(sgl_csr_t *h, const float *d_x, int64_t ldx, float *d_y, int64_t ldy, int d, int vec,
                      int accumulate, hipStream_t st, const EpiHost &eh) {
    const int lanes = d / vec;
    const bool strict = (h->flags & SGL_CSR_STRICT_ORDER) != 0;
    // Lane layout (measured on MI355X, profiles/r01_sweep*.log): rows wider than 64 floats are gathered one
    // non-zero per step by the whole wavefront (R = 1): as fast as two half-wave slots and it keeps the reference's
    // sequential fmaf order; narrower rows pack R = 64/GROUP non-zeros per step or the lanes would idle.
    int group = 64, nch = 1;
    if (lanes > 64) {
        const int need = (lanes + 63) / 64;
        nch = need <= 2 ? need : 4;
    } else if (!strict && lanes <= 16) {
        group = 8;
        while (group < lanes) group <<= 1;
    }
    const int64_t forced = sgl::tuning("spmm_group", 0);
    if (forced == 8 || forced == 16 || forced == 32 || forced == 64) {
        if (nch == 1 && forced >= lanes) group = (int)forced;
    }
    // gathers in flight per lane: 16 for the one-row-per-step layout, 8 for the packed ones (0 = this default).  A row's
    // remainder (nnz mod U) is gathered one dependent load at a time, so short rows want smaller batches: measured
    // (profiles/r02_flat_*.log) 51 nnz/row: U=16 8.73 ms vs U=8 8.79; 30 nnz/row (papers100M-shaped shard): U=8 37.2 ms
    // vs U=16 38.1; 6 nnz/row: U=4 11.27 vs U=8 11.36 vs U=16 13.9.  (A walk that batches across row ends was measured
    // too: within 1 % of this one with the right U, 2-4 % slower for d = 147 and in strict order -- not kept.)
    int ulevel = (group == 64 && nch == 1) ? 2 : 1;
    if (group == 64 && nch == 1 && h->n_rows > 0) {
        const double avg = (double)h->nnz / (double)h->n_rows;
        if (avg < 12.0) ulevel = 0;
        else if (avg < 40.0) ulevel = 1;
    }
    const int64_t un = sgl::tuning("spmm_unroll", 0);
    if (un == 1) ulevel = 0;
    if (un == 2) ulevel = 2;
    if (un == 3) ulevel = 1;
    if (un == 4) ulevel = 3;   // 32 gathers in flight (one-row-per-step layout only)
    const bool nt = sgl::tuning("spmm_nt", 0) != 0;
    int waves = (int)sgl::tuning("spmm_waves", 0);
    if (waves != 1 && waves != 2 && waves != 4) waves = 4;

    SpmmArgs a;
    a.items = h->d_items;
    a.pieces = h->d_pieces;
    a.rowptr = h->d_rowptr;
    a.col = h->d_col;
    a.val = h->d_val;
    a.x = d_x;
    a.y = d_y;
    a.ldx = ldx;
    a.ldy = ldy;
    a.ldp = (d + 3) / 4 * 4;
    a.n_items = (int32_t)h->n_items;
    a.n_pieces = (int32_t)h->n_pieces;
    a.d = d;
    a.accumulate = accumulate;
    a.waves = waves;
    a.res = eh.res;
    a.ldres = eh.ldres;
    a.epi_alpha = eh.alpha;
    a.epi_lo = eh.lo;
    a.epi_hi = eh.hi;
    a.epi = eh.on;
    a.acc = eh.acc;
    a.ldacc = eh.ldacc;
    a.acc_w = eh.acc_w;
    a.acc_div = eh.acc_div;
    a.acc_mode = eh.acc_mode;
    a.n_more = eh.n_more;
    a.row_mask = eh.row_mask;
    for (int q = 0; q < 7; ++q) a.y_more[q] = eh.y_more[q];
    a.rowmap = h->d_rowmap;
    a.piece_blocks = (int32_t)((h->n_pieces + waves - 1) / waves);
    const int64_t item_blocks = (h->n_items + waves - 1) / waves;
    a.xcd_remap = (!(h->flags & SGL_CSR_NO_XCD_REMAP) && sgl::tuning("spmm_xcd_remap", 1) != 0) ? 1 : 0;
    a.item_blocks_per_xcd = (int32_t)((item_blocks + 7) / 8);
    const int64_t grid64 = a.piece_blocks + (a.xcd_remap ? (int64_t)a.item_blocks_per_xcd * 8 : item_blocks);
    if (grid64 >= INT32_MAX) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_spmm_f32: grid too large");

    if (h->n_pieces > 0) {
        const size_t need = (size_t)h->n_pieces * (size_t)a.ldp;
        if (need > h->partial_cap) {
            // grow-only, and the outgrown buffer is kept until the handle dies: a ChainGraph captured earlier has its
            // address baked in and may be replayed after a wider eager call on the same handle
            if (h->d_partial) h->retired.push_back(h->d_partial);
            h->d_partial = nullptr;
            h->partial_cap = 0;
            SGL_HIP_CHECK(hipMalloc((void **)&h->d_partial, need * sizeof(float)));
            h->partial_cap = need;
        }
    }
    a.partial = h->d_partial;
    if (grid64 == 0) return SGL_OK;
    hipError_t e;
    if (vec == 4)
        e = launch_group<4>(a, (int)grid64, st, nt, ulevel, group, nch);
    else if (vec == 2)
        e = launch_group<2>(a, (int)grid64, st, nt, ulevel, group, nch);
    else
        e = launch_group<1>(a, (int)grid64, st, nt, ulevel, group, nch);
    if (e != hipSuccess) return sgl::fail((int)e, "sgl_spmm_f32: kernel launch failed: %s", hipGetErrorString(e));
    if (h->n_long > 0) {
        const int64_t fg = (int64_t)((d + 255) / 256) * h->n_long;
        if (fg >= INT32_MAX) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_spmm_f32: fix-up grid too large");
        Epilogue fe;
        fe.res = nullptr;
        fe.alpha = eh.alpha;
        fe.lo = eh.lo;
        fe.hi = eh.hi;
        fe.on = eh.on;
        fe.acc = eh.acc;
        fe.ldacc = eh.ldacc;
        fe.acc_w = eh.acc_w;
        fe.acc_div = eh.acc_div;
        fe.acc_mode = eh.acc_mode;
        MultiOut fmo;
        fmo.n = eh.n_more;
        fmo.mask = eh.row_mask;
        for (int q = 0; q < 7; ++q) fmo.p[q] = eh.y_more[q];
        hipLaunchKernelGGL(spmm_fixup_kernel, dim3((unsigned)fg), dim3(256), 0, st, h->d_rowmap ? h->d_long_out : h->d_long_row,
                           h->d_long_first, h->d_partial, a.ldp,
                           d_y, ldy, d, accumulate, eh.res, eh.ldres, fe, fmo);
        e = hipGetLastError();
        if (e != hipSuccess) return sgl::fail((int)e, "sgl_spmm_f32: fix-up launch failed: %s", hipGetErrorString(e));
    }
    return SGL_OK;
}

static int spmm_impl(sgl_csr_t *h, const float *d_x, int64_t ldx, float *d_y, int64_t ldy, int64_t d, int accumulate,
                     void *stream, EpiHost eh, const char *who) {
    if (!h) return sgl::fail(SGL_ERR_INVALID, "%s: NULL handle", who);
    SGL_REQUIRE(d >= 0 && d < INT32_MAX, "%s: bad d", who);
    if (d == 0 || h->n_rows == 0) return SGL_OK;
    SGL_REQUIRE(d_x && d_y, "%s: NULL X or Y", who);
    if (h->d_rowmap && eh.n_more > 0)
        return sgl::fail(SGL_ERR_UNSUPPORTED, "%s: a row-mapped handle does not support replicas", who);
    SGL_REQUIRE(ldx >= d && ldy >= d, "%s: leading dimension smaller than d", who);
    SGL_REQUIRE(aligned_to(d_x, 4) && aligned_to(d_y, 4), "%s: X/Y not 4-byte aligned", who);
    hipStream_t st = sgl::as_stream(stream);
    // one launch covers up to 64 lanes x 4 chunks x VEC columns; wider matrices go in column slices
    int vec = pick_vec(d_x, ldx, d_y, ldy, d);
    for (int q = 0; q < eh.n_more; ++q) {
        SGL_REQUIRE(eh.y_more[q] && aligned_to(eh.y_more[q], 4), "%s: bad replica pointer", who);
        if (vec == 4 && !aligned_to(eh.y_more[q], 16)) vec = aligned_to(eh.y_more[q], 8) && d % 2 == 0 ? 2 : 1;
        if (vec == 2 && !aligned_to(eh.y_more[q], 8)) vec = 1;
    }
    const int64_t vcap = sgl::tuning("spmm_vec", 0);   // experiments: cap the lane width (2 or 1 floats)
    if ((vcap == 1 || vcap == 2) && vcap < vec) vec = (int)vcap;
    if (eh.on && eh.res) {
        SGL_REQUIRE(eh.ldres >= d && aligned_to(eh.res, 4), "%s: bad residual matrix", who);
        if (vec == 4 && !(eh.ldres % 4 == 0 && aligned_to(eh.res, 16))) vec = (eh.ldres % 2 == 0 && aligned_to(eh.res, 8) && d % 2 == 0) ? 2 : 1;
        if (vec == 2 && !(eh.ldres % 2 == 0 && aligned_to(eh.res, 8))) vec = 1;
    }
    if (eh.acc) {
        SGL_REQUIRE(eh.ldacc >= d && aligned_to(eh.acc, 4), "%s: bad accumulator matrix", who);
        if (vec == 4 && !(eh.ldacc % 4 == 0 && aligned_to(eh.acc, 16))) vec = (eh.ldacc % 2 == 0 && aligned_to(eh.acc, 8) && d % 2 == 0) ? 2 : 1;
        if (vec == 2 && !(eh.ldacc % 2 == 0 && aligned_to(eh.acc, 8))) vec = 1;
    }
    const int64_t max_cols = 64 * 4 * vec;
    for (int64_t c0 = 0; c0 < d; c0 += max_cols) {
        const int dc = (int)std::min<int64_t>(max_cols, d - c0);
        EpiHost es = eh;
        if (es.res) es.res += c0;
        if (es.acc) es.acc += c0;
        for (int q = 0; q < es.n_more; ++q) es.y_more[q] += c0;
        int rc = spmm_slice(h, d_x + c0, ldx, d_y + c0, ldy, dc, vec, accumulate, st, es);
        if (rc != SGL_OK) return rc;
    }
    return SGL_OK;
}

SGL_EXPORT int sgl_spmm_f32(sgl_csr_t *h, const float *d_x, int64_t ldx, float *d_y, int64_t ldy, int64_t d,
                            int accumulate, void *stream) {
    return spmm_impl(h, d_x, ldx, d_y, ldy, d, accumulate, stream, EpiHost(), "sgl_spmm_f32");
}

// Y_0 = Y_1 = ... = A X: the product is stored into n_out matrices (same leading dimension).  Matrix 0 is normally
// local; the others may live in PEER GPUs' memory (IPC / symmetric-memory mappings): the producing wavefront pushes
// its finished rows straight into every peer's replica of the feature block over xGMI -- the all-gather between
// hops without copy kernels, staging buffers or a separate communication phase.
SGL_EXPORT int sgl_spmm_multi_f32(sgl_csr_t *h, const float *d_x, int64_t ldx, int n_out, float *const *h_y, int64_t ldy,
                                  int64_t d, const uint8_t *d_row_mask, void *stream) {
    SGL_REQUIRE(n_out >= 1 && n_out <= 8 && h_y, "sgl_spmm_multi_f32: n_out must be in [1, 8]");
    EpiHost eh;
    eh.row_mask = d_row_mask;
    eh.n_more = n_out - 1;
    for (int q = 1; q < n_out; ++q) eh.y_more[q - 1] = h_y[q];
    return spmm_impl(h, d_x, ldx, h_y[0], ldy, d, 0, stream, eh, "sgl_spmm_multi_f32");
}

// X_1 = A X_0, X_2 = A X_1, ... : the whole hop loop of GraphOp.propagate (base_op.py:29-35) issued from one call, so a
// small graph (Pubmed: tens of microseconds of kernel time per hop) is not dominated by per-hop host overhead.
SGL_EXPORT int sgl_spmm_chain_f32(sgl_csr_t *h, int n_hops, const float *d_x0, int64_t ldx0, float *const *h_y,
                                  const int64_t *h_ldy, int64_t d, void *stream) {
    if (!h) return sgl::fail(SGL_ERR_INVALID, "sgl_spmm_chain_f32: NULL handle");
    SGL_REQUIRE(n_hops >= 0 && (n_hops == 0 || (h_y && h_ldy)), "sgl_spmm_chain_f32: bad hop arrays");
    SGL_REQUIRE(n_hops == 0 || h->n_rows == h->n_cols, "sgl_spmm_chain_f32: repeated products need a square matrix");
    const float *cur = d_x0;
    int64_t ldc = ldx0;
    for (int k = 0; k < n_hops; ++k) {
        int rc = spmm_impl(h, cur, ldc, h_y[k], h_ldy[k], d, 0, stream, EpiHost(), "sgl_spmm_chain_f32");
        if (rc != SGL_OK) return rc;
        cur = h_y[k];
        ldc = h_ldy[k];
    }
    return SGL_OK;
}

// ---- hipGraph of a whole propagate(): for small graphs (Pubmed / Cora sized) each hop is a few tens of microseconds
// of kernel time, so the k launches (+ fix-ups) are captured once and replayed with a single hipGraphLaunch.
struct sgl_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::shared_ptr<std::atomic<uint64_t>> epoch;   // the handle's change counter and its value at capture time
    uint64_t captured = 0;
};

SGL_EXPORT int sgl_chain_graph_create(sgl_graph_t **out, sgl_csr_t *h, int n_hops, const float *d_x0, int64_t ldx0,
                                      float *const *h_y, const int64_t *h_ldy, int64_t d) {
    if (!out) return sgl::fail(SGL_ERR_INVALID, "sgl_chain_graph_create: NULL out");
    *out = nullptr;
    SGL_REQUIRE(h && n_hops >= 1 && h_y && h_ldy, "sgl_chain_graph_create: bad arguments");
    // everything that may allocate or synchronise happens BEFORE the capture: one eager run sizes the split-row
    // workspace (and validates the arguments)
    int rc = sgl_spmm_chain_f32(h, n_hops, d_x0, ldx0, h_y, h_ldy, d, nullptr);
    if (rc != SGL_OK) return rc;
    SGL_HIP_CHECK(hipDeviceSynchronize());
    hipStream_t cap = nullptr;
    SGL_HIP_CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    sgl_graph_t *g = new (std::nothrow) sgl_graph_t();
    if (!g) {
        (void)hipStreamDestroy(cap);
        return sgl::fail(SGL_ERR_ALLOC, "sgl_chain_graph_create: out of memory");
    }
    hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
        rc = sgl_spmm_chain_f32(h, n_hops, d_x0, ldx0, h_y, h_ldy, d, cap);
        e = hipStreamEndCapture(cap, &g->graph);
        if (rc != SGL_OK && e == hipSuccess) e = hipErrorUnknown;
    }
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    (void)hipStreamDestroy(cap);
    if (e != hipSuccess) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        return sgl::fail((int)e, "sgl_chain_graph_create: capture/instantiate failed: %s", hipGetErrorString(e));
    }
    g->epoch = h->epoch;
    g->captured = h->epoch->load();
    *out = g;
    return SGL_OK;
}

SGL_EXPORT int sgl_chain_graph_launch(sgl_graph_t *g, void *stream) {
    if (!g || !g->exec) return sgl::fail(SGL_ERR_INVALID, "sgl_chain_graph_launch: NULL graph");
    if (g->epoch && g->epoch->load() != g->captured)
        return sgl::fail(SGL_ERR_INVALID, "sgl_chain_graph_launch: the adjacency handle changed after the capture (new values / row map, "
                                          "or it was destroyed): capture the chain again");
    SGL_HIP_CHECK(hipGraphLaunch(g->exec, sgl::as_stream(stream)));
    return SGL_OK;
}

SGL_EXPORT int sgl_chain_graph_destroy(sgl_graph_t *g) {
    if (!g) return SGL_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return SGL_OK;
}

// Y = A X  and, in the same pass,  ACC <- ACC + w * Y  [ / divisor ]  or  ACC <- max / min(ACC, Y): the running aggregate of
// the Sum / Mean / SimpleWeighted / Max / Min MessageOps (message_op/{sum,mean,simple_weighted,max,min}_message_op.py) kept
// up to date where each row is produced, so those aggregators need no pass of their own over the hop matrices and no hop
// needs to be kept.
SGL_EXPORT int sgl_spmm_acc_f32(sgl_csr_t *h, const float *d_x, int64_t ldx, float *d_y, int64_t ldy, int64_t d, float *d_acc,
                                int64_t ldacc, float w, int mode, float divisor, void *stream) {
    SGL_REQUIRE(d_acc != nullptr, "sgl_spmm_acc_f32: NULL accumulator");
    SGL_REQUIRE(!(divisor == 0.f), "sgl_spmm_acc_f32: zero divisor");
    SGL_REQUIRE(mode >= SGL_ACC_SUM && mode <= SGL_ACC_MIN, "sgl_spmm_acc_f32: unknown mode %d", mode);
    SGL_REQUIRE(mode < SGL_ACC_MAX || divisor == 1.f, "sgl_spmm_acc_f32: max / min take no divisor");
    EpiHost eh;
    eh.acc = d_acc;
    eh.ldacc = ldacc;
    eh.acc_w = w;
    eh.acc_div = divisor;
    eh.acc_mode = mode >= SGL_ACC_MAX ? (3 | (mode == SGL_ACC_MIN ? 8 : 0)) : ((mode == SGL_ACC_WSUM ? 2 : 1) | (divisor != 1.f ? 4 : 0));
    return spmm_impl(h, d_x, ldx, d_y, ldy, d, 0, stream, eh, "sgl_spmm_acc_f32");
}

SGL_EXPORT int sgl_spmm_axpb_clamp_f32(sgl_csr_t *h, const float *d_x, int64_t ldx, float *d_y, int64_t ldy, int64_t d,
                                       float alpha, const float *d_res, int64_t ldres, float lo, float hi,
                                       void *stream) {
    EpiHost eh;
    eh.on = 1;
    eh.alpha = alpha;
    eh.lo = lo;
    eh.hi = hi;
    eh.res = d_res;
    eh.ldres = ldres;
    SGL_REQUIRE(!(lo > hi), "sgl_spmm_axpb_clamp_f32: lo > hi");
    return spmm_impl(h, d_x, ldx, d_y, ldy, d, 0, stream, eh, "sgl_spmm_axpb_clamp_f32");
}
