// Per-hop aggregators (MessageOp._combine) for gfx950: row-wise, single pass over the H hop matrices.
// Reference semantics: sgl/operators/message_op/*.py and sgl/operators/utils.py:91-116 (table in SURVEY.md App. A).
// All kernels are HBM-streaming: 16-byte loads per lane, consecutive lanes on consecutive addresses of a row.
// Arithmetic order mirrors the reference's torch-CPU order where that is defined (sum/mean/max/min are bit-exact).
#include "sgl_common.h"

namespace {

struct Hops {
    const float *p[SGL_MAX_HOPS];
    int64_t ld[SGL_MAX_HOPS];
};
struct HopsOut {
    float *p[SGL_MAX_HOPS];
    int64_t ld[SGL_MAX_HOPS];
};

using f4 = float __attribute__((ext_vector_type(4)));

template <int VEC>
struct Vt;
template <>
struct Vt<1> {
    using type = float;
};
template <>
struct Vt<4> {
    using type = f4;
};

template <int VEC, typename F>
__device__ __forceinline__ typename Vt<VEC>::type vmap2(const typename Vt<VEC>::type &a, const typename Vt<VEC>::type &b, F f) {
    typename Vt<VEC>::type r;
    if constexpr (VEC == 1) {
        r = f(a, b);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) r[e] = f(a[e], b[e]);
    }
    return r;
}

__device__ __forceinline__ float nan_max(float r, float x) { return (x > r || x != x) ? x : r; }
__device__ __forceinline__ float nan_min(float r, float x) { return (x < r || x != x) ? x : r; }

// ---- elementwise reductions over hops --------------------------------------------------------------------
template <int OP, int VEC>
__global__ __launch_bounds__(256) void hop_reduce_kernel(const Hops hx, const int n_hops, const float *__restrict__ w,
                                                         float *__restrict__ out, const int64_t ldo, const int64_t n,
                                                         const int d) {
    using V = typename Vt<VEC>::type;
    const int dv = d / VEC;
    const int64_t total = n * (int64_t)dv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / dv;
        const int col = (int)(i - row * dv) * VEC;
        V acc;
        {
            const V x0 = __builtin_nontemporal_load(reinterpret_cast<const V *>(hx.p[0] + row * hx.ld[0] + col));
            if constexpr (OP == SGL_REDUCE_SUM || OP == SGL_REDUCE_MEAN) {
                // Python sum() starts from int 0: 0 + X_s  (sum_message_op.py:10)
                V z;
                if constexpr (VEC == 1) z = 0.f; else z = (V){0.f, 0.f, 0.f, 0.f};
                acc = vmap2<VEC>(z, x0, [](float a, float b) { return __fadd_rn(a, b); });
            } else if constexpr (OP == SGL_REDUCE_WSUM) {
                const float w0 = w[0];
                acc = vmap2<VEC>(x0, x0, [w0](float a, float) { return __fmul_rn(a, w0); });
            } else {
                acc = x0;
            }
        }
        for (int h = 1; h < n_hops; ++h) {
            const V x = __builtin_nontemporal_load(reinterpret_cast<const V *>(hx.p[h] + row * hx.ld[h] + col));
            if constexpr (OP == SGL_REDUCE_SUM || OP == SGL_REDUCE_MEAN) {
                acc = vmap2<VEC>(acc, x, [](float a, float b) { return __fadd_rn(a, b); });
            } else if constexpr (OP == SGL_REDUCE_MAX) {
                acc = vmap2<VEC>(acc, x, [](float a, float b) { return nan_max(a, b); });
            } else if constexpr (OP == SGL_REDUCE_MIN) {
                acc = vmap2<VEC>(acc, x, [](float a, float b) { return nan_min(a, b); });
            } else {  // WSUM: rounded product, then add (operators/utils.py:100-101: mul, then sum)
                const float wh = w[h];
                acc = vmap2<VEC>(acc, x, [wh](float a, float b) { return __fadd_rn(a, __fmul_rn(b, wh)); });
            }
        }
        if constexpr (OP == SGL_REDUCE_MEAN) {
            const float hf = (float)n_hops;
            acc = vmap2<VEC>(acc, acc, [hf](float a, float) { return __fdiv_rn(a, hf); });  // true division
        }
        __builtin_nontemporal_store(acc, reinterpret_cast<V *>(out + row * ldo + col));
    }
}

// Backward of max / min over hops (max_message_op.py:12 / min_message_op.py:12: torch.stack(...).max(0)[0]): the gradient goes to ONE
// hop per element -- the first NaN if the element has one, otherwise the first hop that attains the extremum (torch's index rule) --
// and zeros to the others.  One pass: H hop reads + the incoming gradient, one write per requested dX_h.
template <bool IS_MAX, int VEC>
__global__ __launch_bounds__(256) void hop_select_bwd_kernel(const Hops hx, const int n_hops, const float *__restrict__ g,
                                                             const int64_t ldg, const HopsOut dx, const int64_t n, const int d) {
    using V = typename Vt<VEC>::type;
    const int dv = d / VEC;
    const int64_t total = n * (int64_t)dv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / dv;
        const int col = (int)(i - row * dv) * VEC;
        int sel[VEC];
        float best[VEC];
        bool isn[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sel[e] = -1;
            best[e] = 0.f;
            isn[e] = false;
        }
        for (int h = 0; h < n_hops; ++h) {
            const V xv = __builtin_nontemporal_load(reinterpret_cast<const V *>(hx.p[h] + row * hx.ld[h] + col));
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float x;
                if constexpr (VEC == 1) x = xv; else x = xv[e];
                if (!isn[e]) {
                    if (x != x) {
                        sel[e] = h;
                        isn[e] = true;
                    } else if (sel[e] < 0 || (IS_MAX ? x > best[e] : x < best[e])) {
                        sel[e] = h;
                        best[e] = x;
                    }
                }
            }
        }
        const V gv = __builtin_nontemporal_load(reinterpret_cast<const V *>(g + row * ldg + col));
        for (int h = 0; h < n_hops; ++h) {
            if (dx.p[h] == nullptr) continue;
            V o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if constexpr (VEC == 1) o = (sel[0] == h) ? gv : 0.f;
                else o[e] = (sel[e] == h) ? gv[e] : 0.f;
            }
            __builtin_nontemporal_store(o, reinterpret_cast<V *>(dx.p[h] + row * dx.ld[h] + col));
        }
    }
}

// out_k = sum_j W[k, j] X_j for k < n_out: a small dense matrix applied across the HOP dimension, every input element read once for
// all outputs.  What it is for: the hop matrices of PprGraphOp(K, r, alpha) are polynomials in the Laplacian's --
// ((1 - alpha) A_hat + alpha I)^k X = sum_j C(k, j) (1 - alpha)^j alpha^(k - j) A_hat^j X -- so every alpha of a PaSca-style sweep
// (sgl/search/search_config.py:14-15) follows from ONE propagation chain by a triangular mix of its K + 1 hop matrices: K + 1
// streams read, K written, instead of K more SpMMs.  Zero weights are skipped (a NaN / Inf in a hop that does not enter an output
// must not reach it); the sum is one fma chain in j order.
template <int NIN, int VEC>
__global__ __launch_bounds__(256) void hop_lincomb_kernel(const Hops hx, const int n_in, const HopsOut outs, const int n_out,
                                                          const float *__restrict__ w, const int ldw, const int64_t n, const int d) {
    using V = typename Vt<VEC>::type;
    const int dv = d / VEC;
    const int64_t total = n * (int64_t)dv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / dv;
        const int col = (int)(i - row * dv) * VEC;
        V x[NIN];
#pragma unroll
        for (int j = 0; j < NIN; ++j)
            if (j < n_in) x[j] = __builtin_nontemporal_load(reinterpret_cast<const V *>(hx.p[j] + row * hx.ld[j] + col));
        for (int k = 0; k < n_out; ++k) {
            V acc;
            if constexpr (VEC == 1) acc = 0.f; else acc = (V){0.f, 0.f, 0.f, 0.f};
            const float *wk = w + (int64_t)k * ldw;
#pragma unroll
            for (int j = 0; j < NIN; ++j)
                if (j < n_in) {
                    const float wkj = wk[j];                    // uniform: a scalar load
                    if (wkj != 0.f) acc = vmap2<VEC>(acc, x[j], [wkj](float a, float b) { return __builtin_fmaf(wkj, b, a); });
                }
            __builtin_nontemporal_store(acc, reinterpret_cast<V *>(outs.p[k] + row * outs.ld[k] + col));
        }
    }
}

// out[n,k] = sum_h W[n,h] X_h[n,k]; FMA: fma chain from 0 (bmm-like); !FMA: rounded product then add (NAFS loop)
template <int VEC, bool FMA>
__global__ __launch_bounds__(256) void hop_wsum2d_kernel(const Hops hx, const int n_hops, const float *__restrict__ w,
                                                         const int64_t ldw, float *__restrict__ out, const int64_t ldo,
                                                         const int64_t n, const int d) {
    using V = typename Vt<VEC>::type;
    const int dv = d / VEC;
    const int64_t total = n * (int64_t)dv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / dv;
        const int col = (int)(i - row * dv) * VEC;
        V acc;
        if constexpr (VEC == 1) acc = 0.f; else acc = (V){0.f, 0.f, 0.f, 0.f};
        const float *wr = w + row * ldw;
        for (int h = 0; h < n_hops; ++h) {
            const V x = __builtin_nontemporal_load(reinterpret_cast<const V *>(hx.p[h] + row * hx.ld[h] + col));
            const float wh = wr[h];
            if constexpr (FMA)
                acc = vmap2<VEC>(acc, x, [wh](float a, float b) { return __builtin_fmaf(wh, b, a); });
            else
                acc = vmap2<VEC>(acc, x, [wh](float a, float b) { return __fadd_rn(a, __fmul_rn(wh, b)); });
        }
        __builtin_nontemporal_store(acc, reinterpret_cast<V *>(out + row * ldo + col));
    }
}

// dX_h[n,k] = W[n,h] * dOut[n,k]
template <int VEC>
__global__ __launch_bounds__(256) void hop_wsum2d_dx_kernel(const HopsOut dx, const int n_hops, const float *__restrict__ w,
                                                            const int64_t ldw, const float *__restrict__ dout,
                                                            const int64_t lddo, const int64_t n, const int d) {
    using V = typename Vt<VEC>::type;
    const int dv = d / VEC;
    const int64_t total = n * (int64_t)dv;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / dv;
        const int col = (int)(i - row * dv) * VEC;
        const V g = __builtin_nontemporal_load(reinterpret_cast<const V *>(dout + row * lddo + col));
        const float *wr = w + row * ldw;
        for (int h = 0; h < n_hops; ++h) {
            if (dx.p[h] == nullptr) continue;
            const float wh = wr[h];
            const V r = vmap2<VEC>(g, g, [wh](float a, float) { return a * wh; });
            __builtin_nontemporal_store(r, reinterpret_cast<V *>(dx.p[h] + row * dx.ld[h] + col));
        }
    }
}

// 16-byte row accesses are legal for any d when every row pitch is a multiple of 4 floats (the vector that
// straddles column d stays inside the row's own padding); the elements beyond d are masked out of reductions.
// NT: streaming (non-temporal) hint -- every hop element is read exactly once; measured +3-10 % on the elementwise and
// fused-NAFS kernels, neutral-to-negative on the row-dot and concat kernels, which therefore do not use it
// (profiles/r02_aggregators.log).
template <int VEC, bool NT = false>
__device__ __forceinline__ typename Vt<VEC>::type load_masked(const float *p, int c, int d) {
    using V = typename Vt<VEC>::type;
    V v = NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(p + c)) : *reinterpret_cast<const V *>(p + c);
    if constexpr (VEC == 4) {
        if (c + 4 > d) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e >= d) v[e] = 0.f;
        }
    }
    return v;
}

// ---- row-wise reductions: LPR lanes cooperate on one row, 64/LPR rows per wavefront ------------------------
// All-lanes sum over groups of LPR consecutive lanes.  Inside a 16-lane row the exchange is done by the VALU's DPP
// modifiers (quad permutes, half-row / row mirrors) -- no LDS-crossbar instruction; only the steps that cross rows
// (LPR = 32, 64) use gfx950's v_permlane16_swap / v_permlane32_swap (VALU as well).  (With all steps on ds_bpermute the fused NAFS kernel, 2H reductions per row, ran at
// 0.48-0.52 of the streaming rate.)
template <int CTRL>
__device__ __forceinline__ float dpp_xchg(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(LPR == 8 || LPR == 16 || LPR == 32 || LPR == 64, "groups of 8 / 16 / 32 / 64 lanes");
    v += dpp_xchg<0xB1>(v);                        // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_xchg<0x4E>(v);                        // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_xchg<0x141>(v);                       // row_half_mirror: lane i <-> 7 - i (the other quad of the 8)
    if constexpr (LPR >= 16) v += dpp_xchg<0x140>(v);   // row_mirror: lane i <-> 15 - i (the other half of the row)
    if constexpr (LPR >= 32) {   // gfx950 v_permlane16_swap: odd rows of the first operand <-> even rows of the second;
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);      // with both operands = v the two results are v[lane] and v[lane ^ 16]
    }
    if constexpr (LPR >= 64) {   // v_permlane32_swap: upper half of the first operand <-> lower half of the second
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}

// dW[n,h] = <dOut[n,:], X_h[n,:]>
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void hop_rowdot_kernel(const Hops hx, const int n_hops, const float *__restrict__ g,
                                                         const int64_t ldg, float *__restrict__ dw, const int64_t lddw,
                                                         const int64_t n, const int d) {
    using V = typename Vt<VEC>::type;
    constexpr int RPB = 256 / LPR;  // rows per block
    const int l = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < n;
    const int64_t r = live ? row : 0;
    for (int h = 0; h < n_hops; ++h) {
        float acc = 0.f;
        if (live) {
            for (int c = l * VEC; c < d; c += LPR * VEC) {
                const V gv = load_masked<VEC>(g + r * ldg, c, d);
                const V xv = load_masked<VEC>(hx.p[h] + r * hx.ld[h], c, d);
                if constexpr (VEC == 1) {
                    acc = __builtin_fmaf(gv, xv, acc);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc = __builtin_fmaf(gv[e], xv[e], acc);
                }
            }
        }
        acc = group_sum<LPR>(acc);
        if (live && l == 0) dw[r * lddw + h] = acc;
    }
}

// The same with the H hop vectors of the row in registers: dOut is loaded once (not once per hop), the H x loads are in
// flight together and the H butterflies interleave.  d <= LPR * 4 * CH, H <= HMAX.
// GU: the second operand's rows are only dword-aligned (autograd's dense [n, d] gradient with d % 4 != 0): it is read with
// dword-aligned 16-byte loads (legal on gfx950, split by the hardware -- but it is one stream of H + 1, cheaper than the
// padded copy that would otherwise precede the kernel), the partial vector at column d element by element.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int LPR, int CH, int HMAX, bool GU = false>
__global__ __launch_bounds__(256) void hop_rowdot_reg_kernel(const Hops hx, const int n_hops, const float *__restrict__ g,
                                                             const int64_t ldg, float *__restrict__ dw, const int64_t lddw,
                                                             const int64_t n, const int d) {
    constexpr int RPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < n;
    const int64_t r = live ? row : 0;
    f4 gv[CH], xv[HMAX][CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * LPR + l) * 4;
        const bool on = live && col < d;
        if constexpr (GU) {
            gv[c] = (f4){0.f, 0.f, 0.f, 0.f};
            if (on) {
                const float *gp = g + r * ldg + col;
                if (col + 4 <= d) {
                    const f4u t = *reinterpret_cast<const f4u *>(gp);
                    gv[c] = (f4){t[0], t[1], t[2], t[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e)
                        if (col + e < d) gv[c][e] = gp[e];
                }
            }
        } else {
            gv[c] = on ? load_masked<4>(g + r * ldg, col, d) : (f4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            xv[h][c] = (on && h < n_hops) ? load_masked<4>(hx.p[h] + r * hx.ld[h], col, d) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    float acc[HMAX];
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
        acc[h] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[h] = __builtin_fmaf(gv[c][e], xv[h][c][e], acc[h]);
    }
#pragma unroll
    for (int h = 0; h < HMAX; ++h) acc[h] = group_sum<LPR>(acc[h]);   // independent chains: the scheduler interleaves them
    if (live && l == 0) {
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) dw[r * lddw + h] = acc[h];
    }
}

// NAFS weights: W[n,h] = softmax_h( <X0,Xh> / (|Xh|+1e-10) / (|X0|+1e-10) )   (over_smooth_distance_op.py:12-22)
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void nafs_weight_kernel(const Hops hx, const int n_hops, float *__restrict__ wout,
                                                          const int64_t ldw, const int64_t n, const int d) {
    using V = typename Vt<VEC>::type;
    constexpr int RPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < n;
    const int64_t r = live ? row : 0;
    float n0 = 0.f;
    float run_max = -INFINITY;
    // pass A: c_h for every hop (kept in the W row), running max
    for (int h = 0; h < n_hops; ++h) {
        float dot = 0.f, sq = 0.f;
        if (live) {
            for (int c = l * VEC; c < d; c += LPR * VEC) {
                const V a = load_masked<VEC>(hx.p[0] + r * hx.ld[0], c, d);
                const V b = load_masked<VEC>(hx.p[h] + r * hx.ld[h], c, d);
                if constexpr (VEC == 1) {
                    dot = __builtin_fmaf(a, b, dot);
                    sq = __builtin_fmaf(b, b, sq);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        dot = __builtin_fmaf(a[e], b[e], dot);
                        sq = __builtin_fmaf(b[e], b[e], sq);
                    }
                }
            }
        }
        dot = group_sum<LPR>(dot);
        sq = group_sum<LPR>(sq);
        const float nh = __fadd_rn(__fsqrt_rn(sq), 1e-10f);
        if (h == 0) n0 = nh;
        const float ch = __fdiv_rn(__fdiv_rn(dot, nh), n0);
        run_max = fmaxf(run_max, ch);
        if (live && l == 0) wout[r * ldw + h] = ch;
    }
    // pass B: softmax over the H scores of this row (lane 0 of the group; H is small)
    if (live && l == 0) {
        float *wr = wout + r * ldw;
        float sum = 0.f;
        for (int h = 0; h < n_hops; ++h) {
            const float e = expf(wr[h] - run_max);
            wr[h] = e;
            sum += e;
        }
        for (int h = 0; h < n_hops; ++h) wr[h] = __fdiv_rn(wr[h], sum);
    }
}

// Register budget of the register-resident row kernels: few hop vectors per lane -> insist on 8 workgroups per CU (<= 64 VGPRs);
// left at 4 the scheduler spends the 128 registers it is allowed on speculation (HMAX = 6: 128 VGPRs, 4 waves per SIMD) instead of
// the ~46 the kernel needs: NAFS at d = 128, H = 6 0.715 -> 0.739 of peak.  At MANY hops these kernels were VALU-issue-bound (see
// "one hop per lane" below), which is why two restructurings that bought wavefronts with extra instructions lost (round 3): an
// online softmax without a score array (0.58 vs 0.67) and one row per wavefront with the hops split over the half-waves
// (v_permlane32_swap exchanges; 8 waves, but 0.57 / 0.48 vs 0.67 / 0.62) -- profiles/r03_aggregators_{online_gate,hop_split}_experiment.log.
#define ROWREG_MIN_BLOCKS(HMAX, CH) (((HMAX) * (CH) <= 8) ? 8 : (((HMAX) * (CH) <= 16) ? 4 : 2))
// 12 hop vectors per lane (H = 11: BASELINE configs 4 / 5) sit right at a register boundary: the gate needs 70-73 VGPRs depending on
// what its epilogue carries; held to 72 (7 waves per SIMD, where it was measured in round 3: one register more costs a whole
// wavefront of occupancy and ~4 % at d = 128, H = 11).  Only where that does not spill (groups of >= 16 lanes).
#define GATE_MIN_BLOCKS(LPR, HMAX, CH) (((HMAX) * (CH) <= 8) ? 8 : (((CH) == 1 && (HMAX) <= 12 && (LPR) >= 16) ? 7 : ROWREG_MIN_BLOCKS(HMAX, CH)))

// the recursive gate carries a second vector and two scalars per hop: 5 waves per SIMD (<= 96 VGPRs; 6 spill at 12) up to 12 hop vectors without
// spilling; its array form (more hops than lanes in a group: 8-lane groups only) keeps three scalars per hop and gets 2
#define RECUR_MIN_BLOCKS(LPR, HMAX, CH) (((HMAX) * (CH) <= 8) ? 8 : (((HMAX) > (LPR)) ? 2 : (((CH) == 1 && (HMAX) <= 12 && (LPR) >= 16) ? 5 : ROWREG_MIN_BLOCKS(HMAX, CH))))

// ---- lanes x chunks of a register-resident row kernel ------------------------------------------------------------------------
// A row of d floats is ceil(d / 4) 16-byte slots; LPR lanes take CH slots each (slot (c * LPR + l) of the row for lane l, chunk c),
// 64 / LPR rows per wavefront.  Power-of-two groups leave slots idle when the row is not a power of two wide, and idle slots still
// cost their share of every load and VALU instruction: d = 147 (BASELINE config 3: 100 features + 47 label columns = 37 slots) on
// 32 lanes x 2 chunks idles 27 of 64.  Narrow groups with more chunks per lane fit such rows far better -- 8 lanes x 5 chunks
// (40 slots, 8 rows per wavefront, every load instruction of a lane group is one whole 128-byte line) or 16 x 3 (48 slots) -- at
// the price of CH x HMAX hop vectors in registers, so they are instantiated for few hops only (<= 6 / <= 12) and chosen when they
// at least halve the idle slots.  Measured at d = 147 (profiles/r04_aggregators_layouts.log): 16 x 3 is as fast as 32 x 2 at 6 hops
// and 12-15 % faster at 11 (row-dot 0.597 -> 0.661 of peak, gate 0.539 -> 0.613, NAFS 0.536 -> 0.612); 8 x 5 (151 VGPRs, 3 waves
// per SIMD) pays only in the jk-score kernel (0.585 -> 0.657 at 6 hops) and is instantiated for that kernel alone.  At 6 hops the
// gate / NAFS kernels are NOT issue-bound -- halving their VALU instructions and quartering their wavefronts changed nothing
// (profiles/r04_agg_pmc.md) -- what they lost against the plain sum was the partly written last line of the output row
// (store_row above).
struct RowLayout {
    int lpr, ch;
};


// ---- per-row scalars, one hop per lane -------------------------------------------------------------------------------------
// After the row reductions every lane of a row's group holds all H per-hop scalars.  Evaluating sigmoid / softmax / the IEEE
// divisions hop after hop costs H instruction sequences per WAVEFRONT (every lane repeats them): ~1 050 VALU instructions at
// H = 11, i.e. 85 % of the VALU issue slots at the streaming rate (profiles/r03_agg_pmc.md) -- the row kernels were issue-bound.
// With H <= LPR lane l of the group takes hop l (it keeps hop l's reduced scalars as they are produced): ONE sequence per
// wavefront, then the weights are broadcast back -- and no per-hop score array stays in registers (NAFS at 12 hop vectors:
// 82 -> 62 VGPRs, 5 -> 8 wavefronts per SIMD).  The
// arithmetic (operations, their order, IEEE division, the hop-ordered sum) is unchanged: results are bit-identical.
template <int LPR>
__device__ __forceinline__ float from_lane(const float v, const int h) {   // value of lane h of this lane's group
    if constexpr (LPR == 64) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), h));
    } else {
        const int lane = (int)(threadIdx.x & 63);
        return __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & ~(LPR - 1)) + h) << 2, __float_as_int(v)));
    }
}
template <int LPR>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, dpp_xchg<0xB1>(v));
    v = fmaxf(v, dpp_xchg<0x4E>(v));
    v = fmaxf(v, dpp_xchg<0x141>(v));
    if constexpr (LPR >= 16) v = fmaxf(v, dpp_xchg<0x140>(v));
    if constexpr (LPR >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    if constexpr (LPR >= 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}

// Output row of a register-resident row kernel.  dw = the columns the kernel writes: exactly d, or -- when the caller declared the
// tail of the row's pitch to be padding (out_cols() below) -- d + pad, the pad columns as zeros.  Why: a row of d = 147 floats on a
// 160-float pitch ends 52 bytes short of its last 128-byte line, and a line that is only partly written costs a read-modify-write in
// the ECC-protected HBM: the output write of the gate / NAFS kernels ran at 2.7 TB/s at d = 147 against 5.8 TB/s at d = 160
// (profiles/r04_aggregators.log; the element-wise kernels always streamed whole pitches).
template <int LPR, int CH>
__device__ __forceinline__ void store_row(float *__restrict__ orow, const f4 (&acc)[CH], const int l, const bool live, const int d,
                                          const int dw) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * LPR + l) * 4;
        if (live && col < dw) {
            f4 v = acc[c];
            if (col + 4 <= dw) {                    // whole vector; what lies beyond d is padding the kernel owns: zeros
                if (col + 4 > d) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e >= d) v[e] = 0.f;
                }
                // non-temporal: with plain (write-back) stores these kernels are 7-20 % slower at every width -- also at widths whose
                // rows share lines with their neighbours (d = 100: 0.906 -> 0.935 ms, d = 147: 1.98 -> 2.42 ms, d = 128: 1.51 -> 1.65 ms;
                // profiles/r04_aggregators_plain_stores_experiment.log)
                __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(orow + col));
            } else {                                // dw == d, the vector straddles it: never write past the caller's d columns
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < dw) orow[col + e] = v[e];
            }
        }
    }
}

// NAFS hop SWEEP: every prefix of the hop list in ONE pass.  The clustering / link-prediction tasks evaluate hops = 0, 1, ..., K
// (tasks/node_clustering.py:139,176-178: _k_hop_cluster(hop) for every hop of the range), each from scratch: sum_h h SpMMs and as
// many aggregations per r.  The cosine score c_j = <X_0, X_j> / (|X_j| + 1e-10) / (|X_0| + 1e-10) of hop j does not depend on how
// many hops follow, and softmax_j(c)_j = e^{c_j} / sum_{i <= h} e^{c_i}, so with a running numerator sum_{j <= h} e^{c_j} X_j and
// denominator the feature matrix of EVERY prefix h falls out of one stream over the K + 1 hop matrices (|c| <= 1: no max to
// subtract).  Each hop element is read once; a row is emitted (numerator / denominator) after the hops whose bit is set in
// `emit`, into outs[rank of the bit], where it is combined with what the earlier r values of the ensemble left there:
//   0 store   1 add (Python's sum(): ((0 + f_r0) + f_r1) + ...)   2 add, then true division by `divisor` (the last r of 'mean')
//   3 max (values only, NaN like torch)                                   (node_clustering.py:242-249)
// Registers: X_0, the numerator and kPrefixUnroll hop vectors in flight per lane -- independent of the number of hops.
constexpr int kPrefixUnroll = 4;

template <int LPR, int CH>
__global__ __launch_bounds__(256) void nafs_prefix_kernel(const Hops hx, const int n_hops, const uint64_t emit, const HopsOut outs,
                                                          const int combine, const float divisor, const int64_t n, const int d,
                                                          const int dw) {
    constexpr int RPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < n;
    const int64_t r = live ? row : 0;
    f4 x0[CH], acc[CH];
    bool on[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        on[c] = live && ((c * LPR + l) * 4 < d);
        x0[c] = (f4){0.f, 0.f, 0.f, 0.f};
        acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
        if (on[c]) x0[c] = load_masked<4, true>(hx.p[0] + r * hx.ld[0], (c * LPR + l) * 4, d);
    }
    float n0 = 1.f, den = 0.f;
    for (int hb = 0; hb < n_hops; hb += kPrefixUnroll) {
        f4 x[kPrefixUnroll][CH];
#pragma unroll
        for (int u = 0; u < kPrefixUnroll; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int h = hb + u;
                x[u][c] = (h == 0) ? x0[c] : (f4){0.f, 0.f, 0.f, 0.f};
                if (h > 0 && h < n_hops && on[c]) x[u][c] = load_masked<4, true>(hx.p[h] + r * hx.ld[h], (c * LPR + l) * 4, d);
            }
#pragma unroll
        for (int u = 0; u < kPrefixUnroll; ++u) {
            const int h = hb + u;
            if (h >= n_hops) break;
            float dot = 0.f, sq = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dot = __builtin_fmaf(x0[c][e], x[u][c][e], dot);
                    sq = __builtin_fmaf(x[u][c][e], x[u][c][e], sq);
                }
            dot = group_sum<LPR>(dot);
            sq = group_sum<LPR>(sq);
            const float nh = __fadd_rn(__fsqrt_rn(sq), 1e-10f);
            if (h == 0) n0 = nh;
            const float ex = expf(__fdiv_rn(__fdiv_rn(dot, nh), n0));
            den += ex;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c][e] = __builtin_fmaf(ex, x[u][c][e], acc[c][e]);
            if ((emit >> h) & 1ull) {
                const int k = __popcll(emit & ((1ull << h) - 1ull));
                float *__restrict__ orow = outs.p[k] + r * outs.ld[k];
                const float inv = __fdiv_rn(1.f, den);
                f4 o[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[c][e] = acc[c][e] * inv;
                    if (combine != 0 && on[c]) {
                        const f4 old = load_masked<4>(orow, (c * LPR + l) * 4, d);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (combine == 3) o[c][e] = nan_max(old[e], o[c][e]);
                            else o[c][e] = __fadd_rn(old[e], o[c][e]);
                            if (combine == 2) o[c][e] = __fdiv_rn(o[c][e], divisor);
                        }
                    }
                }
                store_row<LPR, CH>(orow, o, l, live, d, dw);
            }
        }
    }
}

// Fused NAFS: one pass over the H hop rows held in registers -> cosine scores -> softmax -> weighted sum.
// LPR lanes per row, CH float4 chunks per lane (d <= LPR*4*CH), H <= HMAX.  Same arithmetic as the two-pass path.
template <int LPR, int CH, int HMAX>
__global__ __launch_bounds__(256, ROWREG_MIN_BLOCKS(HMAX, CH)) void nafs_fused_kernel(const Hops hx, const int n_hops, float *__restrict__ out,
                                                         const int64_t ldo, float *__restrict__ wout, const int64_t ldw,
                                                         const int64_t n, const int d, const int dw) {
    constexpr int RPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < n;
    const int64_t r = live ? row : 0;
    f4 x[HMAX][CH];
    bool on[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) on[c] = live && ((c * LPR + l) * 4 < d);
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            x[h][c] = (f4){0.f, 0.f, 0.f, 0.f};
            if (h < n_hops && on[c]) x[h][c] = load_masked<4, true>(hx.p[h] + r * hx.ld[h], (c * LPR + l) * 4, d);
        }
    }
    f4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (HMAX <= LPR) {
        // lane l of the row's group collects <x_0, x_l> and |x_l|^2 and owns hop l from here on (see "one hop per lane")
        float dl = 0.f, ql = 0.f;
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
                float dot = 0.f, sq = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dot = __builtin_fmaf(x[0][c][e], x[h][c][e], dot);
                        sq = __builtin_fmaf(x[h][c][e], x[h][c][e], sq);
                    }
                dot = group_sum<LPR>(dot);
                sq = group_sum<LPR>(sq);
                dl = (l == h) ? dot : dl;
                ql = (l == h) ? sq : ql;
            }
        const bool mine = l < n_hops;
        const float nh = __fadd_rn(__fsqrt_rn(ql), 1e-10f);
        const float n0 = from_lane<LPR>(nh, 0);
        const float sc = mine ? __fdiv_rn(__fdiv_rn(dl, nh), n0) : -INFINITY;
        const float run_max = group_max<LPR>(sc);
        const float ex = mine ? expf(sc - run_max) : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) sum += from_lane<LPR>(ex, h);     // in hop order, like the sequential formulation
        const float wl = __fdiv_rn(ex, sum);
        if (wout && live && mine) wout[r * ldw + l] = wl;
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
                const float w = from_lane<LPR>(wl, h);
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][e] = __fadd_rn(acc[c][e], __fmul_rn(w, x[h][c][e]));
            }
    } else {
        float score[HMAX];
        float n0 = 0.f, run_max = -INFINITY;
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            score[h] = -INFINITY;
            if (h < n_hops) {
                float dot = 0.f, sq = 0.f;
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dot = __builtin_fmaf(x[0][c][e], x[h][c][e], dot);
                        sq = __builtin_fmaf(x[h][c][e], x[h][c][e], sq);
                    }
                dot = group_sum<LPR>(dot);
                sq = group_sum<LPR>(sq);
                const float nh = __fadd_rn(__fsqrt_rn(sq), 1e-10f);
                if (h == 0) n0 = nh;
                score[h] = __fdiv_rn(__fdiv_rn(dot, nh), n0);
                run_max = fmaxf(run_max, score[h]);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
                score[h] = expf(score[h] - run_max);
                sum += score[h];
            }
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
                const float w = __fdiv_rn(score[h], sum);
                if (wout && live && l == 0) wout[r * ldw + h] = w;
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][e] = __fadd_rn(acc[c][e], __fmul_rn(w, x[h][c][e]));
            }
    }
    store_row<LPR, CH>(out + r * ldo, acc, l, live, d, dw);
}

// Fused learnable gate (LearnableWeightedMessageOp 'gate', learnable_weighted_messahe_op.py:67-71 + two_dim_weighted_add): one
// pass over the H hop rows held in registers -> score_h = <X_h[n], v> + b -> sigmoid -> softmax over the hops -> weighted sum.
// Each hop element is read ONCE (the two-pass form reads every hop for the scores and again for the sum).  The dot products use
// the lane layout and order of hop_rowdot_reg_kernel and the sum the FMA chain of hop_wsum2d_kernel, so the result equals the
// two-pass path up to the rounding of expf.  wout [n, H] = the softmax weights, gout [n, H] = the sigmoid outputs (what the
// backward needs besides the hops).
template <int LPR, int CH, int HMAX>
__global__ __launch_bounds__(256, GATE_MIN_BLOCKS(LPR, HMAX, CH)) void gate_fused_kernel(const Hops hx, const int n_hops, const float *__restrict__ vec,
                                                         const float bias_arg, const float *__restrict__ bias_ptr,
                                                         float *__restrict__ out, const int64_t ldo,
                                                         float *__restrict__ wout, const int64_t ldw, float *__restrict__ gout,
                                                         const int64_t ldg, const int64_t n, const int d, const int dw) {
    constexpr int RPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < n;
    const int64_t r = live ? row : 0;
    const float bias = bias_ptr ? *bias_ptr : bias_arg;       // a bias that lives on the device is read here: no host round trip
    f4 x[HMAX][CH], vv[CH];
    bool on[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * LPR + l) * 4;
        on[c] = live && col < d;
        vv[c] = (col < d) ? load_masked<4>(vec, col, d) : (f4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            x[h][c] = (f4){0.f, 0.f, 0.f, 0.f};
            if (h < n_hops && on[c]) x[h][c] = load_masked<4, true>(hx.p[h] + r * hx.ld[h], (c * LPR + l) * 4, d);
        }
    }
    f4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (HMAX <= LPR) {
        // lane l of the row's group collects the score of hop l and owns that hop from here on (see "one hop per lane")
        float sl = 0.f;
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) sc = __builtin_fmaf(vv[c][e], x[h][c][e], sc);
            sc = group_sum<LPR>(sc);
            sl = (l == h) ? sc : sl;
        }
        const bool mine = l < n_hops;
        const float g = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-__fadd_rn(sl, bias))));     // sigmoid(Linear(x))
        const float run_max = group_max<LPR>(mine ? g : -INFINITY);
        const float ex = mine ? expf(g - run_max) : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) sum += from_lane<LPR>(ex, h);     // in hop order, like the sequential formulation
        const float wl = __fdiv_rn(ex, sum);
        if (live && mine) {
            if (gout) gout[r * ldg + l] = g;
            if (wout) wout[r * ldw + l] = wl;
        }
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
                const float w = from_lane<LPR>(wl, h);
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][e] = __builtin_fmaf(w, x[h][c][e], acc[c][e]);
            }
    } else {
        float score[HMAX];
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            score[h] = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) score[h] = __builtin_fmaf(vv[c][e], x[h][c][e], score[h]);
        }
#pragma unroll
        for (int h = 0; h < HMAX; ++h) score[h] = group_sum<LPR>(score[h]);   // independent chains: interleaved by the scheduler
        float run_max = -INFINITY;
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
                score[h] = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-__fadd_rn(score[h], bias))));     // sigmoid(Linear(x))
                run_max = fmaxf(run_max, score[h]);
                if (gout && live && l == 0) gout[r * ldg + h] = score[h];
            }
        float sum = 0.f;
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            if (h < n_hops) {
                score[h] = expf(score[h] - run_max);
                sum += score[h];
            }
        }
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
                const float w = __fdiv_rn(score[h], sum);
                if (wout && live && l == 0) wout[r * ldw + h] = w;
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][e] = __builtin_fmaf(w, x[h][c][e], acc[c][e]);
            }
    }
    store_row<LPR, CH>(out + r * ldo, acc, l, live, d, dw);
}

__device__ __forceinline__ float fast_exp(const float x) { return __builtin_amdgcn_exp2f(__fmul_rn(x, 1.4426950408889634f)); }
__device__ __forceinline__ float fast_sigmoid(const float x) { return __builtin_amdgcn_rcpf(__fadd_rn(1.f, fast_exp(-x))); }

// Fused recursive gate (IterateLearnableWeightedMessageOp 'recursive', iterate_learnable_weighted_message_op.py:28-51; GAMLP-R).
// The reference walks the hops: step i scores Linear([X_i || acc]) -> sigmoid, appends it to the (already soft-maxed) weights of
// the steps before, soft-maxes all i + 1 of them again and rebuilds acc = sum_j w_j X_j -- H (H + 3) / 2 reads of a hop matrix
// and H accumulator writes.  The recursion is local to a row and acc is always a weighted sum of that row's hops, so
//     Linear([X_i || acc_{i-1}]) = a_i + sum_{j < i} w_j c_j + b,      a_h = <X_h, w_x>,  c_h = <X_h, w_acc>
// and with the hop rows in registers the whole operator is ONE pass: 2 H row-dots, the recursion on the H scalar pairs (lane h of
// the row's group owns hop h, see "one hop per lane"; the reductions over hops stay inside one 16-lane DPP row), the final
// weighted sum.  vec = [w_x | w_acc], each zero-padded to whole float4 (dv floats apart).  wout / aout / cout [n, H]: the final
// weights and the two score matrices, which is all the backward needs besides the hops.
template <int LPR, int CH, int HMAX>
__global__ __launch_bounds__(256, RECUR_MIN_BLOCKS(LPR, HMAX, CH)) void recursive_fused_kernel(const Hops hx, const int n_hops, const float *__restrict__ vec, const int dv,
                                                              const float bias_arg, const float *__restrict__ bias_ptr,
                                                              float *__restrict__ out, const int64_t ldo,
                                                              float *__restrict__ wout, const int64_t ldw, float *__restrict__ aout,
                                                              const int64_t lda, float *__restrict__ cout, const int64_t ldc,
                                                              const int64_t n, const int d, const int dw) {
    constexpr int RPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < n;
    const int64_t r = live ? row : 0;
    const float bias = bias_ptr ? *bias_ptr : bias_arg;
    f4 x[HMAX][CH], vx[CH], va[CH];
    bool on[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * LPR + l) * 4;
        on[c] = live && col < d;
        vx[c] = (col < d) ? load_masked<4>(vec, col, d) : (f4){0.f, 0.f, 0.f, 0.f};
        va[c] = (col < d) ? load_masked<4>(vec + dv, col, d) : (f4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            x[h][c] = (f4){0.f, 0.f, 0.f, 0.f};
            if (h < n_hops && on[c]) x[h][c] = load_masked<4, true>(hx.p[h] + r * hx.ld[h], (c * LPR + l) * 4, d);
        }
    }
    f4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (HMAX <= LPR) {
        constexpr int SL = LPR < 16 ? LPR : 16;          // the hop lanes 0 .. H-1 of a group lie in its first DPP row
        float al = 0.f, cl = 0.f;
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            float sa = 0.f, sc = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sa = __builtin_fmaf(vx[c][e], x[h][c][e], sa);
                    sc = __builtin_fmaf(va[c][e], x[h][c][e], sc);
                }
            sa = group_sum<LPR>(sa);
            sc = group_sum<LPR>(sc);
            al = (l == h) ? sa : al;
            cl = (l == h) ? sc : cl;
        }
        float wl = (l == 0) ? 1.f : 0.f;                 // step 0: the soft-max of one score
        // One step = ~25 VALU instructions per wavefront: every lane evaluates the sigmoid with ITS a_l (only lane i's is used, so
        // a_i is never broadcast), the soft-max runs without the max subtraction (its inputs are soft-max weights and a sigmoid,
        // all in (0, 1)) and exp / reciprocal are the hardware's v_exp_f32 / v_rcp_f32 (1 ulp): with IEEE divisions, expf and the
        // max the recursion alone was ~55 instructions per step and the kernel VALU-issue-bound at 11 hops.
#pragma unroll
        for (int i = 1; i < HMAX; ++i)
            if (i < n_hops) {
                const float t = group_sum<SL>((l < i) ? __fmul_rn(wl, cl) : 0.f);          // <acc_{i-1}, w_acc>
                const float s = fast_sigmoid(__fadd_rn(__fadd_rn(al, t), bias));
                const float ex = (l < i) ? fast_exp(wl) : ((l == i) ? fast_exp(s) : 0.f);
                const float sum = group_sum<SL>(ex);     // its own statement: a cross-lane reduction inside an arm of ?: runs only
                                                         // in the lanes that take the arm, and a butterfly with idle lanes is wrong
                wl = (l <= i) ? __fmul_rn(ex, __builtin_amdgcn_rcpf(sum)) : 0.f;   // (lanes beyond the first DPP row sum zeros)
            }
        if (live && l < n_hops) {
            if (wout) wout[r * ldw + l] = wl;
            if (aout) aout[r * lda + l] = al;
            if (cout) cout[r * ldc + l] = cl;
        }
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
                const float w = from_lane<LPR>(wl, h);
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][e] = __builtin_fmaf(w, x[h][c][e], acc[c][e]);
            }
    } else {
        float a[HMAX], cc[HMAX], w[HMAX];
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            float sa = 0.f, sc = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sa = __builtin_fmaf(vx[c][e], x[h][c][e], sa);
                    sc = __builtin_fmaf(va[c][e], x[h][c][e], sc);
                }
            a[h] = group_sum<LPR>(sa);
            cc[h] = group_sum<LPR>(sc);
            w[h] = 0.f;
        }
        w[0] = 1.f;
#pragma unroll
        for (int i = 1; i < HMAX; ++i)
            if (i < n_hops) {
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < i; ++j) t = __fadd_rn(t, __fmul_rn(w[j], cc[j]));
                const float s = fast_sigmoid(__fadd_rn(__fadd_rn(a[i], t), bias));
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    w[j] = fast_exp(w[j]);
                    sum += w[j];
                }
                w[i] = fast_exp(s);
                sum += w[i];
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int j = 0; j <= i; ++j) w[j] = __fmul_rn(w[j], inv);
            }
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
                if (live && l == 0) {
                    if (wout) wout[r * ldw + h] = w[h];
                    if (aout) aout[r * lda + h] = a[h];
                    if (cout) cout[r * ldc + h] = cc[h];
                }
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][e] = __builtin_fmaf(w[h], x[h][c][e], acc[c][e]);
            }
    }
    store_row<LPR, CH>(out + r * ldo, acc, l, live, d, dw);
}

// Backward of the recursion above on the per-hop scalars: given A, C [n, H], the bias and G = dL/dW (W = the final weights), one
// thread per row re-runs the H - 1 steps keeping every step's weights, then walks them backwards:
//   step i:  z = [w^{i-1}_0 .. w^{i-1}_{i-1}, s_i],  s_i = sigmoid(a_i + sum_{j<i} w^{i-1}_j c_j + b),  w^i = softmax(z)
//   back  :  gz_k = w^i_k (gw_k - sum_m gw_m w^i_m);  p = gz_i s_i (1 - s_i);  da_i = p;  db += p;
//            dc_j += p w^{i-1}_j,  gw_j <- gz_j + p c_j   (j < i)
// (torch autograd over the same [n, H] recursion is ~140 launches of [n, H]-sized kernels per training step.)
template <int HMAX>
__global__ __launch_bounds__(256) void recursive_scalar_bwd_kernel(const int n_hops, const float *__restrict__ a, const int64_t lda,
                                                                   const float *__restrict__ c, const int64_t ldc, const float bias_arg,
                                                                   const float *__restrict__ bias_ptr, const float *__restrict__ g,
                                                                   const int64_t ldg, float *__restrict__ da, const int64_t ldda,
                                                                   float *__restrict__ dc, const int64_t lddc, float *__restrict__ db,
                                                                   const int64_t n) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float bias = bias_ptr ? *bias_ptr : bias_arg;
    float av[HMAX], cv[HMAX], gw[HMAX], dcv[HMAX], sv[HMAX];
    float hist[HMAX][HMAX];                               // hist[i][k] = w^i_k, k <= i
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
        const bool in = h < n_hops;
        av[h] = in ? a[r * lda + h] : 0.f;
        cv[h] = in ? c[r * ldc + h] : 0.f;
        gw[h] = in ? g[r * ldg + h] : 0.f;
        dcv[h] = 0.f;
        sv[h] = 0.f;
    }
    hist[0][0] = 1.f;
#pragma unroll
    for (int i = 1; i < HMAX; ++i)
        if (i < n_hops) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < i; ++j) t = __fadd_rn(t, __fmul_rn(hist[i - 1][j], cv[j]));
            const float s = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-__fadd_rn(__fadd_rn(av[i], t), bias))));
            sv[i] = s;
            float m = s;
#pragma unroll
            for (int j = 0; j < i; ++j) m = fmaxf(m, hist[i - 1][j]);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < i; ++j) {
                hist[i][j] = expf(hist[i - 1][j] - m);
                sum += hist[i][j];
            }
            hist[i][i] = expf(s - m);
            sum += hist[i][i];
#pragma unroll
            for (int j = 0; j <= i; ++j) hist[i][j] = __fdiv_rn(hist[i][j], sum);
        }
    float dbr = 0.f;
#pragma unroll
    for (int i = HMAX - 1; i >= 1; --i)
        if (i < n_hops) {
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k <= i; ++k) dot += gw[k] * hist[i][k];
            const float p = hist[i][i] * (gw[i] - dot) * sv[i] * (1.f - sv[i]);
            if (da) da[r * ldda + i] = p;
            dbr += p;
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const float gz = hist[i][j] * (gw[j] - dot);
                dcv[j] += p * hist[i - 1][j];
                gw[j] = gz + p * cv[j];
            }
        }
    if (da) da[r * ldda] = 0.f;                           // step 0 is the soft-max of ONE score: a_0 never reaches the weights
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
        if (dc && h < n_hops) dc[r * lddc + h] = dcv[h];
    if (db) db[r] = dbr;
}

// Scores of the 'ori_ref' / 'jk' gates (learnable_weighted_messahe_op.py:73-86) in ONE pass over the hop list:
//   P[n, h - h0] = <X_h[n], v>                 for the adopted hops h in [h0, h1)        (the per-hop part of Linear([ref || x_h]))
//   A[n]         = sum_j <X_j[n], U[j, :]>     over the hops j with bit j of u_mask set  (the shared reference part: the
//                                              reference builds hstack(feat_list) -- or feat_list[0] -- and repeats it H times)
// Every hop row is read once for both; U ([n_hops, ldu], a few KB) is read through the caches.
template <int LPR, int CH, int HMAX>
__global__ __launch_bounds__(256) void hop_rowdot2_reg_kernel(const Hops hx, const int n_hops, const float *__restrict__ u,
                                                              const int64_t ldu, const unsigned long long u_mask,
                                                              const float *__restrict__ vec, const int h0, const int h1,
                                                              float *__restrict__ p, const int64_t ldp, float *__restrict__ a,
                                                              const int64_t n, const int d) {
    constexpr int RPB = 256 / LPR;
    constexpr int SLOTS = LPR * CH;                 // 16-byte slots of a row this layout reaches
    // U is the same for every row: staged in LDS once per block (H x SLOTS vectors, a few KB) instead of H x CH global loads per
    // lane -- those doubled the kernel's L1 accesses and made it the slowest of the row kernels (jk scores at d = 147: 0.48 of
    // peak against 0.69 for the plain row-dot; profiles/r04_agg_pmc.md: TCP_TOTAL_CACHE_ACCESSES 3.8e8 against 1.9e8)
    __shared__ f4 us[HMAX][SLOTS];
    for (int i = threadIdx.x; i < HMAX * SLOTS; i += 256) {
        const int h = i / SLOTS, col = (i - h * SLOTS) * 4;
        us[h][i - h * SLOTS] = (h < n_hops && ((u_mask >> h) & 1ull) && col < d) ? load_masked<4>(u + (int64_t)h * ldu, col, d)
                                                                                  : (f4){0.f, 0.f, 0.f, 0.f};
    }
    const int l = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < n;
    const int64_t r = live ? row : 0;
    f4 x[HMAX][CH], vv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * LPR + l) * 4;
        const bool on = live && col < d;
        vv[c] = (col < d) ? load_masked<4>(vec, col, d) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            x[h][c] = (on && h < n_hops) ? load_masked<4>(hx.p[h] + r * hx.ld[h], col, d) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    float acc[HMAX];
    float shared = 0.f;
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
        acc[h] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[h] = __builtin_fmaf(vv[c][e], x[h][c][e], acc[h]);
        if (h < n_hops && ((u_mask >> h) & 1ull)) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const f4 uv = us[h][c * LPR + l];   // zeros beyond d
#pragma unroll
                for (int e = 0; e < 4; ++e) shared = __builtin_fmaf(uv[e], x[h][c][e], shared);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < HMAX; ++h) acc[h] = group_sum<LPR>(acc[h]);
    shared = group_sum<LPR>(shared);
    if (live && l == 0) {
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h >= h0 && h < h1) p[r * ldp + (h - h0)] = acc[h];
        if (a) a[r] = shared;
    }
}

// out[i,:] = X[idx[i],:].  A thread copies one 16-byte vector of U rows (U index loads, U row loads, U stores: independent).
// The destination is written once and read later by someone else (the transport, the MLP), while the source rows are re-read --
// by every peer that gathers them at the pack step of the need-aware exchange -- so the stores are non-temporal: they do not
// displace the source in L2 / Infinity Cache (pack of the S1 job on 8 ranks: 0.157 -> 0.145 ms at 64 columns, 0.268 -> 0.223 at
// 100; with repeats adjacent 0.141 -> 0.091 ms = 5.1 TB/s written; profiles/r03_pack_order.log).
// With `dst` the copy is out[dst[i],:] = X[idx[i],:] (n_out destination rows): the pack step of the need-aware exchange walks its
// (source row, send-buffer row) pairs in SOURCE order, so a row that several peers gather is read from HBM once and hit in L1 / L2
// for the others.
template <int LPR, int VEC, int U>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ x, const int64_t ldx,
                                                          const int64_t n_rows, const int64_t *__restrict__ idx,
                                                          const int64_t *__restrict__ dst, const int64_t n_out,
                                                          const int64_t n_idx, float *__restrict__ out,
                                                          const int64_t ldo, const int d, const int dz) {
    // d = columns written per row, dz <= d = columns that are DATA: [dz, d) is the destination row's own padding and is written as
    // zeros -- never copied from the source, whose columns beyond dz may be somebody's data (a column view of a wider matrix)
    using V = typename Vt<VEC>::type;
    constexpr int RPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t i0 = (int64_t)blockIdx.x * (RPB * U) + threadIdx.x / LPR;
    int64_t src[U], to[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * RPB;
        int64_t s = i < n_idx ? idx[i] : 0;
        if (s < 0) s += n_rows;                       // python-style negative index
        if (s < 0 || s >= n_rows) __builtin_trap();   // out of range: abort the kernel loudly (like torch's assert)
        src[u] = s;
        to[u] = i;
        if (dst && i < n_idx) {
            to[u] = dst[i];
            if (to[u] < 0 || to[u] >= n_out) __builtin_trap();
        }
    }
    for (int c = l * VEC; c < d; c += LPR * VEC) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c < dz) {
                v[u] = *reinterpret_cast<const V *>(x + src[u] * ldx + c);
                if constexpr (VEC == 4) {
                    if (c + 4 > dz) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c + e >= dz) v[u][e] = 0.f;
                    }
                }
            } else {
                if constexpr (VEC == 4) v[u] = (V){0.f, 0.f, 0.f, 0.f}; else v[u] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + (int64_t)u * RPB;
            if (i < n_idx) __builtin_nontemporal_store(v[u], reinterpret_cast<V *>(out + to[u] * ldo + c));
        }
    }
}

// out_h[i,:] = X_h[idx[i],:] for EVERY hop matrix h in one launch (the training feed of the learnable aggregators,
// models/base_model.py:58-60: the same rows of all K + 1 hop matrices).  A thread owns one 16-byte vector of U rows: it loads the
// U indices ONCE and then, per batch of HB hops, issues HB x U independent row loads before the HB x U stores -- the memory-level
// parallelism of U x HB rows per thread for U index loads and U x HB vector registers, in one launch whose grid does not shrink with H.
template <int LPR, int U, int HB>
__global__ __launch_bounds__(256) void gather_hops_kernel(const Hops hx, const HopsOut ho, const int n_hops, const int64_t n_rows,
                                                          const int64_t *__restrict__ idx, const int64_t n_idx, const int d,
                                                          const int dz) {
    constexpr int RPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t i0 = (int64_t)blockIdx.x * (RPB * U) + threadIdx.x / LPR;
    int64_t src[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * RPB;
        int64_t s = i < n_idx ? idx[i] : 0;
        if (s < 0) s += n_rows;
        if (s < 0 || s >= n_rows) __builtin_trap();
        src[u] = s;
    }
    for (int c = l * 4; c < d; c += LPR * 4) {
        for (int hb = 0; hb < n_hops; hb += HB) {
            f4 v[HB][U];
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const int h = hb + k < n_hops ? hb + k : n_hops - 1;      // (clamped: the surplus loads of the last batch are dropped below)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (c < dz) {
                        v[k][u] = *reinterpret_cast<const f4 *>(hx.p[h] + src[u] * hx.ld[h] + c);
                        if (c + 4 > dz) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c + e >= dz) v[k][u][e] = 0.f;
                        }
                    } else {
                        v[k][u] = (f4){0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                if (hb + k < n_hops) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int64_t i = i0 + (int64_t)u * RPB;
                        if (i < n_idx) __builtin_nontemporal_store(v[k][u], reinterpret_cast<f4 *>(ho.p[hb + k] + i * ho.ld[hb + k] + c));
                    }
                }
            }
        }
    }
}

// The same copy with the hop in blockIdx.y: every (hop, block of rows) is a workgroup of gather_rows_kernel's shape -- U index loads,
// U row loads, U stores per thread -- in ONE grid, so the launch is H times larger than a hop's own (no launch gaps, one ramp) and
// every thread is the short three-round-trip program of the per-hop kernel.  The indices are re-read per hop (n_idx x 8 bytes: L2).
template <int LPR, int U>
__global__ __launch_bounds__(256) void gather_hops_y_kernel(const Hops hx, const HopsOut ho, const int64_t n_rows,
                                                            const int64_t *__restrict__ idx, const int64_t n_idx, const int d,
                                                            const int dz) {
    constexpr int RPB = 256 / LPR;
    const int h = blockIdx.y;
    const float *__restrict__ x = hx.p[h];
    const int64_t ldx = hx.ld[h];
    float *__restrict__ out = ho.p[h];
    const int64_t ldo = ho.ld[h];
    const int l = threadIdx.x % LPR;
    const int64_t i0 = (int64_t)blockIdx.x * (RPB * U) + threadIdx.x / LPR;
    int64_t src[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * RPB;
        int64_t s = i < n_idx ? idx[i] : 0;
        if (s < 0) s += n_rows;
        if (s < 0 || s >= n_rows) __builtin_trap();
        src[u] = s;
    }
    for (int c = l * 4; c < d; c += LPR * 4) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c < dz) {
                v[u] = *reinterpret_cast<const f4 *>(x + src[u] * ldx + c);
                if (c + 4 > dz) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e >= dz) v[u][e] = 0.f;
                }
            } else {
                v[u] = (f4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + (int64_t)u * RPB;
            if (i < n_idx) __builtin_nontemporal_store(v[u], reinterpret_cast<f4 *>(out + i * ldo + c));
        }
    }
}

// out[:, h*d + k] = X_h[:, k]
template <int VEC>
__global__ __launch_bounds__(256) void hop_concat_kernel(const Hops hx, const int n_hops, float *__restrict__ out,
                                                         const int64_t ldo, const int64_t n, const int d) {
    using V = typename Vt<VEC>::type;
    const int dv = d / VEC;
    const int64_t per_hop = n * (int64_t)dv;
    const int64_t total = per_hop * n_hops;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / ((int64_t)dv * n_hops);
        const int rem = (int)(i - row * (int64_t)dv * n_hops);
        const int h = rem / dv;
        const int col = (rem - h * dv) * VEC;
        *reinterpret_cast<V *>(out + row * ldo + (int64_t)h * d + col) =
            *reinterpret_cast<const V *>(hx.p[h] + row * hx.ld[h] + col);
    }
}

// Any d, rows of >= 256 floats: the output row is assembled in LDS.  One block per tile of kConcatTile output floats of one
// row: every ALIGNED 16-byte vector of the source rows that overlaps the tile is loaded exactly once (thread-strided over
// the <= H + 1 segments the tile cuts), its floats are dropped at their (dword-aligned) place in LDS, and after the barrier
// the tile leaves as aligned 16-byte stores.  Every source byte is read once and every output byte written once with full
// vectors on both sides -- the per-thread funnel version below reads each source vector twice through L1.
constexpr int kConcatTile = 1024;
constexpr int kConcatRows = 8;     // rows per block: R independent 16-byte loads in flight per thread (a block with one is latency-bound)
__global__ __launch_bounds__(256) void hop_concat_lds_kernel(const Hops hx, const int n_hops, float *__restrict__ out,
                                                             const int64_t ldo, const int64_t n, const int d,
                                                             const int tiles_per_row, const int width_w) {
    __shared__ float tile[kConcatRows][kConcatTile];
    const int width = d * n_hops;                   // width_w >= width: the columns written (pad columns of the pitch as zeros,
                                                    // see out_cols(): a partly written last line costs a read-modify-write)
    const int64_t rb = blockIdx.x / tiles_per_row;                      // row block
    const int o0 = (int)(blockIdx.x - rb * tiles_per_row) * kConcatTile;
    const int o1 = max(min(o0 + kConcatTile, width), o0);
    const int o1w = min(o0 + kConcatTile, width_w);
    const int64_t row0 = rb * kConcatRows;
    const int rows = (int)min<int64_t>(kConcatRows, n - row0);
    // segments of the tile: hop h covers output floats [max(o0, h d), min(o1, (h+1) d))
    const int h0 = o0 / d, h1 = (o1 > o0) ? (o1 - 1) / d : h0 - 1;
    int done = 0;                                   // aligned source vectors of the previous segments
    const int t = threadIdx.x;
    for (int j = o1 - o0 + t; j < o1w - o0; j += 256) {                  // the pad columns of this tile: zeros
#pragma unroll
        for (int r = 0; r < kConcatRows; ++r) tile[r][j] = 0.f;
    }
    for (int h = h0; h <= h1; ++h) {
        const int k0 = max(o0 - h * d, 0), k1 = min(o1 - h * d, d);       // source floats [k0, k1) of hop h
        const int a0 = k0 & ~3;
        const int nv = (k1 - a0 + 3) >> 2;
        const float *src = hx.p[h] + row0 * hx.ld[h];
        // threads [done, done + nv) modulo 256 take this segment's vectors: consecutive threads, consecutive vectors
        for (int v = (t - done) & 255; v < nv; v += 256) {
            const int a = a0 + v * 4;
            f4 x[kConcatRows];
#pragma unroll
            for (int r = 0; r < kConcatRows; ++r)                        // inside the row's 4-float pitch
                x[r] = (r < rows) ? *reinterpret_cast<const f4 *>(src + r * hx.ld[h] + a) : (f4){0.f, 0.f, 0.f, 0.f};
            const int o = h * d + a - o0;
#pragma unroll
            for (int r = 0; r < kConcatRows; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (a + e >= k0 && a + e < k1) tile[r][o + e] = x[r][e];
        }
        done += nv;
    }
    __syncthreads();
    const int c = t * 4;
#pragma unroll
    for (int r = 0; r < kConcatRows; ++r) {
        if (r >= rows) break;
        float *orow = out + (row0 + r) * ldo + o0;
        if (o0 + c + 4 <= o1w) {
            *reinterpret_cast<f4 *>(orow + c) = *reinterpret_cast<const f4 *>(&tile[r][c]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (o0 + c + e < o1w) orow[c + e] = tile[r][c + e];
        }
    }
}

// Whole rows per block (rows of <= kConcatRowCap floats): a block assembles R = kConcatLds / W complete output rows (W = the columns
// written, pad included) and its 256 threads share the work FLAT -- phase 1 over all (row, hop, aligned source vector) triples,
// phase 2 over all (row, aligned output vector) pairs -- so no thread idles because a 1024-float tile is only partly used (a
// 1617-float row cut into 1024 + 593 keeps 42 % of the second tile's threads idle).  Measured against the tile kernel
// (profiles/r04_concat_rows_vs_tiles.log): d = 147, H = 11 6.94 -> 6.41-6.56 ms, d = 501, H = 5 10.4 -> 10.0-10.3 ms, equal at
// d = 147, H = 6 (882 of 1024), 5 % SLOWER at d = 250, H = 4 where the row is exactly one tile -- so it runs where the tiles would
// be less than 85 % full.  Same loads, same LDS placement, same stores as the tile kernel: bit-identical.
constexpr int kConcatLds = 8192;                    // floats of LDS per block (32 KB)
constexpr int kConcatRowCap = 4096;                 // widest row this kernel takes (then 2 rows per block)
__global__ __launch_bounds__(256) void hop_concat_rows_kernel(const Hops hx, const int n_hops, float *__restrict__ out,
                                                              const int64_t ldo, const int64_t n, const int d, const int width_w,
                                                              const int W, const int R, const int pitch_v) {
    __shared__ float tile[kConcatLds];
    const int width = d * n_hops;
    const int64_t row0 = (int64_t)blockIdx.x * R;
    const int rows = (int)min<int64_t>(R, n - row0);
    const int t = threadIdx.x;
    const int nv = (d + 3) >> 2;                    // aligned 16-byte vectors of one source row
    for (int i = t; i < rows * (W - width); i += 256) {                  // pad columns: zeros
        const int r = i / (W - width);
        tile[r * W + width + (i - r * (W - width))] = 0.f;
    }
    // phase 1.  pitch_v > 0 (every hop has the same pitch of pitch_v vectors, at most a line more than the row): (hop h, row r,
    // vector j of the PITCH) -- the block's rows are contiguous in every hop matrix, so consecutive threads read consecutive
    // addresses ACROSS rows and every wavefront load is one contiguous kilobyte of one hop (the access pattern of the element-wise
    // kernels); the pitch's pad vectors are skipped.  Otherwise (row r, hop h, vector j): consecutive threads take consecutive
    // vectors of one source row.
    const int per_row = pitch_v > 0 ? pitch_v : n_hops * nv;
    const int total = pitch_v > 0 ? n_hops * rows * pitch_v : rows * per_row;
    constexpr int U = 4;                            // independent loads in flight per thread
    for (int base = t; base < total; base += 256 * U) {
        f4 x[U];
        int o[U], k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * 256;
            x[u] = (f4){0.f, 0.f, 0.f, 0.f};
            o[u] = -1;
            k[u] = 0;
            if (i < total) {
                int r, h, j;
                if (pitch_v > 0) {
                    h = i / (rows * pitch_v);
                    const int rem = i - h * (rows * pitch_v);
                    r = rem / pitch_v;
                    j = rem - r * pitch_v;
                } else {
                    r = i / per_row;
                    const int rem = i - r * per_row;
                    h = rem / nv;
                    j = rem - h * nv;
                }
                if (j < nv) {
                    x[u] = *reinterpret_cast<const f4 *>(hx.p[h] + (row0 + r) * hx.ld[h] + 4 * j);   // inside the row's 4-float pitch
                    o[u] = r * W + h * d + 4 * j;
                    k[u] = d - 4 * j;               // floats of this vector that belong to the row (>= 1)
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (o[u] >= 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < k[u]) tile[o[u] + e] = x[u][e];
            }
    }
    __syncthreads();
    // phase 2: (row r, output vector v): aligned 16-byte LDS reads and global stores
    const int wv = (width_w + 3) >> 2;
    for (int i = t; i < rows * wv; i += 256) {
        const int r = i / wv;
        const int c = (i - r * wv) * 4;
        float *orow = out + (row0 + r) * ldo;
        if (c + 4 <= width_w) {
            *reinterpret_cast<f4 *>(orow + c) = *reinterpret_cast<const f4 *>(&tile[r * W + c]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < width_w) orow[c + e] = tile[r * W + c + e];
        }
    }
}

// The same for ANY d (d % 4 != 0: the hops' segments start at arbitrary 4-byte offsets of the output row): one thread per
// ALIGNED 16-byte vector of the output row; its four floats are consecutive in one source row except where the vector
// straddles a hop boundary, so they are fetched with one dword-aligned 16-byte load (legal on gfx950: vector memory
// accesses only need dword alignment) or, at the boundaries, element by element.

__global__ __launch_bounds__(256) void hop_concat_any_kernel(const Hops hx, const int n_hops, float *__restrict__ out,
                                                             const int64_t ldo, const int64_t n, const int d, const int width_w) {
    // d >= 4.  One thread per aligned 16-byte vector of the output row.  Its floats sit at a dword-aligned position of
    // ONE source row, except in the vector that straddles a hop boundary (H per row, i.e. in nearly every wavefront:
    // that path must stay cheap -- no divisions, no element-wise recomputation of the hop index).
    const int width = d * n_hops;
    const int vecs = (width_w + 3) / 4;             // width_w >= width: pad columns of the pitch are written as zeros (out_cols())
    const int64_t total = n * (int64_t)vecs;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / vecs;
        const int c = (int)(i - row * vecs) * 4;
        if (c >= width) {                           // a vector of padding
            *reinterpret_cast<f4 *>(out + row * ldo + c) = (f4){0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const int h = c / d;
        const int k = c - h * d;
        const int rem = d - k;                      // floats of this vector that belong to hop h (>= 1)
        // two ALIGNED 16-byte loads around the source position and a funnel select (a misaligned 16-byte access is split
        // by the hardware: measured 0.35 of peak)
        const float *src = hx.p[h] + row * hx.ld[h];
        const int sft = k & 3;
        const int a = k - sft;
        const f4 v0 = *reinterpret_cast<const f4 *>(src + a);
        f4 r = v0;
        if (sft) {
            f4 v1 = (f4){0.f, 0.f, 0.f, 0.f};
            if (a + 8 <= hx.ld[h]) v1 = *reinterpret_cast<const f4 *>(src + a + 4);
            r[0] = (sft == 1) ? v0[1] : (sft == 2) ? v0[2] : v0[3];
            r[1] = (sft == 1) ? v0[2] : (sft == 2) ? v0[3] : v1[0];
            r[2] = (sft == 1) ? v0[3] : (sft == 2) ? v1[0] : v1[1];
            r[3] = (sft == 1) ? v1[0] : (sft == 2) ? v1[1] : v1[2];
        }
        float *op = out + row * ldo + c;
        if (rem >= 4) {
            *reinterpret_cast<f4 *>(op) = r;
        } else {
            if (h + 1 < n_hops) {                   // the tail of the vector is the head of the next hop's row
                const float *nx = hx.p[h + 1] + row * hx.ld[h + 1];
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (e >= rem) r[e] = nx[e - rem];
                *reinterpret_cast<f4 *>(op) = r;
            } else if (c + 4 <= width_w) {          // last vector of the row, its tail is padding: zeros
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (e >= rem) r[e] = 0.f;
                *reinterpret_cast<f4 *>(op) = r;
            } else {                                // last vector of the row: only the floats inside the row
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < rem) op[e] = r[e];
            }
        }
    }
}

// dw[h] = sum_{n,k} dOut[n,k] X_h[n,k]: per-block partials, then a sequential (deterministic) second level
constexpr int kW1dBlocks = 8192;   // a grid-stride stream wants >= 8k blocks on this part (profiles/r02_stream_patterns.log)

__global__ __launch_bounds__(256) void hop_dot_partial_kernel(const Hops hx, const int n_hops, const float *__restrict__ g,
                                                              const int64_t ldg, float *__restrict__ scratch,
                                                              const int64_t n, const int d) {
    __shared__ float red[4];
    const int64_t total = n * (int64_t)d;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int h = 0; h < n_hops; ++h) {
        float acc = 0.f;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int64_t row = i / d;
            const int col = (int)(i - row * d);
            acc = __builtin_fmaf(g[row * ldg + col], hx.p[h][row * hx.ld[h] + col], acc);
        }
        acc = group_sum<64>(acc);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) scratch[(int64_t)blockIdx.x * n_hops + h] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// Single pass for 16-byte rows (any d: padded rows, masked tail): every block walks a contiguous range of (row, vector)
// positions, dOut's vector is loaded once and multiplied into all H hop vectors (H + 1 streams instead of the 2 H of the
// scalar kernel above, which re-reads dOut per hop), per-thread partial sums for all hops live in registers.
// GU: dOut's rows are only dword-aligned (autograd's dense gradient, d % 4 != 0) -> dword-aligned vector loads.
template <int HMAX, bool GU>
__global__ __launch_bounds__(256) void hop_dot_partial_vec_kernel(const Hops hx, const int n_hops, const float *__restrict__ g,
                                                                  const int64_t ldg, float *__restrict__ scratch,
                                                                  const int64_t n, const int d, const int64_t per_block) {
    __shared__ float red[4][HMAX];
    const int dv = (d + 3) >> 2;
    const int64_t total = n * (int64_t)dv;
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = min(lo + per_block, total);
    float acc[HMAX];
#pragma unroll
    for (int h = 0; h < HMAX; ++h) acc[h] = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int64_t row = i / dv;
        const int col = (int)(i - row * dv) * 4;
        f4 gv;
        if constexpr (GU) {
            const float *gp = g + row * ldg + col;
            if (col + 4 <= d) {
                const f4u t = *reinterpret_cast<const f4u *>(gp);
                gv = (f4){t[0], t[1], t[2], t[3]};
            } else {
                gv = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    if (col + e < d) gv[e] = gp[e];
            }
        } else {
            gv = load_masked<4, true>(g + row * ldg, col, d);
        }
        f4 xv[HMAX];
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) xv[h] = load_masked<4, true>(hx.p[h] + row * hx.ld[h], col, d);
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
            if (h < n_hops) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[h] = __builtin_fmaf(gv[e], xv[h][e], acc[h]);
            }
    }
#pragma unroll
    for (int h = 0; h < HMAX; ++h) acc[h] = group_sum<64>(acc[h]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int h = 0; h < HMAX; ++h) red[threadIdx.x >> 6][h] = acc[h];
    }
    __syncthreads();
    if (threadIdx.x < n_hops)
        scratch[(int64_t)blockIdx.x * n_hops + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// second level: one wavefront per hop, every lane adds its strided share of the block partials in block order, then a
// fixed butterfly -- deterministic for a given block count
__global__ __launch_bounds__(64) void hop_dot_final_kernel(const float *__restrict__ scratch, const int n_blocks, const int n_hops,
                                                           float *__restrict__ dw) {
    const int h = blockIdx.x;
    float acc = 0.f;
    for (int b = threadIdx.x; b < n_blocks; b += 64) acc += scratch[(int64_t)b * n_hops + h];
    acc = group_sum<64>(acc);
    if (threadIdx.x == 0) dw[h] = acc;
}

// Column sums of the hops weighted per row:  out[h, :] = sum_n W[n, h] * X_h[n, :]  (sw = 1) or sum_n W[n] * X_h[n, :] (sw = 0).
// This is the weight gradient of every row-dot -- dv = X_h^T g[:, h] of the gate / jk / ori_ref scores -- which torch evaluates
// as one transposed GEMV per hop: 0.46 ms EACH for a [50 000, 147] batch slice on its padded pitch (rocBLAS gemvn: 64 GB/s),
// 5.5 of the 8.4 ms of a GAMLP training step (profiles/r04_train_step.log).  Here: one streaming pass over all hops, 16 bytes per
// lane, per-thread partial rows in registers, a fixed two-level reduction (row lanes through LDS in lane order, then the blocks in
// block order): deterministic, no atomics.
template <int HMAX>
__global__ __launch_bounds__(256) void hop_colsum_partial_kernel(const Hops hx, const int n_hops, const float *__restrict__ w,
                                                                 const int64_t ldw, const int sw, float *__restrict__ scratch,
                                                                 const int64_t n, const int d, const int slots,
                                                                 const int64_t per_block) {
    __shared__ f4 red[256];
    const int t = threadIdx.x;
    const int rl_count = 256 / slots;               // rows a block takes per iteration
    const int slot = t % slots, rl = t / slots;
    const bool lane_on = rl < rl_count;
    const int64_t r0 = (int64_t)blockIdx.x * per_block, r1 = min(r0 + per_block, n);
    f4 acc[HMAX];
#pragma unroll
    for (int h = 0; h < HMAX; ++h) acc[h] = (f4){0.f, 0.f, 0.f, 0.f};
    if (lane_on) {
#pragma unroll 1
        for (int64_t r = r0 + rl; r < r1; r += rl_count) {
#pragma unroll
            for (int hb = 0; hb < HMAX; hb += 4) {      // four hop loads in flight at a time: the accumulators are what stays
                f4 x[4];
                float wv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (hb + q < n_hops) {
                        x[q] = load_masked<4, true>(hx.p[hb + q] + r * hx.ld[hb + q], slot * 4, d);
                        wv[q] = w[r * ldw + (int64_t)(hb + q) * sw];
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (hb + q < n_hops) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[hb + q][e] = __builtin_fmaf(wv[q], x[q][e], acc[hb + q][e]);
                    }
            }
        }
    }
    const int dp = slots * 4;
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
        if (h < n_hops) {
            red[t] = acc[h];
            __syncthreads();
            if (rl == 0) {
                f4 s = red[slot];
                for (int q = 1; q < rl_count; ++q) {
                    const f4 o = red[q * slots + slot];
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += o[e];
                }
                *reinterpret_cast<f4 *>(scratch + ((int64_t)blockIdx.x * n_hops + h) * dp + slot * 4) = s;
            }
            __syncthreads();
        }
    }
}

// second level: one wavefront per (hop, 16-byte column slot): lane l adds the partials of blocks l, l + 64, ... in that order, a fixed
// butterfly folds the 64 lanes -- deterministic for a given block count (one thread per column walking all ~1000 partials one
// after the other took three times as long as the streaming first level)
__global__ __launch_bounds__(64) void hop_colsum_final_kernel(const float *__restrict__ scratch, const int n_blocks, const int n_hops,
                                                              const int slots, float *__restrict__ out, const int64_t ldo, const int d) {
    const int h = blockIdx.x / slots, slot = blockIdx.x - h * slots;
    const int dp = slots * 4;
    f4 s = (f4){0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < n_blocks; b += 64) {
        const f4 o = *reinterpret_cast<const f4 *>(scratch + ((int64_t)b * n_hops + h) * dp + slot * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += o[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = group_sum<64>(s[e]);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (slot * 4 + e < d) out[(int64_t)h * ldo + slot * 4 + e] = s[e];
    }
}

bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <typename H>
bool vec4_rows(const H &hx, int n_hops) {   // every hop matrix has 16-byte aligned rows (pitch % 4 == 0)
    for (int h = 0; h < n_hops; ++h)
        if (hx.ld[h] % 4 != 0 || !aligned_to(hx.p[h], 16)) return false;
    return true;
}

int fill_hops(Hops &hx, int n_hops, const float *const *h_x, const int64_t *h_ldx, int64_t d, bool &vec4) {
    if (n_hops < 1 || n_hops > SGL_MAX_HOPS) return sgl::fail(SGL_ERR_INVALID, "n_hops=%d outside [1,%d]", n_hops, SGL_MAX_HOPS);
    if (!h_x) return sgl::fail(SGL_ERR_INVALID, "NULL hop pointer array");
    for (int h = 0; h < n_hops; ++h) {
        hx.p[h] = h_x[h];
        hx.ld[h] = h_ldx ? h_ldx[h] : d;
        if (!hx.p[h]) return sgl::fail(SGL_ERR_INVALID, "hop %d: NULL pointer", h);
        if (hx.ld[h] < d) return sgl::fail(SGL_ERR_INVALID, "hop %d: leading dimension %lld < d", h, (long long)hx.ld[h]);
        if (!aligned_to(hx.p[h], 4)) return sgl::fail(SGL_ERR_INVALID, "hop %d: pointer not 4-byte aligned", h);
        if (hx.ld[h] % 4 != 0 || !aligned_to(hx.p[h], 16)) vec4 = false;
    }
    for (int h = n_hops; h < SGL_MAX_HOPS; ++h) {
        hx.p[h] = nullptr;
        hx.ld[h] = 0;
    }
    return SGL_OK;
}

int stream_grid(int64_t total_threads) {
    // memory-bound elementwise: one 16-byte element per thread.  Measured on MI355X (products shape, 5 streams): a
    // 2048-block grid-stride launch reaches 5.3 TB/s, one element per thread 5.9 TB/s (torch's add: 6.0); the
    // grid-stride loop only remains as the overflow path for > 2^22 blocks.
    int64_t blocks = (total_threads + 255) / 256;
    const int64_t cap = sgl::tuning("agg_blocks", 0) > 0 ? sgl::tuning("agg_blocks", 0) : ((int64_t)1 << 22);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int pick_lpr(int64_t d, int vec) {
    const int64_t lanes = (d + vec - 1) / vec;
    int lpr = 8;
    while (lpr < lanes && lpr < 64) lpr <<= 1;
    return lpr;
}

// Columns a row-producing kernel writes: the d data columns plus the `pad` columns after them that the CALLER declared to be the
// row's own padding (the *_padded_f32 entry points; sgl_amd.device passes the tail of the pitch of the outputs it allocates) --
// written as zeros, so that every line of the row is written whole.  The kernels never guess: with pad = 0 nothing beyond column
// d is touched.  `room` = the columns the lane layout reaches.
int out_cols(int64_t d, int64_t pad, int64_t room) {
    int64_t dw = sgl::tuning("row_whole_lines", 1) != 0 ? d + pad : d;
    if (dw > room) dw = room > d ? room / 4 * 4 : d;
    if (dw > d && dw % 4 != 0) dw = d;              // (validated by the entry points: d + pad is a whole number of vectors)
    return (int)dw;
}

int check_pad(const char *who, int64_t width, int64_t pad, int64_t ldo) {
    if (pad < 0 || width + pad > ldo) return sgl::fail(SGL_ERR_INVALID, "%s: pad_cols=%lld does not fit the output pitch", who, (long long)pad);
    if (pad > 0 && ((width + pad) % 4 != 0 || ldo % 4 != 0))
        return sgl::fail(SGL_ERR_INVALID, "%s: padded rows must be whole 16-byte vectors (width + pad_cols and ldo multiples of 4)", who);
    return SGL_OK;
}

RowLayout pick_row_layout(int64_t d, int n_hops, bool allow_8x5 = false) {
    RowLayout r;
    r.lpr = pick_lpr(d, 4);
    r.ch = (d > r.lpr * 4) ? 2 : 1;
    if (r.lpr == 64 && r.ch == 1 && d > 128 && sgl::tuning("row_lpr32x2", 1) != 0) {   // 2 rows per wavefront
        r.lpr = 32;
        r.ch = 2;
    }
    if (sgl::tuning("row_narrow_groups", 1) != 0 && d <= r.lpr * 4 * r.ch) {
        const int slots = (int)((d + 3) / 4);
        static const int cand[2][3] = {{16, 3, 12}, {8, 5, 6}};       // lanes, chunks, most hops instantiated
        const int64_t mode = sgl::tuning("row_narrow_groups", 1);       // 1: both candidates, 2: 16 x 3 only, 3: 8 x 5 only (measurements)
        for (const auto &c : cand) {
            const int idle = c[0] * c[1] - slots;
            if (((mode == 2 || !allow_8x5) && c[0] == 8) || (mode == 3 && c[0] == 16)) continue;
            if (idle >= 0 && n_hops <= c[2] && 2 * idle <= r.lpr * r.ch - slots) {
                r.lpr = c[0];
                r.ch = c[1];
            }
        }
    }
    return r;
}

// KH(L, C): all even hop counts up to 16; KH12 / KH6: the narrow-group layouts, instantiated up to 12 / 6 hop vectors
#define SGL_ROWREG_DISPATCH(KH, KH12, KH6, lay)                        \
    do {                                                               \
        if ((lay).lpr == 8 && (lay).ch == 5) KH6(8, 5);                \
        else if ((lay).lpr == 16 && (lay).ch == 3) KH12(16, 3);        \
        else if ((lay).lpr == 32 && (lay).ch == 2) KH(32, 2);          \
        else if ((lay).ch == 2) KH(64, 2);                             \
        else if ((lay).lpr == 8) KH(8, 1);                             \
        else if ((lay).lpr == 16) KH(16, 1);                           \
        else if ((lay).lpr == 32) KH(32, 1);                           \
        else KH(64, 1);                                                \
    } while (0)
#define SGL_HOPS_UP_TO_16(K, L, C)                 \
    do {                                           \
        if (n_hops <= 2) K(L, C, 2);               \
        else if (n_hops <= 4) K(L, C, 4);          \
        else if (n_hops <= 6) K(L, C, 6);          \
        else if (n_hops <= 8) K(L, C, 8);          \
        else if (n_hops <= 10) K(L, C, 10);        \
        else if (n_hops <= 12) K(L, C, 12);        \
        else if (n_hops <= 14) K(L, C, 14);        \
        else K(L, C, 16);                          \
    } while (0)
#define SGL_HOPS_UP_TO_12(K, L, C)                 \
    do {                                           \
        if (n_hops <= 2) K(L, C, 2);               \
        else if (n_hops <= 4) K(L, C, 4);          \
        else if (n_hops <= 6) K(L, C, 6);          \
        else if (n_hops <= 8) K(L, C, 8);          \
        else if (n_hops <= 10) K(L, C, 10);        \
        else K(L, C, 12);                          \
    } while (0)
#define SGL_HOPS_UP_TO_6(K, L, C)                  \
    do {                                           \
        if (n_hops <= 2) K(L, C, 2);               \
        else if (n_hops <= 4) K(L, C, 4);          \
        else K(L, C, 6);                           \
    } while (0)

#define SGL_LAUNCH_CHECK(what)                                                                                  \
    do {                                                                                                        \
        hipError_t _e = hipGetLastError();                                                                      \
        if (_e != hipSuccess) return sgl::fail((int)_e, what ": kernel launch failed: %s", hipGetErrorString(_e)); \
    } while (0)

}  // namespace

SGL_EXPORT int sgl_hop_reduce_f32(int op, int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_w,
                                  float *d_out, int64_t ldo, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(op >= SGL_REDUCE_SUM && op <= SGL_REDUCE_WSUM, "sgl_hop_reduce_f32: unknown op %d", op);
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_reduce_f32: bad sizes");
    SGL_REQUIRE(op != SGL_REDUCE_WSUM || d_w, "sgl_hop_reduce_f32: WSUM needs device weights");
    Hops hx;
    bool vec4 = (d % 4 == 0) && (ldo % 4 == 0) && aligned_to(d_out, 16);
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    if (n == 0 || d == 0) return SGL_OK;
    SGL_REQUIRE(d_out && ldo >= d, "sgl_hop_reduce_f32: bad output");
    hipStream_t st = sgl::as_stream(stream);
    const int vec = vec4 ? 4 : 1;
    const int grid = stream_grid(n * (d / vec));
#define SGL_RED(OP)                                                                                               \
    if (vec4)                                                                                                     \
        hipLaunchKernelGGL((hop_reduce_kernel<OP, 4>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_w, d_out, ldo, n, (int)d); \
    else                                                                                                          \
        hipLaunchKernelGGL((hop_reduce_kernel<OP, 1>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_w, d_out, ldo, n, (int)d);
    switch (op) {
        case SGL_REDUCE_SUM: SGL_RED(SGL_REDUCE_SUM) break;
        case SGL_REDUCE_MEAN: SGL_RED(SGL_REDUCE_MEAN) break;
        case SGL_REDUCE_MAX: SGL_RED(SGL_REDUCE_MAX) break;
        case SGL_REDUCE_MIN: SGL_RED(SGL_REDUCE_MIN) break;
        default: SGL_RED(SGL_REDUCE_WSUM) break;
    }
#undef SGL_RED
    SGL_LAUNCH_CHECK("sgl_hop_reduce_f32");
    return SGL_OK;
}

static int wsum2d_impl(bool fma, int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_w, int64_t ldw,
                       float *d_out, int64_t ldo, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_wsum2d_f32: bad sizes");
    Hops hx;
    bool vec4 = (d % 4 == 0) && (ldo % 4 == 0) && aligned_to(d_out, 16);
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    if (n == 0 || d == 0) return SGL_OK;
    SGL_REQUIRE(d_w && ldw >= n_hops, "sgl_hop_wsum2d_f32: bad weights");
    SGL_REQUIRE(d_out && ldo >= d, "sgl_hop_wsum2d_f32: bad output");
    hipStream_t st = sgl::as_stream(stream);
    const int grid = stream_grid(n * (d / (vec4 ? 4 : 1)));
    if (vec4 && fma)
        hipLaunchKernelGGL((hop_wsum2d_kernel<4, true>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_w, ldw, d_out, ldo, n, (int)d);
    else if (vec4)
        hipLaunchKernelGGL((hop_wsum2d_kernel<4, false>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_w, ldw, d_out, ldo, n, (int)d);
    else if (fma)
        hipLaunchKernelGGL((hop_wsum2d_kernel<1, true>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_w, ldw, d_out, ldo, n, (int)d);
    else
        hipLaunchKernelGGL((hop_wsum2d_kernel<1, false>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_w, ldw, d_out, ldo, n, (int)d);
    SGL_LAUNCH_CHECK("sgl_hop_wsum2d_f32");
    return SGL_OK;
}

SGL_EXPORT int sgl_hop_wsum2d_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_w,
                                  int64_t ldw, float *d_out, int64_t ldo, int64_t n, int64_t d, void *stream) {
    return wsum2d_impl(true, n_hops, h_x, h_ldx, d_w, ldw, d_out, ldo, n, d, stream);
}

template <int VEC>
static void launch_rowdot(int lpr, int g_unaligned, hipStream_t st, const Hops &hx, int n_hops,
                          const float *g, int64_t ldg, float *dw, int64_t lddw, int64_t n, int d) {
    if constexpr (VEC == 4) {
        // the row's H hop vectors fit in registers: one load of dOut, all loads in flight, interleaved butterflies
        const RowLayout lay = pick_row_layout(d, n_hops);
        if (n_hops <= 16 && d <= lay.lpr * 4 * lay.ch) {
            const unsigned grid = (unsigned)((n + (256 / lay.lpr) - 1) / (256 / lay.lpr));
#define SGL_RR(L, C, HM)                                                                                                        \
    do {                                                                                                                       \
        if (g_unaligned)                                                                                                       \
            hipLaunchKernelGGL((hop_rowdot_reg_kernel<L, C, HM, true>), dim3(grid), dim3(256), 0, st, hx, n_hops, g, ldg, dw, lddw, n, d); \
        else                                                                                                                   \
            hipLaunchKernelGGL((hop_rowdot_reg_kernel<L, C, HM, false>), dim3(grid), dim3(256), 0, st, hx, n_hops, g, ldg, dw, lddw, n, d); \
    } while (0)
#define SGL_RR_H(L, C) SGL_HOPS_UP_TO_16(SGL_RR, L, C)
#define SGL_RR_H12(L, C) SGL_HOPS_UP_TO_12(SGL_RR, L, C)
#define SGL_RR_H6(L, C) (void)0          /* 8 x 5 is never chosen for this kernel (pick_row_layout) */
            SGL_ROWREG_DISPATCH(SGL_RR_H, SGL_RR_H12, SGL_RR_H6, lay);
#undef SGL_RR_H6
#undef SGL_RR_H12
#undef SGL_RR_H
#undef SGL_RR
            return;
        }
    }
#define SGL_RD(L)                                                                                              \
    hipLaunchKernelGGL((hop_rowdot_kernel<L, VEC>), dim3((unsigned)((n + (256 / L) - 1) / (256 / L))), dim3(256), 0, st, \
                       hx, n_hops, g, ldg, dw, lddw, n, d)
    switch (lpr) {
        case 8: SGL_RD(8); break;
        case 16: SGL_RD(16); break;
        case 32: SGL_RD(32); break;
        default: SGL_RD(64); break;
    }
#undef SGL_RD
}

SGL_EXPORT int sgl_hop_wsum2d_bwd_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_w,
                                      int64_t ldw, const float *d_dout, int64_t lddo, float *d_dw, int64_t lddw,
                                      float *const *h_dx, const int64_t *h_lddx, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_wsum2d_bwd_f32: bad sizes");
    Hops hx;
    bool hops4 = true;                                         // 16-byte row accesses possible (any d, masked tail)
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, hops4);
    if (rc != SGL_OK) return rc;
    const bool g4 = (lddo % 4 == 0) && aligned_to(d_dout, 16);
    // a dword-aligned dOut (autograd's dense gradient, d % 4 != 0) still takes the 16-byte path for the H hop reads when the
    // register-resident kernel applies: it reads dOut with dword-aligned vector loads
    const bool g_unaligned = hops4 && !g4 && n_hops <= 16 && d <= 512;
    const bool row4 = hops4 && (g4 || g_unaligned);
    bool vec4 = hops4 && g4 && (d % 4 == 0);                   // element-wise kernels need whole vectors
    if (n == 0 || d == 0) return SGL_OK;
    SGL_REQUIRE(d_dout && lddo >= d, "sgl_hop_wsum2d_bwd_f32: bad dOut");
    hipStream_t st = sgl::as_stream(stream);
    if (d_dw) {
        SGL_REQUIRE(lddw >= n_hops, "sgl_hop_wsum2d_bwd_f32: lddw < n_hops");
        const int lpr = pick_lpr(d, row4 ? 4 : 1);
        SGL_REQUIRE((n + (256 / lpr) - 1) / (256 / lpr) < INT32_MAX, "sgl_hop_wsum2d_bwd_f32: too many rows");
        if (row4)
            launch_rowdot<4>(lpr, g_unaligned ? 1 : 0, st, hx, n_hops, d_dout, lddo, d_dw, lddw, n, (int)d);
        else
            launch_rowdot<1>(lpr, 0, st, hx, n_hops, d_dout, lddo, d_dw, lddw, n, (int)d);
        SGL_LAUNCH_CHECK("sgl_hop_wsum2d_bwd_f32(dW)");
    }
    if (h_dx) {
        SGL_REQUIRE(d_w && ldw >= n_hops, "sgl_hop_wsum2d_bwd_f32: dX needs W");
        HopsOut dx;
        bool any = false, v4 = vec4;
        for (int h = 0; h < SGL_MAX_HOPS; ++h) {
            dx.p[h] = (h < n_hops) ? h_dx[h] : nullptr;
            dx.ld[h] = (h < n_hops) ? (h_lddx ? h_lddx[h] : d) : 0;
            if (dx.p[h]) {
                any = true;
                if (dx.ld[h] < d) return sgl::fail(SGL_ERR_INVALID, "sgl_hop_wsum2d_bwd_f32: dX ld < d");
                if (dx.ld[h] % 4 != 0 || !aligned_to(dx.p[h], 16)) v4 = false;
            }
        }
        if (any) {
            const int grid = stream_grid(n * (d / (v4 ? 4 : 1)));
            if (v4)
                hipLaunchKernelGGL((hop_wsum2d_dx_kernel<4>), dim3(grid), dim3(256), 0, st, dx, n_hops, d_w, ldw, d_dout, lddo, n, (int)d);
            else
                hipLaunchKernelGGL((hop_wsum2d_dx_kernel<1>), dim3(grid), dim3(256), 0, st, dx, n_hops, d_w, ldw, d_dout, lddo, n, (int)d);
            SGL_LAUNCH_CHECK("sgl_hop_wsum2d_bwd_f32(dX)");
        }
    }
    return SGL_OK;
}

// out[n, h] = <X_h[n, :], v>  : the gate scores of LearnableWeightedMessageOp (Linear(d -> 1) applied to every hop,
// learnable_weighted_messahe_op.py:69-71,74-86) for all hops in one pass -- the row-dot kernel with a row-invariant
// second operand (leading dimension 0).
SGL_EXPORT int sgl_hop_rowdot_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_vec,
                                  float *d_out, int64_t ldo, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_rowdot_f32: bad sizes");
    Hops hx;
    bool row4 = aligned_to(d_vec, 16);
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, row4);
    if (rc != SGL_OK) return rc;
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_vec && d_out && ldo >= n_hops, "sgl_hop_rowdot_f32: bad arguments");
    if (!sgl::launch_fits((n + 3) / 4, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_rowdot_f32: too many rows for one launch (shard the matrix)");
    hipStream_t st = sgl::as_stream(stream);
    if (d == 0) {
        SGL_HIP_CHECK(hipMemset2DAsync(d_out, ldo * sizeof(float), 0, n_hops * sizeof(float), n, st));
        return SGL_OK;
    }
    // the vector is read with the same 16-byte lanes as the rows: it must be readable up to round_up(d, 4) floats
    // (callers pass a zero-padded copy), the masked tail ignores what lies beyond d
    const int lpr = pick_lpr(d, row4 ? 4 : 1);
    SGL_REQUIRE((n + (256 / lpr) - 1) / (256 / lpr) < INT32_MAX, "sgl_hop_rowdot_f32: too many rows");
    if (row4)
        launch_rowdot<4>(lpr, 0, st, hx, n_hops, d_vec, 0, d_out, ldo, n, (int)d);
    else
        launch_rowdot<1>(lpr, 0, st, hx, n_hops, d_vec, 0, d_out, ldo, n, (int)d);
    SGL_LAUNCH_CHECK("sgl_hop_rowdot_f32");
    return SGL_OK;
}

SGL_EXPORT int64_t sgl_hop_wsum1d_bwd_scratch(int n_hops) { return (int64_t)kW1dBlocks * (n_hops > 0 ? n_hops : 1); }

SGL_EXPORT int sgl_hop_wsum1d_bwd_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_dout,
                                      int64_t lddo, float *d_dw, float *d_scratch, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_wsum1d_bwd_f32: bad sizes");
    Hops hx;
    bool vec4 = false;
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    SGL_REQUIRE(d_dw && d_scratch, "sgl_hop_wsum1d_bwd_f32: NULL output/scratch");
    hipStream_t st = sgl::as_stream(stream);
    if (n == 0 || d == 0) {
        SGL_HIP_CHECK(hipMemsetAsync(d_dw, 0, sizeof(float) * n_hops, st));
        return SGL_OK;
    }
    SGL_REQUIRE(d_dout && lddo >= d, "sgl_hop_wsum1d_bwd_f32: bad dOut");
    int blocks;
    const bool g4 = (lddo % 4 == 0) && aligned_to(d_dout, 16);
    if (vec4_rows(hx, n_hops) && n_hops <= 16) {   // the per-thread partials of all hops live in registers
        const int64_t total = n * ((d + 3) / 4);
        blocks = (int)std::min<int64_t>(kW1dBlocks, (total + 255) / 256);
        const int64_t per_block = ((total + blocks - 1) / blocks + 255) / 256 * 256;
        blocks = (int)((total + per_block - 1) / per_block);
#define SGL_DP(HM)                                                                                                             \
    do {                                                                                                                       \
        if (g4)                                                                                                                \
            hipLaunchKernelGGL((hop_dot_partial_vec_kernel<HM, false>), dim3(blocks), dim3(256), 0, st, hx, n_hops, d_dout, lddo, \
                               d_scratch, n, (int)d, per_block);                                                              \
        else                                                                                                                   \
            hipLaunchKernelGGL((hop_dot_partial_vec_kernel<HM, true>), dim3(blocks), dim3(256), 0, st, hx, n_hops, d_dout, lddo, \
                               d_scratch, n, (int)d, per_block);                                                              \
    } while (0)
        if (n_hops <= 4) SGL_DP(4);
        else if (n_hops <= 8) SGL_DP(8);
        else if (n_hops <= 12) SGL_DP(12);
        else SGL_DP(16);
#undef SGL_DP
    } else {
        blocks = (int)std::min<int64_t>(kW1dBlocks, (n * d + 255) / 256);
        hipLaunchKernelGGL(hop_dot_partial_kernel, dim3(blocks), dim3(256), 0, st, hx, n_hops, d_dout, lddo, d_scratch, n, (int)d);
    }
    SGL_LAUNCH_CHECK("sgl_hop_wsum1d_bwd_f32(partial)");
    hipLaunchKernelGGL(hop_dot_final_kernel, dim3(n_hops), dim3(64), 0, st, d_scratch, blocks, n_hops, d_dw);
    SGL_LAUNCH_CHECK("sgl_hop_wsum1d_bwd_f32(final)");
    return SGL_OK;
}

SGL_EXPORT int sgl_hop_select_bwd_f32(int op, int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_gout,
                                     int64_t ldg, float *const *h_dx, const int64_t *h_lddx, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(op == SGL_REDUCE_MAX || op == SGL_REDUCE_MIN, "sgl_hop_select_bwd_f32: op must be SGL_REDUCE_MAX or SGL_REDUCE_MIN");
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_select_bwd_f32: bad sizes");
    Hops hx;
    bool vec4 = (d % 4 == 0) && (ldg % 4 == 0) && aligned_to(d_gout, 16);
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    SGL_REQUIRE(h_dx && h_lddx, "sgl_hop_select_bwd_f32: NULL gradient arrays");
    HopsOut dx;
    for (int h = 0; h < SGL_MAX_HOPS; ++h) {
        dx.p[h] = h < n_hops ? h_dx[h] : nullptr;
        dx.ld[h] = h < n_hops ? h_lddx[h] : 0;
        if (dx.p[h]) {
            SGL_REQUIRE(dx.ld[h] >= d && aligned_to(dx.p[h], 4), "sgl_hop_select_bwd_f32: bad gradient buffer %d", h);
            if (dx.ld[h] % 4 != 0 || !aligned_to(dx.p[h], 16)) vec4 = false;
        }
    }
    if (n == 0 || d == 0) return SGL_OK;
    SGL_REQUIRE(d_gout && ldg >= d, "sgl_hop_select_bwd_f32: bad incoming gradient");
    hipStream_t st = sgl::as_stream(stream);
    const int grid = stream_grid(n * (d / (vec4 ? 4 : 1)));
    if (op == SGL_REDUCE_MAX) {
        if (vec4) hipLaunchKernelGGL((hop_select_bwd_kernel<true, 4>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_gout, ldg, dx, n, (int)d);
        else hipLaunchKernelGGL((hop_select_bwd_kernel<true, 1>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_gout, ldg, dx, n, (int)d);
    } else {
        if (vec4) hipLaunchKernelGGL((hop_select_bwd_kernel<false, 4>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_gout, ldg, dx, n, (int)d);
        else hipLaunchKernelGGL((hop_select_bwd_kernel<false, 1>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_gout, ldg, dx, n, (int)d);
    }
    SGL_LAUNCH_CHECK("sgl_hop_select_bwd_f32");
    return SGL_OK;
}

SGL_EXPORT int sgl_hop_lincomb_f32(int n_in, const float *const *h_x, const int64_t *h_ldx, int n_out, float *const *h_out,
                                   const int64_t *h_ldo, const float *d_w, int64_t ldw, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_lincomb_f32: bad sizes");
    SGL_REQUIRE(n_in >= 1 && n_in <= 16, "sgl_hop_lincomb_f32: between 1 and 16 input matrices (n_in=%d)", n_in);
    SGL_REQUIRE(n_out >= 1 && n_out <= SGL_MAX_HOPS, "sgl_hop_lincomb_f32: between 1 and %d outputs", SGL_MAX_HOPS);
    SGL_REQUIRE(d_w && ldw >= n_in && ldw < INT32_MAX, "sgl_hop_lincomb_f32: the [n_out, ldw] weight matrix is required (ldw >= n_in)");
    SGL_REQUIRE(h_out && h_ldo, "sgl_hop_lincomb_f32: NULL output arrays");
    Hops hx;
    bool vec4 = (d % 4 == 0);
    int rc = fill_hops(hx, n_in, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    HopsOut outs;
    for (int k = 0; k < SGL_MAX_HOPS; ++k) {
        outs.p[k] = k < n_out ? h_out[k] : nullptr;
        outs.ld[k] = k < n_out ? h_ldo[k] : 0;
        if (k < n_out) {
            SGL_REQUIRE(outs.p[k] && outs.ld[k] >= d && aligned_to(outs.p[k], 4), "sgl_hop_lincomb_f32: bad output %d", k);
            if (outs.ld[k] % 4 != 0 || !aligned_to(outs.p[k], 16)) vec4 = false;
            for (int j = 0; j < n_in; ++j)
                SGL_REQUIRE(outs.p[k] != hx.p[j], "sgl_hop_lincomb_f32: output %d aliases input %d", k, j);
        }
    }
    if (n == 0 || d == 0) return SGL_OK;
    hipStream_t st = sgl::as_stream(stream);
    const int grid = stream_grid(n * (d / (vec4 ? 4 : 1)));
#define SGL_LC(NI)                                                                                                                 \
    do {                                                                                                                           \
        if (vec4) hipLaunchKernelGGL((hop_lincomb_kernel<NI, 4>), dim3(grid), dim3(256), 0, st, hx, n_in, outs, n_out, d_w, (int)ldw, n, (int)d); \
        else hipLaunchKernelGGL((hop_lincomb_kernel<NI, 1>), dim3(grid), dim3(256), 0, st, hx, n_in, outs, n_out, d_w, (int)ldw, n, (int)d);      \
    } while (0)
    if (n_in <= 4) SGL_LC(4);
    else if (n_in <= 8) SGL_LC(8);
    else if (n_in <= 12) SGL_LC(12);
    else SGL_LC(16);
#undef SGL_LC
    SGL_LAUNCH_CHECK("sgl_hop_lincomb_f32");
    return SGL_OK;
}

SGL_EXPORT int sgl_hop_concat_padded_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, float *d_out, int64_t ldo,
                                         int64_t pad_cols, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_concat_f32: bad sizes");
    {
        const int prc = check_pad("sgl_hop_concat_padded_f32", d * (n_hops > 0 ? n_hops : 0), pad_cols, ldo);
        if (prc != SGL_OK) return prc;
    }
    Hops hx;
    bool vec4 = (d % 4 == 0) && (ldo % 4 == 0) && aligned_to(d_out, 16);
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    if (n == 0 || d == 0) return SGL_OK;
    SGL_REQUIRE(d_out && ldo >= d * n_hops, "sgl_hop_concat_f32: bad output");
    hipStream_t st = sgl::as_stream(stream);
    const bool out16 = (ldo % 4 == 0) && aligned_to(d_out, 16) && d * n_hops < INT32_MAX;
    if (vec4) {
        const int grid = stream_grid(n * (d / 4) * n_hops);
        hipLaunchKernelGGL((hop_concat_kernel<4>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_out, ldo, n, (int)d);
    } else if (out16 && d >= 4 && vec4_rows(hx, n_hops) && d * n_hops >= 256 && out_cols(d * n_hops, pad_cols, INT32_MAX) <= kConcatRowCap &&
               (sgl::tuning("concat_lds", 1) == 3 ||
                (sgl::tuning("concat_lds", 1) == 1 &&      // where the 1024-float tiles of the kernel below would be < 85 % full
                 100 * (int64_t)out_cols(d * n_hops, pad_cols, INT32_MAX) <
                     85 * (int64_t)kConcatTile * ((out_cols(d * n_hops, pad_cols, INT32_MAX) + kConcatTile - 1) / kConcatTile)))) {
        // any d, rows of 256 ... 4096 floats: whole rows assembled in LDS, the block's threads share the work flat
        const int width_w = out_cols(d * n_hops, pad_cols, INT32_MAX);
        const int W = (width_w + 3) / 4 * 4;
        const int R = kConcatLds / W;
        const int64_t blocks = (n + R - 1) / R;
        if (!sgl::launch_fits(blocks, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_concat_f32: too many rows for one launch (shard the matrix)");
        // every hop on the same pitch, at most one line longer than the row: read whole pitches, contiguously across the block's rows
        int pitch_v = 0;
        if (sgl::tuning("concat_flat_read", 1) != 0 && n > 1) {
            pitch_v = (int)(hx.ld[0] / 4);
            for (int h = 0; h < n_hops; ++h)
                if (hx.ld[h] != hx.ld[0] || hx.ld[h] % 4 != 0 || hx.ld[h] >= d + 32) pitch_v = 0;
        }
        hipLaunchKernelGGL(hop_concat_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, hx, n_hops, d_out, ldo, n, (int)d, width_w, W, R, pitch_v);
    } else if (out16 && d >= 4 && vec4_rows(hx, n_hops) && d * n_hops >= 256 && sgl::tuning("concat_lds", 1) != 0 &&
               sgl::launch_fits((n + kConcatRows - 1) / kConcatRows * ((out_cols(d * n_hops, pad_cols, INT32_MAX) + kConcatTile - 1) / kConcatTile), 256)) {
        // any d, long rows: assembled in LDS, every source vector read once
        const int width_w = out_cols(d * n_hops, pad_cols, INT32_MAX);
        const int tiles = (int)((width_w + kConcatTile - 1) / kConcatTile);
        hipLaunchKernelGGL(hop_concat_lds_kernel, dim3((unsigned)((n + kConcatRows - 1) / kConcatRows * tiles)), dim3(256), 0, st, hx,
                           n_hops, d_out, ldo, n, (int)d, tiles, width_w);
    } else if (out16 && d >= 4 && vec4_rows(hx, n_hops)) {   // any d: aligned 16-byte stores, aligned 16-byte loads + select
        const int width_w = out_cols(d * n_hops, pad_cols, INT32_MAX);
        const int grid = stream_grid(n * (((int64_t)width_w + 3) / 4));
        hipLaunchKernelGGL(hop_concat_any_kernel, dim3(grid), dim3(256), 0, st, hx, n_hops, d_out, ldo, n, (int)d, width_w);
    } else {
        const int grid = stream_grid(n * d * n_hops);
        hipLaunchKernelGGL((hop_concat_kernel<1>), dim3(grid), dim3(256), 0, st, hx, n_hops, d_out, ldo, n, (int)d);
    }
    SGL_LAUNCH_CHECK("sgl_hop_concat_f32");
    return SGL_OK;
}

SGL_EXPORT int sgl_hop_concat_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, float *d_out, int64_t ldo,
                                  int64_t n, int64_t d, void *stream) {
    return sgl_hop_concat_padded_f32(n_hops, h_x, h_ldx, d_out, ldo, 0, n, d, stream);
}

SGL_EXPORT int sgl_nafs_padded_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, float *d_out, int64_t ldo,
                                   int64_t pad_cols, float *d_w_out, int64_t ldw, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_nafs_f32: bad sizes");
    {
        const int prc = check_pad("sgl_nafs_padded_f32", d, d_out ? pad_cols : 0, d_out ? ldo : d);
        if (prc != SGL_OK) return prc;
    }
    SGL_REQUIRE(d_w_out && ldw >= n_hops, "sgl_nafs_f32: the [n, n_hops] weight buffer is required");
    Hops hx;
    bool vec4 = true;   // 16-byte row accesses with a masked tail: needs only 4-float row pitches
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    if (n == 0 || d == 0) return SGL_OK;
    hipStream_t st = sgl::as_stream(stream);
    const int lpr = pick_lpr(d, vec4 ? 4 : 1);
    const int64_t blocks = (n + (256 / lpr) - 1) / (256 / lpr);
    if (!sgl::launch_fits(blocks, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "row-wise kernel: too many rows for one launch (shard the matrix)");
    SGL_REQUIRE(blocks < INT32_MAX, "sgl_nafs_f32: too many rows");
    // single-pass kernel: the H hop rows of a node fit in registers (H <= 16, d <= 512, 16-byte lanes)
    const bool out_vec4 = d_out && (ldo % 4 == 0) && aligned_to(d_out, 16);
    if (vec4 && out_vec4 && n_hops <= 16 && d <= 512 && sgl::tuning("nafs_fused", 1) != 0) {
        const RowLayout lay = pick_row_layout(d, n_hops);
        const int64_t nblocks = (n + (256 / lay.lpr) - 1) / (256 / lay.lpr);
#define SGL_NF(L, C, HM) \
    hipLaunchKernelGGL((nafs_fused_kernel<L, C, HM>), dim3((unsigned)nblocks), dim3(256), 0, st, hx, n_hops, d_out, ldo, d_w_out, ldw, n, (int)d, out_cols(d, pad_cols, (L) * (C) * 4))
#define SGL_NF_H(L, C) SGL_HOPS_UP_TO_16(SGL_NF, L, C)
#define SGL_NF_H12(L, C) SGL_HOPS_UP_TO_12(SGL_NF, L, C)
#define SGL_NF_H6(L, C) (void)0          /* 8 x 5 is never chosen for this kernel (pick_row_layout) */
        SGL_ROWREG_DISPATCH(SGL_NF_H, SGL_NF_H12, SGL_NF_H6, lay);
#undef SGL_NF_H6
#undef SGL_NF_H12
#undef SGL_NF_H
#undef SGL_NF
        SGL_LAUNCH_CHECK("sgl_nafs_f32(fused)");
        return SGL_OK;
    }
#define SGL_NW(L, V) \
    hipLaunchKernelGGL((nafs_weight_kernel<L, V>), dim3((unsigned)blocks), dim3(256), 0, st, hx, n_hops, d_w_out, ldw, n, (int)d)
    if (vec4) {
        switch (lpr) {
            case 8: SGL_NW(8, 4); break;
            case 16: SGL_NW(16, 4); break;
            case 32: SGL_NW(32, 4); break;
            default: SGL_NW(64, 4); break;
        }
    } else {
        switch (lpr) {
            case 8: SGL_NW(8, 1); break;
            case 16: SGL_NW(16, 1); break;
            case 32: SGL_NW(32, 1); break;
            default: SGL_NW(64, 1); break;
        }
    }
#undef SGL_NW
    SGL_LAUNCH_CHECK("sgl_nafs_f32(weights)");
    if (!d_out) return SGL_OK;  // weights only
    // out = sum_h W[:,h] * X_h accumulated in hop order from 0 with rounded products (over_smooth_distance_op.py:27-31)
    return wsum2d_impl(false, n_hops, h_x, h_ldx, d_w_out, ldw, d_out, ldo, n, d, stream);
}

SGL_EXPORT int sgl_nafs_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, float *d_out, int64_t ldo,
                            float *d_w_out, int64_t ldw, int64_t n, int64_t d, void *stream) {
    return sgl_nafs_padded_f32(n_hops, h_x, h_ldx, d_out, ldo, 0, d_w_out, ldw, n, d, stream);
}

SGL_EXPORT int sgl_nafs_prefix_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, uint64_t emit_mask,
                                   float *const *h_out, const int64_t *h_ldo, int64_t pad_cols, int combine, float divisor,
                                   int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d <= 512, "sgl_nafs_prefix_f32: rows of up to 512 floats (d=%lld)", (long long)d);
    SGL_REQUIRE(combine >= 0 && combine <= 3, "sgl_nafs_prefix_f32: combine must be 0 (store), 1 (add), 2 (add, divide) or 3 (max)");
    SGL_REQUIRE(combine != 2 || divisor != 0.f, "sgl_nafs_prefix_f32: zero divisor");
    Hops hx;
    bool vec4 = true;
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    SGL_REQUIRE(vec4, "sgl_nafs_prefix_f32: hop rows must be 16-byte aligned with pitches that are multiples of 4 floats");
    SGL_REQUIRE(emit_mask != 0 && (n_hops == 64 || (emit_mask >> n_hops) == 0), "sgl_nafs_prefix_f32: emit_mask must name hops below n_hops");
    SGL_REQUIRE(h_out && h_ldo, "sgl_nafs_prefix_f32: NULL output arrays");
    HopsOut outs;
    const int n_out = __builtin_popcountll(emit_mask);
    for (int k = 0; k < SGL_MAX_HOPS; ++k) {
        outs.p[k] = k < n_out ? h_out[k] : nullptr;
        outs.ld[k] = k < n_out ? h_ldo[k] : 0;
        if (k < n_out) {
            SGL_REQUIRE(outs.p[k] && outs.ld[k] >= d && outs.ld[k] % 4 == 0 && aligned_to(outs.p[k], 16),
                        "sgl_nafs_prefix_f32: output %d must be 16-byte aligned with a pitch >= d that is a multiple of 4 floats", k);
            const int prc = check_pad("sgl_nafs_prefix_f32", d, pad_cols, outs.ld[k]);
            if (prc != SGL_OK) return prc;
        }
    }
    if (n == 0 || d == 0) return SGL_OK;
    hipStream_t st = sgl::as_stream(stream);
    const RowLayout lay = pick_row_layout(d, 1);
    const int64_t nblocks = (n + (256 / lay.lpr) - 1) / (256 / lay.lpr);
    if (!sgl::launch_fits(nblocks, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "row-wise kernel: too many rows for one launch (shard the matrix)");
#define SGL_NP(L, C) \
    hipLaunchKernelGGL((nafs_prefix_kernel<L, C>), dim3((unsigned)nblocks), dim3(256), 0, st, hx, n_hops, emit_mask, outs, combine, divisor, n, (int)d, out_cols(d, pad_cols, (L) * (C) * 4))
#define SGL_NP_NONE(L, C) (void)0
    SGL_ROWREG_DISPATCH(SGL_NP, SGL_NP, SGL_NP_NONE, lay);
#undef SGL_NP_NONE
#undef SGL_NP
    SGL_LAUNCH_CHECK("sgl_nafs_prefix_f32");
    return SGL_OK;
}

static int copy_rows(const char *who, const float *d_x, int64_t ldx, int64_t n_rows, const int64_t *d_idx, const int64_t *d_dst,
                     int64_t n_out, int64_t n_idx, float *d_out, int64_t ldo, int64_t dz, int64_t pad, void *stream) {
    SGL_REQUIRE(n_idx >= 0 && dz >= 0 && pad >= 0 && dz + pad < INT32_MAX && n_rows >= 0 && n_out >= 0, "%s: bad sizes", who);
    if (n_idx == 0 || dz == 0) return SGL_OK;
    const int64_t d = dz + pad;                     // columns written: the data, then the destination's own padding as zeros
    SGL_REQUIRE(d_x && d_idx && d_out && ldx >= dz && ldo >= d, "%s: bad arguments", who);
    SGL_REQUIRE(pad == 0 || (d % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && aligned_to(d_x, 16) && aligned_to(d_out, 16)),
                "%s: padded rows need 16-byte aligned rows, pitches that are multiples of 4 floats and d + pad_cols a multiple of 4", who);
    const bool vec4 = (d % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned_to(d_x, 16) && aligned_to(d_out, 16);
    hipStream_t st = sgl::as_stream(stream);
    // Lanes per row: the group size that leaves the fewest lane slots idle (a row of 40 vectors -- d = 147 on its 160-float pitch --
    // on 64 lanes idles 24 of them in every instruction; on 8 lanes x 5 iterations none, and a wavefront then has 8 rows = 8
    // independent sets of lines in flight); ties go to the wider group (fewer iterations per row).
    int lpr = 64;
    {
        const int64_t nv = (d + (vec4 ? 3 : 0)) / (vec4 ? 4 : 1);
        int64_t best = -1;
        for (int cand : {64, 32, 16, 8}) {
            const int64_t waste = (nv + cand - 1) / cand * cand - nv;
            if (best < 0 || waste < best) {
                best = waste;
                lpr = cand;
            }
        }
        if (sgl::tuning("gather_lpr", 0) > 0) lpr = (int)sgl::tuning("gather_lpr", 0);
    }
    // Rows per thread: a copy is three dependent memory round trips (index, row, store); a launch that needs several ROUNDS of
    // resident workgroups pays them once per round, which is what a small batch is made of (200 000 rows: 46 us against 26 us
    // for a contiguous copy of the same bytes, profiles/r05_aggregators.log).  U rows per thread -- all index loads, then all row
    // loads, then all stores -- so that the grid is about one or two rounds of the chip (<= 4 096 workgroups), up to 8 (70 VGPRs, 7 waves per SIMD).
    const int rpb = 256 / lpr;
    int u = 1;
    while (u < 8 && (n_idx + (int64_t)rpb * u - 1) / ((int64_t)rpb * u) > 4096) u *= 2;   // (16 rows: 135 VGPRs, 3 waves per SIMD)
    if (sgl::tuning("gather_rows_per_thread", 0) > 0) u = (int)sgl::tuning("gather_rows_per_thread", 0);
    if (u != 1 && u != 2 && u != 4 && u != 8 && u != 16) u = 4;
    const int64_t blocks = (n_idx + (int64_t)rpb * u - 1) / ((int64_t)rpb * u);
    if (!sgl::launch_fits(blocks, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "%s: too many indices for one launch", who);
    SGL_REQUIRE(blocks < INT32_MAX, "%s: too many rows", who);
#define SGL_GRU(L, V, UU)                                                                                                         \
    hipLaunchKernelGGL((gather_rows_kernel<L, V, UU>), dim3((unsigned)blocks), dim3(256), 0, st, d_x, ldx, n_rows, d_idx, d_dst, n_out, \
                       n_idx, d_out, ldo, (int)d, (int)dz)
#define SGL_GR(L, V)                                                                                                             \
    do {                                                                                                                         \
        switch (u) {                                                                                                             \
            case 16: SGL_GRU(L, V, 16); break;                                                                                   \
            case 8: SGL_GRU(L, V, 8); break;                                                                                     \
            case 4: SGL_GRU(L, V, 4); break;                                                                                     \
            case 2: SGL_GRU(L, V, 2); break;                                                                                     \
            default: SGL_GRU(L, V, 1); break;                                                                                    \
        }                                                                                                                        \
    } while (0)
    if (vec4) {
        switch (lpr) {
            case 8: SGL_GR(8, 4); break;
            case 16: SGL_GR(16, 4); break;
            case 32: SGL_GR(32, 4); break;
            default: SGL_GR(64, 4); break;
        }
    } else {
        switch (lpr) {
            case 8: SGL_GR(8, 1); break;
            case 16: SGL_GR(16, 1); break;
            case 32: SGL_GR(32, 1); break;
            default: SGL_GR(64, 1); break;
        }
    }
#undef SGL_GR
#undef SGL_GRU
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sgl::fail((int)e, "%s: kernel launch failed: %s", who, hipGetErrorString(e));
    return SGL_OK;
}

SGL_EXPORT int sgl_gather_rows_f32(const float *d_x, int64_t ldx, int64_t n_rows, const int64_t *d_idx, int64_t n_idx,
                                   float *d_out, int64_t ldo, int64_t d, void *stream) {
    return copy_rows("sgl_gather_rows_f32", d_x, ldx, n_rows, d_idx, nullptr, n_idx, n_idx, d_out, ldo, d, 0, stream);
}

// the same with the destination row's own padding declared: columns [d, d + pad_cols) of every output row are written as ZEROS (whole
// 16-byte vectors, so that every line of the row is written whole); nothing beyond column d of the source is ever read into the result
SGL_EXPORT int sgl_gather_rows_padded_f32(const float *d_x, int64_t ldx, int64_t n_rows, const int64_t *d_idx, int64_t n_idx,
                                          float *d_out, int64_t ldo, int64_t d, int64_t pad_cols, void *stream) {
    return copy_rows("sgl_gather_rows_padded_f32", d_x, ldx, n_rows, d_idx, nullptr, n_idx, n_idx, d_out, ldo, d, pad_cols, stream);
}

// The same rows of EVERY hop matrix in one launch (gather_hops_kernel): out_h = X_h[idx] for h < n_hops, all X_h with n_rows rows,
// all outputs with n_idx rows; columns [d, d + pad_cols) of the output rows are written as zeros (the destination's own padding).
// Needs 16-byte aligned rows and pitches that are multiples of 4 floats on both sides (else SGL_ERR_UNSUPPORTED: gather hop by hop).
SGL_EXPORT int sgl_gather_hops_padded_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, int64_t n_rows, const int64_t *d_idx,
                                          int64_t n_idx, float *const *h_out, const int64_t *h_ldo, int64_t d, int64_t pad_cols, void *stream) {
    SGL_REQUIRE(n_idx >= 0 && d >= 0 && pad_cols >= 0 && d + pad_cols < INT32_MAX && n_rows >= 0, "sgl_gather_hops_padded_f32: bad sizes");
    if (n_idx == 0 || d == 0) return SGL_OK;
    SGL_REQUIRE(d_idx && h_out && h_ldo, "sgl_gather_hops_padded_f32: NULL arguments");
    Hops hx;
    bool vec4 = true;
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    const int64_t dw = d + pad_cols;
    HopsOut ho;
    for (int h = 0; h < SGL_MAX_HOPS; ++h) {
        ho.p[h] = h < n_hops ? h_out[h] : nullptr;
        ho.ld[h] = h < n_hops ? h_ldo[h] : 0;
        if (h < n_hops) {
            SGL_REQUIRE(ho.p[h] && ho.ld[h] >= dw, "sgl_gather_hops_padded_f32: output %d: NULL or pitch < d + pad_cols", h);
            if (ho.ld[h] % 4 != 0 || !aligned_to(ho.p[h], 16)) vec4 = false;
        }
    }
    if (!vec4 || dw % 4 != 0) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_gather_hops_padded_f32: rows are not 16-byte vectors (gather hop by hop)");
    hipStream_t st = sgl::as_stream(stream);
    const int64_t nv = dw / 4;
    int lpr = 64;
    {
        int64_t best = -1;
        for (int cand : {64, 32, 16, 8}) {
            const int64_t waste = (nv + cand - 1) / cand * cand - nv;
            if (best < 0 || waste < best) {
                best = waste;
                lpr = cand;
            }
        }
        if (sgl::tuning("gather_lpr", 0) > 0) lpr = (int)sgl::tuning("gather_lpr", 0);
    }
    const int rpb = 256 / lpr;
    if (sgl::tuning("gather_hops_grid", 1) != 0 && n_hops <= 65535) {
        // hop in blockIdx.y (gather_hops_y_kernel), the default.  Per launch with ten launches queued, 200 000 rows (profiles/
        // r06_gather_hops.log): d = 100, H = 4: 0.122 ms (0.65 of 8 TB/s; hop loop 0.124), d = 147, H = 6: 0.281 (0.63; 0.303),
        // d = 128, H = 11: 0.384 (0.73; 0.403) -- one row per thread, two for rows shorter than four lines (d = 100: 0.129 -> 0.122)
        int uy = (dw * 4 < 512) ? 2 : 1;
        if (sgl::tuning("gather_rows_per_thread", 0) > 0) uy = (int)sgl::tuning("gather_rows_per_thread", 0);
        if (uy != 1 && uy != 2 && uy != 4 && uy != 8) uy = 4;
        const int64_t by = (n_idx + (int64_t)rpb * uy - 1) / ((int64_t)rpb * uy);
        if (!sgl::launch_fits(by * n_hops, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_gather_hops_padded_f32: too many indices for one launch");
#define SGL_GY(L, UU) hipLaunchKernelGGL((gather_hops_y_kernel<L, UU>), dim3((unsigned)by, (unsigned)n_hops), dim3(256), 0, st, hx, ho, n_rows, d_idx, n_idx, (int)dw, (int)d)
#define SGL_GYU(L)                     \
    do {                               \
        if (uy == 1) SGL_GY(L, 1);     \
        else if (uy == 2) SGL_GY(L, 2); \
        else if (uy == 8) SGL_GY(L, 8); \
        else SGL_GY(L, 4);             \
    } while (0)
        switch (lpr) {
            case 8: SGL_GYU(8); break;
            case 16: SGL_GYU(16); break;
            case 32: SGL_GYU(32); break;
            default: SGL_GYU(64); break;
        }
#undef SGL_GYU
#undef SGL_GY
        hipError_t ey = hipGetLastError();
        if (ey != hipSuccess) return sgl::fail((int)ey, "sgl_gather_hops_padded_f32: kernel launch failed: %s", hipGetErrorString(ey));
        return SGL_OK;
    }
    // gather_hops_grid = 0: the first form, a hop loop inside the thread (3-8 % slower than the grid form above, kept for comparison).
    // one row per thread: with HB = 4 hops per batch a thread already keeps 4 independent row loads in flight, and the grid stays
    // n_idx / rows-per-block workgroups whatever H is (measured, profiles/r06_gather_hops.log: 1 row 0.153 ms, 2 rows 0.158, 4 rows
    // 0.164 for 200 000 rows of 4 hops at d = 100; hop by hop 0.162)
    int u = (int)sgl::tuning("gather_rows_per_thread", 0);
    if (u != 1 && u != 2 && u != 4) u = 1;
    const int64_t blocks = (n_idx + (int64_t)rpb * u - 1) / ((int64_t)rpb * u);
    if (!sgl::launch_fits(blocks, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_gather_hops_padded_f32: too many indices for one launch");
#define SGL_GH(L, UU) hipLaunchKernelGGL((gather_hops_kernel<L, UU, 4>), dim3((unsigned)blocks), dim3(256), 0, st, hx, ho, n_hops, n_rows, d_idx, n_idx, (int)dw, (int)d)
#define SGL_GHU(L)                   \
    do {                             \
        if (u == 1) SGL_GH(L, 1);    \
        else if (u == 4) SGL_GH(L, 4); \
        else SGL_GH(L, 2);           \
    } while (0)
    switch (lpr) {
        case 8: SGL_GHU(8); break;
        case 16: SGL_GHU(16); break;
        case 32: SGL_GHU(32); break;
        default: SGL_GHU(64); break;
    }
#undef SGL_GHU
#undef SGL_GH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sgl::fail((int)e, "sgl_gather_hops_padded_f32: kernel launch failed: %s", hipGetErrorString(e));
    return SGL_OK;
}

// out[dst[i], :] = X[src[i], :], i < n_idx (dst entries distinct; every index is range-checked in the kernel, which traps on a
// bad one).  The pack step of the need-aware exchange: the (own row, send-buffer row) pairs sorted by own row.
SGL_EXPORT int sgl_scatter_rows_f32(const float *d_x, int64_t ldx, int64_t n_rows, const int64_t *d_src, const int64_t *d_dst,
                                    int64_t n_idx, float *d_out, int64_t ldo, int64_t n_out_rows, int64_t d, void *stream) {
    SGL_REQUIRE(n_idx == 0 || d_dst != nullptr, "sgl_scatter_rows_f32: NULL destination index");
    return copy_rows("sgl_scatter_rows_f32", d_x, ldx, n_rows, d_src, d_dst, n_out_rows, n_idx, d_out, ldo, d, 0, stream);
}

// ---- per-column content signature ----------------------------------------------------------------------------------------------
// sig[c] = sum over rows r of mix(bits(X[r, c]), r)  (64-bit wrapping sum: order-free, so any parallel schedule gives the same value).
// What GraphOp.propagate compares between two calls over one adjacency: the product is separable by columns, so only the columns
// whose signature moved need new hops (the label-reuse loop of node_classification_with_label_use.py:88-104 rewrites the last C of
// d + C columns between its preprocess() calls).  One streaming read of X.
__device__ __forceinline__ unsigned long long sig_mix(unsigned int bits, unsigned long long r) {
    unsigned long long h = ((unsigned long long)bits ^ (r * 0x9E3779B97F4A7C15ull)) * 0xBF58476D1CE4E5B9ull;
    h ^= h >> 31;
    h *= 0x94D049BB133111EBull;
    return h ^ (h >> 29);
}

__global__ __launch_bounds__(256) void col_signature_kernel(const float *__restrict__ x, const int64_t ld, const int64_t n, const int nv,
                                                            const int nvp_log2, const int64_t rows_per_block,
                                                            unsigned long long *__restrict__ out) {
    // lane group: nvp = 2^nvp_log2 >= nv vector columns x rstep = 256 / nvp rows per step; four row steps in flight per thread; the
    // block folds its row groups in LDS and adds ONE value per column to the result (the first version's 256 atomics per block on
    // 148 addresses took 2.9 ms at [2.45 M, 147] -- the stream itself is 0.3 ms)
    __shared__ unsigned long long red[4][256];
    const int nvp = 1 << nvp_log2;
    const int v = threadIdx.x & (nvp - 1);
    const int rsub = threadIdx.x >> nvp_log2, rstep = 256 >> nvp_log2;
    const bool active = v < nv;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    unsigned long long s[4] = {0, 0, 0, 0};
    if (active) {
        for (int64_t r = r0 + rsub; r < r1; r += 4 * (int64_t)rstep) {
            f4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t ru = r + (int64_t)u * rstep;
                q[u] = (f4){0.f, 0.f, 0.f, 0.f};
                if (ru < r1) q[u] = *reinterpret_cast<const f4 *>(x + ru * ld + v * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t ru = r + (int64_t)u * rstep;
                if (ru < r1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] += sig_mix(__float_as_uint(q[u][k]), (unsigned long long)ru);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    if (active && rsub == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long t = s[k];
            for (int g = 1; g < rstep; ++g) t += red[k][v + g * nvp];
            atomicAdd(out + v * 4 + k, t);   // integer adds: the result does not depend on their order
        }
    }
}

// d_sig[0 .. round_up(d, 4)) <- signatures of the columns of X [n, >= round_up(d, 4)] (16-byte aligned rows, ld % 4 == 0: the pitch
// alloc_rows gives; the up-to-3 pad columns behind d are signed with the rest)
SGL_EXPORT int sgl_col_signature_f32(const float *d_x, int64_t ldx, int64_t n, int64_t d, uint64_t *d_sig, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_col_signature_f32: bad sizes");
    const int64_t dw = (d + 3) / 4 * 4;
    if (dw == 0) return SGL_OK;
    SGL_REQUIRE(d_sig != nullptr, "sgl_col_signature_f32: NULL output");
    hipStream_t st = sgl::as_stream(stream);
    SGL_HIP_CHECK(hipMemsetAsync(d_sig, 0, (size_t)dw * sizeof(uint64_t), st));
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_x && ldx >= dw && ldx % 4 == 0 && aligned_to(d_x, 16),
                "sgl_col_signature_f32: rows must be 16-byte aligned with a pitch that is a multiple of 4 floats and covers round_up(d, 4)");
    for (int64_t c0 = 0; c0 < dw; c0 += 1024) {                       // 256 vector columns per launch
        const int nv = (int)(std::min<int64_t>(1024, dw - c0) / 4);
        int lg = 0;
        while ((1 << lg) < nv) ++lg;
        const int rstep = 256 >> lg;
        const int64_t rows_per_block = std::max<int64_t>((int64_t)rstep * 16, (n + 2047) / 2048);   // <= 2 048 blocks: few atomics per address
        const int64_t blocks = (n + rows_per_block - 1) / rows_per_block;
        hipLaunchKernelGGL(col_signature_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_x + c0, ldx, n, nv, lg, rows_per_block,
                           reinterpret_cast<unsigned long long *>(d_sig) + c0);
    }
    SGL_LAUNCH_CHECK("sgl_col_signature_f32");
    return SGL_OK;
}

// ---- learnable gates -------------------------------------------------------------------------------------------------------
// register-resident row kernels: H <= 16, d <= 512, 16-byte aligned rows.  Anything else -> SGL_ERR_UNSUPPORTED and the caller
// takes the two-pass route (sgl_hop_rowdot_f32 + sgl_hop_wsum2d_f32).
static int hop_gate_impl(bool device_bias_sentinel, int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_vec,
                         float bias, float *d_out, int64_t ldo, int64_t pad_cols, float *d_w_out, int64_t ldw, float *d_g_out,
                         int64_t ldg, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_gate_f32: bad sizes");
    {
        const int prc = check_pad("sgl_hop_gate_padded_f32", d, pad_cols, ldo);
        if (prc != SGL_OK) return prc;
    }
    Hops hx;
    bool vec4 = aligned_to(d_vec, 16);
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    if (n == 0 || d == 0) return SGL_OK;
    SGL_REQUIRE(d_vec && d_out && ldo >= d, "sgl_hop_gate_f32: bad arguments");
    SGL_REQUIRE((!d_w_out || ldw >= n_hops) && (!d_g_out || ldg >= n_hops), "sgl_hop_gate_f32: bad weight / gate buffers");
    if (!(vec4 && n_hops <= 16 && d <= 512 && ldo % 4 == 0 && aligned_to(d_out, 16)))
        return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_gate_f32: needs <= 16 hops, d <= 512 and 16-byte aligned rows (use the two-pass route)");
    hipStream_t st = sgl::as_stream(stream);
    const RowLayout lay = pick_row_layout(d, n_hops);
    const int64_t blocks = (n + (256 / lay.lpr) - 1) / (256 / lay.lpr);
    if (!sgl::launch_fits(blocks, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_gate_f32: too many rows for one launch (shard the matrix)");
    // bias = NaN: the bias is the float that follows the padded vector on the device (d_vec[round_up(d, 4)]) -- a caller whose
    // bias is a device tensor (a torch parameter) needs neither a device-to-host synchronisation nor a new value per launch
    // Only the *_padded entry point reads it that way: a caller of the un-suffixed sgl_hop_gate_f32 whose d_vec holds exactly
    // round_up(d, 4) floats and whose (diverged) bias is NaN gets NaN outputs, as the scalar semantics say, not a read past its vector.
    const float *bias_ptr = (device_bias_sentinel && bias != bias) ? d_vec + (d + 3) / 4 * 4 : nullptr;
#define SGL_GF(L, C, HM) \
    hipLaunchKernelGGL((gate_fused_kernel<L, C, HM>), dim3((unsigned)blocks), dim3(256), 0, st, hx, n_hops, d_vec, bias, bias_ptr, d_out, ldo, d_w_out, ldw, d_g_out, ldg, n, (int)d, out_cols(d, pad_cols, (L) * (C) * 4))
#define SGL_GF_H(L, C) SGL_HOPS_UP_TO_16(SGL_GF, L, C)
#define SGL_GF_H12(L, C) SGL_HOPS_UP_TO_12(SGL_GF, L, C)
#define SGL_GF_H6(L, C) (void)0          /* 8 x 5 is never chosen for this kernel (pick_row_layout) */
    SGL_ROWREG_DISPATCH(SGL_GF_H, SGL_GF_H12, SGL_GF_H6, lay);
#undef SGL_GF_H6
#undef SGL_GF_H12
#undef SGL_GF_H
#undef SGL_GF
    SGL_LAUNCH_CHECK("sgl_hop_gate_f32");
    return SGL_OK;
}

SGL_EXPORT int sgl_hop_gate_padded_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_vec, float bias,
                                       float *d_out, int64_t ldo, int64_t pad_cols, float *d_w_out, int64_t ldw, float *d_g_out,
                                       int64_t ldg, int64_t n, int64_t d, void *stream) {
    return hop_gate_impl(true, n_hops, h_x, h_ldx, d_vec, bias, d_out, ldo, pad_cols, d_w_out, ldw, d_g_out, ldg, n, d, stream);
}

// scalar bias, always: a NaN bias means NaN scores (and d_vec needs only round_up(d, 4) floats)
SGL_EXPORT int sgl_hop_gate_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_vec, float bias,
                                float *d_out, int64_t ldo, float *d_w_out, int64_t ldw, float *d_g_out, int64_t ldg, int64_t n,
                                int64_t d, void *stream) {
    return hop_gate_impl(false, n_hops, h_x, h_ldx, d_vec, bias, d_out, ldo, 0, d_w_out, ldw, d_g_out, ldg, n, d, stream);
}

// GAMLP-R's recursive gate in one pass (recursive_fused_kernel).  d_vec: [w_x | w_acc], each zero-padded to round_up(d, 4)
// floats; bias as in sgl_hop_gate_f32 (NaN: the float right after the two padded vectors).  pad_cols as in the *_padded_f32 entries.
SGL_EXPORT int sgl_hop_recursive_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_vec, float bias,
                                     float *d_out, int64_t ldo, int64_t pad_cols, float *d_w_out, int64_t ldw, float *d_a_out,
                                     int64_t lda, float *d_c_out, int64_t ldc, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_recursive_f32: bad sizes");
    {
        const int prc = check_pad("sgl_hop_recursive_f32", d, pad_cols, ldo);
        if (prc != SGL_OK) return prc;
    }
    Hops hx;
    bool vec4 = aligned_to(d_vec, 16);
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    if (n == 0 || d == 0) return SGL_OK;
    SGL_REQUIRE(d_vec && d_out && ldo >= d, "sgl_hop_recursive_f32: bad arguments");
    SGL_REQUIRE((!d_w_out || ldw >= n_hops) && (!d_a_out || lda >= n_hops) && (!d_c_out || ldc >= n_hops),
                "sgl_hop_recursive_f32: bad weight / score buffers");
    if (!(vec4 && n_hops <= 16 && d <= 512 && ldo % 4 == 0 && aligned_to(d_out, 16)))
        return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_recursive_f32: needs <= 16 hops, d <= 512 and 16-byte aligned rows (use the step-by-step route)");
    hipStream_t st = sgl::as_stream(stream);
    const RowLayout lay = pick_row_layout(d, n_hops);
    const int64_t blocks = (n + (256 / lay.lpr) - 1) / (256 / lay.lpr);
    if (!sgl::launch_fits(blocks, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_recursive_f32: too many rows for one launch (shard the matrix)");
    const int dv = (int)((d + 3) / 4 * 4);
    const float *bias_ptr = (bias != bias) ? d_vec + 2 * dv : nullptr;
#define SGL_RF(L, C, HM) \
    hipLaunchKernelGGL((recursive_fused_kernel<L, C, HM>), dim3((unsigned)blocks), dim3(256), 0, st, hx, n_hops, d_vec, dv, bias, bias_ptr, d_out, ldo, d_w_out, ldw, d_a_out, lda, d_c_out, ldc, n, (int)d, out_cols(d, pad_cols, (L) * (C) * 4))
#define SGL_RF_H(L, C) SGL_HOPS_UP_TO_16(SGL_RF, L, C)
#define SGL_RF_H12(L, C) SGL_HOPS_UP_TO_12(SGL_RF, L, C)
#define SGL_RF_H6(L, C) (void)0          /* 8 x 5 is never chosen for this kernel (pick_row_layout) */
    SGL_ROWREG_DISPATCH(SGL_RF_H, SGL_RF_H12, SGL_RF_H6, lay);
#undef SGL_RF_H6
#undef SGL_RF_H12
#undef SGL_RF_H
#undef SGL_RF
    SGL_LAUNCH_CHECK("sgl_hop_recursive_f32");
    return SGL_OK;
}

// backward of the [n, H] recursion of sgl_hop_recursive_f32 (recursive_scalar_bwd_kernel): dA, dC [n, H] and the per-row bias
// gradient dB [n] (its sum is the Linear's bias gradient; summed by the caller so the result does not depend on a launch shape)
SGL_EXPORT int sgl_hop_recursive_bwd_f32(int n_hops, const float *d_a, int64_t lda, const float *d_c, int64_t ldc, float bias,
                                         const float *d_bias, const float *d_gw, int64_t ldg, float *d_da, int64_t ldda,
                                         float *d_dc, int64_t lddc, float *d_db, int64_t n, void *stream) {
    SGL_REQUIRE(n >= 0 && n_hops >= 1 && n_hops <= 16, "sgl_hop_recursive_bwd_f32: 1 <= n_hops <= 16");
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_a && d_c && d_gw && lda >= n_hops && ldc >= n_hops && ldg >= n_hops, "sgl_hop_recursive_bwd_f32: bad score / gradient matrices");
    SGL_REQUIRE((!d_da || ldda >= n_hops) && (!d_dc || lddc >= n_hops), "sgl_hop_recursive_bwd_f32: bad output matrices");
    SGL_REQUIRE(bias == bias || d_bias, "sgl_hop_recursive_bwd_f32: bias = NaN needs the device bias");
    hipStream_t st = sgl::as_stream(stream);
    const int64_t blocks = (n + 255) / 256;
    if (!sgl::launch_fits(blocks, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_recursive_bwd_f32: too many rows for one launch");
    const float *bias_ptr = (bias != bias) ? d_bias : nullptr;
#define SGL_RB(HM) \
    hipLaunchKernelGGL((recursive_scalar_bwd_kernel<HM>), dim3((unsigned)blocks), dim3(256), 0, st, n_hops, d_a, lda, d_c, ldc, bias, bias_ptr, d_gw, ldg, d_da, ldda, d_dc, lddc, d_db, n)
    if (n_hops <= 4) SGL_RB(4);
    else if (n_hops <= 8) SGL_RB(8);
    else if (n_hops <= 12) SGL_RB(12);
    else SGL_RB(16);
#undef SGL_RB
    SGL_LAUNCH_CHECK("sgl_hop_recursive_bwd_f32");
    return SGL_OK;
}

SGL_EXPORT int sgl_hop_rowdot2_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_u, int64_t ldu,
                                   uint64_t u_mask, const float *d_vec, int h0, int h1, float *d_p, int64_t ldp, float *d_a,
                                   int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_rowdot2_f32: bad sizes");
    Hops hx;
    bool vec4 = aligned_to(d_vec, 16) && (!d_u || (aligned_to(d_u, 16) && ldu % 4 == 0));
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    SGL_REQUIRE(h0 >= 0 && h0 <= h1 && h1 <= n_hops, "sgl_hop_rowdot2_f32: bad hop range");
    if (n == 0) return SGL_OK;
    SGL_REQUIRE(d_vec && (h1 == h0 || (d_p && ldp >= h1 - h0)), "sgl_hop_rowdot2_f32: bad arguments");
    SGL_REQUIRE(u_mask == 0 || (d_u && d_a && ldu >= d), "sgl_hop_rowdot2_f32: the reference part needs U [n_hops, ldu] and A [n]");
    if (n_hops < 64) SGL_REQUIRE((u_mask >> n_hops) == 0, "sgl_hop_rowdot2_f32: u_mask names a hop beyond n_hops");
    if (!(vec4 && n_hops <= 16 && d <= 512 && d > 0))
        return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_rowdot2_f32: needs <= 16 hops, 0 < d <= 512 and 16-byte aligned rows");
    hipStream_t st = sgl::as_stream(stream);
    const RowLayout lay = pick_row_layout(d, n_hops, true);
    const int64_t blocks = (n + (256 / lay.lpr) - 1) / (256 / lay.lpr);
    if (!sgl::launch_fits(blocks, 256)) return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_rowdot2_f32: too many rows for one launch (shard the matrix)");
    float *a_out = u_mask ? d_a : nullptr;
#define SGL_R2(L, C, HM) \
    hipLaunchKernelGGL((hop_rowdot2_reg_kernel<L, C, HM>), dim3((unsigned)blocks), dim3(256), 0, st, hx, n_hops, d_u, ldu, (unsigned long long)u_mask, d_vec, h0, h1, d_p, ldp, a_out, n, (int)d)
#define SGL_R2_H(L, C) SGL_HOPS_UP_TO_16(SGL_R2, L, C)
#define SGL_R2_H12(L, C) SGL_HOPS_UP_TO_12(SGL_R2, L, C)
#define SGL_R2_H6(L, C) SGL_HOPS_UP_TO_6(SGL_R2, L, C)
    SGL_ROWREG_DISPATCH(SGL_R2_H, SGL_R2_H12, SGL_R2_H6, lay);
#undef SGL_R2_H6
#undef SGL_R2_H12
#undef SGL_R2_H
#undef SGL_R2
    SGL_LAUNCH_CHECK("sgl_hop_rowdot2_f32");
    return SGL_OK;
}

// ---- weight gradients of the row-dots ----------------------------------------------------------------------------------------
constexpr int kColsumBlocks = 1024;
static int64_t colsum_per_block(int64_t n, int slots) {
    const int64_t rl = 256 / slots;
    int64_t per = (n + kColsumBlocks - 1) / kColsumBlocks;
    per = std::max<int64_t>(per, rl * 4);
    return (per + rl - 1) / rl * rl;
}

SGL_EXPORT int64_t sgl_hop_colsum_scratch(int n_hops, int64_t n, int64_t d) {
    if (n_hops < 1 || n < 0 || d < 1 || d > 1024) return 0;
    const int slots = (int)((d + 3) / 4);
    const int64_t per = colsum_per_block(n, slots);
    const int64_t blocks = std::max<int64_t>(1, (n + per - 1) / per);
    return blocks * n_hops * slots * 4;
}

SGL_EXPORT int sgl_hop_colsum_f32(int n_hops, const float *const *h_x, const int64_t *h_ldx, const float *d_w, int64_t ldw, int sw,
                                  float *d_out, int64_t ldo, float *d_scratch, int64_t n, int64_t d, void *stream) {
    SGL_REQUIRE(n >= 0 && d >= 0 && d < INT32_MAX, "sgl_hop_colsum_f32: bad sizes");
    Hops hx;
    bool vec4 = true;
    int rc = fill_hops(hx, n_hops, h_x, h_ldx, d, vec4);
    if (rc != SGL_OK) return rc;
    SGL_REQUIRE(d_out && ldo >= d, "sgl_hop_colsum_f32: bad output");
    SGL_REQUIRE(sw == 0 || sw == 1, "sgl_hop_colsum_f32: sw must be 0 (one weight per row) or 1 (one per row and hop)");
    hipStream_t st = sgl::as_stream(stream);
    if (d == 0) return SGL_OK;
    if (n == 0) {
        SGL_HIP_CHECK(hipMemset2DAsync(d_out, ldo * sizeof(float), 0, d * sizeof(float), n_hops, st));
        return SGL_OK;
    }
    SGL_REQUIRE(d_w && (sw == 0 || ldw >= n_hops) && ldw >= 1, "sgl_hop_colsum_f32: bad weights");
    if (!(vec4 && n_hops <= 16 && d <= 1024))
        return sgl::fail(SGL_ERR_UNSUPPORTED, "sgl_hop_colsum_f32: needs <= 16 hops, d <= 1024 and 16-byte aligned rows");
    SGL_REQUIRE(d_scratch, "sgl_hop_colsum_f32: NULL scratch (sgl_hop_colsum_scratch floats)");
    const int slots = (int)((d + 3) / 4);
    const int64_t per = colsum_per_block(n, slots);
    const int blocks = (int)std::max<int64_t>(1, (n + per - 1) / per);
#define SGL_CS(HM) hipLaunchKernelGGL((hop_colsum_partial_kernel<HM>), dim3(blocks), dim3(256), 0, st, hx, n_hops, d_w, ldw, sw, d_scratch, n, (int)d, slots, per)
    if (n_hops <= 4) SGL_CS(4);
    else if (n_hops <= 8) SGL_CS(8);
    else if (n_hops <= 12) SGL_CS(12);
    else SGL_CS(16);
#undef SGL_CS
    SGL_LAUNCH_CHECK("sgl_hop_colsum_f32(partial)");
    hipLaunchKernelGGL(hop_colsum_final_kernel, dim3(n_hops * slots), dim3(64), 0, st, d_scratch, blocks, n_hops, slots, d_out, ldo, (int)d);
    SGL_LAUNCH_CHECK("sgl_hop_colsum_f32(final)");
    return SGL_OK;
}
