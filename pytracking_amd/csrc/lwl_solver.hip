// LWL few-shot learner: `GNSteepestDescent.forward` (ltr/models/meta/steepestdescent.py:32-105) specialised to
// `LWTLResidual` (ltr/models/lwl/loss_residual_modules.py:16-41), one sequence, F <= 16 filters.
//
// The residual  r(w) = [ sw * (apply_filter(feat, w) - label) ,  lambda * w ]  is linear in the filter, so the two
// autograd passes of the reference (steepestdescent.py:70,73) are explicit:
//     g = J^T r = apply_feat_transpose(feat, sw * r_data) + lambda^2 * w
//     h = J g   = [ sw * apply_filter(feat, g) , lambda * g ]
//     alpha = |g|^2 / max(|h|^2 + steplength_reg * |g|^2, 1e-8) ;   w <- w - alpha * g          (:76-88)
// and the scores of the next iterate follow without another pass:  s <- s - alpha * apply_filter(feat, g).
// Per iteration: k_mf_adj -> k_lwl_g -> k_mf_corr -> k_lwl_hh -> k_lwl_upd; no host synchronisation.
#include <algorithm>
#include "common.h"
#include "pt_internal.h"

#define LWL_NBLK 256       // partial sums per reduction (fixed order everywhere)
#define LWL_MAX_ITER 64

struct LwlArgs {
    long N;                // n*F*H*W
    int n, F, C, HW, KK, CKK /* F*C*KK */, NSG, sw_mode;
    float lam, slreg, sw_scalar;
    const float *label, *sw;
    float *s, *sg, *rmap, *gpart, *g, *gT, *ggp, *hhp, *lossp, *w_iters;
    int C4KK64;            // (C/4)*KK*64: live part of the transposed table
    const float* w0;
};

__device__ __forceinline__ const float* lwl_w(const LwlArgs& a, int t) { return t == 0 ? a.w0 : a.w_iters + (long)t * a.CKK; }

__device__ __forceinline__ float lwl_sw(const LwlArgs& a, long e) {
    // loss_residual_modules.py:27-33: scalar sqrt(1/n) | per image | per element
    if (a.sw_mode == 0) return a.sw_scalar;
    if (a.sw_mode == 1) return a.sw[e / ((long)a.F * a.HW)];
    return a.sw[e];
}

// every thread: the sum of the LWL_NBLK partials, fixed order (thread k holds partial k; wave sums, then the waves in order).
// As a serial loop in every thread this was 2 x 256 dependent loads in front of the update kernel.
__device__ __forceinline__ float lwl_sum_parts(const float* p, float* scratch) {
    static_assert(LWL_NBLK == 256 && PT_MF_SQ_PARTS == LWL_NBLK, "one partial per thread of the 256-thread block");
    return block_sum(p[threadIdx.x], scratch);
}

// s_t (in place when t > 0: s <- s - alpha*sg), residual, adjoint input, loss partials; t > 0 also w_t = w_{t-1} - alpha*g
__global__ __launch_bounds__(256) void k_lwl_upd(LwlArgs a, int t, int want_loss, int last) {
    __shared__ float scratch[16];
    float alpha = 0.f;
    // LWL_U strided elements per round: all loads of a round are issued before the first store (one element per round left
    // every round waiting for its own loads: 12 dependent round trips).  Round 4: the FIRST round -- the only one with few
    // samples -- and the filter elements are requested in front of the step-length reduction, which they do not depend on.
    constexpr int LWL_U = 4;
    const long stride = (long)LWL_NBLK * 256;
    const long ef = (long)blockIdx.x * 256 + threadIdx.x;
    const bool skip = last && !want_loss;
    float sv[LWL_U], gv[LWL_U], lb[LWL_U], sw[LWL_U];
    if (!skip) {
#pragma unroll
        for (int u = 0; u < LWL_U; ++u) {
            const long e = min(ef + u * stride, a.N - 1);
            sv[u] = a.s[e];
            gv[u] = t > 0 ? a.sg[e] : 0.f;
            lb[u] = a.label[e];
            sw[u] = lwl_sw(a, e);
        }
    }
    if (t > 0) {
        const float* wp = lwl_w(a, t - 1);
        float* wn = a.w_iters + (long)t * a.CKK;
        constexpr int WU = 2;                                                           // filter elements per thread and round
        float wv[WU], gw[WU];
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const long e = min(ef + u * stride, (long)a.CKK - 1);
            wv[u] = wp[e]; gw[u] = a.g[e];
        }
        const float gg = lwl_sum_parts(a.ggp, scratch), hh = lwl_sum_parts(a.hhp, scratch) + a.lam * a.lam * gg;
        alpha = gg / fmaxf(hh + a.slreg * gg, 1e-8f);                                   // steepestdescent.py:76-80
#pragma unroll
        for (int u = 0; u < WU; ++u)
            if (ef + u * stride < a.CKK) wn[ef + u * stride] = wv[u] - alpha * gw[u];   // :83-88
        for (long e = ef + WU * stride; e < a.CKK; e += stride) wn[e] = wp[e] - alpha * a.g[e];
    }
    if (skip) return;
    float lacc = 0.f;
    for (long e0 = ef; e0 < a.N; e0 += LWL_U * stride) {
        if (e0 != ef) {                                                                 // uniform per thread position: later rounds
#pragma unroll
            for (int u = 0; u < LWL_U; ++u) {
                const long e = min(e0 + u * stride, a.N - 1);
                sv[u] = a.s[e];
                gv[u] = t > 0 ? a.sg[e] : 0.f;
                lb[u] = a.label[e];
                sw[u] = lwl_sw(a, e);
            }
        }
#pragma unroll
        for (int u = 0; u < LWL_U; ++u) {
            const long e = e0 + u * stride;
            if (e < a.N) {
                const float v = t > 0 ? sv[u] - alpha * gv[u] : sv[u];
                if (t > 0) a.s[e] = v;
                const float r = sw[u] * (v - lb[u]);                                    // loss_residual_modules.py:36
                lacc += r * r;
                if (!last) a.rmap[e] = sw[u] * r;
            }
        }
    }
    if (want_loss) {
        const float tot = block_sum(lacc, scratch);
        if (threadIdx.x == 0) a.lossp[(long)t * LWL_NBLK + blockIdx.x] = tot;
    }
}

// g = sum of the sample-group partials + lambda^2 * w_t ; partial |g|^2
__global__ __launch_bounds__(256) void k_lwl_g(LwlArgs a, int t) {
    __shared__ float scratch[16];
    const float* w = lwl_w(a, t);
    float acc = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < a.CKK; e += (long)LWL_NBLK * 256) {
        float v = 0.f;
        const float wv = w[e];
        for (int k0 = 0; k0 < a.NSG; k0 += 8) {                     // fixed-size batches, clamped: 8 partial loads in flight whatever NSG is
            float pv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pv[k] = a.gpart[(long)min(k0 + k, a.NSG - 1) * a.CKK + e];
#pragma unroll
            for (int k = 0; k < 8; ++k) v += k0 + k < a.NSG ? pv[k] : 0.f;
        }
        v += a.lam * a.lam * wv;
        a.g[e] = v;
        {                                                           // the same value in the order k_mf_corr reads it
            const int tap = (int)(e % a.KK), c = (int)((e / a.KK) % a.C), f = (int)(e / ((long)a.KK * a.C));
            a.gT[pt_mf_wt_index(c, f, tap, a.KK)] = v;
        }
        acc += v * v;
    }
    const float tot = block_sum(acc, scratch);
    if (threadIdx.x == 0) a.ggp[blockIdx.x] = tot;
}

// partial |sw * F g|^2
__global__ __launch_bounds__(256) void k_lwl_hh(LwlArgs a) {
    __shared__ float scratch[16];
    float acc = 0.f;
#pragma unroll 4
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < a.N; e += (long)LWL_NBLK * 256) {
        const float h = lwl_sw(a, e) * a.sg[e];
        acc += h * h;
    }
    const float tot = block_sum(acc, scratch);
    if (threadIdx.x == 0) a.hhp[blockIdx.x] = tot;
}

// losses[t] = (sum r_data^2 + lambda^2 |w_t|^2) / (N + F*C*KK)      (steepestdescent.py:28-29)
__global__ __launch_bounds__(256) void k_lwl_loss(LwlArgs a, float* __restrict__ losses) {
    __shared__ float scratch[16];
    const int t = blockIdx.x;
    const float* w = lwl_w(a, t);
    float acc = 0.f;
    for (int e = threadIdx.x; e < a.CKK; e += 256) acc += w[e] * w[e];
    const float ww = block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        float l = 0.f;
        for (int k = 0; k < LWL_NBLK; ++k) l += a.lossp[(long)t * LWL_NBLK + k];
        losses[t] = (l + a.lam * a.lam * ww) / (float)((double)a.N + (double)a.CKK);
    }
}

__global__ void k_mf_sum_groups(const float* __restrict__ gpart, float* __restrict__ out, int groups, long count) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < groups; ++k) s += gpart[(long)k * count + e];
    out[e] = s;
}

// ---------------------------------------------------------------------------------------------------
struct LwlCarve { size_t s, sg, rmap, gpart, g, gT, wT, ggp, hhp, lossp, cpart, total; };

static LwlCarve lwl_carve(int n, int F, int C, int H, int W, int K) {
    LwlCarve c;
    size_t off = 0;
    auto take = [&](size_t nfl) { size_t o = off; off += pt_align_floats(nfl); return o; };
    const size_t N = (size_t)n * F * H * W;
    c.s = take(N); c.sg = take(N); c.rmap = take(N);
    c.gpart = take(pt_mf_gpart_floats(n, F, C, H, W, K));
    c.g = take((size_t)F * C * K * K);
    c.gT = take(pt_mf_wt_floats(C, K)); c.wT = take(pt_mf_wt_floats(C, K));
    c.ggp = take(LWL_NBLK); c.hhp = take(LWL_NBLK);
    c.lossp = take((size_t)(LWL_MAX_ITER + 1) * LWL_NBLK);
    c.cpart = take(pt_mf_corr_part_floats(n, F, C, H, W, K));       // channel-split partial maps of the correlation (few samples)
    c.total = off;
    return c;
}

static int mf_check(int n, int F, int C, int H, int W, int K) {
    if (n <= 0 || F <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return PT_ERR_SHAPE;
    if (pt_mf_groups(n, F, C, H, W, K) == 0) return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

extern "C" size_t pt_lwl_ws_bytes(int n, int F, int C, int H, int W, int K) {
    if (mf_check(n, F, C, H, W, K)) return 0;
    return lwl_carve(n, F, C, H, W, K).total * sizeof(float);
}

extern "C" size_t pt_apply_filter_mf_ws_bytes(int n, int F, int C, int H, int W, int K) {
    if (mf_check(n, F, C, H, W, K)) return 0;
    return (pt_align_floats(pt_mf_wt_floats(C, K)) + pt_align_floats(pt_mf_corr_part_floats(n, F, C, H, W, K))) * sizeof(float);
}

extern "C" int pt_apply_filter_mf_f32(const float* feat, long feat_stride_n, const float* filt, float* scores, int n,
                                      int F, int C, int H, int W, int K, void* ws, size_t ws_bytes, void* stream) {
    if (!feat || !filt || !scores || !ws) return PT_ERR_NULL;
    int rc = mf_check(n, F, C, H, W, K);
    if (rc) return rc;
    if (feat_stride_n < (long)C * H * W) return PT_ERR_SHAPE;
    if (ws_bytes < pt_apply_filter_mf_ws_bytes(n, F, C, H, W, K) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    rc = pt_launch_mf_wtrans(filt, (float*)ws, F, C, K, (hipStream_t)stream);
    if (rc) return rc;
    float* part = pt_mf_corr_part_floats(n, F, C, H, W, K) ? (float*)ws + pt_align_floats(pt_mf_wt_floats(C, K)) : nullptr;
    return pt_launch_mf_corr(feat, feat_stride_n, (const float*)ws, scores, n, F, C, H, W, K, (hipStream_t)stream, 0, 1, part);
}

extern "C" size_t pt_feat_transpose_mf_ws_bytes(int n, int F, int C, int H, int W, int K) {
    if (mf_check(n, F, C, H, W, K)) return 0;
    return pt_align_floats(pt_mf_gpart_floats(n, F, C, H, W, K)) * sizeof(float);
}

extern "C" int pt_feat_transpose_mf_f32(const float* feat, long feat_stride_n, const float* inp, float* grad, int n,
                                        int F, int C, int H, int W, int K, void* ws, size_t ws_bytes, void* stream) {
    if (!feat || !inp || !grad || !ws) return PT_ERR_NULL;
    int rc = mf_check(n, F, C, H, W, K);
    if (rc) return rc;
    if (feat_stride_n < (long)C * H * W) return PT_ERR_SHAPE;
    if (ws_bytes < pt_feat_transpose_mf_ws_bytes(n, F, C, H, W, K) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    rc = pt_launch_mf_adj(feat, feat_stride_n, inp, (float*)ws, n, F, C, H, W, K, st);
    if (rc) return rc;
    const long count = (long)F * C * K * K;
    hipLaunchKernelGGL(k_mf_sum_groups, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, (const float*)ws, grad,
                       pt_mf_groups(n, F, C, H, W, K), count);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" int pt_lwl_gn_solve_f32(const float* w_in, const float* feat, long feat_stride_n, const float* label,
                                   const float* sample_weight, int sw_mode, float filter_reg, float steplength_reg, int n,
                                   int F, int C, int H, int W, int K, int num_iter, float* w_iters, float* losses,
                                   void* ws, size_t ws_bytes, void* stream) {
    if (!w_in || !feat || !label || !w_iters || !ws) return PT_ERR_NULL;
    int rc = mf_check(n, F, C, H, W, K);
    if (rc) return rc;
    if (num_iter < 0 || feat_stride_n < (long)C * H * W) return PT_ERR_SHAPE;
    if (num_iter > LWL_MAX_ITER || sw_mode < 0 || sw_mode > 2) return PT_ERR_UNSUPPORTED;
    if (sw_mode != 0 && !sample_weight) return PT_ERR_NULL;
    LwlCarve cv = lwl_carve(n, F, C, H, W, K);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* base = (float*)ws;
    LwlArgs a;
    a.N = (long)n * F * H * W; a.n = n; a.F = F; a.C = C; a.HW = H * W; a.KK = K * K; a.CKK = F * C * K * K;
    a.NSG = pt_mf_groups(n, F, C, H, W, K); a.sw_mode = sw_mode;
    a.lam = filter_reg; a.slreg = steplength_reg; a.sw_scalar = sqrtf(1.0f / (float)n);
    a.label = label; a.sw = sample_weight;
    a.s = base + cv.s; a.sg = base + cv.sg; a.rmap = base + cv.rmap; a.gpart = base + cv.gpart; a.g = base + cv.g;
    a.gT = base + cv.gT;
    a.ggp = base + cv.ggp; a.hhp = base + cv.hhp; a.lossp = base + cv.lossp; a.w_iters = w_iters; a.w0 = w_in;
    const int want_loss = losses != nullptr;
    if (w_iters != w_in &&
        hipMemcpyAsync(w_iters, w_in, (size_t)a.CKK * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return PT_ERR_LAUNCH;
    if (num_iter == 0 && !want_loss) return PT_OK;
    // the padding of the transposed gradient table is written once (k_lwl_g only touches live entries): by the launch that
    // transposes the start filter (round 4; it was a memset node of its own in front of every solve)
    rc = pt_launch_mf_wtrans(w_in, base + cv.wT, F, C, K, st, 1, a.gT);
    if (rc) return rc;
    float* cpart = pt_mf_corr_part_floats(n, F, C, H, W, K) ? base + cv.cpart : nullptr;
    rc = pt_launch_mf_corr(feat, feat_stride_n, base + cv.wT, a.s, n, F, C, H, W, K, st, 0, 1, cpart);   // s_0 = F w_0
    if (rc) return rc;
    hipLaunchKernelGGL(k_lwl_upd, dim3(LWL_NBLK), dim3(256), 0, st, a, 0, want_loss, (int)(num_iter == 0));
    PT_CHECK_LAUNCH();
    for (int t = 0; t < num_iter; ++t) {
        rc = pt_launch_mf_adj(feat, feat_stride_n, a.rmap, a.gpart, n, F, C, H, W, K, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_lwl_g, dim3(LWL_NBLK), dim3(256), 0, st, a, t);
        PT_CHECK_LAUNCH();
        // F g; with few samples the correlation is split over the channels and the kernel that sums the splits leaves the
        // partials of |sw * F g|^2 as well (one launch less per iteration where launches are what an iteration costs)
        int hh_done = 0;
        const PtMfSq sq = {a.sw, a.sw_mode, a.sw_scalar, (long)a.F * a.HW, a.hhp, &hh_done};
        rc = pt_launch_mf_corr(feat, feat_stride_n, a.gT, a.sg, n, F, C, H, W, K, st, 0, 1, cpart, &sq);
        if (rc) return rc;
        if (!hh_done) {
            hipLaunchKernelGGL(k_lwl_hh, dim3(LWL_NBLK), dim3(256), 0, st, a);
            PT_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(k_lwl_upd, dim3(LWL_NBLK), dim3(256), 0, st, a, t + 1, want_loss, (int)(t + 1 == num_iter));
        PT_CHECK_LAUNCH();
    }
    if (want_loss) {
        hipLaunchKernelGGL(k_lwl_loss, dim3(num_iter + 1), dim3(256), 0, st, a, losses);
        PT_CHECK_LAUNCH();
    }
    return PT_OK;
}
