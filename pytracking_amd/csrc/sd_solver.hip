// Steepest-descent filter optimisers (reference: ltr/models/target_classifier/optimizer.py):
//   DiMPSteepestDescentGN (:85-170), DiMPL2SteepestDescentGN (:211-291), PrDiMPSteepestDescentNewton (:355-439)
// for one sequence, unrolled as launches on one stream with no host synchronisation.
//
// Per iteration the reference makes three passes over the sample memory (apply_filter, apply_feat_transpose,
// apply_filter).  Because apply_filter is linear in the filter, the scores of the next iterate follow from
// quantities already computed:   s_{t+1} = F w_{t+1} = s_t - step*alpha_t * (F g_t),
// so after the first correlation every iteration needs exactly two passes (adjoint, then correlation with
// the gradient) -- the algorithmic minimum SURVEY.md section 8(d) prices (2 feature reads / iteration).
//
//   maps            label / mask / weight maps from the boxes (radial LUTs, Gaussians)       [n WGs]
//   corr(w_0)       partial score maps                                                        [n x KS WGs]
//   pw INIT         s_0, residual map r, its im2col R, per-sample loss                        [n WGs]
//   repeat T times:
//     adj(R)        partial gradients                                                         [C/16 x KSPL WGs]
//     corr(g)       g = sum partials + reg*w_t reduced in the prologue, |g|^2 slices,
//                   partial (F g) maps                                                        [n x KS WGs]
//     pw SGQ        sg = F g ; per-sample curvature term q_i                                  [n WGs]
//     pw UPDATE     alpha ; w_{t+1} ; s_{t+1} ; next residual map + R ; loss                  [n WGs]
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "pt_internal.h"
#include "rbuild.h"
#include "sd_common.h"

enum { PW_INIT = 0, PW_SGQ = 1, PW_UPDATE = 2 };

__global__ void k_sd_maps(SdArgs a) {
    __shared__ float scratch[16];
    __shared__ int amin[2];
    const int i = blockIdx.x;
    if (a.cls_spart && i == a.cls_slot) sd_classify_fin(a);     // uniform per workgroup
    sd_maps_sample(a, i, scratch, amin);
}

// ----------------------------------------------------------------------------------------------------
// pointwise stages: one workgroup per sample
// ----------------------------------------------------------------------------------------------------
// sum of the KS channel-slice partials of one score element (loads issued together, fixed summation order)
__device__ __forceinline__ float sd_sum_slices(const SdArgs& a, int i, int o) {
    const float* p = a.spart + (long)i * a.OO + o;
    const long st = (long)a.n * a.OO;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= a.KS; k += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[(k + q) * st];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; k < a.KS; ++k) s += p[k * st];
    return s;
}

// stage: PW_INIT (s from correlation partials), PW_SGQ, PW_UPDATE (s_{t} = s_{t-1} - step*alpha*sg).
// t = index of the iterate whose scores this launch produces (INIT: 0).  last: no further iteration follows.
__global__ __launch_bounds__(512) void k_sd_pw(SdArgs a, int stage, int t, int last, int want_loss) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [OO] residual map (+ [OO] scores for PrDiMP)
    __shared__ float scratch[16];
    const int i = blockIdx.x;
    const long base = (long)i * a.OO;
    const bool prdimp = a.kind == PT_SD_PRDIMP;
    const int sact = a.kind == PT_SD_DIMP_L2 ? 2 : a.score_act;
    const float swp = prdimp ? (a.has_sw ? a.sw[i] : 1.0f / (float)a.n) : 0.f;

    if (stage == PW_SGQ) {
        float acc = 0.f;
        if (!prdimp) {
            for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
                const float sgv = sd_sum_slices(a, i, o);
                a.sg[base + o] = sgv;
                float act, der;
                act_pair(sact, a.act_param, a.s[base + o], a.mask[base + o], act, der);
                const float q = a.sws[base + o] * (der * sgv);                      // :151-152
                acc += q * q;
            }
            const float tot = block_sum(acc, scratch);
            if (threadIdx.x == 0) a.qs[i] = tot;
        } else {
            float psum = 0.f;
            for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
                const float sgv = sd_sum_slices(a, i, o);
                a.sg[base + o] = sgv;
                lds[o] = sgv;
                psum += a.mask[base + o] * sgv;                                     // :419
            }
            const float tot = block_sum(psum, scratch);
            for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
                const float P = a.mask[base + o], sgv = lds[o];
                const float h = P * sgv - P * tot;                                  // :420
                acc += sgv * h;
            }
            const float ghg = block_sum(acc, scratch);
            if (threadIdx.x == 0) a.qs[i] = swp * fmaxf(ghg, 0.f);                  // :421-422
        }
        return;
    }

    float astep = 0.f;
    if (stage == PW_UPDATE) {
        astep = sd_alpha_step_wave(a, threadIdx.x & 63);       // wave-parallel, the same fixed order as the k_adj2 prologue
        // this workgroup's slice of the filter update  w_t = w_{t-1} - step*alpha*g   (:160)
        const int chunk = (a.CKK + a.n - 1) / a.n;
        const float* wp = sd_w(a, t - 1);
        float* wn = (last && a.w_final) ? a.w_final : a.w_iters + (long)t * a.CKK;
        const int e0 = i * chunk, e1 = min(a.CKK, e0 + chunk);
        for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) wn[e] = wp[e] - astep * a.g[e];
        if (last && !want_loss) return;
    }

    float lacc = 0.f;
    if (!prdimp) {
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            float sv;
            if (stage == PW_INIT) sv = sd_sum_slices(a, i, o);
            else sv = a.s[base + o] - astep * a.sg[base + o];
            a.s[base + o] = sv;
            float act, der;
            act_pair(sact, a.act_param, sv, a.mask[base + o], act, der);
            const float sw = a.sws[base + o];
            const float r = sw * (act - a.label[base + o]);                         // :140
            lacc += r * r;
            lds[o] = der * (sw * r);                                                // :146
        }
    } else {
        float* sv_l = lds + a.OO;
        float mx = a.has_softmax_reg ? a.softmax_reg : -INFINITY;
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            float sv;
            if (stage == PW_INIT) sv = sd_sum_slices(a, i, o);
            else sv = a.s[base + o] - astep * a.sg[base + o];
            a.s[base + o] = sv;
            sv_l[o] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = block_max(mx, scratch);
        float es = 0.f, ls = 0.f;
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            const float e = expf(sv_l[o] - mx);
            lds[o] = e;
            es += e;
            ls += a.label[base + o] * sv_l[o];
        }
        es = block_sum(es, scratch);
        ls = block_sum(ls, scratch);
        if (a.has_softmax_reg) es += expf(a.softmax_reg - mx);                      // activation.py:7-16
        const float inv = 1.0f / es;
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            const float P = lds[o] * inv;
            a.mask[base + o] = P;
            lds[o] = swp * (P - a.label[base + o]);                                 // :408
        }
        lacc = 0.f;
        if (threadIdx.x == 0) lacc = swp * (logf(es) + mx - ls);                    // :393-396
    }
    if (want_loss) {
        const float tot = block_sum(lacc, scratch);
        if (threadIdx.x == 0) a.lossp[(long)t * a.n + i] = tot;
    }
    if (last) return;
    __syncthreads();
    pt_build_R_sample(lds, a.R, i, a.n, a.H, a.W, a.K, a.K, a.OH, a.OW);
}

// losses[t] = sum_i lossp[t][i] + reg * |w_t|^2     (one workgroup per iterate)
__global__ void k_sd_loss(SdArgs a, float* __restrict__ losses) {
    __shared__ float scratch[16];
    const int t = blockIdx.x;
    const float* w = sd_w(a, t);
    float acc = 0.f;
    for (int e = threadIdx.x; e < a.CKK; e += blockDim.x) acc += w[e] * w[e];
    const float wn = block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        float l = 0.f;
        for (int k = 0; k < a.n; ++k) l += a.lossp[(long)t * a.n + k];
        losses[t] = l + a.reg * wn;
    }
}

// ----------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------
struct SdCarve {
    size_t label, mask, sws, s, sg, spart, R, gpart, g, anum, qs, lossp, total;
};

static SdCarve sd_carve(const PtPlan& p, int max_iter) {
    SdCarve c;
    size_t off = 0;
    auto take = [&](size_t nfl) { size_t o = off; off += pt_align_floats(nfl); return o; };
    const size_t nOO = (size_t)p.n * p.OO;
    c.label = take(nOO); c.mask = take(nOO); c.sws = take(nOO); c.s = take(nOO); c.sg = take(nOO);
    c.spart = take(pt_spart_floats(p));
    c.R = take(pt_R_floats(p));
    c.gpart = take(pt_gpart_floats(p));
    c.g = take((size_t)p.C * p.KK);
    c.anum = take(64);
    c.qs = take(p.n);
    c.lossp = take((size_t)(max_iter + 1) * p.n);
    c.total = off;
    return c;
}


// ----------------------------------------------------------------------------------------------------
// fast path (fast_passes.hip): per iteration  adj2 [update prologue fused] -> corr2 [gradient reduction fused] -> SGQ
// ----------------------------------------------------------------------------------------------------
static const float* sd_w_host(const SdArgs& a, int t) { return t == 0 ? a.w0 : a.w_iters + (long)t * a.CKK; }

struct FastCarve {
    size_t label, mask, sws, lms, pk, s0, s1, sg, spart, gpart, g, anum, qs, lossp, total;
};

static FastCarve fast_carve(const PtFast& f, int max_iter) {
    FastCarve c;
    size_t off = 0;
    auto take = [&](size_t nfl) { size_t o = off; off += pt_align_floats(nfl); return o; };
    const size_t nOO = (size_t)f.n * f.OO;
    c.label = take(nOO); c.mask = take(nOO); c.sws = take(nOO); c.lms = take(4 * nOO); c.pk = take(4 * nOO);
    c.s0 = take(nOO); c.s1 = take(nOO); c.sg = take(nOO);
    c.spart = take(pt_fast_spart_floats(f));
    c.gpart = take(pt_fast_gpart_floats(f));
    c.g = take((size_t)f.C * f.KK);
    c.anum = take(64);
    c.qs = take((size_t)f.n);
    c.lossp = take((size_t)(max_iter + 1) * f.n);
    c.total = off;
    return c;
}

// packed operands of the update stage that k_adj2 runs as its prologue (PReg in fast_passes.hip)
__device__ __forceinline__ void fast_pack(const SdArgs& a, long q, float s, float sg) {
    f32x4 v;
    if (a.kind == PT_SD_PRDIMP) {
        v = (f32x4){s, sg, a.label[q], 0.f};
    } else {
        const float sw = a.sws[q], m = a.mask[q], L = a.label[q];
        if (a.kind == PT_SD_DIMP && a.score_act == PT_ACT_BENTPAR) v = (f32x4){s, sg, L, m};
        else { const float w2 = sw * sw; v = (f32x4){w2 * s, w2 * sg, w2 * L, m}; }
    }
    ((f32x4*)a.pk)[q] = v;
}

// s_0 = sum of the 8 channel-range slices; classification epilogue of the inserted slot; label/mask/weight maps
__global__ __launch_bounds__(512) void k_fast_init(SdArgs a) {
    __shared__ float scratch[16];
    __shared__ int amin[2];
    const int i = blockIdx.x;
    const long base = (long)i * a.OO;
    for (int o = threadIdx.x; o < a.OO; o += blockDim.x) a.s[base + o] = sd_sum_slices(a, i, o);
    if (a.cls_spart && i == a.cls_slot) sd_classify_fin(a);     // uniform per workgroup; re-centres bb[slot]
    sd_maps_sample(a, i, scratch, amin);
    __syncthreads();                                            // PrDiMP: label is finalised by other threads
    for (int o = threadIdx.x; o < a.OO; o += blockDim.x) fast_pack(a, base + o, a.s[base + o], 0.f);
}

// F g = sum of the 8 slices; per-sample curvature term (optimizer.py:151-156 / :416-422); packed operands for k_adj2
__global__ __launch_bounds__(512) void k_fast_sgq(SdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float scratch[16];
    const int i = blockIdx.x;
    const long base = (long)i * a.OO;
    float acc = 0.f;
    if (a.kind != PT_SD_PRDIMP) {
        const int sact = a.kind == PT_SD_DIMP_L2 ? 2 : a.score_act;
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            const float sgv = sd_sum_slices(a, i, o), sv = a.s[base + o];
            a.sg[base + o] = sgv;
            float act, der;
            act_pair(sact, a.act_param, sv, a.mask[base + o], act, der);
            const float q = a.sws[base + o] * (der * sgv);
            acc += q * q;
            fast_pack(a, base + o, sv, sgv);
        }
        const float tot = block_sum(acc, scratch);
        if (threadIdx.x == 0) a.qs[i] = tot;
    } else {
        const float swp = a.has_sw ? a.sw[i] : 1.0f / (float)a.n;
        float psum = 0.f;
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            const float sgv = sd_sum_slices(a, i, o);
            a.sg[base + o] = sgv;
            lds[o] = sgv;
            psum += a.mask[base + o] * sgv;                                         // :419
            fast_pack(a, base + o, a.s[base + o], sgv);
        }
        const float tot = block_sum(psum, scratch);
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            const float P = a.mask[base + o], sgv = lds[o];
            acc += sgv * (P * sgv - P * tot);                                       // :420
        }
        const float ghg = block_sum(acc, scratch);
        if (threadIdx.x == 0) a.qs[i] = swp * fmaxf(ghg, 0.f);                      // :421-422
    }
}

// ----------------------------------------------------------------------------------------------------
// Pointwise stages of the fast path, one memory round trip each (round 3).  The first versions above (kept for
// look-up tables too large for LDS and as the reference of the parity tests' history) read their operands where the reference's statements use them: the
// compiler must keep a load behind every earlier store that may alias it, so k_fast_sgq was three dependent round trips
// (slices -> s, mask -> label, sws) and k_fast_init five.  Here every operand of an element is requested before the first
// wait, values stay in registers between the steps, and all stores come last; one element per thread (blockDim >= OO).
// ----------------------------------------------------------------------------------------------------
#define PT_PW_MAXKS 16
// sum of the KS (<= 16) channel-range partials of element (i, o): all loads in flight together, fixed summation order
__device__ __forceinline__ void pw_load_slices(const SdArgs& a, int i, int oc, float (&v)[PT_PW_MAXKS]) {
    const float* p = a.spart + (long)i * a.OO + oc;
    const long st = (long)a.n * a.OO;
#pragma unroll
    for (int k = 0; k < PT_PW_MAXKS; ++k) v[k] = p[(long)min(k, a.KS - 1) * st];
}
__device__ __forceinline__ float pw_sum_slices(const SdArgs& a, const float (&v)[PT_PW_MAXKS]) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PT_PW_MAXKS; ++k) s += k < a.KS ? v[k] : 0.f;
    return s;
}

// F g = sum of the slices; per-sample curvature term (optimizer.py:151-156 / :416-422); packed operands for k_adj2.
// The pointers of the LOADS and the sizes are scalar kernel parameters (preloaded into SGPRs at wave launch, see
// fast_passes.hip: pt_late_issue); what the stores need comes in the late block.
//   DiMP kinds: p3 = lms {label, mask, sws, -};   PrDiMP: p3 = softmax P (mask), p4 = label density
//   pa = n | OO << 16;   pb = KS | kind << 8 | score_act << 12 | has_sw << 16
struct SgqLate { pt_gf sg, pk, qs; pt_gcf sw; };              // global-qualified members: see pt_gcf (common.h)
__global__ __launch_bounds__(1024) void k_fast_sgq2(const float* __restrict__ spart, const float* __restrict__ sp, const float* __restrict__ p3,
                                                    const float* __restrict__ p4, unsigned pa, unsigned pb, float act_param, SgqLate l_arg) {
    __shared__ float scratch[16];
    const int n = (int)(pa & 0xffffu), OO = (int)(pa >> 16);
    const int KS = (int)(pb & 255u), kind = (int)((pb >> 8) & 15u), score_act = (int)((pb >> 12) & 15u);
    const bool has_sw = ((pb >> 16) & 1u) != 0;
    const int i = blockIdx.x, o = threadIdx.x;
    const bool ok = o < OO;
    const int oc = min(o, OO - 1);
    const long q = (long)i * OO + oc;
    const int nthreads = ((OO + 63) >> 6) << 6;
    // PT_SGQ_EXP (experiments only, profiles/r06c_*: where do the 1.9 us between this kernel and an empty one go?):
    //   1: no loads (operands from the thread index)   2: no stores   3: neither   4: no block reduction (q_i of wave 0 only)
#ifndef PT_SGQ_EXP
#define PT_SGQ_EXP 0
#endif
    float v[PT_PW_MAXKS];
    {
        const float* p = spart + q;
        const long st = (long)n * OO;
#pragma unroll
        for (int k = 0; k < PT_PW_MAXKS; ++k) v[k] = (PT_SGQ_EXP & 1) && PT_SGQ_EXP != 4 ? 1e-3f * (float)(o + k) : p[(long)min(k, KS - 1) * st];
    }
    const float sv = (PT_SGQ_EXP & 1) && PT_SGQ_EXP != 4 ? 0.5f : sp[q];
    f32x4 lm = {0, 0, 0, 0};
    float P = 0.f, L = 0.f;
    if ((PT_SGQ_EXP & 1) && PT_SGQ_EXP != 4) { lm = (f32x4){0.1f, 0.5f, 1.0f, 0.f}; P = 0.01f; L = 0.01f; }
    else if (kind != PT_SD_PRDIMP) lm = ((const f32x4*)p3)[q];
    else { P = p3[q]; L = p4[q]; }
    __builtin_amdgcn_sched_barrier(0);
    const SgqLate l = pt_late_args<SgqLate>(48);                    // 4 pointers + 3 dwords = 44 bytes, 8-aligned
    float sgv = 0.f;
#pragma unroll
    for (int k = 0; k < PT_PW_MAXKS; ++k) sgv += k < KS ? v[k] : 0.f;           // fixed order
    if (kind != PT_SD_PRDIMP) {
        const int sact = kind == PT_SD_DIMP_L2 ? 2 : score_act;
        float act, der;
        act_pair(sact, act_param, sv, lm[1], act, der);
        const float qq = lm[2] * (der * sgv);                                       // :151-152
#if PT_SGQ_EXP == 0
        // Round 6 (profiles/r06c_sgq_ablation.txt: the block reduction cost 0.43 us of this launch in the chain -- two barriers with the
        // result stores queued behind them): the element stores go out first, then ONE barrier; the waves that only contribute a
        // partial sum end there, thread 0 adds the partials in block_sum's order (same bits) and stores q_i.
        const float wq = wave_sum(ok ? qq * qq : 0.f);
        if (ok) {
            l.sg[q] = sgv;
            f32x4 pk;
            if (kind == PT_SD_DIMP && score_act == PT_ACT_BENTPAR) pk = (f32x4){sv, sgv, lm[0], lm[1]};
            else { const float w2 = lm[2] * lm[2]; pk = (f32x4){w2 * sv, w2 * sgv, w2 * lm[0], lm[1]}; }
            ((f32x4*)l.pk)[q] = pk;
        }
        if ((o & 63) == 0) scratch[o >> 6] = wq;
        __syncthreads();
        if (o == 0) {
            float tot = 0.f;
            for (int w = 0; w < (nthreads >> 6); ++w) tot += scratch[w];
            l.qs[i] = tot;
        }
#else
        const float tot = PT_SGQ_EXP == 4 ? wave_sum(ok ? qq * qq : 0.f) : block_sum(ok ? qq * qq : 0.f, scratch, nthreads);
        if (ok && !((PT_SGQ_EXP & 2) && PT_SGQ_EXP != 4 && tot != 123.456f)) {
            l.sg[q] = sgv;
            f32x4 pk;
            if (kind == PT_SD_DIMP && score_act == PT_ACT_BENTPAR) pk = (f32x4){sv, sgv, lm[0], lm[1]};
            else { const float w2 = lm[2] * lm[2]; pk = (f32x4){w2 * sv, w2 * sgv, w2 * lm[0], lm[1]}; }
            ((f32x4*)l.pk)[q] = pk;
        }
        if (o == 0) l.qs[i] = tot;
#endif
    } else {
        const float swp = has_sw ? l.sw[i] : 1.0f / (float)n;
        const float tot = block_sum(ok ? P * sgv : 0.f, scratch, nthreads);                   // :419
        const float ghg = block_sum(ok ? sgv * (P * sgv - P * tot) : 0.f, scratch, nthreads); // :420
        if (ok) {
            l.sg[q] = sgv;
            ((f32x4*)l.pk)[q] = (f32x4){sv, sgv, L, 0.f};
        }
        if (o == 0) l.qs[i] = swp * fmaxf(ghg, 0.f);                                // :421-422
    }
}

// s_0 = sum of the slices; classification epilogue of the inserted slot (its score row IS s_0 of that sample);
// label / mask / weight maps (optimizer.py:111-125, 201-208, 331-353); packed operands for k_adj2.
// Scalar kernel parameters (preloaded) = what the first loads need; the rest in the late block.
//   pa = n | OO << 16;  pb = KS | kind << 8 | cls_slot << 12 (0xfffff: none);  pc = OW | K << 12 | num_bins << 16
#ifndef PT_SD_FUSE_INIT
#ifdef PT_STAMPS
#define PT_SD_FUSE_INIT 0      // phase-stamp builds: the stamp pointer moves the init block of k_adj2<.., INIT> in the argument segment
#else
#define PT_SD_FUSE_INIT 1      // 0 (A/B builds): never fold the init stage into the first adjoint pass
#endif
#endif
#ifndef PT_INIT_LUT_HOT
#define PT_INIT_LUT_HOT 1      // 0 (A/B builds): always take the look-up tables through the late argument block
#endif
struct InitLate {
    pt_gcf label_lut, mask_lut, spatial_lut;
    pt_gf s, label, mask, sws, lms, pk, cls_scores, cls_peak, cls_bb;
    int OH, score_act, mask_act, normalize_label;
    float bin_disp, gauss_sigma, hinge_thr, uni_weight, label_shrink, label_thr;
};
static_assert(sizeof(InitLate) <= 3 * 64, "InitLate: three 16-dword blocks");
//   lut3 (preloaded; round 6): the three DiMP look-up tables as ONE contiguous array (label | mask | spatial, 3 x num_bins) when the caller
//   keeps them that way -- their loads then leave with the first ones instead of behind the late argument block (one round trip less in
//   front of the maps); null: the three pointers of the late block
//   slot_dyn (preloaded like the other scalars): when non-null the classification slot comes from this device int instead of pb -- the
//   graph-replayed one-call frame (frame_full.hip) cannot bake a per-frame slot into its captured launches
__global__ __launch_bounds__(1024) void k_fast_init2(const float* __restrict__ spart, const float* __restrict__ bb, const float* __restrict__ swp_,
                                                     const int* __restrict__ slot_dyn, const float* __restrict__ lut3, unsigned pa, unsigned pb,
                                                     unsigned pc, float feat_stride, InitLate l_arg) {
    extern __shared__ __attribute__((aligned(16))) float lut[];    // DiMP: label | mask | spatial look-up tables
    __shared__ float scratch[16];
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ float bbs[4];
    const int n = (int)(pa & 0xffffu), OO = (int)(pa >> 16);
    const int KS = (int)(pb & 255u), kind = (int)((pb >> 8) & 15u);
    const int slot_d = slot_dyn ? *slot_dyn : 0;                    // requested with the first loads
    const int cls_slot = slot_dyn ? slot_d : (int)(pb >> 12);
    const int OW = (int)(pc & 0xfffu), K = (int)((pc >> 12) & 15u), num_bins = (int)(pc >> 16);
    const int i = blockIdx.x, o = threadIdx.x;
    const bool ok = o < OO;
    const int oc = min(o, OO - 1);
    const long q = (long)i * OO + oc;
    const int nthreads = ((OO + 63) >> 6) << 6;
    // ---- everything this element needs, requested at once
    float v[PT_PW_MAXKS];
    {
        const float* p = spart + q;
        const long st = (long)n * OO;
#pragma unroll
        for (int k = 0; k < PT_PW_MAXKS; ++k) v[k] = p[(long)min(k, KS - 1) * st];
    }
    const float* bp = bb + 4 * i;
    float b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
    const float swv = swp_ ? swp_[i] : 1.0f / (float)n;
    const bool lut_hot = kind == PT_SD_DIMP && lut3 != nullptr;     // uniform
    float lv0 = 0.f;
    if (lut_hot) lv0 = lut3[min(o, 3 * num_bins - 1)];
    __builtin_amdgcn_sched_barrier(0);
    const InitLate l = pt_late_args<InitLate>(56);                  // 5 pointers + 4 dwords = 56 bytes
    if (lut_hot) {
        if (o < 3 * num_bins) lut[o] = lv0;
        for (int e = o + nthreads; e < 3 * num_bins; e += nthreads) lut[e] = lut3[e];      // tables longer than the block
    } else if (kind == PT_SD_DIMP) {
        for (int e = o; e < 3 * num_bins; e += nthreads) {
            const int t = e / num_bins, k = e - t * num_bins;
            lut[e] = (t == 0 ? l.label_lut : (t == 1 ? l.mask_lut : l.spatial_lut))[k];
        }
    }
    float s0 = 0.f;
#pragma unroll
    for (int k = 0; k < PT_PW_MAXKS; ++k) s0 += k < KS ? v[k] : 0.f;            // fixed order
    // ---- classification of the test frame (pytracking/libs/dcf.py:156-164: first maximum) re-centres this sample's box
    const bool cls = i == cls_slot;                                 // uniform per workgroup
    if (cls) {
        float best = ok ? s0 : -INFINITY;
        int besti = ok ? o : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(besti, off, 64);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        const int lane = o & 63, wave = o >> 6, nw = nthreads >> 6;
        if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
        __syncthreads();
        if (o == 0) {
            for (int w = 1; w < nw; ++w)
                if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
            const int row = besti / OW, col = besti - row * OW;
            const float off = (float)(K % 2) * 0.5f;
            bbs[0] = ((float)col + off) * feat_stride - b2 * 0.5f;
            bbs[1] = ((float)row + off) * feat_stride - b3 * 0.5f;
            l.cls_peak[0] = (float)row;
            l.cls_peak[1] = (float)col;
            l.cls_bb[4 * cls_slot] = bbs[0];
            l.cls_bb[4 * cls_slot + 1] = bbs[1];
        }
        __syncthreads();
        b0 = bbs[0];
        b1 = bbs[1];
        if (ok) l.cls_scores[o] = s0;
    } else {
        __syncthreads();                                            // look-up tables staged
    }
    // ---- maps
    const float off = (float)(K % 2) * 0.5f;
    const float ctr_r = (b1 + b3 * 0.5f) / feat_stride - off;       // optimizer.py:112-113 (flip -> row first)
    const float ctr_c = (b0 + b2 * 0.5f) / feat_stride - off;
    const int y = oc / OW, x = oc - y * OW;
    const float d0 = (float)y - ctr_r, d1 = (float)x - ctr_c;
    float lb, m = 0.f, sw = 0.f;
    if (kind == PT_SD_DIMP) {
        sd_init_elem_dimp(lut, num_bins, l.mask_act, l.bin_disp, d0, d1, swv, lb, m, sw);
    } else if (kind == PT_SD_DIMP_L2) {
        sd_init_elem_l2(l.gauss_sigma, l.hinge_thr, d0, d1, swv, lb, m, sw);
    } else {                                                                        // PrDiMP label density, :331-353
        float gss;
        if (l.gauss_sigma == 0.f) {
            // one-hot at the grid point closest to the centre (first minimum per axis, as the reference's argmin)
            int r0 = 0, c0 = 0;
            float m0 = INFINITY, m1 = INFINITY;
            for (int yy = 0; yy < l.OH; ++yy) { float d = ((float)yy - ctr_r); d *= d; if (d < m0) { m0 = d; r0 = yy; } }
            for (int xx = 0; xx < OW; ++xx) { float d = ((float)xx - ctr_c); d *= d; if (d < m1) { m1 = d; c0 = xx; } }
            gss = (y == r0 && x == c0) ? 1.0f : 0.0f;
        } else {
            const float s2 = l.gauss_sigma * l.gauss_sigma, coef = -1.0f / (2.0f * s2);
            gss = (expf(coef * d0 * d0) / (2.0f * 3.14159265358979323846f * s2)) * expf(coef * d1 * d1);
        }
        gss = gss > l.label_thr ? gss : 0.f;
        const float tot = block_sum(ok ? gss : 0.f, scratch, nthreads);
        const float inv = l.normalize_label ? 1.0f / (tot + 1e-8f) : 1.0f;
        const float uni = l.uni_weight / (float)OO;
        lb = (1.0f - l.label_shrink) * ((1.0f - l.uni_weight) * (gss * inv) + uni);
    }
    // ---- stores
    if (!ok) return;
    l.s[q] = s0;
    l.label[q] = lb;
    f32x4 pk;
    if (kind == PT_SD_PRDIMP) {
        pk = (f32x4){s0, 0.f, lb, 0.f};
    } else {
        l.mask[q] = m;
        l.sws[q] = sw;
        ((f32x4*)l.lms)[q] = (f32x4){lb, m, sw, 0.f};
        if (kind == PT_SD_DIMP && l.score_act == PT_ACT_BENTPAR) pk = (f32x4){s0, 0.f, lb, m};
        else { const float w2 = sw * sw; pk = (f32x4){w2 * s0, w2 * 0.f, w2 * lb, m}; }
    }
    ((f32x4*)l.pk)[q] = pk;
}

// the last filter update of a solve, w_T = w_{T-1} - step*alpha*g (optimizer.py:160): its operands are requested together with the
// inputs of alpha; n workgroups, each a slice of the filter.  Every argument is a scalar kernel parameter: with
// -amdgpu-kernarg-preload-count the hardware delivers them in SGPRs at wave launch, no scalar-load round trip in front of the
// first vector load (experiment of round 3: does the ~4 us floor of a dependent tiny kernel move?).
__global__ __launch_bounds__(512) void k_fast_final(const float* __restrict__ wp, const float* __restrict__ g, float* __restrict__ wn,
                                                    const float* __restrict__ qs, const float* __restrict__ anum, int n, int CKK, int KS,
                                                    float step, float reg_eps) {
    const int i = blockIdx.x, lane = threadIdx.x & 63;
    const int chunk = (CKK + n - 1) / n;
    const int e = i * chunk + threadIdx.x;
    const bool ok = (int)threadIdx.x < chunk && e < CKK;
    const int ec = ok ? e : 0;
    const float wv = wp[ec], gv = g[ec];
    // alpha as sd_alpha_step_wave computes it: same operands, same order (sd_final_astep: shared with the deferred form in k_corr2)
    const float qh = lane < n ? qs[lane] : 0.f;
    float qt = 0.f;
    for (int k = lane + 64; k < n; k += 64) qt += qs[k];
    const float an = lane < KS ? anum[lane] : 0.f;
    const float astep = sd_final_astep(qh, qt, an, step, reg_eps);
    if (ok) wn[e] = sd_final_apply(wv, gv, astep);
    for (int e2 = e + blockDim.x; e2 < min(CKK, (i + 1) * chunk); e2 += blockDim.x) wn[e2] = sd_final_apply(wp[e2], g[e2], astep);
}

#define PT_SD_MAX_ITER 64

extern "C" size_t pt_sd_ws_bytes(int n, int C, int H, int W, int K) {
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
    const int OH = H + (K + 1) % 2, OW = W + (K + 1) % 2;
    PtPlan p = pt_make_plan(n, C, H, W, K, K, OH, OW);
    size_t tot = sd_carve(p, PT_SD_MAX_ITER).total;
    PtFast f = pt_fast_plan(n, C, H, W, K, K, OH, OW);
    if (f.ok) tot = std::max(tot, fast_carve(f, PT_SD_MAX_ITER).total);
    return tot * sizeof(float);
}

static void sd_fill_params(SdArgs& a, const pt_sd_params* prm, const float* bb, const float* sample_weight, int n, int C,
                           int H, int W, int K, int OH, int OW) {
    a.n = n; a.C = C; a.H = H; a.W = W; a.K = K; a.OH = OH; a.OW = OW; a.OO = OH * OW; a.CKK = C * K * K;
    a.kind = prm->kind; a.score_act = prm->score_act; a.mask_act = prm->mask_act; a.has_sw = sample_weight != nullptr;
    a.has_softmax_reg = prm->has_softmax_reg; a.normalize_label = prm->normalize_label; a.num_bins = prm->num_bins;
    a.step = prm->step_length; a.reg = prm->reg; a.alpha_eps = prm->alpha_eps; a.feat_stride = prm->feat_stride;
    a.bin_disp = prm->bin_displacement; a.act_param = prm->act_param; a.gauss_sigma = prm->gauss_sigma;
    a.hinge_thr = prm->hinge_threshold; a.uni_weight = prm->uni_weight; a.label_shrink = prm->label_shrink;
    a.softmax_reg = prm->softmax_reg; a.label_thr = prm->label_threshold;
    a.bb = bb; a.sw = sample_weight; a.label_lut = prm->label_lut; a.mask_lut = prm->mask_lut;
    a.spatial_lut = prm->spatial_lut;
    a.s_in = nullptr; a.lms = nullptr; a.pk = nullptr; a.R = nullptr; a.cls_stride = 0;
    a.cls_spart = nullptr; a.cls_KS = 0; a.cls_slot = -1; a.cls_scores = nullptr; a.cls_peak = nullptr; a.cls_bb = nullptr;
    a.cls_slot_dyn = nullptr;
}

// Solver state of the fast path laid out in the caller's workspace (shared by the solve and by the measurement replay).
static int sd_fast_setup(const PtFast& f, const pt_sd_params* prm, const float* w_in, const float* bb,
                         const float* sample_weight, float* w_iters, float* w_final, void* ws, size_t ws_bytes, SdArgs& a,
                         float* sbuf[2]) {
    FastCarve cv = fast_carve(f, PT_SD_MAX_ITER);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    float* base = (float*)ws;
    sd_fill_params(a, prm, bb, sample_weight, f.n, f.C, f.H, f.W, f.KH, f.OH, f.OW);
    a.KS = f.KSC; a.KSPL = f.KSPL;
    a.label = base + cv.label; a.mask = base + cv.mask; a.sws = base + cv.sws; a.sg = base + cv.sg;
    a.lms = base + cv.lms; a.pk = base + cv.pk;
    a.spart = base + cv.spart; a.gpart = base + cv.gpart; a.g = base + cv.g; a.anum = base + cv.anum;
    a.qs = base + cv.qs; a.lossp = base + cv.lossp; a.w_iters = w_iters; a.w0 = w_in; a.w_final = w_final;
    sbuf[0] = base + cv.s0; sbuf[1] = base + cv.s1;
    a.s = sbuf[0];
    return PT_OK;
}

// what a pending last update needs from the workspace of the solve that left it (same carve every call)
static int sd_pending_operands(const PtFast& f, const pt_frame_pending* pend, const float* filter, float* w_iters, const SdArgs& a,
                               PtCorrLazy& lz) {
    if (pend->iters < 2 || pend->iters > PT_SD_MAX_ITER) return PT_ERR_SHAPE;    // (a one-iteration solve is never deferred: w_{T-1} would BE the filter)
    lz.w_prev = w_iters + (long)(pend->iters - 1) * a.CKK;
    lz.g = a.g; lz.anum = a.anum; lz.qs = a.qs; lz.step = pend->step_length; lz.reg_eps = pend->reg_eps;
    lz.w_out = const_cast<float*>(filter);
    return PT_OK;
}

static int sd_solve_fast(const PtFast& f, const pt_sd_params* prm, const float* w_in, const float* feat, long stride_n,
                         const float* bb, const float* sample_weight, int num_iter, float* w_iters, float* losses,
                         void* ws, size_t ws_bytes, hipStream_t st, bool copy_w0, float* w_final, const PtClsFin* cls,
                         const float* src, pt_frame_pending* pend, bool defer) {
    SdArgs a;
    float* sbuf[2];
    int rc = sd_fast_setup(f, prm, w_in, bb, sample_weight, w_iters, w_final, ws, ws_bytes, a, sbuf);
    if (rc) return rc;
    const bool have_pending = pend && pend->iters > 0;
    PtCorrLazy lz{};
    if (have_pending || defer) {
        // chain mode: the filter is updated in place (w_final == w_in), no loss read-out, no iterate 0 copy, 4x4 filters
        if (!pend || w_final != w_in || losses || copy_w0 || f.KK != 16 || f.n > 511 || f.CX * 8 > f.corr_threads) return PT_ERR_UNSUPPORTED;
        if (have_pending && (rc = sd_pending_operands(f, pend, w_in, w_iters, a, lz))) return rc;
    }
    const int n = f.n;
    const int slot = cls ? cls->slot : -1;
    if (cls) {
        // the inserted sample's scores under w_in ARE the classification scores of the test frame
        a.cls_spart = a.spart + (long)slot * a.OO; a.cls_KS = a.KS; a.cls_stride = (long)n * a.OO; a.cls_slot = slot;
        a.cls_scores = cls->scores; a.cls_peak = cls->peak; a.cls_bb = cls->mem_bb;
        a.cls_slot_dyn = cls->slot_dyn;
        // a device-resident slot is served by k_fast_init2 only: refuse before anything is queued
        if (cls->slot_dyn && !(a.KS <= PT_PW_MAXKS && (a.kind != PT_SD_DIMP || (size_t)3 * a.num_bins * sizeof(float) <= 48 * 1024))) return PT_ERR_UNSUPPORTED;
    }
    const int want_loss = losses != nullptr;
    const size_t pw_lds = (size_t)a.OO * sizeof(float) * (a.kind == PT_SD_PRDIMP ? 2 : 1);
    if (copy_w0 && w_iters != w_in) {
        if (hipMemcpyAsync(w_iters, w_in, (size_t)a.CKK * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return PT_ERR_LAUNCH;
    }
    if (num_iter == 0 && !want_loss && !cls) {
        if (have_pending) {                                          // nothing of this call would consume it: apply it on its own
            hipLaunchKernelGGL(k_fast_final, dim3(f.n), dim3(512), 0, st, lz.w_prev, lz.g, lz.w_out, lz.qs, lz.anum, f.n, a.CKK, a.KS,
                               lz.step, lz.reg_eps);
            PT_CHECK_LAUNCH();
            pend->iters = 0;
        }
        return PT_OK;
    }

    float* copy_dst = src ? const_cast<float*>(feat) + (long)slot * stride_n : nullptr;
    // (a pending last update of the previous solve is applied HERE: it forms this pass's filter operand and lands in the filter)
    rc = pt_launch_corr2(f, feat, stride_n, w_in, a.spart, st, nullptr, src ? slot : -1, src, copy_dst, have_pending ? &lz : nullptr);
    if (rc) return rc;
    if (have_pending) pend->iters = 0;
    const int pw_threads = ((a.OO + 63) / 64) * 64;                 // one element per thread (OO <= 1024 on this path)
    const size_t lut_lds = a.kind == PT_SD_DIMP ? (size_t)3 * a.num_bins * sizeof(float) : 0;
    const bool pw2 = a.KS <= PT_PW_MAXKS && lut_lds <= 48 * 1024;
    // Round 6 (experiment O, LOST -- profiles/r06o_fused_init_stage.txt): with at least one iteration to run, no loss read-out and nobody
    // waiting on an event behind the init stage, the first adjoint pass can do the init stage itself (k_adj2<.., INIT>): one dependent
    // launch less, bit-identical -- but every one of the 32 channel-block workgroups of a position slice then redoes the maps of its
    // 6-7 samples (8 slice loads + the look-up-table arithmetic per element, six elements per lane): that pass is 15.3 us instead of
    // 7.9 and the frame 4.7 us SLOWER.  Opt-in for experiments: PT_SD_FUSE_INIT=1 in the environment (read per solve).
    const bool fuse_env = std::getenv("PT_SD_FUSE_INIT") && std::getenv("PT_SD_FUSE_INIT")[0] == '1';
    const bool fuse_init = PT_SD_FUSE_INIT && fuse_env && pw2 && num_iter > 0 && !want_loss && !(cls && cls->after_init) &&
                           pt_adj2_init_fusable(f, a);
    if (fuse_init) {
        if (n > 65535 || a.OO > 65535) return PT_ERR_UNSUPPORTED;
    } else if (pw2) {
        if (n > 65535 || a.OO > 65535 || a.OW > 4095 || a.K > 15 || a.num_bins > 65535 || n >= 0xfffff) return PT_ERR_UNSUPPORTED;
        InitLate il;
        il.label_lut = (pt_gcf)a.label_lut; il.mask_lut = (pt_gcf)a.mask_lut; il.spatial_lut = (pt_gcf)a.spatial_lut;
        il.s = (pt_gf)a.s; il.label = (pt_gf)a.label; il.mask = (pt_gf)a.mask; il.sws = (pt_gf)a.sws; il.lms = (pt_gf)a.lms;
        il.pk = (pt_gf)a.pk; il.cls_scores = (pt_gf)a.cls_scores; il.cls_peak = (pt_gf)a.cls_peak; il.cls_bb = (pt_gf)a.cls_bb;
        il.OH = a.OH; il.score_act = a.score_act; il.mask_act = a.mask_act; il.normalize_label = a.normalize_label;
        il.bin_disp = a.bin_disp; il.gauss_sigma = a.gauss_sigma; il.hinge_thr = a.hinge_thr; il.uni_weight = a.uni_weight;
        il.label_shrink = a.label_shrink; il.label_thr = a.label_thr;
        const unsigned cslot = (a.cls_spart && a.cls_slot >= 0) ? (unsigned)a.cls_slot : 0xfffffu;
        hipLaunchKernelGGL(k_fast_init2, dim3(n), dim3(pw_threads), lut_lds, st, (const float*)a.spart, a.bb, a.has_sw ? a.sw : (const float*)nullptr,
                           a.cls_slot_dyn,
                           (PT_INIT_LUT_HOT && a.kind == PT_SD_DIMP && a.mask_lut == a.label_lut + a.num_bins &&
                            a.spatial_lut == a.label_lut + 2 * a.num_bins) ? a.label_lut : (const float*)nullptr,
                           (unsigned)n | ((unsigned)a.OO << 16), (unsigned)a.KS | ((unsigned)a.kind << 8) | (cslot << 12),
                           (unsigned)a.OW | ((unsigned)a.K << 12) | ((unsigned)(a.kind == PT_SD_DIMP ? a.num_bins : 0) << 16), a.feat_stride, il);
    }
    else if (a.cls_slot_dyn) return PT_ERR_UNSUPPORTED;             // (refused before the first correlation: see below)
    else hipLaunchKernelGGL(k_fast_init, dim3(n), dim3(384), 0, st, a);
    if (!fuse_init) PT_CHECK_LAUNCH();
    if (cls && cls->after_init && hipEventRecord((hipEvent_t)cls->after_init, st) != hipSuccess) return PT_ERR_LAUNCH;
    if (num_iter == 0) {
        if (!want_loss) return PT_OK;
        a.R = nullptr;
        hipLaunchKernelGGL(k_sd_pw, dim3(n), dim3(512), pw_lds, st, a, (int)PW_INIT, 0, 1, 1);
        PT_CHECK_LAUNCH();
    }
    for (int t = 0; t < num_iter; ++t) {
        a.s_in = sbuf[t == 0 ? 0 : (t - 1) & 1];
        a.s = sbuf[t & 1];
        rc = pt_launch_adj2_sd(f, feat, stride_n, a, t, want_loss, st, fuse_init && t == 0);   // alpha_{t-1}, w_t, s_t, residual maps
        if (rc) return rc;
        PtCorrFuse fz = {a.gpart, f.KSPL, t == 0 ? w_in : w_iters + (long)t * a.CKK, a.reg, a.g, a.anum, nullptr};
        rc = pt_launch_corr2(f, feat, stride_n, nullptr, a.spart, st, &fz);    // g_t, |g_t|^2, F g_t
        if (rc) return rc;
        if (pw2) {
            const bool pr = a.kind == PT_SD_PRDIMP;
            const SgqLate sl = {(pt_gf)a.sg, (pt_gf)a.pk, (pt_gf)a.qs, (pt_gcf)a.sw};
            hipLaunchKernelGGL(k_fast_sgq2, dim3(n), dim3(pw_threads), 0, st, (const float*)a.spart, (const float*)a.s,
                               (const float*)(pr ? a.mask : a.lms), (const float*)(pr ? a.label : nullptr),
                               (unsigned)n | ((unsigned)a.OO << 16),
                               (unsigned)a.KS | ((unsigned)a.kind << 8) | ((unsigned)a.score_act << 12) | ((unsigned)(a.has_sw ? 1 : 0) << 16),
                               a.act_param, sl);
        }
        else hipLaunchKernelGGL(k_fast_sgq, dim3(n), dim3(384), pw_lds, st, a);
        PT_CHECK_LAUNCH();
    }
    if (num_iter > 1 && defer) {
        // the last update stays pending: (w_{T-1}, g_T, qs, anum) are in the workspace, the next call of the chain applies it
        pend->iters = num_iter; pend->step_length = a.step; pend->reg_eps = a.reg + a.alpha_eps;
    } else if (num_iter > 0) {
        static const bool final_old = std::getenv("PT_SD_FINAL_OLD") != nullptr;     // experiment: the struct-argument kernel
        if (!want_loss && !final_old)
            hipLaunchKernelGGL(k_fast_final, dim3(n), dim3(512), 0, st, sd_w_host(a, num_iter - 1), (const float*)a.g,
                               a.w_final ? a.w_final : a.w_iters + (long)num_iter * a.CKK, (const float*)a.qs, (const float*)a.anum, n, a.CKK,
                               a.KS, a.step, a.reg + a.alpha_eps);
        else hipLaunchKernelGGL(k_sd_pw, dim3(n), dim3(512), pw_lds, st, a, (int)PW_UPDATE, num_iter, 1, want_loss);
        PT_CHECK_LAUNCH();
    }
    if (want_loss) {
        hipLaunchKernelGGL(k_sd_loss, dim3(num_iter + 1), dim3(256), 0, st, a, losses);
        PT_CHECK_LAUNCH();
    }
    return PT_OK;
}

// Internal entry shared by pt_sd_solve_f32 and pt_track_frame_f32.
//   copy_w0  : also materialise iterate 0 in w_iters[0] (public API contract)
//   w_final  : if non-null the last iterate is written there (may alias w_in) instead of w_iters[T]
//   cls      : optional classification epilogue (see SdArgs), run inside the maps launch
int pt_sd_solve_impl(const pt_sd_params* prm, const float* w_in, const float* feat, long feat_stride_n, const float* bb,
                     const float* sample_weight, int n, int C, int H, int W, int K, int num_iter, float* w_iters,
                     float* losses, void* ws, size_t ws_bytes, hipStream_t st, bool copy_w0, float* w_final,
                     const PtClsFin* cls, const float* src, pt_frame_pending* pend, bool defer) {
    if (!prm || !w_in || !feat || !bb || !w_iters || !ws) return PT_ERR_NULL;
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || num_iter < 0) return PT_ERR_SHAPE;
    if (K * K > 16 || num_iter > PT_SD_MAX_ITER) return PT_ERR_UNSUPPORTED;
    if (feat_stride_n < (long)C * H * W) return PT_ERR_SHAPE;
    if (prm->kind == PT_SD_DIMP && (!prm->label_lut || !prm->mask_lut || !prm->spatial_lut || prm->num_bins < 1))
        return PT_ERR_NULL;
    if (prm->kind < PT_SD_DIMP || prm->kind > PT_SD_PRDIMP) return PT_ERR_UNSUPPORTED;
    const int OH = H + (K + 1) % 2, OW = W + (K + 1) % 2;               // optimizer.py:105
    {
        PtFast f = pt_fast_plan(n, C, H, W, K, K, OH, OW);
        if (pt_fast_usable(f, feat, feat_stride_n, w_in, src) && ((uintptr_t)w_iters % 16) == 0 && f.KSPL <= 16)
            return sd_solve_fast(f, prm, w_in, feat, feat_stride_n, bb, sample_weight, num_iter, w_iters, losses, ws,
                                 ws_bytes, st, copy_w0, w_final, cls, src, pend, defer);
    }
    if (defer || (pend && pend->iters > 0)) return PT_ERR_UNSUPPORTED;   // frame chains exist on the fast path only
    if (src) return PT_ERR_UNSUPPORTED;                                 // source override exists on the fast path only
    PtPlan p = pt_make_plan(n, C, H, W, K, K, OH, OW);
    if (p.KS > 64) return PT_ERR_UNSUPPORTED;
    SdCarve cv = sd_carve(p, PT_SD_MAX_ITER);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    float* base = (float*)ws;

    SdArgs a;
    a.n = n; a.C = C; a.H = H; a.W = W; a.K = K; a.OH = OH; a.OW = OW; a.OO = OH * OW; a.CKK = C * K * K;
    a.KS = p.KS; a.KSPL = p.KSPL;
    a.kind = prm->kind; a.score_act = prm->score_act; a.mask_act = prm->mask_act; a.has_sw = sample_weight != nullptr;
    a.has_softmax_reg = prm->has_softmax_reg; a.normalize_label = prm->normalize_label; a.num_bins = prm->num_bins;
    a.step = prm->step_length; a.reg = prm->reg; a.alpha_eps = prm->alpha_eps; a.feat_stride = prm->feat_stride;
    a.bin_disp = prm->bin_displacement; a.act_param = prm->act_param; a.gauss_sigma = prm->gauss_sigma;
    a.hinge_thr = prm->hinge_threshold; a.uni_weight = prm->uni_weight; a.label_shrink = prm->label_shrink;
    a.softmax_reg = prm->softmax_reg; a.label_thr = prm->label_threshold;
    a.bb = bb; a.sw = sample_weight; a.label_lut = prm->label_lut; a.mask_lut = prm->mask_lut;
    a.spatial_lut = prm->spatial_lut;
    a.label = base + cv.label; a.mask = base + cv.mask; a.sws = base + cv.sws; a.s = base + cv.s; a.sg = base + cv.sg;
    a.spart = base + cv.spart; a.R = base + cv.R; a.gpart = base + cv.gpart; a.g = base + cv.g;
    a.anum = base + cv.anum; a.qs = base + cv.qs; a.lossp = base + cv.lossp; a.w_iters = w_iters;
    a.w0 = w_in; a.w_final = w_final;
    a.s_in = nullptr; a.lms = nullptr; a.pk = nullptr; a.cls_stride = 0;
    a.cls_spart = nullptr; a.cls_KS = 0; a.cls_slot = -1; a.cls_scores = nullptr; a.cls_peak = nullptr; a.cls_bb = nullptr;
    a.cls_slot_dyn = nullptr;
    if (cls && cls->slot_dyn) return PT_ERR_UNSUPPORTED;                // a device-resident slot exists on the fast path only
    if (cls) {
        a.cls_spart = cls->spart; a.cls_KS = cls->KS; a.cls_stride = a.OO; a.cls_slot = cls->slot; a.cls_scores = cls->scores;
        a.cls_peak = cls->peak; a.cls_bb = cls->mem_bb;
    }

    const int want_loss = losses != nullptr;
    const size_t pw_lds = (size_t)a.OO * sizeof(float) * (a.kind == PT_SD_PRDIMP ? 2 : 1);

    if (copy_w0 && w_iters != w_in) {
        if (hipMemcpyAsync(w_iters, w_in, (size_t)a.CKK * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return PT_ERR_LAUNCH;
    }
    if (num_iter == 0 && !want_loss && !cls) return PT_OK;

    hipLaunchKernelGGL(k_sd_maps, dim3(n), dim3(256), 0, st, a);
    PT_CHECK_LAUNCH();
    if (cls && cls->after_init && hipEventRecord((hipEvent_t)cls->after_init, st) != hipSuccess) return PT_ERR_LAUNCH;
    if (num_iter == 0 && !want_loss) return PT_OK;
    int rc = pt_launch_corr(p, feat, feat_stride_n, w_in, a.spart, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sd_pw, dim3(n), dim3(512), pw_lds, st, a, (int)PW_INIT, 0, (int)(num_iter == 0), want_loss);
    PT_CHECK_LAUNCH();
    for (int t = 0; t < num_iter; ++t) {
        rc = pt_launch_adj(p, feat, feat_stride_n, a.R, a.gpart, st);
        if (rc) return rc;
        // g_t = sum_k gpart[k] + reg*w_t is reduced in the prologue of the correlation pass (optimizer.py:146-151)
        PtCorrFuse fz = {a.gpart, p.KSPL, t == 0 ? w_in : w_iters + (long)t * a.CKK, a.reg, a.g, a.anum, nullptr};
        rc = pt_launch_corr(p, feat, feat_stride_n, nullptr, a.spart, st, &fz);
        if (rc) return rc;
        hipLaunchKernelGGL(k_sd_pw, dim3(n), dim3(512), pw_lds, st, a, (int)PW_SGQ, t, 0, 0);
        PT_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_sd_pw, dim3(n), dim3(512), pw_lds, st, a, (int)PW_UPDATE, t + 1, (int)(t + 1 == num_iter),
                           want_loss);
        PT_CHECK_LAUNCH();
    }
    if (want_loss) {
        hipLaunchKernelGGL(k_sd_loss, dim3(num_iter + 1), dim3(256), 0, st, a, losses);
        PT_CHECK_LAUNCH();
    }
    return PT_OK;
}

int pt_sd_flush_impl(const pt_frame_pending* pend, float* filter, int n, int C, int H, int W, int K, float* w_iters, void* ws,
                     size_t ws_bytes, hipStream_t st) {
    if (!pend || !filter || !w_iters || !ws) return PT_ERR_NULL;
    if (pend->iters == 0) return PT_OK;
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return PT_ERR_SHAPE;
    const int OH = H + (K + 1) % 2, OW = W + (K + 1) % 2;
    PtFast f = pt_fast_plan(n, C, H, W, K, K, OH, OW);
    if (!f.ok) return PT_ERR_UNSUPPORTED;
    FastCarve cv = fast_carve(f, PT_SD_MAX_ITER);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    if (pend->iters < 2 || pend->iters > PT_SD_MAX_ITER) return PT_ERR_SHAPE;
    float* base = (float*)ws;
    const int CKK = C * K * K;
    hipLaunchKernelGGL(k_fast_final, dim3(n), dim3(512), 0, st, (const float*)(w_iters + (long)(pend->iters - 1) * CKK),
                       (const float*)(base + cv.g), filter, (const float*)(base + cv.qs), (const float*)(base + cv.anum), n, CKK, f.KSC,
                       pend->step_length, pend->reg_eps);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" int pt_sd_solve_f32(const pt_sd_params* prm, const float* w_in, const float* feat, long feat_stride_n,
                               const float* bb, const float* sample_weight, int n, int C, int H, int W, int K,
                               int num_iter, float* w_iters, float* losses, void* ws, size_t ws_bytes, void* stream) {
    return pt_sd_solve_impl(prm, w_in, feat, feat_stride_n, bb, sample_weight, n, C, H, W, K, num_iter, w_iters, losses,
                            ws, ws_bytes, (hipStream_t)stream, /*copy_w0=*/true, /*w_final=*/nullptr, /*cls=*/nullptr,
                            /*src=*/nullptr);
}

// S independent sequences in ONE call (optimizer.py:101-104 `num_sequences`: the reference carries them as a tensor dimension of
// every op).  A single-sequence solve is a chain of dependent, latency-bound launches that leaves most of the chip idle (bench.py
// `multi_sequence`: two sequences on two streams -> 1.3x, four -> 1.5x aggregate), so the sequences are spread round-robin over
// `stream` and the caller's `n_aux` auxiliary streams: every auxiliary stream first waits for what is queued on `stream` (fork
// event), and `stream` waits for all of them before the call returns (join events) -- to the caller the call is ordered on `stream`
// like the single-sequence entry.  Sequences that share a stream run back to back; each needs its own workspace.
extern "C" int pt_sd_solve_batch_f32(const pt_sd_params* prm, int S, const float* const* w_in, const float* const* feat,
                                     long feat_stride_n, const float* const* bb, const float* const* sample_weight, int n, int C,
                                     int H, int W, int K, int num_iter, float* const* w_iters, float* const* losses,
                                     void* const* ws, size_t ws_bytes_each, void* stream, void* const* aux_streams, int n_aux) {
    if (!prm || !w_in || !feat || !bb || !w_iters || !ws) return PT_ERR_NULL;
    if (S <= 0 || n_aux < 0 || (n_aux > 0 && !aux_streams)) return PT_ERR_SHAPE;
    if (n_aux > 15) return PT_ERR_UNSUPPORTED;
    const int lanes = 1 + (S > 1 ? n_aux : 0);
    hipEvent_t fork = nullptr, join[16] = {};
    if (lanes > 1) {
        // every auxiliary stream is validated and has its events BEFORE the fork is recorded: a bad later entry must not leave earlier
        // streams forked and never joined (inside a graph capture that would invalidate the capture)
        for (int l = 1; l < lanes && l <= S - 1; ++l) {
            if (!aux_streams[l - 1] || aux_streams[l - 1] == stream) return PT_ERR_SHAPE;
            for (int m = 1; m < l; ++m)
                if (aux_streams[m - 1] == aux_streams[l - 1]) return PT_ERR_SHAPE;
            if (!pt_stream_events(stream, aux_streams[l - 1], nullptr, &join[l])) return PT_ERR_LAUNCH;
        }
        if (!pt_stream_events(stream, nullptr, &fork, nullptr)) return PT_ERR_LAUNCH;
        if (hipEventRecord(fork, (hipStream_t)stream) != hipSuccess) return PT_ERR_LAUNCH;
        for (int l = 1; l < lanes && l <= S - 1; ++l)
            if (hipStreamWaitEvent((hipStream_t)aux_streams[l - 1], fork, 0) != hipSuccess) return PT_ERR_LAUNCH;
    }
    int rc = PT_OK;
    for (int s = 0; s < S && rc == PT_OK; ++s) {
        const int l = s % lanes;
        void* st = l == 0 ? stream : aux_streams[l - 1];
        rc = pt_sd_solve_impl(prm, w_in[s], feat[s], feat_stride_n, bb[s], sample_weight ? sample_weight[s] : nullptr, n, C, H, W, K,
                              num_iter, w_iters[s], losses ? losses[s] : nullptr, ws[s], ws_bytes_each, (hipStream_t)st,
                              /*copy_w0=*/true, /*w_final=*/nullptr, /*cls=*/nullptr, /*src=*/nullptr);
    }
    // join even after an error: work that was queued on an auxiliary stream must not outlive the call's ordering contract
    for (int l = 1; l < lanes && l <= S - 1; ++l) {
        if (hipEventRecord(join[l], (hipStream_t)aux_streams[l - 1]) != hipSuccess ||
            hipStreamWaitEvent((hipStream_t)stream, join[l], 0) != hipSuccess)
            return rc ? rc : PT_ERR_LAUNCH;
    }
    return rc;
}

// ----------------------------------------------------------------------------------------------------
// Measurement helper behind pt_track_frame_replay_pass_f32 (api.hip; bench.py roofline leg): re-issue ONE feature pass of the solve
// that last ran on this workspace, `reps` times back to back on `stream`, exactly as iteration t = num_iter - 1 of
// pt_track_frame_f32 / pt_sd_solve_f32 launched it (same kernel instantiation, same operands; both passes only read
// the state they were launched on, so re-issuing them is idempotent).  The caller brackets the call with ONE event pair.
//   which = 0: correlation pass with the fused gradient reduction (k_corr2<..., FUSE>)
//   which = 1: adjoint pass with the fused update prologue        (k_adj2<V, ...>)
// ----------------------------------------------------------------------------------------------------
int pt_sd_replay_impl(const pt_sd_params* prm, const float* w_in, const float* feat, long feat_stride_n, const float* bb,
                      const float* sample_weight, int n, int C, int H, int W, int K, int num_iter, float* w_iters, void* ws,
                      size_t ws_bytes, int which, int reps, hipStream_t stream) {
    if (!prm || !w_in || !feat || !bb || !w_iters || !ws) return PT_ERR_NULL;
    if (num_iter < 2 || reps < 1 || which < 0 || which > 1) return PT_ERR_SHAPE;
    const int OH = H + (K + 1) % 2, OW = W + (K + 1) % 2;
    PtFast f = pt_fast_plan(n, C, H, W, K, K, OH, OW);
    if (!pt_fast_usable(f, feat, feat_stride_n, w_in) || f.KSPL > 16) return PT_ERR_UNSUPPORTED;
    SdArgs a;
    float* sbuf[2];
    int rc = sd_fast_setup(f, prm, w_in, bb, sample_weight, w_iters, nullptr, ws, ws_bytes, a, sbuf);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int t = num_iter - 1;
    a.s_in = sbuf[(t - 1) & 1];
    a.s = sbuf[t & 1];
    for (int r = 0; r < reps; ++r) {
        if (which == 1) {
            rc = pt_launch_adj2_sd(f, feat, feat_stride_n, a, t, 0, st);
        } else {
            PtCorrFuse fz = {a.gpart, f.KSPL, w_iters + (long)t * a.CKK, a.reg, a.g, a.anum, nullptr};
            rc = pt_launch_corr2(f, feat, feat_stride_n, nullptr, a.spart, st, &fz);
        }
        if (rc) return rc;
    }
    return PT_OK;
}
