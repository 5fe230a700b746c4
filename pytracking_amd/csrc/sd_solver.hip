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
#include "common.h"
#include "pt_internal.h"
#include "rbuild.h"

enum { PW_INIT = 0, PW_SGQ = 1, PW_UPDATE = 2 };

struct SdArgs {
    // problem
    int n, C, H, W, K, OH, OW, OO, CKK, KS, KSPL;
    int kind, score_act, mask_act, has_sw, has_softmax_reg, normalize_label, num_bins;
    float step, reg, alpha_eps, feat_stride, bin_disp, act_param, gauss_sigma, hinge_thr;
    float uni_weight, label_shrink, softmax_reg, label_thr;
    const float *bb, *sw, *label_lut, *mask_lut, *spatial_lut;
    // workspace
    float *label, *mask, *sws;   // (n,OO) maps.  PrDiMP: mask holds the softmax P, sws unused
    float *s, *sg;               // (n,OO) scores of the current iterate, F g
    float *spart;                // (KS,n,OO)
    float *R;                    // (NG,256)
    float *gpart, *g;            // (KSPL,CKK), (CKK)
    float *anum;                 // (KS) per-channel-slice |g|^2 (written by the corr(g) pass)
    float *qs;                   // (n)
    float *lossp;                // (T+1, n)
    float *w_iters;              // (T+1, CKK)  caller's buffer; iterate 0 lives at w0
    const float *w0;             // initial filter
    float *w_final;              // optional: the last iterate is written here instead of w_iters[T]
    // optional classification epilogue run by the workgroup of sample `cls_slot` before its maps
    // (benchmark frame: sum the classify partials, arg-max, re-centre that sample's box)
    const float *cls_spart;
    int cls_KS, cls_slot;
    float *cls_scores, *cls_peak, *cls_bb;
};

__device__ __forceinline__ const float* sd_w(const SdArgs& a, int t) {
    return t == 0 ? a.w0 : a.w_iters + (long)t * a.CKK;
}

// ----------------------------------------------------------------------------------------------------
// maps: one workgroup per sample
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pl_lut(const float* __restrict__ w, int bins, float t) {
    // DistanceMap (ltr/models/layers/distance.py:17-39) followed by a 1x1 conv over the bins is the
    // piecewise-linear interpolation of the conv weights at t = d / bin_displacement, constant past the last bin.
    const int k0 = (int)floorf(t);
    if (k0 >= bins - 1) return w[bins - 1];
    const float fr = t - (float)k0;
    return w[k0] * (1.0f - fr) + w[k0 + 1] * fr;
}

// sums the classification partials, finds the first maximum (torch.max semantics, pytracking/libs/dcf.py:156-164)
// and re-centres the box of memory slot `cls_slot` on it (inverse of the centre formula of optimizer.py:112-113).
__device__ void sd_classify_fin(const SdArgs& a) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= a.cls_KS; k += 8) {                 // independent loads in flight, fixed summation order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = a.cls_spart[(long)(k + q) * a.OO + o];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; k < a.cls_KS; ++k) s += a.cls_spart[(long)k * a.OO + o];
        a.cls_scores[o] = s;
        if (s > best) { best = s; besti = o; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(besti, off, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        const int row = besti / a.OW, col = besti - row * a.OW;
        a.cls_peak[0] = (float)row;
        a.cls_peak[1] = (float)col;
        const float off = (float)(a.K % 2) * 0.5f;
        float* b = a.cls_bb + 4 * a.cls_slot;
        b[0] = ((float)col + off) * a.feat_stride - b[2] * 0.5f;
        b[1] = ((float)row + off) * a.feat_stride - b[3] * 0.5f;
    }
    __syncthreads();
}

__global__ void k_sd_maps(SdArgs a) {
    __shared__ float scratch[16];
    __shared__ int amin[2];
    const int i = blockIdx.x;
    if (a.cls_spart && i == a.cls_slot) sd_classify_fin(a);     // uniform per workgroup
    const float off = (float)(a.K % 2) * 0.5f;
    const float* b = a.bb + 4 * i;
    const float ctr_r = (b[1] + b[3] * 0.5f) / a.feat_stride - off;     // optimizer.py:112-113 (flip -> row first)
    const float ctr_c = (b[0] + b[2] * 0.5f) / a.feat_stride - off;
    float* label = a.label + (long)i * a.OO;
    if (a.kind == PT_SD_DIMP) {
        float* mask = a.mask + (long)i * a.OO;
        float* sws = a.sws + (long)i * a.OO;
        const float swi = a.has_sw ? sqrtf(a.sw[i]) : sqrtf(1.0f / (float)a.n);   // :122-125
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            const int y = o / a.OW, x = o - y * a.OW;
            const float d0 = (float)y - ctr_r, d1 = (float)x - ctr_c;
            const float t = sqrtf(d0 * d0 + d1 * d1) / a.bin_disp;
            label[o] = pl_lut(a.label_lut, a.num_bins, t);
            float m = pl_lut(a.mask_lut, a.num_bins, t);
            if (a.mask_act == PT_MASK_SIGMOID) m = 1.0f / (1.0f + expf(-m));
            mask[o] = m;
            sws[o] = swi * pl_lut(a.spatial_lut, a.num_bins, t);
        }
    } else if (a.kind == PT_SD_DIMP_L2) {
        float* mask = a.mask + (long)i * a.OO;
        float* sws = a.sws + (long)i * a.OO;
        const float swi = a.has_sw ? sqrtf(a.sw[i]) : sqrtf(1.0f / (float)a.n);   // :249-252
        const float coef = -1.0f / (2.0f * a.gauss_sigma * a.gauss_sigma);
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            const int y = o / a.OW, x = o - y * a.OW;
            const float d0 = (float)y - ctr_r, d1 = (float)x - ctr_c;
            const float gss = expf(coef * d0 * d0) * expf(coef * d1 * d1);       // :201-208
            const float m = gss > a.hinge_thr ? 1.0f : 0.0f;                      // :245
            label[o] = gss * m;
            mask[o] = m;
            sws[o] = swi;
        }
    } else {   // PrDiMP label density, optimizer.py:331-353
        if (a.gauss_sigma == 0.f && threadIdx.x == 0) {
            int b0 = 0, b1 = 0;
            float m0 = INFINITY, m1 = INFINITY;
            for (int y = 0; y < a.OH; ++y) { float d = ((float)y - ctr_r); d *= d; if (d < m0) { m0 = d; b0 = y; } }
            for (int x = 0; x < a.OW; ++x) { float d = ((float)x - ctr_c); d *= d; if (d < m1) { m1 = d; b1 = x; } }
            amin[0] = b0; amin[1] = b1;
        }
        __syncthreads();
        const float s2 = a.gauss_sigma * a.gauss_sigma;
        const float coef = -1.0f / (2.0f * s2);
        float part = 0.f;
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x) {
            const int y = o / a.OW, x = o - y * a.OW;
            float gss;
            if (a.gauss_sigma == 0.f) {
                gss = (y == amin[0] && x == amin[1]) ? 1.0f : 0.0f;
            } else {
                const float d0 = (float)y - ctr_r, d1 = (float)x - ctr_c;
                gss = (expf(coef * d0 * d0) / (2.0f * 3.14159265358979323846f * s2)) * expf(coef * d1 * d1);
            }
            gss = gss > a.label_thr ? gss : 0.f;
            label[o] = gss;
            part += gss;
        }
        const float tot = block_sum(part, scratch);
        const float inv = a.normalize_label ? 1.0f / (tot + 1e-8f) : 1.0f;
        const float uni = a.uni_weight / (float)a.OO;
        for (int o = threadIdx.x; o < a.OO; o += blockDim.x)
            label[o] = (1.0f - a.label_shrink) * ((1.0f - a.uni_weight) * (label[o] * inv) + uni);
    }
}

// ----------------------------------------------------------------------------------------------------
// pointwise stages: one workgroup per sample
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void act_pair(int score_act, float bpar, float x, float am, float& act, float& der) {
    // activation.py:32-66.  score_act 2 = the L2 hinge of optimizer.py:262-263 (mask in {0,1}).
    if (score_act == PT_ACT_RELU) {
        const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
        act = (1.0f - am) * 0.5f * fabsf(x) + (1.0f + am) * 0.5f * x;
        der = (1.0f - am) * 0.5f * sgn + (1.0f + am) * 0.5f;
    } else if (score_act == PT_ACT_BENTPAR) {
        const float rt = sqrtf(x * x + 4.0f * bpar * bpar);
        act = (1.0f - am) * 0.5f * (rt - 2.0f * bpar) + (1.0f + am) * 0.5f * x;
        der = (1.0f - am) * 0.5f * (x / rt) + (1.0f + am) * 0.5f;
    } else {
        act = am * x + (1.0f - am) * fmaxf(x, 0.f);
        der = am + (1.0f - am) * (x > 0.f ? 1.f : 0.f);
    }
}

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

__device__ __forceinline__ float sd_alpha_step(const SdArgs& a) {
    // optimizer.py:155-160 / :425-430: alpha = |g|^2 / max(sum_i q_i + (reg+eps)|g|^2, 1e-8), times the step length
    float den = 0.f;
    for (int k = 0; k < a.n; ++k) den += a.qs[k];
    float a_num = 0.f;
    for (int k = 0; k < a.KS; ++k) a_num += a.anum[k];
    den = fmaxf(den + (a.reg + a.alpha_eps) * a_num, 1e-8f);
    return a.step * (a_num / den);
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
        astep = sd_alpha_step(a);
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

#define PT_SD_MAX_ITER 64

extern "C" size_t pt_sd_ws_bytes(int n, int C, int H, int W, int K) {
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
    const int OH = H + (K + 1) % 2, OW = W + (K + 1) % 2;
    PtPlan p = pt_make_plan(n, C, H, W, K, K, OH, OW);
    return sd_carve(p, PT_SD_MAX_ITER).total * sizeof(float);
}

// Internal entry shared by pt_sd_solve_f32 and pt_track_frame_f32.
//   copy_w0  : also materialise iterate 0 in w_iters[0] (public API contract)
//   w_final  : if non-null the last iterate is written there (may alias w_in) instead of w_iters[T]
//   cls      : optional classification epilogue (see SdArgs), run inside the maps launch
int pt_sd_solve_impl(const pt_sd_params* prm, const float* w_in, const float* feat, long feat_stride_n, const float* bb,
                     const float* sample_weight, int n, int C, int H, int W, int K, int num_iter, float* w_iters,
                     float* losses, void* ws, size_t ws_bytes, hipStream_t st, bool copy_w0, float* w_final,
                     const PtClsFin* cls) {
    if (!prm || !w_in || !feat || !bb || !w_iters || !ws) return PT_ERR_NULL;
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || num_iter < 0) return PT_ERR_SHAPE;
    if (K * K > 16 || num_iter > PT_SD_MAX_ITER) return PT_ERR_UNSUPPORTED;
    if (feat_stride_n < (long)C * H * W) return PT_ERR_SHAPE;
    if (prm->kind == PT_SD_DIMP && (!prm->label_lut || !prm->mask_lut || !prm->spatial_lut || prm->num_bins < 1))
        return PT_ERR_NULL;
    if (prm->kind < PT_SD_DIMP || prm->kind > PT_SD_PRDIMP) return PT_ERR_UNSUPPORTED;
    const int OH = H + (K + 1) % 2, OW = W + (K + 1) % 2;               // optimizer.py:105
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
    a.cls_spart = nullptr; a.cls_KS = 0; a.cls_slot = -1; a.cls_scores = nullptr; a.cls_peak = nullptr; a.cls_bb = nullptr;
    if (cls) {
        a.cls_spart = cls->spart; a.cls_KS = cls->KS; a.cls_slot = cls->slot; a.cls_scores = cls->scores;
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

extern "C" int pt_sd_solve_f32(const pt_sd_params* prm, const float* w_in, const float* feat, long feat_stride_n,
                               const float* bb, const float* sample_weight, int n, int C, int H, int W, int K,
                               int num_iter, float* w_iters, float* losses, void* ws, size_t ws_bytes, void* stream) {
    return pt_sd_solve_impl(prm, w_in, feat, feat_stride_n, bb, sample_weight, n, C, H, W, K, num_iter, w_iters, losses,
                            ws, ws_bytes, (hipStream_t)stream, /*copy_w0=*/true, /*w_final=*/nullptr, /*cls=*/nullptr);
}
