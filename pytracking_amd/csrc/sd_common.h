// Shared pieces of the steepest-descent solvers (reference: ltr/models/target_classifier/optimizer.py): argument block,
// radial look-up tables, score activations, label/mask/weight maps, classification epilogue.
#pragma once
#include "common.h"
#include "pt_internal.h"

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
    const float *s_in;           // fast path: scores of iterate t-1 (ping-pong with s)
    float *lms;                  // fast path: (n,OO,4) packed {label, mask, sws, 0} (one 16-byte load per element)
    float *pk;                   // fast path: (n,OO,4) packed update-stage operands (see PReg in fast_passes.hip)
    float *spart;                // (KS,n,OO)
    float *R;                    // (NG,256)
    float *gpart, *g;            // (KSPL,CKK), (CKK)
    float *anum;                 // (KS) per-channel-slice |g|^2 (written by the corr(g) pass)
    float *qs;                   // (n) per-sample curvature terms q_i
    float *lossp;                // (T+1, n)
    float *w_iters;              // (T+1, CKK)  caller's buffer; iterate 0 lives at w0
    const float *w0;             // initial filter
    float *w_final;              // optional: the last iterate is written here instead of w_iters[T]
    // optional classification epilogue run by the workgroup of sample `cls_slot` before its maps
    // (benchmark frame: sum the classify partials, arg-max, re-centre that sample's box)
    const float *cls_spart;
    int cls_KS, cls_slot;
    long cls_stride;             // floats between consecutive classification slices
    float *cls_scores, *cls_peak, *cls_bb;
    const int* cls_slot_dyn;     // optional: cls_slot is read from this device int (k_fast_init2 only; graph-replayed frames)
};

// (returns the member's own pointer type: plain in the argument structs passed by value, global-qualified in the late-fetched ones)
template <typename A>
__device__ __forceinline__ auto sd_w(const A& a, int t) -> decltype(a.w0) {
    return t == 0 ? a.w0 : (decltype(a.w0))(a.w_iters + (long)t * a.CKK);
}

// ----------------------------------------------------------------------------------------------------
// maps: one workgroup per sample
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pl_lut(const float* __restrict__ w, int bins, float t) {
    // DistanceMap (ltr/models/layers/distance.py:17-39) followed by a 1x1 conv over the bins is the
    // piecewise-linear interpolation of the conv weights at t = d / bin_displacement, constant past the last bin.
    // Branch-free (round 6): both table reads are unconditional (clamped indices) and the constant tail is a select, so that a caller
    // evaluating several tables / elements gets all its reads in flight together instead of branch -> read -> wait per table.  Inside
    // the table the indices and the expression are the ones above: same bits.
    const int k0 = (int)floorf(t);
#if defined(PT_PL_LUT_BRANCHY) && PT_PL_LUT_BRANCHY                 // A/B builds: the first form
    if (k0 >= bins - 1) return w[bins - 1];
    const float fr = t - (float)k0;
    return w[k0] * (1.0f - fr) + w[k0 + 1] * fr;
#else
    const int i0 = max(min(k0, bins - 2), 0), i1 = min(i0 + 1, bins - 1);
    const float w0 = w[i0], w1 = w[i1];
    const float fr = t - (float)i0;
    const float v = w0 * (1.0f - fr) + w1 * fr;
    return k0 >= bins - 1 ? w1 : v;                                 // (i1 == bins - 1 there)
#endif
}

// Label / mask / sample-weight map values of ONE element of the DiMP kinds (optimizer.py:111-125 / :201-208, :245, :249-252), given its
// offset (d0, d1) from the target centre and the sample weight swv.  Shared by the init stage (k_fast_init2) and by the first adjoint
// pass when it builds the maps itself (k_adj2<.., INIT>, fast_passes.hip): one body, so the two produce the same bits.
__device__ __forceinline__ void sd_init_elem_dimp(const float* __restrict__ lut, int num_bins, int mask_act, float bin_disp, float d0, float d1,
                                                  float swv, float& lb, float& m, float& sw) {
    const float t = sqrtf(d0 * d0 + d1 * d1) / bin_disp;
    lb = pl_lut(lut, num_bins, t);
    m = pl_lut(lut + num_bins, num_bins, t);
    if (mask_act == PT_MASK_SIGMOID) m = 1.0f / (1.0f + expf(-m));
    sw = sqrtf(swv) * pl_lut(lut + 2 * num_bins, num_bins, t);                      // :122-125
}
__device__ __forceinline__ void sd_init_elem_l2(float gauss_sigma, float hinge_thr, float d0, float d1, float swv, float& lb, float& m, float& sw) {
    const float coef = -1.0f / (2.0f * gauss_sigma * gauss_sigma);
    const float gss = expf(coef * d0 * d0) * expf(coef * d1 * d1);                  // :201-208
    m = gss > hinge_thr ? 1.0f : 0.0f;                                              // :245
    lb = gss * m;
    sw = sqrtf(swv);                                                                // :249-252
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
            for (int q = 0; q < 8; ++q) v[q] = a.cls_spart[(long)(k + q) * a.cls_stride + o];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; k < a.cls_KS; ++k) s += a.cls_spart[(long)k * a.cls_stride + o];
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

// label / mask / spatial-weight maps of sample i (optimizer.py:111-125, 201-208, 331-353); whole workgroup.
__device__ void sd_maps_sample(const SdArgs& a, int i, float* scratch, int* amin) {
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
            const float lb = pl_lut(a.label_lut, a.num_bins, t);
            float m = pl_lut(a.mask_lut, a.num_bins, t);
            if (a.mask_act == PT_MASK_SIGMOID) m = 1.0f / (1.0f + expf(-m));
            const float sv = swi * pl_lut(a.spatial_lut, a.num_bins, t);
            label[o] = lb;
            mask[o] = m;
            sws[o] = sv;
            if (a.lms) ((f32x4*)a.lms)[(long)i * a.OO + o] = (f32x4){lb, m, sv, 0.f};
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
            if (a.lms) ((f32x4*)a.lms)[(long)i * a.OO + o] = (f32x4){gss * m, m, swi, 0.f};
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


// this lane's share of sum_i q_i (samples lane, lane + 64, ...), fixed order: head (sample `lane`, ONE predicated load that
// nothing waits for here -- the k_adj2 prologue keeps it in flight behind its feature loads) + tail (samples lane + 64, ...;
// only memories of more than 64 samples enter the loop).  Callers add head + tail.
struct SdQLane { float head, tail; };
__device__ __forceinline__ SdQLane sd_q_lane(const float* __restrict__ qs, int n, int lane) {
    SdQLane r = {0.f, 0.f};
    r.head = lane < n ? qs[lane] : 0.f;
    for (int k = lane + 64; k < n; k += 64) r.tail += qs[k];
    return r;
}

// The LAST update of a solve, w_T = w_{T-1} - step*alpha_T*g_T (optimizer.py:155-160), from the operands the last iteration left in
// the workspace: one wave, explicit fused multiply-adds -- so that k_fast_final and the deferred form in the prologue of the next
// frame's first correlation (k_corr2, FUSE < 0) produce the same bits.
__device__ __forceinline__ float sd_final_astep(float q_head, float q_tail, float an, float step, float reg_eps) {
    float den = wave_sum(q_head + q_tail);
    const float a_num = wave_sum(an);
    den = fmaxf(__builtin_fmaf(reg_eps, a_num, den), 1e-8f);
    return step * (a_num / den);
}
__device__ __forceinline__ float sd_final_apply(float w, float g, float astep) { return __builtin_fmaf(-astep, g, w); }

// optimizer.py:155-160 / :425-430: alpha = |g|^2 / max(sum_i q_i + (reg+eps)|g|^2, 1e-8), times the step length;
// from one wave (every lane gets it): wave-parallel fixed-order sums, identical in every workgroup.
__device__ __forceinline__ float sd_alpha_step_wave(const SdArgs& a, int lane) {
    const SdQLane ql = sd_q_lane(a.qs, a.n, lane);
    float den = wave_sum(ql.head + ql.tail);
    const float a_num = wave_sum(lane < a.KS ? a.anum[lane] : 0.f);
    den = fmaxf(den + (a.reg + a.alpha_eps) * a_num, 1e-8f);
    return a.step * (a_num / den);
}
