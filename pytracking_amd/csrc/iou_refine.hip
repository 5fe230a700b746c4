// IoU-guided box refinement on the device:
//   AtomIoUNet.predict_iou (modulation, PrRoIPool 5x5 @ 1/8 and 3x3 @ 1/16, two LinearBlocks, Linear -> IoU)
//                                                             ltr/models/bbreg/atom_iou_net.py:96-136, layers/blocks.py:23-36
//   DiMP.optimize_boxes_default / optimize_boxes_relative     pytracking/tracker/dimp/dimp.py:725-788
//   ATOM.optimize_boxes (per-proposal backtracking)           pytracking/tracker/atom/atom.py:758-836
//   rect_to_rel / rel_to_rect                                 ltr/data/bounding_box_utils.py:4-33
// The reference runs every refinement step as an autograd forward + backward (~60 launches, a host-visible graph per
// step, 5-10 steps per frame).  Here the gradient of the predicted IoU w.r.t. the box is written out: forward pools ->
// split-K GEMM on the matrix cores -> BatchNorm/ReLU/IoU head that also emits dIoU/d(pre-activation) -> GEMM with the
// transposed weights -> PrRoIPool coordinate gradient -> box update, ten launches per step, no host synchronisation
// until the boxes are read.
#include "common.h"
#include "pt_internal.h"
#include "mfma_gemm.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int P3 = 5, P4 = 3;                      // pooled sizes of prroi_pool3t / prroi_pool4t (atom_iou_net.py:31,41)
constexpr float S3 = 1.f / 8, S4 = 1.f / 16;       // their spatial scales

struct IouOff { size_t w3, b3, bn3, w4, b4, bn4, wp, bp, total; };
IouOff iou_layout(const pt_iou_dims* d) {
    IouOff o{};
    size_t c = 0;
    auto take = [&](size_t n) { size_t r = c; c += n; return r; };
    const size_t K3 = (size_t)d->C3 * P3 * P3, K4 = (size_t)d->C4 * P4 * P4;
    o.w3 = take(d->I3 * K3); o.b3 = take(d->I3); o.bn3 = take(4 * (size_t)d->I3);
    o.w4 = take(d->I4 * K4); o.b4 = take(d->I4); o.bn4 = take(4 * (size_t)d->I4);
    o.wp = take((size_t)d->I3 + d->I4); o.bp = take(1);
    o.total = c;
    return o;
}

int iou_check(const pt_iou_dims* d) {
    if (!d) return PT_ERR_NULL;
    if (d->C3 <= 0 || d->C4 <= 0 || d->I3 <= 0 || d->I4 <= 0 || d->H3 <= 0 || d->W3 <= 0 || d->H4 <= 0 || d->W4 <= 0)
        return PT_ERR_SHAPE;
    if ((d->C3 * P3 * P3) % 32 != 0 || (d->C4 * P4 * P4) % 32 != 0 || d->I3 % 32 != 0 || d->I4 % 32 != 0)
        return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

struct IouCarve {
    size_t rois, X3, X4, part3, part4, G3, G4, dX3, dX4, gr3, gr4, msc3, msc4, state, szn, prev, slen, pstep, total;
    int nz3, nz4;
};
constexpr int KSPLIT = 4;                           // K-steps of 64 per split-K slice of the forward GEMMs
constexpr int GSL = 16;                             // element slices of the PrRoIPool coordinate gradient per proposal
IouCarve iou_carve(const pt_iou_dims* d, int P) {
    IouCarve c{};
    const size_t K3 = (size_t)d->C3 * P3 * P3, K4 = (size_t)d->C4 * P4 * P4;
    c.nz3 = (int)(((K3 + 63) / 64 + KSPLIT - 1) / KSPLIT);
    c.nz4 = (int)(((K4 + 63) / 64 + KSPLIT - 1) / KSPLIT);
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += pt_align_floats(n); return r; };
    c.rois = take((size_t)P * 5); c.X3 = take(P * K3); c.X4 = take(P * K4);
    c.part3 = take((size_t)c.nz3 * P * d->I3); c.part4 = take((size_t)c.nz4 * P * d->I4);
    c.G3 = take((size_t)P * d->I3); c.G4 = take((size_t)P * d->I4); c.dX3 = take(P * K3); c.dX4 = take(P * K4);
    c.gr3 = take((size_t)P * GSL * 4); c.gr4 = take((size_t)P * GSL * 4); c.msc3 = take(K3); c.msc4 = take(K4);
    c.state = take((size_t)P * 4); c.szn = take(2);
    c.prev = take(P); c.slen = take((size_t)P * 4); c.pstep = take((size_t)P * 4);
    c.total = o;
    return c;
}

// (R, C) -> (C, R)
__global__ __launch_bounds__(256) void k_transpose(const float* in, float* out, int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < C && r < R) out[(long)c * R + r] = tile[tx][ty + 8 * i];
    }
}

struct SetupArgs {
    const float *boxes, *mod3, *mod4;
    float *state, *szn, *rois, *msc3, *msc4, *prev, *slen, *pstep;
    int P, K3, K4, relative;
    float step[4];
};

// per-column modulation of the pooled features, the optimisation variable (rect or relative) and the first rois
__global__ __launch_bounds__(256) void k_iou_setup(SetupArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < a.K3) a.msc3[idx] = a.mod3[idx / (P3 * P3)];
    if (idx < a.K4) a.msc4[idx] = a.mod4[idx / (P4 * P4)];
    if (idx < a.P) {
        const float x = a.boxes[4 * idx], y = a.boxes[4 * idx + 1], w = a.boxes[4 * idx + 2], h = a.boxes[4 * idx + 3];
        const float sw = a.boxes[2], sh = a.boxes[3];                   // sz_norm = size of the first box (dimp.py:767)
        if (idx == 0) { a.szn[0] = sw; a.szn[1] = sh; }
        a.prev[idx] = -99999999.f;                                      // outputs_prev (atom.py:769)
#pragma unroll
        for (int j = 0; j < 4; ++j) { a.slen[4 * idx + j] = a.step[j]; a.pstep[4 * idx + j] = 0.f; }
        float* s = a.state + 4 * idx;
        if (a.relative) {
            s[0] = (x + 0.5f * w) / sw; s[1] = (y + 0.5f * h) / sh; s[2] = logf(w); s[3] = logf(h);
        } else {
            s[0] = x; s[1] = y; s[2] = w; s[3] = h;
        }
        float* r = a.rois + 5 * idx;
        float rx = x, ry = y, rw = w, rh = h;
        if (a.relative) {                                              // the reference pools at rel_to_rect(rect_to_rel(b))
            rw = expf(s[2]); rh = expf(s[3]); rx = s[0] * sw - 0.5f * rw; ry = s[1] * sh - 0.5f * rh;
        }
        r[0] = 0.f; r[1] = rx; r[2] = ry; r[3] = rx + rw; r[4] = ry + rh;
    }
}

struct HeadArgs {
    const float *part3, *part4, *b3, *bn3, *b4, *bn4, *wp, *bp;
    float *G3, *G4, *iou;
    int P, I3, I4, nz3, nz4;
};

// LinearBlock epilogues (bias, BatchNorm on running statistics, ReLU), the IoU Linear, and dIoU/d(pre-activation) for the
// backward GEMMs; one workgroup per proposal
__global__ __launch_bounds__(256) void k_iou_head(HeadArgs a) {
    __shared__ float scratch[4];
    const int p = blockIdx.x;
    float acc = 0.f;
    for (int n = threadIdx.x; n < a.I3 + a.I4; n += 256) {
        const bool l3 = n < a.I3;
        const int m = l3 ? n : n - a.I3, I = l3 ? a.I3 : a.I4, nz = l3 ? a.nz3 : a.nz4;
        const float* part = l3 ? a.part3 : a.part4;
        const float* bn = l3 ? a.bn3 : a.bn4;
        float pre = (l3 ? a.b3 : a.b4)[m];
#pragma unroll 8
        for (int z = 0; z < nz; ++z) pre += part[((long)z * a.P + p) * I + m];     // independent loads, fixed order
        const float sc = bn[m] / sqrtf(bn[3 * I + m] + 1e-5f);
        const float y = (pre - bn[2 * I + m]) * sc + bn[I + m];
        const float w = a.wp[n];
        acc += w * fmaxf(y, 0.f);
        (l3 ? a.G3 : a.G4)[(long)p * I + m] = y > 0.f ? w * sc : 0.f;
    }
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) a.iou[p] = acc + a.bp[0];
}

struct UpdArgs {
    const float *gr3, *gr4, *szn, *iou;
    float *state, *rois, *boxes_out, *prev, *slen, *pstep;
    int P, relative, last, backtrack;
    float step[4], decay;
};

// d IoU / d [x0,y0,x1,y1] (both levels) -> gradient in the optimisation variable -> ascent step -> next rois
__global__ __launch_bounds__(64) void k_iou_update(UpdArgs a) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= a.P) return;
    float g3[4] = {0.f, 0.f, 0.f, 0.f}, g4[4] = {0.f, 0.f, 0.f, 0.f};   // slice partials, fixed order
    for (int sl = 0; sl < GSL; ++sl)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g3[j] += a.gr3[((long)p * GSL + sl) * 4 + j];
            g4[j] += a.gr4[((long)p * GSL + sl) * 4 + j];
        }
    const float gx0 = g3[0] + g4[0], gy0 = g3[1] + g4[1], gx1 = g3[2] + g4[2], gy1 = g3[3] + g4[3];
    const float gx = gx0 + gx1, gy = gy0 + gy1, gw = gx1, gh = gy1;      // [x, y, x + w, y + h]
    float* s = a.state + 4 * p;
    float dir[4];                                                        // ascent direction in the optimisation variable
    if (a.relative) {
        const float sw = a.szn[0], sh = a.szn[1];
        const float w0 = expf(s[2]), h0 = expf(s[3]);                    // rel_to_rect at the current iterate
        dir[0] = gx * sw; dir[1] = gy * sh; dir[2] = w0 * (gw - 0.5f * gx); dir[3] = h0 * (gh - 0.5f * gy);
    } else {
        dir[0] = gx * s[2]; dir[1] = gy * s[3]; dir[2] = gw * s[2]; dir[3] = gh * s[3];   // grad * [w, h, w, h]
    }
    if (a.backtrack) {
        // ATOM (atom.py:783-795,812-820): a proposal whose predicted IoU did not improve shrinks its own step length and
        // takes the previous step back; the others ascend with their current step length
        const bool up = a.iou[p] > a.prev[p] || a.decay >= 1.f;
        a.prev[p] = a.iou[p];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!up) a.slen[4 * p + j] *= a.decay;
            const float st = up ? a.slen[4 * p + j] * dir[j] : -a.pstep[4 * p + j];
            a.pstep[4 * p + j] = st;
            s[j] += st;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += a.step[j] * dir[j];
    }
    float x, y, w, h;
    if (a.relative) {
        w = expf(s[2]); h = expf(s[3]); x = s[0] * a.szn[0] - 0.5f * w; y = s[1] * a.szn[1] - 0.5f * h;
    } else {
        x = s[0]; y = s[1]; w = s[2]; h = s[3];
    }
    float* r = a.rois + 5 * p;
    r[0] = 0.f; r[1] = x; r[2] = y; r[3] = x + w; r[4] = y + h;
    if (a.last) {
        a.boxes_out[4 * p] = x; a.boxes_out[4 * p + 1] = y; a.boxes_out[4 * p + 2] = w; a.boxes_out[4 * p + 3] = h;
    }
}

}  // namespace

extern "C" size_t pt_iou_param_floats(const pt_iou_dims* d) { return iou_check(d) ? 0 : iou_layout(d).total; }

extern "C" size_t pt_iou_prepared_floats(const pt_iou_dims* d) {
    if (iou_check(d)) return 0;
    return pt_align_floats((size_t)d->I3 * d->C3 * P3 * P3) + pt_align_floats((size_t)d->I4 * d->C4 * P4 * P4);
}

extern "C" int pt_iou_prepare_f32(const pt_iou_dims* d, const float* params, float* prepared, void* stream) {
    if (!params || !prepared) return PT_ERR_NULL;
    int rc = iou_check(d);
    if (rc) return rc;
    const IouOff po = iou_layout(d);
    const int K3 = d->C3 * P3 * P3, K4 = d->C4 * P4 * P4;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_transpose, dim3((K3 + 31) / 32, (d->I3 + 31) / 32), dim3(256), 0, st, params + po.w3, prepared,
                       d->I3, K3);
    PT_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_transpose, dim3((K4 + 31) / 32, (d->I4 + 31) / 32), dim3(256), 0, st, params + po.w4,
                       prepared + pt_align_floats((size_t)d->I3 * K3), d->I4, K4);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" size_t pt_iou_refine_ws_bytes(const pt_iou_dims* d, int P) {
    if (iou_check(d) || P <= 0 || P > 256) return 0;
    return iou_carve(d, P).total * sizeof(float);
}

extern "C" int pt_iou_refine_f32(const pt_iou_dims* d, const float* params, const float* prepared, const float* c3,
                                 const float* c4, const float* mod3, const float* mod4, const float* init_boxes,
                                 float* boxes_out, float* iou_out, int P, int num_iter, const float* step_length4,
                                 float step_decay, int relative, int backtrack, void* ws, size_t ws_bytes,
                                 void* stream) {
    if (!params || !prepared || !c3 || !c4 || !mod3 || !mod4 || !init_boxes || !boxes_out || !iou_out || !step_length4 || !ws)
        return PT_ERR_NULL;
    int rc = iou_check(d);
    if (rc) return rc;
    if (P <= 0 || num_iter <= 0) return PT_ERR_SHAPE;
    if (P > 256) return PT_ERR_UNSUPPORTED;
    const IouCarve cv = iou_carve(d, P);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* base = (float*)ws;
    const IouOff po = iou_layout(d);
    const int K3 = d->C3 * P3 * P3, K4 = d->C4 * P4 * P4, I3 = d->I3, I4 = d->I4;
    const float* W3T = prepared;
    const float* W4T = prepared + pt_align_floats((size_t)I3 * K3);
    SetupArgs sa{init_boxes, mod3, mod4, base + cv.state, base + cv.szn, base + cv.rois, base + cv.msc3, base + cv.msc4,
                 base + cv.prev, base + cv.slen, base + cv.pstep, P, K3, K4, relative,
                 {step_length4[0], step_length4[1], step_length4[2], step_length4[3]}};
    hipLaunchKernelGGL(k_iou_setup, dim3((std::max(std::max(K3, K4), P) + 255) / 256), dim3(256), 0, st, sa);
    PT_CHECK_LAUNCH();
    float step[4] = {step_length4[0], step_length4[1], step_length4[2], step_length4[3]};
    const float* const feats[2] = {c3, c4};
    const float* const mods[2] = {mod3, mod4};
    float* const Xs[2] = {base + cv.X3, base + cv.X4};
    const float* const dXs[2] = {base + cv.dX3, base + cv.dX4};
    float* const grs[2] = {base + cv.gr3, base + cv.gr4};
    const int Cs[2] = {d->C3, d->C4}, Hs[2] = {d->H3, d->H4}, Ws[2] = {d->W3, d->W4}, PHs[2] = {P3, P4};
    const float scales[2] = {S3, S4};
    for (int it = 0; it < num_iter; ++it) {
        // ---- forward (atom_iou_net.py:108-134): both pools + modulation in one launch
        if ((rc = pt_launch_prroi_fwd2(feats, mods, Xs, Cs, Hs, Ws, PHs, scales, base + cv.rois, P, st))) return rc;
        GemmArgs g = gemm_args(base + cv.X3, K3, P, params + po.w3, P, I3, K3, nullptr, base + cv.part3, I3);
        g.ksteps = KSPLIT; g.c_zstride = (long)P * I3;
        GemmArgs g2 = gemm_args(base + cv.X4, K4, P, params + po.w4, P, I4, K4, nullptr, base + cv.part4, I4);
        g2.ksteps = KSPLIT; g2.c_zstride = (long)P * I4;
        if ((rc = launch_gemm_pair(g, g2, st))) return rc;              // both FC layers in one launch
        HeadArgs ha{base + cv.part3, base + cv.part4, params + po.b3, params + po.bn3, params + po.b4, params + po.bn4,
                    params + po.wp, params + po.bp, base + cv.G3, base + cv.G4, iou_out, P, I3, I4, cv.nz3, cv.nz4};
        hipLaunchKernelGGL(k_iou_head, dim3(P), dim3(256), 0, st, ha);
        PT_CHECK_LAUNCH();
        // ---- backward to the box: d pooled = (G W) * modulation, then the PrRoIPool coordinate gradient
        g = gemm_args(base + cv.G3, I3, P, W3T, P, K3, I3, nullptr, base + cv.dX3, K3);
        g.scale = base + cv.msc3;
        g2 = gemm_args(base + cv.G4, I4, P, W4T, P, K4, I4, nullptr, base + cv.dX4, K4);
        g2.scale = base + cv.msc4;
        if ((rc = launch_gemm_pair(g, g2, st))) return rc;
        if ((rc = pt_launch_prroi_bwd_coor2(dXs, feats, grs, Cs, Hs, Ws, PHs, scales, base + cv.rois, P, GSL, st))) return rc;
        UpdArgs ua{base + cv.gr3, base + cv.gr4, base + cv.szn, iou_out, base + cv.state, base + cv.rois, boxes_out,
                   base + cv.prev, base + cv.slen, base + cv.pstep, P, relative, it == num_iter - 1, backtrack,
                   {step[0], step[1], step[2], step[3]}, step_decay};
        hipLaunchKernelGGL(k_iou_update, dim3((P + 63) / 64), dim3(64), 0, st, ua);
        PT_CHECK_LAUNCH();
        for (float& s : step) s *= step_decay;                          // dimp.py:748,779 (unused when backtracking)
    }
    return PT_OK;
}
