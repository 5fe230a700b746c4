// IoU-guided box refinement on the device:
//   AtomIoUNet.predict_iou (modulation, PrRoIPool 5x5 @ 1/8 and 3x3 @ 1/16, two LinearBlocks, Linear -> IoU)
//                                                             ltr/models/bbreg/atom_iou_net.py:96-136, layers/blocks.py:23-36
//   DiMP.optimize_boxes_default / optimize_boxes_relative     pytracking/tracker/dimp/dimp.py:725-788
//   ATOM.optimize_boxes (per-proposal backtracking)           pytracking/tracker/atom/atom.py:758-836
//   rect_to_rel / rel_to_rect                                 ltr/data/bounding_box_utils.py:4-33
// The reference runs every refinement step as an autograd forward + backward (~60 launches, a host-visible graph per
// step, 5-10 steps per frame).  Here the gradient of the predicted IoU w.r.t. the box is written out: forward pools ->
// LinearBlock products on the matrix cores -> BatchNorm/ReLU/IoU head that also emits dIoU/d(pre-activation) -> products with
// the transposed weights -> PrRoIPool coordinate gradient -> box update, no host synchronisation until the boxes are read.
// Two forms: the general six-launch iteration (any proposal count <= 256) and, for the trackers' <= 16 proposals, a fused
// three-launch iteration in which one workgroup owns a chunk of the pooled axis in both directions (second half of this file).
#include "common.h"
#include "pt_internal.h"
#include "mfma_gemm.h"
#include "prroi_dev.h"
#include "frame_mid.h"

#include <math.h>
#include <stdlib.h>
#include <chrono>
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
    size_t fpart3, fpart4, gpart, ipart, st2;       // fused path (P <= FUSED_MAX_P)
    int nz3, nz4, fz3, fz4;
};
constexpr int KSPLIT = 4;                           // K-steps of 64 per split-K slice of the forward GEMMs
constexpr int GSL = 16;                             // element slices of the PrRoIPool coordinate gradient per proposal
constexpr int CK = 64;                              // pooled elements per workgroup of the fused iteration kernels
constexpr int FUSED_MAX_P = 16;                     // proposals of the fused path: one MFMA row tile
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
    c.fz3 = (int)((K3 + CK - 1) / CK); c.fz4 = (int)((K4 + CK - 1) / CK);
    if (P <= FUSED_MAX_P) {
        c.fpart3 = take((size_t)c.fz3 * P * d->I3); c.fpart4 = take((size_t)c.fz4 * P * d->I4);
        c.gpart = take((size_t)(c.fz3 + c.fz4) * P * 4); c.ipart = take((size_t)P * 8);
        c.st2 = take((size_t)2 * FUSED_MAX_P * 16);
    }
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
    float* st0;                     // fused path: record of iteration 0 (s[4], slen[4], pstep[4], prev per proposal), else null
    int use_hb;                     // the initial boxes travel in the argument block (host pointer given, P <= 16)
    float hb[4 * 16];
};

// the optimisation variable (rect or relative), step state and first rois of proposal idx; bx: the P initial boxes
__device__ __forceinline__ void iou_setup_proposal(const SetupArgs& a, int idx, const float* bx) {
    {
        const float x = bx[4 * idx], y = bx[4 * idx + 1], w = bx[4 * idx + 2], h = bx[4 * idx + 3];
        const float sw = bx[2], sh = bx[3];                             // sz_norm = size of the first box (dimp.py:767)
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
        if (a.st0) {
            float* rec = a.st0 + 16 * idx;
#pragma unroll
            for (int j = 0; j < 4; ++j) { rec[j] = s[j]; rec[4 + j] = a.step[j]; rec[8 + j] = 0.f; }
            rec[12] = -99999999.f;
        }
        float* r = a.rois + 5 * idx;
        float rx = x, ry = y, rw = w, rh = h;
        if (a.relative) {                                              // the reference pools at rel_to_rect(rect_to_rel(b))
            rw = expf(s[2]); rh = expf(s[3]); rx = s[0] * sw - 0.5f * rw; ry = s[1] * sh - 0.5f * rh;
        }
        r[0] = 0.f; r[1] = rx; r[2] = ry; r[3] = rx + rw; r[4] = ry + rh;
    }
}

// per-column modulation of the pooled features (unfused route) + the per-proposal set-up
__global__ __launch_bounds__(256) void k_iou_setup(SetupArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < a.K3) a.msc3[idx] = a.mod3[idx / (P3 * P3)];
    if (idx < a.K4) a.msc4[idx] = a.mod4[idx / (P4 * P4)];
    if (idx < a.P) iou_setup_proposal(a, idx, a.use_hb ? a.hb : a.boxes);
}

// The middle of the one-call frame (frame_full.hip): localize_advanced on the frame's score map, the tracker's glue (new position,
// update_state, get_iounet_box, jittered proposals) and the refinement's set-up stage for those proposals -- one workgroup, the
// hand-offs through LDS instead of three dependent launches.
__global__ __launch_bounds__(256) void k_frame_mid(DecideArgs d, GlueArgs g, SetupArgs a) {
    __shared__ Peak sh[4];
    __shared__ float res[16];
    __shared__ float bx[4 * 16];
    localize_decide(d, sh, res);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 15; ++k) d.out[k] = res[k];                  // device copy of the results (workspace)
    }
    __syncthreads();
    if (threadIdx.x < a.P) frame_glue(g, res, bx);
    __syncthreads();
    if (threadIdx.x < a.P) iou_setup_proposal(a, threadIdx.x, bx);
}

// The same with the per-frame part of its arguments -- the localisation constants and the tracker state of the glue, i.e. the whole
// PtFrameMid block -- read from DEVICE memory (graph-replayed one-call frame: a captured launch cannot carry per-frame values; the
// host refreshes the block through a copy node in front of the graph, frame_full.hip).  The block is staged in LDS first.
__global__ __launch_bounds__(256) void k_frame_mid_dyn(const PtFrameMid* __restrict__ mid, SetupArgs a) {
    __shared__ Peak sh[4];
    __shared__ float res[16];
    __shared__ float bx[4 * 16];
    __shared__ __attribute__((aligned(16))) int blk[(sizeof(PtFrameMid) + 3) / 4];
    for (int e = threadIdx.x; e < (int)(sizeof(PtFrameMid) / 4); e += 256) blk[e] = ((const int*)mid)[e];
    __syncthreads();
    const PtFrameMid& m = *reinterpret_cast<const PtFrameMid*>(blk);
    localize_decide(m.dec, sh, res);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 15; ++k) m.dec.out[k] = res[k];
    }
    __syncthreads();
    if (threadIdx.x < a.P) frame_glue(m.glue, res, bx);
    __syncthreads();
    if (threadIdx.x < a.P) iou_setup_proposal(a, threadIdx.x, bx);
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

// one proposal's optimisation state, fetched before the gradient partials are added (one memory round trip, not two)
struct UpdState { float s[4], slen[4], pstep[4], prev, sw, sh; };

__device__ __forceinline__ UpdState iou_update_load(const UpdArgs& a, int p) {
    UpdState u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u.s[j] = a.state[4 * p + j];
        u.slen[j] = a.backtrack ? a.slen[4 * p + j] : 0.f;
        u.pstep[j] = a.backtrack ? a.pstep[4 * p + j] : 0.f;
    }
    u.prev = a.backtrack ? a.prev[p] : 0.f;
    u.sw = a.szn[0]; u.sh = a.szn[1];
    return u;
}

// one proposal: d IoU / d [x0,y0,x1,y1] (both levels summed) -> gradient in the optimisation variable -> ascent step ->
// next rois
__device__ __forceinline__ void iou_update_one(const UpdArgs& a, int p, UpdState u, float gx0, float gy0, float gx1, float gy1,
                                               float iou) {
    const float gx = gx0 + gx1, gy = gy0 + gy1, gw = gx1, gh = gy1;      // [x, y, x + w, y + h]
    float* s = u.s;
    float dir[4];                                                        // ascent direction in the optimisation variable
    if (a.relative) {
        const float w0 = expf(s[2]), h0 = expf(s[3]);                    // rel_to_rect at the current iterate
        dir[0] = gx * u.sw; dir[1] = gy * u.sh; dir[2] = w0 * (gw - 0.5f * gx); dir[3] = h0 * (gh - 0.5f * gy);
    } else {
        dir[0] = gx * s[2]; dir[1] = gy * s[3]; dir[2] = gw * s[2]; dir[3] = gh * s[3];   // grad * [w, h, w, h]
    }
    if (a.backtrack) {
        // ATOM (atom.py:783-795,812-820): a proposal whose predicted IoU did not improve shrinks its own step length and
        // takes the previous step back; the others ascend with their current step length
        const bool up = iou > u.prev || a.decay >= 1.f;
        a.prev[p] = iou;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!up) a.slen[4 * p + j] = u.slen[j] = u.slen[j] * a.decay;
            const float st = up ? u.slen[j] * dir[j] : -u.pstep[j];
            a.pstep[4 * p + j] = st;
            s[j] += st;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += a.step[j] * dir[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) a.state[4 * p + j] = s[j];
    float x, y, w, h;
    if (a.relative) {
        w = expf(s[2]); h = expf(s[3]); x = s[0] * u.sw - 0.5f * w; y = s[1] * u.sh - 0.5f * h;
    } else {
        x = s[0]; y = s[1]; w = s[2]; h = s[3];
    }
    float* r = a.rois + 5 * p;
    r[0] = 0.f; r[1] = x; r[2] = y; r[3] = x + w; r[4] = y + h;
    if (a.last) {
        a.boxes_out[4 * p] = x; a.boxes_out[4 * p + 1] = y; a.boxes_out[4 * p + 2] = w; a.boxes_out[4 * p + 3] = h;
    }
}

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
    iou_update_one(a, p, iou_update_load(a, p), g3[0] + g4[0], g3[1] + g4[1], g3[2] + g4[2], g3[3] + g4[3], a.iou[p]);
}

// ------------------------------------------------------------------------------------------------------------------
// Fused iteration for P <= 16 proposals (every deployed configuration refines 10: dimp.py:691, atom.py:724).  The six-launch
// form above spends its time in launch gaps and in round trips of the pooled tensors through memory; here the pooled axis
// (C * bins of both levels, 8704 values per proposal at ResNet-50 sizes) is cut into chunks of CK = 64 and one workgroup owns a
// chunk in both directions:
//   k_iou_fwd   pools its chunk for all proposals (x modulation) into LDS and multiplies it with the chunk's columns of the
//               LinearBlock weight on the matrix cores (16 x 64 by 64 x I) -> one partial of the pre-activation per chunk;
//   k_iou_head2 adds the partials (4 groups of threads per output, fixed order), BatchNorm, ReLU, the IoU Linear, and
//               dIoU/d(pre-activation);
//   k_iou_bwd   forms its chunk of d IoU / d pooled (16 x I by I x 64, rows of the transposed weight) and immediately contracts
//               it with the PrRoIPool coordinate gradient of the same elements -> four partial sums per proposal and chunk;
//   the box update (chunk partials -> gradient -> step -> RoIs) runs in front of the NEXT k_iou_fwd, k_iou_final after the last one.
// The weights (9 MB at ResNet-50 sizes) are read once per direction and stay in L2 between iterations; nothing of size
// P x K is written to memory any more.

// one pyramid level's operands; the kernels fetch the block of THEIR level in one batch (pt_late_args, common.h) -- indexing
// a by-value argument struct with the run-time level made every field its own scalar load + wait at its point of use
struct Lv {
    const float *feat, *mod, *w, *wt, *G;
    float* part;                    // (chunks, P, I) forward partials
    int C, H, W, K, I, nz;
};
constexpr unsigned LV_OFF = 32;     // byte offset of lv0 in the kernel-argument segment of k_iou_fwd / k_iou_bwd

#ifdef PT_IOU_STAMPS                // experiments only: phase time stamps, tools/exp_iou_stamps.py
static unsigned long long* g_iou_stamps = nullptr;
extern "C" void pt_debug_set_iou_stamps(void* p) { g_iou_stamps = (unsigned long long*)p; }
#define IOU_STAMP(k)                                                                                                     \
    do {                                                                                                                 \
        if (stamps && lane == 0) stamps[((long)blockIdx.x * 16 + w) * 8 + (k)] = (unsigned long long)wall_clock64();     \
    } while (0)
#else
#define IOU_STAMP(k)
#endif

constexpr int FT = 512;                             // threads of k_iou_fwd / k_iou_bwd: 8 waves; wave w serves proposals w and w + 8
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- the box update, folded into the front of the NEXT forward kernel (and a one-workgroup kernel after the last iteration) ----
// A separate update launch cost ~5 us per iteration for 40 sums.  Every workgroup of k_iou_fwd now adds the gradient partials of the
// previous iteration itself (22 KB from L2), moves the boxes, and keeps the RoIs in LDS; workgroup 0 also publishes them (state
// record of the next iteration, RoIs for k_iou_bwd, predicted IoU).  The state is double-buffered by iteration parity: the other
// workgroups may still be reading the record workgroup 0 would otherwise overwrite.
struct UpdLate {
    const float *gpart, *ipart, *bp, *szn, *st_in;
    float *st_out, *rois, *boxes_out, *iou_out, *seq_word;
    int nzt, relative, backtrack, first, last;      // first: no gradient yet, only the RoIs are formed; last: boxes_out is written
    float step[4], decay;
    float seq;                                      // != 0: the final kernel stores it to seq_word after the results (host-polled)
    int pad;
};
static_assert(sizeof(UpdLate) == 32 * 4, "UpdLate layout");
constexpr int ST_REC = 16;                           // floats per proposal record: s[4], slen[4], pstep[4], prev

// all threads of the workgroup; contains two barriers; afterwards rois_s[p][0..4] = RoI of proposal p at the current iterate
__device__ __forceinline__ void iou_step(const UpdLate& U, int P, float (*red)[64], float (*rois_s)[5], bool writer) {
    const int t = threadIdx.x, o = t & 63, grp = t >> 6, n4 = 4 * P;
    const int pc = min(t, P - 1);
    const pt_gptr<const float> rec = pt_global(U.st_in) + ST_REC * pc;
    float s[4], slen[4], pstep[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] = rec[j]; slen[j] = rec[4 + j]; pstep[j] = rec[8 + j]; }
    float prev = rec[12];
    const float sw = pt_global(U.szn)[0], sh = pt_global(U.szn)[1];
    float ip0 = 0.f, ip1 = 0.f, bp = 0.f, sum = 0.f;
    if (!U.first) {
        // the IoU Linear's 2 x HB partial sums of k_iou_head2, added in their fixed order
        float ipv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ipv[j] = pt_global(U.ipart)[8 * pc + j];
        ip0 = (ipv[0] + ipv[1]) + (ipv[2] + ipv[3]); ip1 = (ipv[4] + ipv[5]) + (ipv[6] + ipv[7]);
        bp = pt_global(U.bp)[0];
        const int per = (U.nzt + 7) / 8, z0 = grp * per, z1 = min(U.nzt, z0 + per), oc = min(o, n4 - 1);
        const pt_gptr<const float> gp = pt_global(U.gpart);
        for (int zb = z0; zb < z1; zb += 20) {
            float v[20];
#pragma unroll
            for (int u = 0; u < 20; ++u) v[u] = gp[(long)min(zb + u, z1 - 1) * n4 + oc];
#pragma unroll
            for (int u = 0; u < 20; ++u) sum += zb + u < z1 ? v[u] : 0.f;
        }
    }
    red[grp][o] = sum;
    __syncthreads();
    if (t < P) {
        if (!U.first) {
            float g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                g[j] = ((red[0][4 * t + j] + red[1][4 * t + j]) + (red[2][4 * t + j] + red[3][4 * t + j])) +
                       ((red[4][4 * t + j] + red[5][4 * t + j]) + (red[6][4 * t + j] + red[7][4 * t + j]));
            const float iou = ip0 + ip1 + bp;
            const float gx = g[0] + g[2], gy = g[1] + g[3], gw = g[2], gh = g[3];      // [x, y, x + w, y + h]
            float dir[4];                                                              // ascent direction in the optimisation variable
            if (U.relative) {
                const float w0 = expf(s[2]), h0 = expf(s[3]);                          // rel_to_rect at the current iterate
                dir[0] = gx * sw; dir[1] = gy * sh; dir[2] = w0 * (gw - 0.5f * gx); dir[3] = h0 * (gh - 0.5f * gy);
            } else {
                dir[0] = gx * s[2]; dir[1] = gy * s[3]; dir[2] = gw * s[2]; dir[3] = gh * s[3];   // grad * [w, h, w, h]
            }
            if (U.backtrack) {                                                         // atom.py:783-795,812-820, see iou_update_one
                const bool up = iou > prev || U.decay >= 1.f;
                prev = iou;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!up) slen[j] *= U.decay;
                    const float st = up ? slen[j] * dir[j] : -pstep[j];
                    pstep[j] = st;
                    s[j] += st;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += U.step[j] * dir[j];
            }
            if (writer && U.last) pt_global(U.iou_out)[t] = iou;        // the prediction at the last evaluated iterate
        }
        float x, y, w, h;
        if (U.relative) {
            w = expf(s[2]); h = expf(s[3]); x = s[0] * sw - 0.5f * w; y = s[1] * sh - 0.5f * h;
        } else {
            x = s[0]; y = s[1]; w = s[2]; h = s[3];
        }
        rois_s[t][0] = 0.f; rois_s[t][1] = x; rois_s[t][2] = y; rois_s[t][3] = x + w; rois_s[t][4] = y + h;
        if (writer) {
            const pt_gptr<float> out = pt_global(U.st_out) + ST_REC * t;
#pragma unroll
            for (int j = 0; j < 4; ++j) { out[j] = s[j]; out[4 + j] = slen[j]; out[8 + j] = pstep[j]; }
            out[12] = prev;
            const pt_gptr<float> r = pt_global(U.rois) + 5 * t;
            r[0] = 0.f; r[1] = x; r[2] = y; r[3] = x + w; r[4] = y + h;
            if (U.last) {
                const pt_gptr<float> b = pt_global(U.boxes_out) + 4 * t;
                b[0] = x; b[1] = y; b[2] = w; b[3] = h;
            }
        }
    }
    __syncthreads();
}

// Geometry of one (proposal, bin): everything of the PrRoIPool integral that does not depend on the channel.  A chunk of 64 pooled
// elements is ~2.5 channels x all bins, so computed per element (as the stand-alone op does) the ~300 instructions of hat-function
// weights were repeated for every channel and the kernels were bound by them (a wave issues ~0.5 instructions / ns: measured,
// experiments/icache_probe.hip).  The threads of a workgroup fill this table once; an element then costs its 36 window reads (LDS)
// and ~100 instructions.  The weights are the expressions of prroi_fwd_window_sum / prroi_coor_window_sums, evaluated by another lane.
constexpr int GW = 6;                               // window of the table path: bins that touch <= 6 x 6 pixels (the stand-alone op's two window sizes)
struct BinGeo {
    float wx[GW], wy[GW], hxs[GW], hxe[GW], hys[GW], hye[GW];
    float area, bw, bh;
    int i0, i1, j0, j1, flag;       // flag: 0 = contributes nothing, 1 = within a 6 x 6 pixel window, 2 = larger bin (generic path)
    int pad;                        // 45 words: an odd stride, conflict-free reads across bins
};
static_assert(sizeof(BinGeo) == 45 * 4, "BinGeo layout");
constexpr int GEO_MAX = FUSED_MAX_P * P3 * P3;

// two threads per (proposal, bin): one the column weights, one the row weights (500 of the 512 threads busy at 10 proposals x 25 bins)
// (first5: the RoI of this thread's first entry if the caller requested it earlier -- k_iou_bwd does, in front of its argument fetch)
template <bool BWD, typename RP>
__device__ __forceinline__ void iou_geometry(BinGeo* geo, RP rois, int P, int PH, float scale, int H, int W, const float* first5 = nullptr) {
    const int PP = PH * PH;
    for (int e2 = threadIdx.x; e2 < 2 * P * PP; e2 += FT) {
        const int e = e2 >> 1, rows = e2 & 1;
        const int slot = e / PP, bin = e - slot * PP, pp = bin / PH, q = bin - pp * PH;
        float rr[5];
        if (first5 && e2 == (int)threadIdx.x) {
#pragma unroll
            for (int u = 0; u < 5; ++u) rr[u] = first5[u];
        } else {
#pragma unroll
            for (int u = 0; u < 5; ++u) rr[u] = rois[5 * slot + u];
        }
        const Bin k = make_bin(rr, pp, q, PH, PH, scale, H, W);
        const int nj = k.j1 - k.j0 + 1, ni = k.i1 - k.i0 + 1;
        BinGeo& g = geo[e];
        // the same expressions for either axis: [lo, hi] = the bin's extent, first = its first pixel, cnt = pixels it touches
        const float lo = rows ? k.ys : k.xs, hi = rows ? k.ye : k.xe;
        const int first = rows ? k.j0 : k.i0, cnt = rows ? nj : ni;
        float* wv = rows ? g.wy : g.wx;
        float* hs = rows ? g.hys : g.hxs;
        float* he = rows ? g.hye : g.hxe;
#pragma unroll
        for (int ii = 0; ii < GW; ++ii) {
            const float i = (float)(first + ii);
            const bool in = ii < cnt;
            wv[ii] = in ? hat_cdf(hi - i) - hat_cdf(lo - i) : 0.f;
            if (BWD) {
                hs[ii] = in ? hat(lo - i) : 0.f;
                he[ii] = in ? hat(hi - i) : 0.f;
            }
        }
        if (rows) {
            g.j0 = k.j0; g.j1 = k.j1;
            g.flag = (k.area > 0.f && k.b == 0 && nj > 0 && ni > 0) ? ((nj <= GW && ni <= GW) ? 1 : 2) : 0;
        } else {
            g.area = k.area; g.bw = k.bw; g.bh = k.bh;
            g.i0 = k.i0; g.i1 = k.i1;
        }
    }
}

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// The channel planes a chunk touches (64 consecutive (channel, bin) elements: <= 4 planes of 36 x 36 or 9 of 18 x 18), copied to LDS
// with coalesced 16-byte loads.  Gathering the windows straight from global memory -- 16 pixels (or 4 rows) per element, every lane its
// own address -- kept the vector-memory address unit busy for ~5 us per workgroup (stamps: the same with per-pixel and with per-row
// loads); LDS serves such gathers at full rate.  Larger maps than PLANE_MAX floats fall back to the global gather.
constexpr int PLANE_MAX = 6144, PLANE_IT = PLANE_MAX / (4 * FT);
struct PlaneRegs { f32x4 v[PLANE_IT]; };
// request (into registers; the geometry stage runs under the loads) ...
__device__ __forceinline__ bool stage_planes_load(PlaneRegs& r, pt_gptr<const float> feat, int c_first, int nch, int HW) {
    const int nfl = nch * HW;
    if (nfl > PLANE_MAX) return false;
    const pt_gptr<const float> src = feat + (long)c_first * HW;
#pragma unroll
    for (int it = 0; it < PLANE_IT; ++it) {
        const int i = 4 * (int)threadIdx.x + 4 * FT * it;
        if (i + 3 < nfl) r.v[it] = *(pt_gptr<const f32x4u>)(src + i);
        else
#pragma unroll
            for (int u = 0; u < 4; ++u) r.v[it][u] = i + u < nfl ? src[i + u] : 0.f;
    }
    return true;
}
// ... and store
__device__ __forceinline__ void stage_planes_store(float* planes, const PlaneRegs& r, int nch, int HW) {
    const int nfl = nch * HW;
#pragma unroll
    for (int it = 0; it < PLANE_IT; ++it) {
        const int i = 4 * (int)threadIdx.x + 4 * FT * it;
        if (i < nfl) *(f32x4*)(planes + i) = r.v[it];
    }
}
// the window of an element out of the staged planes, clamped to the pixels the bin touches exactly as prroi_window_load does
__device__ __forceinline__ void lds_window_load(const float* pl, const BinGeo& g, int W, float (&v)[GW][GW]) {
#pragma unroll
    for (int jj = 0; jj < GW; ++jj)
#pragma unroll
        for (int ii = 0; ii < GW; ++ii) v[jj][ii] = pl[min(g.j0 + jj, g.j1) * W + min(g.i0 + ii, g.i1)];
}

// Operand layout of both matrix products (v_mfma_f32_16x16x4_f32: A[m = lane % 16][k = lane / 16], B[k = lane / 16][n = lane % 16]): the
// weight is read along its contiguous axis, lane (c = lane % 16, kg = lane / 16) takes 2 or 4 CONSECUTIVE output columns of row k, and MFMA
// number j of a k-step accumulates the tile of the columns {v * c + j}.  One load instruction then covers 4 rows x 128 / 256 contiguous
// bytes (4 / 8 cache lines); the first version read 16 rows x 64 bytes per instruction.  The k index inside a 16-block is permuted
// (k = 16 h + 4 kg + j) so that the A operand is a float4 per lane as well.
__global__ __launch_bounds__(FT) void k_iou_fwd(int nz0, int P, const float* __restrict__ rois_unused, float* gpart, unsigned long long* stamps,
                                                Lv lv0_arg, Lv lv1_arg, UpdLate upd_arg) {
    __shared__ __attribute__((aligned(16))) float Xs[16][CK + 4];
    __shared__ __attribute__((aligned(16))) float planes[PLANE_MAX + 4];
    __shared__ BinGeo geo[GEO_MAX];
    __shared__ float red[8][64];
    __shared__ float rois_s[FUSED_MAX_P][5];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), c16 = lane & 15, kg = lane >> 4;
    const int l = (int)blockIdx.x >= nz0, zz = (int)blockIdx.x - (l ? nz0 : 0);
    IOU_STAMP(0);
    PtLate<Lv> late_l = pt_late_issue<Lv>(LV_OFF + (l ? (unsigned)sizeof(Lv) : 0u));
    PtLate<UpdLate> late_u = pt_late_issue<UpdLate>(LV_OFF + 2u * (unsigned)sizeof(Lv));
    const Lv L = pt_late_get<Lv>(late_l);
    const UpdLate U = pt_late_get<UpdLate>(late_u);
    const float* rois = &rois_s[0][0];
    const int K = L.K, I = L.I, PH = l ? P4 : P3, PP = l ? P4 * P4 : P3 * P3, H = L.H, W = L.W;
    const float scale = l ? S4 : S3;
    const int k0 = zz * CK, kn = min(CK, K - k0);
    const int c_first = l ? k0 / (P4 * P4) : k0 / (P3 * P3), c_last = l ? (k0 + kn - 1) / (P4 * P4) : (k0 + kn - 1) / (P3 * P3);
    // 1. the chunk's rows of the transposed weight for this wave's 32 output columns: in flight under everything else
    f32x2 wv[4][4];
    const pt_gptr<const float> WT = pt_global(L.wt) + (long)k0 * I + 32 * w + 2 * c16;
    auto load_w = [&](int nb) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[h][j] = *(pt_gptr<const f32x2>)(WT + (long)min(16 * h + 4 * kg + j, kn - 1) * I + nb);
    };
    if (32 * w < I) load_w(0);
    PlaneRegs pr;
    const bool staged = stage_planes_load(pr, pt_global(L.feat), c_first, c_last - c_first + 1, H * W);
    const int gk = k0 + min(lane, kn - 1), c = l ? gk / (P4 * P4) : gk / (P3 * P3), bin = gk - c * PP, pp = bin / PH, q = bin - pp * PH;
    const pt_gptr<const float> f = pt_global(L.feat) + (long)c * H * W;
    const float mo = pt_global(L.mod)[c];
    // 2. the box update of the previous iteration (RoIs -> LDS), then the bin geometry of every proposal, once per workgroup
    iou_step(U, P, red, rois_s, blockIdx.x == 0);
    iou_geometry<false>(geo, rois, P, PH, scale, H, W);
    if (staged) stage_planes_store(planes, pr, c_last - c_first + 1, H * W);
    __syncthreads();
    IOU_STAMP(1);
    // 3. pooling: wave w = proposals w and w + 8, lane = element of the chunk
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int slot = w + 8 * r;
        const int flag = (slot < P && lane < kn) ? geo[min(slot, P - 1) * PP + bin].flag : 0;
        float v = 0.f;
        if (flag == 1 && staged) {
            const BinGeo& g = geo[slot * PP + bin];
            float win[GW][GW];
            lds_window_load(planes + (c - c_first) * H * W, g, W, win);
            float acc = 0.f;
#pragma unroll
            for (int jj = 0; jj < GW; ++jj) {
                float row = 0.f;
#pragma unroll
                for (int ii = 0; ii < GW; ++ii) row += win[jj][ii] * g.wx[ii];
                acc += g.wy[jj] * row;
            }
            v = acc / g.area;
        } else if (flag) {
            v = prroi_fwd_elem(L.feat, rois, slot, c, pp, q, 1, L.C, H, W, PH, PH, scale);
        }
        Xs[slot][lane] = v * mo;
        if (r == 0) IOU_STAMP(6);
    }
    IOU_STAMP(2);
    __syncthreads();
    IOU_STAMP(3);
    f32x4 xa[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) xa[h] = *(const f32x4*)&Xs[c16][16 * h + 4 * kg];
    const pt_gptr<float> part = pt_global(L.part) + (long)zz * P * I;
    for (int nb = 0; nb + 32 * w < I; nb += 256) {
        if (nb) load_w(nb);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = mfma16(xa[h][j], wv[h][j][0], acc0);
                acc1 = mfma16(xa[h][j], wv[h][j][1], acc1);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * kg + i < P) *(pt_gptr<f32x2>)(part + (long)(4 * kg + i) * I + nb + 32 * w + 2 * c16) = f32x2{acc0[i], acc1[i]};
    }
    IOU_STAMP(4);
}

struct Head2Args {
    const float *part[2], *b[2], *bn[2], *wp;
    float *G[2], *ipart;
    int P, I[2], nz[2];
};

// workgroup = (proposal, level, block of 64 outputs); 4 groups of 64 threads each add a quarter of the chunk partials of an output.
// (80 small workgroups instead of 20 of 1024 threads; the per-level fields are selected from both copies -- indexing the argument
// struct with the run-time level costs one scalar load + wait per field at its point of use.)
constexpr int HB = 4;                               // output blocks per (proposal, level): ipart is (P, 2, HB)
static_assert(2 * HB == 8, "iou_step adds 8 partial sums per proposal");
__global__ __launch_bounds__(256) void k_iou_head2(Head2Args a) {
    __shared__ float red[4][64];
    __shared__ float scratch[4];
    const int p = blockIdx.x, l = blockIdx.y, cb = blockIdx.z, t = threadIdx.x, col = t & 63, qtr = t >> 6;
    const int I = l ? a.I[1] : a.I[0], nz = l ? a.nz[1] : a.nz[0], per = (nz + 3) / 4, z0 = qtr * per, z1 = min(nz, z0 + per);
    const float* __restrict__ part = l ? a.part[1] : a.part[0];
    const float* __restrict__ bn = l ? a.bn[1] : a.bn[0];
    const float* __restrict__ bias_p = l ? a.b[1] : a.b[0];
    float* __restrict__ G = l ? a.G[1] : a.G[0];
    const float* __restrict__ wp = a.wp + (l ? a.I[0] : 0);
    float acc = 0.f;
    for (int nb = cb * 64; nb < I; nb += HB * 64) {
        const int n = nb + col, nc = min(n, I - 1);
        // the epilogue's parameters travel with the partials (one round trip, not two)
        const float bias = bias_p[nc], bn_w = bn[nc], bn_b = bn[I + nc], bn_m = bn[2 * I + nc], bn_v = bn[3 * I + nc], wn = wp[nc];
        // all of a thread's partials requested together (a rolled loop waits for each load before the next: the clamp keeps
        // the batch branch-free, the repeated last address costs nothing)
        float s = 0.f;
        for (int zb = z0; zb < z1; zb += 32) {
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = part[((long)min(zb + u, z1 - 1) * a.P + p) * I + nc];
#pragma unroll
            for (int u = 0; u < 32; ++u) s += zb + u < z1 ? v[u] : 0.f;
        }
        red[qtr][col] = s;
        __syncthreads();
        if (qtr == 0 && n < I) {
            const float pre = bias + ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col]));
            const float sc = bn_w / sqrtf(bn_v + 1e-5f);
            const float y = (pre - bn_m) * sc + bn_b;
            acc += wn * fmaxf(y, 0.f);
            G[(long)p * I + n] = y > 0.f ? wn * sc : 0.f;
        }
        __syncthreads();
    }
    acc = block_sum(acc, scratch, 256);
    if (t == 0) a.ipart[(2 * p + l) * HB + cb] = acc;
}

__global__ __launch_bounds__(FT) void k_iou_bwd(int nz0, int P, const float* __restrict__ rois, float* gpart, unsigned long long* stamps,
                                                Lv lv0_arg, Lv lv1_arg) {
    __shared__ float dXs[8][16][CK + 1];
    __shared__ __attribute__((aligned(16))) float planes[PLANE_MAX + 4];
    __shared__ BinGeo geo[GEO_MAX];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), c16 = lane & 15, kg = lane >> 4;
    const int l = (int)blockIdx.x >= nz0, zz = (int)blockIdx.x - (l ? nz0 : 0);
    IOU_STAMP(0);
    // the RoI of this thread's geometry entry: `rois` and P are preloaded parameters, the request goes out before the argument block
    float roi5[5];
    {
        const int slot = min((t >> 1) / (l ? P4 * P4 : P3 * P3), P - 1);
#pragma unroll
        for (int u = 0; u < 5; ++u) roi5[u] = rois[5 * slot + u];
    }
    const Lv L = pt_late_args<Lv>(LV_OFF + (l ? (unsigned)sizeof(Lv) : 0u));
    const int K = L.K, I = L.I, PH = l ? P4 : P3, PP = l ? P4 * P4 : P3 * P3, H = L.H, W = L.W;
    const float scale = l ? S4 : S3;
    const int k0 = zz * CK, kn = min(CK, K - k0);
    const int c_first = l ? k0 / (P4 * P4) : k0 / (P3 * P3), c_last = l ? (k0 + kn - 1) / (P4 * P4) : (k0 + kn - 1) / (P3 * P3);
    // d pooled[p][kk] = sum_n G[p][n] * W[n][kk]: wave w takes 32 of every 256 n (rows of the weight, the chunk's 64 kk contiguous),
    // the eight partial products meet in LDS
    const pt_gptr<const float> grow = pt_global(L.G) + (long)min(c16, P - 1) * I + 32 * w + 4 * kg;
    const pt_gptr<const float> wcol = pt_global(L.w) + k0 + min(4 * c16, kn - 4);
    f32x4 ga[2], wb[2][4];
    auto load_gw = [&](int nb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ga[h] = *(pt_gptr<const f32x4>)(grow + nb + 16 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) wb[h][j] = *(pt_gptr<const f32x4>)(wcol + (long)(nb + 32 * w + 16 * h + 4 * kg + j) * K);
        }
    };
    if (32 * w < I) load_gw(0);
    PlaneRegs pr;
    const bool staged = stage_planes_load(pr, pt_global(L.feat), c_first, c_last - c_first + 1, H * W);
    const int gk = k0 + min(lane, kn - 1), c = l ? gk / (P4 * P4) : gk / (P3 * P3), bin = gk - c * PP, pp = bin / PH, q = bin - pp * PH;
    const pt_gptr<const float> f = pt_global(L.feat) + (long)c * H * W;
    const float mo = pt_global(L.mod)[c];
    iou_geometry<true>(geo, rois, P, PH, scale, H, W, roi5);
    if (staged) stage_planes_store(planes, pr, c_last - c_first + 1, H * W);
    __syncthreads();
    IOU_STAMP(1);
    IOU_STAMP(2);

    f32x4 acc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int nb = 0; nb + 32 * w < I; nb += 256) {
        if (nb) load_gw(nb);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[jj] = mfma16(ga[h][j], wb[h][j][jj], acc[jj]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dXs[w][4 * kg + i][4 * c16 + jj] = acc[jj][i];
    IOU_STAMP(3);
    __syncthreads();
    IOU_STAMP(4);
    // PrRoIPool coordinate gradient of the chunk's elements: wave w = proposals w and w + 8, lane = element
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int slot = w + 8 * r;
        if (slot >= P) break;
        const int flag = lane < kn ? geo[slot * PP + bin].flag : 0;
        float gx0 = 0.f, gy0 = 0.f, gx1 = 0.f, gy1 = 0.f;
        if (flag) {
            float gd = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) gd += dXs[ww][slot][lane];
            gd *= mo;
            if (flag == 1 && staged) {
                const BinGeo& g = geo[slot * PP + bin];
                float win[GW][GW];
                lds_window_load(planes + (c - c_first) * H * W, g, W, win);
                float integ = 0.f, lxs = 0.f, lxe = 0.f, lys = 0.f, lye = 0.f;
#pragma unroll
                for (int jj = 0; jj < GW; ++jj) {
                    float row = 0.f, rxs = 0.f, rxe = 0.f;
#pragma unroll
                    for (int ii = 0; ii < GW; ++ii) {
                        row += win[jj][ii] * g.wx[ii];
                        rxs += win[jj][ii] * g.hxs[ii];
                        rxe += win[jj][ii] * g.hxe[ii];
                    }
                    integ += g.wy[jj] * row;
                    lxs += g.wy[jj] * rxs;
                    lxe += g.wy[jj] * rxe;
                    lys += g.hys[jj] * row;
                    lye += g.hye[jj] * row;
                }
                Bin k;
                k.area = g.area; k.bw = g.bw; k.bh = g.bh;
                prroi_coor_finish(k, integ, lxs, lxe, lys, lye, gd, pp, q, PH, PH, gx0, gy0, gx1, gy1);
            } else {
                const Bin k = make_bin(rois + 5 * slot, pp, q, PH, PH, scale, H, W);
                prroi_coor_elem(L.feat + (long)c * H * W, k, W, gd, pp, q, PH, PH, gx0, gy0, gx1, gy1);
            }
        }
        gx0 = wave_sum(gx0); gy0 = wave_sum(gy0); gx1 = wave_sum(gx1); gy1 = wave_sum(gy1);
        if (lane == 0) {
            float* o = gpart + ((long)blockIdx.x * P + slot) * 4;
            o[0] = gx0 * scale; o[1] = gy0 * scale; o[2] = gx1 * scale; o[3] = gy1 * scale;
        }
    }
    IOU_STAMP(5);
}

// the update after the last iteration: boxes and predicted IoU out
// seq_dyn (graph-replayed frame): the sequence number comes from device memory instead of the captured argument block
__global__ __launch_bounds__(FT) void k_iou_final(int P, UpdLate U, const float* __restrict__ seq_dyn) {
    __shared__ float red[8][64];
    __shared__ float rois_s[FUSED_MAX_P][5];
    const float seq = seq_dyn ? *seq_dyn : U.seq;
    iou_step(U, P, red, rois_s, true);
    if (seq != 0.f && threadIdx.x == 0) {           // boxes_out / iou_out were stored by lanes of this wave: program order + release
        __threadfence_system();
        *(volatile float*)U.seq_word = seq;
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

// init_boxes: device pointer, or (boxes_on_host) a host pointer whose P <= 16 boxes travel in the first kernel's argument block;
// seq / seq_word: see pt_iou_refine_sync_f32
// Every argument / shape / route check of a refinement call, WITHOUT queuing anything (advisor, round 5: pt_track_frame_full_f32 used to
// learn about a bad refinement argument only after the head, the memory insert and the re-optimisation of the same call were queued and
// had mutated the tracker state).  iou_refine_impl starts with this; frame_full.hip calls it in front of its first launch.
//   have_init_boxes : the caller passes initial boxes (otherwise `with_mid`: k_frame_mid forms the proposals inside the launch)
//   seq != 0 / boxes_on_host / with_mid need the fused iteration kernels (P <= FUSED_MAX_P); with_mid also P <= 16
int pt_iou_refine_validate(const pt_iou_dims* d, const float* params, const float* prepared, const float* c3, const float* c4,
                           const float* mod3, const float* mod4, bool have_init_boxes, const float* boxes_out, const float* iou_out, int P,
                           int num_iter, const float* step_length4, const void* ws, size_t ws_bytes, bool boxes_on_host, float seq,
                           bool with_mid) {
    if (!params || !prepared || !c3 || !c4 || !mod3 || !mod4 || (!have_init_boxes && !with_mid) || !boxes_out || !iou_out || !step_length4 || !ws)
        return PT_ERR_NULL;
    int rc = iou_check(d);
    if (rc) return rc;
    if (P <= 0 || num_iter <= 0) return PT_ERR_SHAPE;
    if (P > 256) return PT_ERR_UNSUPPORTED;
    const IouCarve cv = iou_carve(d, P);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    static const bool unfused = [] { const char* e = getenv("PT_IOU_UNFUSED"); return e && e[0] == '1'; }();   // ablation switch
    const bool fused = P <= FUSED_MAX_P && !unfused;
    if ((boxes_on_host || seq != 0.f) && !fused) return PT_ERR_UNSUPPORTED;
    if (with_mid && (!fused || P > 16)) return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

static int iou_refine_impl(const pt_iou_dims* d, const float* params, const float* prepared, const float* c3, const float* c4,
                           const float* mod3, const float* mod4, const float* init_boxes, bool boxes_on_host, float* boxes_out,
                           float* iou_out, int P, int num_iter, const float* step_length4, float step_decay, int relative,
                           int backtrack, void* ws, size_t ws_bytes, float seq, float* seq_word, void* stream,
                           const PtFrameMid* mid = nullptr, const PtFrameMid* mid_dev = nullptr, const float* seq_dyn = nullptr) {
    int rc = pt_iou_refine_validate(d, params, prepared, c3, c4, mod3, mod4, init_boxes != nullptr, boxes_out, iou_out, P, num_iter,
                                    step_length4, ws, ws_bytes, boxes_on_host, seq, mid != nullptr);
    if (rc) return rc;
    const IouCarve cv = iou_carve(d, P);
    hipStream_t st = (hipStream_t)stream;
    float* base = (float*)ws;
    const IouOff po = iou_layout(d);
    const int K3 = d->C3 * P3 * P3, K4 = d->C4 * P4 * P4, I3 = d->I3, I4 = d->I4;
    const float* W3T = prepared;
    const float* W4T = prepared + pt_align_floats((size_t)I3 * K3);
    SetupArgs sa{init_boxes, mod3, mod4, base + cv.state, base + cv.szn, base + cv.rois, base + cv.msc3, base + cv.msc4,
                 base + cv.prev, base + cv.slen, base + cv.pstep, P, K3, K4, relative,
                 {step_length4[0], step_length4[1], step_length4[2], step_length4[3]}, nullptr};
    static const bool unfused = [] { const char* e = getenv("PT_IOU_UNFUSED"); return e && e[0] == '1'; }();   // ablation switch
    const bool fused = P <= FUSED_MAX_P && !unfused;
    if (fused) { sa.st0 = base + cv.st2; sa.K3 = sa.K4 = 0; }          // the per-column modulation tables belong to the unfused path
    if (boxes_on_host) {
        sa.boxes = nullptr; sa.use_hb = 1;
        for (int i = 0; i < 4 * P; ++i) sa.hb[i] = init_boxes[i];
    }
    if (mid) {                                                          // proposals are formed inside the launch (fused route only)
        if (mid_dev) hipLaunchKernelGGL(k_frame_mid_dyn, dim3(1), dim3(256), 0, st, mid_dev, sa);     // per-frame values from device memory
        else hipLaunchKernelGGL(k_frame_mid, dim3(1), dim3(256), 0, st, mid->dec, mid->glue, sa);
    } else {
        hipLaunchKernelGGL(k_iou_setup, dim3((std::max(std::max(sa.K3, sa.K4), P) + 255) / 256), dim3(256), 0, st, sa);
    }
    PT_CHECK_LAUNCH();
    float step[4] = {step_length4[0], step_length4[1], step_length4[2], step_length4[3]};
    const float* const feats[2] = {c3, c4};
    const float* const mods[2] = {mod3, mod4};
    float* const Xs[2] = {base + cv.X3, base + cv.X4};
    const float* const dXs[2] = {base + cv.dX3, base + cv.dX4};
    float* const grs[2] = {base + cv.gr3, base + cv.gr4};
    const int Cs[2] = {d->C3, d->C4}, Hs[2] = {d->H3, d->H4}, Ws[2] = {d->W3, d->W4}, PHs[2] = {P3, P4};
    const float scales[2] = {S3, S4};
    if (fused) {
        const Lv lv0{c3, mod3, params + po.w3, W3T, base + cv.G3, base + cv.fpart3, d->C3, d->H3, d->W3, K3, I3, cv.fz3};
        const Lv lv1{c4, mod4, params + po.w4, W4T, base + cv.G4, base + cv.fpart4, d->C4, d->H4, d->W4, K4, I4, cv.fz4};
        unsigned long long* stamps = nullptr;
#ifdef PT_IOU_STAMPS
        stamps = g_iou_stamps;
#endif
        Head2Args ha{{base + cv.fpart3, base + cv.fpart4}, {params + po.b3, params + po.b4}, {params + po.bn3, params + po.bn4},
                     params + po.wp, {base + cv.G3, base + cv.G4}, base + cv.ipart, P, {I3, I4}, {cv.fz3, cv.fz4}};
        const int nzt = cv.fz3 + cv.fz4;
        float* const stb[2] = {base + cv.st2, base + cv.st2 + FUSED_MAX_P * ST_REC};
        UpdLate up{base + cv.gpart, base + cv.ipart, params + po.bp, base + cv.szn, nullptr, nullptr, base + cv.rois, boxes_out, iou_out,
                   seq_word, nzt, relative, backtrack, 1, 0, {step[0], step[1], step[2], step[3]}, step_decay, seq, 0};
        for (int it = 0; it <= num_iter; ++it) {
            // the update after iteration it - 1 (its step lengths) runs in front of iteration it; after the last one, on its own
            up.st_in = stb[it & 1]; up.st_out = stb[(it + 1) & 1];
            up.first = it == 0; up.last = it == num_iter;
            if (it == num_iter) {
                hipLaunchKernelGGL(k_iou_final, dim3(1), dim3(FT), 0, st, P, up, seq_dyn);
                PT_CHECK_LAUNCH();
                break;
            }
            hipLaunchKernelGGL(k_iou_fwd, dim3(nzt), dim3(FT), 0, st, cv.fz3, P, (const float*)nullptr, base + cv.gpart, stamps, lv0, lv1, up);
            PT_CHECK_LAUNCH();
            hipLaunchKernelGGL(k_iou_head2, dim3(P, 2, HB), dim3(256), 0, st, ha);
            PT_CHECK_LAUNCH();
            hipLaunchKernelGGL(k_iou_bwd, dim3(nzt), dim3(FT), 0, st, cv.fz3, P, (const float*)(base + cv.rois), base + cv.gpart,
                               stamps ? stamps + (size_t)nzt * 16 * 8 : nullptr, lv0, lv1);
            PT_CHECK_LAUNCH();
            if (it > 0)
                for (float& sl : up.step) sl *= step_decay;            // dimp.py:748,779 (unused when backtracking)
        }
        return PT_OK;
    }
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

extern "C" int pt_iou_refine_f32(const pt_iou_dims* d, const float* params, const float* prepared, const float* c3,
                                 const float* c4, const float* mod3, const float* mod4, const float* init_boxes,
                                 float* boxes_out, float* iou_out, int P, int num_iter, const float* step_length4,
                                 float step_decay, int relative, int backtrack, void* ws, size_t ws_bytes,
                                 void* stream) {
    return iou_refine_impl(d, params, prepared, c3, c4, mod3, mod4, init_boxes, false, boxes_out, iou_out, P, num_iter, step_length4,
                           step_decay, relative, backtrack, ws, ws_bytes, 0.f, nullptr, stream);
}

// The trackers' call as it happens per frame (dimp.py:691-722, atom.py:724-756): the proposals are formed on the HOST and the refined
// boxes are needed there before anything else can happen.  `init_boxes_host` (P <= 16 boxes, xywh) travels inside the first kernel's
// argument block (no host-to-device copy); `out_host` is PT_IOU_HOST_FLOATS floats of pinned host memory the device can write: boxes at
// [0, 4P), predicted IoU at [64, 64 + P), and a per-call sequence number at [95] that the last kernel stores after the results
// (system-scope release).  The call returns when that word has arrived -- polled, like pt_localize_advanced_sync_f32; falls back to
// hipStreamSynchronize after 2 s.
extern "C" int pt_iou_refine_sync_f32(const pt_iou_dims* d, const float* params, const float* prepared, const float* c3,
                                      const float* c4, const float* mod3, const float* mod4, const float* init_boxes_host,
                                      float* out_host, int P, int num_iter, const float* step_length4, float step_decay,
                                      int relative, int backtrack, void* ws, size_t ws_bytes, void* stream) {
    if (!out_host || !init_boxes_host) return PT_ERR_NULL;
    if (P > FUSED_MAX_P) return PT_ERR_UNSUPPORTED;
    if (!pt_pinned_host_checked(out_host) || pt_stream_is_capturing(stream)) return PT_ERR_UNSUPPORTED;
    volatile float* word = out_host + 95;
    const float seq = pt_next_seq(word);
    const int rc = iou_refine_impl(d, params, prepared, c3, c4, mod3, mod4, init_boxes_host, true, out_host, out_host + 64, P, num_iter,
                                   step_length4, step_decay, relative, backtrack, ws, ws_bytes, seq, out_host + 95, stream);
    if (rc) return rc;
    return pt_poll_word(word, seq, out_host, stream);
}

// launch half for compositions (frame_full.hip): proposals in DEVICE memory, results + sequence word wherever the caller points
int pt_iou_refine_launch(const pt_iou_dims* d, const float* params, const float* prepared, const float* c3, const float* c4,
                         const float* mod3, const float* mod4, const float* init_boxes_dev, float* boxes_out, float* iou_out, int P,
                         int num_iter, const float* step_length4, float step_decay, int relative, int backtrack, void* ws,
                         size_t ws_bytes, float seq, float* seq_word, void* stream, const void* frame_mid, const void* frame_mid_dev,
                         const float* seq_dyn) {
    return iou_refine_impl(d, params, prepared, c3, c4, mod3, mod4, init_boxes_dev, false, boxes_out, iou_out, P, num_iter,
                           step_length4, step_decay, relative, backtrack, ws, ws_bytes, seq, seq_word, stream,
                           (const PtFrameMid*)frame_mid, (const PtFrameMid*)frame_mid_dev, seq_dyn);
}
