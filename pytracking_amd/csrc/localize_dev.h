// Device side of the score-map localisation shared by localize.hip (k_localize_decide and friends) and the merged mid-frame kernel of
// iou_refine.hip (k_frame_mid): see localize.hip for the semantics (max2d's tie order, Python round() = rint on doubles, float32
// un-fused arithmetic in the reference's order).
#pragma once
#include "common.h"

namespace {

struct Peak { float v; int r, c; };

__device__ __forceinline__ bool better(const Peak& a, const Peak& b) {
    return a.v > b.v || (a.v == b.v && (a.c < b.c || (a.c == b.c && a.r < b.r)));
}

// block-wide arg-max with max2d's tie order; every thread gets the result.  `better` is a strict total order over
// distinct cells, so the butterfly leaves the same winner in every lane; the <= 4 wave winners meet in LDS.
__device__ Peak block_peak(Peak p, Peak* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const Peak o{__shfl_xor(p.v, off, 64), __shfl_xor(p.r, off, 64), __shfl_xor(p.c, off, 64)};
        if (better(o, p)) p = o;
    }
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = p;
    __syncthreads();
    Peak b = sh[0];
    for (int w = 1; w < nw; ++w)
        if (better(sh[w], b)) b = sh[w];
    return b;
}

__device__ Peak scan_map(const float* m, int H, int W, int top, int bottom, int left, int right) {
    Peak best{-INFINITY, H, W};
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
        const int r = i / W, c = i - r * W;
        const bool masked = r >= top && r < bottom && c >= left && c < right;
        const Peak p{masked ? 0.f : m[i], r, c};
        if (better(p, best)) best = p;
    }
    return best;
}

struct DecideArgs {
    const float *scores, *scores_hn;
    float* out;
    int S, H, W;
    float seq;                   // written to out[15] LAST (system scope): what pt_localize_advanced_sync_f32 polls
    pt_localize_params p;
};

// All threads of a 256-thread workgroup call this; thread 0 leaves the 15 results in o[0..14] (o: LDS or global).
__device__ __forceinline__ void localize_decide(const DecideArgs& a, Peak* sh, float* o) {
    const pt_localize_params& q = a.p;
    Peak p1{-INFINITY, 0, 0};
    int s1 = 0;
    for (int s = 0; s < a.S; ++s) {
        const Peak p = block_peak(scan_map(a.scores + (long)s * a.H * a.W, a.H, a.W, 0, 0, 0, 0), sh);
        if (s == 0 || p.v > p1.v) { p1 = p; s1 = s; }
    }
    const double nr = (double)q.neigh_r[s1], nc = (double)q.neigh_c[s1];
    const int top = max((int)rint((double)p1.r - nr / 2), 0);
    const int bottom = min((int)rint((double)p1.r + nr / 2 + 1), a.H);
    const int left = max((int)rint((double)p1.c - nc / 2), 0);
    const int right = min((int)rint((double)p1.c + nc / 2 + 1), a.W);
    const Peak p2 = block_peak(scan_map(a.scores_hn + (long)s1 * a.H * a.W, a.H, a.W, top, bottom, left, right), sh);
    if (threadIdx.x != 0) return;                                        // (the caller synchronises before anybody reads `o`)

    // offsets of both peaks from the map centre and from where the target was in the previous frame
    const float d1r = __fsub_rn((float)p1.r, q.center_r), d1c = __fsub_rn((float)p1.c, q.center_c);
    const float d2r = __fsub_rn((float)p2.r, q.center_r), d2c = __fsub_rn((float)p2.c, q.center_c);
    int code, pick = 1;
    const double m1 = (double)p1.v;
    if (m1 < q.target_not_found_threshold) code = PT_LOC_NOT_FOUND;
    else if (m1 < q.uncertain_threshold) code = PT_LOC_UNCERTAIN;
    else if (m1 < q.hard_sample_threshold) code = PT_LOC_HARD_NEGATIVE;
    else if (p2.v > __fmul_rn(q.distractor_threshold, p1.v)) {
        // two comparable peaks: the one that stayed near the previous position is the target, the other a distractor
        const float e1r = __fsub_rn(d1r, q.prev_r[s1]), e1c = __fsub_rn(d1c, q.prev_c[s1]);
        const float e2r = __fsub_rn(d2r, q.prev_r[s1]), e2c = __fsub_rn(d2c, q.prev_c[s1]);
        const float n1 = __fsqrt_rn(__fadd_rn(__fmul_rn(e1r, e1r), __fmul_rn(e1c, e1c)));
        const float n2 = __fsqrt_rn(__fadd_rn(__fmul_rn(e2r, e2r), __fmul_rn(e2c, e2c)));
        const bool far1 = n1 > q.disp_threshold, near1 = n1 < q.disp_threshold;
        const bool far2 = n2 > q.disp_threshold, near2 = n2 < q.disp_threshold;
        if (far2 && near1) code = PT_LOC_HARD_NEGATIVE;
        else if (near2 && far1) { code = PT_LOC_HARD_NEGATIVE; pick = 2; }
        else code = PT_LOC_UNCERTAIN;
    } else if (p2.v > __fmul_rn(q.hard_negative_threshold, p1.v) && p2.v > q.target_not_found_f32) {
        code = PT_LOC_HARD_NEGATIVE;
    } else {
        code = PT_LOC_NORMAL;
    }
    const float dr = pick == 1 ? d1r : d2r, dc = pick == 1 ? d1c : d2c;
    o[0] = (float)code; o[1] = (float)s1;
    o[2] = (float)(pick == 1 ? p1.r : p2.r); o[3] = (float)(pick == 1 ? p1.c : p2.c);
    o[4] = __fmul_rn(__fmul_rn(dr, q.ratio_r), q.scale[s1]);            // translation_vec = disp * (support / output) * scale
    o[5] = __fmul_rn(__fmul_rn(dc, q.ratio_c), q.scale[s1]);
    o[6] = p1.v; o[7] = (float)p1.r; o[8] = (float)p1.c;
    o[9] = p2.v; o[10] = (float)p2.r; o[11] = (float)p2.c;
    o[12] = (float)pick; o[13] = 0.f; o[14] = 0.f;
}

}  // namespace
