// The part of `DiMP.track` / `refine_target_box` between the reference's two host round trips, as device code
// (pytracking/tracker/dimp/dimp.py:118-131, 486-504, 650-675): new position from the translation vector, update_state's clamp,
// get_iounet_box, the jittered proposals.  All of it is float32 tensor arithmetic on the CPU in the reference; here the same operations
// in the same order, un-fused (__f*_rn), one thread per proposal.  Python scalars that multiply float32 tensors are rounded to float32
// first, as torch's binary ops do.  Used by the merged mid-frame kernel (iou_refine.hip: k_frame_mid), filled by frame_full.hip.
#pragma once
#include "common.h"
#include "localize_dev.h"

namespace {

struct GlueArgs {
    float* host;               // pinned result block (PT_FRAME_HOST_FLOATS)
    float pos[2], target_sz[2], sample_pos[16], sample_scales[8];
    float image_sz[2], img_sample_sz[2];
    float inside_ratio_m_half;                 // float32(target_inside_ratio - 0.5)
    float jitter_pos, jitter_sz;
    int use_classifier, num_random;
    float rand_u[60];
};

// Threads 0 .. P-1 of the workgroup (every thread may call it): L = the 15 localisation results (LDS), boxes = (P, 4) proposals (LDS);
// thread 0 also sends the results, the new position and the initial box to the host block.
__device__ __forceinline__ void frame_glue(const GlueArgs& a, const float* L, float* boxes) {
    const int t = threadIdx.x;
    const int code = (int)L[0];
    const int s = (int)L[1];
    // new_pos = sample_pos[scale_ind] + translation_vec (dimp.py:118)
    float pos[2], ib[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float new_pos = __fadd_rn(a.sample_pos[2 * s + k], L[4 + k]);
        float p = a.pos[k];
        if (code != PT_LOC_NOT_FOUND && a.use_classifier) {                       // update_state(new_pos), dimp.py:493-495
            const float off = __fmul_rn(a.inside_ratio_m_half, a.target_sz[k]);
            p = fmaxf(fminf(new_pos, __fsub_rn(a.image_sz[k], off)), off);
        }
        pos[k] = p;
    }
    // get_iounet_box(self.pos, self.target_sz, sample_pos[scale_ind], sample_scales[scale_ind]), dimp.py:498-504
    const float sc = a.sample_scales[s];
    float ul[2], bsz[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float center = __fadd_rn(__fdiv_rn(__fsub_rn(pos[k], a.sample_pos[2 * s + k]), sc),
                                       __fdiv_rn(__fsub_rn(a.img_sample_sz[k], 1.0f), 2.0f));
        bsz[k] = __fdiv_rn(a.target_sz[k], sc);
        ul[k] = __fsub_rn(center, __fdiv_rn(__fsub_rn(bsz[k], 1.0f), 2.0f));
    }
    ib[0] = ul[1]; ib[1] = ul[0]; ib[2] = bsz[1]; ib[3] = bsz[0];                 // flip: (x, y, w, h)
    const int P = 1 + a.num_random;
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) boxes[k] = ib[k];
        for (int k = 0; k < 15; ++k) a.host[k] = L[k];                            // the localisation results for the host
        a.host[16] = pos[0]; a.host[17] = pos[1];
        for (int k = 0; k < 4; ++k) a.host[18 + k] = ib[k];
    } else if (t < P) {                                                           // dimp.py:663-675
        const float square = __fsqrt_rn(__fmul_rn(ib[2], ib[3]));
        const float rf_pos = __fmul_rn(square, a.jitter_pos), rf_sz = __fmul_rn(square, a.jitter_sz);
        const float min_edge = __fdiv_rn(fminf(ib[2], ib[3]), 3.0f);
        const float* u = a.rand_u + 4 * (t - 1);
        float rb[4];
        rb[0] = __fmul_rn(__fsub_rn(u[0], 0.5f), rf_pos); rb[1] = __fmul_rn(__fsub_rn(u[1], 0.5f), rf_pos);
        rb[2] = __fmul_rn(__fsub_rn(u[2], 0.5f), rf_sz);  rb[3] = __fmul_rn(__fsub_rn(u[3], 0.5f), rf_sz);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float nsz = fmaxf(__fadd_rn(ib[2 + k], rb[2 + k]), min_edge);
            const float ctr = __fadd_rn(__fadd_rn(ib[k], __fdiv_rn(ib[2 + k], 2.0f)), rb[k]);
            boxes[4 * t + k] = __fsub_rn(ctr, __fdiv_rn(nsz, 2.0f));
            boxes[4 * t + 2 + k] = nsz;
        }
    }
}

// what frame_full.hip hands to pt_iou_refine_launch: the localisation and the glue run in front of the refinement's set-up stage, in
// ONE single-workgroup launch (three launches of ~5 us each otherwise)
struct PtFrameMid {
    DecideArgs dec;
    GlueArgs glue;
};

}  // namespace
