// The whole DiMP frame behind the backbone as ONE call with ONE host wait (include/pt_hot.h: pt_track_frame_full_f32).
//
//   head -> classify + memory insert + re-optimisation   (pt_track_frame_head_f32, api.hip)
//   -> k_localize_decide                                  (localize.hip; results into the workspace)
//   -> k_frame_glue                                       (this file)
//   -> IoU-guided refinement, proposals in device memory  (iou_refine.hip; results + sequence word into pinned host memory)
//
// `k_frame_glue` is the part of `DiMP.track` / `refine_target_box` that sits between the two host round trips of the reference
// (pytracking/tracker/dimp/dimp.py:118-131, 486-504, 650-675): new position from the translation vector, update_state's clamp,
// get_iounet_box, the jittered proposals.  All of it is float32 tensor arithmetic on the CPU in the reference; here the same
// operations in the same order, un-fused (__f*_rn), one thread per proposal.  Python scalars that multiply float32 tensors are
// rounded to float32 first, as torch's binary ops do.
#include "common.h"
#include "pt_internal.h"

namespace {

struct GlueArgs {
    const float* loc;          // 16 localisation results (device)
    float* boxes;              // (P, 4) proposals for the refinement (device)
    float* host;               // pinned result block (PT_FRAME_HOST_FLOATS)
    float pos[2], target_sz[2], sample_pos[16], sample_scales[8];
    float image_sz[2], img_sample_sz[2];
    float inside_ratio_m_half;                 // float32(target_inside_ratio - 0.5)
    float jitter_pos, jitter_sz;
    int use_classifier, num_random;
    float rand_u[60];
};

__global__ __launch_bounds__(64) void k_frame_glue(GlueArgs a) {
    const int t = threadIdx.x;
    const float* L = a.loc;
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
        for (int k = 0; k < 4; ++k) a.boxes[k] = ib[k];
        for (int k = 0; k < 16; ++k) a.host[k] = L[k];                            // the localisation results for the host
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
            a.boxes[4 * t + k] = __fsub_rn(ctr, __fdiv_rn(nsz, 2.0f));
            a.boxes[4 * t + 2 + k] = nsz;
        }
    }
}

struct FfCarve { size_t frame, loc, boxes, iou, total; };

}  // namespace

static int ff_check(const pt_frame_full* f) {
    if (!f || !f->sd || !f->loc || !f->glue || !f->iou_dims) return PT_ERR_NULL;
    if (f->glue->num_random < 0 || f->glue->num_random > 15) return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

static FfCarve ff_carve(const pt_frame_full* f) {
    FfCarve c;
    const int P = 1 + f->glue->num_random;
    size_t off = 0;
    c.frame = off; off += pt_align_floats(pt_track_frame_head_ws_bytes(f->n, f->Cin, f->C, f->H, f->W, f->K) / sizeof(float));
    c.loc = off; off += pt_align_floats(16);
    c.boxes = off; off += pt_align_floats(64);
    c.iou = off; off += pt_align_floats(pt_iou_refine_ws_bytes(f->iou_dims, P) / sizeof(float));
    c.total = off;
    return c;
}

extern "C" size_t pt_track_frame_full_ws_bytes(const pt_frame_full* f) {
    if (ff_check(f)) return 0;
    if (pt_track_frame_head_ws_bytes(f->n, f->Cin, f->C, f->H, f->W, f->K) == 0) return 0;
    if (pt_iou_refine_ws_bytes(f->iou_dims, 1 + f->glue->num_random) == 0) return 0;
    return ff_carve(f).total * sizeof(float);
}

static int ff_launch(const pt_frame_full* f, float* out, void* ws, size_t ws_bytes, float seq, void* stream) {
    void* const main_stream = stream;
    int rc = ff_check(f);
    if (rc) return rc;
    if (!out || !ws || !f->scores_out) return PT_ERR_NULL;
    const size_t need = pt_track_frame_full_ws_bytes(f);
    if (need == 0) return PT_ERR_UNSUPPORTED;
    if (ws_bytes < need || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    const FfCarve cv = ff_carve(f);
    float* base = (float*)ws;
    const int OH = f->H + (f->K + 1) % 2, OW = f->W + (f->K + 1) % 2;
    pt_localize_params q;                                        // host-side constants first: nothing is queued if they are invalid
    rc = pt_localize_constants_f32(f->loc, 1, OH, OW, &q);
    if (rc) return rc;
    // 1. head + classification + memory insert + re-optimisation; with a second stream the rest of the frame forks off as soon as the
    //    scores are queued and joins at the end
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void* chain = stream;
    if (f->aux_stream && f->aux_stream != stream) {
        if (!pt_stream_events(f->aux_stream, &ev_fork, &ev_join)) return PT_ERR_LAUNCH;
        chain = f->aux_stream;
    }
    rc = pt_track_frame_head_impl(f->sd, f->filter, f->mem_feat, f->mem_bb, f->sample_weight, f->backbone_feat, f->head_weight_tap_major,
                                  f->norm_scale, f->norm_eps, f->slot, f->n, f->Cin, f->C, f->H, f->W, f->K, f->num_iter, f->scores_out,
                                  f->peak_out, base + cv.frame, (cv.loc - cv.frame) * sizeof(float), stream, ev_fork);
    if (rc) return rc;
    if (ev_fork && hipStreamWaitEvent((hipStream_t)chain, ev_fork, 0) != hipSuccess) return PT_ERR_LAUNCH;
    stream = chain;                                              // everything below runs on the chain's stream
    // 2. localisation of the score map the frame just produced (one scale)
    rc = pt_localize_launch(f->scores_out, nullptr, &q, base + cv.loc, 1, OH, OW, 0.f, stream);
    if (rc) return rc;
    // 3. glue: position update, initial box, proposals
    const pt_frame_glue* g = f->glue;
    GlueArgs a{};
    a.loc = base + cv.loc; a.boxes = base + cv.boxes; a.host = out;
    for (int k = 0; k < 2; ++k) {
        a.pos[k] = f->loc->pos[k]; a.target_sz[k] = f->loc->target_sz[k];
        a.image_sz[k] = g->image_sz[k]; a.img_sample_sz[k] = g->img_sample_sz[k];
    }
    for (int k = 0; k < 16; ++k) a.sample_pos[k] = f->loc->sample_pos[k];
    for (int k = 0; k < 8; ++k) a.sample_scales[k] = f->loc->sample_scales[k];
    a.inside_ratio_m_half = (float)(g->target_inside_ratio - 0.5);
    a.jitter_pos = (float)g->box_jitter_pos; a.jitter_sz = (float)g->box_jitter_sz;
    a.use_classifier = g->use_classifier; a.num_random = g->num_random;
    for (int k = 0; k < 4 * g->num_random; ++k) a.rand_u[k] = g->rand_u[k];
    hipLaunchKernelGGL(k_frame_glue, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    PT_CHECK_LAUNCH();
    // 4. refinement; the last kernel writes boxes, IoU and the sequence word into the result block
    const int P = 1 + g->num_random;
    rc = pt_iou_refine_launch(f->iou_dims, f->iou_params, f->iou_prepared, f->c3, f->c4, f->mod3, f->mod4, base + cv.boxes, out + 32,
                                out + 96, P, f->iou_iter, f->step_length4, f->step_decay, f->relative, 0, base + cv.iou,
                                (cv.total - cv.iou) * sizeof(float), seq, out + 127, stream);
    if (rc) return rc;
    if (ev_join) {                                               // the caller's stream joins the chain
        if (hipEventRecord(ev_join, (hipStream_t)chain) != hipSuccess) return PT_ERR_LAUNCH;
        if (hipStreamWaitEvent((hipStream_t)main_stream, ev_join, 0) != hipSuccess) return PT_ERR_LAUNCH;
    }
    return PT_OK;
}

extern "C" int pt_track_frame_full_launch_f32(const pt_frame_full* f, float* out, void* ws, size_t ws_bytes, void* stream) {
    // seq = -1: the word is still written (so a replayed graph signals completion) but nothing polls it here
    return ff_launch(f, out, ws, ws_bytes, -1.0f, stream);
}

extern "C" int pt_track_frame_full_f32(const pt_frame_full* f, float* out_host, void* ws, size_t ws_bytes, void* stream) {
    if (!out_host) return PT_ERR_NULL;
    if (!pt_pinned_host_checked(out_host) || pt_stream_is_capturing(stream)) return PT_ERR_UNSUPPORTED;
    volatile float* word = out_host + 127;
    const float seq = pt_next_seq(word);
    const int rc = ff_launch(f, out_host, ws, ws_bytes, seq, stream);
    if (rc) return rc;
    return pt_poll_word(word, seq, out_host, stream);
}
