// The whole DiMP frame behind the backbone as ONE call with ONE host wait (include/pt_hot.h: pt_track_frame_full_f32).
//
//   head -> classify + memory insert + re-optimisation   (pt_track_frame_head_f32, api.hip)
//   -> k_frame_mid: localize_advanced + the tracker's glue + the refinement's set-up stage in ONE single-workgroup launch
//      (iou_refine.hip; device code of the three parts: localize_dev.h, frame_mid.h, iou_refine.hip)
//   -> IoU-guided refinement iterations                   (iou_refine.hip; results + sequence word into pinned host memory)
#include <cstring>
#include "common.h"
#include "pt_internal.h"
#include "frame_mid.h"

namespace {

struct FfCarve { size_t frame, loc, boxes, iou, total; };

// Per-frame block of the GRAPH-REPLAYED frame (pt_frame_full.dyn): everything of a frame that is a VALUE -- the localisation constants,
// the tracker state the glue works on, this frame's random numbers, the memory slot, the sequence number -- lives in device memory,
// refreshed by a copy node in front of the captured launches, instead of in kernel arguments a graph would freeze.
struct FfDyn {
    PtFrameMid mid;
    int slot;
    float seq;
    int pad[2];
};

}  // namespace

static int ff_check(const pt_frame_full* f) {
    if (!f || !f->sd || !f->loc || !f->glue || !f->iou_dims) return PT_ERR_NULL;
    if (f->glue->num_random < 0 || f->glue->num_random > 15) return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

// Two-stream mode runs the memory update CONCURRENTLY with the refinement, i.e. it labels the new sample from the classification peak
// instead of from the refined state (dimp.py:139-145): only on the caller's explicit word (include/pt_hot.h)
static int ff_check_order(const pt_frame_full* f, void* stream) {
    if (f->aux_stream && f->aux_stream != stream && f->num_iter > 0 && !f->aux_reordered_update_ok) return PT_ERR_UNSUPPORTED;
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

// the by-value part of the mid-frame launch for this frame: localisation constants + glue state (host side)
static int ff_fill_mid(const pt_frame_full* f, float* out, float* loc_out, PtFrameMid& mid) {
    const int OH = f->H + (f->K + 1) % 2, OW = f->W + (f->K + 1) % 2;
    pt_localize_params q;
    const int rc = pt_localize_constants_f32(f->loc, 1, OH, OW, &q);
    if (rc) return rc;
    const pt_frame_glue* g = f->glue;
    mid = PtFrameMid{};
    mid.dec.scores = f->scores_out; mid.dec.scores_hn = f->scores_out; mid.dec.out = loc_out; mid.dec.S = 1; mid.dec.H = OH;
    mid.dec.W = OW; mid.dec.seq = 0.f; mid.dec.p = q;
    GlueArgs& a = mid.glue;
    a.host = out;
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
    return PT_OK;
}

extern "C" size_t pt_track_frame_full_dyn_bytes(void) { return (sizeof(FfDyn) + 255) & ~(size_t)255; }

// Host side of a replayed frame: writes this frame's block into `dyn_host` (pinned memory the graph's copy node reads; >=
// pt_track_frame_full_dyn_bytes()).  Same argument checks as the launch; `out` / `ws` must be the pointers the graph was captured with.
extern "C" int pt_track_frame_full_dyn_fill_f32(const pt_frame_full* f, float seq, float* out, void* ws, size_t ws_bytes, void* dyn_host) {
    int rc = ff_check(f);
    if (rc) return rc;
    if (!out || !ws || !dyn_host || !f->scores_out) return PT_ERR_NULL;
    if (f->slot < 0 || f->slot >= f->n || seq == 0.f) return PT_ERR_SHAPE;
    const size_t need = pt_track_frame_full_ws_bytes(f);
    if (need == 0) return PT_ERR_UNSUPPORTED;
    if (ws_bytes < need || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    const FfCarve cv = ff_carve(f);
    FfDyn d{};
    if ((rc = ff_fill_mid(f, out, (float*)ws + cv.loc, d.mid))) return rc;
    d.slot = f->slot;
    d.seq = seq;
    memcpy(dyn_host, &d, sizeof(d));
    return PT_OK;
}

static int ff_launch(const pt_frame_full* f, float* out, void* ws, size_t ws_bytes, float seq, void* stream) {
    void* const main_stream = stream;
    int rc = ff_check(f);
    if (rc) return rc;
    if ((rc = ff_check_order(f, stream))) return rc;
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
    // ... and every argument of the refinement (pointers, head dimensions, proposal count, iteration count, fused route, workspace):
    // a refusal must come BEFORE the head, the memory insert and the re-optimisation are queued -- they mutate mem_feat[slot],
    // mem_bb[slot] and the filter, and the host block's sequence word must not go stale on a call that returns an error
    rc = pt_iou_refine_validate(f->iou_dims, f->iou_params, f->iou_prepared, f->c3, f->c4, f->mod3, f->mod4, /*have_init_boxes=*/false,
                                out + 32, out + 96, 1 + f->glue->num_random, f->iou_iter, f->step_length4, base + cv.iou,
                                (cv.total - cv.iou) * sizeof(float), /*boxes_on_host=*/false, seq, /*with_mid=*/true);
    if (rc) return rc;
    // 1. head + classification + memory insert + re-optimisation; with a second stream the rest of the frame forks off as soon as the
    //    scores are queued and joins at the end
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void* chain = stream;
    if (f->aux_stream && f->aux_stream != stream) {
        if (!pt_stream_events(stream, f->aux_stream, &ev_fork, &ev_join)) return PT_ERR_LAUNCH;
        chain = f->aux_stream;
    }
    // dyn: the per-frame VALUES come from the device block (graph replay); the launches below then depend on pointers and shapes only
    const FfDyn* dyn = (const FfDyn*)f->dyn;
    if (dyn && (((uintptr_t)dyn % 16) != 0 || f->aux_stream)) return PT_ERR_UNSUPPORTED;
    rc = pt_track_frame_head_impl(f->sd, f->filter, f->mem_feat, f->mem_bb, f->sample_weight, f->backbone_feat, f->head_weight_tap_major,
                                  f->norm_scale, f->norm_eps, f->slot, f->n, f->Cin, f->C, f->H, f->W, f->K, f->num_iter, f->scores_out,
                                  f->peak_out, base + cv.frame, (cv.loc - cv.frame) * sizeof(float), stream, ev_fork, dyn ? &dyn->slot : nullptr);
    if (rc) return rc;
    if (ev_fork && hipStreamWaitEvent((hipStream_t)chain, ev_fork, 0) != hipSuccess) return PT_ERR_LAUNCH;
    stream = chain;                                              // everything below runs on the chain's stream
    // 2. + 3. localisation of the score map the frame just produced (one scale) and the glue (position update, initial box,
    //    proposals): both run inside the refinement's first launch (k_frame_mid, iou_refine.hip)
    PtFrameMid mid{};
    if ((rc = ff_fill_mid(f, out, base + cv.loc, mid))) return rc;     // (constants were validated above: cannot fail here)
    const pt_frame_glue* g = f->glue;
    // 4. refinement; the last kernel writes boxes, IoU and the sequence word into the result block
    const int P = 1 + g->num_random;
    rc = pt_iou_refine_launch(f->iou_dims, f->iou_params, f->iou_prepared, f->c3, f->c4, f->mod3, f->mod4, base + cv.boxes, out + 32,
                              out + 96, P, f->iou_iter, f->step_length4, f->step_decay, f->relative, 0, base + cv.iou,
                              (cv.total - cv.iou) * sizeof(float), seq, out + 127, stream, &mid, dyn ? (const void*)&dyn->mid : nullptr,
                              dyn ? &dyn->seq : nullptr);
    if (ev_join) {
        // the caller's stream joins the chain -- also when the refinement refused its arguments: whatever was queued on the second
        // stream stays inside the call's ordering contract (and a capture in progress must not be left with an open fork)
        if (hipEventRecord(ev_join, (hipStream_t)chain) != hipSuccess || hipStreamWaitEvent((hipStream_t)main_stream, ev_join, 0) != hipSuccess)
            return rc ? rc : PT_ERR_LAUNCH;
    }
    return rc;
}

extern "C" int pt_host_wait_word_f32(const float* word, float seq, void* stream) {
    if (!word) return PT_ERR_NULL;
    if (!pt_pinned_host_checked(word) || pt_stream_is_capturing(stream)) return PT_ERR_UNSUPPORTED;
    return pt_poll_word((volatile float*)word, seq, word, stream);
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
