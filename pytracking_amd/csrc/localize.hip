// Target localisation on the device: the score-map part of
//   DiMP.localize_advanced / ToMP.localize_advanced   pytracking/tracker/dimp/dimp.py:238-303, tracker/tomp/tomp.py
//   dcf.max2d                                         pytracking/libs/dcf.py:156-164
// as one launch that leaves 8 floats for a single device-to-host copy.  The reference issues two max2d's, a clone, a
// masked fill and about ten .item()/.cpu() synchronisations per frame for the same numbers.
//
// Semantics kept bit-for-bit: max2d takes the maximum over rows first and then over columns, both returning the first
// index on ties, i.e. ties go to the smallest column, then the smallest row; the best scale is the first one on ties;
// the neighbourhood of the first peak is ZEROED (not excluded) before the second search; its bounds use Python's
// round() on doubles = round-half-to-even = rint.
#include <chrono>
#include "common.h"
#include "pt_internal.h"
#include "localize_dev.h"

namespace {

struct LocArgs {
    const float *scores, *scores_hn;
    float* out;
    int S, H, W;
    float nh[8], nw[8];          // target neighbourhood (rows, cols) in score cells, per scale
};

__global__ __launch_bounds__(256) void k_localize(LocArgs a) {
    __shared__ Peak sh[4];
    Peak p1{-INFINITY, 0, 0};
    int s1 = 0;
    for (int s = 0; s < a.S; ++s) {
        const Peak p = block_peak(scan_map(a.scores + (long)s * a.H * a.W, a.H, a.W, 0, 0, 0, 0), sh);
        if (s == 0 || p.v > p1.v) { p1 = p; s1 = s; }               // torch.max over scales: first maximum
    }
    const int top = max((int)rint((double)p1.r - (double)a.nh[s1] / 2), 0);          // Python floats are doubles
    const int bottom = min((int)rint((double)p1.r + (double)a.nh[s1] / 2 + 1), a.H);
    const int left = max((int)rint((double)p1.c - (double)a.nw[s1] / 2), 0);
    const int right = min((int)rint((double)p1.c + (double)a.nw[s1] / 2 + 1), a.W);
    const Peak p2 = block_peak(scan_map(a.scores_hn + (long)s1 * a.H * a.W, a.H, a.W, top, bottom, left, right), sh);
    if (threadIdx.x == 0) {
        a.out[0] = p1.v; a.out[1] = (float)p1.r; a.out[2] = (float)p1.c; a.out[3] = (float)s1;
        a.out[4] = p2.v; a.out[5] = (float)p2.r; a.out[6] = (float)p2.c; a.out[7] = 0.f;
    }
}

__global__ __launch_bounds__(256) void k_max2d(const float* a, float* max_val, long long* argmax, int H, int W) {
    __shared__ Peak sh[4];
    const Peak p = block_peak(scan_map(a + (long)blockIdx.x * H * W, H, W, 0, 0, 0, 0), sh);
    if (threadIdx.x == 0) {
        max_val[blockIdx.x] = p.v;
        argmax[2 * (long)blockIdx.x] = p.r;
        argmax[2 * (long)blockIdx.x + 1] = p.c;
    }
}

// ---------------------------------------------------------------------------------------------------
// Localisation with the outcome decided on the device ("one kernel returning 6 scalars").
// Everything `localize_advanced` derives from the score maps is produced here: both peaks, the case
// (normal / hard_negative / uncertain / not_found, dimp.py:258-303), the displacement of the peak that case selects and
// its translation vector.  The host only supplies per-frame constants (thresholds, the per-scale neighbourhood and the
// previous target offset) and reads 16 floats back -- written straight into pinned host memory when `out` points there.
//
// Arithmetic types follow what the reference's expressions evaluate to under torch: `tensor.item() < python float`
// compares doubles (thresholds on max1); `python float * float32 tensor` and `float32 tensor > python float` are float32
// operations with the scalar rounded to float32 first (distractor / hard-negative / displacement tests).  Products and
// sums are kept un-fused (__fmul_rn / __fadd_rn): torch rounds after every operation.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_localize_decide(DecideArgs a) {
    __shared__ Peak sh[4];
    localize_decide(a, sh, a.out);
    if (threadIdx.x != 0) return;
    __threadfence_system();                                              // results visible before the sequence number
    __hip_atomic_store(a.out + 15, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// Per-frame constants of the localisation from the tracker's host state (dimp.py:241-244,268,285,291).  Every product /
// quotient is a float32 operation in the reference (float32 CPU tensors; Python scalars are rounded to float32 by
// torch's binary ops), so plain `float` arithmetic with contraction off reproduces the values bit for bit.
#pragma clang fp contract(off)
extern "C" int pt_localize_constants_f32(const pt_localize_state* st, int S, int H, int W, pt_localize_params* q) {
    if (!st || !q) return PT_ERR_NULL;
    if (S <= 0 || H <= 0 || W <= 0) return PT_ERR_SHAPE;
    if (S > 8) return PT_ERR_UNSUPPORTED;
    q->target_not_found_threshold = st->target_not_found_threshold;
    q->uncertain_threshold = st->uncertain_threshold;
    q->hard_sample_threshold = st->hard_sample_threshold;
    q->distractor_threshold = (float)st->distractor_threshold;
    q->hard_negative_threshold = (float)st->hard_negative_threshold;
    q->target_not_found_f32 = (float)st->target_not_found_threshold;
    q->disp_threshold = (float)(st->dispalcement_scale * sqrt((double)(H * W)) / 2);
    const float out_r = (float)H - fmodf(st->kernel_size[0] + 1.0f, 2.0f);          // output_sz = score_sz - (K + 1) % 2
    const float out_c = (float)W - fmodf(st->kernel_size[1] + 1.0f, 2.0f);
    q->center_r = ((float)H - 1.0f) / 2.0f;
    q->center_c = ((float)W - 1.0f) / 2.0f;
    q->ratio_r = st->img_support_sz[0] / out_r;
    q->ratio_c = st->img_support_sz[1] / out_c;
    const float tns = (float)st->target_neighborhood_scale;
    const float back_r = out_r / st->img_support_sz[0], back_c = out_c / st->img_support_sz[1];
    for (int s = 0; s < 8; ++s) {
        const bool in = s < S;
        const float sc = in ? st->sample_scales[s] : 1.0f;
        q->scale[s] = sc;
        q->neigh_r[s] = in ? (tns * (st->target_sz[0] / sc)) * back_r : 0.f;
        q->neigh_c[s] = in ? (tns * (st->target_sz[1] / sc)) * back_c : 0.f;
        q->prev_r[s] = in ? (st->pos[0] - st->sample_pos[2 * s]) / (q->ratio_r * sc) : 0.f;
        q->prev_c[s] = in ? (st->pos[1] - st->sample_pos[2 * s + 1]) / (q->ratio_c * sc) : 0.f;
    }
    return PT_OK;
}

int pt_localize_launch(const float* scores, const float* scores_hn, const pt_localize_params* prm, float* out16, int S, int H,
                       int W, float seq, void* stream) {
    if (!scores || !prm || !out16) return PT_ERR_NULL;
    if (S <= 0 || H <= 0 || W <= 0) return PT_ERR_SHAPE;
    if (S > 8) return PT_ERR_UNSUPPORTED;
    DecideArgs a{};
    a.scores = scores; a.scores_hn = scores_hn ? scores_hn : scores; a.out = out16; a.S = S; a.H = H; a.W = W;
    a.seq = seq;
    a.p = *prm;
    hipLaunchKernelGGL(k_localize_decide, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" int pt_localize_decide_f32(const float* scores, const float* scores_hn, const pt_localize_params* prm,
                                      float* out16, int S, int H, int W, void* stream) {
    return pt_localize_launch(scores, scores_hn, prm, out16, S, H, W, 0.f, stream);
}

extern "C" int pt_localize_advanced_f32(const float* scores, const float* scores_hn, const pt_localize_state* st,
                                        float* out16, int S, int H, int W, void* stream) {
    pt_localize_params q;
    const int rc = pt_localize_constants_f32(st, S, H, W, &q);
    if (rc) return rc;
    return pt_localize_launch(scores, scores_hn, &q, out16, S, H, W, 0.f, stream);
}

// The same, and the call returns when the 16 results are readable by the host: `out16_host` must be pinned host memory
// the device can write (hipHostMalloc / torch pin_memory).  The kernel stores a per-call sequence number into
// out16_host[15] after everything else (system-scope release); the host polls that word instead of paying a
// hipStreamSynchronize round trip through the runtime -- the tracker needs these numbers before it can do anything
// else, so this wait IS the frame's critical path.  Falls back to hipStreamSynchronize after 2 s without the word.
extern "C" int pt_localize_advanced_sync_f32(const float* scores, const float* scores_hn, const pt_localize_state* st,
                                             float* out16_host, int S, int H, int W, void* stream) {
    if (!out16_host) return PT_ERR_NULL;
    if (!pt_pinned_host_checked(out16_host) || pt_stream_is_capturing(stream)) return PT_ERR_UNSUPPORTED;
    pt_localize_params q;
    int rc = pt_localize_constants_f32(st, S, H, W, &q);
    if (rc) return rc;
    volatile float* word = out16_host + 15;
    const float seq = pt_next_seq(word);
    rc = pt_localize_launch(scores, scores_hn, &q, out16_host, S, H, W, seq, stream);
    if (rc) return rc;
    return pt_poll_word(word, seq, out16_host, stream);
}

extern "C" int pt_max2d_f32(const float* a, float* max_val, long long* argmax, int n, int H, int W, void* stream) {
    if (!a || !max_val || !argmax) return PT_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return PT_ERR_SHAPE;
    hipLaunchKernelGGL(k_max2d, dim3(n), dim3(256), 0, (hipStream_t)stream, a, max_val, argmax, H, W);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" int pt_localize_f32(const float* scores, const float* scores_hn, const float* neigh_rows,
                               const float* neigh_cols, float* out8, int S, int H, int W, void* stream) {
    if (!scores || !neigh_rows || !neigh_cols || !out8) return PT_ERR_NULL;
    if (S <= 0 || H <= 0 || W <= 0) return PT_ERR_SHAPE;
    if (S > 8) return PT_ERR_UNSUPPORTED;
    LocArgs a{};
    a.scores = scores; a.scores_hn = scores_hn ? scores_hn : scores; a.out = out8; a.S = S; a.H = H; a.W = W;
    for (int s = 0; s < S; ++s) { a.nh[s] = neigh_rows[s]; a.nw[s] = neigh_cols[s]; }
    hipLaunchKernelGGL(k_localize, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    PT_CHECK_LAUNCH();
    return PT_OK;
}
