// Internal launch plans + launcher prototypes shared by the translation units of libpt_hot.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/pt_hot.h"

// Geometry of the two feature passes for one (n, C, H, W, KH, KW) problem.
//   corr pass:  grid (n, KS): one workgroup per (sample, channel slice) -> partial score maps
//               spart[KS][n][OH*OW]; the consumer sums the KS slices in a fixed order.
//   adj  pass:  grid (ceil(C/16), KSPL): one workgroup per (16-channel block, slice of the flattened
//               n*H*W position axis in groups of 16) -> gpart[KSPL][C*KK]; consumer sums KSPL slices.
struct PtPlan {
    int n, C, H, W, KH, KW, OH, OW;
    int HW, KK, OO;
    int KS, cper;        // corr: channel slices, channels per slice (multiple of 4)
    int corr_threads;    // corr block size (one wave per 64-position tile, max 16 waves)
    size_t corr_lds;     // dynamic LDS bytes of the corr kernel
    int NG;              // adj: number of 16-position groups = ceil(n*HW/16)
    int KSPL, gper;      // adj: position slices, groups per slice
    bool vec4;           // HW % 4 == 0 -> 16-byte feature loads
};

PtPlan pt_make_plan(int n, int C, int H, int W, int KH, int KW, int OH, int OW);

// sizes in floats
static inline size_t pt_spart_floats(const PtPlan& p) { return (size_t)p.KS * p.n * p.OO; }
static inline size_t pt_R_floats(const PtPlan& p) { return (size_t)p.NG * 256; }
static inline size_t pt_gpart_floats(const PtPlan& p) { return (size_t)p.KSPL * p.C * p.KK; }
static inline size_t pt_align_floats(size_t x) { return (x + 63) & ~(size_t)63; }   // 256-byte carve granule

// Optional stages fused into the correlation pass (all pointers may be null).
struct PtCorrFuse {
    const float* gpart;   // filter operand = sum_k gpart[k] + reg*w instead of `filt` (KSPL slices of C*KK floats)
    int KSPL;
    const float* w;
    float reg;
    float* g_out;         // (C*KK) reduced gradient, written by the workgroups of sample 0
    float* anum_part;     // (KS)   per-channel-slice sum of g^2, written by the workgroups of sample 0
    float* copy_dst;      // (C,H,W) destination of a copy of sample 0's features (requires n == 1)
    // second problem of the same shape in the same launch (grid.z = 2; generic k_corr only): ATOM's joint Gauss-Newton needs
    // conv_same(c, p_x) and conv_same(dc, x) side by side (optim.py:30-44) -- one launch instead of two dependent ones
    const float* feat2;
    const float* filt2;
    float* spart2;
};

// Input gradient of conv2d(mode='same') for a single filter bank: out[i,k,yy,xx] = sum_{u,v} v[i, yy-u+p, xx-v+p] * filt[k,u,v]
// (the back-projection of ATOM's joint Gauss-Newton, pytracking/tracker/atom/optim.py:46-62 through autograd in the reference).
// It reads the same map as the filter adjoint and is independent of it, so it rides on k_adj's launch as grid.z layers >= 1.
struct PtInputGrad { const float* v; const float* filt; float* out; int n, Kc, H, W, K; };

int pt_launch_corr(const PtPlan& p, const float* feat, long stride_n, const float* filt, float* spart, hipStream_t st,
                   const PtCorrFuse* fuse = nullptr);
int pt_launch_adj(const PtPlan& p, const float* feat, long stride_n, const float* R, float* gpart, hipStream_t st,
                  const PtInputGrad* ig = nullptr);
// scores[i][o] = sum_ks spart[ks][i][o]
int pt_launch_sum_slices(const float* part, float* out, int slices, size_t count, hipStream_t st);
// R (im2col of the residual map, MFMA-B layout) from inp (n, OH, OW)
int pt_launch_build_R(const PtPlan& p, const float* inp, float* R, hipStream_t st);

// ---- XCD-aligned fast path (fast_passes.hip): C in {128,256,512,1024}, H*W % 4 == 0 -------------------------
struct PtFast {
    int ok;
    int n, C, H, W, KH, KW, OH, OW, HW, KK, OO, Q;
    int CX, NK, TF, rem, tiles, left, HWp, corr_threads, nh;   // corr2: grid KSC*n/spw, waves = spw samples x nh halves x tiles
    int spw;                                                    // samples per k_corr2 workgroup (2: sample pairs, see pt_fast_plan)
    int sa_direct;                                              // k_corr2's 4x4 shift-and-add may use uniform tap steps (LDS slack guaranteed)
    int KSC;                                                    // channel ranges = partial score maps per sample: 8 (one per XCD) or 16
    size_t corr_lds;
    int CB, bpx, NG, KSPL, gper, U, PH, PW, ns_max, E, zn; // adj2: grid CB*KSPL, 8 waves x U contiguous groups; zero block
    size_t adj_lds;
};
PtFast pt_fast_plan(int n, int C, int H, int W, int KH, int KW, int OH, int OW);
// The ONE predicate for "this call can take the XCD-aligned path": the shape plan plus what the raw 16-byte buffer loads
// need from the pointers (shared by pt_apply_filter_f32, pt_feat_transpose_f32, pt_sd_solve_impl, pt_track_frame_f32;
// when it is false every caller takes its generic branch).  `filt` / `src` may be null.
static inline bool pt_fast_usable(const PtFast& f, const void* feat, long stride_n, const void* filt = nullptr,
                                  const void* src = nullptr) {
    return f.ok && ((uintptr_t)feat % 16) == 0 && (stride_n % 4) == 0 && (long)f.n * stride_n * 4 < (1L << 31) &&
           ((uintptr_t)src % 16) == 0 && (f.KK != 16 || ((uintptr_t)filt % 16) == 0);
}
static inline size_t pt_fast_spart_floats(const PtFast& p) { return (size_t)16 * p.n * p.OO; }   // room for KSC = 16
static inline size_t pt_fast_gpart_floats(const PtFast& p) { return (size_t)p.KSPL * p.C * p.KK; }
// spart[x][i][OH*OW], x = XCD channel range (8 slices).  `slot`/`src`/`copy_dst`: sample `slot` is read from `src`
// and stored to `copy_dst` while it streams (src == nullptr: no override).
// `lazy`: the filter operand is the deferred last update of the previous solve (k_corr2 FUSE < 0): w_out = w_prev - step*alpha*g with
// alpha from qs (n values, at anum + 64) and anum; the workgroups of sample 0 store it to w_out.  4x4 filters, n <= 511 only.
struct PtCorrLazy { const float* w_prev; const float* g; const float* anum; const float* qs; float step, reg_eps; float* w_out; };
int pt_launch_corr2(const PtFast& p, const float* feat, long stride_n, const float* filt, float* spart, hipStream_t st,
                    const PtCorrFuse* fuse = nullptr, int slot = -1, const float* src = nullptr,
                    float* copy_dst = nullptr, const PtCorrLazy* lazy = nullptr);
int pt_launch_adj2_plain(const PtFast& p, const float* feat, long stride_n, const float* inp, float* gpart,
                         hipStream_t st);
struct SdArgs;
//   fuse_init (t == 0 only): the pass also does the init stage (slices -> s_0, classification epilogue, maps); see pt_adj2_init_fusable
int pt_launch_adj2_sd(const PtFast& p, const float* feat, long stride_n, const SdArgs& sd, int t, int want_loss,
                      hipStream_t st, bool fuse_init = false);
bool pt_adj2_init_fusable(const PtFast& p, const SdArgs& sd);

// ---- multi-filter passes (mf_kernels.hip): F <= 16 filters, odd K, C % 16 == 0 -----------------------------
size_t pt_mf_gpart_floats(int n, int F, int C, int H, int W, int K);     // 0: configuration not covered
int pt_mf_groups(int n, int F, int C, int H, int W, int K);              // sample groups of the adjoint partials
size_t pt_mf_wt_floats(int C, int K);                                    // pre-transposed weight table
// position of weight (filter f, channel c, tap) in that table: [c/4][lane = (c%4)*16 + f][12 taps (9 used) | 1 tap]
__host__ __device__ inline long pt_mf_wt_index(int c, int f, int tap, int KK) {
    return ((long)(c >> 2) * 64 + (c & 3) * 16 + f) * (KK == 1 ? 1 : 12) + tap;
}
int pt_launch_mf_wtrans(const float* filt, float* wT, int F, int C, int K, hipStream_t st, int groups = 1,
                        float* clear2 = nullptr);   // clear2: a second table of pt_mf_wt_floats zeroed by the same launch
// out_stride_n / inp_stride_n: floats between consecutive samples of scores / inp (0 = dense F*H*W); lets a group of
// <= 16 filters be a slice of a wider (n, Ftotal, H, W) tensor; groups > 1: `groups` consecutive banks of F filters in ONE
// launch (grid.z) -- weight tables pt_mf_wt_floats apart, maps F*H*W apart inside a sample, adjoint partials
// pt_mf_gpart_floats apart
int pt_mf_corr_tm_splits(int n, int Ftot, int C, int H, int W);             // channel splits that fill the chip; 0: not covered
int pt_mf_corr_tm_cost(int n, int Ftot, int C, int H, int W);               // rounds x channel chunks per workgroup
int pt_launch_mf_corr_tm(const float* feat, long stride_n, const float* w_tap_major, float* part, int n, int Ftot, int C,
                         int H, int W, int ksplit, hipStream_t st);          // 3x3, weights (Ftot, 9, C); partial maps
int pt_launch_mf_corr1_direct(const float* feat, long stride_n, const float* filt, float* scores, int n, int Ftot, int C,
                              int H, int W, hipStream_t st, long out_stride_n);     // 1x1, weights (Ftot, C) untransposed
int pt_mf_corr_splits(int n, int F, int C, int H, int W, int K);               // channel splits used with a partial workspace
size_t pt_mf_corr_part_floats(int n, int F, int C, int H, int W, int K);         // its size (0: no split for this shape)
// `sq`: optional rider of the channel-split reduction (LWL few-shot learner: |sw * F g|^2 of steepestdescent.py:76-80): when the
// launch goes through the partial-map workspace, the kernel that sums the splits also leaves PT_MF_SQ_PARTS partial sums of
// (weight * score)^2 in sq->out and *sq->done = 1; otherwise sq is left alone (the caller runs its own reduction).
#define PT_MF_SQ_PARTS 256
struct PtMfSq { const float* sw; int sw_mode; float sw_scalar; long per_image; float* out; int* done; };
int pt_launch_mf_corr(const float* feat, long stride_n, const float* wT, float* scores, int n, int F, int C, int H,
                      int W, int K, hipStream_t st, long out_stride_n = 0, int groups = 1, float* part = nullptr,
                      const PtMfSq* sq = nullptr);
int pt_launch_mf_adj(const float* feat, long stride_n, const float* inp, float* gpart, int n, int F, int C, int H, int W,
                     int K, hipStream_t st, long inp_stride_n = 0, int groups = 1);

#define PT_CHECK_LAUNCH()                                    \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return PT_ERR_LAUNCH; \
    } while (0)

// classification epilogue of the benchmark frame (runs inside the solver's maps launch)
struct PtClsFin {
    const float* spart;   // (KS, OH*OW) classification partials
    int KS, slot;
    float *scores, *peak, *mem_bb;
    void* after_init;     // optional hipEvent_t recorded behind the launch that writes scores / peak (frame_full.hip forks there)
    const int* slot_dyn;  // optional: the slot is read from this device int by the init stage (graph-replayed frames); `slot` then only bounds checks
};

// Frame chains (pt_track_frame_chain_f32): `pend` in-out, nullable.  On entry pend->iters > 0 = the previous solve on this workspace
// left its last update pending (that many iterations; its step length and reg + alpha_eps): it is applied in the prologue of this
// solve's first correlation and stored to w_final.  `defer`: this solve leaves ITS last update pending (pend is filled) instead of
// launching it.  Both need the fast path (PT_ERR_UNSUPPORTED otherwise, nothing queued) and w_final == w_in (the chain's filter).
int pt_sd_solve_impl(const pt_sd_params* prm, const float* w_in, const float* feat, long feat_stride_n, const float* bb,
                     const float* sample_weight, int n, int C, int H, int W, int K, int num_iter, float* w_iters,
                     float* losses, void* ws, size_t ws_bytes, hipStream_t st, bool copy_w0, float* w_final,
                     const PtClsFin* cls, const float* src, pt_frame_pending* pend = nullptr, bool defer = false);
// the pending update on its own (end of a chain)
int pt_sd_flush_impl(const pt_frame_pending* pend, float* filter, int n, int C, int H, int W, int K, float* w_iters, void* ws,
                     size_t ws_bytes, hipStream_t st);

int pt_sd_replay_impl(const pt_sd_params* prm, const float* w_in, const float* feat, long feat_stride_n, const float* bb,
                      const float* sample_weight, int n, int C, int H, int W, int K, int num_iter, float* w_iters, void* ws,
                      size_t ws_bytes, int which, int reps, hipStream_t stream);

// Two pyramid levels of one image, same RoIs, one launch each way (prroi.hip; used by the IoU refinement):
//   fwd2:      out[l] (R, C[l], PH[l], PH[l]) = chan_scale[l][c] * PrRoIPool(feat[l])
//   bwd_coor2: part[l][(r * slices + s) * 4 + {x0,y0,x1,y1}] = partial coordinate gradients, summed by the consumer
int pt_launch_prroi_fwd2(const float* const feat[2], const float* const chan_scale[2], float* const out[2],
                         const int C[2], const int H[2], const int W[2], const int PH[2], const float scale[2],
                         const float* rois, int R, hipStream_t st);
int pt_launch_prroi_bwd_coor2(const float* const grad_out[2], const float* const feat[2], float* const part[2],
                              const int C[2], const int H[2], const int W[2], const int PH[2], const float scale[2],
                              const float* rois, int R, int slices, hipStream_t st);


// Result buffers of the *_sync_* / full-frame entry points (pinned host memory the device writes and the host polls).  The pointer
// class is verified once per buffer; the verified set is small, lock-free, shared by all threads AND by all translation units (it
// lives in api.hip).  A buffer is dropped from the set by pt_host_buffer_forget() (the C ABI's release hook) and whenever a poll runs
// into its time-out, so a pinned buffer that was freed and whose address came back as pageable / device memory is verified again.
bool pt_pinned_host_checked(const void* p);
// spin on `word` until it holds `seq` (system-scope acquire); after 2 s: forget the buffer, hipStreamSynchronize, re-check
int pt_poll_word(volatile float* word, float seq, const void* buf, void* stream);
// next value of a result buffer's sequence word (an exactly representable integer >= 1)
static inline float pt_next_seq(volatile float* word) {
    float seq = *word + 1.0f;
    if (!(seq >= 1.0f && seq < 8388608.0f)) seq = 1.0f;
    return seq;
}

// pt_track_frame_head_f32 with an event recorded as soon as the classification scores are queued (api.hip)
int pt_track_frame_head_impl(const pt_sd_params* prm, float* filter, float* mem_feat, float* mem_bb, const float* sample_weight,
                             const float* backbone_feat, const float* head_weight_tap_major, float norm_scale, float norm_eps, int slot,
                             int n, int Cin, int C, int H, int W, int K, int num_iter, float* scores_out, float* peak_out, void* ws,
                             size_t ws_bytes, void* stream, void* after_init_event, const int* slot_dyn = nullptr);

// Two timing-less events per (current device, main stream, auxiliary stream) triple -- fork: recorded on `main_stream`, join: recorded on
// `aux_stream` for `main_stream` to wait on -- created on first use and never destroyed (they hold no memory); thread-safe.  Keyed by the
// device as well (a stream address recycled on another GPU must not meet an event of the first one) and by the PAIR (two host threads
// that share an auxiliary stream under different main streams do not share events).  Either out pointer may be null.
bool pt_stream_events(void* main_stream, void* aux_stream, hipEvent_t* fork, hipEvent_t* join);

// launch halves of the host-polled entry points, for compositions (frame_full.hip)
int pt_localize_launch(const float* scores, const float* scores_hn, const pt_localize_params* prm, float* out16, int S, int H,
                       int W, float seq, void* stream);
int pt_iou_refine_launch(const pt_iou_dims* d, const float* params, const float* prepared, const float* c3, const float* c4,
                         const float* mod3, const float* mod4, const float* init_boxes_dev, float* boxes_out, float* iou_out, int P,
                         int num_iter, const float* step_length4, float step_decay, int relative, int backtrack, void* ws,
                         size_t ws_bytes, float seq, float* seq_word, void* stream, const void* frame_mid = nullptr,
                         const void* frame_mid_dev = nullptr, const float* seq_dyn = nullptr);

// pt_clf_head_f32 with the memory slot read from a device int (graph-replayed one-call frame); tomp.hip
int pt_clf_head_impl(const float* feat, const float* weight_tap_major, float* out, int n, int Cin, int Cout, int H, int W,
                     float norm_scale, float eps, void* ws, size_t ws_bytes, void* stream, const int* slot_dyn, long slot_stride);

// the argument / shape / route checks of a refinement call, nothing queued (iou_refine.hip)
int pt_iou_refine_validate(const pt_iou_dims* d, const float* params, const float* prepared, const float* c3, const float* c4,
                           const float* mod3, const float* mod4, bool have_init_boxes, const float* boxes_out, const float* iou_out, int P,
                           int num_iter, const float* step_length4, const void* ws, size_t ws_bytes, bool boxes_on_host, float seq,
                           bool with_mid);

// A host-polled result cannot be waited for while the stream is being captured into a graph (nothing executes): refuse at once
// instead of spinning into the 2 s fallback.
static inline bool pt_stream_is_capturing(void* stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}
