/*
 * pt_hot.h -- C ABI of libpt_hot.so: the MI355X (gfx950) implementation of PyTracking's per-frame
 * online model-optimisation hot path.
 *
 * The reference (visionml/pytracking) has no FFI of its own on this path -- its operator API is a set of
 * Python callables.  Each entry point below names the reference callable it replaces
 * (paths relative to the reference root); the host-side mirror that binds them lives in
 * pytracking_amd/ (ctypes), and INTEGRATION.md shows the stub a maintainer would add to the reference.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer to contiguous fp32 unless stated;
 *   - the caller owns every buffer; the library never allocates or frees device memory and keeps no
 *     global state; scratch space is passed in (`ws`, sized by the matching *_ws_bytes());
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream), no device
 *     synchronisation, no host callbacks -> every call is hipGraph-capturable;
 *   - returns PT_OK (0) or a negative pt_status; nothing is launched when an argument check fails;
 *   - layouts are the reference's: feature maps NCHW (n, C, H, W), filters (C, KH, KW), boxes xywh.
 */
#ifndef PT_HOT_H
#define PT_HOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_NULL = -1,        /* required pointer is NULL */
    PT_ERR_SHAPE = -2,       /* non-positive or inconsistent dimension */
    PT_ERR_UNSUPPORTED = -3, /* valid request the kernels do not cover (e.g. KH*KW > 16 in the fused solver) */
    PT_ERR_WORKSPACE = -4,   /* workspace too small / misaligned */
    PT_ERR_LAUNCH = -5       /* hipGetLastError() != hipSuccess after a launch */
} pt_status;

const char* pt_strerror(int status);
/* ABI version; bumped when a signature changes. */
int pt_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Filter layer -- ltr/models/layers/filter.py
 * ---------------------------------------------------------------------------------------------- */

/* apply_filter (filter.py:5-57), one sequence, one filter:
 *   scores[i,y,x] = sum_{c,u,v} feat[i,c,y+u-KH/2,x+v-KW/2] * filt[c,u,v]      (zero padded)
 * feat (n,C,H,W) with sample stride `feat_stride_n` floats (= C*H*W when dense; S*C*H*W for a 5-D
 * (n,S,C,H,W) tensor, the caller then loops over sequences); filt (C,KH,KW); scores (n,OH,OW) dense.
 * OH/OW may be the full correlation size H+2*(KH/2)-KH+1 (reference apply_filter) or anything smaller:
 * the top-left OHxOW corner is produced (pytracking/libs/operation.py:17-32 conv2d(mode='same')).
 * Requires KH*KW <= 16. */
size_t pt_apply_filter_ws_bytes(int n, int C, int H, int W, int KH, int KW, int OH, int OW);
int pt_apply_filter_f32(const float* feat, long feat_stride_n, const float* filt, float* scores,
                        int n, int C, int H, int W, int KH, int KW, int OH, int OW,
                        void* ws, size_t ws_bytes, void* stream);

/* apply_feat_transpose (filter.py:91-182; _v2 and _v3 compute the same quantity):
 *   grad[c,u,v] = sum_{i,y,x} feat[i,c,y+u-KH/2,x+v-KW/2] * inp[i,y,x]
 * inp (n,OH,OW) dense; grad (C,KH,KW).  Requires KH*KW <= 16 (16-byte loads when (H*W) % 4 == 0). */
size_t pt_feat_transpose_ws_bytes(int n, int C, int H, int W, int KH, int KW, int OH, int OW);
int pt_feat_transpose_f32(const float* feat, long feat_stride_n, const float* inp, float* grad,
                          int n, int C, int H, int W, int KH, int KW, int OH, int OW,
                          void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Steepest-descent filter optimisers -- ltr/models/target_classifier/optimizer.py
 * ---------------------------------------------------------------------------------------------- */

enum { PT_SD_DIMP = 0, PT_SD_DIMP_L2 = 1, PT_SD_PRDIMP = 2 };
enum { PT_ACT_RELU = 0, PT_ACT_BENTPAR = 1 };       /* score_act (optimizer.py:74-82) */
enum { PT_MASK_SIGMOID = 0, PT_MASK_LINEAR = 1 };   /* mask_act  (optimizer.py:58-66) */

/* Host-side parameter block, read at call time (the trackers mutate the module attributes at run
 * time, pytracking/tracker/dimp/dimp.py:589-602, so nothing is baked in). */
typedef struct pt_sd_params {
    int   kind;            /* PT_SD_* */
    float step_length;     /* exp(log_step_length)                               optimizer.py:108 */
    float reg;             /* max(filter_reg^2, min_filter_reg^2)                optimizer.py:109 */
    float alpha_eps;       /*                                                    optimizer.py:156 */
    float feat_stride;     /*                                                    optimizer.py:113 */
    /* PT_SD_DIMP: learned radial look-up tables = the three 1x1 conv weights (optimizer.py:57-72).
     * DEVICE pointers to `num_bins` floats each. */
    int   num_bins;
    float bin_displacement;
    const float* label_lut;
    const float* mask_lut;
    const float* spatial_lut;
    int   mask_act;        /* PT_MASK_* */
    int   score_act;       /* PT_ACT_*  */
    float act_param;       /* BentIdentPar b */
    /* PT_SD_DIMP_L2 (optimizer.py:174-291) */
    float gauss_sigma;     /* also PT_SD_PRDIMP */
    float hinge_threshold;
    /* PT_SD_PRDIMP (optimizer.py:294-439) */
    float uni_weight;
    int   normalize_label;
    float label_shrink;
    int   has_softmax_reg;
    float softmax_reg;
    float label_threshold;
} pt_sd_params;

/* {DiMPSteepestDescentGN, DiMPL2SteepestDescentGN, PrDiMPSteepestDescentNewton}.forward for one
 * sequence (optimizer.py:85-170, 211-291, 355-439):
 *   w_in (C,K,K) is not modified; w_iters receives all num_iter+1 iterates ((num_iter+1),C,K,K),
 *   w_iters[0] = w_in; losses (num_iter+1 floats) may be NULL (= compute_losses False);
 *   bb (n,4) xywh in crop pixels; sample_weight (n) or NULL.
 * Requires K*K <= 16 (feature loads are 16-byte wide when (H*W) % 4 == 0, scalar otherwise). */
size_t pt_sd_ws_bytes(int n, int C, int H, int W, int K);
int pt_sd_solve_f32(const pt_sd_params* p, const float* w_in, const float* feat, long feat_stride_n,
                    const float* bb, const float* sample_weight,
                    int n, int C, int H, int W, int K, int num_iter,
                    float* w_iters, float* losses, void* ws, size_t ws_bytes, void* stream);

/* S independent sequences in one call (optimizer.py:101-104 `num_sequences`; KYS / multi-object wrappers call the optimiser with
 * S > 1).  Arrays of S pointers (host arrays of device pointers); sample_weight / losses may be NULL as arrays; ws: S workspaces of
 * ws_bytes_each >= pt_sd_ws_bytes each.  Sequence s runs on `stream` (s % (1 + n_aux) == 0) or on aux_streams[s % (1 + n_aux) - 1]:
 * the auxiliary streams wait for everything queued on `stream` before the call and `stream` waits for them before the call
 * returns, so the call is ordered on `stream` like pt_sd_solve_f32.  n_aux = 0: the sequences run back to back. */
int pt_sd_solve_batch_f32(const pt_sd_params* p, int S, const float* const* w_in, const float* const* feat, long feat_stride_n,
                          const float* const* bb, const float* const* sample_weight, int n, int C, int H, int W, int K,
                          int num_iter, float* const* w_iters, float* const* losses, void* const* ws, size_t ws_bytes_each,
                          void* stream, void* const* aux_streams, int n_aux);

/* ------------------------------------------------------------------------------------------------
 * Multi-filter filter layer + LWL few-shot learner -- ltr/models/layers/filter.py (5-D `filter` / `input` branches),
 * ltr/models/meta/steepestdescent.py, ltr/models/lwl/loss_residual_modules.py
 *   F <= 16 filters per sequence, K in {1, 3} (zero padding K/2: output H x W), W <= 256, groups = 1, no dilation.
 * ---------------------------------------------------------------------------------------------- */

/* apply_filter (filter.py:29-34), one sequence, F filters:  filt (F,C,K,K); scores (n,F,H,W) dense
 *   scores[i,f,y,x] = sum_{c,u,v} filt[f,c,u,v] * feat[i,c,y+u-K/2,x+v-K/2] */
size_t pt_apply_filter_mf_ws_bytes(int n, int F, int C, int H, int W, int K);
int pt_apply_filter_mf_f32(const float* feat, long feat_stride_n, const float* filt, float* scores,
                           int n, int F, int C, int H, int W, int K, void* ws, size_t ws_bytes, void* stream);

/* apply_feat_transpose for a 5-D input (filter.py:158-176):  inp (n,F,H,W) dense; grad (F,C,K,K)
 *   grad[f,c,u,v] = sum_{i,y,x} feat[i,c,y+u-K/2,x+v-K/2] * inp[i,f,y,x] */
size_t pt_feat_transpose_mf_ws_bytes(int n, int F, int C, int H, int W, int K);
int pt_feat_transpose_mf_f32(const float* feat, long feat_stride_n, const float* inp, float* grad,
                             int n, int F, int C, int H, int W, int K, void* ws, size_t ws_bytes, void* stream);

/* GNSteepestDescent.forward (steepestdescent.py:32-105) on LWTLResidual (loss_residual_modules.py:16-41), one sequence:
 *   residuals [sw*(apply_filter(feat,w) - label), filter_reg*w]; w_in (F,C,K,K) is not modified; w_iters receives the
 *   num_iter+1 iterates; losses (num_iter+1 floats) may be NULL; label (n,F,H,W);
 *   sw_mode 0: sample_weight ignored (sqrt(1/n), :27-28), 1: (n) per image, 2: (n,F,H,W) per element (:29-33). */
size_t pt_lwl_ws_bytes(int n, int F, int C, int H, int W, int K);
int pt_lwl_gn_solve_f32(const float* w_in, const float* feat, long feat_stride_n, const float* label,
                        const float* sample_weight, int sw_mode, float filter_reg, float steplength_reg,
                        int n, int F, int C, int H, int W, int K, int num_iter,
                        float* w_iters, float* losses, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ATOM conjugate gradient -- pytracking/libs/optimization.py:72-163,227-289 (ConjugateGradient.run +
 * run_CG) specialised to pytracking/tracker/atom/optim.py:71-99 (ConvProblem) with the MLU response
 * activation (ltr/models/layers/activation.py:20-29).
 *   x (C,K,K) is updated IN PLACE (optimization.py:259-260); samples (n,C,H,W); y (n,H,W);
 *   sample_weights (n).  cg_state: 2*C*K*K+4 floats the caller keeps between calls
 *   (p, r_prev, rho, has_p) -- only consulted when direction_forget_factor != 0.
 * ---------------------------------------------------------------------------------------------- */
size_t pt_atom_cg_ws_bytes(int n, int C, int H, int W, int K);
int pt_atom_cg_f32(float* x, const float* samples, long samples_stride_n, const float* y,
                   const float* sample_weights, float filter_reg, float act_min_val,
                   int n, int C, int H, int W, int K, int num_iter, int fletcher_reeves,
                   float direction_forget_factor, float* cg_state,
                   void* ws, size_t ws_bytes, void* stream);

/* ATOM first-frame joint optimisation -- GaussNewtonCG.run (pytracking/libs/optimization.py:328-421) on
 * FactorizedConvProblem (pytracking/tracker/atom/optim.py:6-68), identity projection activation, MLU response activation:
 *   filter (Kc,K,K) and proj (Kc,M) [= the reference's (Kc,M,1,1)] are updated IN PLACE (optimization.py:403-404);
 *   samples (n,M,H,W) uncompressed features; y (n,H,W); sample_weights (n);
 *   cg_iters: HOST array of num_gn CG iteration counts, one Gauss-Newton iteration each (optimization.py:340-358);
 *   CG state is reset at every Gauss-Newton iteration (direction_forget_factor = 0, the tracker's setting). */
size_t pt_atom_gn_ws_bytes(int n, int M, int Kc, int H, int W, int K);
int pt_atom_gn_f32(float* filter, float* proj, const float* samples, long samples_stride_n, const float* y,
                   const float* sample_weights, float filter_reg, float projection_reg, float act_min_val,
                   int n, int M, int Kc, int H, int W, int K, const int* cg_iters, int num_gn, int fletcher_reeves,
                   void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ToMP transformer model predictor -- replaces, for inference,
 *   FilterPredictor.predict_filter / predict_cls_bbreg_filters_parallel  (ltr/models/transformer/filter_predictor.py:50-150)
 *   Transformer.forward, post-norm encoder/decoder layers                (ltr/models/transformer/transformer.py:90-96,172-180,224-238)
 *   PositionEmbeddingSine('lin_sine', avoid_aliazing=True)               (ltr/models/transformer/position_encoding.py:6-58)
 *   DenseBoxRegressor.forward and the Linear of LinearFilterClassifier   (ltr/models/transformer/heads.py:93-98,119-141)
 *
 * Parameters travel as ONE contiguous fp32 device buffer ("pack"), tensors in the reference's own layouts, in this order
 * (names = the reference's state_dict keys below `filter_predictor.`):
 *   for each encoder layer i:  transformer.encoder.layers.i.{self_attn.in_proj_weight (3D,D), self_attn.in_proj_bias (3D),
 *       self_attn.out_proj.weight (D,D), self_attn.out_proj.bias (D), linear1.weight (F,D), linear1.bias (F),
 *       linear2.weight (D,F), linear2.bias (D), norm1.weight, norm1.bias, norm2.weight, norm2.bias (D each)}
 *   for each decoder layer i:  transformer.decoder.layers.i.{self_attn.[4 tensors as above], multihead_attn.[4 tensors],
 *       linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.*, norm2.*, norm3.* (weight, bias)}
 *   transformer.decoder.norm.{weight, bias}
 *   box_encoding.0.{weight (D/4,4), bias}, box_encoding.1.{weight, bias, running_mean, running_var} (D/4 each),
 *   box_encoding.3.{weight (D,D/4), bias}, box_encoding.4.{weight, bias, running_mean, running_var} (D each),
 *   box_encoding.6.{weight (D,D), bias}, query_embed_fg.weight (D), query_embed_test.weight (D)
 * pt_tomp_param_floats() returns the length of that buffer (0: configuration not covered).
 *
 * Shapes: train_feat (n_train, n_seq, D, H, W); test_feat (1, n_seq, D, H, W); train_label (n_train, n_seq, H, W);
 * train_ltrb (n_train, n_seq, 4, H, W); pos (H*W, D) from pt_tomp_posenc_f32 (input independent: compute once, keep).
 *   parallel = 0: predict_filter            -> filters (n_seq, D), enc_feat (n_seq, D, H, W)
 *   parallel = 1: predict_cls_bbreg_filters_parallel (n_seq must be 1, as the reference's mask row 1 assumes):
 *                 filters (2, D) = [cls, bbreg], enc_feat (2, D, H, W) = [cls, bbreg]; the memory frames
 *                 [num_gth_frames, n_train) are masked out as keys for the bbreg row (filter_predictor.py:134-136)
 * Covered: d_model in {128,256,384,512}, head dim in {16,32,64}, dim_ff % 64 == 0, <= 16 layers each, <= 8 batch rows.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pt_tomp_dims {
    int d_model, nhead, dim_ff, n_enc, n_dec;
    int H, W;          /* feature map of every frame */
    int max_res;       /* `feature_sz` of FilterPredictor (anti-aliasing factor of the positional encoding) */
} pt_tomp_dims;
size_t pt_tomp_param_floats(const pt_tomp_dims* dims);
/* Weight-only products of the decoder, folded once per weight update into a second caller-owned buffer of
 * pt_tomp_prepared_floats() floats (one query token per batch row makes self-attention, the query/key product and
 * value + out_proj plain matrix-vector products: W_o W_v, W_k,h^T W_q,h / sqrt(d_h), W_o[:,h] W_v,h and their biases; the
 * decoder starts from zeros, so the first layer's state after self-attention and its folded query are constants of the
 * weights and are stored here as well). */
size_t pt_tomp_prepared_floats(const pt_tomp_dims* dims);
int pt_tomp_prepare_f32(const pt_tomp_dims* dims, const float* params, float* prepared, void* stream);
int pt_tomp_posenc_f32(float* pos, int H, int W, int d_model, int max_res, void* stream);
size_t pt_tomp_predict_ws_bytes(const pt_tomp_dims* dims, int n_train, int n_seq, int parallel);
int pt_tomp_predict_f32(const pt_tomp_dims* dims, const float* params, const float* prepared, const float* pos,
                        const float* train_feat,
                        const float* test_feat, const float* train_label, const float* train_ltrb,
                        int n_train, int n_seq, int parallel, int num_gth_frames, float* filters, float* enc_feat,
                        void* ws, size_t ws_bytes, void* stream);
/* y (B,N) = [relu](x (B,K) weight(N,K)^T + bias): the Linear applied to the predicted filters (heads.py:95,123), B <= 8 */
int pt_tomp_linear_f32(const float* weight, const float* bias, const float* x, float* y, int B, int N, int K, int relu,
                       void* stream);
/* DenseBoxRegressor.forward (heads.py:119-141): feat (n, D, H, W), filter (D) -> ltrb (n, 4, H, W).
 * pack: linear.weight (D,D), linear.bias (D), then for tower conv i = 0..3: conv weight in (out, ky, kx, in) order
 * [= the reference's (out,in,3,3) tensor permuted (0,2,3,1)], conv bias (D), GroupNorm weight (D), GroupNorm bias (D);
 * then bbreg_layer weight (4, ky, kx, in) and bias (4). */
size_t pt_tomp_bbreg_param_floats(int d_model);
size_t pt_tomp_bbreg_ws_bytes(int n, int d_model, int H, int W);
int pt_tomp_bbreg_f32(const float* params, const float* feat, const float* filter, float* ltrb, int n, int d_model,
                      int H, int W, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Classification-feature head -- `residual_bottleneck(num_blocks=0, final_conv=True, l2norm=True)`
 * (ltr/models/target_classifier/features.py:49-73): Conv2d(Cin, Cout, 3, padding=1, bias=False) followed by
 * InstanceL2Norm(size_average=True, eps, scale) (ltr/models/layers/normalization.py:15-20), as DiMP-50 (1024 -> 512,
 * dimpnet.py:169-172) and ToMP (1024 -> 256, tompnet.py:99-102; run on three frames per tracked frame) build it.
 *   feat (n, Cin, H, W); weight_tap_major (Cout, ky, kx, Cin) = the reference's (Cout, Cin, 3, 3) tensor permuted
 *   (0,2,3,1); out (n, Cout, H, W) = conv * norm_scale * sqrt(Cout*H*W / (sum over the image of conv^2 + eps)).
 * Covered: Cin % 64 == 0, Cout % 4 == 0.
 * ---------------------------------------------------------------------------------------------- */
size_t pt_clf_head_ws_bytes(int n, int Cin, int Cout, int H, int W);
int pt_clf_head_f32(const float* feat, const float* weight_tap_major, float* out, int n, int Cin, int Cout, int H, int W,
                    float norm_scale, float eps, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Target localisation on the device -- the score-map arithmetic of DiMP/ToMP `localize_advanced`
 * (pytracking/tracker/dimp/dimp.py:238-303, pytracking/tracker/tomp/tomp.py) and `dcf.max2d`
 * (pytracking/libs/dcf.py:156-164), one launch, results left on the device for a single copy.
 *   pt_max2d_f32: a (n, H, W) -> max_val (n), argmax (n, 2) int64 [row, col]; ties: smallest column, then smallest row
 *     (the order torch.max over rows, then over columns produces).
 *   pt_localize_f32: scores (S, H, W) [S <= 8 scales], scores_hn (S, H, W) or NULL (= scores): the map the second
 *     peak is searched in (`perform_hn_without_windowing`, dimp.py:247-250); neigh_rows / neigh_cols: HOST arrays of S
 *     floats = target_neigh_sz of each scale in score cells (dimp.py:268).
 *     out8 (device) = [max1, row1, col1, scale_ind, max2, row2, col2, 0]: first peak over all scales (first scale on
 *     ties), second peak of scores_hn[scale_ind] with rows [round(row1 - nr/2), round(row1 + nr/2 + 1)) x the same in
 *     columns, clamped to the map, ZEROED (dimp.py:270-278; Python round() = half-to-even).
 * ---------------------------------------------------------------------------------------------- */
int pt_max2d_f32(const float* a, float* max_val, long long* argmax, int n, int H, int W, void* stream);
int pt_localize_f32(const float* scores, const float* scores_hn, const float* neigh_rows, const float* neigh_cols,
                    float* out8, int S, int H, int W, void* stream);

/* pt_localize_decide_f32: the whole of `localize_advanced` in one launch -- both peaks
 * as pt_localize_f32 finds them, then the outcome of dimp.py:258-303 decided on the device.  The caller supplies the
 * per-frame constants below (plain host arithmetic on the tracker's state; no device data is needed to form them):
 *   thresholds on the first peak are compared as doubles (`max_score1.item() < threshold`), the remaining tests are the
 *   reference's float32 tensor expressions (`python float * float32 tensor`, `float32 tensor > python float`);
 *   center = (score_sz - 1) / 2, ratio = img_support_sz / output_sz (dimp.py:241-244), per scale s: scale[s] =
 *   sample_scales[s], neigh = target_neigh_sz (dimp.py:268), prev = prev_target_vec (dimp.py:285), disp_threshold =
 *   dispalcement_scale * sqrt(H * W) / 2 (dimp.py:291).  Absent optional thresholds are passed as -infinity.
 * out16 (device memory OR device-visible pinned host memory; the kernel writes it directly) =
 *   [code, scale_ind, row, col, translation_row, translation_col, max1, row1, col1, max2, row2, col2, peak_chosen, 0, 0, seq]
 *   with code = PT_LOC_*, (row, col) the displacement (`max_disp`) of the chosen peak and translation =
 *   (disp - center) * ratio * scale[scale_ind] in float32 (dimp.py:256,282). */
enum { PT_LOC_NORMAL = 0, PT_LOC_HARD_NEGATIVE = 1, PT_LOC_UNCERTAIN = 2, PT_LOC_NOT_FOUND = 3 };
typedef struct {
    double target_not_found_threshold, uncertain_threshold, hard_sample_threshold;
    float distractor_threshold, hard_negative_threshold, target_not_found_f32, disp_threshold;
    float center_r, center_c, ratio_r, ratio_c;
    float scale[8], neigh_r[8], neigh_c[8], prev_r[8], prev_c[8];
} pt_localize_params;
int pt_localize_decide_f32(const float* scores, const float* scores_hn, const pt_localize_params* prm, float* out16,
                           int S, int H, int W, void* stream);

/* The tracker's host state `localize_advanced` reads (all HOST values): `params.*` thresholds as the Python floats they
 * are (absent optional ones: -infinity), the float32 contents of `kernel_size`, `img_support_sz`, `target_sz`, `pos`
 * ([row, col]) and per scale `sample_scales[s]`, `sample_pos[s] = (row, col)`.
 *   pt_localize_constants_f32: pt_localize_params from that state, in the reference's float32 operation order (host
 *     only: no launch, no device access).
 *   pt_localize_advanced_f32 = constants + pt_localize_decide_f32: one call per frame. */
typedef struct {
    double target_not_found_threshold, uncertain_threshold, hard_sample_threshold, distractor_threshold,
        hard_negative_threshold, target_neighborhood_scale, dispalcement_scale;
    float kernel_size[2], img_support_sz[2], target_sz[2], pos[2];
    float sample_scales[8], sample_pos[16];
} pt_localize_state;
int pt_localize_constants_f32(const pt_localize_state* st, int S, int H, int W, pt_localize_params* prm);
int pt_localize_advanced_f32(const float* scores, const float* scores_hn, const pt_localize_state* st, float* out16,
                             int S, int H, int W, void* stream);
/* pt_localize_advanced_sync_f32: the same, returning when the results are readable by the host.  out16_host MUST be
 * pinned host memory the device can write (hipHostMalloc, torch `pin_memory()`); anything else -> unsupported.  The
 * kernel stores a per-call sequence number into out16_host[15] last and the host polls that word (no runtime
 * synchronisation call on the frame's critical path); only work queued on `stream` BEFORE this launch is waited for. */
int pt_localize_advanced_sync_f32(const float* scores, const float* scores_hn, const pt_localize_state* st,
                                  float* out16_host, int S, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * IoU-guided box refinement -- DiMP.optimize_boxes_default / optimize_boxes_relative
 * (pytracking/tracker/dimp/dimp.py:725-788) on AtomIoUNet.predict_iou (ltr/models/bbreg/atom_iou_net.py:96-136):
 * gradient ascent on the predicted IoU w.r.t. the P proposal boxes, all iterations on the device.
 *   params pack (fp32, the reference's state_dict tensors below `bb_regressor.`, this order): fc3_rt.linear.weight
 *     (I3, C3*25), fc3_rt.linear.bias, fc3_rt.bn.{weight, bias, running_mean, running_var}, fc4_rt.linear.weight
 *     (I4, C4*9), fc4_rt.linear.bias, fc4_rt.bn.{...}, iou_predictor.weight (I3+I4), iou_predictor.bias (1).
 *   prepared: the two Linear weights transposed, written by pt_iou_prepare_f32 once per weight update.
 *   c3 (C3,H3,W3), c4 (C4,H4,W4): IoU features of ONE test image (get_iou_feat); mod3 (C3), mod4 (C4): modulation
 *   vectors (get_modulation); init_boxes / boxes_out (P,4) xywh in image coordinates; iou_out (P) = the prediction of the
 *   last forward pass, as the reference returns it.  step_length4: HOST [s,s,s,s] or [s0,s0,s1,s1] (dimp.py:737-738);
 *   relative = 1: optimise [cx/sw, cy/sh, log w, log h] with [sw, sh] = size of the first box (dimp.py:767-768).
 *   backtrack = 0: DiMP -- one step length for all proposals, multiplied by step_decay after every iteration;
 *   backtrack = 1: ATOM.optimize_boxes (pytracking/tracker/atom/atom.py:758-836) -- a proposal whose predicted IoU did
 *     not improve multiplies ITS step length by step_decay and takes its previous step back (no-op when step_decay >= 1).
 *   Pools are the reference's: 5x5 at 1/8 on c3, 3x3 at 1/16 on c4.  Covered: C3*25, C4*9, I3, I4 multiples of 32, P <= 256.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pt_iou_dims { int C3, C4, I3, I4, H3, W3, H4, W4; } pt_iou_dims;
size_t pt_iou_param_floats(const pt_iou_dims* dims);
size_t pt_iou_prepared_floats(const pt_iou_dims* dims);
int pt_iou_prepare_f32(const pt_iou_dims* dims, const float* params, float* prepared, void* stream);
size_t pt_iou_refine_ws_bytes(const pt_iou_dims* dims, int P);
int pt_iou_refine_f32(const pt_iou_dims* dims, const float* params, const float* prepared, const float* c3, const float* c4,
                      const float* mod3, const float* mod4, const float* init_boxes, float* boxes_out, float* iou_out,
                      int P, int num_iter, const float* step_length4, float step_decay, int relative, int backtrack,
                      void* ws, size_t ws_bytes, void* stream);

/* The same for the per-frame call of the trackers (proposals formed on the host, refined boxes needed on the host:
 * dimp.py:691-722, atom.py:724-756): init_boxes_host = P <= 16 boxes (xywh) in HOST memory, passed inside a kernel argument
 * block; out_host = PT_IOU_HOST_FLOATS floats of pinned host memory: boxes [0, 4P), IoU [64, 64+P), sequence word [95].
 * Returns when the results are readable (polls the sequence word; no hipStreamSynchronize).  PT_ERR_UNSUPPORTED for
 * P > 16 or out_host not device-writable host memory: use pt_iou_refine_f32. */
#define PT_IOU_HOST_FLOATS 96
int pt_iou_refine_sync_f32(const pt_iou_dims* dims, const float* params, const float* prepared, const float* c3, const float* c4,
                           const float* mod3, const float* mod4, const float* init_boxes_host, float* out_host, int P,
                           int num_iter, const float* step_length4, float step_decay, int relative, int backtrack, void* ws,
                           size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Precise RoI Pooling -- replaces ltr/external/PreciseRoIPooling (empty git submodule;
 * import sites ltr/models/target_classifier/initializer.py:4,18,45 and
 * ltr/models/bbreg/atom_iou_net.py:4,31-32,41-42,126-127,157,160).
 *   features (N,C,H,W); rois (R,5) = [batch_idx, x0, y0, x1, y1]; out / grad_out (R,C,PH,PW).
 *   bwd_feat ACCUMULATES into grad_features (caller zeroes it); bwd_coor writes grad_rois (R,5),
 *   column 0 = 0.
 * ---------------------------------------------------------------------------------------------- */
int pt_prroi_fwd_f32(const float* features, const float* rois, float* out,
                     int N, int C, int H, int W, int R, int PH, int PW, float spatial_scale, void* stream);
int pt_prroi_bwd_feat_f32(const float* grad_out, const float* rois, float* grad_features,
                          int N, int C, int H, int W, int R, int PH, int PW, float spatial_scale, void* stream);
int pt_prroi_bwd_coor_f32(const float* grad_out, const float* features, const float* rois, float* grad_rois,
                          int N, int C, int H, int W, int R, int PH, int PW, float spatial_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One synthetic tracking frame of the benchmark workload (BASELINE.json metric):
 *   classify(test_feat) -> on-device arg-max -> overwrite memory slot `slot` (features + box centred on
 *   the peak) -> steepest-descent solve over the n memory samples, all on `stream`, no host sync.
 * mem_feat (n,C,H,W), mem_bb (n,4), sample_weight (n), filter (C,K,K) in/out, scores_out (OH*OW),
 * peak_out (2 floats: row, col).
 * ---------------------------------------------------------------------------------------------- */
size_t pt_track_frame_ws_bytes(int n, int C, int H, int W, int K);
int pt_track_frame_f32(const pt_sd_params* p, float* filter, float* mem_feat, float* mem_bb,
                       const float* sample_weight, const float* test_feat, int slot,
                       int n, int C, int H, int W, int K, int num_iter,
                       float* scores_out, float* peak_out, void* ws, size_t ws_bytes, void* stream);

/* Frame CHAINS (round 6).  pt_track_frame_f32 leaves the last iterate in `filter` before its launches end: the solve's last
 * statement, w_T = w_{T-1} - step*alpha_T*g_T (optimizer.py:155-160), is one dependent launch whose ONLY consumer is the next
 * frame's first correlation (classification + s_0).  In a chain of frames on one stream and one workspace that launch can ride on
 * its consumer: with defer != 0 the call leaves (w_{T-1}, g_T, the operands of alpha_T) in the workspace and describes them in
 * *pending; the next pt_track_frame_chain_f32 call (same filter / workspace / n, C, H, W, K) finds pending->iters > 0, forms w_T in
 * the prologue of its first correlation -- the same expressions on the same operands: the same bits as pt_track_frame_f32 -- and
 * stores it to `filter`; pt_track_frame_flush_f32 applies a pending update on its own (end of the chain, or before anything else
 * reads `filter`).  While an update is pending `filter` holds w_0 of the last solve, NOT its result.
 *   *pending: caller-owned host struct, zero-initialised at the start of a chain; in-out.  A solve of fewer than 2 iterations is
 *   never deferred (pending->iters stays 0).  4x4 filters on the XCD-aligned path only: PT_ERR_UNSUPPORTED otherwise with nothing
 *   queued -- fall back to pt_track_frame_f32 (after a flush). */
typedef struct {
    int iters;            /* 0: nothing pending; else the iteration count T of the solve whose last update is pending */
    float step_length;    /* of that solve */
    float reg_eps;        /* its reg + alpha_eps */
} pt_frame_pending;
int pt_track_frame_chain_f32(const pt_sd_params* p, float* filter, float* mem_feat, float* mem_bb,
                             const float* sample_weight, const float* test_feat, int slot,
                             int n, int C, int H, int W, int K, int num_iter,
                             float* scores_out, float* peak_out, void* ws, size_t ws_bytes,
                             pt_frame_pending* pending, int defer, void* stream);
int pt_track_frame_flush_f32(pt_frame_pending* pending, float* filter, int n, int C, int H, int W, int K,
                             void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The tracking frame with the classification-feature head in front.
 * Replaces: ltr/models/target_classifier/features.py:66-72 (final 3x3 conv + InstanceL2Norm of the test frame) feeding
 * pytracking/tracker/dimp/dimp.py:190-194, 429-441, 605-648 (classify, memory insert, re-optimisation) in one call: the
 * head writes the normalised (C,H,W) feature straight into memory slot `slot`, which the first correlation of the solve
 * then reads like any other sample -- the test feature never exists as a separate tensor.
 *   backbone_feat (Cin,H,W); head_weight_tap_major (C,3,3,Cin) as for pt_clf_head_f32; everything else as pt_track_frame_f32.
 * ---------------------------------------------------------------------------------------------- */
size_t pt_track_frame_head_ws_bytes(int n, int Cin, int C, int H, int W, int K);
int pt_track_frame_head_f32(const pt_sd_params* prm, float* filter, float* mem_feat, float* mem_bb,
                            const float* sample_weight, const float* backbone_feat, const float* head_weight_tap_major,
                            float norm_scale, float norm_eps, int slot, int n, int Cin, int C, int H, int W, int K,
                            int num_iter, float* scores_out, float* peak_out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The WHOLE DiMP frame behind the backbone as one call (SURVEY.md section 8f items 1-3 joined to the frame above):
 *   head -> classify + memory insert + re-optimisation (pt_track_frame_head_f32) -> localize_advanced on the score map
 *   -> the tracker's glue between localisation and refinement ON THE DEVICE -> IoU-guided refinement of the proposals,
 * with ONE host wait at the end.  Replaces the call sequence of `DiMP.track` (pytracking/tracker/dimp/dimp.py:95-131):
 *   get_classification_features (:312-314), classify_target (:190-194), localize_target -> localize_advanced (:196-303),
 *   `new_pos = sample_pos[scale_ind] + translation_vec` (:118), update_state (:486-495), refine_target_box (:650-678:
 *   get_iounet_box :498-504, the jittered proposals :663-675), optimize_boxes_default / _relative (:725-788),
 * whose two host round trips (the localisation result, the refined boxes) and the Python between them are the frame's
 * critical path once the kernels are fast.  The proposals depend on the localisation result, so that dependency moves to
 * the device (`k_frame_glue`): the same float32 operations in the reference's order; the frame's random numbers
 * (`torch.rand(num_init_random_boxes, 4)`, :671) are drawn by the HOST generator as the reference does and passed in.
 * When the localisation says `not_found` the reference skips the refinement (:121): here it still runs and the caller
 * ignores boxes / IoU (the flag is in the result block).  S = 1 scale (the trackers' configuration).
 *
 * out_host: PT_FRAME_HOST_FLOATS floats of pinned host memory (device-writable):
 *   [0,16)   the 16 localisation results of pt_localize_advanced_f32
 *   [16,18)  self.pos after update_state (row, col);  [18,22) init_box (x, y, w, h) of get_iounet_box
 *   [32,32+4P) refined boxes, [96,96+P) predicted IoU, [127] sequence word the call polls.   P = 1 + num_random <= 16.
 * ---------------------------------------------------------------------------------------------- */
#define PT_FRAME_HOST_FLOATS 128
typedef struct pt_frame_glue {
    float image_sz[2];                 /* self.image_sz (rows, cols) */
    float img_sample_sz[2];            /* self.img_sample_sz */
    double target_inside_ratio;        /* params.target_inside_ratio (default 0.2) */
    double box_jitter_pos, box_jitter_sz;   /* params.box_jitter_pos / box_jitter_sz */
    int use_classifier;                /* params.use_classifier (default 1): update_state(new_pos) before the refinement */
    int num_random;                    /* params.num_init_random_boxes, 0 .. 15 */
    float rand_u[60];                  /* torch.rand(num_random, 4) of this frame, row-major */
} pt_frame_glue;
typedef struct pt_frame_full {
    /* head + solver: exactly the arguments of pt_track_frame_head_f32 */
    const pt_sd_params* sd;
    float *filter, *mem_feat, *mem_bb;
    const float *sample_weight, *backbone_feat, *head_weight_tap_major;
    float norm_scale, norm_eps;
    int slot, n, Cin, C, H, W, K, num_iter;
    float *scores_out, *peak_out;
    /* localisation + glue */
    const pt_localize_state* loc;
    const pt_frame_glue* glue;
    /* refinement: exactly the arguments of pt_iou_refine_f32 (boxes come from the glue) */
    const pt_iou_dims* iou_dims;
    const float *iou_params, *iou_prepared, *c3, *c4, *mod3, *mod4;
    int iou_iter, relative;
    float step_length4[4];
    float step_decay;
    /* optional second HIP stream.  The localisation + glue + refinement chain needs only the classification scores, which exist
     * after the frame's first correlation; with aux_stream != NULL that chain forks there (event) and runs on aux_stream CONCURRENTLY
     * with the steepest-descent iterations on `stream`, and `stream` joins it before the call's launches end (so everything queued on
     * `stream` afterwards is ordered behind both chains).  Valid because in this synthetic frame the new
     * sample's label box comes from the classification peak; `DiMP.track` labels the update with the REFINED state (dimp.py:139-145),
     * which makes its update depend on the refinement: leave aux_stream NULL for that order.  On return the result block is
     * host-visible; the filter / memory are complete in `stream` order, not necessarily yet: the call returns when the refinement
     * chain has written the sequence word, the steepest-descent tail on `stream` may still be running, so reading filter / mem_bb /
     * scores from the host or from another stream right after the return needs a wait on `stream` first.
     * Because the update is REORDERED relative to dimp.py:139-145, a call with aux_stream set AND num_iter > 0 is refused
     * (PT_ERR_UNSUPPORTED, nothing queued) unless aux_reordered_update_ok is non-zero. */
    void* aux_stream;
    int aux_reordered_update_ok;
    /* Graph replay (round 6).  A captured frame freezes every kernel ARGUMENT; with dyn != NULL (device memory, 16-byte aligned,
     * pt_track_frame_full_dyn_bytes() bytes) the per-frame VALUES -- the memory slot, the tracker state of the localisation and of the
     * glue (pos, target_sz, sample position / scale, thresholds), this frame's random numbers, the sequence number -- are read by the
     * kernels from that block instead, and the captured launches depend on pointers and shapes only.  Per frame the host writes the
     * block with pt_track_frame_full_dyn_fill_f32 into pinned memory, a copy node at the head of the graph moves it to `dyn`, the graph
     * is replayed and pt_host_wait_word_f32(out + 127, seq) waits for the result block.  Pointers (backbone_feat, c3, c4, mod3 / mod4,
     * scores_out, the result block, the workspace) are those of the capture: the caller's backbone writes into fixed buffers.  One
     * stream only (aux_stream must be NULL); num_iter / iou_iter / the proposal count are properties of the captured graph. */
    void* dyn;
} pt_frame_full;
size_t pt_track_frame_full_ws_bytes(const pt_frame_full* f);
int pt_track_frame_full_f32(const pt_frame_full* f, float* out_host, void* ws, size_t ws_bytes, void* stream);
/* the launches without the wait (graph capture, or a caller that waits on the stream itself) */
int pt_track_frame_full_launch_f32(const pt_frame_full* f, float* out, void* ws, size_t ws_bytes, void* stream);
/* graph replay: size of the per-frame block, and its host-side fill for the frame described by *f (slot, loc, glue; seq != 0) --
 * `out` / `ws` as captured; `dyn_host`: pinned host memory the graph's copy node reads.  Nothing is queued. */
size_t pt_track_frame_full_dyn_bytes(void);
int pt_track_frame_full_dyn_fill_f32(const pt_frame_full* f, float seq, float* out, void* ws, size_t ws_bytes, void* dyn_host);
/* wait until *word == seq (a result block's sequence word in pinned host memory; polled, falls back to hipStreamSynchronize after 2 s) */
int pt_host_wait_word_f32(const float* word, float seq, void* stream);

/* Measurement helper (bench.py): `reps` back-to-back launches of a kernel that only READS `floats` floats of `mem` (16-byte aligned)
 * with every CU -- the streaming floor of one feature pass over that footprint.  scratch2048: 2048 floats of device memory. */
int pt_stream_probe_f32(const float* mem, size_t floats, float* scratch2048, int reps, void* stream);

/* Result buffers of the *_sync_* / full-frame entry points are verified once (pinned, device-writable) and remembered;
 * call this before freeing such a buffer so that a later allocation at the same address is verified again. */
void pt_host_buffer_forget(const void* p);

/* ------------------------------------------------------------------------------------------------
 * Image-patch sampling in front of the backbone.
 * Replaces: pytracking/features/preprocessing.py:54-148 `sample_patch` (strided pre-downsampling, crop with replicate
 * padding, F.interpolate(mode='bilinear')) and :33-51 `sample_patch_multiscale` (S scales in one launch).
 *   im   (C, H, W) float image on the device;   out (S, C, OH, OW)
 *   geom one record per scale, the integers the reference computes on the host (preprocessing.py:83-128):
 *        df = pre-downsampling stride, (os0, os1) = its row / column offset, (tl0, tl1) = top-left corner of the crop in
 *        the down-sampled image (may be negative / beyond the border: replicate padding), crop_h x crop_w = crop size.
 * ---------------------------------------------------------------------------------------------- */
#define PT_PATCH_MAX_SCALES 8
typedef struct pt_patch_geom { int df, os0, os1, tl0, tl1, crop_h, crop_w; } pt_patch_geom;
int pt_sample_patch_f32(const float* im, int C, int H, int W, const pt_patch_geom* geom, int S, float* out, int OH, int OW,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * First-frame augmentation set (second half).
 * Replaces: pytracking/features/preprocessing.py:13-30 `sample_patch_transformed` -- the loop `T(im_patch)` over the
 * transforms of pytracking/features/augmentation.py (Identity :39, FlipHorizontal :44, FlipVertical :52, Translation :60,
 * Scale :73, Rotate :111, Blur :128) followed by `Transform.crop_to_output` (:20-37) -- as built by
 * pytracking/tracker/dimp/dimp.py:329-395 `generate_init_samples` (same in atom.py / tomp.py / kys.py).
 *   patch  (C, EH, EW) base patch on the device (the output of pt_sample_patch_f32 at the augmentation size)
 *   desc   T records on the HOST, one per transform (they travel in the kernel arguments):
 *          kind      PT_AUG_*                         (Translation is PT_AUG_IDENTITY with its shift in the pads)
 *          pad_top / pad_left = floor((OH - th) / 2) + shift[0], floor((OW - tw) / 2) + shift[1]   (augmentation.py:30-33)
 *          th, tw    size of the transformed image before the crop (EH, EW except for PT_AUG_SCALE: :84-87)
 *          fs0, fs1, tap_off0, tap_off1   PT_AUG_BLUR: half filter sizes ceil(2 sigma) and the offsets of the 2 fs + 1
 *                    normalised taps of each axis in `taps` (host, n_taps floats)                                     (:135-140)
 *          m[6]      PT_AUG_ROTATE: the 2x3 matrix that maps OUTPUT to INPUT pixels (the inverse of the matrix the
 *                    reference hands to cv.warpAffine, :121-126); parity of this kind is UNPINNED (no cv2 in the image)
 *   out    (T, C, OH, OW)
 * ---------------------------------------------------------------------------------------------- */
enum { PT_AUG_IDENTITY = 0, PT_AUG_FLIP_H = 1, PT_AUG_FLIP_V = 2, PT_AUG_BLUR = 3, PT_AUG_SCALE = 4, PT_AUG_ROTATE = 5 };
typedef struct pt_aug_desc {
    int kind, pad_top, pad_left, th, tw, fs0, fs1, tap_off0, tap_off1, reserved;
    double m[6];
} pt_aug_desc;
#define PT_AUG_MAX_TRANSFORMS 24   /* per launch; longer lists are split by the library */
#define PT_AUG_MAX_TAPS 256
int pt_augment_patches_f32(const float* patch, int C, int EH, int EW, const pt_aug_desc* desc, int T, const float* taps,
                           int n_taps, float* out, int OH, int OW, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement entry (bench.py roofline leg; not part of the reference's API).  Re-issues ONE feature pass of the
 * solve that pt_track_frame_f32 last ran on `ws` -- the launch of its last iteration, same kernel instantiation and
 * operands -- `reps` times back to back on `stream`, so that the caller can bracket the run with ONE HIP event pair
 * (elapsed / reps = kernel duration + one dependent-launch boundary).  Both passes only read the state they were
 * launched on, so the replay leaves the sequence's state untouched.  Needs num_iter >= 2 and the XCD-aligned path.
 *   which = 0: correlation pass with the fused gradient reduction;  which = 1: adjoint pass with the fused update.
 * ---------------------------------------------------------------------------------------------- */
int pt_track_frame_replay_pass_f32(const pt_sd_params* prm, const float* filter, const float* mem_feat,
                                   const float* mem_bb, const float* sample_weight, int n, int C, int H, int W, int K,
                                   int num_iter, void* ws, size_t ws_bytes, int which, int reps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PT_HOT_H */
