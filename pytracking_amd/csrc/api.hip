// C-ABI entry points that are thin compositions of the kernels: filter-layer ops and the benchmark frame.
#include "common.h"
#include "pt_internal.h"

extern "C" const char* pt_strerror(int status) {
    switch (status) {
        case PT_OK: return "ok";
        case PT_ERR_NULL: return "required pointer is NULL";
        case PT_ERR_SHAPE: return "non-positive or inconsistent dimension";
        case PT_ERR_UNSUPPORTED: return "configuration not covered by the gfx950 kernels";
        case PT_ERR_WORKSPACE: return "workspace too small or not 256-byte aligned";
        case PT_ERR_LAUNCH: return "HIP launch failed";
        default: return "unknown status";
    }
}

extern "C" int pt_abi_version(void) { return 1; }

// ---------------------------------------------------------------------------------------------------
// apply_filter / apply_feat_transpose
// ---------------------------------------------------------------------------------------------------
static int filt_check(int n, int C, int H, int W, int KH, int KW, int OH, int OW) {
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || OH <= 0 || OW <= 0) return PT_ERR_SHAPE;
    if (OH > H + 2 * (KH / 2) - KH + 1 || OW > W + 2 * (KW / 2) - KW + 1) return PT_ERR_SHAPE;
    if (KH * KW > 16) return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

extern "C" size_t pt_apply_filter_ws_bytes(int n, int C, int H, int W, int KH, int KW, int OH, int OW) {
    if (filt_check(n, C, H, W, KH, KW, OH, OW)) return 0;
    PtPlan p = pt_make_plan(n, C, H, W, KH, KW, OH, OW);
    return pt_align_floats(pt_spart_floats(p)) * sizeof(float);
}

extern "C" int pt_apply_filter_f32(const float* feat, long feat_stride_n, const float* filt, float* scores, int n, int C,
                                   int H, int W, int KH, int KW, int OH, int OW, void* ws, size_t ws_bytes,
                                   void* stream) {
    if (!feat || !filt || !scores || !ws) return PT_ERR_NULL;
    int rc = filt_check(n, C, H, W, KH, KW, OH, OW);
    if (rc) return rc;
    if (feat_stride_n < (long)C * H * W) return PT_ERR_SHAPE;
    PtPlan p = pt_make_plan(n, C, H, W, KH, KW, OH, OW);
    if (ws_bytes < pt_align_floats(pt_spart_floats(p)) * sizeof(float) || ((uintptr_t)ws % 256) != 0)
        return PT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* spart = (float*)ws;
    rc = pt_launch_corr(p, feat, feat_stride_n, filt, spart, st);
    if (rc) return rc;
    return pt_launch_sum_slices(spart, scores, p.KS, (size_t)n * p.OO, st);
}

struct FtCarve { size_t R, gpart, total; };
static FtCarve ft_carve(const PtPlan& p) {
    FtCarve c;
    c.R = 0;
    c.gpart = pt_align_floats(pt_R_floats(p));
    c.total = c.gpart + pt_align_floats(pt_gpart_floats(p));
    return c;
}

extern "C" size_t pt_feat_transpose_ws_bytes(int n, int C, int H, int W, int KH, int KW, int OH, int OW) {
    if (filt_check(n, C, H, W, KH, KW, OH, OW)) return 0;
    PtPlan p = pt_make_plan(n, C, H, W, KH, KW, OH, OW);
    return ft_carve(p).total * sizeof(float);
}

extern "C" int pt_feat_transpose_f32(const float* feat, long feat_stride_n, const float* inp, float* grad, int n, int C,
                                     int H, int W, int KH, int KW, int OH, int OW, void* ws, size_t ws_bytes,
                                     void* stream) {
    if (!feat || !inp || !grad || !ws) return PT_ERR_NULL;
    int rc = filt_check(n, C, H, W, KH, KW, OH, OW);
    if (rc) return rc;
    if (feat_stride_n < (long)C * H * W) return PT_ERR_SHAPE;
    PtPlan p = pt_make_plan(n, C, H, W, KH, KW, OH, OW);
    FtCarve cv = ft_carve(p);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* R = (float*)ws + cv.R;
    float* gpart = (float*)ws + cv.gpart;
    rc = pt_launch_build_R(p, inp, R, st);
    if (rc) return rc;
    rc = pt_launch_adj(p, feat, feat_stride_n, R, gpart, st);
    if (rc) return rc;
    return pt_launch_sum_slices(gpart, grad, p.KSPL, (size_t)C * p.KK, st);
}

// ---------------------------------------------------------------------------------------------------
// benchmark frame: classify -> arg-max -> memory insert -> solve   
// ---------------------------------------------------------------------------------------------------
// sums the classification partials, finds the first maximum (torch.max semantics, pytracking/libs/dcf.py:156-164)
// and re-centres the box of memory slot `slot` on it (inverse of the centre formula of optimizer.py:112-113).
__global__ __launch_bounds__(512) void k_classify_fin(const float* __restrict__ spart, int KS, int OH, int OW,
                                                      float* __restrict__ scores, float* __restrict__ peak,
                                                      float* __restrict__ mem_bb, int slot, float feat_stride, int K) {
    __shared__ float bv[8];
    __shared__ int bi[8];
    const int OO = OH * OW;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int o = threadIdx.x; o < OO; o += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < KS; ++k) s += spart[(long)k * OO + o];
        scores[o] = s;
        if (s > best) { best = s; besti = o; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(besti, off, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 8; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        const int row = besti / OW, col = besti - row * OW;
        peak[0] = (float)row;
        peak[1] = (float)col;
        const float off = (float)(K % 2) * 0.5f;
        float* b = mem_bb + 4 * slot;
        b[0] = ((float)col + off) * feat_stride - b[2] * 0.5f;
        b[1] = ((float)row + off) * feat_stride - b[3] * 0.5f;
    }
}

struct TfCarve { size_t spart1, w_iters, sd, total; };
static TfCarve tf_carve(int n, int C, int H, int W, int K) {
    const int OH = H + (K + 1) % 2, OW = W + (K + 1) % 2;
    PtPlan p1 = pt_make_plan(1, C, H, W, K, K, OH, OW);
    TfCarve c;
    size_t off = 0;
    c.spart1 = off; off += pt_align_floats(pt_spart_floats(p1));
    c.w_iters = off; off += pt_align_floats((size_t)(64 + 1) * C * K * K);
    c.sd = off; off += pt_sd_ws_bytes(n, C, H, W, K) / sizeof(float);
    c.total = off;
    return c;
}

extern "C" size_t pt_track_frame_ws_bytes(int n, int C, int H, int W, int K) {
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
    return tf_carve(n, C, H, W, K).total * sizeof(float);
}

extern "C" int pt_track_frame_f32(const pt_sd_params* prm, float* filter, float* mem_feat, float* mem_bb,
                                  const float* sample_weight, const float* test_feat, int slot, int n, int C, int H,
                                  int W, int K, int num_iter, float* scores_out, float* peak_out, void* ws,
                                  size_t ws_bytes, void* stream) {
    if (!prm || !filter || !mem_feat || !mem_bb || !test_feat || !scores_out || !peak_out || !ws) return PT_ERR_NULL;
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || num_iter < 0 || slot < 0 || slot >= n) return PT_ERR_SHAPE;
    if (K * K > 16 || (H * W) % 4 != 0 || num_iter > 64) return PT_ERR_UNSUPPORTED;
    TfCarve cv = tf_carve(n, C, H, W, K);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int OH = H + (K + 1) % 2, OW = W + (K + 1) % 2;
    const long CHW = (long)C * H * W;
    float* base = (float*)ws;
    // 1. classify the test frame with the current filter (dimp.py:190-194 -> linear_filter.py:75-80)
    PtPlan p1 = pt_make_plan(1, C, H, W, K, K, OH, OW);
    int rc = pt_launch_corr(p1, test_feat, CHW, filter, base + cv.spart1, st);
    if (rc) return rc;
    // 2. localise on device and re-centre the box of the slot about to be overwritten
    hipLaunchKernelGGL(k_classify_fin, dim3(1), dim3(512), 0, st, base + cv.spart1, p1.KS, OH, OW, scores_out, peak_out,
                       mem_bb, slot, prm->feat_stride, K);
    PT_CHECK_LAUNCH();
    // 3. memory insert (dimp.py:429-441 update_memory)
    if (hipMemcpyAsync(mem_feat + (long)slot * CHW, test_feat, CHW * sizeof(float), hipMemcpyDeviceToDevice, st) !=
        hipSuccess)
        return PT_ERR_LAUNCH;
    // 4. re-optimise the filter over the whole memory (dimp.py:633-639)
    float* w_iters = base + cv.w_iters;
    rc = pt_sd_solve_f32(prm, filter, mem_feat, CHW, mem_bb, sample_weight, n, C, H, W, K, num_iter, w_iters, nullptr,
                         base + cv.sd, (cv.total - cv.sd) * sizeof(float), st);
    if (rc) return rc;
    if (hipMemcpyAsync(filter, w_iters + (long)num_iter * C * K * K, (size_t)C * K * K * sizeof(float),
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
        return PT_ERR_LAUNCH;
    return PT_OK;
}
