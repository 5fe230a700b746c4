// Image-patch sampling in front of the backbone (reference: pytracking/features/preprocessing.py:54-148 `sample_patch`):
// integer pre-downsampling (strided view) -> crop with replicate padding -> bilinear resize, as ONE gather kernel.
// The integer geometry (down-sampling factor, crop corners, 'inside' shifts) is computed by the caller exactly as the
// reference computes it on the host; this kernel reproduces F.pad(mode='replicate') + F.interpolate(mode='bilinear',
// align_corners=False) of the strided view, including ATen's float arithmetic (scale = in / out in float,
// src = fma(scale, dst + 0.5, -0.5) clamped at 0 -- this image's ATen build contracts that expression, see
// oracle/np_oracle.py: sample_patch_pixels -- row blend of column blends, blends unfused).
#include "common.h"
#include "pt_internal.h"

struct PatchArgs {
    const float* im;
    float* out;
    int C, H, W, OH, OW, S;
    pt_patch_geom g[PT_PATCH_MAX_SCALES];
};

__global__ __launch_bounds__(256) void k_sample_patch(PatchArgs a) {
    const long total = (long)a.S * a.C * a.OH * a.OW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % a.OW);
        long r = e / a.OW;
        const int y = (int)(r % a.OH);
        r /= a.OH;
        const int c = (int)(r % a.C);
        const int s = (int)(r / a.C);
        const pt_patch_geom g = a.g[s];
        // F.interpolate(..., mode='bilinear'): source coordinates in the (crop_h x crop_w) padded patch
        const float sh = (float)g.crop_h / (float)a.OH, sw = (float)g.crop_w / (float)a.OW;
        float fy = __fmaf_rn(sh, (float)y + 0.5f, -0.5f), fx = __fmaf_rn(sw, (float)x + 0.5f, -0.5f);   // ATen's build fuses this one
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < g.crop_h - 1 ? 1 : 0), x1 = x0 + (x0 < g.crop_w - 1 ? 1 : 0);
        const float ly1 = fminf(fmaxf(__fsub_rn(fy, (float)y0), 0.f), 1.f), lx1 = fminf(fmaxf(__fsub_rn(fx, (float)x0), 0.f), 1.f);
        const float ly0 = __fsub_rn(1.f, ly1), lx0 = __fsub_rn(1.f, lx1);
        // patch pixel (r, q) = strided view pixel (clamp(tl + r)) = image pixel (os + df * that)
        const int h2 = (a.H - g.os0 + g.df - 1) / g.df, w2 = (a.W - g.os1 + g.df - 1) / g.df;
        auto row = [&](int rr) { return g.os0 + g.df * min(max(g.tl0 + rr, 0), h2 - 1); };
        auto col = [&](int qq) { return g.os1 + g.df * min(max(g.tl1 + qq, 0), w2 - 1); };
        const float* __restrict__ pc = a.im + (long)c * a.H * a.W;
        const int r0 = row(y0), r1 = row(y1), c0 = col(x0), c1 = col(x1);
        const float p00 = pc[(long)r0 * a.W + c0], p01 = pc[(long)r0 * a.W + c1];
        const float p10 = pc[(long)r1 * a.W + c0], p11 = pc[(long)r1 * a.W + c1];
        const float top = __fadd_rn(__fmul_rn(p00, lx0), __fmul_rn(p01, lx1));
        const float bot = __fadd_rn(__fmul_rn(p10, lx0), __fmul_rn(p11, lx1));
        a.out[e] = __fadd_rn(__fmul_rn(top, ly0), __fmul_rn(bot, ly1));
    }
}

extern "C" int pt_sample_patch_f32(const float* im, int C, int H, int W, const pt_patch_geom* geom, int S, float* out, int OH,
                                   int OW, void* stream) {
    if (!im || !geom || !out) return PT_ERR_NULL;
    if (C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || S <= 0) return PT_ERR_SHAPE;
    if (S > PT_PATCH_MAX_SCALES) return PT_ERR_UNSUPPORTED;
    PatchArgs a;
    a.im = im; a.out = out; a.C = C; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.S = S;
    for (int s = 0; s < S; ++s) {
        const pt_patch_geom& g = geom[s];
        if (g.df < 1 || g.crop_h < 1 || g.crop_w < 1 || g.os0 < 0 || g.os1 < 0 || g.os0 >= g.df || g.os1 >= g.df ||
            g.os0 >= H || g.os1 >= W)
            return PT_ERR_SHAPE;
        a.g[s] = g;
    }
    const long total = (long)S * C * OH * OW;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_sample_patch, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    PT_CHECK_LAUNCH();
    return PT_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// First-frame augmentation set (reference: pytracking/features/preprocessing.py:13-30 `sample_patch_transformed`,
// pytracking/features/augmentation.py:11-147): every transform of the list applied to ONE base patch and cropped / padded
// to the output size (`Transform.crop_to_output`, :20-37: F.pad(mode='replicate') with pads floor/ceil((out - in)/2) +- shift,
// negative pads crop) -- one gather launch for the whole list instead of a Python loop of pads, flips, convolutions and
// resizes on the CPU.
//   Identity / Translation (:39-42, :60-71)  out[y][x] = P[clamp(y - pad_top)][clamp(x - pad_left)]
//   FlipHorizontal / FlipVertical (:44-58)   the same on the mirrored patch
//   Blur (:128-147)                          two zero-padded 1-D Gaussian correlations (rows, then columns), then the crop
//   Scale (:73-97)                           F.interpolate(mode='bilinear') to (h_new, w_new) in ATen's float arithmetic
//   Rotate (:111-126)                        cv2.warpAffine(INTER_LINEAR, BORDER_REPLICATE) restated from OpenCV's published
//                                            fixed-point algorithm (coordinates in 1/1024 px, rounded to 1/32 px, bilinear
//                                            weights k/32); PARITY UNPINNED: cv2 is not available where the goldens are made
// ---------------------------------------------------------------------------------------------------------------------
struct AugArgs {
    const float* patch;
    float* out;
    int C, EH, EW, OH, OW, T;
    pt_aug_desc desc[PT_AUG_MAX_TRANSFORMS];        // by value in the kernel arguments: no staging copy, graph-capturable
    float taps[PT_AUG_MAX_TAPS];
};

__device__ __forceinline__ float aug_px(const float* __restrict__ pc, int EH, int EW, int y, int x) {
    return pc[(long)y * EW + x];
}

// one value of the transformed image (th x tw) of transform d at integer (sy, sx), channel plane pc
__device__ __forceinline__ float aug_value(const AugArgs& a, const pt_aug_desc& d, const float* __restrict__ pc, int sy, int sx) {
    // (a lives in the kernel-argument segment: d and the taps are scalar loads)
    const int EH = a.EH, EW = a.EW;
    switch (d.kind) {
    case PT_AUG_IDENTITY: return aug_px(pc, EH, EW, sy, sx);
    case PT_AUG_FLIP_H: return aug_px(pc, EH, EW, sy, EW - 1 - sx);
    case PT_AUG_FLIP_V: return aug_px(pc, EH, EW, EH - 1 - sy, sx);
    case PT_AUG_BLUR: {
        // im1 = corr(P, f0 along rows, zero padding) rounded to float, then corr(im1, f1 along columns)
        const float* __restrict__ f0 = a.taps + d.tap_off0;
        const float* __restrict__ f1 = a.taps + d.tap_off1;
        float acc = 0.f;
        for (int jx = -d.fs1; jx <= d.fs1; ++jx) {
            const int xx = sx + jx;
            if (xx < 0 || xx >= EW) continue;
            float col = 0.f;
            for (int iy = -d.fs0; iy <= d.fs0; ++iy) {
                const int yy = sy + iy;
                if (yy >= 0 && yy < EH) col = __fadd_rn(col, __fmul_rn(f0[iy + d.fs0], aug_px(pc, EH, EW, yy, xx)));
            }
            acc = __fadd_rn(acc, __fmul_rn(f1[jx + d.fs1], col));
        }
        return acc;
    }
    case PT_AUG_SCALE: {
        const float sh = (float)EH / (float)d.th, sw = (float)EW / (float)d.tw;
        float fy = __fmaf_rn(sh, (float)sy + 0.5f, -0.5f), fx = __fmaf_rn(sw, (float)sx + 0.5f, -0.5f);
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < EH - 1 ? 1 : 0), x1 = x0 + (x0 < EW - 1 ? 1 : 0);
        const float ly1 = fminf(fmaxf(__fsub_rn(fy, (float)y0), 0.f), 1.f), lx1 = fminf(fmaxf(__fsub_rn(fx, (float)x0), 0.f), 1.f);
        const float ly0 = __fsub_rn(1.f, ly1), lx0 = __fsub_rn(1.f, lx1);
        const float top = __fadd_rn(__fmul_rn(aug_px(pc, EH, EW, y0, x0), lx0), __fmul_rn(aug_px(pc, EH, EW, y0, x1), lx1));
        const float bot = __fadd_rn(__fmul_rn(aug_px(pc, EH, EW, y1, x0), lx0), __fmul_rn(aug_px(pc, EH, EW, y1, x1), lx1));
        return __fadd_rn(__fmul_rn(top, ly0), __fmul_rn(bot, ly1));
    }
    case PT_AUG_ROTATE: {
        // cv::warpAffine, INTER_LINEAR, BORDER_REPLICATE: d.m = inverse map (dst -> src), AB_BITS = 10, INTER_BITS = 5
        const int round_delta = 16;                                     // AB_SCALE / INTER_TAB_SIZE / 2
        const int X0 = (int)llrint(d.m[0] * (double)sx * 1024.0), Y0 = (int)llrint(d.m[3] * (double)sx * 1024.0);
        const int Xb = (int)llrint((d.m[1] * (double)sy + d.m[2]) * 1024.0) + round_delta;
        const int Yb = (int)llrint((d.m[4] * (double)sy + d.m[5]) * 1024.0) + round_delta;
        const int X = (X0 + Xb) >> 5, Y = (Y0 + Yb) >> 5;
        const int ix = X >> 5, iy = Y >> 5;
        const float ax = (float)(X & 31) * (1.0f / 32.0f), ay = (float)(Y & 31) * (1.0f / 32.0f);
        auto cy = [&](int v) { return min(max(v, 0), EH - 1); };
        auto cx = [&](int v) { return min(max(v, 0), EW - 1); };
        const float p00 = aug_px(pc, EH, EW, cy(iy), cx(ix)), p01 = aug_px(pc, EH, EW, cy(iy), cx(ix + 1));
        const float p10 = aug_px(pc, EH, EW, cy(iy + 1), cx(ix)), p11 = aug_px(pc, EH, EW, cy(iy + 1), cx(ix + 1));
        // remapBilinear<float>: sum of the four taps with weights (1-ax)(1-ay), ax(1-ay), (1-ax)ay, ax*ay
        const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
        return p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11;
    }
    default: return 0.f;
    }
}

__global__ __launch_bounds__(256) void k_augment(AugArgs a) {
    const long per = (long)a.C * a.OH * a.OW;
    const long total = (long)a.T * per;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % a.OW);
        long r = e / a.OW;
        const int y = (int)(r % a.OH);
        r /= a.OH;
        const int c = (int)(r % a.C);
        const int t = (int)(r / a.C);
        const pt_aug_desc& d = a.desc[t];
        const int sy = min(max(y - d.pad_top, 0), d.th - 1), sx = min(max(x - d.pad_left, 0), d.tw - 1);
        a.out[e] = aug_value(a, d, a.patch + (long)c * a.EH * a.EW, sy, sx);
    }
}

extern "C" int pt_augment_patches_f32(const float* patch, int C, int EH, int EW, const pt_aug_desc* desc, int T, const float* taps,
                                      int n_taps, float* out, int OH, int OW, void* stream) {
    if (!patch || !desc || !out) return PT_ERR_NULL;
    if (C <= 0 || EH <= 0 || EW <= 0 || OH <= 0 || OW <= 0 || T <= 0 || n_taps < 0) return PT_ERR_SHAPE;
    if (n_taps > PT_AUG_MAX_TAPS) return PT_ERR_UNSUPPORTED;
    for (int t = 0; t < T; ++t) {
        const pt_aug_desc& d = desc[t];
        if (d.kind < PT_AUG_IDENTITY || d.kind > PT_AUG_ROTATE) return PT_ERR_UNSUPPORTED;
        if (d.th < 1 || d.tw < 1) return PT_ERR_SHAPE;
        if (d.kind != PT_AUG_SCALE && (d.th != EH || d.tw != EW)) return PT_ERR_SHAPE;
        if (d.kind == PT_AUG_BLUR) {
            if (!taps || d.fs0 < 0 || d.fs1 < 0 || d.tap_off0 < 0 || d.tap_off1 < 0 || d.tap_off0 + 2 * d.fs0 + 1 > n_taps ||
                d.tap_off1 + 2 * d.fs1 + 1 > n_taps)
                return PT_ERR_SHAPE;
        }
    }
    const long per = (long)C * OH * OW;
    for (int t0 = 0; t0 < T; t0 += PT_AUG_MAX_TRANSFORMS) {                // transform lists longer than one launch's argument block
        AugArgs a;
        a.patch = patch; a.out = out + (long)t0 * per;
        a.C = C; a.EH = EH; a.EW = EW; a.OH = OH; a.OW = OW;
        a.T = T - t0 < PT_AUG_MAX_TRANSFORMS ? T - t0 : PT_AUG_MAX_TRANSFORMS;
        for (int t = 0; t < a.T; ++t) a.desc[t] = desc[t0 + t];
        for (int k = 0; k < n_taps; ++k) a.taps[k] = taps[k];
        const long total = (long)a.T * per;
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(k_augment, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        PT_CHECK_LAUNCH();
    }
    return PT_OK;
}
