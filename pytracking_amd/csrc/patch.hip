// Image-patch sampling in front of the backbone (reference: pytracking/features/preprocessing.py:54-148 `sample_patch`):
// integer pre-downsampling (strided view) -> crop with replicate padding -> bilinear resize, as ONE gather kernel.
// The integer geometry (down-sampling factor, crop corners, 'inside' shifts) is computed by the caller exactly as the
// reference computes it on the host; this kernel reproduces F.pad(mode='replicate') + F.interpolate(mode='bilinear',
// align_corners=False) of the strided view, including ATen's float arithmetic (scale = in / out in float,
// src = scale * (dst + 0.5) - 0.5 clamped at 0, row blend of column blends, no fused multiply-add).
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
        float fy = __fsub_rn(__fmul_rn(sh, (float)y + 0.5f), 0.5f), fx = __fsub_rn(__fmul_rn(sw, (float)x + 0.5f), 0.5f);
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
