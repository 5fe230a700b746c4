// Precise RoI Pooling for gfx950.  Replaces the CUDA-only extension the reference imports from the (empty)
// git submodule ltr/external/PreciseRoIPooling (import sites: ltr/models/target_classifier/initializer.py:4,18,45,
// ltr/models/bbreg/atom_iou_net.py:4,31-32,41-42,126-127,157,160).
//
// Definition (Jiang et al., ECCV 2018): with f the bilinear interpolant of the feature map
// (hat basis, zero outside the map) the output of bin (p,q) of RoI r is the exact integral of f over the bin divided
// by the bin area.  The integral separates:  sum_{j,i} F[j,i] * wy_j * wx_i  with
//   wx_i = G(xe - i) - G(xs - i),  G = CDF of the hat function,
// non-zero only for i in [floor(xs), ceil(xe)].  The gradient w.r.t. the bin edges is the boundary line integral
// (Lx, Ly below) minus the mean-value term; RoI coordinates get it through xs = X0 + q*bw etc.
//
// The op is tiny (10 RoIs x 256 ch x 25 bins per IoUNet refine step): it is latency bound, so each kernel is a single flat
// launch and a bin's <= 4 x 4 pixels are requested together (fixed window); no LDS staging is needed, they stay in L1/L2.
#include "common.h"
#include "pt_internal.h"
#include "prroi_dev.h"

__global__ void k_prroi_fwd(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                            int N, int C, int H, int W, int R, int PH, int PW, float scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)R * C * PH * PW;
    if (idx >= total) return;
    const int q = (int)(idx % PW);
    const int p = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((long)PW * PH)) % C);
    const int r = (int)(idx / ((long)PW * PH * C));
    out[idx] = prroi_fwd_elem(feat, rois, r, c, p, q, N, C, H, W, PH, PW, scale);
}

// Two pyramid levels of ONE image pooled with the same RoIs in one launch, each output multiplied by a per-channel
// factor (the IoU head's modulation, atom_iou_net.py:108-110: pool(m * f) = m * pool(f) for a per-channel m).
struct Prroi2 {
    const float* feat[2];
    const float* chan_scale[2];
    const float* gout[2];          // backward only
    float* out[2];                 // forward: pooled (R,C,PH,PH); backward: partial sums (R, slices, 4)
    int C[2], H[2], W[2], PH[2];
    float scale[2];
    const float* rois;
    int R;
};

__global__ void k_prroi_fwd2(Prroi2 a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n0 = (long)a.R * a.C[0] * a.PH[0] * a.PH[0], n1 = (long)a.R * a.C[1] * a.PH[1] * a.PH[1];
    if (idx >= n0 + n1) return;
    const int l = idx >= n0;
    const long e = l ? idx - n0 : idx;
    const int PH = a.PH[l], C = a.C[l];
    const int q = (int)(e % PH), p = (int)((e / PH) % PH), c = (int)((e / ((long)PH * PH)) % C);
    const int r = (int)(e / ((long)PH * PH * C));
    a.out[l][e] = a.chan_scale[l][c] * prroi_fwd_elem(a.feat[l], a.rois, r, c, p, q, 1, C, a.H[l], a.W[l], PH, PH, a.scale[l]);
}

// Gradient w.r.t. the features as a GATHER: one thread per feature element (b, c, j, i) walks the RoIs and bins that
// cover it in a fixed (r, p, q) order and adds its own sum to grad_features -- deterministic (the scatter form needed
// atomicAdd: run-to-run different sums).  d out[r,c,p,q] / d F[b,c,j,i] = wy_j * wx_i / area.
__global__ void k_prroi_bwd_feat(const float* __restrict__ gout, const float* __restrict__ rois,
                                 float* __restrict__ gfeat, int N, int C, int H, int W, int R, int PH, int PW,
                                 float scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * C * H * W;
    if (idx >= total) return;
    const int i = (int)(idx % W);
    const int j = (int)((idx / W) % H);
    const int c = (int)((idx / ((long)W * H)) % C);
    const int b = (int)(idx / ((long)W * H * C));
    float acc = 0.f;
    for (int r = 0; r < R; ++r) {
        const float* __restrict__ roi = rois + 5 * r;
        if ((int)roi[0] != b) continue;
        const float X0 = roi[1] * scale, Y0 = roi[2] * scale, X1 = roi[3] * scale, Y1 = roi[4] * scale;
        // the pixel's hat function reaches [i-1, i+1] x [j-1, j+1]: RoIs that do not touch it contribute nothing
        if (X1 <= (float)i - 1.f || X0 >= (float)i + 1.f || Y1 <= (float)j - 1.f || Y0 >= (float)j + 1.f) continue;
        const float bw = fmaxf(X1 - X0, 0.f) / (float)PW, bh = fmaxf(Y1 - Y0, 0.f) / (float)PH;
        const float area = bw * bh;
        if (!(area > 0.f)) continue;
        const float* __restrict__ g = gout + ((long)r * C + c) * PH * PW;
        for (int p = 0; p < PH; ++p) {
            const float ys = Y0 + (float)p * bh, ye = ys + bh;
            const float wy = hat_cdf(ye - (float)j) - hat_cdf(ys - (float)j);
            if (wy == 0.f) continue;
            for (int q = 0; q < PW; ++q) {
                const float xs = X0 + (float)q * bw, xe = xs + bw;
                const float w = wy * (hat_cdf(xe - (float)i) - hat_cdf(xs - (float)i));
                if (w != 0.f) acc += (g[p * PW + q] / area) * w;
            }
        }
    }
    if (acc != 0.f) gfeat[idx] += acc;
}

// Four coordinate sums of RoI r over elements [e0, e1) of its (c,p,q) range, block-reduced in a fixed order (every
// thread returns them).
__device__ __forceinline__ void prroi_coor_sums(const float* __restrict__ gout, const float* __restrict__ feat,
                                                const float* __restrict__ rois, int r, int e0, int e1, int N, int C, int H,
                                                int W, int PH, int PW, float scale, float* scratch, float out[4]) {
    const int per = C * PH * PW;
    float gx0 = 0.f, gy0 = 0.f, gx1 = 0.f, gy1 = 0.f;
    for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
        const int q = e % PW, p = (e / PW) % PH, c = e / (PW * PH);
        const Bin k = make_bin(rois + 5 * r, p, q, PH, PW, scale, H, W);
        if (!(k.area > 0.f) || k.b < 0 || k.b >= N) continue;
        const float* __restrict__ f = feat + ((long)k.b * C + c) * H * W;
        prroi_coor_elem(f, k, W, gout[(long)r * per + e], p, q, PH, PW, gx0, gy0, gx1, gy1);
    }
    out[0] = block_sum(gx0, scratch) * scale;
    out[1] = block_sum(gy0, scratch) * scale;
    out[2] = block_sum(gx1, scratch) * scale;
    out[3] = block_sum(gy1, scratch) * scale;
}

// one workgroup per RoI: grois (R,5), column 0 = 0
__global__ __launch_bounds__(256) void k_prroi_bwd_coor(const float* __restrict__ gout, const float* __restrict__ feat,
                                                        const float* __restrict__ rois, float* __restrict__ grois,
                                                        int N, int C, int H, int W, int R, int PH, int PW,
                                                        float scale) {
    __shared__ float scratch[16];
    const int r = blockIdx.x;
    float g[4];
    prroi_coor_sums(gout, feat, rois, r, 0, C * PH * PW, N, C, H, W, PH, PW, scale, scratch, g);
    if (threadIdx.x == 0) {
        float* o = grois + 5 * r;
        o[0] = 0.f; o[1] = g[0]; o[2] = g[1]; o[3] = g[2]; o[4] = g[3];
    }
}

// workgroup = (RoI, element slice, pyramid level): partial sums out[level][(r * slices + s) * 4 + {x0,y0,x1,y1}], added
// by the consumer in slice order (a 10-RoI call with one workgroup per RoI keeps 10 CUs busy for > 100 us)
__global__ __launch_bounds__(256) void k_prroi_bwd_coor2(Prroi2 a) {
    __shared__ float scratch[16];
    const int r = blockIdx.x, l = blockIdx.z;
    const int per = a.C[l] * a.PH[l] * a.PH[l];
    const int chunk = (per + gridDim.y - 1) / gridDim.y, e0 = blockIdx.y * chunk, e1 = min(per, e0 + chunk);
    float g[4];
    prroi_coor_sums(a.gout[l], a.feat[l], a.rois, r, e0, e1, 1, a.C[l], a.H[l], a.W[l], a.PH[l], a.PH[l], a.scale[l], scratch, g);
    if (threadIdx.x == 0) {
        float* o = a.out[l] + ((long)r * gridDim.y + blockIdx.y) * 4;
        o[0] = g[0]; o[1] = g[1]; o[2] = g[2]; o[3] = g[3];
    }
}

static int prroi_check(const void* a, const void* b, const void* c, int N, int C, int H, int W, int R, int PH, int PW) {
    if (!a || !b || !c) return PT_ERR_NULL;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || R < 0 || PH <= 0 || PW <= 0) return PT_ERR_SHAPE;
    return PT_OK;
}

extern "C" int pt_prroi_fwd_f32(const float* features, const float* rois, float* out, int N, int C, int H, int W, int R,
                                int PH, int PW, float spatial_scale, void* stream) {
    int rc = prroi_check(features, rois, out, N, C, H, W, R, PH, PW);
    if (rc || R == 0) return rc;
    const long total = (long)R * C * PH * PW;
    hipLaunchKernelGGL(k_prroi_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, features,
                       rois, out, N, C, H, W, R, PH, PW, spatial_scale);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" int pt_prroi_bwd_feat_f32(const float* grad_out, const float* rois, float* grad_features, int N, int C, int H,
                                     int W, int R, int PH, int PW, float spatial_scale, void* stream) {
    int rc = prroi_check(grad_out, rois, grad_features, N, C, H, W, R, PH, PW);
    if (rc || R == 0) return rc;
    const long total = (long)N * C * H * W;
    hipLaunchKernelGGL(k_prroi_bwd_feat, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, rois, grad_features, N, C, H, W, R, PH, PW, spatial_scale);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" int pt_prroi_bwd_coor_f32(const float* grad_out, const float* features, const float* rois, float* grad_rois,
                                     int N, int C, int H, int W, int R, int PH, int PW, float spatial_scale,
                                     void* stream) {
    int rc = prroi_check(grad_out, features, rois, N, C, H, W, R, PH, PW);
    if (rc) return rc;
    if (!grad_rois) return PT_ERR_NULL;
    if (R == 0) return PT_OK;
    hipLaunchKernelGGL(k_prroi_bwd_coor, dim3(R), dim3(256), 0, (hipStream_t)stream, grad_out, features, rois, grad_rois,
                       N, C, H, W, R, PH, PW, spatial_scale);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

int pt_launch_prroi_fwd2(const float* const feat[2], const float* const chan_scale[2], float* const out[2],
                         const int C[2], const int H[2], const int W[2], const int PH[2], const float scale[2],
                         const float* rois, int R, hipStream_t st) {
    Prroi2 a{};
    long total = 0;
    for (int l = 0; l < 2; ++l) {
        a.feat[l] = feat[l]; a.chan_scale[l] = chan_scale[l]; a.out[l] = out[l];
        a.C[l] = C[l]; a.H[l] = H[l]; a.W[l] = W[l]; a.PH[l] = PH[l]; a.scale[l] = scale[l];
        total += (long)R * C[l] * PH[l] * PH[l];
    }
    a.rois = rois; a.R = R;
    hipLaunchKernelGGL(k_prroi_fwd2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

int pt_launch_prroi_bwd_coor2(const float* const grad_out[2], const float* const feat[2], float* const part[2],
                              const int C[2], const int H[2], const int W[2], const int PH[2], const float scale[2],
                              const float* rois, int R, int slices, hipStream_t st) {
    Prroi2 a{};
    for (int l = 0; l < 2; ++l) {
        a.gout[l] = grad_out[l]; a.feat[l] = feat[l]; a.out[l] = part[l];
        a.C[l] = C[l]; a.H[l] = H[l]; a.W[l] = W[l]; a.PH[l] = PH[l]; a.scale[l] = scale[l];
    }
    a.rois = rois; a.R = R;
    hipLaunchKernelGGL(k_prroi_bwd_coor2, dim3(R, slices, 2), dim3(256), 0, st, a);
    PT_CHECK_LAUNCH();
    return PT_OK;
}
