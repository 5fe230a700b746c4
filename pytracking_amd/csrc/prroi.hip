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

__device__ __forceinline__ float hat_cdf(float u) {
    if (u <= -1.f) return 0.f;
    if (u <= 0.f) return 0.5f * (u + 1.f) * (u + 1.f);
    if (u <= 1.f) return 1.f - 0.5f * (1.f - u) * (1.f - u);
    return 1.f;
}
__device__ __forceinline__ float hat(float u) { return fmaxf(0.f, 1.f - fabsf(u)); }

struct Bin {
    float xs, xe, ys, ye, bw, bh, area;
    int b, i0, i1, j0, j1;
};

__device__ __forceinline__ Bin make_bin(const float* __restrict__ roi, int p, int q, int PH, int PW, float scale, int H,
                                        int W) {
    Bin k;
    k.b = (int)roi[0];
    const float X0 = roi[1] * scale, Y0 = roi[2] * scale, X1 = roi[3] * scale, Y1 = roi[4] * scale;
    k.bw = fmaxf(X1 - X0, 0.f) / (float)PW;
    k.bh = fmaxf(Y1 - Y0, 0.f) / (float)PH;
    k.xs = X0 + (float)q * k.bw;
    k.xe = k.xs + k.bw;
    k.ys = Y0 + (float)p * k.bh;
    k.ye = k.ys + k.bh;
    k.area = k.bw * k.bh;
    k.i0 = max(0, (int)floorf(k.xs));
    k.i1 = min(W - 1, (int)ceilf(k.xe));
    k.j0 = max(0, (int)floorf(k.ys));
    k.j1 = min(H - 1, (int)ceilf(k.ye));
    return k;
}

// Integral of one bin over a fixed WIN x WIN pixel window (the bin touches nj x ni <= WIN x WIN pixels): all loads are
// issued before the first wait -- the runtime-bounded loops of the general form make every pixel its own dependent round
// trip (k_prroi_bwd_coor2 18 -> 11 us, k_prroi_fwd2 9.3 -> 8.2 us at 2-3 pixel bins).  Same terms in the same order.
template <int WIN>
__device__ __forceinline__ float prroi_fwd_window(const float* __restrict__ f, const Bin& k, int W, int nj, int ni) {
    float v[WIN][WIN], wx[WIN];
#pragma unroll
    for (int jj = 0; jj < WIN; ++jj)
#pragma unroll
        for (int ii = 0; ii < WIN; ++ii) v[jj][ii] = f[min(k.j0 + jj, k.j1) * W + min(k.i0 + ii, k.i1)];
#pragma unroll
    for (int ii = 0; ii < WIN; ++ii) {
        const float i = (float)(k.i0 + ii);
        wx[ii] = ii < ni ? hat_cdf(k.xe - i) - hat_cdf(k.xs - i) : 0.f;
    }
    float acc = 0.f;
#pragma unroll
    for (int jj = 0; jj < WIN; ++jj) {
        const float j = (float)(k.j0 + jj);
        const float wy = jj < nj ? hat_cdf(k.ye - j) - hat_cdf(k.ys - j) : 0.f;
        float row = 0.f;
#pragma unroll
        for (int ii = 0; ii < WIN; ++ii) row += v[jj][ii] * wx[ii];
        acc += wy * row;
    }
    return acc;
}

// the five sums of the coordinate gradient over the same window
template <int WIN>
__device__ __forceinline__ void prroi_coor_window(const float* __restrict__ f, const Bin& k, int W, int nj, int ni, float& integ,
                                                  float& lxs, float& lxe, float& lys, float& lye) {
    float v[WIN][WIN], wx[WIN], hxs[WIN], hxe[WIN];
#pragma unroll
    for (int jj = 0; jj < WIN; ++jj)
#pragma unroll
        for (int ii = 0; ii < WIN; ++ii) v[jj][ii] = f[min(k.j0 + jj, k.j1) * W + min(k.i0 + ii, k.i1)];
#pragma unroll
    for (int ii = 0; ii < WIN; ++ii) {
        const float i = (float)(k.i0 + ii);
        const bool in = ii < ni;
        wx[ii] = in ? hat_cdf(k.xe - i) - hat_cdf(k.xs - i) : 0.f;
        hxs[ii] = in ? hat(k.xs - i) : 0.f;
        hxe[ii] = in ? hat(k.xe - i) : 0.f;
    }
#pragma unroll
    for (int jj = 0; jj < WIN; ++jj) {
        const float j = (float)(k.j0 + jj);
        const bool in = jj < nj;
        const float wy = in ? hat_cdf(k.ye - j) - hat_cdf(k.ys - j) : 0.f;
        const float hys = in ? hat(k.ys - j) : 0.f, hye = in ? hat(k.ye - j) : 0.f;
        float row = 0.f, rxs = 0.f, rxe = 0.f;
#pragma unroll
        for (int ii = 0; ii < WIN; ++ii) {
            row += v[jj][ii] * wx[ii];
            rxs += v[jj][ii] * hxs[ii];
            rxe += v[jj][ii] * hxe[ii];
        }
        integ += wy * row;
        lxs += wy * rxs;
        lxe += wy * rxe;
        lys += hys * row;
        lye += hye * row;
    }
}

// one output element: bin (p,q) of channel c of RoI r
__device__ __forceinline__ float prroi_fwd_elem(const float* __restrict__ feat, const float* __restrict__ rois, int r, int c,
                                                int p, int q, int N, int C, int H, int W, int PH, int PW, float scale) {
    const Bin k = make_bin(rois + 5 * r, p, q, PH, PW, scale, H, W);
    float acc = 0.f;
    if (k.area > 0.f && k.b >= 0 && k.b < N) {
        const float* __restrict__ f = feat + ((long)k.b * C + c) * H * W;
        const int nj = k.j1 - k.j0 + 1, ni = k.i1 - k.i0 + 1;
        // a bin more than one pixel above / left of (or below / right of) the map touches no pixel: j1 < j0 or i1 < i0,
        // and j1 / i1 may be negative -- the window paths clamp to them, so they must not run (the integral is 0)
        if (nj <= 0 || ni <= 0) acc = 0.f;
        else if (nj <= 4 && ni <= 4) acc = prroi_fwd_window<4>(f, k, W, nj, ni);
        else if (nj <= 6 && ni <= 6) acc = prroi_fwd_window<6>(f, k, W, nj, ni);
        else {
            for (int j = k.j0; j <= k.j1; ++j) {
                const float wy = hat_cdf(k.ye - (float)j) - hat_cdf(k.ys - (float)j);
                float row = 0.f;
                for (int i = k.i0; i <= k.i1; ++i)
                    row += f[j * W + i] * (hat_cdf(k.xe - (float)i) - hat_cdf(k.xs - (float)i));
                acc += wy * row;
            }
        }
        acc /= k.area;
    }
    return acc;
}

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
        float integ = 0.f, lxs = 0.f, lxe = 0.f, lys = 0.f, lye = 0.f;
        const int nj = k.j1 - k.j0 + 1, ni = k.i1 - k.i0 + 1;
        if (nj <= 0 || ni <= 0) continue;          // bin entirely outside the map: every sum is 0 (see prroi_fwd_elem)
        if (nj <= 4 && ni <= 4) prroi_coor_window<4>(f, k, W, nj, ni, integ, lxs, lxe, lys, lye);
        else if (nj <= 6 && ni <= 6) prroi_coor_window<6>(f, k, W, nj, ni, integ, lxs, lxe, lys, lye);
        else {
            for (int j = k.j0; j <= k.j1; ++j) {
                const float wy = hat_cdf(k.ye - (float)j) - hat_cdf(k.ys - (float)j);
                const float hys = hat(k.ys - (float)j), hye = hat(k.ye - (float)j);
                float row = 0.f, rxs = 0.f, rxe = 0.f;
                for (int i = k.i0; i <= k.i1; ++i) {
                    const float v = f[j * W + i];
                    row += v * (hat_cdf(k.xe - (float)i) - hat_cdf(k.xs - (float)i));
                    rxs += v * hat(k.xs - (float)i);
                    rxe += v * hat(k.xe - (float)i);
                }
                integ += wy * row;
                lxs += wy * rxs;
                lxe += wy * rxe;
                lys += hys * row;
                lye += hye * row;
            }
        }
        const float inv = 1.f / k.area;
        const float o = integ * inv;
        const float d_xs = (-lxs + k.bh * o) * inv, d_xe = (lxe - k.bh * o) * inv;
        const float d_ys = (-lys + k.bw * o) * inv, d_ye = (lye - k.bw * o) * inv;
        const float g = gout[(long)r * per + e];
        const float fq = (float)q / (float)PW, fq1 = (float)(q + 1) / (float)PW;
        const float fp = (float)p / (float)PH, fp1 = (float)(p + 1) / (float)PH;
        gx0 += g * (d_xs * (1.f - fq) + d_xe * (1.f - fq1));
        gx1 += g * (d_xs * fq + d_xe * fq1);
        gy0 += g * (d_ys * (1.f - fp) + d_ye * (1.f - fp1));
        gy1 += g * (d_ys * fp + d_ye * fp1);
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
