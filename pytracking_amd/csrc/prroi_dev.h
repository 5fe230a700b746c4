// Device-side pieces of Precise RoI Pooling shared by csrc/prroi.hip (the stand-alone op) and csrc/iou_refine.hip (the fused
// IoU-refinement iteration): bin geometry, the bin integral and the five sums of its coordinate gradient.  See prroi.hip for
// the definition and the reference import sites.
#pragma once
#include "common.h"

// CDF of the hat function.  Written without branches (clamp + select): as nested ifs the compiler emitted three divergent
// branches per call, ~100 per pooled element, and the pooling kernels were bound by them.  Same values bit for bit: at the
// clamped ends the two parabolas give exactly 0 and 1.
__device__ __forceinline__ float hat_cdf(float u) {
    const float c = fminf(fmaxf(u, -1.f), 1.f);
    const float lo = 0.5f * (c + 1.f) * (c + 1.f), hi = 1.f - 0.5f * (1.f - c) * (1.f - c);
    return c <= 0.f ? lo : hi;
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
// the WIN x WIN pixel window of a bin, clamped to the pixels it touches (all loads independent)
template <int WIN, typename FP>
__device__ __forceinline__ void prroi_window_load(FP f, const Bin& k, int W, float (&v)[WIN][WIN]) {
#pragma unroll
    for (int jj = 0; jj < WIN; ++jj)
#pragma unroll
        for (int ii = 0; ii < WIN; ++ii) v[jj][ii] = f[min(k.j0 + jj, k.j1) * W + min(k.i0 + ii, k.i1)];
}

template <int WIN>
__device__ __forceinline__ float prroi_fwd_window_sum(const float (&v)[WIN][WIN], const Bin& k, int nj, int ni) {
    float wx[WIN];
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

template <int WIN>
__device__ __forceinline__ float prroi_fwd_window(const float* __restrict__ f, const Bin& k, int W, int nj, int ni) {
    float v[WIN][WIN];
    prroi_window_load<WIN, const float*>(f, k, W, v);
    return prroi_fwd_window_sum<WIN>(v, k, nj, ni);
}

// the five sums of the coordinate gradient over the same window
template <int WIN>
__device__ __forceinline__ void prroi_coor_window_sums(const float (&v)[WIN][WIN], const Bin& k, int nj, int ni, float& integ,
                                                       float& lxs, float& lxe, float& lys, float& lye) {
    float wx[WIN], hxs[WIN], hxe[WIN];
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

template <int WIN>
__device__ __forceinline__ void prroi_coor_window(const float* __restrict__ f, const Bin& k, int W, int nj, int ni, float& integ,
                                                  float& lxs, float& lxe, float& lys, float& lye) {
    float v[WIN][WIN];
    prroi_window_load<WIN, const float*>(f, k, W, v);
    prroi_coor_window_sums<WIN>(v, k, nj, ni, integ, lxs, lxe, lys, lye);
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

// the five sums of a bin -> its contributions to d/d[x0, y0, x1, y1] under the upstream gradient g
__device__ __forceinline__ void prroi_coor_finish(const Bin& k, float integ, float lxs, float lxe, float lys, float lye, float g,
                                                  int p, int q, int PH, int PW, float& gx0, float& gy0, float& gx1, float& gy1) {
    const float inv = 1.f / k.area;
    const float o = integ * inv;
    const float d_xs = (-lxs + k.bh * o) * inv, d_xe = (lxe - k.bh * o) * inv;
    const float d_ys = (-lys + k.bw * o) * inv, d_ye = (lye - k.bw * o) * inv;
    const float fq = (float)q / (float)PW, fq1 = (float)(q + 1) / (float)PW;
    const float fp = (float)p / (float)PH, fp1 = (float)(p + 1) / (float)PH;
    gx0 += g * (d_xs * (1.f - fq) + d_xe * (1.f - fq1));
    gx1 += g * (d_xs * fq + d_xe * fq1);
    gy0 += g * (d_ys * (1.f - fp) + d_ye * (1.f - fp1));
    gy1 += g * (d_ys * fp + d_ye * fp1);
}

// Coordinate gradient of ONE pooled element (bin (p,q) of channel plane f, upstream gradient g): adds its contributions to
// d/d[x0, y0, x1, y1] (unscaled: the caller multiplies the finished sums by the spatial scale).
__device__ __forceinline__ void prroi_coor_elem(const float* __restrict__ f, const Bin& k, int W, float g, int p, int q, int PH,
                                                int PW, float& gx0, float& gy0, float& gx1, float& gy1) {
    float integ = 0.f, lxs = 0.f, lxe = 0.f, lys = 0.f, lye = 0.f;
    const int nj = k.j1 - k.j0 + 1, ni = k.i1 - k.i0 + 1;
    if (nj <= 0 || ni <= 0) return;                // bin entirely outside the map: every sum is 0 (see prroi_fwd_elem)
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
    prroi_coor_finish(k, integ, lxs, lxe, lys, lye, g, p, q, PH, PW, gx0, gy0, gx1, gy1);
}
