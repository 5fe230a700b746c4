// Device helper: write one sample's slice of R, the "im2col of the residual map" that the adjoint
// pass consumes as its MFMA B operand.
//
//   R[P][tap] = rm[i][y'-u+KH/2][x'-v+KW/2]   (0 outside the OHxOW map, 0 for tap >= KH*KW)
//   with P = i*H*W + y'*W + x' the flattened (sample, position) index and tap = u*KW+v,
//
// stored in the order the 16x16x4 MFMA wants it: 16-position groups g = P>>4, inside a group
// [lane = ((P>>2)&3)*16 + tap][m = P&3], i.e. one float4 per lane per group (coalesced 1 KiB per wave).
#pragma once
#include "common.h"

__device__ __forceinline__ void pt_build_R_sample(const float* rm, float* __restrict__ R, int i, int n, int H, int W,
                                                  int KH, int KW, int OH, int OW) {
    const int HW = H * W, KK = KH * KW, ph = KH / 2, pw = KW / 2;
    if (HW & 3) {                                // ragged maps: element-wise (quads would straddle samples)
        for (int item = threadIdx.x; item < HW * 16; item += blockDim.x) {
            const int pos = item >> 4, tap = item & 15;
            const long P = (long)i * HW + pos;
            float v = 0.f;
            if (tap < KK) {
                const int u = tap / KW, vv = tap - u * KW;
                const int yp = pos / W, xp = pos - yp * W;
                const int yy = yp - u + ph, xx = xp - vv + pw;
                if ((unsigned)yy < (unsigned)OH && (unsigned)xx < (unsigned)OW) v = rm[yy * OW + xx];
            }
            R[(P >> 4) * 256 + ((int)((P >> 2) & 3) * 16 + tap) * 4 + (int)(P & 3)] = v;
        }
        if (i == n - 1) {
            const long total = (long)n * HW, padded = ((total + 15) >> 4) << 4;
            for (int item = threadIdx.x; item < (int)(padded - total) * 16; item += blockDim.x) {
                const long P = total + (item >> 4);
                R[(P >> 4) * 256 + ((int)((P >> 2) & 3) * 16 + (item & 15)) * 4 + (int)(P & 3)] = 0.f;
            }
        }
        return;
    }
    const int nq = HW >> 2;
    for (int item = threadIdx.x; item < nq * 16; item += blockDim.x) {
        const int qd = item >> 4, tap = item & 15;
        const long P = (long)i * HW + 4 * qd;
        const long g = P >> 4;
        const int lane = (int)((P >> 2) & 3) * 16 + tap;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tap < KK) {
            const int u = tap / KW, vv = tap - u * KW;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int pos = 4 * qd + m;
                const int yp = pos / W, xp = pos - yp * W;
                const int yy = yp - u + ph, xx = xp - vv + pw;
                if ((unsigned)yy < (unsigned)OH && (unsigned)xx < (unsigned)OW) v[m] = rm[yy * OW + xx];
            }
        }
        *(f32x4*)(R + g * 256 + lane * 4) = v;
    }
    if (i == n - 1) {                            // zero the unused tail of the last group
        const long total = (long)n * HW;
        const long padded = ((total + 15) >> 4) << 4;
        const int extra_q = (int)((padded - total) >> 2);
        for (int item = threadIdx.x; item < extra_q * 16; item += blockDim.x) {
            const int qd = item >> 4, tap = item & 15;
            const long P = total + 4 * qd;
            const long g = P >> 4;
            const int lane = (int)((P >> 2) & 3) * 16 + tap;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(R + g * 256 + lane * 4) = z;
        }
    }
}
