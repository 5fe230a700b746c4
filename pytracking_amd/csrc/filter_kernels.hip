// Feature passes of the filter layer (reference: ltr/models/layers/filter.py).
//
//   k_corr : apply_filter (filter.py:5-57), single filter per sequence.
//            scores[i,y,x] = sum_{c,u,v} feat[i,c,y+u-p,x+v-p] * filt[c,u,v]
//            computed as a 16 x C x (H*W) contraction on the f32 matrix cores:
//                T[tap][pos] = sum_c filt[c][tap] * feat[i][c][pos]          (MFMA 16x16x4, exact f32)
//            followed by a 16-term shift-and-add of the tap planes held in LDS.
//   k_adj  : apply_feat_transpose (filter.py:91-182), the adjoint w.r.t. the filter:
//                G[c][tap] = sum_P feat[c][P] * R[P][tap],  P = flattened (sample, position)
//            with R the im2col of the residual map (rbuild.h).
//
// Both read each feature element exactly once, 16 bytes per lane, and are HBM/MALL-bandwidth bound
// (AI ~ 7 flop/B per pass at K=4); MFMA is used so the VALU stays free and every loaded value is
// reused 16x (taps) without an LDS round trip.
#include "common.h"
#include "pt_internal.h"
#include "rbuild.h"

// ---------------------------------------------------------------------------------------------------
// corr: grid (n, KS), block = nw*64.  Wave w owns the 64-position tiles w, w+nw, ... of sample i and
// reduces over channels [cs*cper, (cs+1)*cper).  Lane l of a tile: positions 4*(l&15)..+3 (one 16-byte
// load per channel), channel sub-index l>>4 inside the 4-channel k-step.
// ---------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ void k_corr(const float* __restrict__ feat, long stride_n, const float* __restrict__ filt,
                       float* __restrict__ spart, int n, int C, int H, int W, int KH, int KW, int OH, int OW,
                       int cper) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // T[KK][HWp]
    const int i = blockIdx.x, cs = blockIdx.y;
    const int HW = H * W, KK = KH * KW;
    const int ntiles = (HW + 63) >> 6;
    const int HWp = ntiles * 64 + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int kq = lane >> 4, j = lane & 15;
    const int cbeg = cs * cper, cend = min(C, cbeg + cper);
    const float* __restrict__ fi = feat + (long)i * stride_n;
    const bool tapv = j < KK;

    for (int t = wave; t < ntiles; t += nw) {
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
        const int pos = t * 64 + 4 * j;
        const bool pv = pos < HW;
        const float* __restrict__ fp = fi + pos + (long)kq * HW;
        const float* __restrict__ ap = filt + kq * KK + j;
#pragma unroll 8
        for (int c0 = cbeg; c0 < cend; c0 += 4) {
            const bool cv = (c0 + kq) < cend;
            const float a = (tapv && cv) ? ap[c0 * KK] : 0.f;
            f32x4 b = {0, 0, 0, 0};
            if (VEC) {
                if (pv && cv) b = *(const f32x4*)(fp + (long)c0 * HW);
            } else {
                if (cv) {
                    const float* q = fp + (long)c0 * HW;
                    if (pos + 0 < HW) b[0] = q[0];
                    if (pos + 1 < HW) b[1] = q[1];
                    if (pos + 2 < HW) b[2] = q[2];
                    if (pos + 3 < HW) b[3] = q[3];
                }
            }
            acc0 = mfma16(a, b[0], acc0);
            acc1 = mfma16(a, b[1], acc1);
            acc2 = mfma16(a, b[2], acc2);
            acc3 = mfma16(a, b[3], acc3);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            if (row < KK) {
                f32x4 v = {acc0[r], acc1[r], acc2[r], acc3[r]};
                *(f32x4*)(lds + row * HWp + pos) = v;          // positions >= HW land in the tile padding
            }
        }
    }
    __syncthreads();

    const int ph = KH / 2, pw = KW / 2, OO = OH * OW;
    float* __restrict__ out = spart + ((long)cs * n + i) * OO;
    for (int o = threadIdx.x; o < OO; o += blockDim.x) {
        const int y = o / OW, x = o - y * OW;
        float s = 0.f;
        for (int u = 0; u < KH; ++u) {
            const int yy = y + u - ph;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int v = 0; v < KW; ++v) {
                const int xx = x + v - pw;
                if ((unsigned)xx < (unsigned)W) s += lds[(u * KW + v) * HWp + yy * W + xx];
            }
        }
        out[o] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// adj: grid (ceil(C/16), KSPL), block 512 (8 waves).  Wave w accumulates the 16x16 tile
// D[channel][tap] over the 16-position groups gbeg+w, gbeg+w+8, ... of its slice; lane l loads 16 bytes of
// channel cb*16+(l&15) at positions g*16 + 4*(l>>4)..+3 and one float4 of R.
// ---------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(512) void k_adj(const float* __restrict__ feat, long stride_n,
                                             const float* __restrict__ R, float* __restrict__ gpart, int n, int C,
                                             int HW, int KK, int NG, int gper) {
    __shared__ float red[8][256];
    const int cb = blockIdx.x, ks = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, j = lane & 15;
    const int c = cb * 16 + j;
    const bool cv = c < C;
    const int gbeg = ks * gper, gend = min(NG, gbeg + gper);
    const long total = (long)n * HW;
    const float inv_hw = 1.0f / (float)HW;
    const float* __restrict__ fc = feat + (long)c * HW;
    f32x4 accA = {0, 0, 0, 0}, accB = {0, 0, 0, 0};
#pragma unroll 4
    for (int g = gbeg + wave; g < gend; g += 8) {
        const long P = (long)g * 16 + 4 * kq;
        int i = (int)((float)P * inv_hw);
        if ((long)i * HW > P) --i;
        if ((long)(i + 1) * HW <= P) ++i;
        const int pos = (int)(P - (long)i * HW);
        f32x4 a = {0, 0, 0, 0};
        if (VEC) {
            if (cv && P < total) a = *(const f32x4*)(fc + (long)i * stride_n + pos);
        } else if (cv) {
            int im = i, pm = pos;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (P + m < total) a[m] = fc[(long)im * stride_n + pm];
                if (++pm == HW) { pm = 0; ++im; }
            }
        }
        const f32x4 b = *(const f32x4*)(R + (long)g * 256 + lane * 4);
        accA = mfma16(a[0], b[0], accA);
        accB = mfma16(a[1], b[1], accB);
        accA = mfma16(a[2], b[2], accA);
        accB = mfma16(a[3], b[3], accB);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * kq + r) * 16 + j] = accA[r] + accB[r];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int e = threadIdx.x, row = e >> 4, tap = e & 15;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += red[w][e];
        const int ch = cb * 16 + row;
        if (ch < C && tap < KK) gpart[(long)ks * C * KK + (long)ch * KK + tap] = s;
    }
}

__global__ void k_sum_slices(const float* __restrict__ part, float* __restrict__ out, int slices, size_t count) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    float s = 0.f;
    for (int k = 0; k < slices; ++k) s += part[(size_t)k * count + e];
    out[e] = s;
}

__global__ void k_build_R(const float* __restrict__ inp, float* __restrict__ R, int n, int H, int W, int KH, int KW,
                          int OH, int OW) {
    const int i = blockIdx.x;
    pt_build_R_sample(inp + (long)i * OH * OW, R, i, n, H, W, KH, KW, OH, OW);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
PtPlan pt_make_plan(int n, int C, int H, int W, int KH, int KW, int OH, int OW) {
    PtPlan p;
    p.n = n; p.C = C; p.H = H; p.W = W; p.KH = KH; p.KW = KW; p.OH = OH; p.OW = OW;
    p.HW = H * W; p.KK = KH * KW; p.OO = OH * OW;
    p.vec4 = (p.HW % 4) == 0;
    // corr: aim for >= ~512 workgroups, at least 16 channels (4 k-steps) per slice
    int ksteps = pt_ceil_div(C, 4);
    int want = pt_ceil_div(512, n);
    int KS = 1;
    while (KS * 2 <= want && pt_ceil_div(ksteps, KS * 2) >= 4 && KS < 32) KS *= 2;
    p.KS = KS;
    p.cper = pt_ceil_div(ksteps, KS) * 4;
    p.KS = pt_ceil_div(C, p.cper);                // drop empty slices
    int ntiles = pt_ceil_div(p.HW, 64);
    int nw = ntiles < 16 ? ntiles : 16;
    p.corr_threads = nw * 64;
    p.corr_lds = (size_t)p.KK * (ntiles * 64 + 4) * sizeof(float);
    // adj
    p.NG = (int)(((long)n * p.HW + 15) / 16);
    int KSPL = p.NG / 64;
    if (KSPL < 1) KSPL = 1;
    if (KSPL > 8) KSPL = 8;
    p.KSPL = KSPL;
    p.gper = pt_ceil_div(p.NG, KSPL);
    p.KSPL = pt_ceil_div(p.NG, p.gper);
    return p;
}

int pt_launch_corr(const PtPlan& p, const float* feat, long stride_n, const float* filt, float* spart, hipStream_t st) {
    dim3 grid(p.n, p.KS), block(p.corr_threads);
    pt_prof_begin(0, st);
    if (p.vec4 && (stride_n % 4) == 0 && ((uintptr_t)feat % 16) == 0)
        hipLaunchKernelGGL(k_corr<true>, grid, block, p.corr_lds, st, feat, stride_n, filt, spart, p.n, p.C, p.H, p.W,
                           p.KH, p.KW, p.OH, p.OW, p.cper);
    else
        hipLaunchKernelGGL(k_corr<false>, grid, block, p.corr_lds, st, feat, stride_n, filt, spart, p.n, p.C, p.H,
                           p.W, p.KH, p.KW, p.OH, p.OW, p.cper);
    pt_prof_end(0, st);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

int pt_launch_adj(const PtPlan& p, const float* feat, long stride_n, const float* R, float* gpart, hipStream_t st) {
    dim3 grid(pt_ceil_div(p.C, 16), p.KSPL), block(512);
    pt_prof_begin(1, st);
    if (p.vec4 && (stride_n % 4) == 0 && ((uintptr_t)feat % 16) == 0)
        hipLaunchKernelGGL(k_adj<true>, grid, block, 0, st, feat, stride_n, R, gpart, p.n, p.C, p.HW, p.KK, p.NG, p.gper);
    else
        hipLaunchKernelGGL(k_adj<false>, grid, block, 0, st, feat, stride_n, R, gpart, p.n, p.C, p.HW, p.KK, p.NG, p.gper);
    pt_prof_end(1, st);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

int pt_launch_sum_slices(const float* part, float* out, int slices, size_t count, hipStream_t st) {
    const int threads = 256;
    hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((count + threads - 1) / threads)), dim3(threads), 0, st, part,
                       out, slices, count);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

int pt_launch_build_R(const PtPlan& p, const float* inp, float* R, hipStream_t st) {
    hipLaunchKernelGGL(k_build_R, dim3(p.n), dim3(256), 0, st, inp, R, p.n, p.H, p.W, p.KH, p.KW, p.OH, p.OW);
    PT_CHECK_LAUNCH();
    return PT_OK;
}
