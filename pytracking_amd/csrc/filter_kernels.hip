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
// reused 16x (taps) without an LDS round trip.  All loads of a wave's slice are issued before the first
// MFMA (>= 16 KiB in flight per wave) because at ~1 workgroup per CU there is no other latency hiding.
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "pt_internal.h"
#include "rbuild.h"

// ---------------------------------------------------------------------------------------------------
// corr: grid (n, KS), block = nw*64.  Wave w owns the 64-position tiles w, w+nw, ... of sample i and
// reduces over channels [cs*cper, (cs+1)*cper).  Lane l of a tile: positions 4*(l&15)..+3 (one 16-byte
// load per channel), channel sub-index l>>4 inside the 4-channel k-step.
//
// Optional fused stages (PtCorrFuse):
//   * gpart != null : the filter operand is g = sum_k gpart[k] + reg*w (the steepest-descent gradient,
//     optimizer.py:146-148) reduced here instead of in a separate launch; the workgroups of sample 0 also
//     publish g and their slice of |g|^2.
//   * copy_dst != null : every feature vector loaded is also stored to copy_dst (the tracker's memory
//     insert, dimp.py:429-441, rides on the classification pass).
// ---------------------------------------------------------------------------------------------------
template <bool VEC>
__device__ __forceinline__ f32x4 corr_load(const float* __restrict__ q, bool ok, int pos, int HW) {
    f32x4 v = {0, 0, 0, 0};
    if (VEC) {
        if (ok) v = *(const f32x4*)q;
    } else if (ok) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (pos + m < HW) v[m] = q[m];
    }
    return v;
}

template <bool VEC>
__device__ __forceinline__ void corr_store(float* __restrict__ q, bool ok, int pos, int HW, f32x4 v) {
    if (!ok) return;
    if (VEC) {
        *(f32x4*)q = v;
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (pos + m < HW) q[m] = v[m];
    }
}

// filter element e of the zero-padded [cper][16] LDS image: rows past the slice and taps >= KK are zero, so
// the MFMA A operand needs no predicate.
__device__ __forceinline__ float corr_filter_elem(const float* __restrict__ filt, const PtCorrFuse& fz, int C, int KK,
                                                  int cbeg, int cend, int e, bool publish, float& gsq) {
    const int cl = e >> 4, tp = e & 15;
    const bool ok = (cbeg + cl) < cend && tp < KK;
    const long ge = ok ? (long)(cbeg + cl) * KK + tp : 0;
    float v;
    if (fz.gpart) {
        float part[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) part[k] = fz.gpart[(long)min(k, fz.KSPL - 1) * C * KK + ge];   // all in flight
        const float wv = fz.w[ge];
        v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += k < fz.KSPL ? part[k] : 0.f;                              // fixed order
        for (int k = 8; k < fz.KSPL; ++k) v += fz.gpart[(long)k * C * KK + ge];
        v += fz.reg * wv;
        if (ok) {
            gsq += v * v;
            if (publish) fz.g_out[ge] = v;
        }
    } else {
        v = filt[ge];
    }
    return ok ? v : 0.f;
}


template <bool VEC, int NK>
__global__ void k_corr(const float* __restrict__ feat, long stride_n, const float* __restrict__ filt,
                       float* __restrict__ spart, int n, int C, int H, int W, int KH, int KW, int OH, int OW,
                       int cper, PtCorrFuse fz) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // afilt[cper][16] | T[KK][HWp]
    __shared__ float scratch[16];
    constexpr int NKA = NK > 0 ? NK : 1;
    if (blockIdx.z) { feat = fz.feat2; filt = fz.filt2; spart = fz.spart2; }   // second problem of a paired launch (uniform)
    const int i = blockIdx.x, cs = blockIdx.y;
    const int HW = H * W, KK = KH * KW;
    const int ntiles = (HW + 63) >> 6;
    const int HWp = ntiles * 64 + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int kq = lane >> 4, j = lane & 15;
    const int cbeg = cs * cper, cend = min(C, cbeg + cper);
    const int nsl = cper * 16;
    float* __restrict__ afilt = lds;
    float* __restrict__ Tl = lds + nsl;
    const float* __restrict__ fi = feat + (long)i * stride_n;
    const bool publish = fz.gpart && i == 0;

    // ---- filter slice -> registers (tiny; issued before the feature loads so that waiting for it leaves them in flight)
    float fv[4] = {0.f, 0.f, 0.f, 0.f};
    float gsq = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + q * blockDim.x;
        if (e < nsl) fv[q] = corr_filter_elem(filt, fz, C, KK, cbeg, cend, e, publish, gsq);
    }

    // ---- first tile's feature loads: the whole channel slice of this wave in flight.  VEC path: addresses are
    //      clamped instead of predicated (straight-line code -> counted vmcnt waits); clamped channels meet a zero
    //      filter row, clamped positions land in output columns nobody reads.
    f32x4 b[NKA];
    int t = wave;
    if (NK > 0 && t < ntiles) {
        const int pos = t * 64 + 4 * j;
        if (VEC) {
            const float* __restrict__ fp = fi + (pos < HW ? pos : 0);
#pragma unroll
            for (int k = 0; k < NK; ++k) b[k] = *(const f32x4*)(fp + (long)min(cbeg + 4 * k + kq, cend - 1) * HW);
        } else {
            const float* __restrict__ fp = fi + pos + (long)kq * HW;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int c0 = cbeg + 4 * k;
                b[k] = corr_load<false>(fp + (long)c0 * HW, pos < HW && (c0 + kq) < cend, pos, HW);
            }
        }
    }

    // ---- publish the filter slice
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + q * blockDim.x;
        if (e < nsl) afilt[e] = fv[q];
    }
    for (int e = threadIdx.x + 4 * blockDim.x; e < nsl; e += blockDim.x)        // slices larger than 4*blockDim
        afilt[e] = corr_filter_elem(filt, fz, C, KK, cbeg, cend, e, publish, gsq);
    if (publish) {                                                  // uniform per workgroup
        const float tot = block_sum(gsq, scratch);
        if (threadIdx.x == 0) fz.anum_part[cs] = tot;
    }
    __syncthreads();                                                // plain loads stay in flight across the barrier

    for (; t < ntiles; t += nw) {
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
        const int pos = t * 64 + 4 * j;
        const bool pv = pos < HW;
        if (NK > 0) {
            if (t != wave) {
                if (VEC) {
                    const float* __restrict__ fp = fi + (pv ? pos : 0);
#pragma unroll
                    for (int k = 0; k < NK; ++k) b[k] = *(const f32x4*)(fp + (long)min(cbeg + 4 * k + kq, cend - 1) * HW);
                } else {
                    const float* __restrict__ fp = fi + pos + (long)kq * HW;
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const int c0 = cbeg + 4 * k;
                        b[k] = corr_load<false>(fp + (long)c0 * HW, pv && (c0 + kq) < cend, pos, HW);
                    }
                }
            }
            if (fz.copy_dst) {
                float* __restrict__ dp = fz.copy_dst + pos + (long)kq * HW;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int c0 = cbeg + 4 * k;
                    corr_store<VEC>(dp + (long)c0 * HW, pv && (c0 + kq) < cend, pos, HW, b[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const float a = afilt[(4 * k + kq) * 16 + j];
                acc0 = mfma16(a, b[k][0], acc0);
                acc1 = mfma16(a, b[k][1], acc1);
                acc2 = mfma16(a, b[k][2], acc2);
                acc3 = mfma16(a, b[k][3], acc3);
            }
        } else {
            const float* __restrict__ fp = fi + pos + (long)kq * HW;
            for (int c0 = cbeg; c0 < cend; c0 += 4) {
                const bool cv = (c0 + kq) < cend;
                const float a = afilt[(c0 - cbeg + kq) * 16 + j];
                const f32x4 v = corr_load<VEC>(fp + (long)c0 * HW, pv && cv, pos, HW);
                if (fz.copy_dst) corr_store<VEC>(fz.copy_dst + pos + (long)(c0 + kq) * HW, pv && cv, pos, HW, v);
                acc0 = mfma16(a, v[0], acc0);
                acc1 = mfma16(a, v[1], acc1);
                acc2 = mfma16(a, v[2], acc2);
                acc3 = mfma16(a, v[3], acc3);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            if (row < KK) {
                f32x4 v = {acc0[r], acc1[r], acc2[r], acc3[r]};
                *(f32x4*)(Tl + row * HWp + pos) = v;           // positions >= HW land in the tile padding
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
                if ((unsigned)xx < (unsigned)W) s += Tl[(u * KW + v) * HWp + yy * W + xx];
            }
        }
        out[o] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// adj: grid (ceil(C/16), KSPL), block 512 (8 waves).  Wave w accumulates the 16x16 tile
// D[channel][tap] over the 16-position groups gbeg+w, gbeg+w+8, ... of its slice; lane l loads 16 bytes of
// channel cb*16+(l&15) at positions g*16 + 4*(l>>4)..+3 and one float4 of R.  Loads are issued U groups
// (2*U KiB per wave) ahead of the MFMAs that consume them.
// ---------------------------------------------------------------------------------------------------
// conv_same input gradient, jobs = (sample, group of 4 filter-bank rows): every thread one position, 4 outputs from the same 16
// reads of the map; summation over (u, v) in the order of the stand-alone kernel it replaces (bit-identical)
__device__ __forceinline__ void input_grad_jobs(const PtInputGrad& g, int job, int njobs) {
    constexpr int KG = 4;
    const int HW = g.H * g.W, KK = g.K * g.K, p = g.K / 2;
    const int groups = (g.Kc + KG - 1) / KG;
    for (int jb = job; jb < g.n * groups; jb += njobs) {
        const int i = jb / groups, k0 = (jb - i * groups) * KG;
        const float* __restrict__ vm = g.v + (long)i * HW;
        const float* __restrict__ fk = g.filt + (long)k0 * KK;
        for (int pos = threadIdx.x; pos < HW; pos += blockDim.x) {
            const int yy = pos / g.W, xx = pos - yy * g.W;
            float s[KG] = {0.f, 0.f, 0.f, 0.f};
            for (int u = 0; u < g.K; ++u) {
                const int y = yy - u + p;
                if ((unsigned)y >= (unsigned)g.H) continue;
                for (int w = 0; w < g.K; ++w) {
                    const int x = xx - w + p;
                    if ((unsigned)x < (unsigned)g.W) {
                        const float vv = vm[y * g.W + x];
#pragma unroll
                        for (int c = 0; c < KG; ++c) s[c] += vv * fk[(k0 + c < g.Kc ? c : 0) * KK + u * g.K + w];
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < KG; ++c)
                if (k0 + c < g.Kc) g.out[((long)i * g.Kc + k0 + c) * HW + pos] = s[c];
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(512) void k_adj(const float* __restrict__ feat, long stride_n,
                                             const float* __restrict__ R, float* __restrict__ gpart, int n, int C,
                                             int HW, int KK, int NG, int gper, PtInputGrad ig) {
    constexpr int U = 16;
    __shared__ float red[8][256];
    if (blockIdx.z) {                                               // uniform: this workgroup serves the input gradient
        const int per = gridDim.x * gridDim.y;
        input_grad_jobs(ig, (blockIdx.z - 1) * per + blockIdx.y * gridDim.x + blockIdx.x, (gridDim.z - 1) * per);
        return;
    }
    const int cb = blockIdx.x, ks = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, j = lane & 15;
    const int c = cb * 16 + j;
    const bool cv = c < C;
    const int gbeg = ks * gper, gend = min(NG, gbeg + gper);
    const long total = (long)n * HW;
    const float inv_hw = 1.0f / (float)HW;
    const float* __restrict__ fc = feat + (long)min(c, C - 1) * HW;
    f32x4 accA = {0, 0, 0, 0}, accB = {0, 0, 0, 0};
    for (int g0 = gbeg + wave; g0 < gend; g0 += 8 * U) {
        f32x4 a[U], b[U];
        if (VEC) {
            // straight-line: out-of-range groups / positions re-read a valid address and are zeroed afterwards
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int g = min(g0 + 8 * u, gend - 1);
                long P = (long)g * 16 + 4 * kq;
                P = P < total ? P : 0;
                int i = (int)((float)P * inv_hw);
                i -= ((long)i * HW > P) ? 1 : 0;
                i += ((long)(i + 1) * HW <= P) ? 1 : 0;
                const int pos = (int)(P - (long)i * HW);
                a[u] = *(const f32x4*)(fc + (long)i * stride_n + pos);
                b[u] = *(const f32x4*)(R + (long)g * 256 + lane * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int g = g0 + 8 * u;
                // k-validity (group / position in range) zeroes BOTH operands; channel validity only A
                // (lane l carries channel l&15 in A but tap l&15 in B)
                const bool okk = g < gend && ((long)g * 16 + 4 * kq) < total;
                const bool oka = okk && cv;
                accA = mfma16(oka ? a[u][0] : 0.f, okk ? b[u][0] : 0.f, accA);
                accB = mfma16(oka ? a[u][1] : 0.f, okk ? b[u][1] : 0.f, accB);
                accA = mfma16(oka ? a[u][2] : 0.f, okk ? b[u][2] : 0.f, accA);
                accB = mfma16(oka ? a[u][3] : 0.f, okk ? b[u][3] : 0.f, accB);
            }
        } else {
#pragma unroll 2
            for (int u = 0; u < U; ++u) {
                const int g = g0 + 8 * u;
                f32x4 av = {0, 0, 0, 0}, bv = {0, 0, 0, 0};
                if (g < gend) {
                    const long P = (long)g * 16 + 4 * kq;
                    int i = (int)((float)P * inv_hw);
                    if ((long)i * HW > P) --i;
                    if ((long)(i + 1) * HW <= P) ++i;
                    int im = i, pm = (int)(P - (long)i * HW);
                    if (cv) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            if (P + m < total) av[m] = fc[(long)im * stride_n + pm];
                            if (++pm == HW) { pm = 0; ++im; }
                        }
                    }
                    bv = *(const f32x4*)(R + (long)g * 256 + lane * 4);
                }
                accA = mfma16(av[0], bv[0], accA);
                accB = mfma16(av[1], bv[1], accB);
                accA = mfma16(av[2], bv[2], accA);
                accB = mfma16(av[3], bv[3], accB);
            }
        }
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
#pragma unroll 8
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
    // corr: aim for >= ~512 workgroups with 4..16 k-steps (16..64 channels) per channel slice
    int ksteps = pt_ceil_div(C, 4);
    int want = pt_ceil_div(512, n);
    int KS = 1;
    while ((KS * 2 <= want || pt_ceil_div(ksteps, KS) > 16) && pt_ceil_div(ksteps, KS * 2) >= 4 && KS < 64) KS *= 2;
    p.cper = pt_ceil_div(ksteps, KS) * 4;
    p.KS = pt_ceil_div(C, p.cper);                // drop empty slices
    int ntiles = pt_ceil_div(p.HW, 64);
    int nw = ntiles < 16 ? ntiles : 16;
    p.corr_threads = nw * 64;
    p.corr_lds = ((size_t)p.cper * 16 + (size_t)p.KK * (ntiles * 64 + 4)) * sizeof(float);
    // adj
    p.NG = (int)(((long)n * p.HW + 15) / 16);
    // position slices: one pass of the 8 waves covers 128 groups; with few channel blocks (ATOM: C = 64) more slices are
    // needed to put a workgroup on every CU
    int KSPL = p.NG / 64;
    const int cap = std::max(8, 256 / pt_ceil_div(C, 16));
    if (KSPL > cap) KSPL = cap;
    if (KSPL > pt_ceil_div(p.NG, 128)) KSPL = std::max(pt_ceil_div(p.NG, 128), std::min(KSPL, 8));
    if (KSPL < 1) KSPL = 1;
    if (KSPL > 64) KSPL = 64;
    p.gper = pt_ceil_div(p.NG, KSPL);
    p.KSPL = pt_ceil_div(p.NG, p.gper);
    return p;
}

template <bool VEC>
static void corr_dispatch(const PtPlan& p, dim3 grid, dim3 block, hipStream_t st, const float* feat, long stride_n,
                          const float* filt, float* spart, const PtCorrFuse& fz) {
    const int nk = p.cper / 4;
#define PT_CORR_CASE(NKV)                                                                                          \
    hipLaunchKernelGGL((k_corr<VEC, NKV>), grid, block, p.corr_lds, st, feat, stride_n, filt, spart, p.n, p.C, p.H, \
                       p.W, p.KH, p.KW, p.OH, p.OW, p.cper, fz)
    if (nk == 16) PT_CORR_CASE(16);
    else if (nk == 8) PT_CORR_CASE(8);
    else if (nk == 4) PT_CORR_CASE(4);
    else PT_CORR_CASE(0);
#undef PT_CORR_CASE
}

int pt_launch_corr(const PtPlan& p, const float* feat, long stride_n, const float* filt, float* spart, hipStream_t st,
                   const PtCorrFuse* fuse) {
    dim3 grid(p.n, p.KS), block(p.corr_threads);
    PtCorrFuse fz = {nullptr, 0, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (fuse) fz = *fuse;
    if (fz.feat2) {                                                 // paired launch: same shape, second operand set
        if (!fz.filt2 || !fz.spart2 || fz.gpart || fz.copy_dst) return PT_ERR_UNSUPPORTED;
        if (((uintptr_t)fz.feat2 % 16) != ((uintptr_t)feat % 16)) return PT_ERR_UNSUPPORTED;
        grid.z = 2;
    }
    if (fz.copy_dst && p.n != 1) return PT_ERR_SHAPE;
    if (p.vec4 && (stride_n % 4) == 0 && ((uintptr_t)feat % 16) == 0 && ((uintptr_t)fz.copy_dst % 16) == 0)
        corr_dispatch<true>(p, grid, block, st, feat, stride_n, filt, spart, fz);
    else
        corr_dispatch<false>(p, grid, block, st, feat, stride_n, filt, spart, fz);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

int pt_launch_adj(const PtPlan& p, const float* feat, long stride_n, const float* R, float* gpart, hipStream_t st,
                  const PtInputGrad* ig) {
    dim3 grid(pt_ceil_div(p.C, 16), p.KSPL), block(512);
    PtInputGrad g = {nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
    if (ig) {
        // enough extra layers that the (sample, 4-row group) jobs spread over the chip (~2 workgroups per CU)
        g = *ig;
        const int jobs = g.n * ((g.Kc + 3) / 4), per = (int)(grid.x * grid.y);
        grid.z = 1 + std::max(1, std::min(pt_ceil_div(std::min(jobs, 512), per), 64));
    }
    if (p.vec4 && (stride_n % 4) == 0 && ((uintptr_t)feat % 16) == 0)
        hipLaunchKernelGGL(k_adj<true>, grid, block, 0, st, feat, stride_n, R, gpart, p.n, p.C, p.HW, p.KK, p.NG, p.gper, g);
    else
        hipLaunchKernelGGL(k_adj<false>, grid, block, 0, st, feat, stride_n, R, gpart, p.n, p.C, p.HW, p.KK, p.NG, p.gper, g);
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
