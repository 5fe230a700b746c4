// Multi-filter feature passes (reference: ltr/models/layers/filter.py, the 5-D `filter` / `input` branches used by the
// LWL few-shot learner, ltr/models/lwl/linear_filter.py:70-73, loss_residual_modules.py:16-41):
//
//   k_mf_corr : apply_filter(feat (n,C,H,W), filter (F,C,K,K)) -> (n,F,H,W)          filter.py:29-34
//               scores[i,f,y,x] = sum_{c,u,v} filt[f,c,u,v] * feat[i,c,y+u-p,x+v-p]   (K odd, p = K/2, zero padded)
//   k_mf_adj  : apply_feat_transpose(feat, input (n,F,H,W), K) -> (F,C,K,K)           filter.py:158-176
//               grad[f,c,u,v]   = sum_{i,y,x} feat[i,c,y+u-p,x+v-p] * input[i,f,y,x]
//
// With F <= 16 filters both are dense contractions with a 16-wide filter dimension -- M = 16 on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32, exact fp32), the MFMA-bound rows of SURVEY.md section 8 (LWL: 7.4 GFLOP per pass at n = 32,
// 47 us at the 157 TFLOP/s fp32 matrix peak vs 13 us of HBM time).  Feature tiles are staged in LDS with their zero
// padding (row band + halo), so the K*K shifted operands are LDS reads at a constant offset instead of global re-reads.
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "pt_internal.h"

struct MfGeom {
    int n, F, C, H, W, K, p, KK;
    int BR, NB;          // output rows per band, bands per sample
    int PWs;             // padded row stride in LDS (W + K - 1)
    int RSmax;           // staged rows per band (BR + K - 1)
};


__device__ __forceinline__ int mf_fdiv(int v, float inv_d) { return (int)(((float)v + 0.5f) * inv_d); }

// Staging plan of one lane for a (rows x W) block that is contiguous in global memory (full-width rows): item q covers
// VW consecutive floats starting at e = VW*(lane + 64*q).  VW = 4 needs W % 4 == 0 (an item never straddles a row).
template <int VW, int NQ>
struct MfStage {
    int g[NQ];           // float offset inside the channel / filter plane, or -1
    int l[NQ];           // LDS float offset of the first element
    __device__ __forceinline__ void plan(int lane, int first_row, int nrows, int W, float inv_w, int lds_row0, int lds_stride,
                                         int lds_col0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = VW * (lane + 64 * q);
            const int rr = mf_fdiv(e, inv_w), x = e - rr * W;
            const bool ok = e < nrows * W;
            g[q] = ok ? (first_row + rr) * W + x : -1;
            l[q] = (lds_row0 + rr) * lds_stride + x + lds_col0;
        }
    }
};

// ---------------------------------------------------------------------------------------------------
// correlation: grid (NB, n), 256 threads.  Channels are streamed in chunks of MF_CK (MF_KS MFMA k-steps); wave w stages
// channels w, w+4, ... of the chunk (and a quarter of the weights) while the previous chunk is being multiplied.
// LDS: fl[MF_CK][CS] zero-padded feature band (channel stride CS == 16 mod 32: conflict-free ds_read_b32 of the B operand),
//      wl[MF_KS][KK][64] weights in MFMA-A order (lane = kq*16 + f).
// Wave w owns the 16-position tiles w, w+4, ... of the band (NT per wave).  The band height is chosen so that the grid
// holds >= 2 workgroups per CU: a workgroup puts one wave on each SIMD, the second one hides its LDS / barrier latency.
// ---------------------------------------------------------------------------------------------------
#define MF_CK 32
#define MF_KS (MF_CK / 4)
#define MF_CW (MF_CK / 4)  // channels staged per wave
#define MF_NQ 8            // scalar staging items per lane per plane: rows*W <= 512

template <int KK, int NT, int VW>
__global__ __launch_bounds__(256) void k_mf_corr(const float* __restrict__ feat, long stride_n,
                                                 const float* __restrict__ wT, float* __restrict__ scores,
                                                 long out_stride_n, MfGeom g, int CS, long wt_zstride, long out_zstride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    wT += (long)blockIdx.z * wt_zstride;                            // blockIdx.z: group of <= 16 filters of a wider bank
    scores += (long)blockIdx.z * out_zstride;
    constexpr int NQ = MF_NQ / VW;
    constexpr int K = KK == 1 ? 1 : 3;
    float* __restrict__ fl = lds;                                   // [MF_CK][CS]
    float* __restrict__ wl = lds + MF_CK * CS;                      // [MF_KS][KK][64]
    const int band = blockIdx.x, i = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, j = lane & 15;
    const int y0 = band * g.BR, rows = min(g.BR, g.H - y0);
    const int HW = g.H * g.W, npos = rows * g.W;
    const float inv_w = 1.0f / (float)g.W;
    const float* __restrict__ fi = feat + (long)i * stride_n;

    for (int e = threadIdx.x; e < MF_CK * CS; e += 256) fl[e] = 0.f;          // padding (and rows outside the image) stay zero

    // image rows [ys, ye) of the band + halo are contiguous in every channel plane
    const int ys = max(y0 - g.p, 0), ye = min(y0 + rows + g.p, g.H);
    MfStage<VW, NQ> sp;
    sp.plan(lane, ys, ye - ys, g.W, inv_w, ys - (y0 - g.p), g.PWs, g.p);
    // weights: the chunk's [MF_KS][KK][64] block is contiguous in the pre-transposed table wT (k_mf_wtrans): 16-byte loads
    constexpr int WN = (MF_KS * KK * 16 + 255) / 256;               // float4 items per thread

    // ---- tile geometry of this wave
    int t_off[NT];
    bool t_ok[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int pj = 16 * (wave + 4 * q) + j;
        const int r = mf_fdiv(pj, inv_w), x = pj - r * g.W;
        t_ok[q] = pj < npos;
        t_off[q] = t_ok[q] ? r * g.PWs + x : 0;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = (f32x4){0, 0, 0, 0};

    float sv[MF_CW][NQ][VW];
    f32x4 wv[WN];
    // fetch: raw loads only (clamped addresses); the validity masks are applied when the values are written to LDS.
    // A select right behind each load makes the compiler wait for that load before issuing the next one (measured: 26
    // serialised L2 round trips = 3.6 us of a 7.5 us chunk).
    auto fetch = [&](int c0) {
#pragma unroll
        for (int cc = 0; cc < MF_CW; ++cc) {
            const int c = c0 + wave + 4 * cc;
            const float* __restrict__ fc = fi + (long)min(c, g.C - 1) * HW;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (VW == 4) {
                    const f32x4 v = *(const f32x4*)(fc + max(sp.g[q], 0));
#pragma unroll
                    for (int m = 0; m < 4; ++m) sv[cc][q][m] = v[m];
                } else {
                    sv[cc][q][0] = fc[max(sp.g[q], 0)];
                }
            }
        }
        const f32x4* __restrict__ wc = (const f32x4*)(wT + (long)(c0 >> 2) * KK * 64);
#pragma unroll
        for (int q = 0; q < WN; ++q) wv[q] = wc[min((int)threadIdx.x + 256 * q, MF_KS * KK * 16 - 1)];
    };
    auto stage = [&](int c0) {
#pragma unroll
        for (int cc = 0; cc < MF_CW; ++cc) {
            float* __restrict__ fc = fl + (wave + 4 * cc) * CS;
            const bool cok = c0 + wave + 4 * cc < g.C;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (sp.g[q] >= 0) {
#pragma unroll
                    for (int m = 0; m < VW; ++m) fc[sp.l[q] + m] = cok ? sv[cc][q][m] : 0.f;
                }
        }
#pragma unroll
        for (int q = 0; q < WN; ++q) {
            const int e = threadIdx.x + 256 * q;
            if (e < MF_KS * KK * 16) ((f32x4*)wl)[e] = wv[q];
        }
    };

    fetch(0);
    __syncthreads();                                                // zero fill done
    for (int c0 = 0; c0 < g.C; c0 += MF_CK) {
        stage(c0);
        __syncthreads();
        if (c0 + MF_CK < g.C) fetch(c0 + MF_CK);   // next chunk in flight while this one is multiplied
        // LDS operands of k-step ks+1 (KK A values, KK x NT B values) are read while the MFMAs of k-step ks issue: a
        // workgroup has one wave per SIMD, so nothing else hides the LDS latency.  No per-tile branch: a tile beyond the
        // band multiplies zeros (t_ok false), far cheaper than putting every MFMA into its own basic block.
        float av[2][KK], bv[2][KK][NT];
        auto lds_operands = [&](int ks, int set) {
            const float* __restrict__ fb = fl + (4 * ks + kq) * CS;
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                const int u = tap / K, v = tap - u * K;
                av[set][tap] = wl[(ks * KK + tap) * 64 + lane];
#pragma unroll
                for (int q = 0; q < NT; ++q) bv[set][tap][q] = fb[t_off[q] + u * g.PWs + v];
            }
        };
        lds_operands(0, 0);
#pragma unroll
        for (int ks = 0; ks < MF_KS; ++ks) {
            if (ks + 1 < MF_KS) lds_operands(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    acc[q] = mfma16(av[ks & 1][tap], t_ok[q] ? bv[ks & 1][tap][q] : 0.f, acc[q]);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int pj = 16 * (wave + 4 * q) + j;
        if (pj < npos) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * kq + r;
                if (f < g.F) scores[(long)i * out_stride_n + (long)f * HW + (long)y0 * g.W + pj] = acc[q][r];
            }
        }
    }
}

// Weights in the order the correlation consumes them: wT[c/4][tap][c%4][16 filters], zero padded to 16 filters and to a
// multiple of MF_CK channels (one contiguous 16-byte-loadable block per channel chunk).  The old layout makes every lane
// of a weight load hit its own cache line (filter stride C*K*K floats): 18 such loads per chunk cost more than the MFMAs.
__global__ void k_mf_wtrans(const float* __restrict__ filt, float* __restrict__ wT, int F, int C, int KK, int Cpad,
                            long filt_zstride, long wt_zstride) {
    filt += (long)blockIdx.y * filt_zstride;
    wT += (long)blockIdx.y * wt_zstride;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (Cpad >> 2) * KK * 64) return;
    const int ln = e & 63, tk = e >> 6, tap = tk % KK, c4 = tk / KK;
    const int f = ln & 15, c = 4 * c4 + (ln >> 4);
    wT[e] = (f < F && c < C) ? filt[((long)f * C + c) * KK + tap] : 0.f;
}

// ---------------------------------------------------------------------------------------------------
// adjoint: grid (ceil(C/16), NSG), 256 threads.  Workgroup = 16 channels x a group of samples; per (sample, row band) it
// stages the zero-padded feature band fl[16][CS2] and the input band rl[16][RS2] (strides == 2 mod 32: conflict-free
// reads with 16 channels/filters x 2 positions per half-wave) and accumulates  D_tap[f][c] += in[f][pos] * feat[c][pos+tap]
// with K = positions on the matrix cores; wave w takes the 4-position k-steps w, w+4, ...
// Output: gpart[sg][f][c][tap] (summed over sample groups by the consumer, fixed order).
// ---------------------------------------------------------------------------------------------------
template <int KK, int VW>
__global__ __launch_bounds__(256) void k_mf_adj(const float* __restrict__ feat, long stride_n,
                                                const float* __restrict__ inp, long inp_stride_n,
                                                float* __restrict__ gpart, MfGeom g, int CS2, int RS2, int spg,
                                                long inp_zstride, long gp_zstride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    inp += (long)blockIdx.z * inp_zstride;                          // blockIdx.z: group of <= 16 filters of a wider bank
    gpart += (long)blockIdx.z * gp_zstride;
    constexpr int NQ = MF_NQ / VW;
    constexpr int K = KK == 1 ? 1 : 3;
    float* __restrict__ fl = lds;                                   // [16][CS2]
    float* __restrict__ rl = lds + 16 * CS2;                        // [16][RS2]
    float* __restrict__ red = lds;                                  // [4][KK][256], reuses the staging area after the loop
    const int cb = blockIdx.x, sg = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, j = lane & 15;
    const int HW = g.H * g.W;
    const float inv_w = 1.0f / (float)g.W;
    const int i_beg = sg * spg, i_end = min(g.n, i_beg + spg);
    const int nst = (i_end - i_beg) * g.NB;                         // stages = (sample, band) pairs

    f32x4 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = (f32x4){0, 0, 0, 0};

    float sv[4][NQ][VW], rv[4][NQ][VW];
    MfStage<VW, NQ> fp, rp;                                         // plans of the stage whose data sits in sv / rv
    auto fetch = [&](int st) {
        const int i = i_beg + st / g.NB, band = st - (st / g.NB) * g.NB;
        const int y0 = band * g.BR, rows = min(g.BR, g.H - y0);
        const int ys = max(y0 - g.p, 0), ye = min(y0 + rows + g.p, g.H);
        fp.plan(lane, ys, ye - ys, g.W, inv_w, ys - (y0 - g.p), g.PWs, g.p);
        rp.plan(lane, y0, rows, g.W, inv_w, 0, g.W, 0);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = cb * 16 + wave + 4 * cc, f = wave + 4 * cc;
            const float* __restrict__ fc = feat + (long)i * stride_n + (long)min(c, g.C - 1) * HW;
            const float* __restrict__ rc = inp + (long)i * inp_stride_n + (long)min(f, g.F - 1) * HW;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {                          // raw loads; masks are applied in stage()
                if (VW == 4) {
                    const f32x4 a = *(const f32x4*)(fc + max(fp.g[q], 0)), b = *(const f32x4*)(rc + max(rp.g[q], 0));
#pragma unroll
                    for (int m = 0; m < 4; ++m) { sv[cc][q][m] = a[m]; rv[cc][q][m] = b[m]; }
                } else {
                    sv[cc][q][0] = fc[max(fp.g[q], 0)];
                    rv[cc][q][0] = rc[max(rp.g[q], 0)];
                }
            }
        }
    };
    auto stage = [&]() {
        // a band shorter than the previous one (last band of a sample) must not keep stale rows: the whole staging area
        // was cleared by all threads before (see the loop), only valid items are written here
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            float* __restrict__ fc = fl + (wave + 4 * cc) * CS2;
            float* __restrict__ rc = rl + (wave + 4 * cc) * RS2;
            const bool cok = cb * 16 + wave + 4 * cc < g.C, fok = wave + 4 * cc < g.F;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (fp.g[q] >= 0) {
#pragma unroll
                    for (int m = 0; m < VW; ++m) fc[fp.l[q] + m] = cok ? sv[cc][q][m] : 0.f;
                }
                if (rp.g[q] >= 0) {
#pragma unroll
                    for (int m = 0; m < VW; ++m) rc[rp.l[q] + m] = fok ? rv[cc][q][m] : 0.f;
                }
            }
        }
    };

    if (nst > 0) fetch(0);
    for (int st = 0; st < nst; ++st) {
        const int band = st - (st / g.NB) * g.NB;
        // first / last band of a sample have rows outside the image, and the last band may be short: clear first
        // (uniform per workgroup; interior bands overwrite every cell they read)
        if (st == 0 || band == 0 || band == g.NB - 1) {
            for (int e = threadIdx.x; e < 16 * (CS2 + RS2); e += 256) lds[e] = 0.f;
            __syncthreads();
        }
        stage();
        __syncthreads();
        if (st + 1 < nst) fetch(st + 1);
        const int rows = min(g.BR, g.H - band * g.BR), npos = rows * g.W;
        const float* __restrict__ fb = fl + j * CS2;
        const float* __restrict__ rb = rl + j * RS2;
#pragma unroll 2
        for (int s = wave; 4 * s < npos; s += 4) {
            const int pos = 4 * s + kq;
            const bool ok = pos < npos;
            const int pc = ok ? pos : 0;
            const int r = mf_fdiv(pc, inv_w), x = pc - r * g.W;
            const float a0 = rb[pc];
            const float a = ok ? a0 : 0.f;
            const int po = r * g.PWs + x;
            float bv[KK];
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) bv[tap] = fb[po + (tap / K) * g.PWs + (tap % K)];
#pragma unroll
            for (int tap = 0; tap < KK; ++tap)
                acc[tap] = mfma16(a, bv[tap], acc[tap]);            // masked k: a = 0, b finite (LDS holds data or zeros)
        }
        __syncthreads();
    }
    // ---- cross-wave reduction, fixed order
#pragma unroll
    for (int tap = 0; tap < KK; ++tap)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * KK + tap) * 256 + (4 * kq + r) * 16 + j] = acc[tap][r];
    __syncthreads();
    for (int e = threadIdx.x; e < KK * 256; e += 256) {
        const int tap = e >> 8, fc = e & 255, f = fc >> 4, c = fc & 15;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * KK + tap) * 256 + fc];
        if (f < g.F && cb * 16 + c < g.C) gpart[(((long)sg * g.F + f) * g.C + cb * 16 + c) * KK + tap] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct MfPlan {
    int ok;
    MfGeom g;            // correlation geometry
    MfGeom ga;           // adjoint geometry
    int NT;              // correlation: tiles per wave (2 or 4)
    int CS, CS2, RS2, spg, NSG;
    size_t corr_lds, adj_lds;
};

static int mf_pad_to(int v, int mod, int res) {          // smallest x >= v with x % mod == res
    int x = v;
    while ((x % mod) != res) ++x;
    return x;
}

static MfPlan mf_plan(int n, int F, int C, int H, int W, int K) {
    MfPlan p;
    p.ok = 0;
    if (n <= 0 || F <= 0 || F > 16 || C <= 0 || H <= 0 || W <= 0) return p;
    if (K != 1 && K != 3) return p;
    if (W > 256 || (long)n * C * H * W >= (1L << 31) || (long)F * C * K * K >= (1L << 30)) return p;
    MfGeom& g = p.g;
    g.n = n; g.F = F; g.C = C; g.H = H; g.W = W; g.K = K; g.p = K / 2; g.KK = K * K;
    g.PWs = W + K - 1;
    // correlation band: <= 256 positions (4 tiles per wave) and <= 512 staged floats per channel; shrink it until the
    // grid holds ~2 workgroups per CU
    int BR = 256 / W;
    if (BR > H) BR = H;
    while (BR > 1 && ((BR + K - 1) * W > 64 * MF_NQ ||
                      (n * ((H + BR - 1) / BR) < 512 && (BR - 1) * W >= 96)))      // keep >= 6 of a wave quad's 8 tile slots busy
        --BR;
    if ((BR + K - 1) * W > 64 * MF_NQ || BR * W > 256) return p;
    g.BR = BR;
    g.NB = (H + BR - 1) / BR;
    g.RSmax = BR + K - 1;
    p.NT = (BR * W + 15) / 16 <= 8 ? 2 : 4;
    // adjoint band: as above without the grid-size constraint (its grid is channel blocks x sample groups)
    p.ga = g;
    int BRa = 256 / W;
    if (BRa > H) BRa = H;
    while (BRa > 1 && (BRa + K - 1) * W > 64 * MF_NQ) --BRa;
    if ((BRa + K - 1) * W > 64 * MF_NQ) return p;
    p.ga.BR = BRa;
    p.ga.NB = (H + BRa - 1) / BRa;
    p.ga.RSmax = BRa + K - 1;
    p.CS = mf_pad_to(g.RSmax * g.PWs, 32, 16);
    p.CS2 = mf_pad_to(p.ga.RSmax * g.PWs, 32, 2);
    p.RS2 = mf_pad_to(BRa * W, 32, 2);
    p.corr_lds = ((size_t)MF_CK * p.CS + MF_KS * g.KK * 64) * sizeof(float);
    p.adj_lds = std::max((size_t)16 * (p.CS2 + p.RS2), (size_t)4 * g.KK * 256) * sizeof(float);
    const int CBn = (C + 15) / 16;
    int NSG = 512 / CBn;                                 // ~2 workgroups per CU
    if (NSG < 1) NSG = 1;
    if (NSG > n) NSG = n;
    if (NSG > 32) NSG = 32;
    p.spg = (n + NSG - 1) / NSG;
    p.NSG = (n + p.spg - 1) / p.spg;
    if (p.corr_lds > 150 * 1024 || p.adj_lds > 150 * 1024) return p;
    p.ok = 1;
    return p;
}

size_t pt_mf_wt_floats(int C, int K) { return (size_t)(((C + MF_CK - 1) / MF_CK) * MF_CK / 4) * K * K * 64; }

int pt_launch_mf_wtrans(const float* filt, float* wT, int F, int C, int K, hipStream_t st, int groups) {
    const int Cpad = ((C + MF_CK - 1) / MF_CK) * MF_CK, total = (Cpad >> 2) * K * K * 64;
    hipLaunchKernelGGL(k_mf_wtrans, dim3((total + 255) / 256, groups), dim3(256), 0, st, filt, wT, F, C, K * K, Cpad,
                       (long)F * C * K * K, (long)pt_mf_wt_floats(C, K));
    PT_CHECK_LAUNCH();
    return PT_OK;
}

size_t pt_mf_gpart_floats(int n, int F, int C, int H, int W, int K) {
    MfPlan p = mf_plan(n, F, C, H, W, K);
    return p.ok ? (size_t)p.NSG * F * C * K * K : 0;
}
int pt_mf_groups(int n, int F, int C, int H, int W, int K) {
    MfPlan p = mf_plan(n, F, C, H, W, K);
    return p.ok ? p.NSG : 0;
}

static bool mf_vec_ok(const float* a, const float* b, long stride_n, int W) {
    return (W % 4) == 0 && (stride_n % 4) == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0;
}

// wT: weights pre-transposed by pt_launch_mf_wtrans (pt_mf_wt_floats(C, K) floats, 16-byte aligned)
int pt_launch_mf_corr(const float* feat, long stride_n, const float* wT, float* scores, int n, int F, int C, int H,
                      int W, int K, hipStream_t st, long out_stride_n, int groups) {
    MfPlan p = mf_plan(n, F, C, H, W, K);
    if (out_stride_n == 0) out_stride_n = (long)F * H * W;
    if (!p.ok || groups < 1) return PT_ERR_UNSUPPORTED;
    // groups > 1: `groups` banks of F filters each (weight tables back to back, outputs F*H*W apart inside a sample)
    const long wt_zs = (long)pt_mf_wt_floats(C, K), out_zs = (long)F * H * W;
    dim3 grid(p.g.NB, n, groups), block(256);
    const bool vec = mf_vec_ok(feat, feat, stride_n, W);
#define PT_MFC(KKV, NTV, VWV) \
    hipLaunchKernelGGL((k_mf_corr<KKV, NTV, VWV>), grid, block, p.corr_lds, st, feat, stride_n, wT, scores, out_stride_n, p.g, p.CS, wt_zs, out_zs)
    if (K == 1) {
        if (p.NT == 2) { if (vec) PT_MFC(1, 2, 4); else PT_MFC(1, 2, 1); }
        else { if (vec) PT_MFC(1, 4, 4); else PT_MFC(1, 4, 1); }
    } else {
        if (p.NT == 2) { if (vec) PT_MFC(9, 2, 4); else PT_MFC(9, 2, 1); }
        else { if (vec) PT_MFC(9, 4, 4); else PT_MFC(9, 4, 1); }
    }
#undef PT_MFC
    PT_CHECK_LAUNCH();
    return PT_OK;
}

int pt_launch_mf_adj(const float* feat, long stride_n, const float* inp, float* gpart, int n, int F, int C, int H, int W,
                     int K, hipStream_t st, long inp_stride_n, int groups) {
    MfPlan p = mf_plan(n, F, C, H, W, K);
    if (inp_stride_n == 0) inp_stride_n = (long)F * H * W;
    if (!p.ok || groups < 1) return PT_ERR_UNSUPPORTED;
    // groups > 1: banks of F filter maps F*H*W apart inside a sample, partials of a bank NSG*F*C*K*K floats apart
    const long inp_zs = (long)F * H * W, gp_zs = (long)p.NSG * F * C * K * K;
    dim3 grid((C + 15) / 16, p.NSG, groups), block(256);
    const bool vec = mf_vec_ok(feat, inp, stride_n, W) && ((H * W) % 4) == 0 && (inp_stride_n % 4) == 0;
#define PT_MFA(KKV, VWV) \
    hipLaunchKernelGGL((k_mf_adj<KKV, VWV>), grid, block, p.adj_lds, st, feat, stride_n, inp, inp_stride_n, gpart, p.ga, p.CS2, p.RS2, p.spg, inp_zs, gp_zs)
    if (K == 1) { if (vec) PT_MFA(1, 4); else PT_MFA(1, 1); }
    else { if (vec) PT_MFA(9, 4); else PT_MFA(9, 1); }
#undef PT_MFA
    PT_CHECK_LAUNCH();
    return PT_OK;
}

