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
    int PWs;             // padded row stride in LDS (W + K - 1): adjoint
    int RSmax;           // staged rows per band (BR + K - 1)
    int Wp, PS;          // correlation: row length the positions are counted on (roundup4(W)), LDS row stride (Wp + 4)
    int out_vec;         // correlation: 16-byte stores of the scores are legal
};


__device__ __forceinline__ int mf_fdiv(int v, float inv_d) { return (int)(((float)v + 0.5f) * inv_d); }

// 16-byte LDS accesses.  Through a plain cast of `lds + offset` clang derives the alignment from the array (4 bytes for an
// unknown offset) and the backend splits the access into two ds_read2_b32 / ds_write2_b32.
__device__ __forceinline__ f32x4 mf_lds_ld4(const float* p) { return *(const f32x4*)__builtin_assume_aligned(p, 16); }
__device__ __forceinline__ void mf_lds_st4(float* p, f32x4 v) { *(f32x4*)__builtin_assume_aligned(p, 16) = v; }
__device__ __forceinline__ f32x2 mf_lds_ld2(const float* p) { return *(const f32x2*)__builtin_assume_aligned(p, 8); }

// VW (1, 2, 4) consecutive floats: range-checked buffer load / LDS store (VW*4-byte aligned)
template <int VW>
__device__ __forceinline__ void mf_bload(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float* out) {
    if (VW == 4) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
#pragma unroll
        for (int m = 0; m < 4; ++m) out[m] = v[m];
    } else if (VW == 2) {
        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
        out[0] = v[0]; out[1] = v[1];
    } else {
        out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    }
}
template <int VW>
__device__ __forceinline__ void mf_lds_store(float* p, const float* v) {
    if (VW == 4) mf_lds_st4(p, (f32x4){v[0], v[1], v[2], v[3]});
    else if (VW == 2) *(f32x2*)__builtin_assume_aligned(p, 8) = (f32x2){v[0], v[1]};
    else p[0] = v[0];
}

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
// correlation: grid (NB, n), 512 threads, one band of BR output rows per workgroup.
//
// What bounds it (experiments/mfma_issue.hip, profiles/r02i_mfma_issue_microbench.json): on this chip nothing issues in
// the shadow of a v_mfma_f32_16x16x4_f32 -- every other VALU instruction of the wavefront costs ~5 cycles on top of the
// MFMA's 32, every LDS instruction ~4-8.  So the kernel is laid out for FEW instructions per MFMA, not for overlap:
//   * positions are counted on rows padded to Wp (roundup4(W + 1) for 3x3); lane (kq, j) of wave w owns the 4 CONSECUTIVE
//     positions 64w + 4j + q, q = 0..3 (its 4 MFMA tiles).  One ds_read_b128 and two ds_read_b64 of the padded row (the lane's
//     quad and the pairs either side of it; 6 of the 8 floats are used) hold the B operands of all 4 tiles x 3 horizontal taps:
//     9 LDS reads per 36 MFMAs of a k-step at 4 LDS cycles each (round 6; the neighbours used to be two ds_read_b32 with 5-way
//     bank conflicts, 10 cycles each);
//   * the weights sit in LDS as [k-step][lane][12] (9 taps + pad): 3 ds_read_b128 per k-step, conflict-free (lane stride
//     12 dwords); k_mf_wtrans writes the table in exactly this order, staging it is a straight 16-byte copy;
//   * no masks on the operands: lanes beyond the band read offset 0 and only feed output columns that are never stored.
// Channels are streamed in chunks of MF_CK (MF_KS k-steps) through two LDS buffers:
//      fl[MF_CK][CS]   zero-padded band: 4 floats, then rows of PS = Wp floats with the data at column x -- the zero tail of a
//                      row is the right padding of that row and the left padding of the next one, and a float4 of the image
//                      lands 16-byte aligned; CS == 0 (mod 64), see mf_plan
//      wl[MF_KS][64][TP]
// 8 waves: wave w owns position group w & 3 (64 positions) and k-steps 2(w>>2), 2(w>>2)+1 of every chunk -- two waves per
// SIMD, because a wave's LDS / global / barrier time does not overlap its own MFMAs either (measured: with one wave per
// SIMD every component adds linearly, profiles/r02i_lwl_kernel_ablation.txt); the two halves are summed through LDS at
// the end.  Wave w stages channels w and w+8 of a chunk.  Per wave, while chunk c is multiplied out of buffer c&1, chunk
// c+1 goes from registers into the other buffer and the global loads of chunk c+2 are issued.  The chunk's barrier sits in
// front of the wave's second k-step (see the loop).
// ---------------------------------------------------------------------------------------------------
#define MF_CK 16
#define MF_KS (MF_CK / 4)
#define MF_NQ 8            // scalar staging items per lane per plane: rows*W <= 512
#define MF_TP(KK) ((KK) == 1 ? 1 : 12)   // weight floats per lane per k-step (pt_mf_wt_index)
#define MF_CT 512          // threads of the correlation workgroup

// WTM: the weights come tap-major, (filters, 9, C) -- the layout the classification-feature head keeps its 3x3 weights in --
//      and are transposed into the [k-step][lane][12] order on their way into LDS (C % 16 == 0, 16 filters per bank).
// ksplit > 1: blockIdx.y = sample + n * split; split s multiplies its share of the channel chunks and writes its partial map
//      to scores + s * part_zstride (summed by the caller in fixed order): a single frame has too few bands to fill the chip.
// NPG: position groups of 64 per workgroup (4, or 6 = 12 waves for a whole 18x18 map as one band: 360 of 384 slots instead of
//      2 bands of 180 of 256).  Waves = 2 NPG (two k halves), 128 NPG threads.
template <int KK, int VW, bool WTM, int NPG = 4>
__global__ __launch_bounds__(128 * NPG) void k_mf_corr(const float* __restrict__ feat, long stride_n,
                                                   const float* __restrict__ wT, float* __restrict__ scores,
                                                   long out_stride_n, MfGeom g, int CS, long wt_zstride, long out_zstride,
                                                   int ksplit, long part_zstride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(!WTM || KK == 9, "tap-major weights: 3x3 only");
    wT += (long)blockIdx.z * wt_zstride;                            // blockIdx.z: group of <= 16 filters of a wider bank
    scores += (long)blockIdx.z * out_zstride;
    constexpr int NQ = MF_NQ / VW;
    constexpr int K = KK == 1 ? 1 : 3;
    constexpr int TP = MF_TP(KK);
    constexpr int NB6 = K == 1 ? 4 : 6;                             // floats of a padded row one lane reads per (k-step, u)
    constexpr int WITEMS = WTM ? 16 * 9 * MF_KS : MF_KS * 64 * TP / 4;   // 16-byte items of a chunk's weight block
    constexpr int NT = 128 * NPG, NWV = 2 * NPG;                    // threads, waves
    constexpr int WN = (WITEMS + NT - 1) / NT;
    static_assert(MF_KS == 4 && MF_CK == 16 && WN <= 2 && 2 * NWV >= MF_CK, "two k-steps and two staging pieces per wave and chunk");
    const int BUF = MF_CK * CS + MF_KS * 64 * TP;                   // floats per buffer
    const int band = blockIdx.x, i = blockIdx.y % g.n, ksp = blockIdx.y / g.n;
    scores += (long)ksp * part_zstride;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pg = wave % NPG, kh = wave / NPG;
    const int kq = lane >> 4, j = lane & 15;
    const int y0 = band * g.BR, rows = min(g.BR, g.H - y0);
    const int HW = g.H * g.W;
    const float inv_w = 1.0f / (float)g.W;
    const float* __restrict__ fi = feat + (long)i * stride_n;

    // image rows [ys, ye) of the band + halo are contiguous in every channel plane
    const int ys = max(y0 - g.p, 0), ye = min(y0 + rows + g.p, g.H);
    MfStage<VW, NQ> sp;
    sp.plan(lane, ys, ye - ys, g.W, inv_w, ys - (y0 - g.p), g.PS, 4);
    const int nch_all = (g.C + MF_CK - 1) / MF_CK, cps = (nch_all + ksplit - 1) / ksplit;
    const int cbeg = ksp * cps, nchunks = min(nch_all, cbeg + cps) - cbeg;   // >= 1 (the launcher sees to it)

    // ---- the 4 positions of this lane: P0 .. P0 + 3 on rows of Wp
    const int P0 = 64 * pg + 4 * j;
    const int pr = P0 / g.Wp, px = P0 - pr * g.Wp;
    const int t_off = 4 + (pr < rows ? pr * g.PS + px : 0);
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0, 0, 0, 0};

    float sv[2][NQ][VW];
    f32x4 wv[WN];
    // piece cc (0, 1) of a chunk = channel wave + 8 cc (NQ loads) and weight item cc.  Buffer loads: SGPR base (channel
    // plane / chunk block) + invariant VGPR offset, no address arithmetic on the VALU; an item without a source has an
    // out-of-range offset (reads 0) and goes to the thread's dump slot.  A chunk index beyond the last one re-fetches
    // the last chunk; a channel >= C (channel count not a multiple of MF_CK) reads channel C - 1 against zero weights.
    const __amdgpu_buffer_rsrc_t rsF = pt_rsrc(fi, (unsigned)g.C * (unsigned)HW * 4u);
    const __amdgpu_buffer_rsrc_t rsW = pt_rsrc(wT, WTM ? (unsigned)g.F * 9u * (unsigned)g.C * 4u
                                                       : (unsigned)nch_all * (unsigned)(MF_KS * 64 * TP) * 4u);
    const int dump = 2 * BUF + 4 * threadIdx.x;
    unsigned goff[NQ], woff[WN];
    int lrow[NQ], lwt[WN];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        goff[q] = sp.g[q] >= 0 ? 4u * (unsigned)sp.g[q] : 0x80000000u;
        lrow[q] = sp.g[q] >= 0 ? sp.l[q] : -1;
    }
#pragma unroll
    for (int cc = 0; cc < WN; ++cc) {
        const int e = threadIdx.x + NT * cc;
        if (WTM) {                                                  // item = (filter f, tap, k-step): channels 4 ks .. + 3
            const int ks = e & 3, ft = e >> 2, f = ft / 9, tap = ft - 9 * f;
            woff[cc] = e < WITEMS ? 4u * (unsigned)((f * 9 + tap) * g.C + 4 * ks) : 0x80000000u;
            lwt[cc] = e < WITEMS ? MF_CK * CS + (ks * 64 + f) * TP + tap : -1;
        } else {
            woff[cc] = e < WITEMS ? 16u * e : 0x80000000u;
            lwt[cc] = e < WITEMS ? MF_CK * CS + 4 * e : -1;
        }
    }
    auto fetch_piece = [&](int ci, int cc) {
        const int c0 = (cbeg + min(ci, nchunks - 1)) * MF_CK;
        const unsigned soff = (unsigned)min(c0 + min(wave + NWV * cc, MF_CK - 1), g.C - 1) * (unsigned)HW * 4u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) mf_bload<VW>(rsF, goff[q], soff, sv[cc][q]);
        if (cc < WN)
            wv[cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsW, woff[cc], WTM ? (unsigned)c0 * 4u : (unsigned)(c0 >> 2) * (64 * TP * 4), 0));
    };
    auto stage_piece = [&](int buf, int cc) {
        const int row = buf * BUF + (wave + NWV * cc) * CS;
        if (wave + NWV * cc < MF_CK) {                              // (12 waves: the second piece exists for waves 0..3 only)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int at = lrow[q] >= 0 ? row + lrow[q] : dump;
                mf_lds_store<VW>(lds + at, sv[cc][q]);
            }
        }
        if (cc < WN) {
            if (WTM) {                                              // 4 channels of one (filter, tap): lanes kq = 0..3
#pragma unroll
                for (int m = 0; m < 4; ++m) lds[lwt[cc] >= 0 ? buf * BUF + lwt[cc] + m * 16 * TP : dump + m] = wv[cc][m];
            } else {
                mf_lds_st4(lds + (lwt[cc] >= 0 ? buf * BUF + lwt[cc] : dump), wv[cc]);
            }
        }
    };
    // LDS operands of one k-step: the lane's weights of all taps, and per vertical tap NB6 floats of the padded row --
    // NRD single read instructions (read_op), so that they can be placed one by one between the MFMAs.
    float av[2][KK], bv[2][K][NB6];
    constexpr int NRD = KK == 1 ? 2 : 12;
    auto read_op = [&](int buf, int ks, int set, int k) {
        const float* __restrict__ fb = lds + buf * BUF + (4 * ks + kq) * CS + t_off;
        const float* __restrict__ wl = lds + buf * BUF + MF_CK * CS + (ks * 64 + lane) * TP;
        if (KK == 1) {
            if (k == 0) av[set][0] = wl[0];
            if (k == 1) {
                const f32x4 v = mf_lds_ld4(fb);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[set][0][e] = v[e];
            }
            return;
        }
        if (k < 2) {                                                // 9 of the 12 floats: a register that is loaded but
            const f32x4 v = mf_lds_ld4(wl + 4 * k);            // never read gets reused while the load is in flight
#pragma unroll
            for (int e = 0; e < 4; ++e) av[set][4 * k + e] = v[e];
        } else if (k == 2) {
            av[set][8] = mf_lds_ld2(wl + 8)[0];                     // 8-byte read: as a ds_read_b32 at a lane stride of 12 dwords a 4-way conflict
        } else {                                                    // columns px - 1 .. px + 4 of row pr + u: the lane's aligned quad, the
            const int u = (k - 3) / 3, part = (k - 3) % 3;          // pair left of it and the pair right of it.  (The neighbours as two
            if (part == 0) {                                        // ds_read_b32 were 5-way bank conflicts: 32 lanes on 8 banks.)
                const f32x4 v = mf_lds_ld4(fb + u * g.PS);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[set][u][e + 1] = v[e];
            } else if (part == 1) {
                bv[set][u][0] = mf_lds_ld2(fb + u * g.PS - 2)[1];
            } else {
                bv[set][u][5] = mf_lds_ld2(fb + u * g.PS + 4)[0];
            }
        }
    };
    auto lds_operands = [&](int buf, int ks, int set) {
#pragma unroll
        for (int k = 0; k < NRD; ++k) read_op(buf, ks, set, k);
    };
    // One k-step: the MFMAs on operand set `set`, and behind the first ones, one each and pinned there (left alone the
    // scheduler sinks every read to its first use), the memory instructions that go with it: the operand reads of the
    // wave's next k-step (rbuf, rks -> the other set; they need the rest of this k-step to land), the LDS stores of
    // staging piece cc into sbuf, the loads that refill its registers from chunk fci.
    auto kstep = [&](int set, int rbuf, int rks, int sbuf, int cc, int fci) {
        constexpr int NMF = KK * 4;
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            const int tap = m / 4, q = m - 4 * tap, u = tap / K, v = tap - u * K;
            acc[q] = mfma16(av[set][tap], bv[set][u][q + v], acc[q]);
            if (KK > 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (m < NRD) read_op(rbuf, rks, set ^ 1, m);
                if (m == NRD) stage_piece(sbuf, cc);
                if (m == NRD + 3) fetch_piece(fci, cc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (KK == 1) {
            lds_operands(rbuf, rks, set ^ 1);
            stage_piece(sbuf, cc);
            fetch_piece(fci, cc);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    fetch_piece(0, 0);
    fetch_piece(0, 1);
    // padding (and rows outside the image) stay zero; 16-byte stores, issued BEHIND the first loads so that the fill runs under their
    // round trip (round 6: it used to be 2 BUF scalar stores in front of everything -- 27 per thread on the 12-wave head kernel)
    for (int e = 4 * (int)threadIdx.x; e < 2 * BUF; e += 4 * NT) mf_lds_st4(lds + e, (f32x4){0.f, 0.f, 0.f, 0.f});
    __syncthreads();                                                // zero fill done
    stage_piece(0, 0);
    stage_piece(0, 1);
    fetch_piece(1, 0);
    fetch_piece(1, 1);
    stage_piece(1, 0);
    fetch_piece(2, 0);
    __syncthreads();
    lds_operands(0, 2 * kh, 0);
    __builtin_amdgcn_sched_barrier(0);
    // Registers hold chunk c+1 piece 1 and chunk c+2 piece 0 when chunk c starts.  With a k-step's MFMAs go: the reads of
    // the wave's NEXT k-step, one staging piece (registers -> LDS) and the loads that refill those registers.
    //   first k-step : reads of the second one; piece 1 of chunk c+1 -> other buffer; loads piece 1 of chunk c+2
    //   second k-step: reads of chunk c+1's first one from the other buffer; piece 0 of chunk c+2 -> this buffer (nobody
    //                  reads it any more); loads piece 0 of chunk c+3
    // so the chunk's barrier sits between the two.
    const int k0 = 2 * kh, k1 = 2 * kh + 1;
    for (int ci = 0; ci < nchunks; ci += 2) {
        kstep(0, 0, k1, 1, 1, ci + 2);
        __syncthreads();
        kstep(1, 1, k0, 0, 0, ci + 3);
        if (ci + 1 < nchunks) {
            kstep(0, 1, k1, 0, 1, ci + 3);
            __syncthreads();
            kstep(1, 0, k0, 1, 0, ci + 4);
        }
    }
    // sum of the two k halves: waves 4..7 hand their accumulators to waves 0..3 through LDS ([pg][q][r][lane])
    __syncthreads();
    if (kh == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[((pg * 4 + q) * 4 + r) * 64 + lane] = acc[q][r];   // NPG x 16 x 64 floats
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] += lds[((pg * 4 + q) * 4 + r) * 64 + lane];
    // lane (kq, j), tile q, register r: filter 4 kq + r at position P0 + q
    if (pr < rows) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 4 * kq + r;
            float* __restrict__ o = scores + (long)i * out_stride_n + (long)f * HW + (long)(y0 + pr) * g.W + px;
            if (f < g.F) {
                if (VW == 4 && g.out_vec) {
                    if (px < g.W) *(f32x4*)o = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (px + q < g.W) o[q] = acc[q][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 1x1 correlation without LDS staging (ATOM's projection: 256 -> 64 channels on 18x18 maps).  With a single tap nothing is
// reused between positions, so the B operand goes straight from global memory to the lane that multiplies it: lane
// (kq, j) of every wave owns the 4 consecutive positions P0 + q (P0 = 64 blockIdx.x + 4 j) and loads them per channel
// with one 16-byte load (a wave load = 4 rows of 256 contiguous bytes).  The A operand comes from the transposed table
// ([c/4][lane], one coalesced dword per k-step and bank) or, DIRECT, from the (F, C) weight matrix itself (16 bytes per
// lane and bank for 4 k-steps: no transposition launch in front).  A workgroup serves NBK banks of 16 filters with the same B
// registers; its 4 waves split the channels and are summed through LDS.  No barrier in the loop; the banded LDS kernel
// spent 17 us on this (16 chunks x (barrier + exposed load latency)) for 10 MB of input.
// grid (ceil(HW/64), n, ceil(groups/NBK)), 256 threads.  HW % 4 == 0.
// ---------------------------------------------------------------------------------------------------
template <int NBK, bool DIRECT>
__global__ __launch_bounds__(256) void k_mf_corr1(const float* __restrict__ feat, long stride_n, const float* __restrict__ wT,
                                                  float* __restrict__ scores, long out_stride_n, int F, int C, int HW,
                                                  int groups, long wt_zstride, long out_zstride, int out_vec) {
    __shared__ __attribute__((aligned(16))) float red[3][NBK * 16][64];
    constexpr int PD = 2;                                           // 16-channel blocks in flight
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, j = lane & 15;
    const int i = blockIdx.y, b0 = blockIdx.z * NBK;
    const int P0 = 64 * blockIdx.x + 4 * j;
    const bool pok = P0 < HW;
    const int nblk = (C + MF_CK - 1) / MF_CK;                       // the table is zero padded to whole blocks
    const int per = (nblk + 3) / 4, cb0 = wave * per, cb1 = min(nblk, cb0 + per);
    const __amdgpu_buffer_rsrc_t rsF = pt_rsrc(feat + (long)i * stride_n, (unsigned)C * (unsigned)HW * 4u);
    const unsigned boff = pok ? 4u * P0 : 0x80000000u;              // beyond the map: reads 0, never stored
    f32x4 acc[NBK][4];
#pragma unroll
    for (int b = 0; b < NBK; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[b][q] = (f32x4){0, 0, 0, 0};
    // k-step m of block cb multiplies channel 16 cb + 4 m + kq (table order), or 16 cb + 4 kq + m when the weights are
    // read DIRECTly from the (F, C) matrix: then a lane's 4 k-steps are 4 consecutive floats of its filter row (C % 16 == 0)
    f32x4 bq[PD][4];
    float aq[PD][NBK][4];
    auto fetch = [&](int cb, int sl) {
        const int cbc = min(cb, nblk - 1);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int c = min(16 * cbc + (DIRECT ? 4 * kq + m : 4 * m + kq), C - 1);      // a channel >= C meets zero weights
            bq[sl][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsF, boff + (unsigned)c * (unsigned)HW * 4u, 0, 0));
        }
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            const int bank = min(b0 + b, groups - 1);
            if (DIRECT) {
                const f32x4 v = *(const f32x4*)(wT + ((long)bank * 16 + j) * C + 16 * cbc + 4 * kq);
#pragma unroll
                for (int m = 0; m < 4; ++m) aq[sl][b][m] = v[m];
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) aq[sl][b][m] = wT[(long)bank * wt_zstride + (long)(4 * cbc + m) * 64 + lane];
            }
        }
    };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) fetch(cb0 + sl, sl);
    for (int cb = cb0; cb < cb1; cb += PD) {
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) {
            if (cb + sl < cb1) {
                f32x4 bv[4];
                float av[NBK][4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    bv[m] = bq[sl][m];
#pragma unroll
                    for (int b = 0; b < NBK; ++b) av[b][m] = aq[sl][b][m];
                }
                fetch(cb + sl + PD, sl);                            // past the range: clamped, unused
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int b = 0; b < NBK; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[b][q] = mfma16(av[b][m], bv[m][q], acc[b][q]);
            }
        }
    }
    // sum of the 4 channel ranges, fixed order; lane (kq, j), register r: filter 4 kq + r at positions P0 + q
    if (wave > 0) {
#pragma unroll
        for (int b = 0; b < NBK; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave - 1][(b * 4 + q) * 4 + r][lane] = acc[b][q][r];
    }
    __syncthreads();
    if (wave > 0 || !pok) return;
#pragma unroll
    for (int b = 0; b < NBK; ++b) {
        if (b0 + b >= groups) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = acc[b][q][r];
#pragma unroll
                for (int w = 0; w < 3; ++w) v[q] += red[w][(b * 4 + q) * 4 + r][lane];
            }
            const int f = 4 * kq + r;
            if (f < F) {
                float* __restrict__ o = scores + (long)(b0 + b) * out_zstride + (long)i * out_stride_n + (long)f * HW + P0;
                if (out_vec) *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
                else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
            }
        }
    }
}

// Weights in the order the correlation consumes them: wT[c/4][lane = (c%4)*16 + filter][MF_TP taps], zero padded to 16
// filters, to 12 taps and to a multiple of MF_CK channels (one contiguous 16-byte-loadable block per channel chunk).
// clear2 (optional): a second table of the same size that this launch zeroes on the side (the few-shot learner's transposed
// gradient table, whose padding is written once per solve -- no memset node in front of the solve)
__global__ void k_mf_wtrans(const float* __restrict__ filt, float* __restrict__ wT, int F, int C, int KK, int Cpad,
                            long filt_zstride, long wt_zstride, float* __restrict__ clear2) {
    filt += (long)blockIdx.y * filt_zstride;
    wT += (long)blockIdx.y * wt_zstride;
    const int TP = MF_TP(KK);
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (Cpad >> 2) * 64 * TP) return;
    if (clear2 && blockIdx.y == 0) clear2[e] = 0.f;
    const int tap = e % TP, ln = (e / TP) & 63, c4 = e / (TP * 64);
    const int f = ln & 15, c = 4 * c4 + (ln >> 4);
    wT[e] = (tap < KK && f < F && c < C) ? filt[((long)f * C + c) * KK + tap] : 0.f;
}

// ---------------------------------------------------------------------------------------------------
// adjoint: grid (ceil(C/16), NSG), 256 threads.  Workgroup = 16 channels x a group of samples; a stage = one (sample, row
// band): the zero-padded feature band fl[16][CS2] (layout of the correlation: 4 floats, rows of PS = Wp + 4, data at column
// x) and the input band rl[16][RS2] (rows of Wp) in LDS, and  D_tap[f][c] += in[f][pos] * feat[c][pos+tap]  with K =
// positions on the matrix cores.  Same issue model as the correlation (few instructions per MFMA, see there):
//   * positions are counted on rows of Wp and taken in groups of 16; lane kq owns the 4 CONSECUTIVE positions 16g + 4kq + m
//     and feeds them to 4 MFMAs: one ds_read_b128 of the input row, and per vertical tap one ds_read_b128 + two ds_read_b64 of
//     the feature row (the quad and the pairs either side; 6 floats = 4 positions x 3 horizontal taps are used): 10 LDS reads of
//     4 LDS cycles each per 36 MFMAs (round 6; the neighbours were one ds_read2_b32 = two 4-way bank conflicts, 16 cycles);
//   * wave w takes the groups 4 s + (w + stage) % 4 -- rotating, so that a band of 13 groups costs every SIMD 3.25 slots
//     on average; a slot without a group skips its MFMAs;
//   * two LDS buffers; while stage t is multiplied, stage t+1 goes from registers to the other buffer and the buffer loads
//     of stage t+2 are issued, one piece (channel wave + 4k, filter wave + 4k) per slot, pinned behind the slot's first
//     MFMAs.  The loads are range checked against the channel plane: rows above / below the image arrive as zeros, so no
//     buffer is ever cleared.  The stage's barrier sits in front of its last slot (as in the correlation).
// Output: gpart[sg][f][c][tap] (summed over sample groups by the consumer, fixed order).
// ---------------------------------------------------------------------------------------------------
template <int KK, int VW>
__global__ __launch_bounds__(256) void k_mf_adj(const float* __restrict__ feat, long stride_n,
                                                const float* __restrict__ inp, long inp_stride_n,
                                                float* __restrict__ gpart, MfGeom g, int CS2, int RS2, int spg, int bps,
                                                long inp_zstride, long gp_zstride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    inp += (long)blockIdx.z * inp_zstride;                          // blockIdx.z: group of <= 16 filters of a wider bank
    gpart += (long)blockIdx.z * gp_zstride;
    constexpr int K = KK == 1 ? 1 : 3;
    constexpr int NQF = 8 / VW, NQI = 4 / VW;                       // staging items per lane: feature channel / input plane
    constexpr int NB6 = K == 1 ? 4 : 6;
    constexpr int NRD = K == 1 ? 2 : 10;
    constexpr int NWR = NQF + NQI;
    static_assert(NRD + 2 * NWR <= 4 * KK || KK == 1, "memory instructions of a slot fit behind its MFMAs");
    const int BUF = 16 * (CS2 + RS2);
    float* __restrict__ red = lds;                                  // [4][KK][256], reuses the buffers after the loop
    const int cb = blockIdx.x, sg = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, j = lane & 15;
    const int HW = g.H * g.W;
    const float inv_w = 1.0f / (float)g.W, inv_wp = 1.0f / (float)g.Wp;
    // group sg: spg samples with all their bands, or (bps > 1, few samples) one sample's share of the bands
    const int bpg = (g.NB + bps - 1) / bps;
    const int i_beg = bps > 1 ? sg / bps : sg * spg, i_end = bps > 1 ? i_beg + 1 : min(g.n, i_beg + spg);
    const int b_beg = bps > 1 ? (sg % bps) * bpg : 0, b_cnt = bps > 1 ? max(0, min(g.NB, b_beg + bpg) - b_beg) : g.NB;
    const int nst = (i_end - i_beg) * b_cnt;                        // stages = (sample, band) pairs
    const int dump = 2 * BUF + 4 * threadIdx.x;

    f32x4 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = (f32x4){0, 0, 0, 0};
    for (int e = threadIdx.x; e < 2 * BUF; e += 256) lds[e] = 0.f;  // the padding stays zero

    // staging items: item q of a plane covers VW floats from e = VW (lane + 64 q) of the staged rows (full width, contiguous)
    unsigned fsrc[NQF], isrc[NQI];                                  // byte offset in the plane relative to the first staged row
    int fdst[NQF], idst[NQI];                                       // LDS offset inside the channel / filter row, or -1
#pragma unroll
    for (int q = 0; q < NQF; ++q) {
        const int e = VW * (lane + 64 * q);
        const int rr = mf_fdiv(e, inv_w), x = e - rr * g.W;
        const bool ok = e < g.RSmax * g.W;
        fsrc[q] = ok ? 4u * e : 0x40000000u;
        fdst[q] = ok ? 4 + rr * g.PS + x : -1;
    }
#pragma unroll
    for (int q = 0; q < NQI; ++q) {
        const int e = VW * (lane + 64 * q);
        const int rr = mf_fdiv(e, inv_w), x = e - rr * g.W;
        const bool ok = e < g.BR * g.W;
        isrc[q] = ok ? 4u * e : 0x40000000u;
        idst[q] = ok ? rr * g.Wp + x : -1;
    }
    float sv[4][NQF][VW], rv[4][NQI][VW];
    // piece cc of stage st: channel cb*16 + wave + 4 cc and filter wave + 4 cc.  One buffer resource per plane: a row
    // offset below 0 or beyond H*W is out of range and reads 0.  A stage index beyond the last re-fetches the last stage.
    auto fetch_piece = [&](int st, int cc) {
        const int sc = min(st, nst - 1);
        const int si = sc / b_cnt, band = b_beg + sc - si * b_cnt;
        const int i = i_beg + si, y0 = band * g.BR;
        const int c = min(cb * 16 + wave + 4 * cc, g.C - 1), f = min(wave + 4 * cc, g.F - 1);
        const __amdgpu_buffer_rsrc_t rsF = pt_rsrc(feat + (long)i * stride_n + (long)c * HW, (unsigned)HW * 4u);
        const __amdgpu_buffer_rsrc_t rsI = pt_rsrc(inp + (long)i * inp_stride_n + (long)f * HW, (unsigned)HW * 4u);
        const unsigned frow = (unsigned)((y0 - g.p) * g.W * 4), irow = (unsigned)(y0 * g.W * 4);
#pragma unroll
        for (int q = 0; q < NQF; ++q) mf_bload<VW>(rsF, fsrc[q] + frow, 0, sv[cc][q]);
#pragma unroll
        for (int q = 0; q < NQI; ++q) mf_bload<VW>(rsI, isrc[q] + irow, 0, rv[cc][q]);
    };
    auto stage_piece = [&](int buf, int cc) {
        const int frw = buf * BUF + (wave + 4 * cc) * CS2, irw = buf * BUF + 16 * CS2 + (wave + 4 * cc) * RS2;
#pragma unroll
        for (int q = 0; q < NQF; ++q) {
            const int at = fdst[q] >= 0 ? frw + fdst[q] : dump;
            mf_lds_store<VW>(lds + at, sv[cc][q]);
        }
#pragma unroll
        for (int q = 0; q < NQI; ++q) {
            const int at = idst[q] >= 0 ? irw + idst[q] : dump;
            mf_lds_store<VW>(lds + at, rv[cc][q]);
        }
    };
    // slot s of stage st: group 4 s + (wave + st) % 4, positions 16 g + 4 kq + m; operands of its 4 k-steps
    float av[2][4], bv[2][K][NB6];
    int a_at, b_at;                                                 // LDS offsets of the slot whose reads are being issued
    auto slot_address = [&](int buf, int st, int sl) {
        const int gi = 4 * sl + ((wave + st) & 3);
        const int P = 16 * gi < g.BR * g.Wp ? 16 * gi + 4 * kq : 4 * kq;       // beyond the band: group 0 (never multiplied)
        const int r = mf_fdiv(P, inv_wp);
        a_at = buf * BUF + 16 * CS2 + j * RS2 + P;
        b_at = buf * BUF + j * CS2 + 4 + P + 4 * r;                 // 4 + r PS + x, PS = Wp + 4
    };
    auto slot_valid = [&](int st, int sl) {                         // wave uniform
        const int sc = min(st, nst - 1), band = b_beg + sc - (sc / b_cnt) * b_cnt;
        const int rows = min(g.BR, g.H - band * g.BR);
        return st < nst && 16 * (4 * sl + ((wave + st) & 3)) < rows * g.Wp;
    };
    auto read_op = [&](int set, int k) {
        if (k == 0) {
            const f32x4 v = mf_lds_ld4(lds + a_at);
#pragma unroll
            for (int e = 0; e < 4; ++e) av[set][e] = v[e];
        } else if (K == 1) {
            const f32x4 v = mf_lds_ld4(lds + b_at);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[set][0][e] = v[e];
        } else {
            const int u = (k - 1) / 3;
            if ((k - 1) % 3 == 1) {                                 // columns x - 1 and x + 4 from the neighbouring aligned pairs: a
                bv[set][u][0] = mf_lds_ld2(lds + b_at + u * g.PS - 2)[1];      // ds_read2_b32 here was two 4-way bank conflicts
            } else if ((k - 1) % 3 == 2) {
                bv[set][u][5] = mf_lds_ld2(lds + b_at + u * g.PS + 4)[0];
            } else {
                const f32x4 v = mf_lds_ld4(lds + b_at + u * g.PS);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[set][u][e + 1] = v[e];
            }
        }
    };
    // One slot: 4 k-steps x KK taps of MFMAs on operand set `set` (if the slot has a group), and pinned behind the first
    // ones the memory instructions that go with it: the reads of the wave's next slot (addresses in a_at / b_at), the LDS
    // stores of staging piece cc into sbuf, the loads that refill its registers from stage fst.
    auto memory_op = [&](int set, int k, int sbuf, int cc, int fst) {
        if (k < NRD) read_op(set ^ 1, k);
        if (k == NRD) stage_piece(sbuf, cc);
        if (k == NRD + NWR) fetch_piece(fst, cc);
    };
    auto slot = [&](bool valid, int set, int sbuf, int cc, int fst) {
        if (valid && KK > 1) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) {
                    const int u = tap / K, v = tap - u * K;
                    acc[tap] = mfma16(av[set][m], bv[set][u][m + v], acc[tap]);
                    __builtin_amdgcn_sched_barrier(0);
                    memory_op(set, m * KK + tap, sbuf, cc, fst);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            if (valid) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[0] = mfma16(av[set][m], bv[set][0][m], acc[0]);
            }
#pragma unroll
            for (int k = 0; k <= NRD + NWR; ++k) memory_op(set, k, sbuf, cc, fst);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nst > 0) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) fetch_piece(0, cc);
        __syncthreads();                                            // zero fill done
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) stage_piece(0, cc);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) fetch_piece(1, cc);
        stage_piece(1, 0);
        fetch_piece(2, 0);
        __syncthreads();
        slot_address(0, 0, 0);
#pragma unroll
        for (int k = 0; k < NRD; ++k) read_op(0, k);
        __builtin_amdgcn_sched_barrier(0);
    }
    // Registers hold pieces 1..3 of stage t+1 and piece 0 of stage t+2 when stage t starts (piece 0 of stage t+1 is in the
    // other buffer already).  Slots 0..2 write pieces 1..3 of stage t+1 into the other buffer; then the barrier; slot 3
    // reads the first operands of stage t+1 there and writes piece 0 of stage t+2 into this buffer, which nobody reads any
    // more.  (Two stages per iteration so that the buffer index is a literal.)
    for (int st = 0; st < nst; st += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = st + h;
            if (h == 1 && t >= nst) break;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                if (sl == 3) __syncthreads();
                if (sl < 3) slot_address(h, t, sl + 1);
                else slot_address(h ^ 1, t + 1, 0);
                if (sl < 3) slot(slot_valid(t, sl), sl & 1, h ^ 1, sl + 1, t + 2);
                else slot(slot_valid(t, sl), sl & 1, h, 0, t + 3);
            }
        }
    }
    // ---- cross-wave reduction, fixed order
    __syncthreads();
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
    int CS, CS2, RS2, spg, NSG, bps;
    size_t corr_lds, adj_lds;
};

static int mf_pad_to(int v, int mod, int res) {          // smallest x >= v with x % mod == res
    int x = v;
    while ((x % mod) != res) ++x;
    return x;
}

static MfPlan mf_plan(int n, int F, int C, int H, int W, int K, int br_force = 0, int npg = 4) {   // br_force: rows per correlation band; npg: 64-position groups per band
    MfPlan p;
    p.ok = 0;
    if (n <= 0 || F <= 0 || F > 16 || C <= 0 || H <= 0 || W <= 0) return p;
    if (K != 1 && K != 3) return p;
    if (W > 256 || (long)n * C * H * W >= (1L << 31) || (long)C * H * W >= (1L << 30) || (long)F * C * K * K >= (1L << 30)) return p;
    if (K == 1 && ((H * W) % 4) == 0) {
        // 1x1: the map is a flat list of H*W positions.  Count it on rows whose length is a multiple of 4 (float4 staging
        // and stores; 18x18 -> 9 rows of 36), the one that fills the 256-position bands best.
        const int HW = H * W;
        int best = 0, fill = -1;
        for (int d = 4; d <= 256 && d <= HW; d += 4) {
            if (HW % d) continue;
            const int br = std::min(HW / d, 256 / d), fl = br * d;
            if (fl >= fill) { fill = fl; best = d; }
        }
        if (best) { W = best; H = HW / best; }
    }
    MfGeom& g = p.g;
    g.n = n; g.F = F; g.C = C; g.H = H; g.W = W; g.K = K; g.p = K / 2; g.KK = K * K;
    g.PWs = W + K - 1;
    // correlation band: <= 256 positions counted on rows of Wp (4 consecutive positions per lane) and <= 512 staged floats
    // per channel; a small problem gets shorter bands until the grid covers the CUs, as long as half the lanes stay busy
    // Correlation rows in LDS are LINEAR: row stride PS == Wp, data at column x, and Wp >= W + 1 for a 3x3 filter, so that the zero
    // tail of a row is the right padding of that row and the left padding of the next.  The 4 positions of lane (kq, j) then
    // start at dword 4 + 64 pg + 4 j of the channel plane -- 16 consecutive lanes cover 64 consecutive dwords -- and with a
    // plane stride CS == 0 (mod 64) every 16-lane group of a ds_read_b128 (which mixes the lanes of two kq, see
    // MI355X_MICROARCH.md section LDS) touches each of the 64 banks once.  Round 5 counters on the old layout (rows of Wp + 4,
    // CS = 484, neighbours as ds_read_b32): 65 % of the LDS-active cycles were bank conflicts (profiles/r05u_head_pmc.txt).
    const int Wpa = (W + 3) & ~3;                                   // adjoint: rows of roundup4(W), stride + 4 (unchanged)
    g.Wp = K == 1 ? Wpa : ((W + 1 + 3) & ~3);
    g.PS = g.Wp;
    g.out_vec = 0;
    int BR = 64 * npg / g.Wp;
    if (BR > H) BR = H;
    while (BR > 1 && (BR + K - 1) * W > 64 * MF_NQ) --BR;            // what fits
    if (br_force > 0) {
        if (br_force > BR) return p;
        BR = br_force;
    } else {
        while (BR > 1 && n * ((H + BR - 1) / BR) < 256 && (BR - 1) * g.Wp >= 128) --BR;
    }
    if (BR < 1 || (BR + K - 1) * W > 64 * MF_NQ || BR * g.Wp > 64 * npg) return p;
    g.BR = BR;
    g.NB = (H + BR - 1) / BR;
    g.RSmax = BR + K - 1;
    // adjoint band: <= 256 positions on rows of Wp, <= 512 staged feature floats per channel
    p.ga = g;
    p.ga.Wp = Wpa;
    p.ga.PS = Wpa + 4;
    int BRa = 256 / Wpa;
    if (BRa > H) BRa = H;
    while (BRa > 1 && (BRa + K - 1) * W > 64 * MF_NQ) --BRa;
    if (BRa < 1 || (BRa + K - 1) * W > 64 * MF_NQ) return p;
    p.ga.BR = BRa;
    p.ga.NB = (H + BRa - 1) / BRa;
    p.ga.RSmax = BRa + K - 1;
    p.CS = mf_pad_to(4 + g.RSmax * g.PS + 4, 64, 0);    // + 4: the right-hand quad of the last staged row
    // adjoint: lane (kq, j) reads row j at a column that depends on kq only; strides == 8 (mod 16) put the 16 rows of every
    // ds_read_b128 lane group on 16 different bank quads (== 4 mod 8, the round-2 choice, was a 2-way conflict on this chip's groups)
    p.CS2 = mf_pad_to(4 + p.ga.RSmax * p.ga.PS + 4, 16, 8);
    p.RS2 = mf_pad_to((BRa * Wpa + 15) & ~15, 16, 8);   // whole groups of 16 positions; the tail of a row stays zero
    p.corr_lds = std::max(2 * ((size_t)MF_CK * p.CS + MF_KS * 64 * MF_TP(g.KK)) + 4 * 128 * npg, (size_t)1024 * npg) * sizeof(float);   // two buffers + dump slots | k-half sum
    p.adj_lds = std::max((size_t)2 * 16 * (p.CS2 + p.RS2) + 4 * 256, (size_t)4 * g.KK * 256) * sizeof(float);   // two buffers + dump slots | reduction
    const int CBn = (C + 15) / 16;
    int NSG = 512 / CBn;                                 // ~2 workgroups per CU
    if (NSG < 1) NSG = 1;
    if (NSG > n) NSG = n;
    if (NSG > 32) NSG = 32;
    p.spg = (n + NSG - 1) / NSG;
    p.NSG = (n + p.spg - 1) / p.spg;
    p.bps = 1;
    // few samples: a sample's bands over several workgroups -- up to two 256-thread workgroups per CU (with one, every SIMD holds a
    // single wave and nothing overlaps its MFMAs)
    static const int adj_target = [] { const char* e = getenv("PT_MF_ADJ_WGS"); return e ? atoi(e) : 512; }();
    if (p.spg == 1 && (long)n * CBn < adj_target && p.ga.NB > 1) {
        int bps = std::min(p.ga.NB, (int)((adj_target + (long)n * CBn - 1) / ((long)n * CBn)));
        while (bps > 1 && n * bps > 32) --bps;
        const int bpg = (p.ga.NB + bps - 1) / bps;
        bps = (p.ga.NB + bpg - 1) / bpg;                            // no empty share
        if (bps > 1) { p.bps = bps; p.NSG = n * bps; }
    }
    if (p.corr_lds > 150 * 1024 || p.adj_lds > 150 * 1024) return p;
    p.ok = 1;
    return p;
}

size_t pt_mf_wt_floats(int C, int K) { return (size_t)(((C + MF_CK - 1) / MF_CK) * MF_CK / 4) * 64 * MF_TP(K * K); }

int pt_launch_mf_wtrans(const float* filt, float* wT, int F, int C, int K, hipStream_t st, int groups, float* clear2) {
    const int Cpad = ((C + MF_CK - 1) / MF_CK) * MF_CK, total = (Cpad >> 2) * 64 * MF_TP(K * K);
    hipLaunchKernelGGL(k_mf_wtrans, dim3((total + 255) / 256, groups), dim3(256), 0, st, filt, wT, F, C, K * K, Cpad,
                       (long)F * C * K * K, (long)pt_mf_wt_floats(C, K), clear2);
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
// widest staging item: 4 floats, or 2 for an even map width such as 18 (rows and planes 8-byte aligned), else 1
static int mf_vec_width(const float* a, long stride_n, int H, int W) {
    if (mf_vec_ok(a, a, stride_n, W)) return 4;
    if ((W % 2) == 0 && (stride_n % 2) == 0 && ((uintptr_t)a % 8) == 0) return 2;
    return 1;
}

// no LDS staging for a single tap; as many banks per workgroup (shared B registers) as keep the grid >= 256
static int mf_launch_corr1(const float* feat, long stride_n, const float* w, bool direct, float* scores, long out_stride_n,
                           int n, int F, int C, int HW, int groups, long wt_zs, long out_zs, int out_vec, hipStream_t st) {
    const int gx = (HW + 63) / 64;
    int nbk = groups >= 4 ? 4 : groups >= 2 ? 2 : 1;
    while (nbk > 1 && (long)gx * n * ((groups + nbk - 1) / nbk) < 256) nbk >>= 1;
    dim3 g1(gx, n, (groups + nbk - 1) / nbk);
#define PT_MFC1(NBKV, DV) \
    hipLaunchKernelGGL((k_mf_corr1<NBKV, DV>), g1, dim3(256), 0, st, feat, stride_n, w, scores, out_stride_n, F, C, HW, groups, wt_zs, out_zs, out_vec)
    if (direct) { if (nbk == 4) PT_MFC1(4, true); else if (nbk == 2) PT_MFC1(2, true); else PT_MFC1(1, true); }
    else { if (nbk == 4) PT_MFC1(4, false); else if (nbk == 2) PT_MFC1(2, false); else PT_MFC1(1, false); }
#undef PT_MFC1
    PT_CHECK_LAUNCH();
    return PT_OK;
}

// conv1x1 with the weights read from the (Ftot, C) matrix itself, Ftot = 16 * groups rows (no pt_launch_mf_wtrans in front).
// PT_ERR_UNSUPPORTED when the shapes / alignments do not allow it: the caller falls back to wtrans + pt_launch_mf_corr.
int pt_launch_mf_corr1_direct(const float* feat, long stride_n, const float* filt, float* scores, int n, int Ftot, int C,
                              int H, int W, hipStream_t st, long out_stride_n) {
    const int HW = H * W;
    if (n <= 0 || Ftot <= 0 || Ftot % 16 || C % 16 || HW % 4 || stride_n % 4 || out_stride_n % 4) return PT_ERR_UNSUPPORTED;
    if (((uintptr_t)feat % 16) || ((uintptr_t)filt % 16) || ((uintptr_t)scores % 16) || (long)C * HW >= (1L << 30)) return PT_ERR_UNSUPPORTED;
    return mf_launch_corr1(feat, stride_n, filt, true, scores, out_stride_n, n, 16, C, HW, Ftot / 16, 0, (long)16 * HW, 1, st);
}

// Few samples leave the 3x3 correlation too few bands for the chip (n = 1: the test frame of the few-shot learner, n < 16: the
// first frames of a sequence): with a partial-map workspace the channels are split over `pt_mf_corr_splits` workgroups per band
// and summed in fixed order.  1 = no split.
int pt_mf_corr_splits(int n, int F, int C, int H, int W, int K) {
    MfPlan p = mf_plan(n, F, C, H, W, K);
    if (!p.ok || K != 3) return 1;
    const int nch = (C + MF_CK - 1) / MF_CK;
    // The pass takes (rounds of workgroups over the 256 CUs) x (channel chunks per split) chunk-times plus a per-workgroup prologue /
    // epilogue worth about two chunks; n = 8 with the old power-of-two rule ran 320 workgroups = 2 rounds of 8 chunks where 240
    // workgroups of 11 chunks are one round.  Any split count up to nch / 2 may be chosen (PT_MF_KS_POW2=1: the old rule).
    static const bool pow2 = [] { const char* e = getenv("PT_MF_KS_POW2"); return e && e[0] == '1'; }();
    if (pow2) {
        int ks = 1;
        while (ks < 8 && 2 * ks <= nch / 2 && (long)n * p.g.NB * ks < 192) ks *= 2;
        while (ks > 1 && (ks - 1) * ((nch + ks - 1) / ks) >= nch) ks /= 2;     // every split owns at least one chunk
        return ks;
    }
    int best = 1;
    long best_cost = -1;
    for (int ks = 1; ks <= 16 && ks <= nch / 2; ++ks) {
        const int cps = (nch + ks - 1) / ks;
        if ((ks - 1) * cps >= nch) continue;                                   // every split owns at least one chunk
        const long wgs = (long)n * p.g.NB * ks, rounds = (wgs + 255) / 256;
        const long cost = rounds * (cps + 2) * 16 + ks;                        // + ks: the partial maps are written and summed
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ks; }
    }
    return best;
}
size_t pt_mf_corr_part_floats(int n, int F, int C, int H, int W, int K) {
    const int ks = pt_mf_corr_splits(n, F, C, H, W, K);
    return ks > 1 ? (size_t)ks * n * F * H * W : 0;
}
// sum of the <= 16 channel-split partial maps of an element quad: every load requested before the first add (with the run-time
// count as the loop bound each split was its own memory round trip), fixed order
template <int NB>
__device__ __forceinline__ f32x4 mf_parts_batch(const float* __restrict__ part, int parts, long count, long e) {
    f32x4 v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = *(const f32x4*)(part + (long)min(k, parts - 1) * count + e);
    f32x4 s = v[0];
#pragma unroll
    for (int k = 1; k < NB; ++k)
        if (k < parts) s += v[k];
    return s;
}
__device__ __forceinline__ f32x4 mf_parts_quad(const float* __restrict__ part, int parts, long count, long e) {
    if (parts <= 4) return mf_parts_batch<4>(part, parts, count, e);       // uniform: no more clamped re-reads than the next batch size
    if (parts <= 8) return mf_parts_batch<8>(part, parts, count, e);
    return mf_parts_batch<16>(part, parts, count, e);
}
__global__ void k_mf_sum_parts(const float* __restrict__ part, float* __restrict__ out, int parts, long count) {
    const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= count) return;
    *(f32x4*)(out + e) = mf_parts_quad(part, parts, count, e);
}
// the same with the squared-norm rider (PtMfSq): PT_MF_SQ_PARTS workgroups stride over the quads, each leaves one partial sum
__global__ __launch_bounds__(256) void k_mf_sum_parts_sq(const float* __restrict__ part, float* __restrict__ out, int parts, long count,
                                                         PtMfSq q) {
    __shared__ float scratch[16];
    float acc = 0.f;
    for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < count; e += (long)PT_MF_SQ_PARTS * 256 * 4) {
        const f32x4 s = mf_parts_quad(part, parts, count, e);
        *(f32x4*)(out + e) = s;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float w = q.sw_mode == 0 ? q.sw_scalar : (q.sw_mode == 1 ? q.sw[(e + u) / q.per_image] : q.sw[e + u]);
            const float h = w * s[u];
            acc += h * h;
        }
    }
    const float tot = block_sum(acc, scratch);
    if (threadIdx.x == 0) q.out[blockIdx.x] = tot;
}

// wT: weights pre-transposed by pt_launch_mf_wtrans (pt_mf_wt_floats(C, K) floats, 16-byte aligned)
int pt_launch_mf_corr(const float* feat, long stride_n, const float* wT, float* scores, int n, int F, int C, int H,
                      int W, int K, hipStream_t st, long out_stride_n, int groups, float* part, const PtMfSq* sq) {
    MfPlan p = mf_plan(n, F, C, H, W, K);
    if (out_stride_n == 0) out_stride_n = (long)F * H * W;
    if (!p.ok || groups < 1) return PT_ERR_UNSUPPORTED;
    // channel splits into the caller's partial-map workspace (pt_mf_corr_part_floats), then the fixed-order sum
    const int ksp = (part && groups == 1 && out_stride_n == (long)F * H * W && ((F * H * W) % 4) == 0 &&
                     ((uintptr_t)part % 16) == 0 && ((uintptr_t)scores % 16) == 0) ? pt_mf_corr_splits(n, F, C, H, W, K) : 1;
    if (ksp > 16) return PT_ERR_UNSUPPORTED;                            // mf_parts_quad sums at most 16 splits: refused BEFORE anything is queued
    // groups > 1: `groups` banks of F filters each (weight tables back to back, outputs F*H*W apart inside a sample)
    const long wt_zs = (long)pt_mf_wt_floats(C, K), out_zs = (long)F * H * W;
    dim3 grid(p.g.NB, n * ksp, groups), block(MF_CT);
    const bool vec = mf_vec_ok(feat, feat, stride_n, p.g.W);            // the plan's W (a 1x1 map is re-rowed)
    float* dst = ksp > 1 ? part : scores;
    p.g.out_vec = vec && ((uintptr_t)dst % 16) == 0 && (out_stride_n % 4) == 0 && ((H * W) % 4) == 0;
#define PT_MFC(KKV, VWV) \
    hipLaunchKernelGGL((k_mf_corr<KKV, VWV, false>), grid, block, p.corr_lds, st, feat, stride_n, wT, dst, out_stride_n, p.g, p.CS, wt_zs, out_zs, ksp, (long)n * out_stride_n)
    if (K == 1 && vec) return mf_launch_corr1(feat, stride_n, wT, false, scores, out_stride_n, n, F, C, H * W, groups, wt_zs, out_zs, p.g.out_vec, st);
    const int vw = mf_vec_width(feat, stride_n, p.g.H, p.g.W);
    if (K == 1) { PT_MFC(1, 1); }
    else { if (vw == 4) PT_MFC(9, 4); else if (vw == 2) PT_MFC(9, 2); else PT_MFC(9, 1); }
#undef PT_MFC
    PT_CHECK_LAUNCH();
    if (ksp > 1) {
        const long count = (long)n * out_stride_n;
        if (sq && sq->out && sq->done) {
            hipLaunchKernelGGL(k_mf_sum_parts_sq, dim3(PT_MF_SQ_PARTS), dim3(256), 0, st, part, scores, ksp, count, *sq);
            *sq->done = 1;
        } else {
            hipLaunchKernelGGL(k_mf_sum_parts, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, st, part, scores, ksp, count);
        }
        PT_CHECK_LAUNCH();
    }
    return PT_OK;
}

// Band height, position groups (8 or 12 waves) and channel splits for a bank of filters on few samples: the cheapest of
// (rounds of 256 workgroups) x (chunks per workgroup + ~4 chunks of fixed cost) x (waves / 8), balanced bands on a tie.
static int mf_tm_config(int n, int Ftot, int C, int H, int W, int* br_out, int* npg_out, long* cost_out = nullptr) {
    if (Ftot <= 0 || Ftot % 16 || C <= 0 || C % 16 || H <= 0 || W <= 0) return 0;
    const int Wp = (W + 1 + 3) & ~3, nch = C / MF_CK, groups = Ftot / 16;   // rows of the 3x3 correlation in LDS (mf_plan)
    int best_ks = 0, best_br = 0, best_npg = 4;
    long best_cost = -1;
    for (int npg = 4; npg <= ((W % 2) == 0 ? 6 : 4); npg += 2) {    // a chunk costs a SIMD npg/4 as much; 12 waves: 8-byte staging items only (registers)
        for (int br = std::min(H, 64 * npg / std::max(Wp, 1)); br >= 1; --br) {
            if ((br + 2) * W > 64 * MF_NQ) continue;
            const int nb = (H + br - 1) / br;
            if (br != (H + nb - 1) / nb) continue;                  // only the balanced height of each band count
            for (int ks = 1; ks <= 8 && ks <= nch; ks *= 2) {
                if (ks > 1 && (ks - 1) * ((nch + ks - 1) / ks) >= nch) continue;    // every split owns at least one chunk
                const long wgs = (long)n * nb * groups * ks, cost = ((wgs + 255) / 256) * ((nch + ks - 1) / ks + 4) * npg;
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_ks = ks; best_br = br; best_npg = npg; }
            }
        }
    }
    if (br_out) *br_out = best_br;
    if (npg_out) *npg_out = best_npg;
    if (cost_out) *cost_out = best_cost;
    return best_ks;
}

// 3x3 correlation with a bank of Ftot = 16 * groups filters kept tap-major, (Ftot, 9, C), split `ksplit` ways over the
// channels: partial maps part[s][n][Ftot][H][W] (the caller sums them in fixed order).  pt_mf_corr_tm_splits() says how
// many splits this launcher will use (0: shape not covered).
int pt_mf_corr_tm_splits(int n, int Ftot, int C, int H, int W) {
    int br = 0, npg = 4;
    const int ks = mf_tm_config(n, Ftot, C, H, W, &br, &npg);
    if (ks == 0 || br == 0 || !mf_plan(n, 16, C, H, W, 3, br, npg).ok) return 0;
    return ks;
}
// what the shape costs on this launcher, in chunk times of the 8-wave kernel (~2.7 us each on an 18x18 map), fixed part included
int pt_mf_corr_tm_cost(int n, int Ftot, int C, int H, int W) {
    int br = 0, npg = 4;
    long cost = 0;
    const int ks = mf_tm_config(n, Ftot, C, H, W, &br, &npg, &cost);
    if (ks == 0 || br == 0) return 1 << 30;
    return (int)(cost / 4);
}
int pt_launch_mf_corr_tm(const float* feat, long stride_n, const float* w_tap_major, float* part, int n, int Ftot, int C,
                         int H, int W, int ksplit, hipStream_t st) {
    int br = 0, npg = 4;
    if (mf_tm_config(n, Ftot, C, H, W, &br, &npg) != ksplit || ksplit < 1) return PT_ERR_UNSUPPORTED;
    if (((uintptr_t)w_tap_major % 16) || ((uintptr_t)part % 16) || (long)Ftot * 9 * C >= (1L << 30)) return PT_ERR_UNSUPPORTED;
    const int groups = Ftot / 16;
    MfPlan p = mf_plan(n, 16, C, H, W, 3, br, npg);
    if (!p.ok) return PT_ERR_UNSUPPORTED;
    const long out_stride_n = (long)Ftot * H * W, wt_zs = (long)16 * 9 * C, out_zs = (long)16 * H * W;
    dim3 grid(p.g.NB, n * ksplit, groups), block(128 * npg);
    const int vw = mf_vec_width(feat, stride_n, H, W);
    p.g.out_vec = vw == 4 && ((H * W) % 4) == 0;
#define PT_MFT(VWV, NPGV) \
    hipLaunchKernelGGL((k_mf_corr<9, VWV, true, NPGV>), grid, block, p.corr_lds, st, feat, stride_n, w_tap_major, part, out_stride_n, p.g, p.CS, wt_zs, out_zs, ksplit, (long)n * out_stride_n)
    if (npg == 6 && vw == 1) return PT_ERR_UNSUPPORTED;             // scalar staging at 3 waves per SIMD would spill (the caller has a GEMM route)
    if (npg == 6) { if (vw == 4) PT_MFT(4, 6); else PT_MFT(2, 6); }
    else { if (vw == 4) PT_MFT(4, 4); else if (vw == 2) PT_MFT(2, 4); else PT_MFT(1, 4); }
#undef PT_MFT
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
    const bool vec = mf_vec_ok(feat, inp, stride_n, p.ga.W) && ((H * W) % 4) == 0 && (inp_stride_n % 4) == 0;
    const bool vec2 = !vec && (p.ga.W % 2) == 0 && (stride_n % 2) == 0 && (inp_stride_n % 2) == 0 &&
                      ((uintptr_t)feat % 8) == 0 && ((uintptr_t)inp % 8) == 0;
#define PT_MFA(KKV, VWV) \
    hipLaunchKernelGGL((k_mf_adj<KKV, VWV>), grid, block, p.adj_lds, st, feat, stride_n, inp, inp_stride_n, gpart, p.ga, p.CS2, p.RS2, p.spg, p.bps, inp_zs, gp_zs)
    if (K == 1) { if (vec) PT_MFA(1, 4); else PT_MFA(1, 1); }
    else { if (vec) PT_MFA(9, 4); else if (vec2) PT_MFA(9, 2); else PT_MFA(9, 1); }
#undef PT_MFA
    PT_CHECK_LAUNCH();
    return PT_OK;
}

