// fp32 NT GEMM on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) shared by the ToMP predictor, the
// classification-feature head (tomp.hip) and the IoU-guided box refinement (iou_refine.hip).
//   C = A W^T (+bias, affine, ReLU, residual, exp), both operands K-contiguous; LDS-staged tiles; optional "+pos" on the
//   A operand of the leading column blocks; optional 3x3 zero-padded gather on the A operand (implicit GEMM); split-K.
// Header-only (anonymous namespace): each translation unit gets its own instantiations.
#pragma once
#include "common.h"
#include "pt_internal.h"

#include <algorithm>
#include <stdlib.h>

namespace {

constexpr unsigned OOB = 0xFFFFFFF0u;      // raw buffer loads past num_records return 0
typedef unsigned pt_u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// generic NT GEMM on MFMA
// ------------------------------------------------------------------------------------------------------------------
#define GEMM_GN_MAX_SLICES 256
struct GemmArgs {
    const float* A; long lda; unsigned a_bytes;
    const float* Wt; unsigned w_bytes;          // (N, K) row-major
    int M, N, K;
    const float* bias; const float* scale; const float* shift;
    const float* R; float* C; long ldc;         // residual shares ldc and the row map of C
    int c_seg; long c_segstride;                // C/R row of logical row r: (r / c_seg) * c_segstride + r % c_seg
    int relu, expo, nchw;                       // nchw: C[((r / HW) * N + n) * HW + r % HW]
    const float* pos; unsigned pos_bytes; int pos_cols, L, HW;   // A[r][k] + pos[(r % L) % HW][k] for column tiles < pos_cols
    int H, Wd, Cin;                             // MODE 1: 3x3 zero-padded gather, K = 9 * Cin, weights (N, tap, Cin)
    // MODE 1, optional: the gathered operand is relu(GroupNorm(1, Cin)(A)) of the PREVIOUS layer's raw sums, normalised on its way
    // into LDS -- A holds the un-normalised activations, gn_stats the per-slice (sum, sum of squares) pairs of k_gn_reduce
    // (gn_slices <= GEMM_GN_MAX_SLICES per image of gn_count values), gn_gam / gn_bet the affine parameters.  Zero padding stays zero.
    const float* gn_stats; int gn_slices, gn_count; const float* gn_gam; const float* gn_bet;
    int swizzle;                                // XCD-aware workgroup -> tile map (grid.y rounded up to a multiple of 8)
    int ksteps; long c_zstride;                 // split-K: blockIdx.z owns K-steps [z*ksteps, (z+1)*ksteps) and writes its
                                                // partial product to C + z*c_zstride (bias on z = 0 only); 0 = no split
    int batch; long a_zstride, w_zstride;       // batch > 0 (no split-K): blockIdx.z is an independent problem, operands
                                                // a_zstride / w_zstride floats apart, output at C + z*c_zstride
    // k_gemm_ps<.., LNA> (round 6): the A operand is LayerNorm(A) over its K = 256 columns (eps 1e-5, affine ln_gam / ln_bet), formed
    // in registers on the way into LDS; the column-tile-0 workgroups also store the normalised rows to ln_out (row stride K)
    const float* ln_gam; const float* ln_bet; float* ln_out;
};

// ------------------------------------------------------------------------------------------------------------------
// epilogue shared by the GEMM kernels: accumulator tile (MT x NT fragments of 16x16, rows / columns permuted by prow) of
// the wave whose first row / column is (mrow0, ncol0)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gemm_prow(int i) { return (i >= 4 && i < 12) ? 2 * (i - 4) : (i < 4 ? 2 * i + 1 : 2 * i - 15); }

template <int MT, int NT>
__device__ __forceinline__ void gemm_store(const GemmArgs& g, f32x4 (&acc)[MT][NT], int mrow0, int ncol0, int lane, int bz) {
    auto prow = [](int i) { return gemm_prow(i); };
    // epilogue: straight-line.  The row part of every output address is computed once per accumulator row (the segment /
    // NCHW maps cost an integer division each) as a 32-bit BYTE offset for raw buffer accesses: rows / columns outside the
    // matrix get an out-of-range offset, which the hardware drops (stores) or answers with 0 (loads) -- no per-element
    // branches, so the compiler keeps all residual loads and all stores in flight (with `if (row < M)` around every
    // element it put an s_waitcnt vmcnt(0) in front of every store: 0.23 us per element per lane, 18 us for a 128x128
    // tile).  exp() only under a workgroup-uniform branch.
    const __amdgpu_buffer_rsrc_t rsC = pt_rsrc(g.C, 0xFFFFFFE0u), rsR = pt_rsrc(g.R ? g.R : g.C, 0xFFFFFFE0u);
    unsigned rbase[MT][4];
    const bool plain = !g.nchw && g.c_segstride == 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mrow0 + mt * 16 + prow(4 * (lane >> 4) + r);
            const int rc = min(row, g.M - 1);
            long o;
            if (plain) {
                o = (long)rc * g.ldc;
            } else if (g.nchw) {
                const int img = rc / g.HW;
                o = (long)img * g.N * g.HW + (rc - img * g.HW);
            } else {
                const int sg = rc / g.c_seg;
                o = ((long)sg * g.c_segstride + (rc - sg * g.c_seg)) * g.ldc;
            }
            rbase[mt][r] = row < g.M ? (unsigned)(o * 4) : OOB;
        }
    const unsigned zoff = (unsigned)((long)bz * g.c_zstride * 4);
    const unsigned cstep = (unsigned)(g.nchw ? g.HW : 1) * 4u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = ncol0 + nt * 16 + prow(lane & 15);
        const bool cok = col < g.N;
        const int cc = cok ? col : 0;
        const float bv = (g.bias && (g.batch || bz == 0)) ? g.bias[cc] : 0.f;
        const float sc = g.scale ? g.scale[cc] : 1.f, sh = g.shift ? g.shift[cc] : 0.f;
        const unsigned coff = (unsigned)cc * cstep;
        unsigned off[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) off[mt][r] = (cok && rbase[mt][r] != OOB) ? rbase[mt][r] + coff : OOB;
        float v[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = (acc[mt][nt][r] + bv) * sc + sh;
                v[mt][r] = g.relu ? fmaxf(t, 0.f) : t;
            }
        if (g.R) {                                                   // uniform: residual loads all in flight together
            float res[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) res[mt][r] = pt_bload1(rsR, off[mt][r]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[mt][r] += res[mt][r];
        }
        if (g.expo) {                                                // uniform
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[mt][r] = expf(v[mt][r]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[mt][r]), rsC,
                                                      off[mt][r] == OOB ? OOB : off[mt][r] + zoff, 0, 0);
    }
}

// Epilogue for accumulators computed TRANSPOSED (the MFMA's operands swapped: accumulator rows <-> columns n of C, accumulator
// columns <-> rows m of C), plain row-major C only.  Lane (kq, j) then holds, for tile row prow(j), the four columns prow(4 kq + r) --
// {1,3,5,7} / {0,2,4,6} / {8,10,12,14} / {9,11,13,15} for kq = 0..3: the lane pairs (kq ^ 1, i.e. lane ^ 16) hold interleaved halves of
// eight consecutive columns.  Two lane exchanges per accumulator turn them into FOUR CONSECUTIVE columns per lane, which leave as one
// 16-byte store (gemm_store: sixteen 4-byte stores per 32 x 32 wave tile and lane; the FFN's first product writes 15.9 MB that way).
// bias / ReLU / split-K offset as gemm_store; no residual, no affine, no exp, no row maps (the launcher checks).
template <int MT, int NT>
__device__ __forceinline__ void gemm_store_t(const GemmArgs& g, f32x4 (&acc)[MT][NT], int mrow0, int ncol0, int lane, int bz) {
    const int kq = lane >> 4, j = lane & 15;
    const bool low = kq == 1 || kq == 2;                             // holds the EVEN columns of its eight: takes the lower four (partner: odd, upper four)
    const int quarter = kq == 1 ? 0 : (kq == 0 ? 1 : kq);            // which 4-column group of the 16 this lane ends up with
    const __amdgpu_buffer_rsrc_t rsC = pt_rsrc(g.C, 0xFFFFFFE0u);
    const unsigned zoff = (unsigned)((long)bz * g.c_zstride * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = mrow0 + mt * 16 + gemm_prow(j);
        const unsigned rbase = row < g.M ? (unsigned)((long)row * g.ldc * 4) : OOB;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 a = acc[mt][nt];
            // even-column holder (kq = 1, 2): keeps e0, e1, sends e2, e3, receives o0, o1; odd-column holder (kq = 0, 3): the mirror image
            const float s0 = low ? a[2] : a[0], s1 = low ? a[3] : a[1];
            const float r0 = __shfl_xor(s0, 16, 64), r1 = __shfl_xor(s1, 16, 64);
            f32x4 v = low ? f32x4{a[0], r0, a[1], r1} : f32x4{r0, a[2], r1, a[3]};
            const int col = ncol0 + nt * 16 + 4 * quarter;
            if (g.bias && (g.batch || bz == 0)) {
                const f32x4 bv = col + 3 < g.N ? *reinterpret_cast<const f32x4*>(g.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                v += bv;
            }
            if (g.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            const unsigned off = (rbase != OOB && col + 3 < g.N) ? rbase + (unsigned)col * 4u + zoff : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pt_u32x4, v), rsC, off, 0, 0);
        }
    }
}

// one workgroup tile of the problem `g`: (bxi, byi) tile coordinates in a grid of gx column tiles, bzi the split / batch index
template <int BM, int BN, int MODE, int BK = 64>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bxi, const int byi, const int bzi, const int gx) {
    // K advances in steps of 64 (one barrier per 4 MFMA sub-steps of 16): with 32-wide steps the counters showed the
    // wavefronts parked at s_waitcnt / s_barrier for a third of their life and the LDS round trip exposed twice per step
    // (profiles/r01i_tomp_pmc.txt).  Fragment reads of sub-step h+1 are issued before the MFMAs of sub-step h.
    constexpr int LS = BK + 4;                  // LDS row stride 68 / 36 words: 16-byte aligned, 4 mod 32 (see prow below)
    constexpr int RP = 1024 / BK;               // tile rows covered by one pass of the 256 loader threads
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 16, NT = WN / 16, AL = BM / RP, BL = BN / RP;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LS];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // Workgroup -> tile map.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), and each XCD
    // has its own L2: with the natural map and a column-tile count that is a multiple of 8, every XCD walks ALL row tiles
    // of A (8 x 16 MB over the fabric for the FFN's second GEMM).  Swizzled: XCD c owns the row tiles = c mod 8 and sweeps
    // the column tiles, so A crosses the fabric once and only the (small) weight matrix is replicated per XCD.
    int bx = bxi, by = byi;
    if (g.swizzle) {
        const int lin = byi * gx + bxi, xcd = lin & 7, j = lin >> 3;
        by = xcd + 8 * (j / gx);
        bx = j % gx;
        if (by * BM >= g.M) return;                              // padding rows of the rounded-up grid
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int lrow = tid / (BK / 4), lc4 = (tid % (BK / 4)) * 4;   // loader: BK/4 threads cover one row segment of BK floats
    // fused GroupNorm + ReLU on the gathered operand (MODE 1): mean / rstd of the <= 2 images this tile's rows belong to, from the
    // slice statistics (double precision, as the stand-alone k_gn_apply; tree order instead of its serial order)
    // MODE 2 = MODE 1 (3x3 gather) + the fused GroupNorm loader: its own instantiation, so that the plain gather kernel (the
    // classification-feature head over several frames, the tower's first layer) keeps the code it had (with the statistics path
    // as a run-time branch of MODE 1 that kernel was 6 us slower on the ToMP head, same-box A/B, round 4)
    constexpr bool GATHER = MODE >= 1;
    constexpr bool gn = MODE == 2;
    __shared__ double gn_red[MODE == 2 ? 8 : 1];
    __shared__ float gn_mr[4];
    const int img_lo = GATHER ? m0 / g.HW : 0;
    float gmean[2] = {0.f, 0.f}, grstd[2] = {1.f, 1.f};
    const long zb = g.batch ? (long)bzi : 0;
    const __amdgpu_buffer_rsrc_t rsA = pt_rsrc(g.A + zb * g.a_zstride, g.a_bytes), rsW = pt_rsrc(g.Wt + zb * g.w_zstride, g.w_bytes);
    const bool addpos = MODE == 0 && g.pos != nullptr && n0 < g.pos_cols;
    const __amdgpu_buffer_rsrc_t rsP = pt_rsrc(addpos ? g.pos : g.A, addpos ? g.pos_bytes : 16u);

    unsigned aoff[AL], poff[AL], woff[BL];
    int py[AL], px[AL], rimg[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int row = m0 + lrow + RP * i;
        const bool ok = row < g.M;
        if (MODE == 0) {
            aoff[i] = ok ? (unsigned)(((long)row * g.lda + lc4) * 4) : OOB;
            poff[i] = (addpos && ok) ? (unsigned)(((long)((row % g.L) % g.HW) * g.K + lc4) * 4) : OOB;
            rimg[i] = 0;
        } else {
            const int img = row / g.HW, p = row - img * g.HW;
            rimg[i] = min(max(img - img_lo, 0), 1);
            py[i] = ok ? p / g.Wd : -4;                                  // -4: every tap falls outside the map
            px[i] = p - (p / g.Wd) * g.Wd;
            aoff[i] = (unsigned)(((long)img * g.HW * g.lda + lc4) * 4);
            poff[i] = OOB;
        }
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int n = n0 + lrow + RP * i;
        woff[i] = n < g.N ? (unsigned)(((long)n * g.K + lc4) * 4) : OOB;
    }

    // Register ring of depth PD; LDS is double-buffered one step ahead.  The loads are issued unconditionally (`live` =
    // false turns a request into an out-of-range one: zeros, no memory touched) so that the compiler's s_waitcnt vmcnt
    // bookkeeping stays exact -- a fetch inside a branch makes it wait for ALL outstanding loads before the LDS store.
    constexpr int PD = 2;
    f32x4 ra[PD][AL], rp[PD][AL], rb[PD][BL];
    f32x4 rgam[PD], rbet[PD];                                           // fused GroupNorm: gamma / beta of the slot's 4 channels
    unsigned inm[PD] = {0u, 0u};                                        //                  which of its rows lie inside the map
    // Addressing without VALU work per load (the vector instructions of a K-step are issue time taken from the MFMAs:
    // experiments/mfma_issue.hip): the row part of an address is a loop-invariant VGPR (out-of-range for rows outside
    // the matrix), the K-step part goes into the scalar offset of the buffer instruction.  A step past the end re-fetches
    // the last one (never consumed); only a ragged last step (K % BK != 0) needs per-lane offsets, prepared once.
    const int nkt_all = (g.K + BK - 1) / BK;
    const bool tail_live = (nkt_all - 1) * BK + lc4 < g.K;          // this lane's quad of the last K-step is inside K
    unsigned aoffT[AL], poffT[AL], woffT[BL];
#pragma unroll
    for (int i = 0; i < AL; ++i) { aoffT[i] = tail_live ? aoff[i] : OOB; poffT[i] = tail_live ? poff[i] : OOB; }
#pragma unroll
    for (int i = 0; i < BL; ++i) woffT[i] = tail_live ? woff[i] : OOB;
    auto fetch = [&](int kb, int sl, bool live) {
        const int kc = min(kb, nkt_all - 1);
        const bool last = kc == nkt_all - 1;                            // uniform
        const unsigned kbytes = (unsigned)kc * (BK * 4u);
        (void)live;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < AL; ++i)
                ra[sl][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, last ? aoffT[i] : aoff[i], kbytes, 0));
            if (addpos) {
#pragma unroll
                for (int i = 0; i < AL; ++i)
                    rp[sl][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, last ? poffT[i] : poff[i], kbytes, 0));
            }
        } else {
            const int k0 = kc * BK, tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            unsigned m = 0u;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int y = py[i] + dy, x = px[i] + dx;
                const bool in = y >= 0 && y < g.H && x >= 0 && x < g.Wd;
                m |= in ? (1u << i) : 0u;
                ra[sl][i] = pt_bload4(rsA, in ? aoff[i] + (unsigned)(((long)(y * g.Wd + x) * g.lda + c0) * 4) : OOB);
            }
            inm[sl] = m;
            if (gn) {                                                    // uniform
                rgam[sl] = *reinterpret_cast<const f32x4*>(g.gn_gam + c0 + lc4);
                rbet[sl] = *reinterpret_cast<const f32x4*>(g.gn_bet + c0 + lc4);
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i)
            rb[sl][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, last ? woffT[i] : woff[i], kbytes, 0));
    };
    auto stash = [&](int sl, int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            f32x4 v = ra[sl][i];
            if (addpos) v += rp[sl][i];
            if (gn) {                                                    // uniform; the expression of k_gn_apply, padding stays 0
                const float mu = gmean[rimg[i]], rs = grstd[rimg[i]];
                const bool in = ((inm[sl] >> i) & 1u) != 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = in ? fmaxf((v[e] - mu) * rs * rgam[sl][e] + rbet[sl][e], 0.f) : 0.f;
            }
            *reinterpret_cast<f32x4*>(&As[buf][(lrow + RP * i) * LS + lc4]) = v;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + RP * i) * LS + lc4]) = rb[sl][i];
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS fragment reads without bank conflicts.  ds_read_b128 is serviced in four groups of 16 lanes over 64 banks,
    // and the groups are not lane-contiguous: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32, i.e. every group
    // mixes two k-slots (lane >> 4).  With a row stride of 4 mod 64 words the 16-byte bank group of lane (i, kq) is
    // (s * row(i) + quad(kq)) mod 16 with s odd; it is a permutation inside every hardware group when MFMA row i is
    // tile row prow(i) (even rows for i in 4..11, odd rows otherwise) and k-slot kq reads k-quad {0,2,1,3}[kq].  Both
    // operands use the same maps, so the k pairing is intact and the row / column permutation is undone in the epilogue.
    auto prow = [](int i) { return (i >= 4 && i < 12) ? 2 * (i - 4) : (i < 4 ? 2 * i + 1 : 2 * i - 15); };
    const int qoff = ((lane >> 4) & 1) * 2 + (lane >> 5);
    const int aso = (wm * WM + prow(lane & 15)) * LS + qoff * 4, bso = (wn * WN + prow(lane & 15)) * LS + qoff * 4;

    const int nkt = (g.K + BK - 1) / BK;
    constexpr int KSU = 64 / BK;                                         // g.ksteps counts 64-wide steps whatever BK is
    const int kb0 = g.ksteps ? bzi * g.ksteps * KSU : 0;
    const int nk = g.ksteps ? min(nkt, kb0 + g.ksteps * KSU) : nkt;
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) fetch(kb0 + sl, sl, kb0 + sl < nk);
    // (the statistics are only needed by the first LDS store: their round trip runs under the first operand loads)
    if (gn) {
        // one slice per thread (gn_slices <= 256 = the block), double-precision wave + block reduction: a serial sum by one
        // thread (the order of k_gn_apply) cost 3.5 us in front of every tile's first load (profiles/r04s_*)
        const int nimg = g.M / g.HW;
        const int img_hi = min((min(m0 + BM, g.M) - 1) / g.HW, nimg - 1);
        for (int im = 0; im <= img_hi - img_lo && im < 2; ++im) {                       // uniform; one image unless the tile straddles two
            const float* st = g.gn_stats + (long)(img_lo + im) * 2 * g.gn_slices;
            double sd = tid < g.gn_slices ? (double)st[2 * tid] : 0.0, qd = tid < g.gn_slices ? (double)st[2 * tid + 1] : 0.0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { sd += __shfl_xor(sd, off, 64); qd += __shfl_xor(qd, off, 64); }
            if (lane == 0) { gn_red[2 * wave] = sd; gn_red[2 * wave + 1] = qd; }
            __syncthreads();
            if (tid == 0) {
                const double s4 = (gn_red[0] + gn_red[2]) + (gn_red[4] + gn_red[6]), q4 = (gn_red[1] + gn_red[3]) + (gn_red[5] + gn_red[7]);
                const double mean_d = s4 / g.gn_count;
                gn_mr[2 * im] = (float)mean_d;
                gn_mr[2 * im + 1] = (float)(1.0 / sqrt(fmax(q4 / g.gn_count - mean_d * mean_d, 0.0) + 1e-5));
            }
            __syncthreads();
        }
        gmean[0] = gn_mr[0]; grstd[0] = gn_mr[1];
        if (img_hi > img_lo) { gmean[1] = gn_mr[2]; grstd[1] = gn_mr[3]; }
    }
    stash(0, 0);
    __syncthreads();
    for (int t0 = kb0; t0 < nk; t0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int kb = t0 + u;
            const int buf = u & 1;
            fetch(kb + PD, u, kb + PD < nk);                             // slot u (step kb) already sits in LDS[buf]
            if (kb < nk) {                                               // workgroup-uniform; LDS reads + MFMA only
                const float* as = &As[buf][aso];
                const float* bs = &Bs[buf][bso];
                // all fragment reads of the K-step are issued up front (straight-line, so the compiler's lgkmcnt waits
                // are exact and the MFMAs of sub-step h start as soon as ITS fragments have landed): one exposed LDS round
                // trip per K-step instead of one per 16-k sub-step
                f32x4 fa[BK / 16][MT], fb[BK / 16][NT];
#pragma unroll
                for (int hh = 0; hh < BK / 16; ++hh) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) fa[hh][mt] = *reinterpret_cast<const f32x4*>(as + mt * 16 * LS + hh * 16);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) fb[hh][nt] = *reinterpret_cast<const f32x4*>(bs + nt * 16 * LS + hh * 16);
                }
#pragma unroll
                for (int hh = 0; hh < BK / 16; ++hh)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = mfma16(fa[hh][mt][j], fb[hh][nt][j], acc[mt][nt]);
            }
            stash((u + 1) % PD, buf ^ 1);                               // step kb+1 (zeros past the end: never read)
            __syncthreads();
        }
    }

    gemm_store<MT, NT>(g, acc, m0 + wm * WM, n0 + wn * WN, lane, bzi);
}

template <int BM, int BN, int MODE, int BK = 64>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
    gemm_tile<BM, BN, MODE, BK>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// Two independent small problems in one launch (the IoU head's two FC layers, forward and backward: 4 launches -> 2 per
// refinement iteration; each is launch bound at M = 10): column tiles [0, nx0) belong to g0, the rest to g1; a workgroup
// outside its problem's own row / split range leaves.
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_gemm_pair(GemmArgs g0, GemmArgs g1, int nx0, int nx1) {
    const bool first = (int)blockIdx.x < nx0;
    const GemmArgs& g = first ? g0 : g1;
    const int nz = g.ksteps ? ((g.K + 63) / 64 + g.ksteps - 1) / g.ksteps : 1;
    if ((int)blockIdx.y * BM >= g.M || (int)blockIdx.z >= nz) return;
    if (first) gemm_tile<BM, BN, 0, 64>(g0, blockIdx.x, blockIdx.y, blockIdx.z, nx0);
    else gemm_tile<BM, BN, 0, 64>(g1, blockIdx.x - nx0, blockIdx.y, blockIdx.z, nx1);
}

// ------------------------------------------------------------------------------------------------------------------
// 128 x 128 x 32 tiles, 8 wavefronts (2 x 4, each 64 x 32 = 4 x 2 MFMA fragments), for the wide GEMMs (M, N in the
// thousands).  Built on what bounds the fp32 MFMA kernels here (experiments/mfma_issue.hip: nothing but SALU issues behind
// a v_mfma_f32_16x16x4_f32; phases in which all waves touch LDS / wait on a barrier at once are pure loss):
//   * 6 ds_read_b128 fragment reads per 32 MFMAs (4 A + 2 B fragments per 16-k sub-step), no address VALU in the loop (K
//     advances through the scalar offset of the buffer loads);
//   * two LDS stages; while stage s is multiplied, stage s+1 goes from registers into the other buffer and the loads of
//     stage s+2 are issued; ONE barrier per stage, in front of its second sub-step (by then every wave has read its last
//     fragments of the current buffer and written its share of the next), so that sub-step already reads stage s+1;
//   * every memory instruction pinned behind one of the sub-step's first MFMAs (sched_barrier; left alone the scheduler
//     sinks the reads to their first use);
//   * two waves per SIMD.
// MODE 0 only (plain A).  K % 32 == 0.  Same LDS layout / fragment permutation / epilogue as k_gemm.
// ------------------------------------------------------------------------------------------------------------------
constexpr int GB_LS = 36, GB_STAGE = 2 * 128 * GB_LS;               // floats per stage: A tile then W tile

__global__ __launch_bounds__(512) void k_gemm_big(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float gb_lds[];  // [2][GB_STAGE]
    constexpr int MT = 4, NT = 2, WM = 64, WN = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const long zb = g.batch ? (long)blockIdx.z : 0;
    const __amdgpu_buffer_rsrc_t rsA = pt_rsrc(g.A + zb * g.a_zstride, g.a_bytes), rsW = pt_rsrc(g.Wt + zb * g.w_zstride, g.w_bytes);
    // loader: thread t covers 16 bytes of rows t/8 and t/8 + 64 of both tiles
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    unsigned aoff[2], woff[2];
    int lds_at[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + lrow + 64 * i, n = n0 + lrow + 64 * i;
        aoff[i] = row < g.M ? (unsigned)(((long)row * g.lda + lc4) * 4) : OOB;
        woff[i] = n < g.N ? (unsigned)(((long)n * g.K + lc4) * 4) : OOB;
        lds_at[i] = (lrow + 64 * i) * GB_LS + lc4;
    }
    const int nst_all = g.K / 32;
    const int s0 = g.ksteps ? blockIdx.z * g.ksteps * 2 : 0;       // g.ksteps counts 64-wide steps
    const int nst = (g.ksteps ? min(nst_all, s0 + g.ksteps * 2) : nst_all) - s0;
    f32x4 ra[2], rb[2];
    auto fetch_one = [&](int st, int k) {                           // k = 0, 1: A rows; 2, 3: W rows.  Past the end: last stage
        const unsigned kbytes = (unsigned)(s0 + min(st, nst - 1)) * 128u;
        if (k < 2) ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[k], kbytes, 0));
        else rb[k - 2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, woff[k - 2], kbytes, 0));
    };
    auto stash_one = [&](int buf, int k) {
        float* base = gb_lds + buf * GB_STAGE;
        if (k < 2) *reinterpret_cast<f32x4*>(__builtin_assume_aligned(base + lds_at[k], 16)) = ra[k];
        else *reinterpret_cast<f32x4*>(__builtin_assume_aligned(base + 128 * GB_LS + lds_at[k - 2], 16)) = rb[k - 2];
    };
    // fragment addresses (k_gemm's conflict-free maps: MFMA row i <-> tile row prow(i), k-slot kq <-> k-quad {0,2,1,3}[kq])
    const int qoff = ((lane >> 4) & 1) * 2 + (lane >> 5);
    const int aso = (wm * WM + gemm_prow(lane & 15)) * GB_LS + qoff * 4;
    const int bso = 128 * GB_LS + (wn * WN + gemm_prow(lane & 15)) * GB_LS + qoff * 4;
    f32x4 fa[2][MT], fb[2][NT];
    auto frag_one = [&](int buf, int hh, int set, int k) {          // k < MT: A fragment k, else W fragment k - MT
        const float* base = gb_lds + buf * GB_STAGE + hh * 16;
        if (k < MT) fa[set][k] = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(base + aso + k * 16 * GB_LS, 16));
        else fb[set][k - MT] = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(base + bso + (k - MT) * 16 * GB_LS, 16));
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one 16-k sub-step on fragment set `set`; `mem(i)` is issued behind MFMA i and pinned there
    auto substep = [&](int set, auto&& mem) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = mfma16(fa[set][mt][j], fb[set][nt][j], acc[mt][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                    mem((j * MT + mt) * NT + nt);
                    __builtin_amdgcn_sched_barrier(0);
                }
    };
    if (nst > 0) {
        {   // stages 0 and 1 requested together: one exposed memory round trip in front of the first MFMA, not two
            f32x4 pa[2], pb[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) fetch_one(0, k);
#pragma unroll
            for (int k = 0; k < 2; ++k) { pa[k] = ra[k]; pb[k] = rb[k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) fetch_one(1, k);
            float* base = gb_lds;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                *reinterpret_cast<f32x4*>(__builtin_assume_aligned(base + lds_at[k], 16)) = pa[k];
                *reinterpret_cast<f32x4*>(__builtin_assume_aligned(base + 128 * GB_LS + lds_at[k], 16)) = pb[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MT + NT; ++k) frag_one(0, 0, 0, k);
        __builtin_amdgcn_sched_barrier(0);
        // registers hold stage s+1 when stage s starts
        for (int st = 0; st < nst; st += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = st + h;                               // buffer h
                if (h == 1 && s >= nst) break;
                substep(0, [&](int i) {
                    if (i < MT + NT) frag_one(h, 1, 1, i);                          // fragments of this stage's second sub-step
                    else if (i < MT + NT + 4) stash_one(h ^ 1, i - (MT + NT));      // stage s+1 -> the other buffer
                    else if (i < MT + NT + 8) fetch_one(s + 2, i - (MT + NT + 4));  // stage s+2 -> registers
                });
                __syncthreads();
                substep(1, [&](int i) {
                    if (i < MT + NT) frag_one(h ^ 1, 0, 0, i);                      // first fragments of stage s+1
                });
            }
        }
    }
    gemm_store<MT, NT>(g, acc, m0 + wm * WM, n0 + wn * WN, lane, blockIdx.z);
}

// ------------------------------------------------------------------------------------------------------------------
// k_gemm_big's pinned schedule on other tile shapes (round 6): WGM x WGN wavefronts, each MT x NT MFMA fragments, tile
// BM = 16 WGM MT by BN = 16 WGN NT, K stages of 32.  What it is for: a 128 x 128 workgroup of the ToMP FFN's first product
// (K = 256: 8 stages) spends as long in its prologue (two memory round trips) and epilogue (16384 result stores) as in its K loop,
// and with 256 such workgroups a CU holds ONE, so nothing hides them (24 us where the matrix pipe needs 13).  128 x 64 tiles
// (4 x 2 waves of 32 x 32) make 512 workgroups of 55 KB LDS and <= 128 VGPRs: TWO per CU, each other's prologue / epilogue under
// the other's MFMAs.  The same kernel with 4 K splits serves the second product (K = 2048: 256 workgroups x 16 stages).
// TG: the M tile index is blockIdx.x (workgroups are dealt to the XCDs round-robin in x: all column tiles / K splits of a row tile
// then share that XCD's L2 copy of the A slab -- the long-K product's A operand is 16 MB).
// MODE 0 only (plain A).  K % 32 == 0.  Same LDS layout / fragment permutation / epilogue as k_gemm.
// ------------------------------------------------------------------------------------------------------------------
// SW: accumulators transposed (MFMA operands swapped) and stored with 16-byte stores (gemm_store_t; plain row-major C, N % 4 == 0).
// LNA (round 6, ToMP encoder: norm1 folded into the FFN's first product): A := LayerNorm_K(A) with K == 256 == 8 stages.  The workgroup
// needs its whole A slab anyway (K is the full row), so the slab is requested UP FRONT into registers (NPA x 8 16-byte pieces per thread:
// the same bytes the staged loader would have fetched), the row statistics are formed across the 8 lanes that share a row (two passes:
// mean, then centred squares -- the LayerNorm kernel's formula), the pieces are normalised in place and go to LDS stage by stage where
// the staged loader's registers would have gone.  One launch and one round trip through memory less per encoder layer.
template <int WGM, int WGN, int MT, int NT, bool TG, bool SW = false, bool LNA = false>
__global__ __launch_bounds__(64 * WGM * WGN) void k_gemm_ps(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float gb_lds[];  // [2][STAGE]
    constexpr int NTHR = 64 * WGM * WGN, WM = 16 * MT, WN = 16 * NT, BM = WGM * WM, BN = WGN * WN, LS = GB_LS;
    constexpr int STAGE = (BM + BN) * LS, NP = (BM + BN) * 8 / NTHR, NPA = BM * 8 / NTHR;
    static_assert((BM + BN) * 8 % NTHR == 0 && BM * 8 % NTHR == 0, "whole loader pieces per thread, A / W split on a piece boundary");
    static_assert(MT + NT + 2 * NP <= 4 * MT * NT, "memory instructions of a stage fit behind its first sub-step's MFMAs");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WGN, wn = wave % WGN;
    const int m0 = (TG ? blockIdx.x : blockIdx.y) * BM, n0 = (TG ? blockIdx.y : blockIdx.x) * BN;
    const long zb = g.batch ? (long)blockIdx.z : 0;
    const __amdgpu_buffer_rsrc_t rsA = pt_rsrc(g.A + zb * g.a_zstride, g.a_bytes), rsW = pt_rsrc(g.Wt + zb * g.w_zstride, g.w_bytes);
    // "+pos" on the A operand of the leading column tiles (the q | k blocks of the attention's input projection), as k_gemm does it
    const bool addpos = g.pos != nullptr && n0 < g.pos_cols;        // workgroup-uniform
    const __amdgpu_buffer_rsrc_t rsP = pt_rsrc(addpos ? g.pos : g.A, addpos ? g.pos_bytes : 16u);
    // loader: piece p = tid + NTHR k covers 16 bytes (k-quad p & 7) of tile row p >> 3; rows [0, BM) are A, [BM, BM + BN) are W
    unsigned goff[NP], poff[NPA];
    int lds_at[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = tid + NTHR * k, r = p >> 3, lc4 = (p & 7) * 4;
        if (k < NPA) {
            const int row = m0 + r;
            goff[k] = row < g.M ? (unsigned)(((long)row * g.lda + lc4) * 4) : OOB;
            poff[k] = (addpos && row < g.M) ? (unsigned)(((long)((row % g.L) % g.HW) * g.K + lc4) * 4) : OOB;
        } else {
            const int n = n0 + r - BM;
            goff[k] = n < g.N ? (unsigned)(((long)n * g.K + lc4) * 4) : OOB;
        }
        lds_at[k] = r * LS + lc4;
    }
    constexpr int LN_ST = 8;                                        // LNA: K == 256, no split
    const int nst_all = LNA ? LN_ST : g.K / 32;
    const int s0 = (!LNA && g.ksteps) ? blockIdx.z * g.ksteps * 2 : 0;   // g.ksteps counts 64-wide steps
    const int nst = LNA ? LN_ST : (g.ksteps ? min(nst_all, s0 + g.ksteps * 2) : nst_all) - s0;
    f32x4 rr[NP], rp[NPA];
    f32x4 ra[LNA ? NPA : 1][LNA ? LN_ST : 1];                       // LNA: the A slab of this thread (row p >> 3, k-quad p & 7 of every stage)
    auto fetch_one = [&](int st, int k) {                           // past the end: the last stage again (never consumed)
        if (LNA && k < NPA) return;                                 // (compile-time k at every call site)
        const unsigned kbytes = (unsigned)(s0 + min(st, nst - 1)) * 128u;
        rr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k < NPA ? rsA : rsW, goff[k], kbytes, 0));
        if (k < NPA && addpos) rp[k < NPA ? k : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, poff[k < NPA ? k : 0], kbytes, 0));
    };
    auto stash_one = [&](int buf, int k, int st = 0) {              // st: the stage being stashed (LNA; compile time: its loops are unrolled)
        f32x4 v = rr[k];
        if (LNA && k < NPA) v = ra[LNA ? (k < NPA ? k : 0) : 0][LNA ? min(st, LN_ST - 1) : 0];
        if (k < NPA && addpos) v += rp[k < NPA ? k : 0];
        *reinterpret_cast<f32x4*>(__builtin_assume_aligned(gb_lds + buf * STAGE + lds_at[k], 16)) = v;
    };
    if constexpr (LNA) {
        float* __restrict__ lnp = gb_lds + 2 * STAGE;               // gamma[256] | beta[256]
        // ---- the slab, gamma / beta, then the first two W stages: everything requested before the first wait
#pragma unroll
        for (int k = 0; k < NPA; ++k)
#pragma unroll
            for (int st = 0; st < LN_ST; ++st)
                ra[k][st] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, goff[k], (unsigned)st * 128u, 0));
        f32x4 gbv = {0, 0, 0, 0};
        if (tid < 128) gbv = *reinterpret_cast<const f32x4*>((tid < 64 ? g.ln_gam : g.ln_bet) + 4 * (tid & 63));
        if (tid < 128) *reinterpret_cast<f32x4*>(__builtin_assume_aligned(lnp + 4 * tid, 16)) = gbv;
        // ---- row statistics: a row's 256 values sit in the 8 lanes tid & 7 (k-quad) x 8 stages
        float mean[NPA], rstd[NPA];
#pragma unroll
        for (int k = 0; k < NPA; ++k) {
            float sm = 0.f;
#pragma unroll
            for (int st = 0; st < LN_ST; ++st) sm += (ra[k][st][0] + ra[k][st][1]) + (ra[k][st][2] + ra[k][st][3]);
            sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
            mean[k] = sm / 256.0f;
            float q = 0.f;
#pragma unroll
            for (int st = 0; st < LN_ST; ++st)
#pragma unroll
                for (int e = 0; e < 4; ++e) q += (ra[k][st][e] - mean[k]) * (ra[k][st][e] - mean[k]);
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
            rstd[k] = rsqrtf(q / 256.0f + 1e-5f);
        }
        __syncthreads();                                            // gamma / beta staged
        const bool keep = (TG ? blockIdx.y : blockIdx.x) == 0 && g.ln_out != nullptr;   // uniform: column tile 0 stores the normalised rows
        const __amdgpu_buffer_rsrc_t rsO = pt_rsrc(g.ln_out ? g.ln_out : g.C, g.ln_out ? g.a_bytes : 16u);
#pragma unroll
        for (int st = 0; st < LN_ST; ++st) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(lnp + 32 * st + 4 * (tid & 7), 16));
            const f32x4 bv = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(lnp + 256 + 32 * st + 4 * (tid & 7), 16));
#pragma unroll
            for (int k = 0; k < NPA; ++k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[k][st][e] = (ra[k][st][e] - mean[k]) * rstd[k] * gv[e] + bv[e];
                if (keep) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pt_u32x4, ra[k][st]), rsO, goff[k], (unsigned)st * 128u, 0);
            }
        }
    }
    const int qoff = ((lane >> 4) & 1) * 2 + (lane >> 5);
    const int aso = (wm * WM + gemm_prow(lane & 15)) * LS + qoff * 4;
    const int bso = BM * LS + (wn * WN + gemm_prow(lane & 15)) * LS + qoff * 4;
    f32x4 fa[2][MT], fb[2][NT];
    auto frag_one = [&](int buf, int hh, int set, int k) {          // k < MT: A fragment k, else W fragment k - MT
        const float* base = gb_lds + buf * STAGE + hh * 16;
        if (k < MT) fa[set][k] = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(base + aso + k * 16 * LS, 16));
        else fb[set][k - MT] = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(base + bso + (k - MT) * 16 * LS, 16));
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto substep = [&](int set, auto&& mem) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = SW ? mfma16(fb[set][nt][j], fa[set][mt][j], acc[mt][nt]) : mfma16(fa[set][mt][j], fb[set][nt][j], acc[mt][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                    mem((j * MT + mt) * NT + nt);
                    __builtin_amdgcn_sched_barrier(0);
                }
    };
    if (nst > 0) {
        {   // stages 0 and 1 requested together: one exposed memory round trip in front of the first MFMA
            f32x4 p0[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) fetch_one(0, k);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                p0[k] = (LNA && k < NPA) ? ra[LNA ? (k < NPA ? k : 0) : 0][0] : rr[k];
                if (k < NPA && addpos) p0[k] += rp[k < NPA ? k : 0];
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) fetch_one(1, k);
#pragma unroll
            for (int k = 0; k < NP; ++k) *reinterpret_cast<f32x4*>(__builtin_assume_aligned(gb_lds + lds_at[k], 16)) = p0[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MT + NT; ++k) frag_one(0, 0, 0, k);
        __builtin_amdgcn_sched_barrier(0);
        auto stage_pair = [&](int st) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = st + h;                               // buffer h
                if (h == 1 && s >= nst) break;
                substep(0, [&](int i) {
                    if (i < MT + NT) frag_one(h, 1, 1, i);                          // fragments of this stage's second sub-step
                    else if (i < MT + NT + NP) stash_one(h ^ 1, i - (MT + NT), s + 1);   // stage s+1 -> the other buffer
                    else if (i < MT + NT + 2 * NP) fetch_one(s + 2, i - (MT + NT + NP));   // stage s+2 -> registers
                });
                __syncthreads();
                substep(1, [&](int i) {
                    if (i < MT + NT) frag_one(h ^ 1, 0, 0, i);                      // first fragments of stage s+1
                });
            }
        };
        if constexpr (LNA) {                                        // 8 stages, unrolled: the slab registers are indexed by the stage
#pragma unroll
            for (int st = 0; st < LN_ST; st += 2) stage_pair(st);
        } else {
            for (int st = 0; st < nst; st += 2) stage_pair(st);
        }
    }
    if (SW) gemm_store_t<MT, NT>(g, acc, m0 + wm * WM, n0 + wn * WN, lane, blockIdx.z);
    else gemm_store<MT, NT>(g, acc, m0 + wm * WM, n0 + wn * WN, lane, blockIdx.z);
}

GemmArgs gemm_args(const float* A, long lda, long a_rows, const float* Wt, int M, int N, int K, const float* bias,
                   float* C, long ldc) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.a_bytes = (unsigned)std::min<long>(a_rows * lda * 4, 0xFFFFFFE0L);
    g.Wt = Wt; g.w_bytes = (unsigned)((long)N * K * 4);
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.C = C; g.ldc = ldc;
    g.c_seg = M > 0 ? M : 1; g.c_segstride = 0; g.HW = 1; g.L = 1;
    return g;
}

// Workgroup tile by shape (experiments/gemm_tiles.hip, profiles/r02e_gemm_tiles.txt, r02m_gemm_tiles.txt; M = 1944 rows of the ToMP
// encoder).  Wide outputs (N >= 1024, many rows) go to k_gemm_big: its K loop runs at 91 % of the fp32 MFMA peak (ffn2 shape,
// one workgroup per 64 K-stages), but at K = 256 a workgroup has only 8 stages and prologue + epilogue weigh as much as they
// do for the small tiles: 24.1 vs 27.8 us on ffn1.  Below that:
// 32x32x64 everywhere except wide outputs (N >= 1024, many rows), where 64x64 tiles with 32-wide K steps halve the operand
// traffic per flop and two workgroups still fit a CU: FFN first GEMM 36.3 -> 28.0 us.  Larger tiles (128x64, 128x128) run
// their K steps at 67-74 % of the MFMA rate but leave too few workgroups at these sizes.
// experiment knob: smallest N that takes the 64 x 64 tile kernel (PT_GEMM_T64N, default 1024)
static int pt_gemm_tile64_min_n() {
    static const int v = [] { const char* e = getenv("PT_GEMM_T64N"); return e ? atoi(e) : 1024; }();
    return v;
}

// experiment knob: tile of long-K narrow-N products (PT_GEMM_LONGK: 0 = 32 x 32, 1 = 32 x 64, 2 = 64 x 32)
static int pt_gemm_longk_tile() {
    static const int v = [] { const char* e = getenv("PT_GEMM_LONGK"); return e ? atoi(e) : 0; }();
    return v;
}

// gemm_store_t's conditions: plain row-major C, 16-byte aligned rows and bias, nothing but bias / ReLU / split offset in the epilogue
// (PT_GEMM_WIDE_STORE=0: the 4-byte epilogue, A/B knob)
static bool wide_store_ok(const GemmArgs& g) {
    static const bool on = [] { const char* e = getenv("PT_GEMM_WIDE_STORE"); return !(e && e[0] == '0'); }();
    return on && !g.R && !g.scale && !g.shift && !g.expo && !g.nchw && g.c_segstride == 0 && g.N % 4 == 0 && g.ldc % 4 == 0 && g.c_zstride % 4 == 0 &&
           ((uintptr_t)g.C % 16) == 0 && (!g.bias || ((uintptr_t)g.bias % 16) == 0);
}

int launch_gemm(const GemmArgs& g, hipStream_t st, bool conv = false) {
    // K % 4: a loader thread fetches 4 consecutive k (16 bytes); quads past K are not requested at all (zeros)
    if (g.K % 4 != 0 || (conv && g.K % 32 != 0) || g.M <= 0 || g.N <= 0 || (g.batch && g.ksteps)) return PT_ERR_UNSUPPORTED;
    // the epilogue addresses C (and R) with 32-bit byte offsets through a raw buffer descriptor
    const int nz = g.batch ? g.batch : g.ksteps ? ((g.K + 63) / 64 + g.ksteps - 1) / g.ksteps : 1;
    const long rows = g.nchw ? (long)g.M * g.N : (g.c_segstride ? ((long)(g.M / g.c_seg) + 1) * g.c_segstride * g.ldc
                                                                   : (long)g.M * g.ldc);
    if ((rows + (long)(nz - 1) * g.c_zstride) * 4 >= 0xFFFFFFE0L) return PT_ERR_UNSUPPORTED;
    // round 6: 128 x 64 tiles, two workgroups per CU (PT_GEMM_PS=0: the round-5 routes), for
    //   * wide outputs at short K (FFN first product: N = 2048, K = 256 -> 512 workgroups of 8 stages)
    //   * split-K products the caller asked to be cut in >= 4 (FFN second product: 256 workgroups of 16 stages, M tiles on x)
    static const bool ps_on = [] { const char* e = getenv("PT_GEMM_PS"); return !(e && e[0] == '0'); }();
    static const bool ps_qkv = [] { const char* e = getenv("PT_GEMM_PS_QKV"); return !(e && e[0] == '0'); }();
    if (ps_on && ps_qkv && !conv && g.pos && g.M >= 1024 && g.K % 32 == 0 && !g.batch && !g.nchw && nz == 1 && g.N % 64 == 0 && g.pos_cols % 64 == 0 &&
        g.c_segstride == 0 && g.N >= 512) {
        // the attention's input projection (N = 3 D = 768, K = 256, "+pos" on the q | k column tiles): 64 x 64 tiles of 4 waves, 37 KB LDS,
        // up to four workgroups per CU; the 32 x 32 route took 16.2 us for 0.76 GFLOP (1464 workgroups = 1.45 rounds)
        constexpr size_t lds = 2 * (64 + 64) * GB_LS * sizeof(float);
        if (wide_store_ok(g)) hipLaunchKernelGGL((k_gemm_ps<2, 2, 2, 2, false, true>), dim3((g.N + 63) / 64, (g.M + 63) / 64, 1), dim3(256), lds, st, g);
        else hipLaunchKernelGGL((k_gemm_ps<2, 2, 2, 2, false>), dim3((g.N + 63) / 64, (g.M + 63) / 64, 1), dim3(256), lds, st, g);
        PT_CHECK_LAUNCH();
        return PT_OK;
    }
    if (g.ln_gam) {
        // LayerNorm on the A operand: served by the 128 x 64 pinned-schedule kernel only (never dropped silently)
        if (!(ps_on && !conv && g.K == 256 && g.lda == 256 && !g.pos && !g.batch && !g.nchw && nz == 1 && !g.ksteps && g.ln_bet && wide_store_ok(g) &&
              ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.ln_gam % 16) == 0 && ((uintptr_t)g.ln_bet % 16) == 0 && ((uintptr_t)g.ln_out % 16) == 0))
            return PT_ERR_UNSUPPORTED;
        constexpr size_t lds = (2 * (128 + 64) * GB_LS + 512) * sizeof(float);
        hipLaunchKernelGGL((k_gemm_ps<4, 2, 2, 2, false, true, true>), dim3((g.N + 63) / 64, (g.M + 127) / 128, 1), dim3(512), lds, st, g);
        PT_CHECK_LAUNCH();
        return PT_OK;
    }
    if (ps_on && !conv && g.M >= 1024 && g.K % 32 == 0 && !g.pos && !g.batch && !g.nchw && ((g.N >= 1024 && nz == 1) || (nz >= 4 && g.N % 64 == 0))) {
        constexpr size_t lds = 2 * (128 + 64) * GB_LS * sizeof(float);
        const bool sw = wide_store_ok(g);
        if (nz == 1 && sw) hipLaunchKernelGGL((k_gemm_ps<4, 2, 2, 2, false, true>), dim3((g.N + 63) / 64, (g.M + 127) / 128, 1), dim3(512), lds, st, g);
        else if (nz == 1) hipLaunchKernelGGL((k_gemm_ps<4, 2, 2, 2, false>), dim3((g.N + 63) / 64, (g.M + 127) / 128, 1), dim3(512), lds, st, g);
        else if (sw) hipLaunchKernelGGL((k_gemm_ps<4, 2, 2, 2, true, true>), dim3((g.M + 127) / 128, (g.N + 63) / 64, nz), dim3(512), lds, st, g);
        else hipLaunchKernelGGL((k_gemm_ps<4, 2, 2, 2, true>), dim3((g.M + 127) / 128, (g.N + 63) / 64, nz), dim3(512), lds, st, g);
        PT_CHECK_LAUNCH();
        return PT_OK;
    }
    if (!conv && g.N >= 1024 && g.M >= 1024 && g.K % 32 == 0 && !g.pos && !g.batch) {
        hipLaunchKernelGGL(k_gemm_big, dim3((g.N + 127) / 128, (g.M + 127) / 128, nz), dim3(512), 2 * GB_STAGE * sizeof(float),
                           st, g);
        PT_CHECK_LAUNCH();
        return PT_OK;
    }
    if (conv) {
        if (g.Cin % 64 != 0) return PT_ERR_UNSUPPORTED;
        if (g.gn_stats) {
            // the fused loader reads gn_gam / gn_bet as 16-byte vectors and divides by gn_count
            if (g.gn_slices > GEMM_GN_MAX_SLICES || g.gn_slices < 1 || !g.gn_gam || !g.gn_bet || g.HW < 32 || g.gn_count <= 0 ||
                ((uintptr_t)g.gn_gam % 16) != 0 || ((uintptr_t)g.gn_bet % 16) != 0)
                return PT_ERR_UNSUPPORTED;
            hipLaunchKernelGGL((k_gemm<32, 32, 2>), dim3((g.N + 31) / 32, (g.M + 31) / 32, nz), dim3(256), 0, st, g);
        } else {
            hipLaunchKernelGGL((k_gemm<32, 32, 1>), dim3((g.N + 31) / 32, (g.M + 31) / 32, nz), dim3(256), 0, st, g);
        }
    } else if (g.N >= pt_gemm_tile64_min_n() && g.M >= 1024 && nz == 1 && g.K % 32 == 0) {
        GemmArgs gs = g;
        const int gy = (g.M + 63) / 64;
        gs.swizzle = gy >= 16;
        hipLaunchKernelGGL((k_gemm<64, 64, 0, 32>), dim3((g.N + 63) / 64, gs.swizzle ? (gy + 7) / 8 * 8 : gy, 1), dim3(256), 0,
                           st, gs);
    } else if (pt_gemm_longk_tile() && !g.batch && nz == 1 && g.K >= 1024 && g.K % 32 == 0 && g.M >= 1024 && g.N >= 64) {
        // long-K, narrow-N products (the FFN's second GEMM: 1944 x 256 x 2048): 32 x 32 tiles re-read both operands through L2
        // 488 times; a 2:1 tile halves one operand's traffic at the same workgroup count per CU
        GemmArgs gs = g;
        if (pt_gemm_longk_tile() == 1) {
            const int gy = (g.M + 31) / 32;
            gs.swizzle = gy >= 16;
            hipLaunchKernelGGL((k_gemm<32, 64, 0>), dim3((g.N + 63) / 64, gs.swizzle ? (gy + 7) / 8 * 8 : gy, 1), dim3(256), 0, st, gs);
        } else {
            const int gy = (g.M + 63) / 64;
            gs.swizzle = gy >= 16;
            hipLaunchKernelGGL((k_gemm<64, 32, 0>), dim3((g.N + 31) / 32, gs.swizzle ? (gy + 7) / 8 * 8 : gy, 1), dim3(256), 0, st, gs);
        }
    } else {
        GemmArgs gs = g;
        const int gy = (g.M + 31) / 32;
        gs.swizzle = nz == 1 && gy >= 16;                            // worth it once every XCD gets >= 2 row tiles
        if (g.batch) gs.ksteps = 0;
        hipLaunchKernelGGL((k_gemm<32, 32, 0>), dim3((g.N + 31) / 32, gs.swizzle ? (gy + 7) / 8 * 8 : gy, nz), dim3(256), 0,
                           st, gs);
    }
    PT_CHECK_LAUNCH();
    return PT_OK;
}

// two plain GEMMs (no swizzle, no conv, no batch) as one launch of 32 x 32 tiles
int launch_gemm_pair(const GemmArgs& g0, const GemmArgs& g1, hipStream_t st) {
    for (const GemmArgs* g : {&g0, &g1})
        if (g->K % 4 != 0 || g->M <= 0 || g->N <= 0 || g->batch || g->pos || g->swizzle) return PT_ERR_UNSUPPORTED;
    auto nzf = [](const GemmArgs& g) { return g.ksteps ? ((g.K + 63) / 64 + g.ksteps - 1) / g.ksteps : 1; };
    const int nx0 = (g0.N + 31) / 32, nx1 = (g1.N + 31) / 32;
    const int gy = std::max((g0.M + 31) / 32, (g1.M + 31) / 32), nz = std::max(nzf(g0), nzf(g1));
    hipLaunchKernelGGL((k_gemm_pair<32, 32>), dim3(nx0 + nx1, gy, nz), dim3(256), 0, st, g0, g1, nx0, nx1);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

}  // namespace
