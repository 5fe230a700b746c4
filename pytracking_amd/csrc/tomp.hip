// ToMP transformer model predictor on gfx950 (BASELINE configs[3]).
//
// Replaces, for inference:
//   FilterPredictor.predict_filter / predict_cls_bbreg_filters_parallel   ltr/models/transformer/filter_predictor.py:50-150
//   Transformer.forward + post-norm encoder / decoder layers              ltr/models/transformer/transformer.py:90-96,172-180,224-238
//   PositionEmbeddingSine('lin_sine', avoid_aliazing)                     ltr/models/transformer/position_encoding.py:6-58
//   DenseBoxRegressor.forward, the Linear of LinearFilterClassifier       ltr/models/transformer/heads.py:93-98,119-141
//
// This path is MFMA-bound (45 GFLOP per frame, AI >> 100 flop/B), unlike the solver passes: every dense contraction runs
// on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation -- the reference computes in fp32 and north_star
// asks for 1e-4).  Layout: tokens are rows, `(batch row, token) x channel` row-major, so every nn.Linear is an NT GEMM
// with both operands K-contiguous; the NCHW <-> token-major transposes happen once on the way in and once on the way out.
//
//   k_gemm      C = A W^T (+bias, affine, ReLU, residual, exp), LDS-staged 64x64 / 64x32 / 32x32 tiles, register
//               prefetch of the next K-step, optional "+pos" on the A operand of the q/k column blocks, optional 3x3
//               gather on the A operand (implicit GEMM for the regression tower)
//   k_attn      fp32 flash attention: S^T = K Q^T per 16-key tile so that the MFMA accumulator of the score tile IS the
//               B operand of the PV product (no LDS round trip for P), online softmax per 64-key chunk
//   decoder     one query token per batch row: the K/V projections of the memory are never formed --
//               q.(Wk m + bk) = (Wk^T q).m + const and sum_l p_l (Wv m_l + bv) = Wv (sum_l p_l m_l) + bv, and the weight
//               products this leaves (Wo Wv, Wk_h^T Wq_h, Wo[:,h] Wv_h) are folded once per weight update
//               (pt_tomp_prepare_f32), so a layer is 5 GEMVs plus two MFMA passes over the memory
#include "common.h"
#include "pt_internal.h"
#include "mfma_gemm.h"

#include <algorithm>
#include <math.h>
#include <stdint.h>

namespace {

// ------------------------------------------------------------------------------------------------------------------
// fp32 flash attention over the packed (rows, 3D) q|k|v projections
// ------------------------------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* qkv; unsigned qkv_bytes;
    float* out;
    int L, D, nhead;
    float scale;
    int mlo[8], mhi[8];                         // keys [mlo[b], mhi[b]) of batch row b are padding (never attended)
};

// Workgroup = 64 queries x 2 key halves: wavefronts 0-3 own the even 64-key chunks, 4-7 the odd ones (two wavefronts per
// SIMD, so one half's softmax / LDS waits hide behind the other's MFMAs -- B*nhead*L/16 query tiles alone are only one
// wavefront per SIMD); the halves' (max, sum, O^T) are merged through LDS at the end.
// PT_ATTN_EXP (experiments only, WRONG results, timing: profiles/r06q_attention_ablation.txt) -- bit mask:
//   1: no exponentials (P = S - m)      2: K / V staged once (no stash, no barriers after the first iteration)
//   4: fragments read from LDS once per iteration instead of per tile      8: K / V fetched from memory once
#ifndef PT_ATTN_EXP
#define PT_ATTN_EXP 0
#endif
#ifndef PT_ATTN_PIPE_EXP
#define PT_ATTN_PIPE_EXP 0   // 1 (experiment, no gain: r06q): exponentials of the next key tile issued between the PV MFMAs of the current one
#endif
template <int HD>
__global__ __launch_bounds__(512) void k_attn(AttnArgs a) {
    constexpr int KS = HD + 4, NV = HD / 4, NF4 = HD / 16, DT = HD / 16;   // NV floats of q/k per lane, NF4 float4s
    // LDS strides: K rows 4 mod 64 words, V rows 16 mod 32 words; together with the key permutation prow() below they
    // make the ds_read_b128 K fragments and the ds_read_b32 V operands bank-conflict free (see k_gemm for the lane
    // groups of ds_read_b128)
    constexpr int VS = HD <= 48 ? 48 : 80;
    __shared__ __attribute__((aligned(16))) float Ks[128 * KS];
    __shared__ __attribute__((aligned(16))) float Vs[128 * VS];
    auto prow = [](int i) { return (i >= 4 && i < 12) ? 2 * (i - 4) : (i < 4 ? 2 * i + 1 : 2 * i - 15); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int half = wave >> 2, h = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * 64 + (wave & 3) * 16;
    const int ld = 3 * a.D;
    const __amdgpu_buffer_rsrc_t rs = pt_rsrc(a.qkv, a.qkv_bytes);
    const long rowbase = (long)b * a.L;
    const int mlo = a.mlo[b], mhi = a.mhi[b];

    // Q fragment (B operand of S^T = K Q^T): lane (query li, k-slot kq) holds q[query][kq*NV .. +NV), pre-scaled
    float qf[NV];
    {
        const int q = min(q0 + li, a.L - 1);
        const unsigned off = (unsigned)(((rowbase + q) * ld + h * HD + kq * NV) * 4);
#pragma unroll
        for (int v = 0; v < NF4; ++v) {
            const f32x4 t = pt_bload4(rs, off + 16u * v);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[4 * v + e] = t[e] * a.scale;
        }
    }
    // cooperative K / V loads of 128 keys (one chunk per half) per iteration
    constexpr int LPT = (128 * HD / 4) / 512;                     // float4 per thread per operand
    constexpr int F4R = HD / 4;                                   // float4 per row
    f32x4 rk[LPT], rv[LPT];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = tid + 512 * i, r = idx / F4R, c4 = (idx - r * F4R) * 4, key = c0 + r;
            const unsigned off = key < a.L ? (unsigned)(((rowbase + key) * ld + a.D + h * HD + c4) * 4) : OOB;
            rk[i] = pt_bload4(rs, off);
            rv[i] = pt_bload4(rs, off == OOB ? OOB : off + (unsigned)a.D * 4u);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = tid + 512 * i, r = idx / F4R, c4 = (idx - r * F4R) * 4;
            *reinterpret_cast<f32x4*>(&Ks[r * KS + c4]) = rk[i];
            *reinterpret_cast<f32x4*>(&Vs[r * VS + c4]) = rv[i];
        }
    };

    f32x4 ot[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) ot[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // Padding keys (round 4).  A 128-key iteration that lies entirely inside the padding range [mlo, mhi) contributes exp(-inf) = 0
    // to every query and leaves (max, sum, O) bit for bit as they were: it is not fetched at all (ToMP's second batch row masks 324
    // of its 972 keys: two of its eight iterations).  A 64-key half chunk inside the range is skipped by its four wavefronts; one
    // that touches neither the range nor the end of the sequence runs without the per-key tests.
    const int nit = (a.L + 127) / 128;
    auto live = [&](int it) { return !(it * 128 >= mlo && it * 128 + 128 <= mhi); };
    int pk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = prow(4 * kq + r);
    int it = 0;
    bool first = true;                                            // (ablation builds only)
    while (it < nit && !live(it)) ++it;
    if (it < nit) fetch(it * 128);
    while (it < nit) {
        int nx = it + 1;
        while (nx < nit && !live(nx)) ++nx;
        if (!(PT_ATTN_EXP & 2) || first) {
            __syncthreads();                                      // previous iteration's LDS reads are done
            stash();
            __syncthreads();
        }
        if (nx < nit && !((PT_ATTN_EXP & 8) && !first)) fetch(nx * 128);
        first = false;
        const int c0 = it * 128 + half * 64;
        it = nx;
        if (c0 >= a.L || (c0 >= mlo && c0 + 64 <= mhi)) continue;  // wave-uniform: nothing but padding in this half chunk
        const bool tests = c0 + 64 > a.L || (c0 < mhi && c0 + 64 > mlo);
        const float* Kh = Ks + half * 64 * KS;
        const float* Vh = Vs + half * 64 * VS;
        // ---- S^T tiles: rows = keys, cols = queries.  The K fragments of tile kt+1 (and the V operands of the first PV
        //      tile) are read while tile kt is multiplied; the sched_barriers keep the scheduler from sinking the reads to
        //      their first use (it did: one exposed LDS round trip per 2 PV MFMAs, per 8 S MFMAs).
        f32x4 st[4];
        float mc = -INFINITY;
        f32x4 kf[2][NF4];
        float vf[2][4][DT];
        auto read_k = [&](int kt, int set) {
            const float* kp = Kh + (kt * 16 + prow(li)) * KS + kq * NV;   // S^T row i of the tile <-> key prow(i)
#pragma unroll
            for (int v = 0; v < NF4; ++v) kf[set][v] = *reinterpret_cast<const f32x4*>(kp + 4 * v);
        };
        auto read_v = [&](int kt, int set) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vp = Vh + (kt * 16 + prow(4 * kq + r)) * VS + li;
#pragma unroll
                for (int d = 0; d < DT; ++d) vf[set][r][d] = vp[16 * d];
            }
        };
        read_k(0, 0);
        if (PT_ATTN_EXP & 4) { read_k(1, 1); read_v(0, 0); read_v(1, 1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (PT_ATTN_EXP & 4) {}
            else if (kt < 3) read_k(kt + 1, (kt + 1) & 1);
            else read_v(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < NF4; ++v) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sc = mfma16(kf[kt & 1][v][e], qf[4 * v + e], sc);
            }
            if (tests) {                                          // wave-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = c0 + kt * 16 + pk[r];
                    if (key >= a.L || (key >= mlo && key < mhi)) sc[r] = -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mc = fmaxf(mc, sc[r]);
            st[kt] = sc;
            __builtin_amdgcn_sched_barrier(0);
        }
        mc = fmaxf(mc, __shfl_xor(mc, 16, 64));
        mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
        const float m_new = fmaxf(m_run, mc);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = (PT_ATTN_EXP & 1) ? 0.5f : __expf(m_run - m_use);                 // m_run = -inf -> 0
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < DT; ++d) ot[d] *= alpha;
        m_run = m_new;
        // ---- P^T = exp(S^T - m) feeds the PV product straight from the accumulator registers:
        //      O^T[d][query] += sum_key V[key][d] * P^T[key][query]; k-slot kq of MFMA r <-> key prow(4*kq + r) of the tile
        __builtin_amdgcn_sched_barrier(0);
#if PT_ATTN_PIPE_EXP
        // Round 6 experiment (profiles/r06q_attention_ablation.txt: the exponentials are 2.3 us of the launch): the four exponentials of
        // tile kt + 1 issued one behind each PV MFMA pair of tile kt (ISA: M M E x 12), only tile 0's in front.  Same bits -- and NO gain
        // (0.8875-0.8890 vs 0.8850-0.8878 ms): the transcendental issue does not overlap the matrix pipe's issue on this SIMD.
        auto pexp = [&](float sv) { return (PT_ATTN_EXP & 1) ? sv - m_use : __expf(sv - m_use); };
        float pc[4], pn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pc[r] = pexp(st[0][r]);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < 3 && !(PT_ATTN_EXP & 4)) read_v(kt + 1, (kt + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                l_run += pc[r];
#pragma unroll
                for (int d = 0; d < DT; ++d) ot[d] = mfma16(vf[kt & 1][r][d], pc[r], ot[d]);
                __builtin_amdgcn_sched_barrier(0);
                if (kt < 3) pn[r] = pexp(st[kt < 3 ? kt + 1 : 3][r]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) pc[r] = pn[r];
        }
#else
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < 3 && !(PT_ATTN_EXP & 4)) read_v(kt + 1, (kt + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = (PT_ATTN_EXP & 1) ? st[kt][r] - m_use : __expf(st[kt][r] - m_use);
                l_run += pe;
#pragma unroll
                for (int d = 0; d < DT; ++d) ot[d] = mfma16(vf[kt & 1][r][d], pe, ot[d]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    // ---- merge the two key halves (the exchange area aliases the K staging buffer)
    __syncthreads();
    float* xch = Ks + ((wave & 3) * 64 + lane) * (4 * DT + 2);
    if (half == 1) {
        xch[0] = m_run;
        xch[1] = l_run;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) xch[2 + 4 * d + r] = ot[d][r];
    }
    __syncthreads();
    if (half == 0) {
        const float m1 = xch[0], l1 = xch[1];
        const float m = fmaxf(m_run, m1);
        const float a0 = __expf(m_run - m), a1 = m1 == -INFINITY ? 0.f : __expf(m1 - m);
        const float inv = 1.f / (l_run * a0 + l1 * a1);
        const int q = q0 + li;
        if (q < a.L) {
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                f32x4 o;                                           // rows d*16 + 4*kq + r of O^T = 4 consecutive channels
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (ot[d][r] * a0 + xch[2 + 4 * d + r] * a1) * inv;
                *reinterpret_cast<f32x4*>(&a.out[(rowbase + q) * a.D + h * HD + d * 16 + 4 * kq]) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// row LayerNorm, one wavefront per row (D % 64 == 0, D <= 512); in-place safe
// ------------------------------------------------------------------------------------------------------------------
// PER = D / 64 known at compile time: the row is ONE vector load per lane and gamma / beta are requested with it (with a run-time
// count the loop stayed rolled -- a memory round trip per element -- and gamma / beta were read behind both reductions)
// in2 / in3 (optional, same layout): further addends of the row -- the second half of a split-K product and the residual the
// GEMM epilogue would have added (v = (in + in2) + in3, fixed order)
template <int PER>
__global__ __launch_bounds__(256) void k_ln_rows_t(const float* in, float* out, const float* __restrict__ gam,
                                                   const float* __restrict__ bet, int rows, const float* in2 = nullptr,
                                                   const float* in3 = nullptr, const float* gam2 = nullptr, const float* bet2 = nullptr,
                                                   const float* in4 = nullptr, const float* in5 = nullptr) {
    constexpr int D = 64 * PER;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[PER], g[PER], b[PER];
    const long ro = (long)row * D + lane * PER;
    const float* src = in + ro;
    if (PER == 4) {
        const f32x4 t = *(const f32x4*)src, tg = *(const f32x4*)(gam + lane * 4), tb = *(const f32x4*)(bet + lane * 4);
        f32x4 t2 = {0, 0, 0, 0}, t3 = {0, 0, 0, 0}, t4 = {0, 0, 0, 0}, t5 = {0, 0, 0, 0};
        if (in2) t2 = *(const f32x4*)(in2 + ro);                     // uniform branches: all loads requested together
        if (in3) t3 = *(const f32x4*)(in3 + ro);
        if (in4) { t4 = *(const f32x4*)(in4 + ro); t5 = *(const f32x4*)(in5 + ro); }   // splits 3 and 4 of a four-way split-K product (both or neither)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = in4 ? (((t[i] + t2[i]) + t4[i]) + t5[i]) + t3[i] : (in2 ? (t[i] + t2[i]) + t3[i] : t[i]);
            g[i] = tg[i]; b[i] = tb[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            v[i] = src[i];
            if (in2) v[i] = (v[i] + in2[ro + i]) + (in3 ? in3[ro + i] : 0.f);
            g[i] = gam[lane * PER + i]; b[i] = bet[lane * PER + i];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) q += (v[i] - mean) * (v[i] - mean);
    const float rstd = rsqrtf(wave_sum(q) / D + 1e-5f);
    float* dst = out + (long)row * D + lane * PER;
    if (gam2) {                                                     // uniform: a second LayerNorm on the result (the decoder's norm3 + final norm)
        float o[PER], s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) { o[i] = (v[i] - mean) * rstd * g[i] + b[i]; s2 += o[i]; }
        const float mean2 = wave_sum(s2) / D;
        float q2 = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) q2 += (o[i] - mean2) * (o[i] - mean2);
        const float rstd2 = rsqrtf(wave_sum(q2) / D + 1e-5f);
#pragma unroll
        for (int i = 0; i < PER; ++i) dst[i] = (o[i] - mean2) * rstd2 * gam2[lane * PER + i] + bet2[lane * PER + i];
        return;
    }
    if (PER == 4) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (v[i] - mean) * rstd * g[i] + b[i];
        *(f32x4*)dst = o;
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) dst[i] = (v[i] - mean) * rstd * g[i] + b[i];
    }
}

__global__ __launch_bounds__(256) void k_ln_rows(const float* in, float* out, const float* gam, const float* bet,
                                                 int rows, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int per = D / 64;
    float v[8];
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
        v[i] = in[(long)row * D + lane * per + i];
        s += v[i];
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
    for (int i = 0; i < per; ++i) q += (v[i] - mean) * (v[i] - mean);
    const float rstd = rsqrtf(wave_sum(q) / D + 1e-5f);
    for (int i = 0; i < per; ++i) {
        const int c = lane * per + i;
        out[(long)row * D + c] = (v[i] - mean) * rstd * gam[c] + bet[c];
    }
}

// The FFN's second product (K = dim_ff) runs as two K-halves whose partial sums, bias included, meet in the LayerNorm that follows
// (which then also adds the residual): 976 instead of 488 workgroups of half the K loop, ToMP frame 1.006 -> 0.996 ms (round 4,
// profiles/r04e_*).  PT_TOMP_FFN2_SPLIT=0 restores the single product (A/B knob).
static int pt_tomp_ffn2_split() {
    static const int v = [] { const char* e = getenv("PT_TOMP_FFN2_SPLIT"); return e ? atoi(e) : 4; }();   // round 6: 4 (k_gemm_ps); 2 = round 4's halves
    return v;
}

// in2 / in3 are only served by the 256-wide vector path: callers check ln_rows_wide_ok() first
static bool ln_rows_wide_ok(const float* in, const float* out, const float* gam, const float* bet, int D) {
    return D == 256 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)gam % 16 == 0) && ((uintptr_t)bet % 16 == 0);
}

// Returns PT_ERR_UNSUPPORTED (nothing queued) when extra operands are given that only the 256-wide vector kernel serves and that
// kernel cannot be used: dropping in2 / in3 / the second norm silently would lose the split-K half, the residual or a LayerNorm.
static int launch_ln_rows(const float* in, float* out, const float* gam, const float* bet, int rows, int D, hipStream_t st,
                          const float* in2 = nullptr, const float* in3 = nullptr, const float* gam2 = nullptr,
                          const float* bet2 = nullptr, const float* in4 = nullptr, const float* in5 = nullptr) {
    const dim3 grid((rows + 3) / 4), block(256);
    const bool al = ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)gam % 16 == 0) && ((uintptr_t)bet % 16 == 0) &&
                    ((uintptr_t)in2 % 16 == 0) && ((uintptr_t)in3 % 16 == 0) && ((uintptr_t)gam2 % 16 == 0) && ((uintptr_t)bet2 % 16 == 0) &&
                    ((uintptr_t)in4 % 16 == 0) && ((uintptr_t)in5 % 16 == 0);
    if (!(D == 256 && al) && (in2 || in3 || gam2 || bet2 || in4 || in5)) return PT_ERR_UNSUPPORTED;
    if ((in4 == nullptr) != (in5 == nullptr) || (in4 && !in2)) return PT_ERR_UNSUPPORTED;
    if (D == 256 && al) hipLaunchKernelGGL(k_ln_rows_t<4>, grid, block, 0, st, in, out, gam, bet, rows, in2, in3, gam2, bet2, in4, in5);
    else if (D == 128) hipLaunchKernelGGL(k_ln_rows_t<2>, grid, block, 0, st, in, out, gam, bet, rows);
    else if (D == 64) hipLaunchKernelGGL(k_ln_rows_t<1>, grid, block, 0, st, in, out, gam, bet, rows);
    else hipLaunchKernelGGL(k_ln_rows, grid, block, 0, st, in, out, gam, bet, rows, D);
    return PT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// token build (filter_predictor.py:113-128)
// ------------------------------------------------------------------------------------------------------------------
struct TokArgs {
    const float *train, *test, *label, *fg, *testtok;
    float* X;
    int nf, ns, dup, D, HW, L;
    float* zero; int nzero;         // a buffer this launch clears on the side (the decoder's initial state: no memset node per frame)
    // two more side jobs (round 4): `rep` copies of cp_src[k] (cp_n[k] floats) into cp_dst[k] -- the first decoder layer's state
    // after self-attention and its folded query do not depend on the frame (the decoder starts from zeros) and come from the
    // prepared buffer instead of two GEMV launches
    const float* cp_src[2]; float* cp_dst[2]; int cp_n[2]; int rep;
};

// NCHW feature maps -> token-major rows, + fg_token * label (memory frames) or + test_token (test frame)
__global__ __launch_bounds__(256) void k_tomp_tokens(TokArgs a) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int i = threadIdx.x; i < a.nzero; i += 256) a.zero[i] = 0.f;
    if (blockIdx.z == 0 && blockIdx.y >= 1 && blockIdx.y <= 2 && a.cp_src[blockIdx.y - 1]) {     // spread over the blocks of row y
        const int k = blockIdx.y - 1, n = a.cp_n[k];
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n * a.rep; i += gridDim.x * 256) a.cp_dst[k][i] = a.cp_src[k][i % n];
    }
    const int B = a.dup ? 2 : a.ns;
    const int fr = blockIdx.z / B, b = blockIdx.z - fr * B, s = a.dup ? 0 : b;
    const bool train = fr < a.nf;
    const float* src = train ? a.train + ((long)(fr * a.ns + s) * a.D) * a.HW : a.test + ((long)s * a.D) * a.HW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, p = p0 + tx;
        tile[ty + 8 * i][tx] = (c < a.D && p < a.HW) ? src[(long)c * a.HW + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + ty + 8 * i, c = c0 + tx;
        if (p >= a.HW || c >= a.D) continue;
        float v = tile[tx][ty + 8 * i];
        v += train ? a.fg[c] * a.label[(long)(fr * a.ns + s) * a.HW + p] : a.testtok[c];
        a.X[((long)b * a.L + (long)fr * a.HW + p) * a.D + c] = v;
    }
}

struct BoxArgs {
    const float *ltrb, *w1, *b1, *bn1, *bn2;     // bn: [weight, bias, running_mean, running_var] of D4 / D floats
    float *E1, *scale2, *shift2;
    int nf, ns, dup, D, D4, HW;
};

// first layer of the ltrb MLP (Conv1d 4 -> D/4, BatchNorm1d on running statistics, ReLU) and the folded scale / shift
// of the second BatchNorm for the epilogue of the next GEMM (filter_predictor.py:6-17)
__global__ __launch_bounds__(256) void k_tomp_box1(BoxArgs a) {
    const int B = a.dup ? 2 : a.ns, Ltr = a.nf * a.HW;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < a.D; c += 256) {
            const float sc = a.bn2[c] / sqrtf(a.bn2[3 * a.D + c] + 1e-5f);
            a.scale2[c] = sc;
            a.shift2[c] = a.bn2[a.D + c] - a.bn2[2 * a.D + c] * sc;
        }
    }
    if (idx >= (long)B * Ltr * a.D4) return;
    const int j = (int)(idx % a.D4);
    const long row = idx / a.D4;
    const int b = (int)(row / Ltr), l = (int)(row - (long)b * Ltr), fr = l / a.HW, p = l - fr * a.HW;
    const int s = a.dup ? 0 : b;
    const float* t = a.ltrb + ((long)(fr * a.ns + s) * 4) * a.HW + p;
    float e = a.b1[j];
#pragma unroll
    for (int i = 0; i < 4; ++i) e += a.w1[j * 4 + i] * t[(long)i * a.HW];
    e = (e - a.bn1[2 * a.D4 + j]) / sqrtf(a.bn1[3 * a.D4 + j] + 1e-5f) * a.bn1[j] + a.bn1[a.D4 + j];
    a.E1[idx] = fmaxf(e, 0.f);
}

// token-major rows of the test frame -> NCHW (enc_opt, filter_predictor.py:142-147)
__global__ __launch_bounds__(256) void k_tokens_to_nchw(const float* X, float* out, int B, int L, int D, int HW,
                                                        int l0) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (p < HW && c < D) ? X[((long)b * L + l0 + p) * D + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, p = p0 + tx;
        if (c < D && p < HW) out[((long)b * D + c) * HW + p] = tile[tx][ty + 8 * i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// positional encoding (position_encoding.py:6-58), written once per map size and cached by the caller
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_tomp_posenc(float* pos, int H, int W, int D, double factor) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W * D) return;
    const int p = idx / D, c = idx - p * D, depth = D / 4;
    const int y = p / W, x = p - y * W;
    const bool cosine = c >= 2 * depth;
    const int cc = cosine ? c - 2 * depth : c;
    const int i = cc / 2 + 1, isy = cc & 1;
    const float coord = isy ? ((float)(y + 1) - 0.5f) / ((float)H + 1e-6f) : ((float)(x + 1) - 0.5f) / ((float)W + 1e-6f);
    const float arg = (float)((double)i * factor * 3.14159265358979323846) * coord;   // scalar rounded to fp32, fp32 product
    pos[idx] = cosine ? cosf(arg) : sinf(arg);
}

// ------------------------------------------------------------------------------------------------------------------
// decoder: GEMV family (one wavefront per output feature; B <= 8 batch rows ride along in registers)
// ------------------------------------------------------------------------------------------------------------------
struct GemvArgs {
    const float* Wt; const float* bias; int N, K;
    const float* x;                                 // (B, K) input rows
    const float *xg, *xb;                           // optional LayerNorm applied to the input rows (K = model width)
    const float* xadd;                              // optional (K) vector added to every input row after the LayerNorm
    const float* res; const float *rg, *rb;         // optional residual rows (B, N), optionally LayerNorm'ed
    int relu, B;
    float* out;
};

template <int V>
__device__ __forceinline__ void ldv(const float* p, float* out) {      // V consecutive floats, 16-byte aligned when V = 4
    if (V == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; out[3] = t[3];
    } else {
        out[0] = p[0];
    }
}

template <int V>
__device__ __forceinline__ void wave_row_stats(const float* x, int K, int lane, float& mean, float& rstd) {
    float s = 0.f;
    for (int k = lane * V; k < K; k += 64 * V) {
        float t[V];
        ldv<V>(x + k, t);
#pragma unroll
        for (int e = 0; e < V; ++e) s += t[e];
    }
    mean = wave_sum(s) / K;
    float q = 0.f;
    for (int k = lane * V; k < K; k += 64 * V) {
        float t[V];
        ldv<V>(x + k, t);
#pragma unroll
        for (int e = 0; e < V; ++e) q += (t[e] - mean) * (t[e] - mean);
    }
    rstd = rsqrtf(wave_sum(q) / K + 1e-5f);
}

template <int V>
__global__ __launch_bounds__(256) void k_gemv(GemvArgs a) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    float acc[8], mean[8], rstd[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        acc[b] = 0.f;
        mean[b] = 0.f;
        rstd[b] = 1.f;
    }
    // all weight loads of the row are issued before anything that has to wait (LayerNorm statistics of the input and of
    // the residual are two dependent reductions each): the chain is latency, not bandwidth
    constexpr int WCH = 8;
    const float* w = a.Wt + (long)n * a.K;
    const int nch = a.K / (64 * V);
    const bool pre = nch <= WCH;
    float wreg[WCH][V];
    if (pre) {
#pragma unroll
        for (int ch = 0; ch < WCH; ++ch)
            if (ch < nch) ldv<V>(w + ch * 64 * V + lane * V, wreg[ch]);
    }
    const float bias = a.bias ? a.bias[n] : 0.f;
    float rterm[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) rterm[b] = 0.f;
    // <= 2 batch rows of <= 64*V*WCH elements (the decoder's shapes): the input rows and the residual rows are read ONCE,
    // together with the weights, and their LayerNorm statistics come from registers -- otherwise every statistic is two
    // dependent sweeps through memory and the dot product a third
    const bool cached = pre && a.B <= 2 && (!a.rg || a.N <= 64 * V * WCH);
    float xc[2][WCH][V];
    if (cached) {
        float rc[2][WCH][V];
        const int nrc = a.rg ? a.N / (64 * V) : 0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b >= a.B) continue;
#pragma unroll
            for (int ch = 0; ch < WCH; ++ch) {
                if (ch < nch) ldv<V>(a.x + (long)b * a.K + ch * 64 * V + lane * V, xc[b][ch]);
                if (ch < nrc) ldv<V>(a.res + (long)b * a.N + ch * 64 * V + lane * V, rc[b][ch]);
            }
        }
        float rn[2] = {0.f, 0.f};
        if (a.res) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (b < a.B) rn[b] = a.res[(long)b * a.N + n];
        }
        auto stats = [&](float (&v)[WCH][V], int nc, int len, float& m, float& rs) {     // same two-pass formula, from registers
            float s = 0.f;
#pragma unroll
            for (int ch = 0; ch < WCH; ++ch)
                if (ch < nc) {
#pragma unroll
                    for (int e = 0; e < V; ++e) s += v[ch][e];
                }
            m = wave_sum(s) / len;
            float q = 0.f;
#pragma unroll
            for (int ch = 0; ch < WCH; ++ch)
                if (ch < nc) {
#pragma unroll
                    for (int e = 0; e < V; ++e) q += (v[ch][e] - m) * (v[ch][e] - m);
                }
            rs = rsqrtf(wave_sum(q) / len + 1e-5f);
        };
        if (a.res) {
            const float rgn = a.rg ? a.rg[n] : 1.f, rbn = a.rg ? a.rb[n] : 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (b >= a.B) continue;
                float m = 0.f, rs = 1.f;
                if (a.rg) stats(rc[b], nrc, a.N, m, rs);
                rterm[b] = (rn[b] - m) * rs * rgn + rbn;
            }
        }
        if (a.xg) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (b < a.B) stats(xc[b], nch, a.K, mean[b], rstd[b]);
        }
    } else {
        if (a.res) {
            const float rgn = a.rg ? a.rg[n] : 1.f, rbn = a.rg ? a.rb[n] : 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b >= a.B) continue;
                const float* r = a.res + (long)b * a.N;
                float m = 0.f, rs = 1.f;
                if (a.rg) wave_row_stats<V>(r, a.N, lane, m, rs);
                rterm[b] = (r[n] - m) * rs * rgn + rbn;
            }
        }
        if (a.xg) {
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (b < a.B) wave_row_stats<V>(a.x + (long)b * a.K, a.K, lane, mean[b], rstd[b]);
        }
    }
    auto body = [&](int k, const float* wv, int ch) {
        float gv[V], bv[V], av[V];
#pragma unroll
        for (int e = 0; e < V; ++e) gv[e] = 1.f, bv[e] = 0.f, av[e] = 0.f;
        if (a.xg) {
            ldv<V>(a.xg + k, gv);
            ldv<V>(a.xb + k, bv);
        }
        if (a.xadd) ldv<V>(a.xadd + k, av);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b >= a.B) continue;
            float xr[V];
            if (cached && b < 2 && ch >= 0) {
#pragma unroll
                for (int e = 0; e < V; ++e) xr[e] = xc[b][ch][e];
            } else {
                ldv<V>(a.x + (long)b * a.K + k, xr);
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float xv = (xr[e] - mean[b]) * rstd[b] * gv[e] + bv[e];
                xv += av[e];
                acc[b] += wv[e] * xv;
            }
        }
    };
    if (pre) {
#pragma unroll
        for (int ch = 0; ch < WCH; ++ch)
            if (ch < nch) body(ch * 64 * V + lane * V, wreg[ch], ch);
    } else {
        for (int k = lane * V; k < a.K; k += 64 * V) {
            float wv[V];
            ldv<V>(w + k, wv);
            body(k, wv, -1);
        }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        if (b >= a.B) continue;
        float v = wave_sum(acc[b]) + bias;
        if (a.relu) v = fmaxf(v, 0.f);
        v += rterm[b];
        if (lane == 0) a.out[(long)b * a.N + n] = v;
    }
}

// The decoder's GEMVs (<= 2 batch rows, K = 256 * NCH): the `cached` path of k_gemv as its own kernel.  What the prologue needs to
// REQUEST every operand arrives as preloaded scalar parameters (six pointers, two packed words = 14 dwords); bias / residual-LayerNorm
// parameters / output pointer are fetched behind the loads (pt_late_args).  The LayerNorm gamma / beta of the input and the additive
// vector are requested with the weights instead of behind the statistics.  Same arithmetic in the same order as k_gemv.
struct GemvLate { pt_gcf bias, rg, rb; pt_gf out; };
template <int NCH>
__global__ __launch_bounds__(256) void k_gemv_dec(const float* __restrict__ Wt, const float* __restrict__ x, const float* __restrict__ xg,
                                                  const float* __restrict__ xb, const float* __restrict__ xadd,
                                                  const float* __restrict__ res, unsigned nk, unsigned flags, GemvLate l_arg) {
    constexpr int V = 4, WCH = 8;
    const int N = (int)(nk & 0xffffu), K = (int)(nk >> 16);
    const int B = (int)(flags & 15u), nrc = (int)((flags >> 8) & 15u);
    const bool relu = (flags >> 4) & 1u, has_rg = (flags >> 5) & 1u;
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = Wt + (long)n * K;
    float wreg[NCH][V], xc[2][NCH][V], gv[NCH][V], bv[NCH][V], av[NCH][V], rc[2][WCH][V];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int k = ch * 64 * V + lane * V;
        ldv<V>(w + k, wreg[ch]);
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (b < B) ldv<V>(x + (long)b * K + k, xc[b][ch]);
#pragma unroll
        for (int e = 0; e < V; ++e) gv[ch][e] = 1.f, bv[ch][e] = 0.f, av[ch][e] = 0.f;
        if (xg) {
            ldv<V>(xg + k, gv[ch]);
            ldv<V>(xb + k, bv[ch]);
        }
        if (xadd) ldv<V>(xadd + k, av[ch]);
    }
    float rn[2] = {0.f, 0.f};
    if (res) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b >= B) continue;
            rn[b] = res[(long)b * N + n];
            if (has_rg) {
#pragma unroll
                for (int ch = 0; ch < WCH; ++ch)
                    if (ch < nrc) ldv<V>(res + (long)b * N + ch * 64 * V + lane * V, rc[b][ch]);
            }
        }
    }
    const GemvLate l = pt_late_args<GemvLate>(56);                  // 6 pointers + 2 dwords = 56 bytes
    const float bias = l.bias ? l.bias[n] : 0.f;
    const float rgn = has_rg ? l.rg[n] : 1.f, rbn = has_rg ? l.rb[n] : 0.f;
    float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f}, rterm[2] = {0.f, 0.f};
    if (res) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b >= B) continue;
            float m = 0.f, rs = 1.f;
            if (has_rg) {                                           // the two-pass formula of wave_row_stats, from registers
                float s = 0.f;
#pragma unroll
                for (int ch = 0; ch < WCH; ++ch)
                    if (ch < nrc) {
#pragma unroll
                        for (int e = 0; e < V; ++e) s += rc[b][ch][e];
                    }
                m = wave_sum(s) / N;
                float q = 0.f;
#pragma unroll
                for (int ch = 0; ch < WCH; ++ch)
                    if (ch < nrc) {
#pragma unroll
                        for (int e = 0; e < V; ++e) q += (rc[b][ch][e] - m) * (rc[b][ch][e] - m);
                    }
                rs = rsqrtf(wave_sum(q) / N + 1e-5f);
            }
            rterm[b] = (rn[b] - m) * rs * rgn + rbn;
        }
    }
    if (xg) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b >= B) continue;
            float s = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int e = 0; e < V; ++e) s += xc[b][ch][e];
            mean[b] = wave_sum(s) / K;
            float q = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int e = 0; e < V; ++e) q += (xc[b][ch][e] - mean[b]) * (xc[b][ch][e] - mean[b]);
            rstd[b] = rsqrtf(wave_sum(q) / K + 1e-5f);
        }
    }
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b >= B) continue;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float xv = (xc[b][ch][e] - mean[b]) * rstd[b] * gv[ch][e] + bv[ch][e];
                xv += av[ch][e];
                acc[b] += wreg[ch][e] * xv;
            }
        }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b >= B) continue;
        float v = wave_sum(acc[b]) + bias;
        if (relu) v = fmaxf(v, 0.f);
        v += rterm[b];
        if (lane == 0) l.out[(long)b * N + n] = v;
    }
}

int launch_gemv(const GemvArgs& a, hipStream_t st) {
    if (a.B > 8 || a.K % 64 != 0 || (a.rg && a.N % 64 != 0)) return PT_ERR_UNSUPPORTED;
    const bool v4 = a.K % 256 == 0 && (!a.rg || a.N % 256 == 0);
    const int nch = a.K / 256;
    auto al16 = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
    const bool dec = v4 && a.B <= 2 && (nch == 1 || nch == 2 || nch == 4 || nch == 8) && a.N < 65536 && (!a.rg || (a.res && a.N <= 2048)) &&
                     al16(a.Wt) && al16(a.x) && al16(a.xg) && al16(a.xb) && al16(a.xadd) && al16(a.res);
    if (dec) {
        const unsigned nk = (unsigned)a.N | ((unsigned)a.K << 16);
        const unsigned flags = (unsigned)a.B | ((a.relu ? 1u : 0u) << 4) | ((a.rg ? 1u : 0u) << 5) | ((unsigned)(a.rg ? a.N / 256 : 0) << 8);
        const GemvLate l{(pt_gcf)a.bias, (pt_gcf)a.rg, (pt_gcf)a.rb, (pt_gf)a.out};
        const dim3 grid((a.N + 3) / 4), block(256);
#define PT_GEMV_DEC(NC) hipLaunchKernelGGL((k_gemv_dec<NC>), grid, block, 0, st, a.Wt, a.x, a.xg, a.xb, a.xadd, a.res, nk, flags, l)
        if (nch == 1) PT_GEMV_DEC(1);
        else if (nch == 2) PT_GEMV_DEC(2);
        else if (nch == 4) PT_GEMV_DEC(4);
        else PT_GEMV_DEC(8);
#undef PT_GEMV_DEC
    } else if (v4)
        hipLaunchKernelGGL((k_gemv<4>), dim3((a.N + 3) / 4), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((k_gemv<1>), dim3((a.N + 3) / 4), dim3(256), 0, st, a);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

// One-time folding of the decoder's weight products (pt_tomp_prepare_f32).  With a single query token per batch row
//   self-attention      out_proj(v_proj(t))                  = (Wo Wv) t + (Wo bv + bo)
//   query . key         q_h . (Wk_h (m + pos))               = (Wk_h^T Wq_h x + Wk_h^T bq_h) . (m + pos)   [+ const in l]
//   value + out_proj    Wo concat_h(Wv_h ctx_h + bv_h) + bo  = sum_h (Wo[:, h] Wv_h) ctx_h + (Wo bv + bo)
// so a decoder layer is 5 GEMVs + 2 passes over the memory instead of 8 + 2.  Plain loops: runs once per weight update.
__global__ void k_prep_sa(const float* Wo, const float* bo, const float* Wv, const float* bv, float* Wsa, float* bsa,
                          int D) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= D * D) return;
    const int i = idx / D, j = idx - i * D;
    float s = 0.f;
    for (int k = 0; k < D; ++k) s += Wo[i * D + k] * Wv[k * D + j];
    Wsa[idx] = s;
    if (j == 0) {
        float t = bo[i];
        for (int k = 0; k < D; ++k) t += Wo[i * D + k] * bv[k];
        bsa[i] = t;
    }
}

__global__ void k_prep_mq(const float* Wq, const float* bq, const float* Wk, float* Mq, float* cq, int D, int nhead,
                          float scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)nhead * D * D) return;
    const int HD = D / nhead, h = (int)(idx / ((long)D * D)), c = (int)((idx / D) % D), k = (int)(idx % D);
    float s = 0.f, t = 0.f;
    for (int j = 0; j < HD; ++j) {
        const float wk = Wk[(long)(h * HD + j) * D + c];
        s += wk * Wq[(long)(h * HD + j) * D + k];
        t += wk * bq[h * HD + j];
    }
    Mq[idx] = s * scale;
    if (k == 0) cq[h * D + c] = t * scale;
}

__global__ void k_prep_ov(const float* Wo, const float* bo, const float* Wv, const float* bv, float* Wov, float* bov,
                          int D, int nhead) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)D * nhead * D) return;
    const int HD = D / nhead, n = (int)(idx / ((long)nhead * D)), h = (int)((idx / D) % nhead), c = (int)(idx % D);
    float s = 0.f;
    for (int j = 0; j < HD; ++j) s += Wo[(long)n * D + h * HD + j] * Wv[(long)(h * HD + j) * D + c];
    Wov[idx] = s;
    if (h == 0 && c == 0) {
        float t = bo[n];
        for (int j = 0; j < D; ++j) t += Wo[(long)n * D + j] * bv[j];
        bov[n] = t;
    }
}

struct DecAttnArgs {
    const float *mem, *pos, *qk;
    float *scores, *ctx;
    int B, L, D, nhead, HW;
    int mlo[8], mhi[8];
};

// scores[b][h][l] = qk[b][h] . (mem[b][l] + pos[l % HW]) as a (16 rows x 16 heads) MFMA tile per wavefront:
// A = 16 memory rows (+pos), B = the folded queries of the (<= 16) heads, K = channels
__global__ __launch_bounds__(64) void k_dec_scores(DecAttnArgs a) {
    const int lane = threadIdx.x, li = lane & 15, kq = lane >> 4, b = blockIdx.y, l0 = blockIdx.x * 16;
    const int l = min(l0 + li, a.L - 1);
    const __amdgpu_buffer_rsrc_t rm = pt_rsrc(a.mem, (unsigned)((long)a.B * a.L * a.D * 4));
    const __amdgpu_buffer_rsrc_t rp = pt_rsrc(a.pos, (unsigned)((long)a.HW * a.D * 4));
    const __amdgpu_buffer_rsrc_t rq = pt_rsrc(a.qk, (unsigned)((long)a.B * a.nhead * a.D * 4));
    const unsigned om = (unsigned)((((long)b * a.L + l) * a.D + 4 * kq) * 4);
    const unsigned op = (unsigned)(((long)(l % a.HW) * a.D + 4 * kq) * 4);
    const unsigned oq = li < a.nhead ? (unsigned)((((long)b * a.nhead + li) * a.D + 4 * kq) * 4) : OOB;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nks = a.D / 16;                                     // D % 64 == 0: whole batches of 4 k-steps
    for (int k0 = 0; k0 < nks; k0 += 4) {                         // 12 loads in flight, then 16 MFMAs (24 in flight: measured slower, round 4)
        f32x4 m[4], pp[4], qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            m[u] = pt_bload4(rm, om + 64u * (k0 + u));
            pp[u] = pt_bload4(rp, op + 64u * (k0 + u));
            qv[u] = pt_bload4(rq, oq == OOB ? OOB : oq + 64u * (k0 + u));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma16(m[u][e] + pp[u][e], qv[u][e], acc);
    }
    if (li < a.nhead) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ll = l0 + 4 * kq + r;
            if (ll < a.L)
                a.scores[((long)b * a.nhead + li) * a.L + ll] = (ll >= a.mlo[b] && ll < a.mhi[b]) ? -INFINITY : acc[r];
        }
    }
}

// ctx[b][h][c] = sum_l softmax(scores[b][h])[l] * mem[b][l][c]: grid (D/16, B), 16 wavefronts.  Prologue: wavefront h
// turns scores[b][h] into exp(s - max) in LDS and keeps 1/sum; then every wavefront contracts its share of the memory
// rows on MFMA (A = probabilities of the <= 16 heads, B = 16 channels of 4 memory rows), partial tiles reduced in LDS
__global__ __launch_bounds__(1024) void k_dec_ctx(DecAttnArgs a) {
    extern __shared__ float sm[];                                // nhead * Lp probabilities, 16 inverse sums, 16 x 256 partials
    const int Lp = (a.L + 3) & ~3;
    float* p = sm;
    float* inv = sm + (size_t)a.nhead * Lp;
    float* part = inv + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int b = blockIdx.y, c0 = blockIdx.x * 16;
    // the first round of memory rows is requested in front of the softmax (it does not depend on it)
    const __amdgpu_buffer_rsrc_t rm = pt_rsrc(a.mem, (unsigned)((long)a.B * a.L * a.D * 4));
    const int nks = Lp / 4;
    float mv0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k2 = wave + 16 * u, l = 4 * k2 + kq;
        mv0[u] = pt_bload1(rm, (k2 < nks && l < a.L) ? (unsigned)((((long)b * a.L + l) * a.D + c0 + li) * 4) : OOB);
    }
    if (wave < a.nhead) {
        // the head's score row in ONE batch of loads (L <= 1408 by the LDS budget the launcher checks: 22 per lane), kept in
        // registers for both softmax passes.  As two strided loops with a run-time bound (round 3) every element was its own
        // memory round trip, twice: 32 dependent round trips = the kernel's 10 us (round 4: profiles/r04x_*)
        constexpr int NSV = 22;
        const float* s = a.scores + ((long)b * a.nhead + wave) * a.L;
        if (a.L > 64 * NSV) {                                       // uniform: longer rows (few heads) keep the two strided passes
            float mx = -INFINITY;
            for (int l = lane; l < a.L; l += 64) mx = fmaxf(mx, s[l]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int l = lane; l < Lp; l += 64) {
                const float e = l < a.L ? __expf(s[l] - mx) : 0.f;
                p[(size_t)wave * Lp + l] = e;
                sum += e;
            }
            sum = wave_sum(sum);
            if (lane == 0) inv[wave] = 1.f / sum;
        } else {
        float sv[NSV];
#pragma unroll
        for (int u = 0; u < NSV; ++u) sv[u] = s[min(lane + 64 * u, a.L - 1)];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < NSV; ++u) mx = fmaxf(mx, lane + 64 * u < a.L ? sv[u] : -INFINITY);
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < NSV; ++u) {
            const int l = lane + 64 * u;
            const float e = l < a.L ? __expf(sv[u] - mx) : 0.f;
            if (l < Lp) p[(size_t)wave * Lp + l] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) inv[wave] = 1.f / sum;
        }
    }
    __syncthreads();
    const float hsel = li < a.nhead ? 1.f : 0.f;
    const float* ph = p + (size_t)min(li, a.nhead - 1) * Lp;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = wave; ks < nks; ks += 64) {                     // this wavefront's k-steps ks, ks+16, ks+32, ks+48:
        float mv[4], pv[4];                                       // four loads in flight, then four MFMAs (sixteen: measured slower)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k2 = ks + 16 * u, l = 4 * k2 + kq;
            const bool live = k2 < nks;
            if (ks == wave) mv[u] = mv0[u];                       // uniform: the round requested in front of the softmax
            else mv[u] = pt_bload1(rm, (live && l < a.L) ? (unsigned)((((long)b * a.L + l) * a.D + c0 + li) * 4) : OOB);
            pv[u] = live ? ph[l] * hsel : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = mfma16(pv[u], mv[u], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave * 256 + (4 * kq + r) * 16 + li] = acc[r];
    __syncthreads();
    if (tid < 256) {
        const int h = tid >> 4, c = tid & 15;
        if (h < a.nhead) {
            float t = 0.f;
            for (int w = 0; w < 16; ++w) t += part[w * 256 + tid];
            a.ctx[((long)b * a.nhead + h) * a.D + c0 + c] = t * inv[h];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// box-regression head pieces (heads.py:119-141)
// ------------------------------------------------------------------------------------------------------------------
// att[p] = feat[:, p] . filt;  T[p][c] = att[p] * feat[c][p] (token-major), one workgroup per (32 positions, image)
__global__ __launch_bounds__(256) void k_reg_attend(const float* feat, const float* filt, float* T, int D, int HW) {
    extern __shared__ float sm[];                                // D x 33 tile + 32 attention values
    float* tile = sm;
    float* att = sm + (size_t)D * 33;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, p0 = blockIdx.x * 32, img = blockIdx.y;
    const float* src = feat + (long)img * D * HW;
    float part = 0.f;
    // 16 channels per thread and batch, clamped: all loads of a batch in flight (with the run-time bound as the only loop every
    // channel was its own round trip: 32 of them at D = 256, 13 us for a 330 KB map; round 4)
    const int px = min(p0 + tx, HW - 1);
    const bool pin = p0 + tx < HW;
    for (int cb = 0; cb < D; cb += 128) {
        float v[16], fv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = min(cb + ty + 8 * u, D - 1);
            v[u] = src[(long)c * HW + px];
            fv[u] = filt[c];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = cb + ty + 8 * u;
            if (c < D) {
                const float x = pin ? v[u] : 0.f;
                tile[c * 33 + tx] = x;
                part += x * fv[u];
            }
        }
    }
    __shared__ float red[8][33];
    red[ty][tx] = part;
    __syncthreads();
    if (ty == 0) {
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += red[i][tx];
        att[tx] = s;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i;
        if (p >= HW) continue;
        for (int c = tx; c < D; c += 32) T[((long)img * HW + p) * D + c] = att[i] * tile[c * 33 + i];
    }
}

// GroupNorm(1, C) + ReLU of the regression tower, fed by the split-K partial products of the 3x3 convolution:
//   k_gn_reduce: x = sum_z part[z] (fixed order), written once, + per-slice (sum, sum of squares) of one image
//   k_gn_apply:  mean / variance from the slice sums (double, fixed order), normalise + affine + ReLU in place
// grid (slices of 4096 elements, images); token-major (HW, C) blocks
constexpr int GN_SLICE = 1024;

__global__ __launch_bounds__(256) void k_gn_reduce(const float* __restrict__ part, int nz, long zstride,
                                                   float* __restrict__ x, float* __restrict__ stats, int n_per_img) {
    __shared__ float scratch[4];
    const int img = blockIdx.y, i0 = blockIdx.x * GN_SLICE;
    const long base = (long)img * n_per_img;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int u = 0; u < GN_SLICE / 1024; ++u) {
        const int i = i0 + (u * 256 + threadIdx.x) * 4;
        if (i < n_per_img) {
            // the split partials in ONE batch of loads, fixed order (a run-time bound leaves fewer than 8 iterations -- the
            // feature head's 8 channel splits -- to a remainder loop of one dependent load per round trip; round 4)
            f32x4 pv[9];                                            // 8 channel splits (feature head) or 9 taps (tower): one batch
#pragma unroll
            for (int z = 0; z < 9; ++z) pv[z] = *reinterpret_cast<const f32x4*>(part + (long)min(z, nz - 1) * zstride + base + i);
            f32x4 v = pv[0];
#pragma unroll
            for (int z = 1; z < 9; ++z)
                if (z < nz) v += pv[z];
            for (int z = 9; z < nz; ++z) v += *reinterpret_cast<const f32x4*>(part + z * zstride + base + i);
            *reinterpret_cast<f32x4*>(x + base + i) = v;
            s += (v[0] + v[1]) + (v[2] + v[3]);
            q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    }
    s = block_sum(s, scratch);
    q = block_sum(q, scratch);
    if (threadIdx.x == 0) {
        stats[((long)img * gridDim.x + blockIdx.x) * 2] = s;
        stats[((long)img * gridDim.x + blockIdx.x) * 2 + 1] = q;
    }
}

__global__ __launch_bounds__(256) void k_gn_apply(float* __restrict__ x, const float* __restrict__ stats,
                                                  const float* __restrict__ gam, const float* __restrict__ bet, int D,
                                                  int n_per_img) {
    const int img = blockIdx.y, i0 = blockIdx.x * GN_SLICE;
    double s = 0.0, q = 0.0;
    for (unsigned k = 0; k < gridDim.x; ++k) {
        s += (double)stats[((long)img * gridDim.x + k) * 2];
        q += (double)stats[((long)img * gridDim.x + k) * 2 + 1];
    }
    const double mean_d = s / n_per_img;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(fmax(q / n_per_img - mean_d * mean_d, 0.0) + 1e-5));
    const long base = (long)img * n_per_img;
#pragma unroll
    for (int u = 0; u < GN_SLICE / 1024; ++u) {
        const int i = i0 + (u * 256 + threadIdx.x) * 4;
        if (i < n_per_img) {
            const int c = i % D;
            f32x4 v = *reinterpret_cast<const f32x4*>(x + base + i);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gam + c), b4 = *reinterpret_cast<const f32x4*>(bet + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf((v[e] - mean) * rstd * g4[e] + b4[e], 0.f);
            *reinterpret_cast<f32x4*>(x + base + i) = v;
        }
    }
}

// ltrb[img][n][p] = exp(sum_z part[z][img*HW + p][n]) (heads.py:134)
__global__ __launch_bounds__(256) void k_reg_finish(const float* __restrict__ part, int nz, long zstride,
                                                    float* __restrict__ out, int M, int HW) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * 4) return;
    float v = part[idx];
    for (int z = 1; z < nz; ++z) v += part[z * zstride + idx];
    const int row = idx >> 2, n = idx & 3, img = row / HW;
    out[((long)img * 4 + n) * HW + (row - img * HW)] = expf(v);
}

// ------------------------------------------------------------------------------------------------------------------
// parameter pack layout (documented in include/pt_hot.h)
// ------------------------------------------------------------------------------------------------------------------
struct MhaOff { size_t w_in, b_in, w_out, b_out; };
struct EncOff { MhaOff sa; size_t w1, b1, w2, b2, n1g, n1b, n2g, n2b; };
struct DecOff { MhaOff sa, ca; size_t w1, b1, w2, b2, n1g, n1b, n2g, n2b, n3g, n3b; };
struct PackOff {
    EncOff enc[16];
    DecOff dec[16];
    size_t dng, dnb, bw1, bb1, bn1, bw2, bb2, bn2, bw3, bb3, fg, testtok, total;
};

PackOff pack_layout(int D, int ff, int n_enc, int n_dec) {
    PackOff o{};
    size_t c = 0;
    auto take = [&](size_t n) { size_t r = c; c += n; return r; };
    auto mha = [&](MhaOff& m) {
        m.w_in = take((size_t)3 * D * D); m.b_in = take((size_t)3 * D); m.w_out = take((size_t)D * D); m.b_out = take(D);
    };
    for (int i = 0; i < n_enc; ++i) {
        EncOff& e = o.enc[i];
        mha(e.sa);
        e.w1 = take((size_t)ff * D); e.b1 = take(ff); e.w2 = take((size_t)D * ff); e.b2 = take(D);
        e.n1g = take(D); e.n1b = take(D); e.n2g = take(D); e.n2b = take(D);
    }
    for (int i = 0; i < n_dec; ++i) {
        DecOff& d = o.dec[i];
        mha(d.sa);
        mha(d.ca);
        d.w1 = take((size_t)ff * D); d.b1 = take(ff); d.w2 = take((size_t)D * ff); d.b2 = take(D);
        d.n1g = take(D); d.n1b = take(D); d.n2g = take(D); d.n2b = take(D); d.n3g = take(D); d.n3b = take(D);
    }
    o.dng = take(D); o.dnb = take(D);
    const int D4 = D / 4;
    o.bw1 = take((size_t)D4 * 4); o.bb1 = take(D4); o.bn1 = take((size_t)4 * D4);
    o.bw2 = take((size_t)D * D4); o.bb2 = take(D); o.bn2 = take((size_t)4 * D);
    o.bw3 = take((size_t)D * D); o.bb3 = take(D);
    o.fg = take(D); o.testtok = take(D);
    o.total = c;
    return o;
}

struct PrepOff { size_t wsa, bsa, mq, cq, wov, bov; };
struct PrepLayout { PrepOff dec[16]; size_t qk0; size_t total; };   // qk0: the first decoder layer's folded query (below), (nhead, D)
PrepLayout prep_layout(int D, int nhead, int n_dec) {
    PrepLayout o{};
    size_t c = 0;
    auto take = [&](size_t n) { size_t r = c; c += pt_align_floats(n); return r; };
    for (int i = 0; i < n_dec; ++i) {
        PrepOff& p = o.dec[i];
        p.wsa = take((size_t)D * D); p.bsa = take(D); p.mq = take((size_t)nhead * D * D); p.cq = take((size_t)nhead * D);
        p.wov = take((size_t)D * nhead * D); p.bov = take(D);
    }
    o.qk0 = take((size_t)nhead * D);
    o.total = c;
    return o;
}

int dims_check(const pt_tomp_dims* d) {
    if (!d) return PT_ERR_NULL;
    if (d->d_model <= 0 || d->nhead <= 0 || d->dim_ff <= 0 || d->n_enc < 0 || d->n_dec < 0 || d->H <= 0 || d->W <= 0 ||
        d->max_res <= 0)
        return PT_ERR_SHAPE;
    if (d->d_model % d->nhead != 0) return PT_ERR_SHAPE;
    const int hd = d->d_model / d->nhead;
    if ((hd != 16 && hd != 32 && hd != 64) || d->d_model % 128 != 0 || d->d_model > 512 || d->dim_ff % 64 != 0 ||
        d->n_enc > 16 || d->n_dec > 16 || d->nhead > 16)
        return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

struct WsCarve {
    size_t X, QKV, AO, Y, Hd, E1, E2, bnsc, bnsh, zero, a, qk, scores, ctx, P[2], P1, P2, hdn, total;
};

WsCarve ws_carve(const pt_tomp_dims* d, int B, int nf) {
    WsCarve w{};
    const size_t D = d->d_model, HW = (size_t)d->H * d->W, L = (nf + 1) * HW, rows = B * L;
    size_t c = 0;
    auto take = [&](size_t n) { size_t r = c; c += pt_align_floats(n); return r; };
    w.X = take(rows * D); w.QKV = take(rows * 3 * D); w.AO = take(rows * D); w.Y = take(rows * D);
    w.Hd = take(rows * d->dim_ff);
    w.E1 = take((size_t)B * nf * HW * (D / 4)); w.E2 = take((size_t)B * nf * HW * D);
    w.bnsc = take(D); w.bnsh = take(D);
    w.zero = take(B * D); w.a = take(B * D); w.qk = take((size_t)B * d->nhead * D);
    w.scores = take((size_t)B * d->nhead * L); w.ctx = take((size_t)B * d->nhead * D);
    w.P[0] = take(B * D); w.P[1] = take(B * D); w.P1 = take(B * D); w.P2 = take(B * D); w.hdn = take((size_t)B * d->dim_ff);
    w.total = c;
    return w;
}

}  // namespace

// ==================================================================================================================
// C ABI
// ==================================================================================================================
extern "C" size_t pt_tomp_param_floats(const pt_tomp_dims* d) {
    if (dims_check(d)) return 0;
    return pack_layout(d->d_model, d->dim_ff, d->n_enc, d->n_dec).total;
}

extern "C" int pt_tomp_posenc_f32(float* pos, int H, int W, int d_model, int max_res, void* stream) {
    if (!pos) return PT_ERR_NULL;
    if (H <= 0 || W <= 0 || d_model <= 0 || d_model % 4 != 0 || max_res <= 0) return PT_ERR_SHAPE;
    const int total = H * W * d_model;
    const double factor = (double)max_res / (double)(d_model / 4);
    hipLaunchKernelGGL(k_tomp_posenc, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, H, W, d_model,
                       factor);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" size_t pt_tomp_predict_ws_bytes(const pt_tomp_dims* d, int n_train, int n_seq, int parallel) {
    if (dims_check(d) || n_train <= 0 || n_seq <= 0) return 0;
    const int B = parallel ? 2 : n_seq;
    if (B > 8 || (parallel && n_seq != 1)) return 0;
    if ((size_t)d->nhead * ((size_t)(n_train + 1) * d->H * d->W + 4) > 11000) return 0;   // LDS of k_dec_ctx
    return ws_carve(d, B, n_train).total * sizeof(float);
}

extern "C" int pt_tomp_linear_f32(const float* weight, const float* bias, const float* x, float* y, int B, int N, int K,
                                  int relu, void* stream) {
    if (!weight || !x || !y) return PT_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0) return PT_ERR_SHAPE;
    GemvArgs a{};
    a.Wt = weight; a.bias = bias; a.N = N; a.K = K; a.x = x; a.relu = relu; a.B = B; a.out = y;
    return launch_gemv(a, (hipStream_t)stream);
}

extern "C" size_t pt_tomp_prepared_floats(const pt_tomp_dims* d) {
    if (dims_check(d)) return 0;
    return std::max<size_t>(prep_layout(d->d_model, d->nhead, d->n_dec).total, 64);
}

extern "C" int pt_tomp_prepare_f32(const pt_tomp_dims* d, const float* params, float* prepared, void* stream) {
    if (!params || !prepared) return PT_ERR_NULL;
    int rc = dims_check(d);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int D = d->d_model, NH = d->nhead;
    const PackOff po = pack_layout(D, d->dim_ff, d->n_enc, d->n_dec);
    const PrepLayout pl = prep_layout(D, NH, d->n_dec);
    const float scale = 1.0f / sqrtf((float)(D / NH));
    for (int i = 0; i < d->n_dec; ++i) {
        const DecOff& dc = po.dec[i];
        const PrepOff& pp = pl.dec[i];
        const float *sa_w = params + dc.sa.w_in, *sa_b = params + dc.sa.b_in, *ca_w = params + dc.ca.w_in,
                    *ca_b = params + dc.ca.b_in;
        hipLaunchKernelGGL(k_prep_sa, dim3((D * D + 255) / 256), dim3(256), 0, st, params + dc.sa.w_out,
                           params + dc.sa.b_out, sa_w + (size_t)2 * D * D, sa_b + 2 * D, prepared + pp.wsa, prepared + pp.bsa, D);
        PT_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_prep_mq, dim3((unsigned)(((long)NH * D * D + 255) / 256)), dim3(256), 0, st, ca_w, ca_b,
                           ca_w + (size_t)D * D, prepared + pp.mq, prepared + pp.cq, D, NH, scale);
        PT_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_prep_ov, dim3((unsigned)(((long)NH * D * D + 255) / 256)), dim3(256), 0, st,
                           params + dc.ca.w_out, params + dc.ca.b_out, ca_w + (size_t)2 * D * D, ca_b + 2 * D,
                           prepared + pp.wov, prepared + pp.bov, D, NH);
        PT_CHECK_LAUNCH();
    }
    if (d->n_dec > 0) {
        // The decoder starts from zeros (transformer.py:224-238), so layer 0's self-attention output is its folded bias,
        // Pa = (Wo bv + bo), and its folded query Mq (LN1(Pa) + query_pos) + cq is a constant of the weights: one GEMV here
        // instead of two per frame (the same kernel, one row -> bit-identical to what the per-frame launches produced)
        const DecOff& dc = po.dec[0];
        const PrepOff& pp = pl.dec[0];
        GemvArgs v{};
        v.Wt = prepared + pp.mq; v.bias = prepared + pp.cq; v.N = NH * D; v.K = D; v.x = prepared + pp.bsa; v.xg = params + dc.n1g;
        v.xb = params + dc.n1b; v.xadd = params + po.fg; v.B = 1; v.out = prepared + pl.qk0;
        const int rc2 = launch_gemv(v, st);
        if (rc2) return rc2;
    }
    return PT_OK;
}

extern "C" int pt_tomp_predict_f32(const pt_tomp_dims* d, const float* params, const float* prepared, const float* pos,
                                   const float* train_feat,
                                   const float* test_feat, const float* train_label, const float* train_ltrb,
                                   int n_train, int n_seq, int parallel, int num_gth_frames, float* filters,
                                   float* enc_feat, void* ws, size_t ws_bytes, void* stream) {
    if (!params || !prepared || !pos || !train_feat || !test_feat || !train_label || !train_ltrb || !filters || !enc_feat || !ws)
        return PT_ERR_NULL;
    int rc = dims_check(d);
    if (rc) return rc;
    if (n_train <= 0 || n_seq <= 0 || num_gth_frames < 0 || num_gth_frames > n_train) return PT_ERR_SHAPE;
    const int B = parallel ? 2 : n_seq;
    if (B > 8 || (parallel && n_seq != 1)) return PT_ERR_UNSUPPORTED;
    const int D = d->d_model, ff = d->dim_ff, NH = d->nhead, HD = D / NH, HW = d->H * d->W, Ltr = n_train * HW,
              L = Ltr + HW, rows = B * L, D4 = D / 4;
    if ((size_t)NH * (L + 4) > 11000) return PT_ERR_UNSUPPORTED;
    const WsCarve cv = ws_carve(d, B, n_train);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* base = (float*)ws;
    const PackOff po = pack_layout(D, ff, d->n_enc, d->n_dec);
    const float* P = params;
    float *X = base + cv.X, *QKV = base + cv.QKV, *AO = base + cv.AO, *Y = base + cv.Y, *Hd = base + cv.Hd;

    bool dec0_const = false;                                       // decoder layer 0's first two GEMVs replaced by prepared constants
    // ---- tokens (filter_predictor.py:113-128)
    {
        TokArgs t{train_feat, test_feat, train_label, P + po.fg, P + po.testtok, X, n_train, n_seq, parallel ? 1 : 0, D,
                  HW, L, base + cv.zero, B * D, {nullptr, nullptr}, {nullptr, nullptr}, {0, 0}, B};
        static const bool dec0_off = [] { const char* e = getenv("PT_TOMP_DEC0"); return e && e[0] == '0'; }();   // A/B knob
        if (d->n_dec > 0 && (D + 31) / 32 >= 3 && !dec0_off) {    // layer 0 of the decoder: state after self-attention, folded query
            const PrepLayout pl0 = prep_layout(D, NH, d->n_dec);
            t.cp_src[0] = prepared + pl0.dec[0].bsa; t.cp_dst[0] = base + cv.P1; t.cp_n[0] = D;
            t.cp_src[1] = prepared + pl0.qk0; t.cp_dst[1] = base + cv.qk; t.cp_n[1] = NH * D;
        }
        dec0_const = t.cp_src[0] != nullptr;
        hipLaunchKernelGGL(k_tomp_tokens, dim3((HW + 31) / 32, (D + 31) / 32, (n_train + 1) * B), dim3(256), 0, st, t);
        PT_CHECK_LAUNCH();
        BoxArgs bx{train_ltrb, P + po.bw1, P + po.bb1, P + po.bn1, P + po.bn2, base + cv.E1, base + cv.bnsc,
                   base + cv.bnsh, n_train, n_seq, parallel ? 1 : 0, D, D4, HW};
        const long tot = (long)B * Ltr * D4;
        hipLaunchKernelGGL(k_tomp_box1, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, bx);
        PT_CHECK_LAUNCH();
        GemmArgs g = gemm_args(base + cv.E1, D4, (long)B * Ltr, P + po.bw2, B * Ltr, D, D4, P + po.bb2, base + cv.E2, D);
        g.scale = base + cv.bnsc; g.shift = base + cv.bnsh; g.relu = 1;
        if ((rc = launch_gemm(g, st))) return rc;
        g = gemm_args(base + cv.E2, D, (long)B * Ltr, P + po.bw3, B * Ltr, D, D, P + po.bb3, X, D);
        g.R = X; g.c_seg = Ltr; g.c_segstride = L;
        if ((rc = launch_gemm(g, st))) return rc;
    }
    AttnArgs at{};
    at.qkv = QKV; at.qkv_bytes = (unsigned)((long)rows * 3 * D * 4); at.out = AO; at.L = L; at.D = D; at.nhead = NH;
    at.scale = 1.0f / sqrtf((float)HD);
    for (int b = 0; b < 8; ++b) at.mlo[b] = at.mhi[b] = 0;
    if (parallel) {                                               // filter_predictor.py:134-136: batch row 1 only
        at.mlo[1] = num_gth_frames * HW;
        at.mhi[1] = L - HW;
    }
    // ---- encoder (transformer.py:172-180), post-norm
    for (int i = 0; i < d->n_enc; ++i) {
        const EncOff& e = po.enc[i];
        GemmArgs g = gemm_args(X, D, rows, P + e.sa.w_in, rows, 3 * D, D, P + e.sa.b_in, QKV, 3 * D);
        g.pos = pos; g.pos_bytes = (unsigned)((long)HW * D * 4); g.pos_cols = 2 * D; g.L = L; g.HW = HW;
        if ((rc = launch_gemm(g, st))) return rc;
        const dim3 ag((L + 63) / 64, NH, B);
        if (HD == 32) hipLaunchKernelGGL((k_attn<32>), ag, dim3(512), 0, st, at);
        else if (HD == 16) hipLaunchKernelGGL((k_attn<16>), ag, dim3(512), 0, st, at);
        else hipLaunchKernelGGL((k_attn<64>), ag, dim3(512), 0, st, at);
        PT_CHECK_LAUNCH();
        g = gemm_args(AO, D, rows, P + e.sa.w_out, rows, D, D, P + e.sa.b_out, Y, D);
        g.R = X;
        if ((rc = launch_gemm(g, st))) return rc;
        // round 6, experiment P (LOST, profiles/r06p_tomp_ln1_fold.txt; opt-in PT_TOMP_LN1_FOLD=1): norm1 riding on the FFN's first
        // product (k_gemm_ps<.., LNA>: the A slab normalised in registers, the column-tile-0 workgroups store X for the residual of
        // norm2) saves a launch per layer, but the slab-up-front prologue costs that product 4.6 us (23.5 -> 28.1) against the 4.6 us
        // LayerNorm launch (~3 in the chain): frame 0.885 -> 0.894 ms.
        static const bool ln1_fold = [] { const char* e = getenv("PT_TOMP_LN1_FOLD"); return e && e[0] == '1'; }();
        g = gemm_args(X, D, rows, P + e.w1, rows, ff, D, P + e.b1, Hd, ff);
        g.relu = 1;
        bool folded = false;
        if (ln1_fold && D == 256 && rows >= 1024 && ff >= 1024) {
            GemmArgs gl = g;
            gl.A = Y; gl.ln_gam = P + e.n1g; gl.ln_bet = P + e.n1b; gl.ln_out = X;
            const int rl = launch_gemm(gl, st);
            if (rl == PT_OK) folded = true;
            else if (rl != PT_ERR_UNSUPPORTED) return rl;
        }
        if (!folded) {
            if ((rc = launch_ln_rows(Y, X, P + e.n1g, P + e.n1b, rows, D, st))) return rc;
            PT_CHECK_LAUNCH();
            if ((rc = launch_gemm(g, st))) return rc;
        }
        const long rD = (long)rows * D;
        if (pt_tomp_ffn2_split() == 4 && (ff / 64) % 4 == 0 && rows >= 1024 && D % 64 == 0 && cv.AO == cv.QKV + 3 * (size_t)rD &&
            ln_rows_wide_ok(QKV, X, P + e.n2g, P + e.n2b, D)) {
            // round 6: four K-quarters on 128 x 64 tiles (k_gemm_ps: 256 workgroups x 16 stages, M tiles on x) into the four row blocks
            // QKV[0..2], AO -- all free here -- that meet, with the residual, in the LayerNorm
            g = gemm_args(Hd, ff, rows, P + e.w2, rows, D, ff, P + e.b2, QKV, D);
            g.ksteps = ff / 64 / 4; g.c_zstride = rD;
            if ((rc = launch_gemm(g, st))) return rc;
            if ((rc = launch_ln_rows(QKV, X, P + e.n2g, P + e.n2b, rows, D, st, QKV + rD, X, nullptr, nullptr, QKV + 2 * rD, AO))) return rc;
            PT_CHECK_LAUNCH();
            continue;
        }
        if (pt_tomp_ffn2_split() == 2 && (ff / 64) % 2 == 0 && cv.Y > cv.AO && ln_rows_wide_ok(AO, X, P + e.n2g, P + e.n2b, D)) {
            g = gemm_args(Hd, ff, rows, P + e.w2, rows, D, ff, P + e.b2, AO, D);
            g.ksteps = ff / 64 / 2; g.c_zstride = (long)(cv.Y - cv.AO);          // halves -> AO, Y (bias rides on the first)
            if ((rc = launch_gemm(g, st))) return rc;
            if ((rc = launch_ln_rows(AO, X, P + e.n2g, P + e.n2b, rows, D, st, Y, X))) return rc;
            PT_CHECK_LAUNCH();
            continue;
        }
        g = gemm_args(Hd, ff, rows, P + e.w2, rows, D, ff, P + e.b2, Y, D);
        g.R = X;
        if ((rc = launch_gemm(g, st))) return rc;
        if ((rc = launch_ln_rows(Y, X, P + e.n2g, P + e.n2b, rows, D, st))) return rc;
        PT_CHECK_LAUNCH();
    }
    // ---- decoder (transformer.py:224-238), one query per batch row; T = LayerNorm(pre) is applied by the consumers
    // (base + cv.zero was cleared by k_tomp_tokens)
    DecAttnArgs da{};
    da.mem = X; da.pos = pos; da.qk = base + cv.qk; da.scores = base + cv.scores; da.ctx = base + cv.ctx;
    da.B = B; da.L = L; da.D = D; da.nhead = NH; da.HW = HW;
    for (int b = 0; b < 8; ++b) { da.mlo[b] = at.mlo[b]; da.mhi[b] = at.mhi[b]; }
    const float* qpos = P + po.fg;                                 // query_embed_fg_decoder IS query_embed_fg (:35)
    const float* tpre = base + cv.zero;                            // pre-LayerNorm state entering the layer
    const float *tg = nullptr, *tb = nullptr;                      // its LayerNorm (none for the initial zeros)
    const PrepLayout pl = prep_layout(D, NH, d->n_dec);
    for (int i = 0; i < d->n_dec; ++i) {
        const DecOff& dc = po.dec[i];
        const PrepOff& pp = pl.dec[i];
        float *Pa = base + cv.P1, *Pb = base + cv.P2, *Pc = base + cv.P[i & 1];
        GemvArgs v{};
        // self-attention over one token (softmax of a single key is 1), folded: Pa = t + (Wo Wv) t + (Wo bv + bo)
        v.Wt = prepared + pp.wsa; v.bias = prepared + pp.bsa; v.N = D; v.K = D; v.x = tpre; v.xg = tg; v.xb = tb;
        v.res = tpre; v.rg = tg; v.rb = tb; v.B = B; v.out = Pa;
        if (!(i == 0 && dec0_const) && (rc = launch_gemv(v, st))) return rc;        // layer 0: written by k_tomp_tokens
        // cross-attention query folded onto the keys: qk[b][h] = Wk_h^T (Wq_h (LN1(Pa) + query_pos) + bq_h) / sqrt(HD)
        v = GemvArgs{};
        v.Wt = prepared + pp.mq; v.bias = prepared + pp.cq; v.N = NH * D; v.K = D; v.x = Pa; v.xg = P + dc.n1g;
        v.xb = P + dc.n1b; v.xadd = qpos; v.B = B; v.out = base + cv.qk;     // one query_pos row for all batch rows
        if (!(i == 0 && dec0_const) && (rc = launch_gemv(v, st))) return rc;
        hipLaunchKernelGGL(k_dec_scores, dim3((L + 15) / 16, B), dim3(64), 0, st, da);
        PT_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_dec_ctx, dim3(D / 16, B), dim3(1024),
                           ((size_t)NH * ((L + 3) & ~3) + 16 + 16 * 256) * sizeof(float), st, da);
        PT_CHECK_LAUNCH();
        // value projection + out_proj folded over the heads' attention-weighted memories, residual LN1(Pa)
        v = GemvArgs{};
        v.Wt = prepared + pp.wov; v.bias = prepared + pp.bov; v.N = D; v.K = NH * D; v.x = base + cv.ctx;
        v.res = Pa; v.rg = P + dc.n1g; v.rb = P + dc.n1b; v.B = B; v.out = Pb;
        if ((rc = launch_gemv(v, st))) return rc;
        // feed-forward on LN2(Pb)
        v = GemvArgs{};
        v.Wt = P + dc.w1; v.bias = P + dc.b1; v.N = ff; v.K = D; v.x = Pb; v.xg = P + dc.n2g;
        v.xb = P + dc.n2b; v.relu = 1; v.B = B; v.out = base + cv.hdn;
        if ((rc = launch_gemv(v, st))) return rc;
        v = GemvArgs{};
        v.Wt = P + dc.w2; v.bias = P + dc.b2; v.N = D; v.K = ff; v.x = base + cv.hdn;
        v.res = Pb; v.rg = P + dc.n2g; v.rb = P + dc.n2b; v.B = B; v.out = Pc;
        if ((rc = launch_gemv(v, st))) return rc;
        tpre = Pc; tg = P + dc.n3g; tb = P + dc.n3b;
    }
    // norm3 of the last layer, then the decoder's final norm (transformer.py:141-142)
    if (d->n_dec > 0 && ln_rows_wide_ok(tpre, filters, tg, tb, D)) {
        if ((rc = launch_ln_rows(tpre, filters, tg, tb, B, D, st, nullptr, nullptr, P + po.dng, P + po.dnb))) return rc;   // both norms, one launch
        PT_CHECK_LAUNCH();
    } else {
        if (d->n_dec > 0) {
            if ((rc = launch_ln_rows(tpre, base + cv.a, tg, tb, B, D, st))) return rc;
            PT_CHECK_LAUNCH();
            tpre = base + cv.a;
        }
        if ((rc = launch_ln_rows(tpre, filters, P + po.dng, P + po.dnb, B, D, st))) return rc;
        PT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_tokens_to_nchw, dim3((HW + 31) / 32, (D + 31) / 32, B), dim3(256), 0, st, X, enc_feat, B, L, D,
                       HW, Ltr);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

// ---- DenseBoxRegressor (heads.py:119-141) ------------------------------------------------------------------------
// reg pack: linear.weight (D,D), linear.bias (D), 4 x [conv weight (D, 9, D) tap-major, conv bias (D), GroupNorm weight
// (D), GroupNorm bias (D)], bbreg_layer weight (4, 9, D), bbreg_layer bias (4)
namespace {
struct RegOff { size_t lw, lb, cw[4], cb[4], gg[4], gb[4], fw, fb, total; };
RegOff reg_layout(int D) {
    RegOff o{};
    size_t c = 0;
    auto take = [&](size_t n) { size_t r = c; c += n; return r; };
    o.lw = take((size_t)D * D); o.lb = take(D);
    for (int i = 0; i < 4; ++i) { o.cw[i] = take((size_t)D * 9 * D); o.cb[i] = take(D); o.gg[i] = take(D); o.gb[i] = take(D); }
    o.fw = take((size_t)4 * 9 * D); o.fb = take(4);
    o.total = c;
    return o;
}
}  // namespace

extern "C" size_t pt_tomp_bbreg_param_floats(int d_model) {
    return d_model > 0 && d_model % 32 == 0 ? reg_layout(d_model).total : 0;
}

namespace {
struct RegCarve { size_t fproj, T0, T1, part, stats, total; int slices; };
RegCarve reg_carve(int n, int D, int H, int W) {
    RegCarve c{};
    const size_t M = (size_t)n * H * W;
    size_t o = 0;
    auto take = [&](size_t k) { size_t r = o; o += pt_align_floats(k); return r; };
    c.slices = (int)(((size_t)H * W * D + GN_SLICE - 1) / GN_SLICE);
    c.fproj = take(D); c.T0 = take(M * D); c.T1 = take(M * D); c.part = take(9 * M * D);
    c.stats = take((size_t)2 * n * c.slices);
    c.total = o;
    return c;
}
}  // namespace

extern "C" size_t pt_tomp_bbreg_ws_bytes(int n, int d_model, int H, int W) {
    if (n <= 0 || d_model <= 0 || d_model % 64 != 0 || d_model > 512 || H <= 0 || W <= 0) return 0;
    return reg_carve(n, d_model, H, W).total * sizeof(float);
}

extern "C" int pt_tomp_bbreg_f32(const float* params, const float* feat, const float* filter, float* ltrb, int n,
                                 int d_model, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (!params || !feat || !filter || !ltrb || !ws) return PT_ERR_NULL;
    if (n <= 0 || d_model <= 0 || H <= 0 || W <= 0) return PT_ERR_SHAPE;
    if (d_model % 64 != 0 || d_model > 512) return PT_ERR_UNSUPPORTED;
    const int D = d_model, HW = H * W, M = n * HW;
    const RegCarve cv = reg_carve(n, D, H, W);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const RegOff ro = reg_layout(D);
    float* base = (float*)ws;
    float *fproj = base + cv.fproj, *part = base + cv.part, *stats = base + cv.stats;
    GemvArgs v{};
    v.Wt = params + ro.lw; v.bias = params + ro.lb; v.N = D; v.K = D; v.x = filter; v.B = 1; v.out = fproj;
    int rc = launch_gemv(v, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_reg_attend, dim3((HW + 31) / 32, n), dim3(256), ((size_t)D * 33 + 32) * sizeof(float), st, feat,
                       fproj, base + cv.T0, D, HW);
    PT_CHECK_LAUNCH();
    // each 3x3 convolution = 9 per-tap partial GEMMs (one K-slice of Cin per tap) reduced by the GroupNorm statistics pass
    float *src = base + cv.T0, *dst = base + cv.T1;
    const long zs = (long)M * D;
    // Round 4: GroupNorm + ReLU of layer i are applied by layer i + 1's loader (mfma_gemm.h: gn_*): the tower is conv -> reduce per
    // layer, the activations between two layers exist only as raw sums + slice statistics (k_gn_apply stays for shapes the fused
    // loader does not take: more than GEMM_GN_MAX_SLICES slices per image, maps smaller than a 32-row tile)
    const bool fuse_gn = cv.slices <= GEMM_GN_MAX_SLICES && HW >= 32 && ((uintptr_t)(params + ro.gg[0]) % 16) == 0 &&
                         ((uintptr_t)(params + ro.gb[0]) % 16) == 0 && (D % 4) == 0;
    auto with_gn = [&](GemmArgs& g, int prev) {
        if (!fuse_gn || prev < 0) return;
        g.gn_stats = stats; g.gn_slices = cv.slices; g.gn_count = HW * D;
        g.gn_gam = params + ro.gg[prev]; g.gn_bet = params + ro.gb[prev];
    };
    for (int i = 0; i < 4; ++i) {
        GemmArgs g = gemm_args(src, D, M, params + ro.cw[i], M, D, 9 * D, params + ro.cb[i], part, D);
        g.H = H; g.Wd = W; g.Cin = D; g.HW = HW; g.ksteps = D / 64; g.c_zstride = zs;
        with_gn(g, i - 1);
        if ((rc = launch_gemm(g, st, true))) return rc;
        hipLaunchKernelGGL(k_gn_reduce, dim3(cv.slices, n), dim3(256), 0, st, part, 9, zs, dst, stats, HW * D);
        PT_CHECK_LAUNCH();
        if (!fuse_gn) {
            hipLaunchKernelGGL(k_gn_apply, dim3(cv.slices, n), dim3(256), 0, st, dst, stats, params + ro.gg[i],
                               params + ro.gb[i], D, HW * D);
            PT_CHECK_LAUNCH();
        }
        std::swap(src, dst);
    }
    GemmArgs g = gemm_args(src, D, M, params + ro.fw, M, 4, 9 * D, params + ro.fb, part, 4);
    g.H = H; g.Wd = W; g.Cin = D; g.HW = HW; g.ksteps = D / 64; g.c_zstride = (long)M * 4;
    with_gn(g, 3);
    if ((rc = launch_gemm(g, st, true))) return rc;
    hipLaunchKernelGGL(k_reg_finish, dim3((M * 4 + 255) / 256), dim3(256), 0, st, part, 9, (long)M * 4, ltrb, M, HW);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

// ==================================================================================================================
// Classification-feature head: `residual_bottleneck(num_blocks=0, final_conv=True,
// l2norm=True)` = Conv2d(Cin, Cout, 3, padding=1, bias=False) + InstanceL2Norm (ltr/models/target_classifier/
// features.py:49-73, ltr/models/layers/normalization.py:15-20).  DiMP-50 runs it once per frame (1024 -> 512), ToMP on
// the test frame and on both memory frames every frame (1024 -> 256, tomp.py:289-290).
//   NCHW -> token-major transpose, 9 per-tap partial GEMMs on the MFMA kernel above (split-K), fixed-order reduction
//   with the per-image sum of squares, normalise + transpose back to NCHW.
// ==================================================================================================================
namespace {

__global__ __launch_bounds__(256) void k_nchw_to_tokens(const float* in, float* X, int C, int HW) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, img = blockIdx.z;
    const float* src = in + (long)img * C * HW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, p = p0 + tx;
        tile[ty + 8 * i][tx] = (c < C && p < HW) ? src[(long)c * HW + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + ty + 8 * i, c = c0 + tx;
        if (p < HW && c < C) X[((long)img * HW + p) * C + c] = tile[tx][ty + 8 * i];
    }
}

// out[img][c][p] = x[img*HW + p][c] * scale * sqrt(C*HW / (sum x^2 + eps)); the per-image sum of squares comes from the
// slice partials of k_gn_reduce (double, fixed order)
__global__ __launch_bounds__(256) void k_head_finish(const float* X, const float* stats, int slices, float* out, int C,
                                                     int HW, float scale, float eps) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, img = blockIdx.z;
    __shared__ double sh[256];
    double q = 0.0;                                                 // fixed order: thread t sums slices t, t+256, ...; then a tree
    for (int k = threadIdx.x; k < slices; k += 256) q += (double)stats[((long)img * slices + k) * 2 + 1];
    sh[threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    const float f = scale * (float)sqrt((double)C * HW / (sh[0] + (double)eps));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (p < HW && c < C) ? X[((long)img * HW + p) * C + c] * f : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, p = p0 + tx;
        if (c < C && p < HW) out[((long)img * C + c) * HW + p] = tile[tx][ty + 8 * i];
    }
}

// out = y * scale * sqrt(C*HW / (sum y^2 + eps)) for maps that already are (n, C, H, W)
// slot_dyn (graph-replayed one-call frame, frame_full.hip): the output lands slot_stride * *slot_dyn floats further -- the memory slot of
// this frame is read from a device descriptor instead of being baked into the captured launch
__global__ __launch_bounds__(256) void k_head_scale(const float* __restrict__ Y, const float* __restrict__ stats, int slices,
                                                    float* __restrict__ out, int C, int HW, float scale, float eps,
                                                    const int* __restrict__ slot_dyn, long slot_stride) {
    __shared__ double sh[256];
    if (slot_dyn) out += (long)(*slot_dyn) * slot_stride;
    const int img = blockIdx.y;
    double q = 0.0;                                                 // fixed order: thread t sums slices t, t+256, ...; then a tree
    for (int k = threadIdx.x; k < slices; k += 256) q += (double)stats[((long)img * slices + k) * 2 + 1];
    sh[threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    const float f = scale * (float)sqrt((double)C * HW / (sh[0] + (double)eps));
    const long base = (long)img * C * HW, i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i < (long)C * HW) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Y + base + i);
        *reinterpret_cast<f32x4*>(out + base + i) = v * f;
    }
}

struct HeadCarve { size_t X, part, Y, stats, total; int slices; };
HeadCarve head_carve(int n, int Cin, int Cout, int H, int W) {
    HeadCarve c{};
    const size_t M = (size_t)n * H * W;
    size_t o = 0;
    auto take = [&](size_t k) { size_t r = o; o += pt_align_floats(k); return r; };
    c.slices = (int)(((size_t)H * W * Cout + GN_SLICE - 1) / GN_SLICE);
    c.X = take(M * Cin); c.part = take(9 * M * Cout); c.Y = take(M * Cout); c.stats = take((size_t)2 * n * c.slices);
    c.total = o;
    return c;
}

int head_check(int n, int Cin, int Cout, int H, int W) {
    if (n <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return PT_ERR_SHAPE;
    if (Cin % 64 != 0 || Cout % 4 != 0 || (size_t)n * H * W * Cin * 4 > 0xFFFFFFE0ull ||
        (size_t)Cout * 9 * Cin * 4 > 0xFFFFFFE0ull)
        return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

}  // namespace

extern "C" size_t pt_clf_head_ws_bytes(int n, int Cin, int Cout, int H, int W) {
    if (head_check(n, Cin, Cout, H, W)) return 0;
    return head_carve(n, Cin, Cout, H, W).total * sizeof(float);
}

extern "C" int pt_clf_head_f32(const float* feat, const float* weight_tap_major, float* out, int n, int Cin, int Cout,
                               int H, int W, float norm_scale, float eps, void* ws, size_t ws_bytes, void* stream) {
    return pt_clf_head_impl(feat, weight_tap_major, out, n, Cin, Cout, H, W, norm_scale, eps, ws, ws_bytes, stream, nullptr, 0);
}

// slot_dyn != nullptr (one frame, banded-correlation route only): `out` is the BASE of the sample memory and the slot comes from the device
int pt_clf_head_impl(const float* feat, const float* weight_tap_major, float* out, int n, int Cin, int Cout, int H, int W,
                     float norm_scale, float eps, void* ws, size_t ws_bytes, void* stream, const int* slot_dyn, long slot_stride) {
    if (!feat || !weight_tap_major || !out || !ws) return PT_ERR_NULL;
    int rc = head_check(n, Cin, Cout, H, W);
    if (rc) return rc;
    const HeadCarve cv = head_carve(n, Cin, Cout, H, W);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* base = (float*)ws;
    const int HW = H * W, M = n * HW;
    // The maps stay (n, C, H, W) end to end when the banded correlation kernel covers the shape (csrc/mf_kernels.hip: 16 output
    // channels per workgroup, weights transposed on their way into LDS, channel splits summed here): no token transposes,
    // 3 launches instead of 4, and the MFMA loop built on the issue model (66.6 -> 4x us on the DiMP-50 head).
    // (one frame of DiMP-50: 256 workgroups x 16 chunks; ToMP's three frames would need 32 chunk times and stay on the
    // GEMM path below, 76 vs 90 us)
    const int ks = ((Cout * HW) % 4 == 0 && ((uintptr_t)out % 16) == 0 && pt_mf_corr_tm_cost(n, Cout, Cin, H, W) <= 22)
                       ? pt_mf_corr_tm_splits(n, Cout, Cin, H, W) : 0;
    if (slot_dyn && !(ks > 0 && ks <= 9 && n == 1)) return PT_ERR_UNSUPPORTED;
    if (ks > 0 && ks <= 9) {
        if ((rc = pt_launch_mf_corr_tm(feat, (long)Cin * HW, weight_tap_major, base + cv.part, n, Cout, Cin, H, W, ks, st)) == PT_OK) {
            hipLaunchKernelGGL(k_gn_reduce, dim3(cv.slices, n), dim3(256), 0, st, base + cv.part, ks, (long)M * Cout, base + cv.Y,
                               base + cv.stats, HW * Cout);
            PT_CHECK_LAUNCH();
            hipLaunchKernelGGL(k_head_scale, dim3((Cout * HW / 4 + 255) / 256, n), dim3(256), 0, st, base + cv.Y, base + cv.stats,
                               cv.slices, out, Cout, HW, norm_scale, eps, slot_dyn, slot_stride);
            PT_CHECK_LAUNCH();
            return PT_OK;
        }
    }
    if (slot_dyn) return PT_ERR_UNSUPPORTED;                           // the token route writes through k_head_finish: static slot only
    hipLaunchKernelGGL(k_nchw_to_tokens, dim3((HW + 31) / 32, (Cin + 31) / 32, n), dim3(256), 0, st, feat, base + cv.X, Cin,
                       HW);
    PT_CHECK_LAUNCH();
    const long zs = (long)M * Cout;
    GemmArgs g = gemm_args(base + cv.X, Cin, M, weight_tap_major, M, Cout, 9 * Cin, nullptr, base + cv.part, Cout);
    g.H = H; g.Wd = W; g.Cin = Cin; g.HW = HW; g.ksteps = Cin / 64; g.c_zstride = zs;
    if ((rc = launch_gemm(g, st, true))) return rc;
    hipLaunchKernelGGL(k_gn_reduce, dim3(cv.slices, n), dim3(256), 0, st, base + cv.part, 9, zs, base + cv.Y,
                       base + cv.stats, HW * Cout);
    PT_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_head_finish, dim3((HW + 31) / 32, (Cout + 31) / 32, n), dim3(256), 0, st, base + cv.Y,
                       base + cv.stats, cv.slices, out, Cout, HW, norm_scale, eps);
    PT_CHECK_LAUNCH();
    return PT_OK;
}
