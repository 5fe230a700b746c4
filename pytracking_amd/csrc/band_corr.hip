// Position-band correlation pass of the steepest-descent solvers: apply_filter (ltr/models/layers/filter.py:5-57) with
// ALL channels of a sample inside one workgroup, so that the score map it produces is complete and everything that
// follows per position can ride on the pass:
//     F g -> per-sample curvature terms (optimizer.py:151-156 / :416-422) -> packed operands of the next update stage.
// With the channel-range split of k_corr2 the curvature term q_i = sum_o (sws * der * (F g)[o])^2 is not additive over
// workgroups (each holds a partial F g), which forced a pointwise launch (k_fast_sgq) after every correlation; with a
// position-band split it is a plain sum over the bands of a sample, folded into the next adjoint launch's prologue.
//
//   workgroup = (sample i, band b of OUTPUT rows [out0, out1)).  It needs the input rows [out0 - KH/2, out1 + KH - 1 - KH/2)
//               of all C channels: neighbouring bands overlap by KH - 1 rows (18x18, K = 4, 4 bands: 27 rows read for
//               18 -> 1.5x through L2; the bands of a sample run on ONE XCD (sample i on XCD i % 8), so the overlap is an
//               L2 hit and fabric traffic stays 1x).
//   waves     = (tile t of 64 positions) x (quarter q of the k-steps): T_q[tap][pos] = sum_{c in quarter} g[c][tap] feat[c][pos]
//               on v_mfma_f32_16x16x4_f32; the four quarters are summed through LDS in a fixed tree, then the 16-tap
//               shift-and-add gives the band's rows of F g.
//   prologue  = g = sum_k gpart[k] + reg * w_t for ALL channels (optimizer.py:146-148), |g|^2.
//
// STATUS (round 3): correct (same goldens as the default path, tests/test_gpu_parity.py::test_band_correlation_variant_golden)
// but NOT the default.  Ablation on the MI355X (profiles/r03c_band_corr_experiment.txt): 15.2 us per launch = 5.1 us of
// launch shell + 3.0 (gradient reduction) + 3.0 (feature loads) + 2.7 (MFMA) + 2.0 (quarter sum, shift-and-add, tail),
// additive: each of the 200 workgroups owns a CU and has to take in 288 KB of gradient partials + 258 KB of features,
// i.e. >= 3.5 us at the CU's 64 B/clk before the first MFMA can retire, and nothing overlaps inside a single-workgroup CU.
// k_corr2 + k_fast_sgq spread 46 MB over 256 CUs and take 10.1 + 4.9 us.  Selected with PT_SD_BAND=1.
#include "common.h"
#include "pt_internal.h"
#include "sd_common.h"

PtBand pt_band_plan(const PtFast& f) {
    PtBand p;
    p.ok = 0;
    if (!f.ok) return p;
    const int ph = f.KH / 2, tail = f.KH - 1 - ph;
    p.SPX = pt_ceil_div(f.n, 8);
    // all bands of the samples of one XCD on distinct CUs (32 per XCD); >= 4 bands so that 50 samples fill the chip
    int B = 32 / p.SPX;
    if (B < 4) B = 4;
    if (B > 8) B = 8;
    if (B > f.OH) B = f.OH;
    for (; B <= 8 && B <= f.OH; ++B) {
        // smallest band height (in input rows) M for which a greedy cut needs <= B bands
        for (int M = 1; M <= f.H; ++M) {
            int nb = 0, a = 0;
            bool fits = true;
            while (a < f.OH && nb < 8) {
                const int in0 = a - ph > 0 ? a - ph : 0;
                int b = in0 + M - tail;                       // last output row + 1 whose inputs end at in0 + M
                if (in0 + M >= f.H) b = f.OH;                 // the band reaches the bottom of the map
                if (b > f.OH) b = f.OH;
                if (b <= a) { fits = false; break; }
                p.out0[nb] = a; p.out1[nb] = b;
                p.in0[nb] = in0;
                p.in1[nb] = b + tail < f.H ? b + tail : f.H;
                a = b;
                ++nb;
            }
            if (!fits || a < f.OH || nb > B) continue;
            int cnt = 0, nout = 0;
            for (int k = 0; k < nb; ++k) {
                const int p0a = (p.in0[k] * f.W) & ~3;
                int p1a = (p.in1[k] * f.W + 3) & ~3;
                if (p1a > f.HW) p1a = f.HW;
                cnt = cnt > p1a - p0a ? cnt : p1a - p0a;
                const int no = (p.out1[k] - p.out0[k]) * f.OW;
                nout = nout > no ? nout : no;
            }
            const int TL = pt_ceil_div(cnt, 64);
            if (TL > 3) break;                               // larger M only grows it: try more bands
            p.B = nb;
            p.TL = TL < 2 ? 2 : TL;
            p.NKW = f.C / 16;
            p.threads = 256 * p.TL;
            p.TP = 64 * p.TL + 4;
            if (nout > p.threads) break;
            p.lds = ((size_t)f.C * 16 + (size_t)3 * 16 * p.TP) * sizeof(float);
            if (p.lds > 150 * 1024) return p;
            p.ok = 1;
            return p;
        }
    }
    return p;
}

struct Corr3Args {
    const float* feat; long stride_n; const float* filt; float* s_out;
    int n, C, H, W, KH, KW, OH, OW, B, TP;
    int in0[8], in1[8], out0[8], out1[8];
    // fused gradient reduction (optimizer.py:146-148): filter operand = sum_k gpart[k] + reg*w
    const float* gpart; int KSPL; const float* w; float reg; float* g_out; float* anum;
    // source override: sample `slot` is read from `src` (C,H,W) and stored to copy_dst (the memory slot)
    int slot; const float* src; float* copy_dst;
    // pointwise tail (fused form only)
    int kind, score_act, has_sw; float act_param;
    const float *s_cur, *lms, *label, *P, *sw;
    float *sg, *pk, *qs;
};

// NKW: k-steps (4 channels each) per wave = C / 16.  TL: 64-position tiles per band.  FUSE = 0: filter operand read from
// `filt`, the band's scores go to s_out.  FUSE = 8 / 16: operand = sum of <= FUSE gradient partials + reg*w, pointwise tail.
template <int NKW, int TL, int FUSE>
__global__ __launch_bounds__(256 * TL) void k_corr3(Corr3Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // g[C][16] | T[3][16][TP]
    __shared__ float scratch[16], scratch2[16];
    constexpr int NT = 256 * TL;
    constexpr int FP = FUSE > 0 ? FUSE : 1;
    const int blk = blockIdx.x, x = blk & 7, sl = blk >> 3;
    const int i = 8 * (sl / a.B) + x, band = sl - (sl / a.B) * a.B;
    if (i >= a.n) return;                                           // uniform per workgroup
    const int HW = a.H * a.W, KK = a.KH * a.KW, OO = a.OH * a.OW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, j = lane & 15;
    const int q = wave & 3, t = wave >> 2;
    const int in0 = a.in0[band], in1 = a.in1[band], out0 = a.out0[band], out1 = a.out1[band];
    const int p0a = (in0 * a.W) & ~3;
    const int p1a = min(HW, (in1 * a.W + 3) & ~3);
    float* __restrict__ gl = lds;
    float* __restrict__ T0 = lds + a.C * 16;
    const bool over = a.src != nullptr && i == a.slot;
    const float* __restrict__ fi = over ? a.src : a.feat + (long)i * a.stride_n;
    const bool publish = FUSE > 0 && i == 0 && band == 0;

    // ---- pointwise operands of this thread's output element (written by the previous launch): requested first
    const int nout = (out1 - out0) * a.OW;
    const int ol = min((int)threadIdx.x, nout - 1);
    const long eo = (long)i * OO + (long)out0 * a.OW + ol;
    float sv = 0.f, Pv = 0.f, Lv = 0.f;
    f32x4 lm = {0, 0, 0, 0};
    if (FUSE > 0) {
        sv = a.s_cur[eo];
        if (a.kind == PT_SD_PRDIMP) { Pv = a.P[eo]; Lv = a.label[eo]; }
        else lm = ((const f32x4*)a.lms)[eo];
    }

    // ---- feature slice of this wave: NKW float4.  The first CD k-steps are requested before the filter is staged, the
    //      rest CD k-steps ahead of the MFMAs that consume them.
    constexpr int CDM = TL == 2 ? 32 : 16;                          // 8 waves: 256 VGPRs per lane; 12 waves: 168
    constexpr int CD = NKW < CDM ? NKW : CDM;                       // register ring: k-steps in flight (all of them up to C = 512)
    constexpr int PRE = CD < 8 ? CD : 8;                            // requested before the filter operand is staged
    const int cbase = 4 * (q * NKW) + kq;
    const int pl = 64 * t + 4 * j;                                  // position inside the band
    const bool pv = p0a + pl < p1a;
    const __amdgpu_buffer_rsrc_t fr = pt_rsrc(fi, (unsigned)a.C * HW * 4u);
    const unsigned fo = pv ? ((unsigned)cbase * HW + (unsigned)(p0a + pl)) * 4u : 0x80000000u;   // no position: reads 0
    f32x4 bq[CD];
    auto ldq = [&](int k) { bq[k % CD] = pt_bload4(fr, fo + (unsigned)(4 * k) * HW * 4u); };
#pragma unroll
    for (int k = 0; k < PRE; ++k) ldq(k);

    // ---- filter operand for all channels -> LDS [c][16 taps]
    float gsq = 0.f;
    if (KK == 16) {
        constexpr int R4 = (64 * NKW + NT - 1) / NT;                // float4 pieces per thread
        const int n4 = a.C * 4;
        if (FUSE > 0) {
            const unsigned ckk_b = (unsigned)a.C * 16u * 4u;
            const __amdgpu_buffer_rsrc_t rg = pt_rsrc(a.gpart, (unsigned)a.KSPL * ckk_b);
            // pieces reduced per batch (registers: RB * (FUSE + 1) float4): at C = 512 with <= 8 partials every load of
            // the reduction is in flight at once
            constexpr int VB = TL == 2 ? 36 : 18;
            constexpr int RBx = VB / (FP + 1) < 1 ? 1 : VB / (FP + 1);
            constexpr int RB = RBx < R4 ? RBx : R4;
#pragma unroll
            for (int r0 = 0; r0 < R4; r0 += RB) {
                f32x4 part[RB][FP], wv[RB];
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    if (r0 + rr < R4) {
                        const int ec = min((int)threadIdx.x + (r0 + rr) * NT, n4 - 1);
#pragma unroll
                        for (int k = 0; k < FP; ++k)
                            part[rr][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                rg, (unsigned)ec * 16u, (unsigned)min(k, a.KSPL - 1) * ckk_b, 0));
                        wv[rr] = ((const f32x4*)a.w)[ec];
                    }
                }
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    if (r0 + rr < R4) {
                        const int e = (int)threadIdx.x + (r0 + rr) * NT;
                        f32x4 v = {0, 0, 0, 0};
#pragma unroll
                        for (int k = 0; k < FP; ++k)
                            if (k < a.KSPL) v += part[rr][k];                       // fixed order
                        v += a.reg * wv[rr];
                        if (e < n4) {
                            gsq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                            ((f32x4*)gl)[e] = v;
                            if (publish) ((f32x4*)a.g_out)[e] = v;
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R4; ++r) {
                const int e = (int)threadIdx.x + r * NT;
                if (e < n4) ((f32x4*)gl)[e] = ((const f32x4*)a.filt)[e];
            }
        }
    } else {
        for (int e = threadIdx.x; e < a.C * 16; e += NT) {
            const int c = e >> 4, tp = e & 15;
            float v = 0.f;
            if (tp < KK) {
                const long ge = (long)c * KK + tp;
                if (FUSE > 0) {
                    const long CKK = (long)a.C * KK;
                    for (int k = 0; k < a.KSPL; ++k) v += a.gpart[(long)k * CKK + ge];
                    v += a.reg * a.w[ge];
                    gsq += v * v;
                    if (publish) a.g_out[ge] = v;
                } else {
                    v = a.filt[ge];
                }
            }
            gl[e] = v;
        }
    }
    // the rest of the wave's feature slice goes out now (the reduction's registers are free again): by the time the
    // operand is visible in LDS most of it has arrived, and the MFMA chain never waits for a load it has not yet issued
#pragma unroll
    for (int k = PRE; k < CD; ++k) ldq(k);
    if (publish) {                                                  // uniform per workgroup
        const float tot = block_sum(gsq, scratch);
        if (threadIdx.x == 0) a.anum[0] = tot;
    }
    __syncthreads();

    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    float* __restrict__ dp = (over && a.copy_dst && pv) ? a.copy_dst + (long)cbase * HW + p0a + pl : nullptr;
#pragma unroll
    for (int k = 0; k < NKW; ++k) {
        const float av = gl[(4 * (q * NKW + k) + kq) * 16 + j];
        const f32x4 b = bq[k % CD];
        if (dp) *(f32x4*)(dp + (long)(4 * k) * HW) = b;             // memory insert rides on the pass (dimp.py:429-441)
        if (k + CD < NKW) ldq(k + CD);
        acc0 = mfma16(av, b[0], acc0);
        acc1 = mfma16(av, b[1], acc1);
        acc2 = mfma16(av, b[2], acc2);
        acc3 = mfma16(av, b[3], acc3);
    }


    // ---- sum of the four k-step quarters in a fixed order (((q0 + q1) + q2) + q3): quarters 1..3 park their tiles in
    //      LDS, quarter 0 adds them and leaves the complete tap planes in T0 (cells it has just read itself)
    auto put = [&](float* __restrict__ T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 v = {acc0[r], acc1[r], acc2[r], acc3[r]};
            *(f32x4*)__builtin_assume_aligned(T + (4 * kq + r) * a.TP + pl, 16) = v;
        }
    };
    if (q > 0) put(T0 + (q - 1) * 16 * a.TP);
    __syncthreads();
    if (q == 0) {
        f32x4 v[3][4];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                v[m][r] = *(const f32x4*)__builtin_assume_aligned(T0 + m * 16 * a.TP + (4 * kq + r) * a.TP + pl, 16);
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[r] += v[m][r][0]; acc1[r] += v[m][r][1]; acc2[r] += v[m][r][2]; acc3[r] += v[m][r][3]; }
        put(T0);
    }
    __syncthreads();

    // ---- shift-and-add of the tap planes -> this band's rows of the score map; one output element per thread
    const int ph = a.KH / 2, pw = a.KW / 2;
    float s = 0.f;
    if ((int)threadIdx.x < nout) {
        const int yl = ol / a.OW, xx0 = ol - yl * a.OW, y = out0 + yl;
        if (a.KH == 4 && a.KW == 4) {
            float tv[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int yy = y + u - 2;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int xx = xx0 + v - 2;
                    const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
                    const float tvv = T0[(u * 4 + v) * a.TP + (ok ? yy * a.W + xx - p0a : 0)];
                    tv[u * 4 + v] = ok ? tvv : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) s += tv[e];
        } else {
            for (int u = 0; u < a.KH; ++u) {
                const int yy = y + u - ph;
                if ((unsigned)yy >= (unsigned)a.H) continue;
                for (int v = 0; v < a.KW; ++v) {
                    const int xx = xx0 + v - pw;
                    if ((unsigned)xx < (unsigned)a.W) s += T0[(u * a.KW + v) * a.TP + yy * a.W + xx - p0a];
                }
            }
        }
    }
    if (FUSE == 0) {
        if ((int)threadIdx.x < nout) a.s_out[eo] = s;
        return;
    }

    // ---- pointwise tail (what k_fast_sgq did per sample): F g, curvature partials of this band, packed operands
    const bool live = (int)threadIdx.x < nout;
    float acc_a = 0.f, acc_b = 0.f;
    f32x4 pkv;
    if (a.kind == PT_SD_PRDIMP) {
        acc_a = live ? Pv * s * s : 0.f;                             // sum_o P (F g)^2           (:419-420)
        acc_b = live ? Pv * s : 0.f;                                 // sum_o P (F g)
        pkv = (f32x4){sv, s, Lv, 0.f};
    } else {
        const int sact = a.kind == PT_SD_DIMP_L2 ? 2 : a.score_act;
        float act, der;
        act_pair(sact, a.act_param, sv, lm[1], act, der);
        const float qv = lm[2] * (der * s);                          // :151-152
        acc_a = live ? qv * qv : 0.f;
        if (a.kind == PT_SD_DIMP && a.score_act == PT_ACT_BENTPAR) pkv = (f32x4){sv, s, lm[0], lm[1]};
        else { const float w2 = lm[2] * lm[2]; pkv = (f32x4){w2 * sv, w2 * s, w2 * lm[0], lm[1]}; }
    }
    if (live) {
        a.sg[eo] = s;
        ((f32x4*)a.pk)[eo] = pkv;
    }
    // both band sums through one pair of barriers, fixed order (waves, then wave index)
    acc_a = wave_sum(acc_a);
    acc_b = wave_sum(acc_b);
    if (lane == 0) { scratch[wave] = acc_a; scratch2[wave] = acc_b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
        for (int w = 0; w < 4 * TL; ++w) { ta += scratch[w]; tb += scratch2[w]; }
        float* qo = a.qs + ((long)i * a.B + band) * 2;
        qo[0] = ta;
        qo[1] = tb;
    }
}

int pt_launch_corr3(const PtFast& f, const PtBand& p, const float* feat, long stride_n, const float* filt, float* s_out,
                    hipStream_t st, const PtCorrFuse* fuse, const SdArgs* sd, int slot, const float* src, float* copy_dst) {
    Corr3Args a;
    a.feat = feat; a.stride_n = stride_n; a.filt = filt; a.s_out = s_out;
    a.n = f.n; a.C = f.C; a.H = f.H; a.W = f.W; a.KH = f.KH; a.KW = f.KW; a.OH = f.OH; a.OW = f.OW; a.B = p.B; a.TP = p.TP;
    for (int k = 0; k < 8; ++k) { a.in0[k] = p.in0[k]; a.in1[k] = p.in1[k]; a.out0[k] = p.out0[k]; a.out1[k] = p.out1[k]; }
    a.gpart = nullptr; a.KSPL = 0; a.w = nullptr; a.reg = 0.f; a.g_out = nullptr; a.anum = nullptr;
    a.slot = slot; a.src = src; a.copy_dst = copy_dst;
    a.kind = 0; a.score_act = 0; a.has_sw = 0; a.act_param = 0.f;
    a.s_cur = a.lms = a.label = a.P = a.sw = nullptr; a.sg = a.pk = a.qs = nullptr;
    if (fuse) {
        if (!sd) return PT_ERR_NULL;
        a.gpart = fuse->gpart; a.KSPL = fuse->KSPL; a.w = fuse->w; a.reg = fuse->reg; a.g_out = fuse->g_out; a.anum = fuse->anum_part;
        a.kind = sd->kind; a.score_act = sd->score_act; a.has_sw = sd->has_sw; a.act_param = sd->act_param;
        a.s_cur = sd->s; a.lms = sd->lms; a.label = sd->label; a.P = sd->mask; a.sw = sd->sw;
        a.sg = sd->sg; a.pk = sd->pk; a.qs = sd->qs;
        if (a.KSPL > 16) return PT_ERR_UNSUPPORTED;
    }
    if (((uintptr_t)feat % 16) || (stride_n % 4) || ((uintptr_t)src % 16) || ((uintptr_t)copy_dst % 16)) return PT_ERR_UNSUPPORTED;
    if ((long)f.n * stride_n * 4 >= (1L << 31)) return PT_ERR_UNSUPPORTED;
    if (f.KK == 16 && (((uintptr_t)filt % 16) || ((uintptr_t)a.gpart % 16) || ((uintptr_t)a.w % 16) || ((uintptr_t)a.g_out % 16)))
        return PT_ERR_UNSUPPORTED;
    dim3 grid(8 * p.SPX * p.B), block(p.threads);
#define PT_C3F(NKWV, TLV)                                                                                       \
    do {                                                                                                        \
        if (!a.gpart) hipLaunchKernelGGL((k_corr3<NKWV, TLV, 0>), grid, block, p.lds, st, a);                   \
        else if (a.KSPL <= 8) hipLaunchKernelGGL((k_corr3<NKWV, TLV, 8>), grid, block, p.lds, st, a);           \
        else hipLaunchKernelGGL((k_corr3<NKWV, TLV, 16>), grid, block, p.lds, st, a);                           \
    } while (0)
#define PT_C3(NKWV)                       \
    do {                                  \
        if (p.TL == 2) PT_C3F(NKWV, 2);   \
        else PT_C3F(NKWV, 3);             \
    } while (0)
    if (p.NKW == 8) PT_C3(8);
    else if (p.NKW == 16) PT_C3(16);
    else if (p.NKW == 32) PT_C3(32);
    else if (p.NKW == 64) PT_C3(64);
    else return PT_ERR_UNSUPPORTED;
#undef PT_C3F
#undef PT_C3
    PT_CHECK_LAUNCH();
    return PT_OK;
}
