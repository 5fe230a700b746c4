// XCD-aligned feature passes for the common tracker shapes (C in {128,256,512,1024}, H*W % 4 == 0).
//
// Both passes stream the whole sample memory once.  MI355X has 8 XCDs with a private 4 MiB L2 each and dispatches
// workgroup b to XCD b % 8 (observed placement, used for speed only -- nothing below depends on it for correctness).
// Channel range [x*C/8, (x+1)*C/8) of EVERY sample is always processed by workgroups with b % 8 == x, in the
// correlation pass and in the adjoint pass alike, so consecutive passes of a solve re-read the eighth of the memory
// their XCD already holds in L2 (DiMP-50: 4.15 MB per XCD; profiles/r01_l2_probe.txt: 3.7 us per aligned pass vs
// 5.9 us when the ownership rotates).
//
//   k_corr2 : apply_filter (ltr/models/layers/filter.py:5-57).  Workgroup = (sample i, XCD x); its waves are
//             (64-position tile t, half h of the XCD's k-steps).  T[h][tap][pos] = sum_c filt[c][tap]*feat[i][c][pos]
//             on v_mfma_f32_16x16x4_f32, then the 16-tap shift-and-add from LDS -> spart[x][i][OH*OW].
//             A trailing <= 4 quads (18x18: positions 320..323) use ONE MFMA per k-step with the positions as the
//             16 columns instead of a mostly empty 64-position tile.
//   k_adj2  : apply_feat_transpose (filter.py:91-182).  Workgroup = (16-channel block cb of XCD x, position slice ks);
//             wave w owns U contiguous 16-position groups.  G[c][tap] = sum_P feat[c][P] * r[P shifted by tap]; the B
//             operand is gathered from zero-padded residual maps the workgroup builds in LDS -- there is no im2col
//             buffer in HBM.  In the solver the maps come from the fused update prologue (alpha, s_{t}, residual).
#include <type_traits>
#include "common.h"
#include "pt_internal.h"
#include "sd_common.h"

typedef float f32x2a4 __attribute__((ext_vector_type(2), aligned(4)));   // 8-byte LDS access at 4-byte alignment (ds_read_b64)

#ifndef PT_ADJ_WAVES
#define PT_ADJ_WAVES 8                     // waves per k_adj2 workgroup (16 measured no better at 18x18, worse at 22x22)
#endif
#ifndef PT_ADJ_UMAX
#define PT_ADJ_UMAX (128 / PT_ADJ_WAVES)   // 16-position groups per wave of the common instantiations
#endif
// 22x22 / 23x23 maps (PrDiMP-50: E == 9): up to 24 groups per wave, so that n = 50 is EIGHT position slices = 256 workgroups, one
// per CU, instead of sixteen = 512 (round 4: k_adj2 15.1 -> see profiles/HISTORY.md section 7; the correlation's prologue then sums 8 gradient
// partials instead of 16).  The loads are pipelined PD groups ahead, so the register cost of a longer run is the unrolled loop only.
#ifndef PT_ADJ_UMAX_WIDE
#define PT_ADJ_UMAX_WIDE (192 / PT_ADJ_WAVES)
#endif
// experiment knobs (profiles/r03f_*): feature loads requested ahead of the MFMAs / register cap of k_corr2
#ifndef PT_C2_CD
#define PT_C2_CD 2       // round 3 sweep (profiles/r03l_*): 2 / 3 / 4 / 6 / 8 ahead -> 8.72 / 8.97 / 8.99 / 9.49 / 9.76 us
#endif
#ifndef PT_C2_MINW
#define PT_C2_MINW 6
#endif
#ifndef PT_ADJ_PD
#define PT_ADJ_PD 3      // round 3 sweep: 3 / 4 / 6 / 8 / 12 / 16 ahead -> 8.08 / 8.36 / 8.50 / 8.67 / 8.84 / 9.0 us
#endif
#ifndef PT_C2_BAR
#define PT_C2_BAR 0      // 1: workgroup barrier between the filter-operand loads and the first feature loads
#endif
#ifndef PT_C2_PAIR
#define PT_C2_PAIR 1     // 1: sample-pair workgroups in k_corr2 where the plan allows them (pt_fast_plan); 0: round 3's one sample per workgroup
#endif
#ifndef PT_C2_R16
#define PT_C2_R16 0      // 1: 16 channel ranges -- the two k-step halves of an XCD's range as separate 5-wave workgroups
#endif
#ifndef PT_ADJ_BAR
#define PT_ADJ_BAR 0     // 1: workgroup barrier between the small update-stage loads and the first feature loads
#endif
#ifndef PT_ADJ_G2
#define PT_ADJ_G2 0      // 1: residual cells of a quad gathered as two 8-byte LDS reads at 4-byte alignment (even map widths).  Measured in
//                          round 4 and NOT kept: k_adj2 7.6 -> 12.0 us (PrDiMP 12.7 -> 17.7) -- a misaligned ds_read_b64 costs far more
//                          than the two scalar reads and two address adds it replaces; as two ds_read2_b32 (dword pairs, no alignment demand): 7.55 -> 7.94 us
//                          (profiles/r04j_adjoint_pair_gather_ab.txt) -- fewer instructions in the loop do not make it faster
#endif
#ifndef PT_C2_AV_UPFRONT
#define PT_C2_AV_UPFRONT 1 // 1: k_corr2 reads its NK filter-operand values from LDS in one batch in front of the MFMA loop (round 6)
#endif
#ifndef PT_C2_SA_DIRECT
#define PT_C2_SA_DIRECT 1 // 1: k_corr2's 4x4 shift-and-add with uniform tap steps and row / column masks (round 6); 0: per-tap index arithmetic
#endif
#ifndef PT_ADJ_EARLY
#define PT_ADJ_EARLY 1   // 1: first feature loads in front of the LDS work; 0: behind barrier 1
#endif

// Phase time stamps (experiments only, -DPT_STAMPS; tools/exp_stamps.py): every wave records the 100 MHz device wall clock at
// up to 8 points into a buffer registered with pt_debug_set_stamps(); layout [workgroup][wave (16 slots)][8].
#ifdef PT_STAMPS
static unsigned long long* g_pt_stamps = nullptr;
extern "C" void pt_debug_set_stamps(void* p) { g_pt_stamps = (unsigned long long*)p; }
#define PT_STAMP_ARG unsigned long long* stamps;
#define PT_STAMP_SET(a) (a).stamps = g_pt_stamps
#define PT_STAMP(a, k)                                                                                       \
    do {                                                                                                     \
        if ((a).stamps && (threadIdx.x & 63) == 0)                                                           \
            (a).stamps[((long)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (k)] = (unsigned long long)wall_clock64(); \
    } while (0)
#else
#define PT_STAMP_ARG
#define PT_STAMP_SET(a)
#define PT_STAMP(a, k)
#endif
// second stamp set (-DPT_STAMPS=2): the k_adj2 prologue in detail; the coarse k_adj2 stamps are then off
#if defined(PT_STAMPS) && PT_STAMPS == 2
#define PT_STAMP_A(a, k)
#define PT_STAMP_B(a, k) PT_STAMP(a, k)
#else
#define PT_STAMP_A(a, k) PT_STAMP(a, k)
#define PT_STAMP_B(a, k)
#endif

// ---------------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------------
PtFast pt_fast_plan(int n, int C, int H, int W, int KH, int KW, int OH, int OW) {
    PtFast p;
    p.ok = 0;
    p.n = n; p.C = C; p.H = H; p.W = W; p.KH = KH; p.KW = KW; p.OH = OH; p.OW = OW;
    p.HW = H * W; p.KK = KH * KW; p.OO = OH * OW;
    if ((p.HW % 4) != 0 || W < 4 || p.KK > 16) return p;
    if (C != 64 && C != 128 && C != 256 && C != 512 && C != 1024) return p;
    if ((long)n * p.HW >= (1L << 20)) return p;
    if ((long)n * C * p.HW * 4 >= (1L << 31)) return p;             // 32-bit byte offsets of the raw buffer loads
    p.Q = p.HW / 4;
    p.TF = p.Q / 16;
    p.rem = p.Q % 16;
    p.left = (p.rem > 0 && p.rem <= 4) ? 1 : 0;
    p.tiles = p.TF + ((p.rem > 0 && !p.left) ? 1 : 0);
    if (p.TF < 1 || p.tiles > 16) return p;
    // k-step halves: two waves per tile when that keeps the workgroup at <= 10 waves (two workgroups per CU, <= 80 VGPRs);
    // larger maps (22x22: 8 tiles) use one wave per tile with all of the XCD's k-steps
    p.nh = 2 * p.tiles <= 10 ? 2 : 1;
    p.CX = C / 8;
    p.KSC = 8;
    if (C == 64) { p.CX = C; p.KSC = 1; }                           // ATOM's compressed samples: a workgroup takes ALL channels of a sample
    if (PT_C2_R16 && p.nh == 2 && n > 16 && C > 64) { p.nh = 1; p.CX = C / 16; p.KSC = 16; }
    // Sample pairs (round 4).  With more samples than one workgroup per CU (8 n > 256) the pass lasted as long as a CU holding two
    // 10-wave workgroups -- two prologues, two filter reductions, their MFMA phases colliding -- while other CUs held one.  A pair
    // workgroup takes TWO samples with the same 10 waves: wave = (sample of the pair, tile), each with ALL k-steps of the XCD's
    // channel range (no k-step halves: no second tap-plane set to add), the filter operand reduced and staged once for both.
    // 8 n / 2 workgroups, at most one per CU up to n = 64.  Even n only (the pair index is formed before the argument block arrives).
    p.spw = 1;
    // Maps of 6-8 tiles (22x22) already run one wave per tile with all k-steps; their pairs (16 waves) were measured SLOWER
    // (PrDiMP-50: 11.25 vs 10.83 us, profiles/r04i_*) and keep one sample per workgroup.
    if (PT_C2_PAIR && p.nh == 2 && p.KSC == 8 && p.CX / 4 <= 16 && (n % 2) == 0 && 8 * n > 256) { p.nh = 1; p.spw = 2; }
    p.NK = p.CX / 4 / p.nh;
    if (p.NK > 16) return p;
    p.HWp = 64 * (p.TF + (p.rem > 0 ? 1 : 0)) + 4;
    p.corr_threads = p.spw * p.nh * p.tiles * 64;
    p.corr_lds = ((size_t)p.CX * 16 + (size_t)p.spw * p.nh * p.KK * p.HWp) * sizeof(float);
    // direct 4x4 shift-and-add (k_corr2): masked taps may read up to 2W+2 floats in front of the first tap plane (the filter operand is
    // there) and behind the last one (its padding, else slack added here)
    p.sa_direct = (KH == 4 && KW == 4 && p.CX * 16 >= 2 * W + 2) ? 1 : 0;
    if (p.sa_direct && p.HWp < p.HW + 2 * W + 2) p.corr_lds += (size_t)(p.HW + 2 * W + 2 - p.HWp) * sizeof(float);
    if (p.corr_lds > 150 * 1024) return p;
    p.CB = C / 16;
    p.bpx = p.CB / 8;                                               // 0: fewer channel blocks than XCDs (workgroup b: block b % CB, slice b / CB)
    p.NG = (int)(((long)n * p.HW + 15) / 16);
    p.KSPL = 0;
    // position slices: 8 or 16 (the fused gradient reduction of k_corr2 sums at most 16 partials); with few channel blocks up to
    // 64, so that CB * KSPL workgroups still fill the chip (ATOM: 4 x 64) -- such plans serve the passes, not the SD solver
    p.E = p.OO <= 384 ? 6 : (p.OO <= 576 ? 9 : 16);
    const int umax = p.E == 9 ? PT_ADJ_UMAX_WIDE : PT_ADJ_UMAX;
    for (int ks = 8; ks <= (p.CB < 8 ? 64 : 16); ks *= 2) {
        const int gper = pt_ceil_div(p.NG, ks), U = pt_ceil_div(gper, PT_ADJ_WAVES);
        if (U <= umax) { p.gper = gper; p.U = U; p.KSPL = pt_ceil_div(p.NG, gper); break; }
    }
    if (p.KSPL == 0) return p;
    p.PH = H + KH - 1;
    // padded row stride of the residual maps in LDS: >= W + KW - 1 and == 8 (mod 16), so that the 16 taps x 2 quads a
    // ds_read_b32 half-wave gathers (offsets v + u*PW + 4*kq) fall into 32 distinct banks for the usual 4x4 filter
    p.PW = W + KW - 1;
    while ((p.PW % 16) != 8) ++p.PW;
    p.ns_max = (p.gper * 16 + p.HW - 1) / p.HW + 1;
    // zero block behind the maps: a masked quad gathers zeros for every tap offset; sized so that the quad table that
    // follows is 16-byte aligned
    p.zn = (KH - 1) * p.PW + KW + 4;
    while (((p.ns_max * p.PH * p.PW + p.zn) % 4) != 0) ++p.zn;
    p.adj_lds = ((size_t)p.ns_max * p.PH * p.PW + p.zn + (size_t)5 * 4 * PT_ADJ_WAVES * umax) * sizeof(float);
    if (p.adj_lds > 96 * 1024 || p.OO > 1024) return p;
    // what the packed kernel parameters of the two passes can hold (k_corr2: h_dims / h_geo, k_adj2: h_g1 .. h_g4)
    if (H > 255 || W > 255 || n > 65535 || p.gper > 65535 || p.PH > 63 || p.PW > 63 || p.zn > 255 || p.ns_max > 63 || KH > 7 || KW > 7)
        return p;
    p.ok = 1;
    return p;
}

// exact floor(v / d) for the small non-negative integers of this file (v * (1/d) is >= 0.5/d away from an integer)
__device__ __forceinline__ int fdiv(int v, float inv_d) { return (int)(((float)v + 0.5f) * inv_d); }

// ---------------------------------------------------------------------------------------------------
// k_corr2
// ---------------------------------------------------------------------------------------------------
struct Corr2Args {                  // pointer members global-qualified: fetched late (pt_late_args), see pt_gcf in common.h
    pt_gcf feat; long stride_n; pt_gcf filt; pt_gf spart;
    int n, C, H, W, KH, KW, OH, OW, CX, TF, rem, tiles, HWp, nh, KSC;
    int sa_direct;   // 4x4 shift-and-add with uniform tap steps (the plan guarantees 2W+2 floats in front of and behind the tap planes)
    // fused gradient reduction (optimizer.py:146-148): filter operand = sum_k gpart[k] + reg*w
    pt_gcf gpart; int KSPL; pt_gcf w; float reg; float step; pt_gf g_out; pt_gf anum_part;   // FUSE < 0: reg = reg + alpha_eps, step = step length of the pending solve
    // source override: sample `slot` is read from `src` (C,H,W) and stored to copy_dst (the memory slot)
    int slot; pt_gcf src; pt_gf copy_dst;
    PT_STAMP_ARG
};

// generic (slow) form of one filter-operand element; only used for slices larger than 4 elements per thread
__device__ __forceinline__ float corr2_filter_elem(const Corr2Args& a, int KK, int cx0, int e, bool publish, float& gsq) {
    const int cl = e >> 4, tp = e & 15;
    const bool ok = tp < KK;
    const long ge = (long)(cx0 + cl) * KK + (ok ? tp : 0);
    float v;
    if (a.gpart) {
        const long CKK = (long)a.C * KK;
        v = 0.f;
        for (int k = 0; k < a.KSPL; ++k) v += a.gpart[(long)k * CKK + ge];
        v += a.reg * a.w[ge];
        if (ok) {
            gsq += v * v;
            if (publish) a.g_out[ge] = v;
        }
    } else {
        v = a.filt[ge];
    }
    return ok ? v : 0.f;
}

// FUSE = 0: filter operand read from `filt`.  FUSE = 8 / 16 / 32: operand = sum of <= FUSE gradient partials + reg*w
// (optimizer.py:146-148), every load of the reduction issued before the first wait.
// Two 10-wave workgroups per CU put up to 6 waves on one SIMD (3+3): <= 80 VGPRs for the k-half instantiations (NH == 2) of the
// common channel counts; the one-wave-per-tile ones (NH == 1: large maps, sample pairs) run at most 4 waves per SIMD.
// K16: the filter has 16 taps (4x4, the trackers' size) -- a compile-time fact, so that the prologue is ONE basic block: with
// the tap count as a run-time branch the compiler started the partial sum inside the branch and put `s_waitcnt vmcnt(8)`
// -- a whole memory round trip -- in front of the first feature load (round 3, profiles/r03g_pass_phase_stamps.txt).
// NH: k-step halves per tile (2 for 18x18 maps, 1 for 22x22) -- compile time as well: as a run-time value every tap of the
// shift-and-add became `ds_read; branch; ds_read; s_waitcnt lgkmcnt(0)`, 16 serialised LDS round trips (0.9 us of the pass).
// FUSE < 0 (round 6, 4x4 filters only): the DEFERRED last update of the previous solve.  A frame chain leaves (w_{T-1}, g_T, the operands
// of alpha_T) in the workspace instead of spending a dependent launch (k_fast_final) on w_T = w_{T-1} - step*alpha_T*g_T, whose only
// consumer is the next frame's first correlation: here the operand is formed from h_filt = w_{T-1}, h_w = g_T, alpha_T from the n
// per-sample curvature terms and the channel-range sums of g^2 (h_w + h_aux: anum, 64 floats further: qs) with the expressions of
// k_fast_final (sd_final_astep / sd_final_apply: same bits), and the workgroups of sample 0 store w_T into a.g_out = the filter.
template <int NK, bool LEFT, int FUSE, bool K16, int NH>
__global__ __launch_bounds__(1024, (NK <= 8 && NH == 2 ? PT_C2_MINW : 4)) void k_corr2(const float* h_feat, long h_stride, const float* h_filt, const float* h_w, const float* h_src, int h_slot,
                                                                              unsigned h_dims, unsigned h_geo, unsigned h_aux, Corr2Args a_arg) {
    // The h_* parameters repeat what the prologue needs to REQUEST its operands (13 dwords).  Scalar kernel parameters are
    // preloaded into SGPRs at wave launch (-amdgpu-kernarg-preload-count), the argument block `a` arrives by scalar loads about
    // 1.2 us later (measured in round 3: a kernel whose arguments are all preloaded is that much shorter, eager and in graph
    // replay alike) -- by then the filter partials and the first feature tiles are on their way.
    //   h_filt = gradient partials (FUSE > 0) or the filter (FUSE == 0);  h_dims = C << 16 | H*W;
    //   h_geo  = tiles | TF << 5 | rem << 10 | KSPL << 14 | (16 channel ranges) << 20 | (1 channel range: C = 64) << 21 |
    //            (sample pairs: workgroup b >> 3 takes samples 2 (b >> 3) and 2 (b >> 3) + 1) << 22 | (FUSE < 0: n) << 23
    //   h_aux  = FUSE < 0: float offset of the anum block from h_w
    extern __shared__ __attribute__((aligned(16))) float lds[];     // afilt[CX][16] | T[2][KK][HWp]
    __shared__ float scratch[16];
    PT_STAMP(a_arg, 0);
    constexpr int EPT = 2;
    constexpr int FP = FUSE > 0 ? FUSE : 1;
    const int hC = (int)(h_dims >> 16), HW = (int)(h_dims & 0xffffu);
    const int h_tiles = (int)(h_geo & 31u), h_TF = (int)((h_geo >> 5) & 31u), h_rem = (int)((h_geo >> 10) & 15u);
    const int h_KSPL = (int)((h_geo >> 14) & 63u);
    const bool ksc16 = ((h_geo >> 20) & 1u) != 0, ksc1 = ((h_geo >> 21) & 1u) != 0, pair = ((h_geo >> 22) & 1u) != 0;
    const int wps = NH * h_tiles;                                   // waves per sample
    const int nthreads = (pair ? 2 : 1) * wps * 64;
    const int hCX = ksc1 ? hC : (ksc16 ? hC >> 4 : hC >> 3), hHWp = 64 * (h_TF + (h_rem > 0 ? 1 : 0)) + 4;
    // 16 ranges: 2x and 2x + 1 both run on XCD x (the adjoint pass owns channels [x C/8, (x+1) C/8) there)
    const int b = blockIdx.x, xc = b & 7, q = b >> 3;
    const int x = ksc1 ? 0 : (ksc16 ? 2 * xc + (q & 1) : xc), i0 = ksc1 ? b : (ksc16 ? q >> 1 : q);
    const int KK = K16 ? 16 : a_arg.KH * a_arg.KW;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar branches
    const int kq = lane >> 4, j = lane & 15;
    const int ws = (pair && wave >= wps) ? 1 : 0, wl = wave - ws * wps;   // which sample of the pair, wave inside the sample
    const int i = pair ? 2 * i0 + ws : i0;
    const int h = wl >= h_tiles ? 1 : 0, t = wl - h * h_tiles;
    const int cx0 = hCX * x;
    const int nsl = hCX * 16;
    float* __restrict__ afilt = lds;
    float* __restrict__ Tl = lds + nsl + (long)(ws * NH + h) * KK * hHWp;
    const bool over = h_src != nullptr && i == h_slot;
    const float* __restrict__ fi = over ? h_src : h_feat + (long)i * h_stride;
    const bool publish = FUSE != 0 && i0 == 0;                      // uniform per workgroup
    const int h_n = (int)(h_geo >> 23);                             // FUSE < 0: samples of the pending solve
    float lz_q = 0.f, lz_an = 0.f;

    // ---- filter operand: straight-line, clamped addresses, all loads in flight together.  With 16 taps the slice
    //      is one contiguous run of CX*16 floats: 16-byte loads, one per thread; otherwise up to EPT scalars.
    constexpr bool k16 = K16;
    const int n4 = nsl >> 1;                                        // 8-byte pieces (keeps the register count low
    const int e4 = min((int)threadIdx.x, n4 - 1);                   //  enough for two workgroups per CU)
    f32x2 part4[FP], wv4 = {0, 0};
    float part[EPT][FP], wv[EPT];
    long gev[EPT];
    bool okv[EPT];
    if (k16) {
        const long g4 = ((long)cx0 * 16 >> 1) + e4;
        if (FUSE > 0) {
            // partial k in the scalar offset of the buffer load: no 64-bit address arithmetic per load on the VALU
            const unsigned ckk_b = (unsigned)hC * 16u * 4u;
            const __amdgpu_buffer_rsrc_t rg = pt_rsrc(h_filt, (unsigned)h_KSPL * ckk_b);
#pragma unroll
            for (int k = 0; k < FP; ++k)
                part4[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rg, (unsigned)g4 * 8u, (unsigned)min(k, h_KSPL - 1) * ckk_b, 0));
            wv4 = ((const f32x2*)h_w)[g4];
        } else if (FUSE < 0) {
            part4[0] = ((const f32x2*)h_filt)[g4];                  // w_{T-1}
            wv4 = ((const f32x2*)h_w)[g4];                          // g_T
            const float* __restrict__ anp = h_w + h_aux;
            lz_an = lane < (ksc16 ? 16 : (ksc1 ? 1 : 8)) ? anp[lane] : 0.f;
            lz_q = lane < h_n ? anp[64 + lane] : 0.f;
        } else {
            part4[0] = ((const f32x2*)h_filt)[g4];
        }
    } else {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = threadIdx.x + q * nthreads;
            const int ec = min(e, nsl - 1);
            const int cl = ec >> 4, tp = ec & 15;
            okv[q] = e < nsl && tp < KK;
            gev[q] = (long)(cx0 + cl) * KK + (tp < KK ? tp : 0);
            if (FUSE > 0) {
                const long CKK = (long)hC * KK;
#pragma unroll
                for (int k = 0; k < FP; ++k) part[q][k] = h_filt[(long)min(k, h_KSPL - 1) * CKK + gev[q]];
                wv[q] = h_w[gev[q]];
            } else {
                part[q][0] = h_filt[gev[q]];
            }
        }
    }

    // ---- feature slice of this wave: NK float4 (+ NK scalars for the trailing quads).  A wave stalls at a load it
    //      cannot issue (the CU accepts ~45 B/clk), so only the first CD k-steps are requested before the filter is
    //      staged; the rest are issued CD k-steps ahead of the MFMAs that consume them.
    constexpr int CD = NK < PT_C2_CD ? NK : PT_C2_CD;    // 6 or 8 ahead: register spills under the 80-VGPR cap, 10.3 vs 10.15 us
    const int cbase = cx0 + 4 * (h * NK) + kq;
    const int pos = 64 * t + 4 * j;
    const bool pv = pos < HW;
    const __amdgpu_buffer_rsrc_t fr = pt_rsrc(fi, (unsigned)hC * HW * 4u);
    const unsigned fo = ((unsigned)cbase * HW + (pv ? pos : 0)) * 4u;
    const int lpos = 64 * h_TF + j;
    const bool lv = LEFT && t == 0 && j < 4 * h_rem;
    const unsigned lo = lv ? ((unsigned)cbase * HW + lpos) * 4u : 0xFFFFFFF0u - 64u * (unsigned)HW * 4u;   // no position: reads 0
    f32x4 bq[NK];
    float bl[NK];
    auto ldq = [&](int k) {
        bq[k] = pt_bload4(fr, fo + (unsigned)(4 * k) * HW * 4u);
        if (LEFT && t == 0) bl[k] = pt_bload1(fr, lo + (unsigned)(4 * k) * HW * 4u);
    };
    if (PT_C2_BAR) __syncthreads();
    // the leftover-quad scalars first: the filter reduction below then waits for vmcnt(CD) on every wave -- the 16-byte loads stay
    // in flight behind it -- instead of a count that fits the waves without the leftover tile
    if (LEFT && t == 0) {
#pragma unroll
        for (int k = 0; k < CD; ++k) bl[k] = pt_bload1(fr, lo + (unsigned)(4 * k) * HW * 4u);
    }
#pragma unroll
    for (int k = 0; k < CD; ++k) bq[k] = pt_bload4(fr, fo + (unsigned)(4 * k) * HW * 4u);
    __builtin_amdgcn_sched_barrier(0);                              // nothing that needs the argument block above this line
    const Corr2Args a = pt_late_args<Corr2Args>(56);                // 5 pointers / longs + 3 dwords = 52 bytes, 8-aligned
    PT_STAMP(a, 1);

    // ---- reduce + publish the filter slice
    float gsq = 0.f;
    if (k16) {
        f32x2 v4;
        if (FUSE > 0) {
            v4 = (f32x2){0, 0};
#pragma unroll
            for (int k = 0; k < FP; ++k)
                if (k < h_KSPL) v4 += part4[k];                                     // fixed order
            v4 += a.reg * wv4;
            if ((int)threadIdx.x < n4) {
                gsq = v4[0] * v4[0] + v4[1] * v4[1];
                if (publish) ((f32x2*)a.g_out)[((long)cx0 * 16 >> 1) + e4] = v4;
            }
        } else if (FUSE < 0) {
            float qt = 0.f;
            for (int k = lane + 64; k < h_n; k += 64) qt += (h_w + h_aux)[64 + k];   // memories of more than 64 samples
            const float astep = sd_final_astep(lz_q, qt, lz_an, a.step, a.reg);
            v4 = (f32x2){sd_final_apply(part4[0][0], wv4[0], astep), sd_final_apply(part4[0][1], wv4[1], astep)};
            if ((int)threadIdx.x < n4 && publish) ((f32x2*)a.g_out)[((long)cx0 * 16 >> 1) + e4] = v4;
        } else {
            v4 = part4[0];
        }
        if ((int)threadIdx.x < n4) ((f32x2*)afilt)[e4] = v4;
        for (int q4 = threadIdx.x + nthreads; q4 < n4; q4 += nthreads)          // slices larger than the block
            for (int m = 0; m < 2; ++m) afilt[2 * q4 + m] = corr2_filter_elem(a, KK, cx0, 2 * q4 + m, publish, gsq);
    } else {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = threadIdx.x + q * nthreads;
            float v;
            if (FUSE > 0) {
                v = 0.f;
#pragma unroll
                for (int k = 0; k < FP; ++k) v += k < h_KSPL ? part[q][k] : 0.f;   // fixed order
                v += a.reg * wv[q];
                if (okv[q]) {
                    gsq += v * v;
                    if (publish) a.g_out[gev[q]] = v;
                }
            } else {
                v = part[q][0];
            }
            if (e < nsl) afilt[e] = okv[q] ? v : 0.f;
        }
        for (int e = threadIdx.x + EPT * nthreads; e < nsl; e += nthreads)
            afilt[e] = corr2_filter_elem(a, KK, cx0, e, publish, gsq);
    }
    if (FUSE > 0 && publish) {                                      // uniform per workgroup
        const float tot = block_sum(gsq, scratch, nthreads);
        if (threadIdx.x == 0) a.anum_part[x] = tot;
    }
    PT_STAMP(a, 2);
    __syncthreads();
    PT_STAMP(a, 3);

    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0}, accL = {0, 0, 0, 0};
#if PT_C2_AV_UPFRONT
    // the wave's NK filter-operand values in one batch of LDS reads in front of the loop (round 6: inside the loop every k-step was
    // read -> wait -> 4 MFMAs, the uniform leftover-tile branches keep the compiler from hoisting the next read over them)
    float avv[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) avv[k] = afilt[(4 * (h * NK + k) + kq) * 16 + j];
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (k + CD < NK) ldq(k + CD);
#if PT_C2_AV_UPFRONT
        const float av = avv[k];
#else
        const float av = afilt[(4 * (h * NK + k) + kq) * 16 + j];
#endif
        acc0 = mfma16(av, bq[k][0], acc0);
        acc1 = mfma16(av, bq[k][1], acc1);
        acc2 = mfma16(av, bq[k][2], acc2);
        acc3 = mfma16(av, bq[k][3], acc3);
        if (LEFT && t == 0) accL = mfma16(av, bl[k], accL);
    }
    PT_STAMP(a, 4);
    // ---- memory insert rides on the pass (pytracking/tracker/dimp/dimp.py:429-441)
    if (over && a.copy_dst) {
        const pt_gf dp = a.copy_dst + pos + (long)cbase * HW;
        if (pv) {
#pragma unroll
            for (int k = 0; k < NK; ++k) *(f32x4*)(dp + (long)(4 * k) * HW) = bq[k];
        }
        if (lv) {
            const pt_gf dl = a.copy_dst + lpos + (long)cbase * HW;
#pragma unroll
            for (int k = 0; k < NK; ++k) dl[(long)(4 * k) * HW] = bl[k];
        }
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r;
        if (row < KK) {
            f32x4 v = {acc0[r], acc1[r], acc2[r], acc3[r]};
            *(f32x4*)(Tl + row * hHWp + pos) = v;                 // columns >= HW land in the padding
            if (LEFT && t == 0) Tl[row * hHWp + lpos] = accL[r];  // j >= 4*rem: padding as well (lpos < HWp)
        }
    }
    PT_STAMP(a, 5);
    __syncthreads();
    PT_STAMP(a, 6);

    // ---- shift-and-add of the tap planes (both halves), fixed order
    const int ph = a.KH / 2, pw = a.KW / 2, OO = a.OH * a.OW;
    const float inv_ow = 1.0f / (float)a.OW;
    const int nsm = pair ? 2 : 1;                                   // samples whose tap planes this workgroup holds
    if (a.KH == 4 && a.KW == 4) {                                   // the trackers' filter size: fully unrolled
      for (int s2 = 0; s2 < nsm; ++s2) {                            // uniform: one round of the block per sample of the pair
        const float* __restrict__ T0 = lds + nsl + (long)s2 * NH * KK * hHWp;
        const float* __restrict__ T1 = T0 + (long)KK * hHWp;      // second k-step half (NH == 2)
#if defined(PT_C2_OUT_GLOBAL) && PT_C2_OUT_GLOBAL
        const auto out = pt_global((float*)a.spart) + ((long)x * a.n + (pair ? 2 * i0 + s2 : i0)) * OO;   // global_store instead of flat_store (A/B knob)
#else
        const pt_gf out = a.spart + ((long)x * a.n + (pair ? 2 * i0 + s2 : i0)) * OO;
#endif
        // Round 6 (profiles/r06l_pass_ablation.txt: this loop is 0.9 us of the pass, bound by VALU issue -- ~100 vector instructions per output
        // around 16 LDS reads): the 16 taps of an output sit at  base + u (4 HWp + W) + v (HWp + 1)  with a per-output base and UNIFORM
        // steps, and whether a tap lies inside the map is (row u ok) & (column v ok): 8 compares and one add per tap instead of two compares,
        // a multiply-add and two selects.  An invalid tap reads a finite value somewhere inside the LDS allocation (the filter operand and
        // earlier planes lie in front of it; behind the last plane its own padding, or -- 22x22: HWp = 516 < HW + 2 W + 2 -- slack the plan
        // adds to the allocation) and is replaced by 0 through a BIT mask, so uninitialised slack cannot leak a NaN.  Same values added in
        // the same order: same bits.
        const bool direct = PT_C2_SA_DIRECT && a.sa_direct;                                         // uniform
        if (direct) {
            const int su = 4 * hHWp + a.W, sv = hHWp + 1;
            for (int o = threadIdx.x; o < OO; o += nthreads) {
                const int y = fdiv(o, inv_ow), xx0 = o - y * a.OW;
                const int base = (y - 2) * a.W + (xx0 - 2);
                // all 16 (32) LDS reads are requested before the first use: written as a select around the load, the compiler made every
                // read conditional (branch, read, wait -- 16 serialised LDS round trips; and the per-tap form below waits after every one
                // or two reads as well).  The masks are applied as bit masks so that the reads stay unconditional.
                float t0[16], t1[16];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int idx = base + (u * su + v * sv);
                        t0[u * 4 + v] = T0[idx];
                        if (NH == 2) t1[u * 4 + v] = T1[idx];
                    }
                int rok[4], cok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rok[u] = (unsigned)(y + u - 2) < (unsigned)a.H ? -1 : 0;
#pragma unroll
                for (int v = 0; v < 4; ++v) cok[v] = (unsigned)(xx0 + v - 2) < (unsigned)a.W ? -1 : 0;
                __builtin_amdgcn_sched_barrier(0);
                float tv[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float tsum = NH == 2 ? t0[q] + t1[q] : t0[q];
                    tv[q] = __builtin_bit_cast(float, __builtin_bit_cast(int, tsum) & (rok[q >> 2] & cok[q & 3]));
                }
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) s += tv[q];
                out[o] = s;
            }
            continue;
        }
        for (int o = threadIdx.x; o < OO; o += nthreads) {
            const int y = fdiv(o, inv_ow), xx0 = o - y * a.OW;
            float tv[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int yy = y + u - 2;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int xx = xx0 + v - 2;
                    const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
                    const int idx = (u * 4 + v) * hHWp + (ok ? yy * a.W + xx : 0);
                    const float tsum = NH == 2 ? T0[idx] + T1[idx] : T0[idx];
                    tv[u * 4 + v] = ok ? tsum : 0.f;
                }
            }
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += tv[q];
#if defined(PT_C2_EXP) && PT_C2_EXP == 1
            s = T0[o];                                               // ABLATION (wrong results, timing only): no shift-and-add
#endif
            out[o] = s;
        }
      }
        PT_STAMP(a, 7);
        return;
    }
    // other filter sizes
    for (int s2 = 0; s2 < nsm; ++s2) {
      const float* __restrict__ T0 = lds + nsl + (long)s2 * NH * KK * hHWp;
      const float* __restrict__ T1 = T0 + (long)KK * hHWp;
      const pt_gf out = a.spart + ((long)x * a.n + (pair ? 2 * i0 + s2 : i0)) * OO;
      for (int o = threadIdx.x; o < OO; o += nthreads) {
        const int y = fdiv(o, inv_ow), xx0 = o - y * a.OW;
        float s = 0.f;
        for (int u = 0; u < a.KH; ++u) {
            const int yy = y + u - ph;
            if ((unsigned)yy >= (unsigned)a.H) continue;
            for (int v = 0; v < a.KW; ++v) {
                const int xx = xx0 + v - pw;
                if ((unsigned)xx < (unsigned)a.W) {
                    const int idx = (u * a.KW + v) * hHWp + yy * a.W + xx;
                    s += NH == 2 ? T0[idx] + T1[idx] : T0[idx];
                }
            }
        }
        out[o] = s;
      }
    }
}

int pt_launch_corr2(const PtFast& p, const float* feat, long stride_n, const float* filt, float* spart, hipStream_t st,
                    const PtCorrFuse* fuse, int slot, const float* src, float* copy_dst, const PtCorrLazy* lazy) {
    Corr2Args a;
    a.feat = (pt_gcf)feat; a.stride_n = stride_n; a.filt = (pt_gcf)filt; a.spart = (pt_gf)spart;
    a.n = p.n; a.C = p.C; a.H = p.H; a.W = p.W; a.KH = p.KH; a.KW = p.KW; a.OH = p.OH; a.OW = p.OW;
    a.CX = p.CX; a.TF = p.TF; a.rem = p.rem; a.tiles = p.tiles; a.HWp = p.HWp; a.nh = p.nh; a.KSC = p.KSC;
    a.sa_direct = p.sa_direct;
    a.gpart = nullptr; a.KSPL = 0; a.w = nullptr; a.reg = 0.f; a.g_out = nullptr; a.anum_part = nullptr;
    if (fuse) {
        a.gpart = (pt_gcf)fuse->gpart; a.KSPL = fuse->KSPL; a.w = (pt_gcf)fuse->w; a.reg = fuse->reg; a.g_out = (pt_gf)fuse->g_out;
        a.anum_part = (pt_gf)fuse->anum_part;
    }
    a.slot = slot; a.src = (pt_gcf)src; a.copy_dst = (pt_gf)copy_dst;
    a.step = 0.f;
    unsigned h_aux = 0, h_nbits = 0;
    if (lazy) {
        // the deferred last update of the previous solve forms the filter operand (k_corr2, FUSE < 0); `filt` is not read
        if (fuse || p.KK != 16 || p.n > 511 || p.CX * 8 > p.corr_threads || !lazy->w_prev || !lazy->g || !lazy->anum || !lazy->w_out ||
            lazy->qs != lazy->anum + 64 || lazy->anum < lazy->g || (lazy->anum - lazy->g) > 0x3fffffffL)
            return PT_ERR_UNSUPPORTED;
        if (((uintptr_t)lazy->w_prev % 16) || ((uintptr_t)lazy->g % 16) || ((uintptr_t)lazy->w_out % 16)) return PT_ERR_UNSUPPORTED;
        a.filt = (pt_gcf)lazy->w_prev; a.w = (pt_gcf)lazy->g; a.g_out = (pt_gf)lazy->w_out; a.reg = lazy->reg_eps; a.step = lazy->step;
        h_aux = (unsigned)(lazy->anum - lazy->g);
        h_nbits = (unsigned)p.n << 23;
    }
    PT_STAMP_SET(a);
    if (((uintptr_t)feat % 16) || (stride_n % 4) || ((uintptr_t)src % 16) || ((uintptr_t)copy_dst % 16)) return PT_ERR_UNSUPPORTED;
    if ((long)p.n * stride_n * 4 >= (1L << 31)) return PT_ERR_UNSUPPORTED;
    if (p.KK == 16 && (((uintptr_t)filt % 16) || ((uintptr_t)a.gpart % 16) || ((uintptr_t)a.w % 16) || ((uintptr_t)a.g_out % 16)))
        return PT_ERR_UNSUPPORTED;
    dim3 grid(p.KSC * (p.n / p.spw)), block(p.corr_threads);
    if (p.C >= (1 << 16) || p.HW >= (1 << 16) || p.tiles > 31 || p.TF > 31 || p.rem > 15 || a.KSPL > 16) return PT_ERR_UNSUPPORTED;
    const unsigned h_dims = ((unsigned)p.C << 16) | (unsigned)p.HW;
    const unsigned h_geo = (unsigned)p.tiles | ((unsigned)p.TF << 5) | ((unsigned)p.rem << 10) | ((unsigned)a.KSPL << 14) |
                           ((p.KSC == 16 ? 1u : 0u) << 20) | ((p.KSC == 1 ? 1u : 0u) << 21) | ((p.spw == 2 ? 1u : 0u) << 22) | h_nbits;
#define PT_C2_HOT(FPTR) (const float*)a.feat, a.stride_n, (const float*)(FPTR), (const float*)a.w, (const float*)a.src, a.slot, h_dims, h_geo, h_aux
#define PT_C2G(NKV, LF, KF, NHV)                                                                                    \
    do {                                                                                                         \
        if (lazy) { if constexpr (KF) hipLaunchKernelGGL((k_corr2<NKV, LF, -1, true, NHV>), grid, block, p.corr_lds, st, PT_C2_HOT(a.filt), a); } \
        else if (!a.gpart) hipLaunchKernelGGL((k_corr2<NKV, LF, 0, KF, NHV>), grid, block, p.corr_lds, st, PT_C2_HOT(a.filt), a);             \
        else if (a.KSPL <= 8) hipLaunchKernelGGL((k_corr2<NKV, LF, 8, KF, NHV>), grid, block, p.corr_lds, st, PT_C2_HOT(a.gpart), a);     \
        else hipLaunchKernelGGL((k_corr2<NKV, LF, 16, KF, NHV>), grid, block, p.corr_lds, st, PT_C2_HOT(a.gpart), a);                     \
    } while (0)
#define PT_C2H(NKV, LF, KF)              \
    do {                                 \
        if (p.nh == 2) PT_C2G(NKV, LF, KF, 2); \
        else PT_C2G(NKV, LF, KF, 1);     \
    } while (0)
#define PT_C2F(NKV, LF)                  \
    do {                                 \
        if (p.KK == 16) PT_C2H(NKV, LF, true); \
        else PT_C2H(NKV, LF, false);     \
    } while (0)
#define PT_C2(NKV)                  \
    do {                            \
        if (p.left) PT_C2F(NKV, true); \
        else PT_C2F(NKV, false);    \
    } while (0)
    if (p.NK == 2) PT_C2(2);
    else if (p.NK == 4) PT_C2(4);
    else if (p.NK == 8) PT_C2(8);
    else PT_C2(16);
#undef PT_C2_HOT
#undef PT_C2G
#undef PT_C2H
#undef PT_C2F
#undef PT_C2
    PT_CHECK_LAUNCH();
    return PT_OK;
}

// ---------------------------------------------------------------------------------------------------
// k_adj2
// ---------------------------------------------------------------------------------------------------
struct Adj2Args {
    const float* feat; long stride_n; float* gpart;
    int n, C, H, W, KH, KW, OH, OW, NG, gper, U, bpx, PH, PW, ns_max, zn;
    const float* inp;        // V_PLAIN: residual maps (n, OH, OW)
    SdArgs sd;               // solver variants: solver state
    int t, want_loss;        //                  iterate index of the maps this launch builds
    PT_STAMP_ARG
};

// What k_adj2 needs BEHIND its prologue, compact (46 dwords = three scalar 16-dword loads): fetched late, see pt_late_issue.
// Same member names as Adj2Args / SdArgs so that the update-stage helpers below read either.
struct Adj2SdLate {
    int n, CKK, has_sw, has_softmax_reg;
    float step, reg, alpha_eps, act_param, softmax_reg;
    int pad0;
    pt_gcf sw, s_in, w0;
    pt_gf mask, sws, s, sg, lms, g, lossp, w_iters;
};
struct Adj2Late {
    pt_gf gpart;
    int C, KH, KW, OH, OW, PH, PW, ns_max, zn, t, want_loss, pad1;
    Adj2SdLate sd;
    PT_STAMP_ARG
};
#ifndef PT_STAMPS
static_assert(sizeof(Adj2Late) == 46 * 4, "Adj2Late layout");
#endif

// residual-map providers of k_adj2
enum { V_PLAIN = 0, V_DIMP_RELU = 1, V_DIMP_BENT = 2, V_L2 = 3, V_PRDIMP = 4 };

// per-lane inputs of the update stage for one sample: E strided elements per lane.
//   pk : packed by k_fast_init / k_fast_sgq, ONE 16-byte load per element:
//          DiMP relu / L2 : {sws^2 * s, sws^2 * (F g), sws^2 * label, mask}
//          DiMP bentpar   : {s, F g, label, mask}  (+ sw)          PrDiMP : {s, F g, label, -}
//   s, sg : raw scores / F g, only loaded by the wave that owns the sample (it stores s_t; the loss needs more raw maps,
//           which that wave loads late)
template <int E>
struct PReg { f32x4 pk[E]; float s[E], sg[E], sw[E]; };

// hot part: the packed operands through pointers that arrived as preloaded kernel parameters (no argument block needed)
template <int V, int E>
__device__ __forceinline__ void sdp_load_hot(const float* __restrict__ pkp, int OO, int i, int lane, PReg<E>& r) {
    const long base = (long)i * OO;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const long q = base + min(lane + 64 * e, OO - 1);
        if (V == V_PLAIN) r.s[e] = pkp[q];
        else r.pk[e] = ((const f32x4*)pkp)[q];
    }
}
// the rest: bentpar weights, and the raw maps of the wave that owns the sample
template <int V, int E, typename A>
__device__ __forceinline__ void sdp_load_rest(const A& a, int i, int lane, bool home, PReg<E>& r) {
    const int OO = a.OH * a.OW;
    const long base = (long)i * OO;
    // t == 0: there is no F g yet; astep is 0 and any finite operand does
    // (global-qualified accesses: a flat operation in front of the LDS-only barriers would have to be waited for, see pt_lds_barrier)
    const auto sgp = a.t > 0 ? pt_global((const float*)a.sd.sg) : pt_global((const float*)a.sd.s_in);
    if (V == V_DIMP_BENT) {
#pragma unroll
        for (int e = 0; e < E; ++e) r.sw[e] = pt_global((const float*)a.sd.sws)[base + min(lane + 64 * e, OO - 1)];
    }
    if (V != V_PLAIN && home) {                                     // uniform per wave
        const auto sin = pt_global((const float*)a.sd.s_in);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const long q = base + min(lane + 64 * e, OO - 1);
            r.s[e] = sin[q];
            r.sg[e] = sgp[q];
        }
    }
}
template <int V, int E, typename A>
__device__ __forceinline__ void sdp_load(const A& a, const float* __restrict__ pkp, int i, int lane, bool home, PReg<E>& r) {
    sdp_load_hot<V, E>(pkp, a.OH * a.OW, i, lane, r);
    sdp_load_rest<V, E>(a, i, lane, home, r);
}

// ---- Fused init stage (round 6, experiment O): the FIRST adjoint pass of a solve builds the label / mask / weight maps itself ----------
// The init stage (k_fast_init2, sd_solver.hip) sums the channel-range slices of s_0 = F w, re-centres the box of the classification slot on
// the arg-max and evaluates the maps -- all per sample, nothing global -- and its only hurry is the first adjoint pass, whose update
// stage has one wave per sample anyway.  k_adj2<.., INIT = true> takes the slices / the boxes / the look-up tables through the three
// preloaded pointers the t > 0 launches use for (pk, qs, anum), forms the packed operands in registers and lets the sample's home wave
// store what the later stages read (s_0, label, mask, sws, lms, classification outputs): one dependent launch less per frame.
// Every workgroup of a position slice redoes the maps of its <= 8 samples (32 channel blocks: L2 traffic, no HBM traffic).
// DiMP kinds only (PrDiMP's label needs a block-wide normalisation); <= 8 slices; tables of <= 127 bins in one contiguous array.
struct Adj2InitLate {                                              // 16 dwords: ONE scalar load, fetched behind Adj2Late
    pt_gf label, cls_scores, cls_peak, cls_bb;
    const int* slot_dyn;
    int slot, mask_act;
    float feat_stride, bin_disp, gauss_sigma, hinge_thr;
};
static_assert(sizeof(Adj2InitLate) == 64, "Adj2InitLate: one 16-dword block");
#define PT_INIT_MAXKS 8
template <int E>
struct IReg { float sl[E][PT_INIT_MAXKS]; float bb[4]; };

// hot part: the slices of s_0 and the box of sample i, through preloaded pointers.  Slice k in the scalar offset of the buffer load.
template <int E>
__device__ __forceinline__ void sdi_load_hot(const float* __restrict__ spart, const float* __restrict__ bb, int n, int OO, int KS, int i, int lane,
                                             IReg<E>& r) {
    const unsigned stb = (unsigned)n * (unsigned)OO * 4u;
    const __amdgpu_buffer_rsrc_t rs = pt_rsrc(spart, (unsigned)KS * stb);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned vo = ((unsigned)i * (unsigned)OO + (unsigned)min(lane + 64 * e, OO - 1)) * 4u;
#pragma unroll
        for (int k = 0; k < PT_INIT_MAXKS; ++k)
            r.sl[e][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (unsigned)min(k, KS - 1) * stb, 0));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) r.bb[k] = bb[4 * i + k];
}

// s_0, classification epilogue (pytracking/libs/dcf.py:156-164: first maximum), maps (optimizer.py:111-125 / :201-252) -> packed operands
// of the update stage, exactly as k_fast_init2 forms them; `lutl`: the look-up tables in LDS.
template <int V, int E, typename A>
__device__ __forceinline__ void sdi_make(const A& a, const Adj2InitLate& il, const float* __restrict__ lutl, int num_bins, int KS, int i, int lane,
                                         bool home, const IReg<E>& ir, PReg<E>& r) {
    const int OO = a.OH * a.OW;
    const long base = (long)i * OO;
    const float inv_ow = 1.0f / (float)a.OW;
    float s0[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < PT_INIT_MAXKS; ++k) sm += k < KS ? ir.sl[e][k] : 0.f;   // fixed order
        s0[e] = sm;
    }
    float b0 = ir.bb[0], b1 = ir.bb[1];
    const float b2 = ir.bb[2], b3 = ir.bb[3];
    const float off = (float)(a.KH % 2) * 0.5f;
    const int slot = il.slot_dyn ? *pt_global(il.slot_dyn) : il.slot;
    if (i == slot) {                                                // uniform per wave
        float best = -INFINITY;
        int besti = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int o = lane + 64 * e;
            if (o < OO && (s0[e] > best || (s0[e] == best && o < besti))) { best = s0[e]; besti = o; }
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) {
            const float ov = __shfl_xor(best, sh, 64);
            const int oi = __shfl_xor(besti, sh, 64);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        const int row = besti / a.OW, col = besti - row * a.OW;
        b0 = ((float)col + off) * il.feat_stride - b2 * 0.5f;
        b1 = ((float)row + off) * il.feat_stride - b3 * 0.5f;
        if (home) {
            if (lane == 0) {
                pt_global((float*)il.cls_peak)[0] = (float)row;
                pt_global((float*)il.cls_peak)[1] = (float)col;
                pt_global((float*)il.cls_bb)[4 * slot] = b0;
                pt_global((float*)il.cls_bb)[4 * slot + 1] = b1;
            }
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (lane + 64 * e < OO) pt_global((float*)il.cls_scores)[lane + 64 * e] = s0[e];
        }
    }
    const float ctr_r = (b1 + b3 * 0.5f) / il.feat_stride - off;   // optimizer.py:112-113 (flip -> row first)
    const float ctr_c = (b0 + b2 * 0.5f) / il.feat_stride - off;
    const float swv = a.sd.has_sw ? pt_global((const float*)a.sd.sw)[i] : 1.0f / (float)a.sd.n;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int o = lane + 64 * e, oc = min(o, OO - 1);
        const int y = fdiv(oc, inv_ow), x = oc - y * a.OW;
        const float d0 = (float)y - ctr_r, d1 = (float)x - ctr_c;
        float lb, m, sw;
        if (V == V_L2) sd_init_elem_l2(il.gauss_sigma, il.hinge_thr, d0, d1, swv, lb, m, sw);
        else sd_init_elem_dimp(lutl, num_bins, il.mask_act, il.bin_disp, d0, d1, swv, lb, m, sw);
        if (V == V_DIMP_BENT) {
            r.pk[e] = (f32x4){s0[e], 0.f, lb, m};
            r.sw[e] = sw;
        } else {
            const float w2 = sw * sw;
            r.pk[e] = (f32x4){w2 * s0[e], w2 * 0.f, w2 * lb, m};
        }
        r.s[e] = s0[e];
        r.sg[e] = 0.f;
        if (home && o < OO) {                                       // what the later stages read (k_fast_sgq2: s, lms; bentpar passes: sws)
            pt_global((float*)a.sd.s)[base + o] = s0[e];
            pt_global((float*)il.label)[base + o] = lb;
            pt_global((float*)a.sd.mask)[base + o] = m;
            pt_global((float*)a.sd.sws)[base + o] = sw;
            pt_global((f32x4*)a.sd.lms)[base + o] = (f32x4){lb, m, sw, 0.f};
        }
    }
}

// Update stage of the steepest-descent iteration for sample i, executed by one wave
// (optimizer.py:137-146,160 / :403-408,430): s_t = s_{t-1} - step*alpha*(F g); residual map -> zero-padded LDS map;
// the owning workgroup (`home`) also stores s_t (and the PrDiMP softmax) and the sample's loss term.
template <int V, int E, typename A>
__device__ __forceinline__ void sdp_compute(const A& a, int i, int lane, const PReg<E>& r, float astep,
                                            float* __restrict__ map, int oy, int ox, bool home) {
    const auto& sd = a.sd;
    const int OO = a.OH * a.OW;
    const long base = (long)i * OO;
    const float inv_ow = 1.0f / (float)a.OW;
    float val[E], aux[E];
    float lacc = 0.f;
    if (V == V_PLAIN) {
#pragma unroll
        for (int e = 0; e < E; ++e) val[e] = r.s[e];
    } else if (V == V_DIMP_RELU || V == V_L2) {
        // With u = sws^2 * s_t the residual map entry  der * sws * (sws * (act - label))  of optimizer.py:140,146 is
        //   s_t > 0 : u - a3        s_t < 0 : m * (m*u - a3)        s_t == 0 : der0 * (0 - a3),   a3 = sws^2 * label,
        // der0 = (1+m)/2 for LeakyReluPar (sign(0) = 0, activation.py:43-44), m for the L2 hinge (optimizer.py:262-263).
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float u = r.pk[e][0] - astep * r.pk[e][1], a3 = r.pk[e][2], m = r.pk[e][3];
            const float d0 = V == V_DIMP_RELU ? (1.0f + m) * 0.5f : m;
            val[e] = u > 0.f ? u - a3 : (u < 0.f ? m * (m * u - a3) : d0 * (0.f - a3));
        }
    } else if (V == V_DIMP_BENT) {                                  // activation.py:49-66
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float x = r.pk[e][0] - astep * r.pk[e][1], L = r.pk[e][2], am = r.pk[e][3];
            const float bp = sd.act_param, rt = sqrtf(x * x + 4.0f * bp * bp);
            const float act = (1.0f - am) * 0.5f * (rt - 2.0f * bp) + (1.0f + am) * 0.5f * x;
            const float der = (1.0f - am) * 0.5f * (x / rt) + (1.0f + am) * 0.5f;
            val[e] = der * (r.sw[e] * (r.sw[e] * (act - L)));
        }
    } else {
        const float swp = sd.has_sw ? pt_global((const float*)sd.sw)[i] : 1.0f / (float)sd.n;   // :387-390
        float sv[E];
        float mx = sd.has_softmax_reg ? sd.softmax_reg : -INFINITY;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            sv[e] = r.pk[e][0] - astep * r.pk[e][1];
            if (lane + 64 * e < OO) mx = fmaxf(mx, sv[e]);
        }
        mx = wave_max(mx);
        float es = 0.f, ls = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool ok = lane + 64 * e < OO;
            aux[e] = ok ? expf(sv[e] - mx) : 0.f;
            es += aux[e];
            ls += ok ? r.pk[e][2] * sv[e] : 0.f;
        }
        es = wave_sum(es);
        ls = wave_sum(ls);
        if (sd.has_softmax_reg) es += expf(sd.softmax_reg - mx);                    // activation.py:7-16
        const float inv = 1.0f / es;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            aux[e] *= inv;                                                          // softmax P
            val[e] = swp * (aux[e] - r.pk[e][2]);                                   // :408
        }
        lacc = swp * (logf(es) + mx - ls);                                          // :393-396 (same in every lane)
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int o = lane + 64 * e;
        if (o < OO) {
            const int y = fdiv(o, inv_ow), x = o - y * a.OW;
            map[(y + oy) * a.PW + x + ox] = val[e];
        }
    }
    if (V != V_PLAIN && home) {                                     // uniform per wave
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int o = lane + 64 * e;
            if (o < OO) {                                           // global_store (not flat): the LDS-only barrier behind the update stage
                if (a.t > 0) pt_global((float*)sd.s)[base + o] = r.s[e] - astep * r.sg[e];   // must not have to wait for these (pt_lds_barrier)
                if (V == V_PRDIMP) pt_global((float*)sd.mask)[base + o] = aux[e];
            }
        }
        if (a.want_loss) {
            if (V != V_PRDIMP) {                                    // sum_o (sws*(act - label))^2 from the raw maps (:140-143)
                const int sact = V == V_L2 ? 2 : (V == V_DIMP_BENT ? PT_ACT_BENTPAR : PT_ACT_RELU);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const f32x4 lm = pt_global((const f32x4*)sd.lms)[base + min(lane + 64 * e, OO - 1)];   // {label, mask, sws, -}
                    float act, der;
                    act_pair(sact, sd.act_param, r.s[e] - astep * r.sg[e], lm[1], act, der);
                    const float rr = lm[2] * (act - lm[0]);
                    lacc += (lane + 64 * e < OO) ? rr * rr : 0.f;
                }
                lacc = wave_sum(lacc);
            }
            if (lane == 0) pt_global((float*)sd.lossp)[(long)a.t * sd.n + i] = lacc;
        }
    }
}

// UM: 16-position groups per wave (compile-time trip count of the pipelined loop; 12 for the 22x22 PrDiMP geometry)
//   INIT (round 6): the launch of iterate 0 also does the init stage -- h_pk = slices of s_0, h_qs = boxes, h_anum = the three look-up
//   tables as one array (null: L2 kind), the `t` bits of h_g1 carry num_bins (t is 0); see sdi_make above.  i_arg is only read by
//   the INIT instantiations (same signature for all: the late-fetch offsets are checked against the code object's metadata)
template <int V, int E, int UM, bool INIT = false>
__global__ __launch_bounds__(PT_ADJ_WAVES * 64) void k_adj2(const float* h_feat, long h_stride, const float* h_pk, const float* h_qs, const float* h_anum, unsigned h_g1,
                                                                unsigned h_g2, unsigned h_g3, unsigned h_g4, Adj2Late a_arg,
                                                                Adj2InitLate i_arg) {
    // h_*: everything up to the first barrier, as scalar kernel parameters preloaded into SGPRs at wave launch (14 dwords; the
    // argument block arrives by scalar loads ~1.2 us later -- see k_corr2 -- and is fetched behind the prologue, pt_late_issue):
    //   h_pk = packed update-stage operands (V_PLAIN: the input maps);
    //   h_g1 = H | W << 8 | U << 16 | bpx << 21 | t << 25;  h_g2 = n | gper << 16;  h_g3 = OO | (C / 16) << 16 | KS << 24
    //   h_g4 = PH | PW << 6 | zn << 12 | ns_max << 20 | KH << 26 | KW << 29
    extern __shared__ __attribute__((aligned(16))) float maps[];    // [ns_max][PH][PW] zero-padded residual maps, zn zeros, quad table
    __shared__ float red[PT_ADJ_WAVES][256];
    PT_STAMP_A(a_arg, 0);
    PT_STAMP_B(a_arg, 0);
    const int hH = (int)(h_g1 & 255u), hW = (int)((h_g1 >> 8) & 255u), hU = (int)((h_g1 >> 16) & 31u), hbpx = (int)((h_g1 >> 21) & 15u);
    const int ht = INIT ? 0 : (int)(h_g1 >> 25);
    const int h_bins = INIT ? (int)(h_g1 >> 25) : 0;
    const int hn = (int)(h_g2 & 0xffffu), hgper = (int)(h_g2 >> 16);
    const int hOO = (int)(h_g3 & 0xffffu), hC = 16 * (int)((h_g3 >> 16) & 255u), hKS = (int)((h_g3 >> 24) & 127u);
    const int hPH = (int)(h_g4 & 63u), hPW = (int)((h_g4 >> 6) & 63u), hzn = (int)((h_g4 >> 12) & 255u), hns_max = (int)((h_g4 >> 20) & 63u);
    const int hKH = (int)((h_g4 >> 26) & 7u), hKW = (int)((h_g4 >> 29) & 7u);
    const int b = blockIdx.x, x = b & 7, rr = b >> 3;
    const int hCB = (int)((h_g3 >> 16) & 255u);
    // channel block and position slice of this workgroup; XCD x (= b % 8) always meets the same channel blocks
    const int cb = hbpx > 0 ? hbpx * x + rr % hbpx : b % hCB, ks = hbpx > 0 ? rr / hbpx : b / hCB;
    const int HW = hH * hW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, j = lane & 15;
    const int total = hn * HW;
    const int hNG = (total + 15) >> 4;
    const int gbeg = ks * hgper, gend = min(hNG, gbeg + hgper);
    const float inv_hw = 1.0f / (float)HW, inv_w = 1.0f / (float)hW;
    const int i_lo = fdiv(gbeg * 16, inv_hw);
    const int i_hi = min(hn - 1, fdiv(gend * 16 - 1, inv_hw));
    const int ns = i_hi - i_lo + 1;
    constexpr int PD = UM < 8 ? UM : (UM == 8 ? 8 : (PT_ADJ_PD < UM ? PT_ADJ_PD : UM));
    const int c = cb * 16 + j;
    const __amdgpu_buffer_rsrc_t fr = pt_rsrc(h_feat, (unsigned)(((long)(hn - 1) * h_stride + (long)hC * HW) * 4));
    const unsigned chw4 = 4u * (unsigned)c * (unsigned)HW;
    const int Pm = 16 * gbeg < total ? 16 * gbeg : 0;
    // where quad (k-quarter tk, wave tw, group tu) of this workgroup lives: sample, position inside the sample (a masked
    // quad re-reads a line already fetched)
    auto quad_at = [&](int tk, int tw, int tu, int& i, int& p0) {
        const int g = gbeg + tw * hU + tu;
        const int P0 = 16 * g + 4 * tk;
        const bool okk = tu < hU && g < gend && P0 < total;
        const int Pc = okk ? P0 : Pm + 4 * tk;
        i = fdiv(Pc, inv_hw);
        p0 = Pc - i * HW;
        return okk;
    };

    // ---- Order of the prologue (round 3, profiles/r03g_pass_phase_stamps.txt: the first feature load used to leave 2.7 us
    //      after the kernel started -- behind the argument fetch, the table build, the update-stage loads and a
    //      `s_waitcnt vmcnt(0)` for alpha):
    //      (1) the small, L2-resident inputs of the update stage are requested first: the memory counter retires in order, so
    //          waiting for them later leaves the feature loads behind them in flight;
    //      (2) the first PD feature loads go out with offsets computed directly (one division per group);
    //      (3) the argument block is requested;
    //      (4) only then the LDS work (zeroed maps, quad table) and the barrier -- all of it on preloaded parameters;
    //      (5) the argument block is waited for; alpha and the update stage follow.
    PReg<E> pr;
    const bool have = wave < ns;
    const int hg0 = ((i_lo + wave) * HW) >> 4;                      // the slice holding a sample's first group owns it
    const bool home0 = have && cb == 0 && hg0 >= gbeg && hg0 < gend;
    IReg<INIT ? E : 1> ir;
    float lutv = 0.f;
    if constexpr (INIT) {
        // the table element first: the counter retires in order, and the tables are staged into LDS in front of the first barrier
        if (V != V_L2) lutv = h_anum[min((int)threadIdx.x, 3 * h_bins - 1)];
        sdi_load_hot<E>(h_pk, h_qs, hn, hOO, hKS, i_lo + min(wave, ns - 1), lane, ir);
    } else {
        sdp_load_hot<V, E>(h_pk, hOO, i_lo + min(wave, ns - 1), lane, pr);
    }
    float an_in = 0.f;
    SdQLane q_in = {0.f, 0.f};
    if (V != V_PLAIN && ht > 0) {                                   // optimizer.py:155-160 / :425-430 (operands of alpha)
        q_in.head = lane < hn ? h_qs[lane] : 0.f;
        an_in = lane < hKS ? h_anum[lane] : 0.f;
    }
    PT_STAMP_B(a_arg, 1);
    if (PT_ADJ_BAR) __syncthreads();
    f32x4 av[UM];
    if (PT_ADJ_EARLY) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            int qi, qp;
            quad_at(kq, wave, u, qi, qp);
            av[u] = pt_bload4(fr, ((unsigned)qi * (unsigned)h_stride + (unsigned)qp) * 4u + chw4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);                              // nothing that needs the argument block above this line
    PtLate<Adj2Late> late = pt_late_issue<Adj2Late>(56);            // 5 pointers / longs + 4 dwords = 56 bytes
    PtLate<Adj2InitLate> late_i;
#ifndef PT_STAMPS
    static_assert(56 + sizeof(Adj2Late) == 240, "offset of the init block in the kernel-argument segment");
#endif
    if constexpr (INIT) late_i = pt_late_issue<Adj2InitLate>(240);
    const int KK = hKH * hKW, PHPW = hPH * hPW;
    const int ZB = hns_max * PHPW;                                  // hzn zeros: what masked quads gather (any tap)
    // residual coordinate of (position (y,x), tap (u,v)) is (y-u+KH/2, x-v+KW/2); in the padded map the residual
    // pixel (yy,xx) sits at (yy + oy, xx + ox)
    const int oy = hKH - 1 - hKH / 2, ox = hKW - 1 - hKW / 2;
    const int uj = j / hKW, vj = j - uj * hKW;
    // lanes j >= K*K feed accumulator columns that are never stored: any in-range tap offset does
    const unsigned tapoff4 = j < KK ? 4u * (unsigned)((hKH - 1 - uj) * hPW + (hKW - 1 - vj)) : 0u;

    PT_STAMP_B(a_arg, 2);
    for (int e = threadIdx.x; e < ns * PHPW; e += (PT_ADJ_WAVES * 64)) maps[e] = 0.f;
    for (int e = threadIdx.x; e < hzn; e += (PT_ADJ_WAVES * 64)) maps[ZB + e] = 0.f;
    PT_STAMP_B(a_arg, 3);
    // Where a quad of positions lives does not depend on the channel or the tap: the workgroup computes it once per quad
    // (sample, feature offset, the 4 residual-map cells incl. the row wrap inside the quad) into an LDS table instead of
    // every lane redoing two divisions and the wrap selects per group -- the pass is bound by VALU issue, not by memory
    // (rocprofv3 counters, profiles/r02j_solver_instruction_mix.txt: 1020 VALU instructions per wave around 64 MFMAs).
    //   tabA[kq][wave][u]   byte offset of the quad in the feature tensor (channel 0)
    //   tabI[kq][wave][u]   byte offsets of its 4 cells in `maps` for tap (KH-1, KW-1); + tapoff4 per lane.  A masked quad
    //                       re-reads a line already fetched and points at the zero block behind the maps.
    int* __restrict__ tabA = (int*)(maps + ZB + hzn);
    int* __restrict__ tabI = tabA + 4 * PT_ADJ_WAVES * UM;
    for (int e = threadIdx.x; e < 4 * PT_ADJ_WAVES * UM; e += (PT_ADJ_WAVES * 64)) {
        const int tu = e % UM, tw = (e / UM) % PT_ADJ_WAVES, tk = e / (UM * PT_ADJ_WAVES);
        int i, p0;
        const bool okk = quad_at(tk, tw, tu, i, p0);
        tabA[e] = (int)(((unsigned)i * (unsigned)h_stride + (unsigned)p0) * 4u);
        const int y0 = fdiv(p0, inv_w), x0 = p0 - y0 * hW;
        // a quad may wrap to the next feature row: one row further in the padded map is PW - W cells more
        const int base = (i - i_lo) * PHPW + y0 * hPW + x0, wr = hPW - hW;
#pragma unroll
        for (int k = 0; k < 4; ++k) tabI[4 * e + k] = 4 * (okk ? base + k + (x0 + k >= hW ? wr : 0) : ZB);
    }
    float* __restrict__ lutl = (float*)(tabI + 16 * PT_ADJ_WAVES * UM);   // INIT: look-up tables behind the quad table
    if constexpr (INIT) {
        if (V != V_L2 && (int)threadIdx.x < 3 * h_bins) lutl[threadIdx.x] = lutv;
    }
    const Adj2Late a = pt_late_get<Adj2Late>(late);
    Adj2InitLate il;
    if constexpr (INIT) il = pt_late_get<Adj2InitLate>(late_i);
    else sdp_load_rest<V, E>(a, i_lo + min(wave, ns - 1), lane, home0, pr);
    const bool wupd = V != V_PLAIN && a.t > 0 && ks == 0;
    const long wge = (long)cb * 16 * KK + min((int)threadIdx.x, 16 * KK - 1);
    float w_prev = 0.f, g_prev = 0.f;
    if (V != V_PLAIN && a.t > 0) {
        for (int k = lane + 64; k < hn; k += 64) q_in.tail += h_qs[k];          // memories of more than 64 samples
        if (wupd) {                                                 // uniform per workgroup; consumed at the end of the kernel
            w_prev = pt_global((const float*)sd_w(a.sd, a.t - 1))[wge];   // global-qualified: these two stay in flight across both barriers
            g_prev = pt_global((const float*)a.sd.g)[wge];
        }
    }
    PT_STAMP_A(a, 1);
    PT_STAMP_B(a, 4);
    pt_lds_barrier();                                               // maps zeroed, quad table written
    PT_STAMP_A(a, 2);
    PT_STAMP_B(a, 5);

    const int tq = (kq * PT_ADJ_WAVES + wave) * UM;
    int foff[UM];
#pragma unroll
    for (int u = 0; u < UM; u += 4) {
        const i32x4 v = *(const i32x4*)__builtin_assume_aligned(tabA + tq + u, 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) foff[u + k] = v[k];
    }
    float bv[UM][4];
    i32x4 cell[UM];
    auto issue = [&](int u) { av[u] = pt_bload4(fr, (unsigned)foff[u] + chw4); };
    auto cells = [&](int u) { cell[u] = *(const i32x4*)__builtin_assume_aligned(tabI + 4 * (tq + u), 16); };
    if (!PT_ADJ_EARLY) {
#pragma unroll
        for (int u = 0; u < PD; ++u) issue(u);
    }
    float astep = 0.f;
    if (V != V_PLAIN && a.t > 0) {
        const float a_num = wave_sum(an_in);
        const float den = fmaxf(wave_sum(q_in.head + q_in.tail) + (a.sd.reg + a.sd.alpha_eps) * a_num, 1e-8f);
        astep = a.sd.step * (a_num / den);
    }
    if constexpr (INIT) {
        if (have) sdi_make<V, E>(a, il, lutl, h_bins, hKS, i_lo + wave, lane, home0, ir, pr);
    }
    if (have) sdp_compute<V, E>(a, i_lo + wave, lane, pr, astep, maps + wave * PHPW, oy, ox, home0);
    for (int sl = wave + PT_ADJ_WAVES; sl < ns; sl += PT_ADJ_WAVES) {   // more samples than waves (tiny maps)
        const int i = i_lo + sl;
        const int hg = (i * HW) >> 4;
        const bool home = cb == 0 && hg >= gbeg && hg < gend;
        if constexpr (INIT) {
            sdi_load_hot<E>(h_pk, h_qs, hn, hOO, hKS, i, lane, ir);
            sdi_make<V, E>(a, il, lutl, h_bins, hKS, i, lane, home, ir, pr);
        } else {
            sdp_load<V, E>(a, h_pk, i, lane, home, pr);
        }
        sdp_compute<V, E>(a, i, lane, pr, astep, maps + sl * PHPW, oy, ox, home);
    }
    PT_STAMP_A(a, 3);
    PT_STAMP_B(a, 6);
    pt_lds_barrier();                                               // residual maps built (the home waves' result stores stay in flight)
    PT_STAMP_A(a, 4);

    // ---- G[c][tap] += feat[c][P] * r[P shifted by tap] over the U contiguous 16-position groups of this wave.
    //      A wave stalls at a load it cannot issue (the CU's memory pipeline accepts ~20-45 B/clk), so the loads are
    //      software-pipelined PD groups ahead of the MFMAs that consume them instead of being issued all up front.
    //      The loop is bound by instruction issue (3.3 of the pass's 7.6 us: 11 vector / LDS instructions around 4 MFMAs per group,
    //      profiles/r04a_pmc_counters.txt).  Experiment PT_ADJ_G2 (round 4, lost): with an even map width a row wrap can only fall
    //      between a quad's second and third position, so cells (0,1) and (2,3) are neighbours in the padded map and the four scalar
    //      gathers can be two 8-byte LDS reads -- at 4-byte alignment, which the LDS serves far slower than four aligned dwords.
    f32x4 accA = {0, 0, 0, 0}, accB = {0, 0, 0, 0};
    auto mfma_loop = [&](auto pair_tag) {
        constexpr bool G2 = decltype(pair_tag)::value;
        auto gather = [&](int u) {
            if constexpr (G2) {
                const f32x2a4 lo = *(const f32x2a4*)((const char*)maps + ((unsigned)cell[u][0] + tapoff4));
                const f32x2a4 hi = *(const f32x2a4*)((const char*)maps + ((unsigned)cell[u][2] + tapoff4));
                bv[u][0] = lo[0]; bv[u][1] = lo[1]; bv[u][2] = hi[0]; bv[u][3] = hi[1];
            } else {
#if defined(PT_ADJ_EXP) && PT_ADJ_EXP == 1
                // ABLATION (wrong results, timing only; profiles/r06l_*): the four scalar gathers of a group as ONE aligned 16-byte read -- what a
                // layout with four shifted copies of every residual map could reach at best
                const f32x4 qv = *(const f32x4*)__builtin_assume_aligned((const char*)maps + (((unsigned)cell[u][0] + tapoff4) & ~15u), 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) bv[u][k] = qv[k];
#else
#pragma unroll
                for (int k = 0; k < 4; ++k) bv[u][k] = *(const float*)((const char*)maps + ((unsigned)cell[u][k] + tapoff4));
#endif
            }
        };
        cells(0);
        if (UM > 1) cells(1);
        gather(0);
#pragma unroll
        for (int u = 0; u < UM; ++u) {
            if (u + PD < UM) issue(u + PD);
            if (u + 2 < UM) cells(u + 2);
            if (u + 1 < UM) gather(u + 1);
            // masked quads multiply a finite, re-read feature value by a gathered zero
            accA = mfma16(av[u][0], bv[u][0], accA);
            accB = mfma16(av[u][1], bv[u][1], accB);
            accA = mfma16(av[u][2], bv[u][2], accA);
            accB = mfma16(av[u][3], bv[u][3], accB);
        }
    };
    if (PT_ADJ_G2 && (hW & 1) == 0) mfma_loop(std::true_type{});    // uniform
    else mfma_loop(std::false_type{});
    PT_STAMP_A(a, 5);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * kq + r) * 16 + j] = accA[r] + accB[r];
    __syncthreads();
    PT_STAMP_A(a, 6);
    if (threadIdx.x < 256) {
        const int e = threadIdx.x, row = e >> 4, tap = e & 15;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < PT_ADJ_WAVES; ++w) s += red[w][e];
        if (tap < KK) a.gpart[(long)ks * a.C * KK + (long)(cb * 16 + row) * KK + tap] = s;
    }
    // w_t = w_{t-1} - step*alpha*g (:160): nothing in this launch reads it, so its two loads (requested behind the feature
    // loads, the counter retires in order) are only waited for here
    if (wupd && (int)threadIdx.x < 16 * KK) a.sd.w_iters[(long)a.t * a.sd.CKK + wge] = w_prev - astep * g_prev;
    PT_STAMP_A(a, 7);
    PT_STAMP_B(a, 7);
}

static void adj2_fill(const PtFast& p, Adj2Args& a, const float* feat, long stride_n, float* gpart) {
    a.feat = feat; a.stride_n = stride_n; a.gpart = gpart;
    a.n = p.n; a.C = p.C; a.H = p.H; a.W = p.W; a.KH = p.KH; a.KW = p.KW; a.OH = p.OH; a.OW = p.OW;
    a.NG = p.NG; a.gper = p.gper; a.U = p.U; a.bpx = p.bpx; a.PH = p.PH; a.PW = p.PW; a.ns_max = p.ns_max; a.zn = p.zn;
    a.inp = nullptr; a.t = 0; a.want_loss = 0;
    PT_STAMP_SET(a);
}

// Can the first adjoint pass of this solve do the init stage itself (k_adj2<.., INIT>)?
bool pt_adj2_init_fusable(const PtFast& p, const SdArgs& sd) {
    if (sd.kind != PT_SD_DIMP && sd.kind != PT_SD_DIMP_L2) return false;
    if (p.E > 9 || sd.KS > PT_INIT_MAXKS || PT_ADJ_WAVES != 8) return false;
    if (sd.kind == PT_SD_DIMP) {
        if (sd.num_bins < 1 || sd.num_bins > 127 || 3 * sd.num_bins > PT_ADJ_WAVES * 64) return false;
        if (sd.mask_lut != sd.label_lut + sd.num_bins || sd.spatial_lut != sd.label_lut + 2 * sd.num_bins) return false;   // one array
    }
    return (long)sd.KS * sd.n * sd.OO * 4 < (1L << 31);
}

template <int V>
static int adj2_dispatch(const PtFast& p, const Adj2Args& a, hipStream_t st, bool fuse_init = false) {
    dim3 grid(p.CB * p.KSPL), block(PT_ADJ_WAVES * 64);
    if (p.H > 255 || p.W > 255 || p.U > 31 || p.bpx > 15 || a.t > 127 || p.n > 65535 || p.gper > 65535 || p.OO > 65535 || p.CB > 255 ||
        a.sd.KS > 127 || p.PH > 63 || p.PW > 63 || p.zn > 255 || p.ns_max > 63 || p.KH > 7 || p.KW > 7)
        return PT_ERR_UNSUPPORTED;
    // (fused init stage: t is 0 and its bits carry the number of look-up table bins)
    const unsigned g1 = (unsigned)p.H | ((unsigned)p.W << 8) | ((unsigned)p.U << 16) | ((unsigned)p.bpx << 21) |
                        ((unsigned)(fuse_init ? (a.sd.kind == PT_SD_DIMP ? a.sd.num_bins : 0) : a.t) << 25);
    const unsigned g2 = (unsigned)p.n | ((unsigned)p.gper << 16);
    const unsigned g3 = (unsigned)p.OO | ((unsigned)p.CB << 16) | ((unsigned)(V == V_PLAIN ? 0 : a.sd.KS) << 24);
    const unsigned g4 = (unsigned)p.PH | ((unsigned)p.PW << 6) | ((unsigned)p.zn << 12) | ((unsigned)p.ns_max << 20) | ((unsigned)p.KH << 26) |
                        ((unsigned)p.KW << 29);
    const float* pkp = V == V_PLAIN ? a.inp : a.sd.pk;
    const float* qsp = V == V_PLAIN ? nullptr : a.sd.qs;
    const float* anp = V == V_PLAIN ? nullptr : a.sd.anum;
    Adj2Late l = {};
    l.gpart = (pt_gf)a.gpart; l.C = a.C; l.KH = a.KH; l.KW = a.KW; l.OH = a.OH; l.OW = a.OW; l.PH = a.PH; l.PW = a.PW;
    l.ns_max = a.ns_max; l.zn = a.zn; l.t = a.t; l.want_loss = a.want_loss;
    const SdArgs& sd = a.sd;
    l.sd.n = sd.n; l.sd.CKK = sd.CKK; l.sd.has_sw = sd.has_sw; l.sd.has_softmax_reg = sd.has_softmax_reg;
    l.sd.step = sd.step; l.sd.reg = sd.reg; l.sd.alpha_eps = sd.alpha_eps; l.sd.act_param = sd.act_param; l.sd.softmax_reg = sd.softmax_reg;
    l.sd.sw = (pt_gcf)sd.sw; l.sd.s_in = (pt_gcf)sd.s_in; l.sd.w0 = (pt_gcf)sd.w0; l.sd.mask = (pt_gf)sd.mask; l.sd.sws = (pt_gf)sd.sws;
    l.sd.s = (pt_gf)sd.s; l.sd.sg = (pt_gf)sd.sg; l.sd.lms = (pt_gf)sd.lms; l.sd.g = (pt_gf)sd.g; l.sd.lossp = (pt_gf)sd.lossp;
    l.sd.w_iters = (pt_gf)sd.w_iters;
#ifdef PT_STAMPS
    l.stamps = a.stamps;
#endif
    if constexpr (V == V_DIMP_RELU || V == V_DIMP_BENT || V == V_L2) {
        if (fuse_init) {
            if (!pt_adj2_init_fusable(p, sd) || a.t != 0 || a.want_loss) return PT_ERR_UNSUPPORTED;
            Adj2InitLate il;
            il.label = (pt_gf)sd.label; il.cls_scores = (pt_gf)sd.cls_scores; il.cls_peak = (pt_gf)sd.cls_peak; il.cls_bb = (pt_gf)sd.cls_bb;
            il.slot_dyn = sd.cls_slot_dyn; il.slot = (sd.cls_spart && sd.cls_slot >= 0) ? sd.cls_slot : -1; il.mask_act = sd.mask_act;
            il.feat_stride = sd.feat_stride; il.bin_disp = sd.bin_disp; il.gauss_sigma = sd.gauss_sigma; il.hinge_thr = sd.hinge_thr;
            const float* lut3 = sd.kind == PT_SD_DIMP ? sd.label_lut : nullptr;
            const size_t lds = p.adj_lds + (size_t)3 * (sd.kind == PT_SD_DIMP ? sd.num_bins : 0) * sizeof(float);
#define PT_A2_INIT a.feat, a.stride_n, (const float*)sd.spart, sd.bb, lut3, g1, g2, g3, g4, l, il
            if (p.E == 6) hipLaunchKernelGGL((k_adj2<V, 6, 16, true>), grid, block, lds, st, PT_A2_INIT);
            else if (p.U <= 12) hipLaunchKernelGGL((k_adj2<V, 9, 12, true>), grid, block, lds, st, PT_A2_INIT);
            else if (p.U <= 16) hipLaunchKernelGGL((k_adj2<V, 9, 16, true>), grid, block, lds, st, PT_A2_INIT);
            else hipLaunchKernelGGL((k_adj2<V, 9, PT_ADJ_UMAX_WIDE, true>), grid, block, lds, st, PT_A2_INIT);
#undef PT_A2_INIT
            return PT_OK;
        }
    } else if (fuse_init) return PT_ERR_UNSUPPORTED;
#define PT_A2_HOT a.feat, a.stride_n, pkp, qsp, anp, g1, g2, g3, g4, l, Adj2InitLate{}
    if (p.E == 6 && p.U <= 12 && PT_ADJ_WAVES != 8) hipLaunchKernelGGL((k_adj2<V, 6, 12>), grid, block, p.adj_lds, st, PT_A2_HOT);   // experiment builds only
    else if (p.E == 6) hipLaunchKernelGGL((k_adj2<V, 6, 16>), grid, block, p.adj_lds, st, PT_A2_HOT);
    else if (p.E == 9 && p.U <= 12) hipLaunchKernelGGL((k_adj2<V, 9, 12>), grid, block, p.adj_lds, st, PT_A2_HOT);
    else if (p.E == 9 && p.U <= 16) hipLaunchKernelGGL((k_adj2<V, 9, 16>), grid, block, p.adj_lds, st, PT_A2_HOT);
    else if (p.E == 9) hipLaunchKernelGGL((k_adj2<V, 9, PT_ADJ_UMAX_WIDE>), grid, block, p.adj_lds, st, PT_A2_HOT);
    else hipLaunchKernelGGL((k_adj2<V, 16, 16>), grid, block, p.adj_lds, st, PT_A2_HOT);
#undef PT_A2_HOT
    return PT_OK;
}

int pt_launch_adj2_plain(const PtFast& p, const float* feat, long stride_n, const float* inp, float* gpart,
                         hipStream_t st) {
    if (((uintptr_t)feat % 16) || (stride_n % 4) || (long)p.n * stride_n * 4 >= (1L << 31)) return PT_ERR_UNSUPPORTED;
    Adj2Args a;
    adj2_fill(p, a, feat, stride_n, gpart);
    a.inp = inp;
    a.sd = SdArgs();
    const int rc = adj2_dispatch<V_PLAIN>(p, a, st);
    if (rc) return rc;
    PT_CHECK_LAUNCH();
    return PT_OK;
}

int pt_launch_adj2_sd(const PtFast& p, const float* feat, long stride_n, const SdArgs& sd, int t, int want_loss,
                      hipStream_t st, bool fuse_init) {
    if (((uintptr_t)feat % 16) || (stride_n % 4) || (long)p.n * stride_n * 4 >= (1L << 31)) return PT_ERR_UNSUPPORTED;
    Adj2Args a;
    adj2_fill(p, a, feat, stride_n, sd.gpart);
    a.sd = sd;
    a.t = t;
    a.want_loss = want_loss;
    int rc;
    if (sd.kind == PT_SD_PRDIMP) rc = adj2_dispatch<V_PRDIMP>(p, a, st, fuse_init);
    else if (sd.kind == PT_SD_DIMP_L2) rc = adj2_dispatch<V_L2>(p, a, st, fuse_init);
    else if (sd.score_act == PT_ACT_BENTPAR) rc = adj2_dispatch<V_DIMP_BENT>(p, a, st, fuse_init);
    else rc = adj2_dispatch<V_DIMP_RELU>(p, a, st, fuse_init);
    if (rc) return rc;
    PT_CHECK_LAUNCH();
    return PT_OK;
}
