// ATOM online filter update: `ConjugateGradient.run` + `ConjugateGradientBase.run_CG`
// (pytracking/libs/optimization.py:227-289, 72-163) specialised to `ConvProblem`
// (pytracking/tracker/atom/optim.py:71-99) with the MLU response activation
// (ltr/models/layers/activation.py:20-29; pytracking/tracker/atom/atom.py:451-452,467-468).
//
// The reference obtains A(p) = J^T J p with two torch.autograd.grad calls per CG iteration
// (optimization.py:278-280).  Here the Gauss-Newton operator is explicit:
//     residuals  f(x) = [ sqrt(sw_i) * (MLU(conv_same(samples, x)) - y) ,  sqrt(lambda) * x ]
//     J p   = [ d .* conv_same(samples, p) , sqrt(lambda) p ],   d = sqrt(sw_i) * MLU'(s0)
//     J^T u = adj(samples, d .* u_data) + sqrt(lambda) u_reg
// so one CG iteration is one correlation pass + one adjoint pass over the sample memory (the same two
// kernels the DiMP solver uses, with the 'same' crop OH=H, OW=W of pytracking/libs/operation.py:17-32),
// and the scalar recurrences run in a single-workgroup kernel without host round trips (the reference's
// `check_zero(...).item()` at optimization.py:108 is a device-side flag here).
#include "common.h"
#include "pt_internal.h"
#include "rbuild.h"
#include <algorithm>

struct CgArgs {
    int n, C, H, W, K, HW, CKK, KS, KSPL, num_iter, fletcher_reeves;
    float lambda, act_min, forget;
    float* x;
    const float *y, *sw;
    float *d, *spart, *R, *gpart, *r, *delta, *scal;   // scal: [0]=rho [1]=stop
    float *p, *r_prev, *st;                            // persistent: st[0]=rho, st[1]=has_p
};

__device__ __forceinline__ float mlu_f(float x, float mn) {
    const float yv = x >= 0.f ? x : x / mn;                 // leaky_relu(x, 1/min_val)
    return yv > 0.f ? yv : mn * (expf(yv) - 1.f);           // elu(., min_val)
}
__device__ __forceinline__ float mlu_d(float x, float mn) { return x >= 0.f ? 1.f : expf(x / mn); }

// mode 0: linearisation point (s0 from conv(x)) -> d, input of J^T f0.   mode 1: J p -> input of J^T (J p).
__global__ __launch_bounds__(512) void k_atom_pw(CgArgs a, int mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int i = blockIdx.x;
    const long base = (long)i * a.HW;
    const float sq = sqrtf(a.sw[i]);
    for (int o = threadIdx.x; o < a.HW; o += blockDim.x) {
        float sv = 0.f;
#pragma unroll 8
        for (int k = 0; k < a.KS; ++k) sv += a.spart[((long)k * a.n + i) * a.HW + o];
        if (mode == 0) {
            const float dv = sq * mlu_d(sv, a.act_min);
            a.d[base + o] = dv;
            lds[o] = dv * (sq * (mlu_f(sv, a.act_min) - a.y[base + o]));
        } else {
            const float dv = a.d[base + o];
            lds[o] = dv * (dv * sv);
        }
    }
    __syncthreads();
    pt_build_R_sample(lds, a.R, i, a.n, a.H, a.W, a.K, a.K, a.H, a.W);
}

__device__ float cg_dot(const float* u, const float* v, int n, float* scratch) {
    float acc = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) acc += u[e] * v[e];
    return block_sum(acc, scratch);
}

// Computes the next search direction from the residual r (optimization.py:100-125).
__device__ void cg_direction(const CgArgs& a, float* scratch) {
    const float rho1 = a.st[0];
    const float rho = cg_dot(a.r, a.r, a.CKK, scratch);          // z = M2(M1(r)) = r for ConvProblem
    float rho2 = 0.f;
    const bool has_p = a.st[1] != 0.f;
    if (has_p && !a.fletcher_reeves) rho2 = cg_dot(a.r_prev, a.r, a.CKK, scratch);
    __syncthreads();
    if (rho == 0.f) {                                            // :108-113  stop, keep what we have
        for (int e = threadIdx.x; e < a.CKK; e += blockDim.x) a.x[e] += a.delta[e];
        if (threadIdx.x == 0) { a.st[0] = rho; a.scal[1] = 1.f; }
        return;
    }
    float beta = 0.f;
    if (has_p) {
        beta = a.fletcher_reeves ? rho / rho1 : (rho - rho2) / rho1;   // :118-122
        beta = fmaxf(beta, 0.f);                                       // :124
    }
    for (int e = threadIdx.x; e < a.CKK; e += blockDim.x) a.p[e] = has_p ? a.r[e] + beta * a.p[e] : a.r[e];
    if (threadIdx.x == 0) { a.st[0] = rho; a.st[1] = 1.f; }
}

// phase 0: right-hand side b = -J^T f0 (optimization.py:262-265), state reset/forgetting (:82-85), first direction.
// phase 1: step ii: q = A(p), alpha, delta, residual update (:127-146), then next direction or x += delta (:259-260).
__global__ __launch_bounds__(1024) void k_atom_vec(CgArgs a, int phase, int ii) {
    __shared__ float scratch[16];
    if (phase == 0) {
        if (threadIdx.x == 0) {
            a.scal[1] = 0.f;
            if (a.forget == 0.f) { a.st[0] = 1.f; a.st[1] = 0.f; }
            else if (a.st[1] != 0.f) a.st[0] = a.st[0] / a.forget;
        }
        for (int e = threadIdx.x; e < a.CKK; e += blockDim.x) {
            float v = 0.f;
#pragma unroll 8
            for (int k = 0; k < a.KSPL; ++k) v += a.gpart[(long)k * a.CKK + e];   // unrolled: 8 partial loads in flight
            a.r[e] = -(v + a.lambda * a.x[e]);
            a.delta[e] = 0.f;
        }
        __syncthreads();
        cg_direction(a, scratch);
        return;
    }
    if (a.scal[1] != 0.f) return;
    // q = J^T J p
    float acc = 0.f;
    for (int e = threadIdx.x; e < a.CKK; e += blockDim.x) {
        float v = 0.f;
#pragma unroll 8
        for (int k = 0; k < a.KSPL; ++k) v += a.gpart[(long)k * a.CKK + e];
        v += a.lambda * a.p[e];
        a.gpart[e] = v;                       // slice 0 now holds q (each thread only rewrites what it read)
        acc += a.p[e] * v;
    }
    const float pq = block_sum(acc, scratch);
    const float alpha = a.st[0] / pq;                                   // :131
    const bool more = ii < a.num_iter - 1;
    for (int e = threadIdx.x; e < a.CKK; e += blockDim.x) {
        const float re = a.r[e];
        if (!a.fletcher_reeves) a.r_prev[e] = re;                       // :136-137
        a.delta[e] += alpha * a.p[e];                                   // :140-143
        if (more) a.r[e] = re - alpha * a.gpart[e];                     // :145-146
    }
    __syncthreads();
    if (more) {
        cg_direction(a, scratch);
    } else {
        for (int e = threadIdx.x; e < a.CKK; e += blockDim.x) a.x[e] += a.delta[e];
    }
}

// ----------------------------------------------------------------------------------------------------
// Fast path (round 3) for ATOM's shape -- C = 64 compressed channels, 4x4 filter: three launches per CG step instead of four
//   k_adj2 (fast_passes.hip, plain)   J^T u as 64 position-slice partials per 16-channel block            [256 workgroups]
//   k_acg_red                         sum of the partials (+ lambda * p / x), partial dot products p.q      [CKK/64 workgroups]
//   k_acg_fwd                         prologue: the CG recurrences (alpha, delta, r, rho, beta, next direction p) on the 1024
//                                     filter elements, computed by EVERY workgroup in the same fixed order (workgroup 0 stores
//                                     the state); body: conv_same(samples, p) for one sample with ALL channels, so the score
//                                     map is complete in the workgroup; epilogue: the pointwise stage d .* (d .* s) -> the
//                                     residual map the next adjoint reads                                   [n workgroups]
// The generic path below (any C, K) keeps the separate pointwise / vector kernels and the im2col buffer.
// State block (floats): p [CKK] | r_prev [CKK] | st [16]: rho, has_p, stop | r [CKK] | delta [CKK]; the caller's cg_state is the
// same layout's prefix (p, r_prev, st), two more blocks live in the workspace and alternate between steps.
// ----------------------------------------------------------------------------------------------------
#define ACG_ST 16
struct AcgLate {
    pt_gcf y, sw;
    pt_gf d, rmap, nxt, x;
    pt_gcf q, pqp;
    int n, H, W, OH, OW, fr, nred, pad;
    float act_min, forget;
};
static_assert(sizeof(AcgLate) <= 2 * 64, "AcgLate: two 16-dword blocks");

// phase 1: r = -(sum_k gpart[k] + lambda x) (optimization.py:262-265).  phase 2: q = sum_k gpart[k] + lambda p, partial p.q.
__global__ __launch_bounds__(256) void k_acg_red(const float* __restrict__ gpart, const float* __restrict__ px, float* __restrict__ out,
                                                 float* __restrict__ pqp, const float* __restrict__ st, int KSPL, int CKK, int phase,
                                                 float lambda) {
    __shared__ float part[4][64];
    if (phase == 2 && st[2] != 0.f) return;                              // stopped (rho == 0 earlier)
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6, e = blockIdx.x * 64 + el;
    const int ec = min(e, CKK - 1);
    const float pv = px[ec];
    float acc = 0.f;
    for (int k0 = g; k0 < KSPL; k0 += 64) {                              // partials g, g+4, ...: 16 loads in flight (the tracker's 64
        float v[16];                                                     // position slices: ONE round trip; round 3 took two of 8)
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = gpart[(long)min(k0 + 4 * u, KSPL - 1) * CKK + ec];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += k0 + 4 * u < KSPL ? v[u] : 0.f;
    }
    part[g][el] = acc;
    __syncthreads();
    if (g == 0) {
        const float v = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el] + lambda * pv;
        if (e < CKK) out[e] = phase == 1 ? -v : v;
        if (phase == 2) {
            const float d = wave_sum(e < CKK ? pv * v : 0.f);
            if (el == 0) pqp[blockIdx.x] = d;
        }
    }
}

// FOLD (round 5): the partial-sum launch between the adjoint and this kernel is gone -- in phases 1 / 2 `h_in` points at the adjoint's
// KSPL position-slice partials (KSPL in h_geo bits 16..22) and every workgroup adds them itself (256 KB from its XCD's L2, one round of
// 16-byte loads issued in front of everything else, two halves met in LDS in a fixed order), forms r = -(sum + lambda x) or
// q = sum + lambda p and the dot product p.q with the block reduction the recurrences already use.  13 instead of 19 launches per
// 5-iteration update.  MEASURED SLOWER (106.4 vs 97.7 us, profiles/r05p_atom_fold_ab.txt) and therefore off unless PT_ACG_FOLD=1.
#ifndef PT_ACG_INFLIGHT
#define PT_ACG_INFLIGHT 16                 // 16-byte partial loads in flight per thread in the folded sum (32 per thread at KSPL = 64)
#endif
template <bool LEFT, bool FOLD>
__global__ __launch_bounds__(640, 2) void k_acg_fwd(const float* h_feat, long h_stride, const float* h_cur, const float* h_in, unsigned h_dims,
                                                    unsigned h_geo, float h_lambda, AcgLate l_arg) {
    // h_in: phase 0 the filter x; phase 1 the right-hand side r (k_acg_red); phase 2 q (FOLD: the adjoint's partials in phases 1, 2).
    // h_cur: state block of the previous step.
    // h_dims = C << 16 | H*W;  h_geo = tiles | TF << 5 | rem << 10 | phase << 14 | KSPL << 16
    extern __shared__ __attribute__((aligned(16))) float lds[];     // afilt[C][16] | T[2][16][HWp]
    __shared__ float scratch[16];
    constexpr int NK = 8;
    const int C = (int)(h_dims >> 16), HW = (int)(h_dims & 0xffffu);
    const int tiles = (int)(h_geo & 31u), TF = (int)((h_geo >> 5) & 31u), rem = (int)((h_geo >> 10) & 15u), phase = (int)((h_geo >> 14) & 3u);
    const int HWp = 64 * (TF + (rem > 0 ? 1 : 0)) + 4, nthreads = 2 * tiles * 64, CKK = C * 16, nsl = CKK;
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane >> 4, j = lane & 15;
    const int h = wave >= tiles ? 1 : 0, t = wave - h * tiles;
    float* __restrict__ afilt = lds;
    float* __restrict__ Tl = lds + nsl + (long)h * 16 * HWp;
    const float* st_c = h_cur + 2 * CKK;
    if (phase == 2 && st_c[2] != 0.f) {                             // stopped (rho == 0 earlier): hand the state on, nothing else to do
        if (blockIdx.x == 0) {
            const AcgLate ls = pt_late_args<AcgLate>(48);
            for (int e = threadIdx.x; e < 4 * CKK + ACG_ST; e += nthreads) ls.nxt[e] = h_cur[e];
        }
        return;
    }

    // ---- FOLD: this workgroup's share of the partial sums goes out first (the memory counter retires in order: the feature loads
    //      behind it stay in flight while these are waited for)
    const int ncol = CKK >> 2;                                      // 16-byte columns of a partial
    const int fsub = nthreads / ncol;                               // partial subsets summed side by side (2 for 1024 elements)
    const int fs = (int)threadIdx.x / ncol, fcol = (int)threadIdx.x - fs * ncol;
    f32x4 facc = {0.f, 0.f, 0.f, 0.f};
    if (FOLD && phase >= 1 && fs < fsub) {
        const int KSPL = (int)((h_geo >> 16) & 127u);
        const f32x4* __restrict__ gp = (const f32x4*)h_in + fcol;
        for (int k0 = fs; k0 < KSPL; k0 += PT_ACG_INFLIGHT * fsub) {
            f32x4 v[PT_ACG_INFLIGHT];
#pragma unroll
            for (int u = 0; u < PT_ACG_INFLIGHT; ++u) v[u] = gp[(long)min(k0 + u * fsub, KSPL - 1) * ncol];
#pragma unroll
            for (int u = 0; u < PT_ACG_INFLIGHT; ++u) {
                const bool ok = k0 + u * fsub < KSPL;
#pragma unroll
                for (int r = 0; r < 4; ++r) facc[r] += ok ? v[u][r] : 0.f;
            }
        }
    }
    // ---- operands of the vector stage (2 filter elements per thread) and the first feature tiles
    const int e0 = threadIdx.x, e1 = threadIdx.x + nthreads;
    const int c0 = min(e0, CKK - 1), c1 = min(e1, CKK - 1);
    float in0 = 0.f, in1 = 0.f;
    if (!FOLD || phase == 0) { in0 = h_in[c0]; in1 = h_in[c1]; }
    float p0 = 0.f, p1 = 0.f, r0 = 0.f, r1 = 0.f, d0 = 0.f, d1 = 0.f, rp0 = 0.f, rp1 = 0.f, st0 = 0.f, st1 = 0.f;
    if (phase >= 1) {
        p0 = h_cur[c0]; p1 = h_cur[c1];
        st0 = st_c[0]; st1 = st_c[1];
        rp0 = h_cur[CKK + c0]; rp1 = h_cur[CKK + c1];
        if (phase == 2) {
            r0 = h_cur[2 * CKK + ACG_ST + c0]; r1 = h_cur[2 * CKK + ACG_ST + c1];
            d0 = h_cur[3 * CKK + ACG_ST + c0]; d1 = h_cur[3 * CKK + ACG_ST + c1];
        }
    }
    constexpr int CD = 2;
    const int cbase = 4 * (h * NK) + kq;
    const int pos = 64 * t + 4 * j;
    const bool pv = pos < HW;
    const float* __restrict__ fi = h_feat + (long)i * h_stride;
    const __amdgpu_buffer_rsrc_t fr = pt_rsrc(fi, (unsigned)C * HW * 4u);
    const unsigned fo = ((unsigned)cbase * HW + (pv ? pos : 0)) * 4u;
    const int lpos = 64 * TF + j;
    const bool lv = LEFT && t == 0 && j < 4 * rem;
    const unsigned lo = lv ? ((unsigned)cbase * HW + lpos) * 4u : 0xFFFFFFF0u - 64u * (unsigned)HW * 4u;
    f32x4 bq[NK];
    float bl[NK];
    if (LEFT && t == 0) {
#pragma unroll
        for (int k = 0; k < CD; ++k) bl[k] = pt_bload1(fr, lo + (unsigned)(4 * k) * HW * 4u);
    }
#pragma unroll
    for (int k = 0; k < CD; ++k) bq[k] = pt_bload4(fr, fo + (unsigned)(4 * k) * HW * 4u);
    __builtin_amdgcn_sched_barrier(0);
    const AcgLate l = pt_late_args<AcgLate>(48);                    // 4 pointers / longs + 3 dwords = 44 bytes, 8-aligned
    // pointwise operands of this sample's score elements (one per thread)
    const int OO = l.OH * l.OW;
    const int oc = min((int)threadIdx.x, OO - 1);
    const long qo = (long)i * OO + oc;
    const float sq = sqrtf(l.sw[i]);
    const float pw_in = phase == 0 ? l.y[qo] : l.d[qo];
    float pq = 0.f;
    if (!FOLD && phase == 2) pq = lane < l.nred ? l.pqp[lane] : 0.f;
    if (FOLD && phase >= 1) {                                       // the halves meet in LDS (the tap-plane region is free until the MFMAs)
        float* __restrict__ fsum = lds + nsl;
        if (fs < fsub) *(f32x4*)(fsum + (long)fs * CKK + 4 * fcol) = facc;
        __syncthreads();
        float s0 = 0.f, s1 = 0.f;
        for (int q = 0; q < fsub; ++q) { s0 += fsum[(long)q * CKK + c0]; s1 += fsum[(long)q * CKK + c1]; }
        if (phase == 1) {                                           // r = -(sum_k gpart[k] + lambda x)   (optimization.py:262-265)
            in0 = -(s0 + h_lambda * l.x[c0]); in1 = -(s1 + h_lambda * l.x[c1]);
        } else {                                                    // q = sum_k gpart[k] + lambda p
            in0 = s0 + h_lambda * p0; in1 = s1 + h_lambda * p1;
        }
        __syncthreads();                                            // fsum is the tap-plane buffer again from here on
    }

    // ---- the conjugate-gradient recurrences (optimization.py:100-146), identical in every workgroup
    float f0, f1;                                                   // this thread's two elements of the filter operand
    float stop = 0.f;
    if (phase == 0) {
        f0 = in0; f1 = in1;
    } else {
        float rn0, rn1, dn0 = 0.f, dn1 = 0.f, rpn0, rpn1, rho1;
        bool has_p;
        if (phase == 1) {                                           // state reset / forgetting (:82-85), r = right-hand side
            if (l.forget == 0.f) { rho1 = 1.f; has_p = false; }
            else { has_p = st1 != 0.f; rho1 = has_p ? st0 / l.forget : st0; }
            rn0 = in0; rn1 = in1;
            rpn0 = rp0; rpn1 = rp1;
        } else {                                                    // step: alpha, delta, residual (:127-146)
            float pqs;
            if (FOLD) pqs = block_sum((e0 < CKK ? p0 * in0 : 0.f) + (e1 < CKK ? p1 * in1 : 0.f), scratch, nthreads);
            else pqs = wave_sum(pq);
            const float alpha = st0 / pqs;                          // :131
            dn0 = d0 + alpha * p0; dn1 = d1 + alpha * p1;           // :140-143
            rn0 = r0 - alpha * in0; rn1 = r1 - alpha * in1;         // :145-146
            rpn0 = l.fr ? rp0 : r0; rpn1 = l.fr ? rp1 : r1;         // :136-137
            rho1 = st0; has_p = true;
        }
        const bool v0 = e0 < CKK, v1 = e1 < CKK;
        const float rho = block_sum((v0 ? rn0 * rn0 : 0.f) + (v1 ? rn1 * rn1 : 0.f), scratch, nthreads);   // z = r for ConvProblem
        float rho2 = 0.f;
        if (has_p && !l.fr) rho2 = block_sum((v0 ? rpn0 * rn0 : 0.f) + (v1 ? rpn1 * rn1 : 0.f), scratch, nthreads);
        float beta = 0.f;
        if (rho == 0.f) {                                           // :108-113  stop, keep what we have
            stop = 1.f;
        } else if (has_p) {
            beta = l.fr ? rho / rho1 : (rho - rho2) / rho1;         // :118-122
            beta = fmaxf(beta, 0.f);                                // :124
        }
        f0 = has_p ? rn0 + beta * p0 : rn0;
        f1 = has_p ? rn1 + beta * p1 : rn1;
        if (i == 0) {                                               // one workgroup stores the state of this step
            const pt_gf nx = l.nxt;
            if (stop != 0.f) {
                if (v0) l.x[e0] += dn0;
                if (v1) l.x[e1] += dn1;
                f0 = p0; f1 = p1;                                   // the direction is kept as it was
            }
            if (v0) { nx[e0] = f0; nx[CKK + e0] = rpn0; nx[2 * CKK + ACG_ST + e0] = rn0; nx[3 * CKK + ACG_ST + e0] = dn0; }
            if (v1) { nx[e1] = f1; nx[CKK + e1] = rpn1; nx[2 * CKK + ACG_ST + e1] = rn1; nx[3 * CKK + ACG_ST + e1] = dn1; }
            if (threadIdx.x == 0) {
                nx[2 * CKK + 0] = rho;
                nx[2 * CKK + 1] = (stop != 0.f && !has_p) ? 0.f : 1.f;
                nx[2 * CKK + 2] = stop;
            }
        }
        if (stop != 0.f) return;                                    // uniform: the residual maps are not needed any more
    }
    if (e0 < nsl) afilt[e0] = f0;
    if (e1 < nsl) afilt[e1] = f1;
    __syncthreads();

    // ---- T[half][tap][pos] = sum_c p[c][tap] * feat[i][c][pos] (as k_corr2)
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0}, accL = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (k + CD < NK) {
            bq[k + CD] = pt_bload4(fr, fo + (unsigned)(4 * (k + CD)) * HW * 4u);
            if (LEFT && t == 0) bl[k + CD] = pt_bload1(fr, lo + (unsigned)(4 * (k + CD)) * HW * 4u);
        }
        const float av = afilt[(4 * (h * NK + k) + kq) * 16 + j];
        acc0 = mfma16(av, bq[k][0], acc0);
        acc1 = mfma16(av, bq[k][1], acc1);
        acc2 = mfma16(av, bq[k][2], acc2);
        acc3 = mfma16(av, bq[k][3], acc3);
        if (LEFT && t == 0) accL = mfma16(av, bl[k], accL);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r;
        f32x4 v = {acc0[r], acc1[r], acc2[r], acc3[r]};
        *(f32x4*)(Tl + row * HWp + pos) = v;
        if (LEFT && t == 0) Tl[row * HWp + lpos] = accL[r];
    }
    __syncthreads();

    // ---- shift-and-add ('same' crop, pytracking/libs/operation.py:17-32) + the pointwise stage of the Gauss-Newton operator
    const float* __restrict__ T0 = lds + nsl;
    const float* __restrict__ T1 = T0 + 16 * HWp;
    const int o = threadIdx.x;
    if (o < OO) {
        const int y = o / l.OW, xx0 = o - y * l.OW;
        float tv[16];
        if (HWp >= HW + 2 * l.W + 2 && nsl >= 2 * l.W + 2) {        // uniform.  As k_corr2 (fast_passes.hip, round 6): uniform tap steps, all 32 LDS reads
            const int su = 4 * HWp + l.W, sv_ = HWp + 1;            // requested before the first use, validity as row / column bit masks; same sums
            const int base = (y - 2) * l.W + (xx0 - 2);
            float t0[16], t1[16];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int idx = base + (u * su + v * sv_);
                    t0[u * 4 + v] = T0[idx];
                    t1[u * 4 + v] = T1[idx];
                }
            int rok[4], cok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) rok[u] = (unsigned)(y + u - 2) < (unsigned)l.H ? -1 : 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) cok[v] = (unsigned)(xx0 + v - 2) < (unsigned)l.W ? -1 : 0;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                tv[q] = __builtin_bit_cast(float, __builtin_bit_cast(int, t0[q] + t1[q]) & (rok[q >> 2] & cok[q & 3]));
        } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int yy = y + u - 2;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int xx = xx0 + v - 2;
                const bool ok = (unsigned)yy < (unsigned)l.H && (unsigned)xx < (unsigned)l.W;
                const int idx = (u * 4 + v) * HWp + (ok ? yy * l.W + xx : 0);
                const float tsum = T0[idx] + T1[idx];
                tv[u * 4 + v] = ok ? tsum : 0.f;
            }
        }
        }
        float sv = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) sv += tv[q];
        if (phase == 0) {                                           // linearisation point: d, input of J^T f0
            const float dv = sq * mlu_d(sv, l.act_min);
            l.d[qo] = dv;
            l.rmap[qo] = dv * (sq * (mlu_f(sv, l.act_min) - pw_in));
        } else {                                                    // J p -> input of J^T (J p)
            l.rmap[qo] = pw_in * (pw_in * sv);
        }
    }
}

// FOLD variant of the last step: p.q from the adjoint's partials directly (1024 threads = 16-byte columns x 4 partial subsets, one round
// of loads, met in LDS), then as k_acg_final
__global__ __launch_bounds__(1024) void k_acg_final_fold(const float* __restrict__ cur, const float* __restrict__ gpart, int KSPL, float lambda,
                                                         float* __restrict__ x, float* __restrict__ cg_state, int CKK, int fr) {
    __shared__ __attribute__((aligned(16))) float fsum[4 * 1280];
    __shared__ float scratch[16];
    const float* st = cur + 2 * CKK;
    const bool stopped = st[2] != 0.f;
    const int ncol = CKK >> 2, fsub = min(4, 1024 / ncol);
    const int fs = (int)threadIdx.x / ncol, fcol = (int)threadIdx.x - fs * ncol;
    float alpha = 0.f;
    if (!stopped) {                                                  // uniform
        f32x4 facc = {0.f, 0.f, 0.f, 0.f};
        if (fs < fsub) {
            const f32x4* __restrict__ gp = (const f32x4*)gpart + fcol;
            for (int k0 = fs; k0 < KSPL; k0 += 16 * fsub) {
                f32x4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = gp[(long)min(k0 + u * fsub, KSPL - 1) * ncol];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const bool ok = k0 + u * fsub < KSPL;
#pragma unroll
                    for (int r = 0; r < 4; ++r) facc[r] += ok ? v[u][r] : 0.f;
                }
            }
            *(f32x4*)(fsum + (long)fs * CKK + 4 * fcol) = facc;
        }
        __syncthreads();
        float acc = 0.f;
        for (int e = threadIdx.x; e < CKK; e += 1024) {
            float sv = 0.f;
            for (int q = 0; q < fsub; ++q) sv += fsum[(long)q * CKK + e];
            const float pv = cur[e];
            acc += pv * (sv + lambda * pv);
        }
        alpha = st[0] / block_sum(acc, scratch);
    }
    for (int e = threadIdx.x; e < CKK; e += blockDim.x) {
        const float pv = cur[e];
        if (!stopped) x[e] += cur[3 * CKK + ACG_ST + e] + alpha * pv;
        cg_state[e] = pv;
        cg_state[CKK + e] = (!stopped && !fr) ? cur[2 * CKK + ACG_ST + e] : cur[CKK + e];
    }
    if (threadIdx.x == 0) { cg_state[2 * CKK] = st[0]; cg_state[2 * CKK + 1] = st[1]; }
}

// the last step of a solve (optimization.py:127-143, 259-260): alpha, delta, x += delta; persistent state back to the caller
__global__ __launch_bounds__(1024) void k_acg_final(const float* __restrict__ cur, const float* __restrict__ q, const float* __restrict__ pqp,
                                                    float* __restrict__ x, float* __restrict__ cg_state, int CKK, int nred, int fr) {
    const float* st = cur + 2 * CKK;
    const int lane = threadIdx.x & 63;
    const bool stopped = st[2] != 0.f;
    float alpha = 0.f;
    if (!stopped) alpha = st[0] / wave_sum(lane < nred ? pqp[lane] : 0.f);
    for (int e = threadIdx.x; e < CKK; e += blockDim.x) {
        const float pv = cur[e];
        if (!stopped) x[e] += cur[3 * CKK + ACG_ST + e] + alpha * pv;
        cg_state[e] = pv;
        cg_state[CKK + e] = (!stopped && !fr) ? cur[2 * CKK + ACG_ST + e] : cur[CKK + e];
    }
    if (threadIdx.x == 0) { cg_state[2 * CKK] = st[0]; cg_state[2 * CKK + 1] = st[1]; }
}

struct CgCarve { size_t d, spart, R, gpart, r, delta, scal, total; };

static CgCarve cg_carve(const PtPlan& p) {
    CgCarve c;
    size_t off = 0;
    auto take = [&](size_t nfl) { size_t o = off; off += pt_align_floats(nfl); return o; };
    c.d = take((size_t)p.n * p.HW);
    c.spart = take(pt_spart_floats(p));
    c.R = take(pt_R_floats(p));
    c.gpart = take(pt_gpart_floats(p));
    c.r = take((size_t)p.C * p.KK);
    c.delta = take((size_t)p.C * p.KK);
    c.scal = take(64);
    c.total = off;
    return c;
}

// workspace of the fast path
struct AcgCarve { size_t d, rmap, gpart, q, pqp, s0, s1, total; };
static AcgCarve acg_carve(const PtFast& f) {
    AcgCarve c;
    size_t off = 0;
    auto take = [&](size_t nfl) { size_t o = off; off += pt_align_floats(nfl); return o; };
    const size_t CKK = (size_t)f.C * f.KK;
    c.d = take((size_t)f.n * f.OO); c.rmap = take((size_t)f.n * f.OO);
    c.gpart = take(pt_fast_gpart_floats(f));
    c.q = take(CKK); c.pqp = take(64);
    c.s0 = take(4 * CKK + ACG_ST); c.s1 = take(4 * CKK + ACG_ST);
    c.total = off;
    return c;
}
static bool acg_fast_ok(const PtFast& f) {
    return f.ok && f.KSC == 1 && f.KK == 16 && f.nh == 2 && f.NK == 8 && f.C * f.KK <= 2 * f.corr_threads && f.C * f.KK <= 64 * 64 &&
           f.OO <= f.corr_threads && f.KSPL <= 64;
}

extern "C" size_t pt_atom_cg_ws_bytes(int n, int C, int H, int W, int K) {
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
    PtPlan p = pt_make_plan(n, C, H, W, K, K, H, W);
    size_t tot = cg_carve(p).total;
    PtFast f = pt_fast_plan(n, C, H, W, K, K, H, W);
    if (acg_fast_ok(f)) tot = std::max(tot, acg_carve(f).total);
    return tot * sizeof(float);
}

static int acg_solve_fast(const PtFast& f, float* x, const float* samples, long stride_n, const float* y, const float* sw, float lambda,
                          float act_min, int num_iter, int fr, float forget, float* cg_state, float* base, hipStream_t st) {
    const AcgCarve cv = acg_carve(f);
    const int n = f.n, CKK = f.C * f.KK, nred = (CKK + 63) / 64;
    float* S[2] = {base + cv.s0, base + cv.s1};
    float* gpart = base + cv.gpart;
    float* q = base + cv.q;
    float* pqp = base + cv.pqp;
    AcgLate l;
    l.y = (pt_gcf)y; l.sw = (pt_gcf)sw; l.d = (pt_gf)(base + cv.d); l.rmap = (pt_gf)(base + cv.rmap); l.x = (pt_gf)x; l.q = (pt_gcf)q;
    l.pqp = (pt_gcf)pqp;
    l.n = n; l.H = f.H; l.W = f.W; l.OH = f.OH; l.OW = f.OW; l.fr = fr; l.nred = nred; l.pad = 0;
    l.act_min = act_min; l.forget = forget;
    const unsigned dims = ((unsigned)f.C << 16) | (unsigned)f.HW;
    const unsigned geo0 = (unsigned)f.tiles | ((unsigned)f.TF << 5) | ((unsigned)f.rem << 10);
    // A/B knob, OFF by default: measured in round 5 and lost -- 106.4 vs 97.7 us per 5-iteration update on the same box
    // (profiles/r05p_atom_fold_ab.txt): every one of the 250 workgroups pulls the 256 KB of partials through its XCD's L2 (8 MB per XCD
    // and step), which costs more than the dependent launch it removes.  Kept as the record of the experiment; parity-tested once with
    // the knob on (tests/test_gpu_parity.py::test_atom_cg_folded_partial_sum_knob).
    static const bool fold_env = [] { const char* e = getenv("PT_ACG_FOLD"); return e && e[0] == '1'; }();
    // the folded sum keeps 4-float columns of a partial per thread and parks fsub * CKK floats in the tap-plane buffer
    const int HWp = 64 * (f.TF + (f.rem > 0 ? 1 : 0)) + 4;
    const bool fold = fold_env && (CKK % 4) == 0 && f.corr_threads >= CKK / 4 && (size_t)(f.corr_threads / (CKK / 4)) * CKK <= (size_t)2 * 16 * HWp &&
                      CKK <= 1280 && ((uintptr_t)gpart % 16) == 0;
    auto fwd = [&](int phase, const float* cur, const float* in, float* nxt) {
        l.nxt = (pt_gf)nxt;
        const unsigned geo = geo0 | ((unsigned)phase << 14) | ((unsigned)f.KSPL << 16);
        if (fold) {
            if (f.left) hipLaunchKernelGGL((k_acg_fwd<true, true>), dim3(n), dim3(f.corr_threads), f.corr_lds, st, samples, stride_n, cur, in, dims, geo, lambda, l);
            else hipLaunchKernelGGL((k_acg_fwd<false, true>), dim3(n), dim3(f.corr_threads), f.corr_lds, st, samples, stride_n, cur, in, dims, geo, lambda, l);
        } else {
            if (f.left) hipLaunchKernelGGL((k_acg_fwd<true, false>), dim3(n), dim3(f.corr_threads), f.corr_lds, st, samples, stride_n, cur, in, dims, geo, lambda, l);
            else hipLaunchKernelGGL((k_acg_fwd<false, false>), dim3(n), dim3(f.corr_threads), f.corr_lds, st, samples, stride_n, cur, in, dims, geo, lambda, l);
        }
    };
    if (fold) {
        // fwd(0) -> adjoint -> [fwd(1 / 2) -> adjoint] x num_iter -> final: the partials go straight from the adjoint to their consumer
        fwd(0, cg_state, x, nullptr);
        PT_CHECK_LAUNCH();
        int rc = pt_launch_adj2_plain(f, samples, stride_n, (const float*)l.rmap, gpart, st);
        if (rc) return rc;
        fwd(1, cg_state, gpart, S[0]);
        PT_CHECK_LAUNCH();
        rc = pt_launch_adj2_plain(f, samples, stride_n, (const float*)l.rmap, gpart, st);
        if (rc) return rc;
        int cur = 0;
        for (int ii = 0; ii < num_iter - 1; ++ii) {
            fwd(2, S[cur], gpart, S[cur ^ 1]);
            PT_CHECK_LAUNCH();
            cur ^= 1;
            rc = pt_launch_adj2_plain(f, samples, stride_n, (const float*)l.rmap, gpart, st);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_acg_final_fold, dim3(1), dim3(1024), 0, st, (const float*)S[cur], (const float*)gpart, f.KSPL, lambda, x, cg_state, CKK, fr);
        PT_CHECK_LAUNCH();
        return PT_OK;
    }
    // linearisation point: s0 = conv(x), d, J^T f0
    fwd(0, cg_state, x, nullptr);
    PT_CHECK_LAUNCH();
    int rc = pt_launch_adj2_plain(f, samples, stride_n, (const float*)l.rmap, gpart, st);
    if (rc) return rc;
    // right-hand side b = -J^T f0 (:262-265), first direction, J p
    hipLaunchKernelGGL(k_acg_red, dim3(nred), dim3(256), 0, st, (const float*)gpart, (const float*)x, q, pqp, (const float*)(cg_state + 2 * CKK), f.KSPL, CKK, 1, lambda);
    PT_CHECK_LAUNCH();
    fwd(1, cg_state, q, S[0]);
    PT_CHECK_LAUNCH();
    rc = pt_launch_adj2_plain(f, samples, stride_n, (const float*)l.rmap, gpart, st);
    if (rc) return rc;
    int cur = 0;
    for (int ii = 0; ii < num_iter; ++ii) {
        hipLaunchKernelGGL(k_acg_red, dim3(nred), dim3(256), 0, st, (const float*)gpart, (const float*)S[cur], q, pqp, (const float*)(S[cur] + 2 * CKK), f.KSPL, CKK, 2, lambda);
        PT_CHECK_LAUNCH();
        if (ii == num_iter - 1) break;
        fwd(2, S[cur], q, S[cur ^ 1]);
        PT_CHECK_LAUNCH();
        cur ^= 1;
        rc = pt_launch_adj2_plain(f, samples, stride_n, (const float*)l.rmap, gpart, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_acg_final, dim3(1), dim3(1024), 0, st, (const float*)S[cur], (const float*)q, (const float*)pqp, x, cg_state, CKK, nred, fr);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" int pt_atom_cg_f32(float* x, const float* samples, long samples_stride_n, const float* y,
                              const float* sample_weights, float filter_reg, float act_min_val, int n, int C, int H,
                              int W, int K, int num_iter, int fletcher_reeves, float direction_forget_factor,
                              float* cg_state, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !samples || !y || !sample_weights || !cg_state || !ws) return PT_ERR_NULL;
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || num_iter < 0) return PT_ERR_SHAPE;
    if (K * K > 16) return PT_ERR_UNSUPPORTED;
    if (samples_stride_n < (long)C * H * W) return PT_ERR_SHAPE;
    if (num_iter == 0) return PT_OK;                                    // optimization.py:230-231
    hipStream_t st = (hipStream_t)stream;
    if (ws_bytes < pt_atom_cg_ws_bytes(n, C, H, W, K) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    {
        PtFast f = pt_fast_plan(n, C, H, W, K, K, H, W);
        if (acg_fast_ok(f) && pt_fast_usable(f, samples, samples_stride_n, x) && ((uintptr_t)cg_state % 16) == 0)
            return acg_solve_fast(f, x, samples, samples_stride_n, y, sample_weights, filter_reg, act_min_val, num_iter, fletcher_reeves,
                                  direction_forget_factor, cg_state, (float*)ws, st);
    }
    PtPlan p = pt_make_plan(n, C, H, W, K, K, H, W);                    // conv2d(mode='same'): OH=H, OW=W
    CgCarve cv = cg_carve(p);
    float* base = (float*)ws;
    CgArgs a;
    a.n = n; a.C = C; a.H = H; a.W = W; a.K = K; a.HW = H * W; a.CKK = C * K * K; a.KS = p.KS; a.KSPL = p.KSPL;
    a.num_iter = num_iter; a.fletcher_reeves = fletcher_reeves;
    a.lambda = filter_reg; a.act_min = act_min_val; a.forget = direction_forget_factor;
    a.x = x; a.y = y; a.sw = sample_weights;
    a.d = base + cv.d; a.spart = base + cv.spart; a.R = base + cv.R; a.gpart = base + cv.gpart;
    a.r = base + cv.r; a.delta = base + cv.delta; a.scal = base + cv.scal;
    a.p = cg_state; a.r_prev = cg_state + a.CKK; a.st = cg_state + 2 * (size_t)a.CKK;
    const size_t lds = (size_t)a.HW * sizeof(float);

    int rc = pt_launch_corr(p, samples, samples_stride_n, x, a.spart, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_atom_pw, dim3(n), dim3(512), lds, st, a, 0);
    PT_CHECK_LAUNCH();
    rc = pt_launch_adj(p, samples, samples_stride_n, a.R, a.gpart, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_atom_vec, dim3(1), dim3(1024), 0, st, a, 0, 0);
    PT_CHECK_LAUNCH();
    for (int ii = 0; ii < num_iter; ++ii) {
        rc = pt_launch_corr(p, samples, samples_stride_n, a.p, a.spart, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_atom_pw, dim3(n), dim3(512), lds, st, a, 1);
        PT_CHECK_LAUNCH();
        rc = pt_launch_adj(p, samples, samples_stride_n, a.R, a.gpart, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_atom_vec, dim3(1), dim3(1024), 0, st, a, 1, ii);
        PT_CHECK_LAUNCH();
    }
    return PT_OK;
}
