// ATOM first-frame joint optimisation of (filter, projection matrix): `GaussNewtonCG.run`
// (pytracking/libs/optimization.py:328-421, run_CG :72-163) on `FactorizedConvProblem`
// (pytracking/tracker/atom/optim.py:6-68) with identity projection activation and MLU response activation
// (pytracking/tracker/atom/atom.py:443-470).
//
//   residuals  f(x, P) = [ sqrt(sw_i) * (MLU(conv_same(c_i, x)) - y_i) ,  sqrt(lf) * x ,  sqrt(lP) * P ],   c = conv1x1(S, P)
//
// The reference obtains J^T J p with two autograd passes per CG iteration (optimization.py:410-412).  Here the Jacobian of
// one linearisation (d = sqrt(sw) * MLU'(s), c) is explicit:
//     J (px, pP)  = [ d .* (conv_same(c, px) + conv_same(conv1x1(S, pP), x)) ,  sqrt(lf) px ,  sqrt(lP) pP ]
//     J^T (u, ux, uP) = ( adj(c, d.*u) + sqrt(lf) ux ,   sum_i (dL/dc_i) S_i^T + sqrt(lP) uP ),   dL/dc = conv_same^T(d.*u, x)
// built from kernels that already exist: conv1x1 and its P-gradient are the multi-filter passes with K = 1
// (mf_kernels.hip, 16 projection rows per launch), conv_same / adj are the generic single-filter passes
// (filter_kernels.hip); new here are the input-gradient of conv_same, the pointwise stage and the two-block CG
// recurrences (joint inner product, diagonal preconditioner 1/[lf, lP], optim.py:48-68).  No host synchronisation.
#include <algorithm>
#include "common.h"
#include "pt_internal.h"
#include "rbuild.h"
#include "mfma_gemm.h"

struct GnArgs {
    int n, M, Kc, H, W, K, HW, KK, NF /* Kc*KK */, NP /* Kc*M */, NV;
    int KS, KSPL, NSG, NGRP;
    float lf, lP, act_min;
    int fletcher_reeves;
    float *f, *P;                       // variables, updated in place (optimization.py:403-404)
    const float *y, *sw;
    float *d, *v;                       // (n,HW) linearisation scale, output-side map
    float *sp1, *sp2;                   // correlation partials (KS, n, HW) each
    float *R;                           // im2col of v for the filter adjoint
    float *gpf;                         // (KSPL, NF) filter-adjoint partials
    float *gpP;                         // (NGRP, NSG, 16*M) projection-adjoint partials
    float *c, *dc, *gc;                 // (n,Kc,HW) compressed samples, their direction, their gradient
    float *r, *p, *q, *delta, *rprev;   // (NV) CG vectors: [filter block | projection block]
    float *scal;                        // [0] rho, [1] stop, [2] has_p
    float *pqpart;                      // (ceil(NV/256)) per-workgroup partial sums of <p, q> (k_gn_gather)
};

__device__ __forceinline__ float gn_mlu(float x, float mn) {
    const float yv = x >= 0.f ? x : x / mn;
    return yv > 0.f ? yv : mn * (expf(yv) - 1.f);
}
__device__ __forceinline__ float gn_mlu_d(float x, float mn) { return x >= 0.f ? 1.f : expf(x / mn); }

// mode 0: linearisation point: s = conv_same(c, x) -> d, v = d * sqrt(sw)*(MLU(s) - y)     (input of J^T f0)
// mode 1: J p: t = conv_same(c, px) + conv_same(dc, x) -> v = d * (d * t)                  (input of J^T (J p))
__global__ __launch_bounds__(512) void k_gn_pw(GnArgs a, int mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int i = blockIdx.x;
    const long base = (long)i * a.HW;
    const float sq = sqrtf(a.sw[i]);
    // channel-slice partials in fixed-size batches with clamped indices: a run-time trip count makes the compiler's unrolled body
    // start at 8 iterations and hands fewer (KS = 4 at ATOM's sizes) to a remainder loop of one dependent load per round trip
    // (round 4: 8 round trips per element here, 7.4 us per launch)
    const long sstride = (long)a.n * a.HW;
    auto slice_sum = [&](const float* __restrict__ sp, int o) {
        float t = 0.f;
        for (int k0 = 0; k0 < a.KS; k0 += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = sp[(long)min(k0 + k, a.KS - 1) * sstride + (long)i * a.HW + o];
#pragma unroll
            for (int k = 0; k < 8; ++k) t += k0 + k < a.KS ? v[k] : 0.f;               // same order as the plain loop
        }
        return t;
    };
    for (int o = threadIdx.x; o < a.HW; o += blockDim.x) {
        float val;
        if (mode == 0) {
            const float yv = a.y[base + o];
            const float t = slice_sum(a.sp1, o);
            const float dv = sq * gn_mlu_d(t, a.act_min);
            a.d[base + o] = dv;
            val = dv * (sq * (gn_mlu(t, a.act_min) - yv));
        } else {
            const float dv = a.d[base + o];
            const float t = slice_sum(a.sp1, o), t2 = slice_sum(a.sp2, o);
            val = dv * (dv * (t + t2));
        }
        a.v[base + o] = val;
        lds[o] = val;
    }
    __syncthreads();
    pt_build_R_sample(lds, a.R, i, a.n, a.H, a.W, a.K, a.K, a.H, a.W);
}

__device__ float gn_dot2(const float* u, const float* v, int n, float* scratch) {
    float acc = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) acc += u[e] * v[e];
    return block_sum(acc, scratch);
}

// element e of J^T(.) assembled from the pass partials (+ the regularisation part `reg * vec[e]`)
__device__ __forceinline__ float gn_gather(const GnArgs& a, int e, const float* vec) {
    // partial slabs in fixed-size batches of 8 with clamped indices (all loads of a batch in flight, fixed summation order); with a
    // run-time trip count the unrolled loop's remainder -- 6 of the 30 per-sample slabs -- was one dependent load per round trip
    float s = 0.f;
    const float rv = vec[e];
    if (e < a.NF) {
        for (int k0 = 0; k0 < a.KSPL; k0 += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = a.gpf[(long)min(k0 + k, a.KSPL - 1) * a.NF + e];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += k0 + k < a.KSPL ? v[k] : 0.f;
        }
        return s + a.lf * rv;
    }
    const int ep = e - a.NF, row = ep / a.M, m = ep - row * a.M, grp = row >> 4, rl = row & 15;
    const int Fg = min(16, a.Kc - 16 * grp);
    const float* gp = a.gpP + (long)grp * a.NSG * 16 * a.M + (long)rl * a.M + m;
    const long kst = (long)Fg * a.M;
    for (int k0 = 0; k0 < a.NSG; k0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = gp[(long)min(k0 + k, a.NSG - 1) * kst];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += k0 + k < a.NSG ? v[k] : 0.f;
    }
    return s + a.lP * rv;
}

// search direction from the residual r with the diagonal preconditioner (optimization.py:97-125, optim.py:67-68)
#define GN_NE 20   // elements per thread of the register-resident CG step (NV <= 20480)
__device__ void gn_direction(const GnArgs& a, float* scratch) {
    const float rho1 = a.scal[0];
    float acc = 0.f, acc2 = 0.f;
    const bool has_p = a.scal[2] != 0.f;
    if (a.NV <= 1024 * GN_NE && blockDim.x == 1024) {
        // on registers (one round of loads, the two sums, one round of stores; same summation order as the sweeps below, which
        // paid one memory round trip per 1024 elements and array -- the first direction of every Gauss-Newton iteration, round 4)
        const bool pr = has_p && !a.fletcher_reeves;
        float re[GN_NE], pe[GN_NE], rp[GN_NE];
#pragma unroll
        for (int k = 0; k < GN_NE; ++k) {
            const int ec = min((int)threadIdx.x + 1024 * k, a.NV - 1);
            re[k] = a.r[ec];
            pe[k] = has_p ? a.p[ec] : 0.f;
            rp[k] = pr ? a.rprev[ec] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < GN_NE; ++k) {
            const int e = threadIdx.x + 1024 * k;
            if (e < a.NV) {
                const float z = re[k] / (e < a.NF ? a.lf : a.lP);
                acc += re[k] * z;
                if (pr) acc2 += rp[k] * z;
            }
        }
        const float rho = block_sum(acc, scratch);
        const float rho2 = block_sum(acc2, scratch);
        __syncthreads();
        if (rho == 0.f) {                                               // :108-113
            if (threadIdx.x == 0) { a.scal[0] = rho; a.scal[1] = 1.f; }
            return;
        }
        float beta = 0.f;
        if (has_p) beta = fmaxf(a.fletcher_reeves ? rho / rho1 : (rho - rho2) / rho1, 0.f);   // :118-124
#pragma unroll
        for (int k = 0; k < GN_NE; ++k) {
            const int e = threadIdx.x + 1024 * k;
            if (e < a.NV) {
                const float z = re[k] / (e < a.NF ? a.lf : a.lP);
                a.p[e] = has_p ? z + beta * pe[k] : z;
            }
        }
        if (threadIdx.x == 0) { a.scal[0] = rho; a.scal[2] = 1.f; }
        return;
    }
    for (int e = threadIdx.x; e < a.NV; e += blockDim.x) {
        const float z = a.r[e] / (e < a.NF ? a.lf : a.lP);
        acc += a.r[e] * z;
        if (has_p && !a.fletcher_reeves) acc2 += a.rprev[e] * z;
    }
    const float rho = block_sum(acc, scratch);
    const float rho2 = block_sum(acc2, scratch);
    __syncthreads();
    if (rho == 0.f) {                                                   // :108-113
        if (threadIdx.x == 0) { a.scal[0] = rho; a.scal[1] = 1.f; }
        return;
    }
    float beta = 0.f;
    if (has_p) beta = fmaxf(a.fletcher_reeves ? rho / rho1 : (rho - rho2) / rho1, 0.f);   // :118-124
    for (int e = threadIdx.x; e < a.NV; e += blockDim.x) {
        const float z = a.r[e] / (e < a.NF ? a.lf : a.lP);
        a.p[e] = has_p ? z + beta * a.p[e] : z;
    }
    if (threadIdx.x == 0) { a.scal[0] = rho; a.scal[2] = 1.f; }
}

// Assembling J^T(.) from the pass partials is the wide part of a CG step (NV = Kc*K*K + Kc*M elements, each the sum of up
// to NSG partial slabs): one workgroup per 256 elements.  As a single workgroup (round 1) this was 146 us per step, 86 %
// of the first-frame solve.
//   phase 0: r = -(J^T f0 + reg * x), delta = 0        phase 1: q = J^T J p (+ reg * p), per-workgroup partial of <p, q>
__global__ __launch_bounds__(256) void k_gn_gather(GnArgs a, int phase) {
    __shared__ float scratch[16];
    const int e = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (e < a.NV) {
        if (phase == 0) {
            a.r[e] = -gn_gather(a, e, e < a.NF ? a.f : a.P - a.NF);
            a.delta[e] = 0.f;
        } else {
            const float qv = gn_gather(a, e, a.p);
            a.q[e] = qv;
            acc = a.p[e] * qv;
        }
    }
    if (phase == 1) {
        const float tot = block_sum(acc, scratch);
        if (threadIdx.x == 0) a.pqpart[blockIdx.x] = tot;
    }
}

// The recurrences of a CG step on the assembled vectors (one workgroup; fixed summation order):
// phase 0: state reset (:82-83), first direction
// phase 1: alpha, delta, residual (:127-146); then the next direction, or x += delta after the last one
__global__ __launch_bounds__(1024) void k_gn_vec(GnArgs a, int phase, int ii, int num_iter) {
    __shared__ float scratch[16];
    if (phase == 0) {
        if (threadIdx.x == 0) { a.scal[0] = 1.f; a.scal[1] = 0.f; a.scal[2] = 0.f; }
        __syncthreads();
        gn_direction(a, scratch);
        return;
    }
    if (a.scal[1] != 0.f) {                                             // CG stopped early: apply what we have once
        if (ii == num_iter - 1)
            for (int e = threadIdx.x; e < a.NV; e += blockDim.x) (e < a.NF ? a.f[e] : a.P[e - a.NF]) += a.delta[e];
        return;
    }
    const int nparts = (a.NV + 255) / 256;
    float acc = 0.f;
    for (int k = threadIdx.x; k < nparts; k += blockDim.x) acc += a.pqpart[k];
    const bool more = ii < num_iter - 1;
    if (more && a.NV <= 1024 * GN_NE) {
        // The whole step on registers: every thread owns elements tid + 1024 k.  One round of loads (all independent, in
        // flight together), the recurrences, two block sums, one round of stores -- instead of three dependent sweeps
        // through memory by a single workgroup (21 -> 9 us per CG step at NV = 17408).  Same summation order as below.
        float re[GN_NE], pe[GN_NE], qe[GN_NE], de[GN_NE], rp[GN_NE];
        const float rho1 = a.scal[0];
        const bool has_p = a.scal[2] != 0.f, pr = has_p && !a.fletcher_reeves;
#pragma unroll
        for (int k = 0; k < GN_NE; ++k) {
            const int e = threadIdx.x + 1024 * k;
            const bool ok = e < a.NV;
            re[k] = ok ? a.r[e] : 0.f;
            pe[k] = ok ? a.p[e] : 0.f;
            qe[k] = ok ? a.q[e] : 0.f;
            de[k] = ok ? a.delta[e] : 0.f;
        }
        const float pq = block_sum(acc, scratch);
        const float alpha = rho1 / pq;                                  // :131
        float acc1 = 0.f, acc2 = 0.f;
#pragma unroll
        for (int k = 0; k < GN_NE; ++k) {
            const int e = threadIdx.x + 1024 * k;
            rp[k] = re[k];
            de[k] += alpha * pe[k];
            re[k] -= alpha * qe[k];
            const float z = re[k] / (e < a.NF ? a.lf : a.lP);
            acc1 += re[k] * z;
            if (pr) acc2 += rp[k] * z;
        }
        const float rho = block_sum(acc1, scratch);
        const float rho2 = block_sum(acc2, scratch);
        float beta = 0.f;
        if (has_p) beta = fmaxf(a.fletcher_reeves ? rho / rho1 : (rho - rho2) / rho1, 0.f);   // :118-124
#pragma unroll
        for (int k = 0; k < GN_NE; ++k) {
            const int e = threadIdx.x + 1024 * k;
            if (e < a.NV) {
                if (!a.fletcher_reeves) a.rprev[e] = rp[k];
                a.delta[e] = de[k];
                a.r[e] = re[k];
                if (rho != 0.f) {
                    const float z = re[k] / (e < a.NF ? a.lf : a.lP);
                    a.p[e] = has_p ? z + beta * pe[k] : z;
                }
            }
        }
        if (threadIdx.x == 0) {
            a.scal[0] = rho;
            if (rho == 0.f) a.scal[1] = 1.f;                            // :108-113
            else a.scal[2] = 1.f;
        }
        return;
    }
    const float pq = block_sum(acc, scratch);
    const float alpha = a.scal[0] / pq;                                 // :131
    if (!more && a.NV <= 1024 * GN_NE) {
        // last CG step of a Gauss-Newton iteration, on registers like the others: delta += alpha p, then x += delta (:143-146,
        // :403-404).  As two strided sweeps with a run-time bound this was 17 dependent round trips per array (round 4)
        float re[GN_NE], pe[GN_NE], de[GN_NE], xe[GN_NE];
#pragma unroll
        for (int k = 0; k < GN_NE; ++k) {
            const int e = threadIdx.x + 1024 * k, ec = min(e, a.NV - 1);
            re[k] = a.r[ec];
            pe[k] = a.p[ec];
            de[k] = a.delta[ec];
            xe[k] = ec < a.NF ? a.f[ec] : a.P[ec - a.NF];
        }
#pragma unroll
        for (int k = 0; k < GN_NE; ++k) {
            const int e = threadIdx.x + 1024 * k;
            if (e < a.NV) {
                if (!a.fletcher_reeves) a.rprev[e] = re[k];
                const float dn = de[k] + alpha * pe[k];
                a.delta[e] = dn;
                (e < a.NF ? a.f[e] : a.P[e - a.NF]) = xe[k] + dn;
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < a.NV; e += blockDim.x) {
        const float re = a.r[e];
        if (!a.fletcher_reeves) a.rprev[e] = re;
        a.delta[e] += alpha * a.p[e];
        if (more) a.r[e] = re - alpha * a.q[e];
    }
    __syncthreads();
    if (more) {
        gn_direction(a, scratch);                                       // may raise the stop flag (rho == 0)
    } else {
        for (int e = threadIdx.x; e < a.NV; e += blockDim.x) (e < a.NF ? a.f[e] : a.P[e - a.NF]) += a.delta[e];
    }
}

// ---------------------------------------------------------------------------------------------------
struct GnCarve { size_t d, v, sp1, sp2, R, gpf, gpP, c, dc, gc, wT, r, p, q, delta, rprev, scal, pqpart, total; };

static GnCarve gn_carve(const PtPlan& pl, int n, int M, int Kc, int H, int W, int K) {
    GnCarve c;
    size_t off = 0;
    auto take = [&](size_t nfl) { size_t o = off; off += pt_align_floats(nfl); return o; };
    const size_t HW = (size_t)H * W, NV = (size_t)Kc * K * K + (size_t)Kc * M;
    const int ngrp = (Kc + 15) / 16;
    c.d = take(n * HW); c.v = take(n * HW);
    c.sp1 = take(pt_spart_floats(pl)); c.sp2 = take(pt_spart_floats(pl));
    c.R = take(pt_R_floats(pl));
    c.gpf = take(pt_gpart_floats(pl));
    c.gpP = take((size_t)ngrp * pt_mf_gpart_floats(n, 16, M, H, W, 1));
    c.c = take(n * Kc * HW); c.dc = take(n * Kc * HW); c.gc = take(n * Kc * HW);
    c.wT = take((size_t)ngrp * pt_mf_wt_floats(M, 1));
    c.r = take(NV); c.p = take(NV); c.q = take(NV); c.delta = take(NV); c.rprev = take(NV);
    c.scal = take(64);
    c.pqpart = take((NV + 255) / 256);
    c.total = off;
    return c;
}

static int gn_check(int n, int M, int Kc, int H, int W, int K) {
    if (n <= 0 || M <= 0 || Kc <= 0 || H <= 0 || W <= 0 || K <= 0) return PT_ERR_SHAPE;
    if (K * K > 16) return PT_ERR_UNSUPPORTED;
    if (pt_mf_groups(n, std::min(Kc, 16), M, H, W, 1) == 0) return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

extern "C" size_t pt_atom_gn_ws_bytes(int n, int M, int Kc, int H, int W, int K) {
    if (gn_check(n, M, Kc, H, W, K)) return 0;
    PtPlan pl = pt_make_plan(n, Kc, H, W, K, K, H, W);
    return gn_carve(pl, n, M, Kc, H, W, K).total * sizeof(float);
}

// conv1x1(S, rows of `proj`) -> out (n,Kc,H,W), 16 projection rows per launch
static int gn_project(const GnArgs& a, const float* samples, long stride_n, const float* proj, float* out, float* wT,
                      hipStream_t st) {
    if (pt_launch_mf_corr1_direct(samples, stride_n, proj, out, a.n, a.Kc, a.M, a.H, a.W, st, (long)a.Kc * a.HW) == PT_OK)
        return PT_OK;                                               // weights read in place, no transposition launch
    if (a.Kc % 16 == 0) {                                           // all banks of 16 projection rows in one launch each
        int rc = pt_launch_mf_wtrans(proj, wT, 16, a.M, 1, st, a.NGRP);
        if (rc) return rc;
        return pt_launch_mf_corr(samples, stride_n, wT, out, a.n, 16, a.M, a.H, a.W, 1, st, (long)a.Kc * a.HW, a.NGRP);
    }
    for (int g = 0; g < a.NGRP; ++g) {
        const int Fg = std::min(16, a.Kc - 16 * g);
        int rc = pt_launch_mf_wtrans(proj + (long)16 * g * a.M, wT, Fg, a.M, 1, st);
        if (rc) return rc;
        rc = pt_launch_mf_corr(samples, stride_n, wT, out + (long)16 * g * a.HW, a.n, Fg, a.M, a.H, a.W, 1, st,
                               (long)a.Kc * a.HW);
        if (rc) return rc;
    }
    return PT_OK;
}

// J^T of the data part for the map in a.v / a.R: filter-adjoint partials -> gpf, projection-adjoint partials -> gpP
static int gn_jt(const GnArgs& a, const PtPlan& pl, const float* samples, long stride_n, hipStream_t st) {
    // the filter adjoint and dL/dc = conv_same^T(v, x) (filter_kernels.hip: input_grad_jobs) read the same map and do not depend on
    // each other: ONE launch (round 4; they were two dependent ones of 8.1 + 9.3 us)
    const PtInputGrad ig = {a.v, a.f, a.gc, a.n, a.Kc, a.H, a.W, a.K};
    int rc = pt_launch_adj(pl, a.c, (long)a.Kc * a.HW, a.R, a.gpf, st, &ig);
    if (rc) return rc;
    if (a.Kc % 16 == 0 && a.NSG == a.n && a.HW % 4 == 0 && stride_n % 4 == 0 && ((uintptr_t)samples % 16) == 0) {
        // one partial per sample: a batch of n small NT GEMMs  gpP[., i][f][m] = sum_pos gc[i][f][pos] * S[i][m][pos]
        // (K = H*W contiguous in both operands) instead of the banded LDS kernel, whose fixed cost per workgroup
        // dominates on an 18x18 map (50 us vs 6)
        GemmArgs ga = gemm_args(a.gc, a.HW, a.Kc, samples, a.Kc, a.M, a.HW, nullptr, a.gpP, a.M);
        ga.batch = a.n; ga.a_zstride = (long)a.Kc * a.HW; ga.w_zstride = stride_n;
        ga.c_seg = 16; ga.c_segstride = (long)a.NSG * 16; ga.c_zstride = (long)16 * a.M;
        return launch_gemm(ga, st);
    }
    if (a.Kc % 16 == 0)
        return pt_launch_mf_adj(samples, stride_n, a.gc, a.gpP, a.n, 16, a.M, a.H, a.W, 1, st, (long)a.Kc * a.HW, a.NGRP);
    for (int g = 0; g < a.NGRP; ++g) {
        const int Fg = std::min(16, a.Kc - 16 * g);
        rc = pt_launch_mf_adj(samples, stride_n, a.gc + (long)16 * g * a.HW, a.gpP + (long)g * a.NSG * 16 * a.M, a.n, Fg,
                              a.M, a.H, a.W, 1, st, (long)a.Kc * a.HW);
        if (rc) return rc;
    }
    return PT_OK;
}

extern "C" int pt_atom_gn_f32(float* filter, float* proj, const float* samples, long samples_stride_n, const float* y,
                              const float* sample_weights, float filter_reg, float projection_reg, float act_min_val,
                              int n, int M, int Kc, int H, int W, int K, const int* cg_iters, int num_gn,
                              int fletcher_reeves, void* ws, size_t ws_bytes, void* stream) {
    if (!filter || !proj || !samples || !y || !sample_weights || !ws || (num_gn > 0 && !cg_iters)) return PT_ERR_NULL;
    int rc = gn_check(n, M, Kc, H, W, K);
    if (rc) return rc;
    if (num_gn < 0 || samples_stride_n < (long)M * H * W) return PT_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    PtPlan pl = pt_make_plan(n, Kc, H, W, K, K, H, W);                  // conv2d(mode='same'): OH = H, OW = W
    GnCarve cv = gn_carve(pl, n, M, Kc, H, W, K);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
    float* base = (float*)ws;
    GnArgs a;
    a.n = n; a.M = M; a.Kc = Kc; a.H = H; a.W = W; a.K = K; a.HW = H * W; a.KK = K * K;
    a.NF = Kc * K * K; a.NP = Kc * M; a.NV = a.NF + a.NP;
    a.KS = pl.KS; a.KSPL = pl.KSPL; a.NGRP = (Kc + 15) / 16; a.NSG = pt_mf_groups(n, std::min(Kc, 16), M, H, W, 1);
    a.lf = filter_reg; a.lP = projection_reg; a.act_min = act_min_val; a.fletcher_reeves = fletcher_reeves;
    a.f = filter; a.P = proj; a.y = y; a.sw = sample_weights;
    a.d = base + cv.d; a.v = base + cv.v; a.sp1 = base + cv.sp1; a.sp2 = base + cv.sp2; a.R = base + cv.R;
    a.gpf = base + cv.gpf; a.gpP = base + cv.gpP; a.c = base + cv.c; a.dc = base + cv.dc; a.gc = base + cv.gc;
    a.r = base + cv.r; a.p = base + cv.p; a.q = base + cv.q; a.delta = base + cv.delta; a.rprev = base + cv.rprev;
    a.scal = base + cv.scal; a.pqpart = base + cv.pqpart;
    float* wT = base + cv.wT;
    const size_t pw_lds = (size_t)a.HW * sizeof(float);
    const long cs = (long)Kc * a.HW;
    for (int gi = 0; gi < num_gn; ++gi) {
        const int ncg = cg_iters[gi];
        if (ncg <= 0) continue;
        // ---- linearise at (x, P): c, s, d, right-hand side                                   optimization.py:373-392
        rc = gn_project(a, samples, samples_stride_n, proj, a.c, wT, st);
        if (rc) return rc;
        rc = pt_launch_corr(pl, a.c, cs, filter, a.sp1, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gn_pw, dim3(n), dim3(512), pw_lds, st, a, 0);
        PT_CHECK_LAUNCH();
        rc = gn_jt(a, pl, samples, samples_stride_n, st);
        if (rc) return rc;
        const unsigned ngw = (unsigned)((a.NV + 255) / 256);
        hipLaunchKernelGGL(k_gn_gather, dim3(ngw), dim3(256), 0, st, a, 0);
        PT_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_gn_vec, dim3(1), dim3(1024), 0, st, a, 0, 0, ncg);
        PT_CHECK_LAUNCH();
        // ---- conjugate gradient on J^T J delta = b                                          optimization.py:72-163
        for (int ii = 0; ii < ncg; ++ii) {
            rc = gn_project(a, samples, samples_stride_n, a.p + a.NF, a.dc, wT, st);           // conv1x1(S, pP)
            if (rc) return rc;
            // conv_same(c, px) -> sp1 and conv_same(dc, x) -> sp2 in one paired launch (same shape, two operand sets)
            const PtCorrFuse pair = {nullptr, 0, nullptr, 0.f, nullptr, nullptr, nullptr, a.dc, filter, a.sp2};
            rc = pt_launch_corr(pl, a.c, cs, a.p, a.sp1, st, &pair);
            if (rc == PT_ERR_UNSUPPORTED) {
                rc = pt_launch_corr(pl, a.c, cs, a.p, a.sp1, st);                              // conv_same(c, px)
                if (rc) return rc;
                rc = pt_launch_corr(pl, a.dc, cs, filter, a.sp2, st);                          // conv_same(dc, x)
            }
            if (rc) return rc;
            hipLaunchKernelGGL(k_gn_pw, dim3(n), dim3(512), pw_lds, st, a, 1);
            PT_CHECK_LAUNCH();
            rc = gn_jt(a, pl, samples, samples_stride_n, st);
            if (rc) return rc;
            hipLaunchKernelGGL(k_gn_gather, dim3(ngw), dim3(256), 0, st, a, 1);
            PT_CHECK_LAUNCH();
            hipLaunchKernelGGL(k_gn_vec, dim3(1), dim3(1024), 0, st, a, 1, ii, ncg);
            PT_CHECK_LAUNCH();
        }
    }
    return PT_OK;
}
