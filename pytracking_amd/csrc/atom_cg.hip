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

extern "C" size_t pt_atom_cg_ws_bytes(int n, int C, int H, int W, int K) {
    if (n <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
    PtPlan p = pt_make_plan(n, C, H, W, K, K, H, W);
    return cg_carve(p).total * sizeof(float);
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
    PtPlan p = pt_make_plan(n, C, H, W, K, K, H, W);                    // conv2d(mode='same'): OH=H, OW=W
    CgCarve cv = cg_carve(p);
    if (ws_bytes < cv.total * sizeof(float) || ((uintptr_t)ws % 256) != 0) return PT_ERR_WORKSPACE;
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
