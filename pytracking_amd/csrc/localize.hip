// Target localisation on the device: the score-map part of
//   DiMP.localize_advanced / ToMP.localize_advanced   pytracking/tracker/dimp/dimp.py:238-303, tracker/tomp/tomp.py
//   dcf.max2d                                         pytracking/libs/dcf.py:156-164
// as one launch that leaves 8 floats for a single device-to-host copy.  The reference issues two max2d's, a clone, a
// masked fill and about ten .item()/.cpu() synchronisations per frame for the same numbers.
//
// Semantics kept bit-for-bit: max2d takes the maximum over rows first and then over columns, both returning the first
// index on ties, i.e. ties go to the smallest column, then the smallest row; the best scale is the first one on ties;
// the neighbourhood of the first peak is ZEROED (not excluded) before the second search; its bounds use Python's
// round() on doubles = round-half-to-even = rint.
#include "common.h"
#include "pt_internal.h"

namespace {

struct Peak { float v; int r, c; };

__device__ __forceinline__ bool better(const Peak& a, const Peak& b) {
    return a.v > b.v || (a.v == b.v && (a.c < b.c || (a.c == b.c && a.r < b.r)));
}

// block-wide arg-max with max2d's tie order; every thread gets the result
__device__ Peak block_peak(Peak p, Peak* sh) {
    const int tid = threadIdx.x;
    __syncthreads();
    sh[tid] = p;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (tid < s && better(sh[tid + s], sh[tid])) sh[tid] = sh[tid + s];
        __syncthreads();
    }
    return sh[0];
}

__device__ Peak scan_map(const float* m, int H, int W, int top, int bottom, int left, int right) {
    Peak best{-INFINITY, H, W};
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
        const int r = i / W, c = i - r * W;
        const bool masked = r >= top && r < bottom && c >= left && c < right;
        const Peak p{masked ? 0.f : m[i], r, c};
        if (better(p, best)) best = p;
    }
    return best;
}

struct LocArgs {
    const float *scores, *scores_hn;
    float* out;
    int S, H, W;
    float nh[8], nw[8];          // target neighbourhood (rows, cols) in score cells, per scale
};

__global__ __launch_bounds__(256) void k_localize(LocArgs a) {
    __shared__ Peak sh[256];
    Peak p1{-INFINITY, 0, 0};
    int s1 = 0;
    for (int s = 0; s < a.S; ++s) {
        const Peak p = block_peak(scan_map(a.scores + (long)s * a.H * a.W, a.H, a.W, 0, 0, 0, 0), sh);
        if (s == 0 || p.v > p1.v) { p1 = p; s1 = s; }               // torch.max over scales: first maximum
    }
    const int top = max((int)rint((double)p1.r - (double)a.nh[s1] / 2), 0);          // Python floats are doubles
    const int bottom = min((int)rint((double)p1.r + (double)a.nh[s1] / 2 + 1), a.H);
    const int left = max((int)rint((double)p1.c - (double)a.nw[s1] / 2), 0);
    const int right = min((int)rint((double)p1.c + (double)a.nw[s1] / 2 + 1), a.W);
    const Peak p2 = block_peak(scan_map(a.scores_hn + (long)s1 * a.H * a.W, a.H, a.W, top, bottom, left, right), sh);
    if (threadIdx.x == 0) {
        a.out[0] = p1.v; a.out[1] = (float)p1.r; a.out[2] = (float)p1.c; a.out[3] = (float)s1;
        a.out[4] = p2.v; a.out[5] = (float)p2.r; a.out[6] = (float)p2.c; a.out[7] = 0.f;
    }
}

__global__ __launch_bounds__(256) void k_max2d(const float* a, float* max_val, long long* argmax, int H, int W) {
    __shared__ Peak sh[256];
    const Peak p = block_peak(scan_map(a + (long)blockIdx.x * H * W, H, W, 0, 0, 0, 0), sh);
    if (threadIdx.x == 0) {
        max_val[blockIdx.x] = p.v;
        argmax[2 * (long)blockIdx.x] = p.r;
        argmax[2 * (long)blockIdx.x + 1] = p.c;
    }
}

}  // namespace

extern "C" int pt_max2d_f32(const float* a, float* max_val, long long* argmax, int n, int H, int W, void* stream) {
    if (!a || !max_val || !argmax) return PT_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return PT_ERR_SHAPE;
    hipLaunchKernelGGL(k_max2d, dim3(n), dim3(256), 0, (hipStream_t)stream, a, max_val, argmax, H, W);
    PT_CHECK_LAUNCH();
    return PT_OK;
}

extern "C" int pt_localize_f32(const float* scores, const float* scores_hn, const float* neigh_rows,
                               const float* neigh_cols, float* out8, int S, int H, int W, void* stream) {
    if (!scores || !neigh_rows || !neigh_cols || !out8) return PT_ERR_NULL;
    if (S <= 0 || H <= 0 || W <= 0) return PT_ERR_SHAPE;
    if (S > 8) return PT_ERR_UNSUPPORTED;
    LocArgs a{};
    a.scores = scores; a.scores_hn = scores_hn ? scores_hn : scores; a.out = out8; a.S = S; a.H = H; a.W = W;
    for (int s = 0; s < S; ++s) { a.nh[s] = neigh_rows[s]; a.nw[s] = neigh_cols[s]; }
    hipLaunchKernelGGL(k_localize, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    PT_CHECK_LAUNCH();
    return PT_OK;
}
