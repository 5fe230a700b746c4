// Micro-benchmark (not part of the product): how fast does MI355X deliver the DiMP-50 sample memory (n x 512 x 18 x 18 fp32) to the
// CUs in the ACCESS PATTERNS of the two solver passes, with nothing else in the kernel?  Device-side duration = last wave's end
// stamp - first wave's entry stamp (100 MHz wall clock), so the host launch floor is not in the number.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/stream_probe.hip -o experiments/stream_probe && ./experiments/stream_probe [n]
//   pattern A (k_adj2):  256 workgroups x 512 threads; lane (kq, j) of wave w reads channel 16*cb + j, positions 16*g + 4*kq .. +3 of
//                        group g = ks*gper + w*U + u, u = 0..15: one 16-byte load per group, INF of them in flight
//   pattern C (k_corr2): 8n workgroups x 640 threads; lane (kq, j) of wave (t, h) reads channel 64*x + 4*(8h + k) + kq, positions
//                        64*t + 4*j .. +3, k = 0..7
//   pattern L (linear):  512 x 256 threads, each XCD streams its channel range of every sample with consecutive 16-byte vectors
// cold = consecutive launches alternate between two copies of the memory (nothing can survive in an L2 between launches).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define HW 324
#define C 512

__device__ __forceinline__ void stamp(unsigned long long* st, int k) {
    if ((threadIdx.x & 63) == 0) st[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + k] = (unsigned long long)wall_clock64();
}

template <int INF>
__global__ __launch_bounds__(512) void k_adj_pattern(const float* __restrict__ feat, int n, int gper, int U, unsigned long long* st, float* out) {
    stamp(st, 0);
    const int b = blockIdx.x, x = b & 7, rr = b >> 3, cb = 4 * x + (rr & 3), ks = rr >> 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4, j = lane & 15;
    const int total = n * HW;
    const int c = cb * 16 + j;
    f32x4 v[16];
    f32x4 acc = {0, 0, 0, 0};
    auto addr = [&](int u) {
        const int g = ks * gper + wave * U + u;
        int P0 = 16 * g + 4 * kq;
        if (u >= U || P0 >= total) P0 = 4 * kq;
        const int i = P0 / HW, p0 = P0 - i * HW;
        return (const f32x4*)(feat + ((size_t)i * C + c) * HW + p0);
    };
#pragma unroll
    for (int u = 0; u < INF; ++u) v[u] = *addr(u);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (u + INF < 16) v[u + INF] = *addr(u + INF);
        acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
    stamp(st, 1);
}

template <int INF>
__global__ __launch_bounds__(640) void k_corr_pattern(const float* __restrict__ feat, int n, unsigned long long* st, float* out) {
    stamp(st, 0);
    const int b = blockIdx.x, x = b & 7, i = b >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4, j = lane & 15;
    const int h = wave >= 5 ? 1 : 0, t = wave - 5 * h;
    const int pos = 64 * t + 4 * j;
    const float* base = feat + ((size_t)i * C + 64 * x + 32 * h + kq) * HW + pos;
    f32x4 v[8];
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < INF; ++k) v[k] = *(const f32x4*)(base + (size_t)(4 * k) * HW);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k + INF < 8) v[k + INF] = *(const f32x4*)(base + (size_t)(4 * (k + INF)) * HW);
        acc += v[k];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
    stamp(st, 1);
}

// per XCD 64 workgroups; the XCD's slice of sample i = 64 channels x 324 = 5184 vectors, streamed 256 vectors at a time
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ feat, int n, unsigned long long* st, float* out) {
    stamp(st, 0);
    const int b = blockIdx.x, x = b & 7, idx = b >> 3, per = gridDim.x >> 3;
    const int vec_per = 64 * HW / 4;                      // 5184
    const long tot = (long)n * vec_per;
    const long chunk = (tot + per - 1) / per;
    const long lo = idx * chunk, hi = std::min(tot, lo + chunk);
    f32x4 acc = {0, 0, 0, 0};
    for (long e0 = lo; e0 < hi; e0 += 8 * 256) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            long e = e0 + u * 256 + threadIdx.x;
            if (e >= hi) e = lo;
            const long i = e / vec_per, r = e - i * vec_per;
            v[u] = *(const f32x4*)(feat + ((size_t)i * C + 64 * x) * HW + 4 * r);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
    stamp(st, 1);
}

__global__ void k_flush(f32x4* p, size_t nvec) {         // touch another 256 MB in between (optional)
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nvec; e += (size_t)gridDim.x * blockDim.x) p[e] += 1.0f;
}

template <typename F>
void run(const char* name, int wgs, int waves, double bytes, F launch) {
    unsigned long long* st;
    hipMalloc(&st, (size_t)wgs * 16 * 2 * 8);
    std::vector<unsigned long long> h((size_t)wgs * 16 * 2);
    for (int w = 0; w < 5; ++w) launch(w, st);
    hipDeviceSynchronize();
    double sum = 0, best = 1e9;
    const int R = 20;
    for (int r = 0; r < R; ++r) {
        hipMemset(st, 0, (size_t)wgs * 16 * 2 * 8);
        hipDeviceSynchronize();
        launch(r, st);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < wgs; ++b)
            for (int w = 0; w < waves; ++w) {
                lo = std::min(lo, h[((size_t)b * 16 + w) * 2]);
                hi = std::max(hi, h[((size_t)b * 16 + w) * 2 + 1]);
            }
        const double us = (hi - lo) * 0.01;
        sum += us; best = std::min(best, us);
    }
    // event period of back-to-back launches
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    for (int r = 0; r < 200; ++r) launch(r, st);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("%-58s device %6.2f us (best %5.2f) = %6.2f TB/s | back-to-back period %6.2f us\n", name, sum / R, best, bytes / (sum / R) * 1e-6,
           ms * 1e3 / 200);
    hipFree(st);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 50;
    const size_t nfl = (size_t)n * C * HW;
    float *A, *B, *out;
    hipMalloc(&A, nfl * 4 + 4096); hipMalloc(&B, nfl * 4 + 4096); hipMalloc(&out, 1 << 16);
    hipMemset(A, 0, nfl * 4 + 4096); hipMemset(B, 0, nfl * 4 + 4096);
    const double bytes = (double)nfl * 4;
    const int NG = (n * HW + 15) / 16, gper = (NG + 7) / 8, U = (gper + 7) / 8;
    printf("n = %d: %.2f MB per pass, adj pattern gper %d U %d\n", n, bytes * 1e-6, gper, U);
    for (int cold = 0; cold < 2; ++cold) {
        auto buf = [&](int r) { return (cold && (r & 1)) ? B : A; };
        const char* tag = cold ? "cold (alternating copies)" : "same buffer";
        printf("--- %s\n", tag);
        run("linear 512 x 256, 8 in flight", 512, 4, bytes, [&](int r, unsigned long long* st) { hipLaunchKernelGGL(k_linear, dim3(512), dim3(256), 0, 0, buf(r), n, st, out); });
        run("linear 1024 x 256, 8 in flight", 1024, 4, bytes, [&](int r, unsigned long long* st) { hipLaunchKernelGGL(k_linear, dim3(1024), dim3(256), 0, 0, buf(r), n, st, out); });
        run("corr pattern 8n x 640, 8 in flight", 8 * n, 10, bytes, [&](int r, unsigned long long* st) { hipLaunchKernelGGL(k_corr_pattern<8>, dim3(8 * n), dim3(640), 0, 0, buf(r), n, st, out); });
        run("corr pattern 8n x 640, 4 in flight", 8 * n, 10, bytes, [&](int r, unsigned long long* st) { hipLaunchKernelGGL(k_corr_pattern<4>, dim3(8 * n), dim3(640), 0, 0, buf(r), n, st, out); });
        run("corr pattern 8n x 640, 2 in flight", 8 * n, 10, bytes, [&](int r, unsigned long long* st) { hipLaunchKernelGGL(k_corr_pattern<2>, dim3(8 * n), dim3(640), 0, 0, buf(r), n, st, out); });
        run("adj pattern 256 x 512, 16 in flight", 256, 8, bytes, [&](int r, unsigned long long* st) { hipLaunchKernelGGL(k_adj_pattern<16>, dim3(256), dim3(512), 0, 0, buf(r), n, gper, U, st, out); });
        run("adj pattern 256 x 512, 8 in flight", 256, 8, bytes, [&](int r, unsigned long long* st) { hipLaunchKernelGGL(k_adj_pattern<8>, dim3(256), dim3(512), 0, 0, buf(r), n, gper, U, st, out); });
        run("adj pattern 256 x 512, 4 in flight", 256, 8, bytes, [&](int r, unsigned long long* st) { hipLaunchKernelGGL(k_adj_pattern<4>, dim3(256), dim3(512), 0, 0, buf(r), n, gper, U, st, out); });
    }
    return 0;
}
