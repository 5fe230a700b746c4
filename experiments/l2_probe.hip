// Micro-benchmark (not part of the product): how fast can repeated passes over the DiMP-50 sample memory
// (33.2 MB fp32) be served on MI355X, depending on whether the same XCD re-reads the same eighth of it
// (per-XCD L2 = 4 MiB, block b runs on XCD b % 8) and on the workgroup geometry?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/l2_probe.hip -o experiments/l2_probe && ./experiments/l2_probe
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// WG b: xcd = (b + rot) % 8 owns vectors [xcd*per_xcd, (xcd+1)*per_xcd); the WGs of one XCD split that range evenly.
// Each thread issues UN independent 16-byte loads per trip.
template <int UN>
__global__ void k_read_xcd(const f32x4* __restrict__ p, size_t per_xcd, int rot, int reverse, float* out) {
    const int b = blockIdx.x, xcd = (b + rot) & 7, idx = b >> 3, per = gridDim.x >> 3;
    const size_t chunk = (per_xcd + per - 1) / per;
    const size_t lo = (size_t)xcd * per_xcd + (size_t)idx * chunk;
    const size_t hi = min((size_t)(xcd + 1) * per_xcd, lo + chunk);
    f32x4 acc = {0, 0, 0, 0};
    const size_t span = (size_t)UN * blockDim.x;
    for (size_t base = lo; base < hi; base += span) {
        f32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            if (reverse) i = hi - 1 - (i - lo);
            v[u] = p[(i < hi && i >= lo) ? i : lo];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

// strided ownership, as k_corr2/k_adj2 use it: XCD x owns the x-th eighth of EVERY sample (n samples of `svec` vectors)
__global__ void k_read_xcd_strided(const f32x4* __restrict__ p, int n, size_t svec, int rot, float* out) {
    const int b = blockIdx.x, xcd = (b + rot) & 7, idx = b >> 3, per = gridDim.x >> 3;
    const size_t eighth = svec / 8;
    f32x4 acc = {0, 0, 0, 0};
    for (int i = idx; i < n; i += per) {
        const f32x4* q = p + (size_t)i * svec + (size_t)xcd * eighth;
        for (size_t e = threadIdx.x; e < eighth; e += 8 * blockDim.x) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { size_t k = e + (size_t)u * blockDim.x; v[u] = q[k < eighth ? k : 0]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

__global__ void k_empty(float* out) { if (threadIdx.x == 9999) out[0] = 1.f; }

// chain of dependent tiny kernels: load 50 floats -> wave reduce -> store
__global__ void k_tiny(const float* in, float* out) {
    float a = in[threadIdx.x & 63];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (threadIdx.x == 0) out[blockIdx.x] = a;
}

template <typename F>
float time_it(const char* name, F f, int reps = 200, double bytes = 0) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f(i);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f(i);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    float us = ms * 1e3f / reps;
    if (bytes > 0) printf("%-64s %8.2f us   %7.1f GB/s\n", name, us, bytes / us * 1e-3);
    else printf("%-64s %8.2f us\n", name, us);
    return us;
}

int main() {
    const size_t full = (size_t)50 * 512 * 324;          // floats
    float *feat, *out;
    hipMalloc(&feat, full * 4 * 2);
    hipMalloc(&out, 1 << 20);
    hipMemset(feat, 0, full * 4 * 2);
    std::vector<float> h(full * 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(feat, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    time_it("empty kernel (1 WG)", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, out); });
    time_it("empty kernel (256 WG x 512)", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, 0, out); });
    time_it("tiny dependent kernel (50 WG x 64)", [&](int) { hipLaunchKernelGGL(k_tiny, dim3(50), dim3(64), 0, 0, out, out + 4096); });
    time_it("tiny dependent kernel (1 WG x 64)", [&](int) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, 0, out, out + 4096); });

    {
        const size_t svec = (size_t)512 * 324 / 4;
        const double B = 50.0 * svec * 16.0;
        time_it("STRIDED ownership aligned, 400 WGs x 512", [&](int) { hipLaunchKernelGGL(k_read_xcd_strided, dim3(400), dim3(512), 0, 0, (const f32x4*)feat, 50, svec, 0, out); }, 200, B);
        time_it("STRIDED ownership rotating, 400 WGs x 512", [&](int k) { hipLaunchKernelGGL(k_read_xcd_strided, dim3(400), dim3(512), 0, 0, (const f32x4*)feat, 50, svec, k % 8, out); }, 200, B);
        const size_t nvec = full / 4, per_xcd = nvec / 8;
        time_it("CONTIG  ownership aligned, 512 WGs x 256", [&](int) { hipLaunchKernelGGL(k_read_xcd<8>, dim3(512), dim3(256), 0, 0, (const f32x4*)feat, per_xcd, 0, 0, out); }, 200, B);
        time_it("CONTIG  ownership rotating, 512 WGs x 256", [&](int k) { hipLaunchKernelGGL(k_read_xcd<8>, dim3(512), dim3(256), 0, 0, (const f32x4*)feat, per_xcd, k % 8, 0, out); }, 200, B);
        if (getenv("L2_PROBE_SHORT")) return 0;
    }
    for (double frac : {1.0, 0.9, 0.75, 0.5, 2.0}) {
        const size_t nvec = (size_t)(full * frac) / 4 / 8 * 8;
        const size_t per_xcd = nvec / 8;
        const double B = nvec * 16.0;
        printf("--- working set %.1f MB (%.2f MB per XCD) ---\n", B * 1e-6, B * 1e-6 / 8);
        for (int wgs : {256, 512, 1024, 2048}) {
            for (int threads : {256, 512}) {
                char nm[128];
                snprintf(nm, 128, "aligned   WGs=%d x %d, 8 loads/thr", wgs, threads);
                time_it(nm, [&](int) { hipLaunchKernelGGL(k_read_xcd<8>, dim3(wgs), dim3(threads), 0, 0, (const f32x4*)feat, per_xcd, 0, 0, out); }, 200, B);
            }
        }
        time_it("aligned   WGs=512 x 256, 16 loads/thr", [&](int) { hipLaunchKernelGGL(k_read_xcd<16>, dim3(512), dim3(256), 0, 0, (const f32x4*)feat, per_xcd, 0, 0, out); }, 200, B);
        time_it("aligned   WGs=512 x 256, 4 loads/thr", [&](int) { hipLaunchKernelGGL(k_read_xcd<4>, dim3(512), dim3(256), 0, 0, (const f32x4*)feat, per_xcd, 0, 0, out); }, 200, B);
        time_it("aligned, alternating direction WGs=512 x 256, 8 loads/thr", [&](int k) { hipLaunchKernelGGL(k_read_xcd<8>, dim3(512), dim3(256), 0, 0, (const f32x4*)feat, per_xcd, 0, k & 1, out); }, 200, B);
        time_it("rotating XCD ownership WGs=512 x 256, 8 loads/thr", [&](int k) { hipLaunchKernelGGL(k_read_xcd<8>, dim3(512), dim3(256), 0, 0, (const f32x4*)feat, per_xcd, k % 8, 0, out); }, 200, B);
        time_it("rotating XCD ownership WGs=1024 x 256, 8 loads/thr", [&](int k) { hipLaunchKernelGGL(k_read_xcd<8>, dim3(1024), dim3(256), 0, 0, (const f32x4*)feat, per_xcd, k % 8, 0, out); }, 200, B);
    }
    return 0;
}
