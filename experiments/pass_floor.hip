// Micro-benchmark (not part of the product): what does one pass over the 33 MB DiMP-50 sample memory cost on
// MI355X, for a plain streaming read and for the corr/adj kernels, back-to-back on one stream?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/pass_floor.hip -o experiments/pass_floor && ./experiments/pass_floor
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#define PT_EXPERIMENT 1
#ifndef PT_ABL
#define PT_ABL 0
#endif
#include "../pytracking_amd/csrc/filter_kernels.hip"

void pt_prof_begin(int, hipStream_t) {}
void pt_prof_end(int, hipStream_t) {}

__global__ void k_read(const f32x4* __restrict__ p, size_t nvec, float* out) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}
template <int UN>
__global__ void k_read_unrolled(const f32x4* __restrict__ p, size_t nvec, float* out) {
    // each thread issues UN independent 16-byte loads up front (like the pass kernels), one "item" per wave-slot
    f32x4 v[UN];
    const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int u = 0; u < UN; ++u) { size_t i = base + u * stride; v[u] = p[i < nvec ? i : 0]; }
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < UN; ++u) acc += v[u];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}
__global__ void k_empty(float* out) { if (threadIdx.x == 9999) out[0] = 1.f; }
__global__ void k_tiny(const float* in, float* out, int n) {      // load -> reduce -> store, 50 WGs
    __shared__ float sc[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += in[blockIdx.x * n + i];
    a = block_sum(a, sc);
    if (threadIdx.x == 0) out[blockIdx.x] = a;
}

template <typename F>
float time_it(const char* name, F f, int reps = 200, double bytes = 0) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    float us = ms * 1e3f / reps;
    if (bytes > 0) printf("%-44s %8.2f us   %7.1f GB/s\n", name, us, bytes / us * 1e-3);
    else printf("%-44s %8.2f us\n", name, us);
    return us;
}

int main() {
    const int n = 50, C = 512, H = 18, W = 18, K = 4, OH = 19, OW = 19;
    const size_t nfeat = (size_t)n * C * H * W;
    float *feat, *filt, *spart, *R, *gpart, *out, *w;
    hipMalloc(&feat, nfeat * 4); hipMalloc(&filt, C * 16 * 4); hipMalloc(&w, C * 16 * 4);
    PtPlan p = pt_make_plan(n, C, H, W, K, K, OH, OW);
    hipMalloc(&spart, pt_spart_floats(p) * 4); hipMalloc(&R, pt_R_floats(p) * 4); hipMalloc(&gpart, pt_gpart_floats(p) * 4);
    hipMalloc(&out, 1 << 20);
    std::vector<float> h(nfeat);
    for (size_t i = 0; i < nfeat; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(feat, h.data(), nfeat * 4, hipMemcpyHostToDevice);
    hipMemset(filt, 0, C * 16 * 4); hipMemset(w, 0, C * 16 * 4); hipMemset(R, 0, pt_R_floats(p) * 4); hipMemset(gpart, 0, pt_gpart_floats(p) * 4);
    printf("plan: KS=%d cper=%d corr_threads=%d lds=%zu | NG=%d KSPL=%d gper=%d\n", p.KS, p.cper, p.corr_threads, p.corr_lds, p.NG, p.KSPL, p.gper);
    const double B = nfeat * 4.0;
    time_it("empty kernel (1 WG)", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, out); });
    time_it("empty kernel (400 WG x 384)", [&] { hipLaunchKernelGGL(k_empty, dim3(400), dim3(384), 0, 0, out); });
    time_it("tiny load-reduce-store (50 WG x 512, 361 el)", [&] { hipLaunchKernelGGL(k_tiny, dim3(50), dim3(512), 0, 0, feat, out, 361); });
    for (int wgs : {256, 512, 1024, 2048, 4096})
        time_it(("stream read, grid-stride, WGs=" + std::to_string(wgs)).c_str(),
                [&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, 0, (const f32x4*)feat, nfeat / 4, out); }, 200, B);
    {
        const size_t nvec = nfeat / 4;
        auto run = [&](auto tag, int un, int threads) {
            const int wgs = (int)((nvec + (size_t)un * threads - 1) / ((size_t)un * threads));
            char nm[96]; snprintf(nm, 96, "stream read, %d loads/thread upfront, %dx%d", un, wgs, threads);
            time_it(nm, [&] { hipLaunchKernelGGL(k_read_unrolled<decltype(tag)::value>, dim3(wgs), dim3(threads), 0, 0, (const f32x4*)feat, nvec, out); }, 200, B);
        };
        run(std::integral_constant<int, 4>{}, 4, 256);
        run(std::integral_constant<int, 8>{}, 8, 256);
        run(std::integral_constant<int, 16>{}, 16, 256);
        run(std::integral_constant<int, 16>{}, 16, 384);
        run(std::integral_constant<int, 32>{}, 32, 256);
    }
    time_it("k_corr plain (n=50)", [&] { pt_launch_corr(p, feat, (long)C * H * W, filt, spart, 0); }, 200, B);
    PtCorrFuse fz = {gpart, p.KSPL, w, 0.01f, filt, out, nullptr};
    time_it("k_corr fused-g (n=50)", [&] { pt_launch_corr(p, feat, (long)C * H * W, nullptr, spart, 0, &fz); }, 200, B);
    time_it("k_adj (n=50)", [&] { pt_launch_adj(p, feat, (long)C * H * W, R, gpart, 0); }, 200, B);
    // alternating passes like the solver does
    time_it("adj + corr fused-g alternating (per pair)", [&] { pt_launch_adj(p, feat, (long)C * H * W, R, gpart, 0); pt_launch_corr(p, feat, (long)C * H * W, nullptr, spart, 0, &fz); }, 200, 2 * B);
    return 0;
}
