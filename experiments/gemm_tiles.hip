// Measurement aid (not part of the product): workgroup-tile shapes of the shared fp32 MFMA GEMM (csrc/mfma_gemm.h) on the
// shapes of this repository -- ToMP encoder (M = 1944), classification-feature head as a plain GEMM (M = 324, K = 9216,
// split-K), IoU head.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pytracking_amd/csrc experiments/gemm_tiles.hip -o experiments/gemm_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../pytracking_amd/csrc/mfma_gemm.h"

static float* dalloc(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) % 2001 - 1000) * 1e-3f * scale; }
    float* d;
    hipMalloc(&d, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}

template <int BM, int BN, int BK>
static float run(const char* tag, int M, int N, int K, int ksteps_per_z, const float* A, const float* W, float* C, const float* ref,
                 int reps) {
    GemmArgs g = gemm_args(A, K, M, W, M, N, K, nullptr, C, N);
    const int nk64 = (K + 63) / 64;                              // split-K is counted in 64-wide steps whatever BK is
    int nz = 1;
    if (ksteps_per_z > 0) { g.ksteps = ksteps_per_z; g.c_zstride = (long)M * N; nz = (nk64 + ksteps_per_z - 1) / ksteps_per_z; }
    const int gy = (M + BM - 1) / BM;
    g.swizzle = nz == 1 && gy >= 16;
    dim3 grid((N + BN - 1) / BN, g.swizzle ? (gy + 7) / 8 * 8 : gy, nz), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gemm<BM, BN, 0, BK>), grid, block, 0, 0, g);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%-28s launch failed\n", tag); return -1.f; }
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gemm<BM, BN, 0, BK>), grid, block, 0, 0, g);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const float us = ms * 1e3f / reps;
    // check (sum of the split-K slabs) against the reference on a sample of entries
    std::vector<float> h((size_t)M * N * nz);
    hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
    double err = 0;
    for (size_t i = 0; i < (size_t)M * N; i += 97) {
        double s = 0;
        for (int z = 0; z < nz; ++z) s += h[(size_t)z * M * N + i];
        err = fmax(err, fabs(s - ref[i]));
    }
    printf("%-28s tile %3dx%3dx%2d z=%d grid %4d  %7.2f us  %6.1f TFLOP/s  err %.1e\n", tag, BM, BN, BK, nz,
           (int)(grid.x * grid.y * grid.z), us, 2.0 * M * N * K / us / 1e6, err);
    return us;
}

// the 8-wave 128x128x32 kernel with the pinned schedule (k_gemm_big)
static float run_big(const char* tag, int M, int N, int K, int ksteps_per_z, const float* A, const float* W, float* C, const float* ref,
                     int reps) {
    GemmArgs g = gemm_args(A, K, M, W, M, N, K, nullptr, C, N);
    const int nk64 = (K + 63) / 64;
    int nz = 1;
    if (ksteps_per_z > 0) { g.ksteps = ksteps_per_z; g.c_zstride = (long)M * N; nz = (nk64 + ksteps_per_z - 1) / ksteps_per_z; }
    dim3 grid((N + 127) / 128, (M + 127) / 128, nz), block(512);
    const size_t lds = 2 * GB_STAGE * sizeof(float);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_gemm_big, grid, block, lds, 0, g);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%-28s big: launch failed\n", tag); return -1.f; }
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gemm_big, grid, block, lds, 0, g);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const float us = ms * 1e3f / reps;
    std::vector<float> h((size_t)M * N * nz);
    hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
    double err = 0;
    for (size_t i = 0; i < (size_t)M * N; i += 97) {
        double s = 0;
        for (int z = 0; z < nz; ++z) s += h[(size_t)z * M * N + i];
        err = fmax(err, fabs(s - ref[i]));
    }
    printf("%-28s BIG  128x128x32 8 waves z=%d grid %4d  %7.2f us  %6.1f TFLOP/s  err %.1e\n", tag, nz,
           (int)(grid.x * grid.y * grid.z), us, 2.0 * M * N * K / us / 1e6, err);
    return us;
}

int main() {
    struct Shape { const char* name; int M, N, K; } shapes[] = {{"qkv 1944x768x256", 1944, 768, 256}, {"out 1944x256x256", 1944, 256, 256},
        {"ffn1 1944x2048x256", 1944, 2048, 256}, {"ffn2 1944x256x2048", 1944, 256, 2048}, {"head 324x512x9216", 324, 512, 9216},
        {"iou 10x256x6400", 10, 256, 6400}};
    for (auto& s : shapes) {
        float* A = dalloc((size_t)s.M * s.K, 1, 1.f);
        float* W = dalloc((size_t)s.N * s.K, 2, 0.05f);
        float* C;
        hipMalloc(&C, (size_t)s.M * s.N * 4 * 72);
        // reference = the 32x32 kernel without split-K (exact fp32 chain), on the host copy
        std::vector<float> ref((size_t)s.M * s.N);
        {
            GemmArgs g = gemm_args(A, s.K, s.M, W, s.M, s.N, s.K, nullptr, C, s.N);
            hipLaunchKernelGGL((k_gemm<32, 32, 0, 64>), dim3((s.N + 31) / 32, (s.M + 31) / 32, 1), dim3(256), 0, 0, g);
            hipMemcpy(ref.data(), C, ref.size() * 4, hipMemcpyDeviceToHost);
        }
        const int R = 50;
        const bool deepk = s.K >= 2048;
        run<32, 32, 64>(s.name, s.M, s.N, s.K, 0, A, W, C, ref.data(), R);
        if (deepk) run<32, 32, 64>(s.name, s.M, s.N, s.K, s.K / 64 / 4, A, W, C, ref.data(), R);
        if (deepk) run<64, 64, 64>(s.name, s.M, s.N, s.K, s.K / 64 / 8, A, W, C, ref.data(), R);
        run<64, 64, 64>(s.name, s.M, s.N, s.K, 0, A, W, C, ref.data(), R);
        run<64, 64, 32>(s.name, s.M, s.N, s.K, 0, A, W, C, ref.data(), R);
        if (deepk) run<64, 64, 32>(s.name, s.M, s.N, s.K, s.K / 64 / 8, A, W, C, ref.data(), R);
        run<128, 64, 32>(s.name, s.M, s.N, s.K, 0, A, W, C, ref.data(), R);
        if (deepk) run<128, 64, 32>(s.name, s.M, s.N, s.K, s.K / 64 / 8, A, W, C, ref.data(), R);
        run<128, 128, 32>(s.name, s.M, s.N, s.K, 0, A, W, C, ref.data(), R);
        if (deepk) run<128, 128, 32>(s.name, s.M, s.N, s.K, s.K / 64 / 8, A, W, C, ref.data(), R);
        if (deepk) run<128, 128, 32>(s.name, s.M, s.N, s.K, s.K / 64 / 16, A, W, C, ref.data(), R);
        run_big(s.name, s.M, s.N, s.K, 0, A, W, C, ref.data(), R);
        if (deepk) run_big(s.name, s.M, s.N, s.K, s.K / 64 / 8, A, W, C, ref.data(), R);
        if (deepk) run_big(s.name, s.M, s.N, s.K, s.K / 64 / 16, A, W, C, ref.data(), R);
        hipFree(A); hipFree(W); hipFree(C);
    }
    return 0;
}
