// How fast can this chip issue v_mfma_f32_16x16x4_f32 at all?  Register-only kernel (no memory, no LDS): every wavefront
// runs ITER x 16 MFMAs on NACC independent accumulators.  Reference point for the MFMA-busy ceiling of ~0.45 that every
// MFMA kernel of this repository shows.   hipcc --offload-arch=gfx950 -O3 experiments/mfma_peak.hip -o experiments/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_peak(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NACC], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC>
void run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_peak<NACC>), dim3(blocks), dim3(256), 0, 0, out, 64, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_peak<NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 16 * 2048.0;
    printf("{\"nacc\": %d, \"blocks\": %d, \"waves_per_simd\": %.1f, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", NACC, blocks,
           blocks * 4 / 1024.0, ms, flops / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 8192 * 256 * sizeof(float));
    for (int blocks : {256, 512, 1024, 2048}) {
        run<1>(blocks, 20000, out);
        run<2>(blocks, 20000, out);
        run<4>(blocks, 20000, out);
    }
    return 0;
}
