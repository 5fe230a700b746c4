// What can a wavefront issue in the shadow of a v_mfma_f32_16x16x4_f32 (8 passes = 32 cycles)?  Register-only MFMA stream
// with NV independent VALU instructions (v_add_u32, or the quarter-rate v_mad_u64_u32 when W64) and NL ds_read_b32 behind
// every MFMA, at 1 and 2 wavefronts per SIMD.  The LWL correlation kernel carries ~2.4 VALU + 1.3 LDS + 0.5 SALU
// (and, 4th parameter, NS s_add_u32)
// instructions per MFMA and reaches ~50 % MFMA issue rate per wavefront: this tells whether that mix alone explains it.
//   hipcc --offload-arch=gfx950 -O3 experiments/mfma_issue.hip -o experiments/mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, int W64, int NS = 0>
__global__ __launch_bounds__(256) void k_issue(float* out, int iters, float a0, float b0) {
    __shared__ float sm[4096];
    for (int e = threadIdx.x; e < 4096; e += 256) sm[e] = 1.f;
    __syncthreads();
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    unsigned x[8] = {1, 2, 3, 4, 5, 6, 7, 8}, y = threadIdx.x;
    unsigned long long z[4] = {1, 2, 3, 4};
    float r[4] = {0, 0, 0, 0};
    unsigned sx = iters;
    const unsigned addr = (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (W64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(z[v & 3]) : "v"(y) : "vcc");
                else asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[(u * NV + v) & 7]) : "v"(y));
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[(u * NL + l) & 3]) : "v"(addr), "i"(256 * ((u * 2 + l) & 15)));
#pragma unroll
            for (int k = 0; k < NS; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx) : : "scc");
            if (NL && (u & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    f32x4 s = acc[0] + acc[1];
    unsigned xs = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) xs += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + (float)xs + (float)sx + (float)(z[0] + z[1] + z[2] + z[3]) + r[0] + r[1] + r[2] + r[3];
}

template <int NV, int NL, int W64, int NS = 0>
void run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_issue<NV, NL, W64, NS>), dim3(blocks), dim3(256), 0, 0, out, 64, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_issue<NV, NL, W64, NS>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 16 * 2048.0;
    printf("{\"salu_per_mfma\": %d, \"valu_per_mfma\": %d, \"valu64\": %d, \"lds_per_mfma\": %d, \"waves_per_simd\": %.0f, \"TFLOPs\": %.1f, \"cycles_per_mfma_at_2.4GHz\": %.1f}\n",
           NS, NV, W64, NL, blocks * 4 / 1024.0, flops / ms / 1e9, 32.0 * 157.3 / (flops / ms / 1e9) * (blocks >= 512 ? 1 : 1));
}

int main() {
    float* out;
    hipMalloc(&out, 8192 * 256 * sizeof(float));
    for (int blocks : {256, 512}) {
        run<0, 0, 0>(blocks, 5000, out);
        run<1, 0, 0>(blocks, 5000, out);
        run<2, 0, 0>(blocks, 5000, out);
        run<4, 0, 0>(blocks, 5000, out);
        run<6, 0, 0>(blocks, 5000, out);
        run<8, 0, 0>(blocks, 5000, out);
        run<1, 0, 1>(blocks, 5000, out);
        run<2, 0, 1>(blocks, 5000, out);
        run<0, 1, 0>(blocks, 5000, out);
        run<0, 2, 0>(blocks, 5000, out);
        run<2, 1, 0>(blocks, 5000, out);
        run<3, 2, 0>(blocks, 5000, out);
        run<0, 0, 0, 1>(blocks, 5000, out);
        run<0, 0, 0, 2>(blocks, 5000, out);
        run<0, 0, 0, 4>(blocks, 5000, out);
        run<0, 1, 0, 2>(blocks, 5000, out);
    }
    return 0;
}
