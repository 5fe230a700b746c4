// Second reference point for the GEMM inner loop: the SAME per-K-step structure as k_gemm<32,32> (8 ds_read_b128 fragment
// reads, 16 MFMAs, 4 ds_write_b128, one s_barrier; 4 wavefronts per workgroup) but WITHOUT global loads.  If this runs
// near the MFMA peak, the global-load path is what holds k_gemm at ~0.45; if it does not, the LDS / barrier structure is.
//   hipcc --offload-arch=gfx950 -O3 experiments/mfma_lds_peak.hip -o experiments/mfma_lds_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WRITES, int BARRIER>
__global__ __launch_bounds__(256) void k_step(float* out, int steps) {
    __shared__ __attribute__((aligned(16))) float As[2][32 * 68];
    __shared__ __attribute__((aligned(16))) float Bs[2][32 * 68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 32 * 68; i += 256) { (&As[0][0])[i] = 1e-3f * (i & 7); (&Bs[0][0])[i] = 1e-3f; }
    __syncthreads();
    const int aso = ((wave >> 1) * 16 + (lane & 15)) * 68 + (lane >> 4) * 4, bso = ((wave & 1) * 16 + (lane & 15)) * 68 + (lane >> 4) * 4;
    const int lrow = tid >> 4, lc4 = (tid & 15) * 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 w0 = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        f32x4 fa[4], fb[4];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            fa[hh] = *reinterpret_cast<const f32x4*>(&As[buf][aso + hh * 16]);
            fb[hh] = *reinterpret_cast<const f32x4*>(&Bs[buf][bso + hh * 16]);
        }
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[hh][j], fb[hh][j], acc, 0, 0, 0);
        if (WRITES) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<f32x4*>(&As[buf ^ 1][(lrow + 16 * i) * 68 + lc4]) = w0;
                *reinterpret_cast<f32x4*>(&Bs[buf ^ 1][(lrow + 16 * i) * 68 + lc4]) = w0;
            }
        }
        if (BARRIER) __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int WRITES, int BARRIER>
void run(int blocks, int steps, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_step<WRITES, BARRIER>), dim3(blocks), dim3(256), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_step<WRITES, BARRIER>), dim3(blocks), dim3(256), 0, 0, out, steps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * steps * 16 * 2048.0;
    printf("{\"lds_writes\": %d, \"barrier\": %d, \"workgroups_per_cu\": %.1f, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", WRITES, BARRIER,
           blocks / 256.0, ms, flops / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    for (int blocks : {256, 512, 1024}) {
        run<0, 0>(blocks, 20000, out);
        run<1, 0>(blocks, 20000, out);
        run<1, 1>(blocks, 20000, out);
    }
    return 0;
}
