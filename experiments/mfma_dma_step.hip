// Third reference point: the k_gemm<32,32> K-step with the operands brought in by LDS-DMA
// (`buffer_load_dwordx4 ... lds` = __builtin_amdgcn_raw_ptr_buffer_load_lds, 1 KiB per wavefront instruction, no VGPR
// round trip, no ds_write) instead of buffer loads -> registers -> ds_write_b128.  Unpadded 64-float rows; bank conflicts
// are avoided by an XOR swizzle of the 16-byte quad index with the row (the lane -> global address map applies it on the
// way in, the fragment reads on the way out).  Throughput probe only (no result check).
//   hipcc --offload-arch=gfx950 -O3 experiments/mfma_dma_step.hip -o experiments/mfma_dma_step
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ __launch_bounds__(256) void k_step(const float* A, const float* W, float* out, int steps, int K, int rowsA) {
    __shared__ __attribute__((aligned(16))) float As[2][32 * 64];
    __shared__ __attribute__((aligned(16))) float Bs[2][32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)rowsA * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (unsigned)rowsA * K * 4, 0x00020000);
    const int m0 = (blockIdx.x * 32) % (rowsA - 32);
    // loader: wavefront w, instruction i covers tile rows 8w + 4i .. +3; lane = (row_local, quad')
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * wave + 4 * i + (lane >> 4), quad = (lane & 15) ^ (row & 15);
            const unsigned off = (unsigned)(((m0 + row) * K + (s * 64) % K + quad * 4) * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)&As[buf][(8 * wave + 4 * i) * 64], 16, off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)&Bs[buf][(8 * wave + 4 * i) * 64], 16, off, 0, 0, 0);
        }
    };
    const int ra_ = (wave >> 1) * 16 + (lane & 15), rb_ = (wave & 1) * 16 + (lane & 15), kq = lane >> 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        issue(s + 1, buf ^ 1);
        f32x4 fa[4], fb[4];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            fa[hh] = *reinterpret_cast<const f32x4*>(&As[buf][ra_ * 64 + (((hh * 4 + kq) ^ (ra_ & 15)) * 4)]);
            fb[hh] = *reinterpret_cast<const f32x4*>(&Bs[buf][rb_ * 64 + (((hh * 4 + kq) ^ (rb_ & 15)) * 4)]);
        }
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[hh][j], fb[hh][j], acc, 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);                    // vmcnt(0): the DMA of step s+1 has landed
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    const int K = 2048, rows = 2048;
    float *A, *W, *out;
    hipMalloc(&A, (size_t)rows * K * 4);
    hipMalloc(&W, (size_t)rows * K * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMemset(A, 0, (size_t)rows * K * 4);
    hipMemset(W, 0, (size_t)rows * K * 4);
    for (int blocks : {256, 512, 1024}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, 0, A, W, out, 16, K, rows);
        hipDeviceSynchronize();
        const int steps = 8000;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, 0, A, W, out, steps, K, rows);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("{\"staging\": \"lds-dma\", \"workgroups_per_cu\": %.1f, \"ms\": %.3f, \"TFLOPs\": %.1f, \"err\": \"%s\"}\n", blocks / 256.0, ms,
               (double)blocks * 4 * steps * 16 * 2048.0 / ms / 1e9, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
