// Shared device helpers for libpt_hot (gfx950 only: wave64, MFMA f32 16x16x4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PT_WAVE 64

// D(16x16) += A(16x4) * B(4x16), exact f32 (fmaf chain in k order).
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; holds D[4*(l>>4)+r][l&15], r=0..3.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Deterministic block-wide sum; `scratch` needs blockDim.x/64 floats of LDS.  Every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += scratch[w];
    return t;
}

__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = scratch[0];
    for (int w = 1; w < nw; ++w) t = fmaxf(t, scratch[w]);
    return t;
}

static inline int pt_ceil_div(int a, int b) { return (a + b - 1) / b; }
