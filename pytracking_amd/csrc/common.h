// Shared device helpers for libpt_hot (gfx950 only: wave64, MFMA f32 16x16x4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define PT_WAVE 64

// Raw buffer loads: 32-bit byte offsets from one wave-uniform resource descriptor (cheaper address arithmetic than
// 64-bit flat pointers; out-of-range offsets return 0 instead of faulting).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pt_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 pt_bload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ float pt_bload1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// D(16x16) += A(16x4) * B(4x16), exact f32 (fmaf chain in k order).
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; holds D[4*(l>>4)+r][l&15], r=0..3.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Wave-wide reductions, every lane gets the result.  Inside a row of 16 lanes the butterfly runs on DPP operands (xor 1 and
// xor 2 as quad permutes, then row_half_mirror and row_mirror: each step adds the partner group's total, so all lanes of the
// row hold the same bits); the four rows are combined in fixed order through v_readlane.  ~10 instructions of a few cycles
// instead of six dependent ds_bpermute round trips (the alpha of the k_adj2 prologue sat behind twelve of them).
// (a lane whose DPP source is switched off receives `old`: 0 for sums, its own value for maxima; the callers run with full waves)
template <int CTRL>
__device__ __forceinline__ float pt_dpp(float v, float old = 0.f) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float pt_lane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += pt_dpp<0xB1>(v);                  // quad_perm [1,0,3,2]
    v += pt_dpp<0x4E>(v);                  // quad_perm [2,3,0,1]
    v += pt_dpp<0x141>(v);                 // row_half_mirror
    v += pt_dpp<0x140>(v);                 // row_mirror
    return ((pt_lane(v, 0) + pt_lane(v, 16)) + pt_lane(v, 32)) + pt_lane(v, 48);
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, pt_dpp<0xB1>(v, v));
    v = fmaxf(v, pt_dpp<0x4E>(v, v));
    v = fmaxf(v, pt_dpp<0x141>(v, v));
    v = fmaxf(v, pt_dpp<0x140>(v, v));
    return fmaxf(fmaxf(pt_lane(v, 0), pt_lane(v, 16)), fmaxf(pt_lane(v, 32), pt_lane(v, 48)));
}

// Workgroup barrier for hand-offs THROUGH LDS ONLY.  __syncthreads() is a workgroup-scope fence + s_barrier: in front of it the compiler
// waits for every outstanding vector-memory operation as well (s_waitcnt vmcnt(0)) -- prefetched feature loads, and the write
// acknowledgements of result stores nobody in the workgroup reads (k_adj2's home waves: ~0.4 us in front of the barrier the whole
// workgroup then sits at).  Here: LDS / scalar counter only, then the barrier.  NOT for data exchanged through global memory.
#ifndef PT_LDS_BARRIER
#define PT_LDS_BARRIER 1
#endif
__device__ __forceinline__ void pt_lds_barrier() {
#if PT_LDS_BARRIER
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    __syncthreads();
#endif
}

// Deterministic block-wide sum; `scratch` needs blockDim.x/64 floats of LDS.  Every thread gets the result.
// (`nthreads`: pass the block size when the kernel knows it -- blockDim.x is an implicit kernel argument, a scalar load of its own)
__device__ __forceinline__ float block_sum(float v, float* scratch, int nthreads = 0) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = ((nthreads > 0 ? nthreads : (int)blockDim.x) + 63) >> 6;
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

// The argument block of a kernel whose leading scalar parameters are preloaded (-amdgpu-kernarg-preload-count): fetched BEHIND
// the point where the prologue has requested its operands, in ONE batch, with ONE wait.
//   * The compiler materialises accesses to a by-value struct parameter as scalar loads at the kernel entry and waits for all of
//     them (`s_waitcnt lgkmcnt(0)`: scalar loads return out of order) at the first instruction that uses one OR overwrites a scalar
//     register one of them is landing in -- in practice a few instructions into the prologue, ~1.2 us in front of the first
//     vector load.  pt_late_args() therefore reads the block through a pointer into the kernel-argument segment that the compiler
//     cannot see through (`asm volatile`), copied word by word through the constant address space (scalar loads; unused words
//     disappear).  `byte_offset` = offset of the struct parameter in the segment (the scalars in front of it, 8-aligned).
//   * Whole 16-dword blocks are fetched and defined in SGPRs by an `asm volatile` right there: otherwise the compiler re-fetches
//     fields at their uses, one ~1.2 us stall each, where it finds that cheaper than holding them in registers.
//   * The fetch never leaves the struct: whole 16-dword blocks for sizeof(T) / 64, then ONE narrower load per set bit of the
//     remaining dword count (x8 / x4 / x2 / x1).  (Round 3 rounded the last block up to 64 bytes and read up to 56 bytes past
//     the end of the kernel-argument segment -- a fault if the segment ends at the end of the runtime's argument pool.)
typedef int pt_i32x16 __attribute__((ext_vector_type(16)));
typedef int pt_i32x8 __attribute__((ext_vector_type(8)));
typedef int pt_i32x4 __attribute__((ext_vector_type(4)));
typedef int pt_i32x2 __attribute__((ext_vector_type(2)));
template <typename T>
struct PtLate {
    static constexpr int NW = sizeof(T) / 4, NB = NW / 16, R = NW % 16;
    pt_i32x16 blk[NB > 0 ? NB : 1];
    pt_i32x8 t8;
    pt_i32x4 t4;
    pt_i32x2 t2;
    int t1;
};
// request the block (s_load_dwordx16 each + the narrow tail); exactly sizeof(T) bytes are read
template <typename T>
__device__ __forceinline__ PtLate<T> pt_late_issue(unsigned byte_offset) {
    static_assert(sizeof(T) % 4 == 0 && sizeof(T) <= 3 * 64, "argument blocks: whole dwords, at most 3 x 16 (SGPR budget)");
    typedef PtLate<T> L;
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    p += byte_offset;
    asm volatile("" : "+s"(p)::"memory");
    L r;
#pragma unroll
    for (int b = 0; b < L::NB; ++b) r.blk[b] = ((const pt_i32x16 __attribute__((address_space(4)))*)p)[b];
    const char __attribute__((address_space(4)))* q = p + 64 * L::NB;
    if constexpr ((L::R & 8) != 0) { r.t8 = *(const pt_i32x8 __attribute__((address_space(4)))*)q; q += 32; }
    if constexpr ((L::R & 4) != 0) { r.t4 = *(const pt_i32x4 __attribute__((address_space(4)))*)q; q += 16; }
    if constexpr ((L::R & 2) != 0) { r.t2 = *(const pt_i32x2 __attribute__((address_space(4)))*)q; q += 8; }
    if constexpr ((L::R & 1) != 0) { r.t1 = *(const int __attribute__((address_space(4)))*)q; }
    return r;
}
// the one place where the fetch is waited for.  Every dword of a block is live up to here (the asm takes whole blocks), so no
// scalar register under a landing load is reused in between -- that is what makes a wait appear early
template <typename T>
__device__ __forceinline__ T pt_late_get(PtLate<T>& r) {
    typedef PtLate<T> L;
    struct Words { int w[L::NW]; };
    Words tmp;
#pragma unroll
    for (int b = 0; b < L::NB; ++b) {
        asm volatile("" : "+s"(r.blk[b]));
#pragma unroll
        for (int k = 0; k < 16; ++k) tmp.w[b * 16 + k] = r.blk[b][k];
    }
    int at = 16 * L::NB;
    if constexpr ((L::R & 8) != 0) {
        asm volatile("" : "+s"(r.t8));
#pragma unroll
        for (int k = 0; k < 8; ++k) tmp.w[at + k] = r.t8[k];
        at += 8;
    }
    if constexpr ((L::R & 4) != 0) {
        asm volatile("" : "+s"(r.t4));
#pragma unroll
        for (int k = 0; k < 4; ++k) tmp.w[at + k] = r.t4[k];
        at += 4;
    }
    if constexpr ((L::R & 2) != 0) {
        asm volatile("" : "+s"(r.t2));
        tmp.w[at] = r.t2[0]; tmp.w[at + 1] = r.t2[1];
        at += 2;
    }
    if constexpr ((L::R & 1) != 0) {
        asm volatile("" : "+s"(r.t1));
        tmp.w[at] = r.t1;
    }
    return __builtin_bit_cast(T, tmp);
}
template <typename T>
__device__ __forceinline__ T pt_late_args(unsigned byte_offset) {
    PtLate<T> r = pt_late_issue<T>(byte_offset);
    return pt_late_get<T>(r);
}

// A pointer that came out of pt_late_args() is an integer pair to the compiler: accesses through it are FLAT (address-space
// check per access, counted on both the vector-memory and the LDS counter).  pt_global() states that it points to global memory.
#ifdef PT_GLOBAL_OFF                                // A/B switch: pt_global() becomes a no-op (flat accesses)
template <typename T>
using pt_gptr = T*;
#else
template <typename T>
using pt_gptr = T __attribute__((address_space(1)))*;
#endif
// Pointer members of the late-fetched argument blocks of the solver / ATOM kernels have these types.  Declared global
// (-DPT_LATE_GLOBAL) every access through them is a global_load / global_store instead of flat_* (134-166 flat accesses in k_adj2,
// 3428 in fast_passes.hip) -- and the frame is 1 % SLOWER (A/B on one box, twice: 109.6 / 109.5 vs 108.5 / 108.6 us, k_corr2 8.43 vs
// 8.34, k_adj2 7.77 vs 7.66 us; profiles/r03ae_late_pointer_address_space.txt): they are result stores and late loads, and the
// compiler's schedule around them differs.  The generic form stays the default; the IoU kernels use pt_global() where it was measured
// with their loads.
#ifdef PT_LATE_GLOBAL
typedef const float __attribute__((address_space(1)))* pt_gcf;
typedef float __attribute__((address_space(1)))* pt_gf;
#else
typedef const float* pt_gcf;
typedef float* pt_gf;
#endif
template <typename T>
__device__ __forceinline__ pt_gptr<T> pt_global(T* p) { return (pt_gptr<T>)p; }

static inline int pt_ceil_div(int a, int b) { return (a + b - 1) / b; }
