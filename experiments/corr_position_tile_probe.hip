// Micro-benchmark (not part of the product), round 6: what would a POSITION-COMPLETE correlation pass cost?
//
// The steepest-descent iteration is three launches (adjoint -> correlation -> pointwise k_fast_sgq2) because the correlation's
// workgroups own a channel range of a sample: F g of a position is only complete after the 8 channel-range partial maps are summed,
// and alpha needs sum_i |D_i F g|^2 -- hence the pointwise launch in between.  A correlation workgroup that takes ALL 512 channels of
// a (sample, row band) would finish F g, q_i in-workgroup and the pointwise launch would disappear (18 -> 13 launches per frame).
// The price (judge's list): every workgroup needs the WHOLE reduced gradient (8 partials x 512 x 16 floats =
// 256 KB instead of a 1/8 slice), row bands need a 3-row halo for the 4x4 shift-and-add, and there are n x R workgroups streaming
// 663 KB / R each instead of 8n / 2 streaming 166 KB.  This probe measures the stripped kernel: filter operand reduction, feature
// stream, MFMAs, tap planes into LDS -- no shift-and-add, no q_i epilogue -- for R = 1, 2, 3 row bands, with and without the
// 8-partial reduction.  Decision rule: worth building if probe <= k_corr2 + k_fast_sgq2 in-chain (8.2 + 2.9 us).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/corr_position_tile_probe.hip -o experiments/corr_position_tile_probe
//   rocprofv3 --kernel-trace --stats -d out -- ./experiments/corr_position_tile_probe        (per-kernel durations)
// XCD placement: workgroup b runs on XCD b % 8 (observed); sample i is served by XCD i % 8 in every launch, so a pass re-reads what its
// XCD's L2 holds from the previous one (4.15 MB per XCD), like the product's channel-range ownership.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define HW 324
#define WD 18
#define C 512
#define NPART 8

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// R row bands per sample; TILES 64-position tiles per band; KG k-step groups per tile (waves = TILES * KG); REDUCE: 8 partials + reg*w
template <int R, int TILES, int KG, bool REDUCE>
__global__ __launch_bounds__(1024) void k_probe(const float* __restrict__ feat, const float* __restrict__ gpart, const float* __restrict__ w,
                                               float* __restrict__ out, int n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // afilt[C][16] | T[KG][16][64*TILES + 4]
    constexpr int HWP = 64 * TILES + 4, NK = C / 4 / KG, CD = 4;
    const int b = blockIdx.x, x = b & 7, q = b >> 3, i = x + 8 * (q / R), part = q % R;
    if (i >= n) return;
    // output rows of the band and the feature rows its 4x4 shift-and-add touches (2 above, 1 below)
    const int o_lo = part * 19 / R, o_hi = (part + 1) * 19 / R;
    const int r_lo = max(o_lo - 2, 0), r_hi = min(o_hi + 1, 18);
    const int p_lo = (r_lo * WD) & ~3, p_hi = r_hi * WD;             // positions [p_lo, p_hi), 16-byte aligned start
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, j = lane & 15;
    const int t = wave % TILES, h = wave / TILES;
    float* __restrict__ afilt = lds;
    float* __restrict__ Tl = lds + C * 16 + h * 16 * HWP;
    // ---- filter operand: 2048 16-byte pieces over 1024 threads
    f32x4 fv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int piece = tid + 1024 * e;
        if (REDUCE) {
            f32x4 s = {0, 0, 0, 0}, pv[NPART];
#pragma unroll
            for (int k = 0; k < NPART; ++k) pv[k] = ((const f32x4*)(gpart + (size_t)k * C * 16))[piece];
            const f32x4 wv = ((const f32x4*)w)[piece];
#pragma unroll
            for (int k = 0; k < NPART; ++k) s += pv[k];
            fv[e] = s + 0.01f * wv;
        } else {
            fv[e] = ((const f32x4*)w)[piece];
        }
    }
    // ---- feature slice of this wave: NK k-steps, CD ahead
    const int pos = p_lo + 64 * t + 4 * j;
    const bool pv_ok = pos < p_hi && wave < TILES * KG;
    const float* __restrict__ fb = feat + ((size_t)i * C + (size_t)(h * NK * 4 + kq)) * HW + (pv_ok ? pos : p_lo);
    f32x4 bq[NK];
#pragma unroll
    for (int k = 0; k < CD; ++k) bq[k] = *(const f32x4*)(fb + (size_t)(4 * k) * HW);
#pragma unroll
    for (int e = 0; e < 2; ++e) ((f32x4*)afilt)[tid + 1024 * e] = fv[e];
    __syncthreads();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    if (wave < TILES * KG) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (k + CD < NK) bq[k + CD] = *(const f32x4*)(fb + (size_t)(4 * (k + CD)) * HW);
            const float av = afilt[(4 * (h * NK + k) + kq) * 16 + j];
            acc0 = mfma16(av, bq[k][0], acc0);
            acc1 = mfma16(av, bq[k][1], acc1);
            acc2 = mfma16(av, bq[k][2], acc2);
            acc3 = mfma16(av, bq[k][3], acc3);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 v = {acc0[r], acc1[r], acc2[r], acc3[r]};
            *(f32x4*)(Tl + (4 * kq + r) * HWP + 64 * t + 4 * j) = v;
        }
    }
    __syncthreads();
    // stand-in for the epilogue: one value per thread leaves the workgroup
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < KG; ++g) s += lds[C * 16 + g * 16 * HWP + (tid % (16 * HWP))];
    out[(size_t)b * 1024 + tid] = s;
}

template <int R, int TILES, int KG, bool REDUCE>
static void run(const char* name, const float* feat, const float* gpart, const float* w, float* out, int n, int reps) {
    constexpr int HWP = 64 * TILES + 4;
    const size_t ldsb = (size_t)(C * 16 + KG * 16 * HWP) * sizeof(float);
    CHECK(hipFuncSetAttribute((const void*)k_probe<R, TILES, KG, REDUCE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const int groups = (n + 7) / 8;                                   // samples per XCD, rounded up
    dim3 grid(8 * groups * R), block(1024);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((k_probe<R, TILES, KG, REDUCE>), grid, block, ldsb, 0, feat, gpart, w, out, n);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((k_probe<R, TILES, KG, REDUCE>), grid, block, ldsb, 0, feat, gpart, w, out, n);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"probe\": \"%s\", \"row_bands\": %d, \"tiles\": %d, \"k_groups\": %d, \"reduce_8_partials\": %s, \"workgroups\": %d, \"waves\": %d, "
           "\"lds_bytes\": %zu, \"back_to_back_period_us\": %.2f}\n", name, R, TILES, KG, REDUCE ? "true" : "false", (int)grid.x, TILES * KG, ldsb,
           1e3 * ms / reps);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 50, reps = 200;
    std::vector<float> hf((size_t)n * C * HW), hg((size_t)NPART * C * 16), hw((size_t)C * 16);
    for (size_t k = 0; k < hf.size(); ++k) hf[k] = (float)((k * 2654435761u) % 1000) * 1e-3f - 0.5f;
    for (size_t k = 0; k < hg.size(); ++k) hg[k] = (float)((k * 40503u) % 1000) * 1e-4f;
    for (size_t k = 0; k < hw.size(); ++k) hw[k] = (float)((k * 9973u) % 1000) * 1e-3f;
    float *feat, *gpart, *w, *out;
    CHECK(hipMalloc(&feat, hf.size() * 4)); CHECK(hipMalloc(&gpart, hg.size() * 4)); CHECK(hipMalloc(&w, hw.size() * 4));
    CHECK(hipMalloc(&out, (size_t)8 * 8 * 3 * 1024 * 4 * 4));
    CHECK(hipMemcpy(feat, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(gpart, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    // whole sample: 324 positions -> 6 tiles x 2 k-groups (12 waves); halves: <= 11 rows + alignment -> 4 tiles x 4; thirds: <= 9 rows -> 3 x 4
    run<1, 6, 2, true>("whole_sample", feat, gpart, w, out, n, reps);
    run<1, 6, 2, false>("whole_sample", feat, gpart, w, out, n, reps);
    run<2, 4, 4, true>("row_halves", feat, gpart, w, out, n, reps);
    run<2, 4, 4, false>("row_halves", feat, gpart, w, out, n, reps);
    run<3, 3, 4, true>("row_thirds", feat, gpart, w, out, n, reps);
    run<3, 3, 4, false>("row_thirds", feat, gpart, w, out, n, reps);
    return 0;
}
