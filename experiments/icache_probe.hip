// Experiment: what does the FIRST pass through a stretch of straight-line code cost after a kernel launch?  Every workgroup runs the
// same block of N independent-of-memory VALU instructions twice and records the 100 MHz wall clock around each pass; between launches
// a different kernel runs (as in the per-frame graphs).  If the instruction cache is invalidated / cold at a dispatch, pass 1 pays the
// fetch of the code from L2 and pass 2 does not.
//   hipcc --offload-arch=gfx950 -O3 experiments/icache_probe.hip -o experiments/icache_probe && experiments/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define BLOCK(v) asm volatile(R256("v_fma_f32 %0, %0, %0, %0\n") R256("v_fma_f32 %0, %0, %0, %0\n") R256("v_fma_f32 %0, %0, %0, %0\n") R256("v_fma_f32 %0, %0, %0, %0\n") : "+v"(v))

__global__ __launch_bounds__(256) void k_probe(unsigned long long* out, float* sink, int passes) {
    float v = threadIdx.x * 1e-9f;
    unsigned long long t[5];
    t[0] = wall_clock64();
    for (int p = 0; p < passes && p < 4; ++p) {
        BLOCK(v);                                   // 1024 instructions, 8 KB of code
        t[p + 1] = wall_clock64();
    }
    if ((threadIdx.x & 63) == 0)
        for (int p = 0; p <= passes && p < 5; ++p) out[((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + p] = t[p];
    if (v == 123.f) sink[0] = v;
}

__global__ void k_other(float* x) { x[threadIdx.x] += 1.f; }

int main() {
    const int NB = 256, passes = 3;
    unsigned long long* d; float* sink;
    hipMalloc(&d, NB * 4 * 8 * 8); hipMalloc(&sink, 4096);
    hipMemset(sink, 0, 4096);
    std::vector<unsigned long long> h(NB * 4 * 8);
    for (int waves = 1; waves <= 4; waves *= 2) {
        double acc[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
        const int reps = 20;
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(k_other, dim3(64), dim3(256), 0, 0, sink);
            hipLaunchKernelGGL(k_probe, dim3(NB), dim3(64 * waves), 0, 0, d, sink, passes);
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            for (int p = 0; p < passes; ++p) {
                double s = 0, m = 0;
                for (int b = 0; b < NB; ++b) {
                    const double dt = (double)(h[(b * 4) * 8 + p + 1] - h[(b * 4) * 8 + p]) * 0.01;
                    s += dt; m = std::max(m, dt);
                }
                acc[p] += s / NB; mx[p] = std::max(mx[p], m);
            }
        }
        printf("%d wave(s) per workgroup, 1024 v_fma per pass: pass 1 %.2f us (max %.2f), pass 2 %.2f (max %.2f), pass 3 %.2f (max %.2f)\n", waves,
               acc[0] / reps, mx[0], acc[1] / reps, mx[1], acc[2] / reps, mx[2]);
    }
    return 0;
}
