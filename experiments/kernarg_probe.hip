// Micro-benchmark (not part of the product): what does a dependent tiny kernel cost on MI355X depending on HOW its arguments
// arrive?  A chain of 200 dependent launches (each reads one float through a pointer argument, adds fields of its argument
// block, writes one float), eager and as a hipGraph:
//   struct   : 200-byte argument block by value           (scalar loads from the kernel-argument segment)
//   preload  : 14 scalar parameters                        (-amdgpu-kernarg-preload-count=14: SGPRs at wave launch)
//   devblock : pointer (preloaded) to the same block in ordinary device memory (scalar loads through the cache hierarchy)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=14 experiments/kernarg_probe.hip -o experiments/kernarg_probe
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>

struct Blk { const float* in; float* out; int n; float z[46]; };          // 8 + 8 + 4 + 184 = 204 -> 208 bytes

__global__ void k_struct(Blk a) {
    const int i = threadIdx.x;
    if (i < a.n) a.out[i] = a.in[i] + a.z[3] + a.z[40];
}
__global__ void k_preload(const float* in, float* out, int n, float z3, float z40, int p5, int p6, int p7, int p8, int p9, int p10) {
    const int i = threadIdx.x;
    if (i < n) out[i] = in[i] + z3 + z40 + (float)(p5 + p6 + p7 + p8 + p9 + p10);
}
__global__ void k_devblock(const Blk __attribute__((address_space(4)))* a) {
    const int i = threadIdx.x;
    if (i < a->n) a->out[i] = a->in[i] + a->z[3] + a->z[40];
}
__global__ void k_stream(const float4* p, size_t nvec, float* out) {      // evicts caches in between (optional)
    float acc = 0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nvec; e += (size_t)gridDim.x * blockDim.x) acc += p[e].x;
    if (acc == 123.456f) out[0] = acc;
}

template <typename F>
void run(const char* name, F launch, hipStream_t st, int chain, bool with_stream, const float4* big, size_t nvec, float* sink) {
    auto body = [&]() {
        for (int k = 0; k < chain; ++k) {
            launch(k);
            if (with_stream) hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, big, nvec, sink);
        }
    };
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    body();
    hipStreamSynchronize(st);
    hipEventRecord(a, st); body(); hipEventRecord(b, st); hipEventSynchronize(b);
    float ms_e = 0; hipEventElapsedTime(&ms_e, a, b);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    body();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms_g = 0; hipEventElapsedTime(&ms_g, a, b);
    printf("%-44s eager %6.2f us/launch   graph %6.2f us/launch\n", name, ms_e * 1e3 / chain, ms_g * 1e3 / (5 * chain));
}

int main() {
    hipStream_t st; hipStreamCreate(&st);
    float *buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
    Blk h = {}; h.in = buf; h.out = buf + 1024; h.n = 64;
    Blk* d; hipMalloc(&d, sizeof(Blk)); hipMemcpy(d, &h, sizeof(Blk), hipMemcpyHostToDevice);
    const size_t nvec = (33u << 20) / 16;
    float4* big; hipMalloc(&big, nvec * 16); hipMemset(big, 0, nvec * 16);
    for (int ws = 0; ws < 2; ++ws) {
        printf("--- %s\n", ws ? "a 33 MB streaming kernel between the links (its time included in both columns)" : "chain of tiny kernels");
        const int chain = 200;
        run("struct by value (208 B)", [&](int k) { Blk a = h; a.in = buf + (k & 1) * 1024; a.out = buf + ((k + 1) & 1) * 1024; hipLaunchKernelGGL(k_struct, dim3(50), dim3(64), 0, st, a); }, st, chain, ws, big, nvec, buf + 4096);
        run("11 scalar parameters (preloaded)", [&](int k) { hipLaunchKernelGGL(k_preload, dim3(50), dim3(64), 0, st, (const float*)(buf + (k & 1) * 1024), buf + ((k + 1) & 1) * 1024, 64, 0.f, 0.f, 0, 0, 0, 0, 0, 0); }, st, chain, ws, big, nvec, buf + 4096);
        run("pointer to the block in device memory", [&](int k) { hipLaunchKernelGGL(k_devblock, dim3(50), dim3(64), 0, st, (const Blk __attribute__((address_space(4)))*)d); }, st, chain, ws, big, nvec, buf + 4096);
    }
    return 0;
}
