// Micro-benchmark (not part of the product): cost of the XCD-aligned passes over the DiMP-50 sample memory, with
// ablations selected at compile time (-DPT_ABL=...): corr2 1 = no MFMA, 2 = no shift-add epilogue; adj2 8 = no gather+MFMA,
// 16 = no update-stage compute.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w experiments/fast_floor.hip -o experiments/fast_floor
#include <cstdio>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#define PT_EXPERIMENT 1
#include "../pytracking_amd/csrc/fast_passes.hip"

void pt_prof_begin(int, hipStream_t) {}
void pt_prof_end(int, hipStream_t) {}

template <typename F>
float time_it(const char* name, F f, int reps = 300, double bytes = 0) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    float us = ms * 1e3f / reps;
    printf("%-52s %8.2f us   %7.1f GB/s\n", name, us, bytes / us * 1e-3);
    return us;
}

int main(int argc, char** argv) {
    const int n = argc > 2 ? atoi(argv[2]) : 50, C = 512, H = argc > 1 ? atoi(argv[1]) : 18, W = H, K = 4, OH = H + 1, OW = W + 1;
    const size_t nfeat = (size_t)n * C * H * W;
    PtFast f = pt_fast_plan(n, C, H, W, K, K, OH, OW);
    printf("n=%d ", n); printf("fast ok=%d tiles=%d left=%d NK=%d corr_lds=%zu | KSPL=%d gper=%d U=%d ns_max=%d adj_lds=%zu E=%d\n", f.ok, f.tiles, f.left, f.NK, f.corr_lds, f.KSPL, f.gper, f.U, f.ns_max, f.adj_lds, f.E);
    float *feat, *filt, *spart, *gpart, *w, *g, *anum, *maps;
    hipMalloc(&feat, nfeat * 4); hipMalloc(&filt, C * 16 * 4); hipMalloc(&w, C * 16 * 4 * 8); hipMalloc(&g, C * 16 * 4);
    hipMalloc(&spart, pt_fast_spart_floats(f) * 4); hipMalloc(&gpart, pt_fast_gpart_floats(f) * 4); hipMalloc(&anum, 4096);
    const size_t nOO = (size_t)n * OH * OW;
    hipMalloc(&maps, nOO * 4 * 16);
    hipMemset(maps, 0, nOO * 4 * 16);
    std::vector<float> h(nfeat);
    for (size_t i = 0; i < nfeat; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(feat, h.data(), nfeat * 4, hipMemcpyHostToDevice);
    hipMemset(filt, 0, C * 16 * 4); hipMemset(w, 0, C * 16 * 4 * 8); hipMemset(gpart, 0, pt_fast_gpart_floats(f) * 4); hipMemset(anum, 0, 4096);
    const double B = nfeat * 4.0;
    const long sn = (long)C * H * W;
    SdArgs sd = SdArgs();
    sd.n = n; sd.C = C; sd.H = H; sd.W = W; sd.K = K; sd.OH = OH; sd.OW = OW; sd.OO = OH * OW; sd.CKK = C * 16; sd.KS = 8; sd.KSPL = f.KSPL;
    sd.kind = PT_SD_DIMP; sd.score_act = PT_ACT_RELU; sd.step = 0.9f; sd.reg = 0.01f;
    sd.label = maps; sd.mask = maps + nOO; sd.sws = maps + 2 * nOO; sd.s = maps + 3 * nOO; sd.s_in = maps + 4 * nOO; sd.sg = maps + 5 * nOO; sd.lms = maps + 6 * nOO; sd.pk = maps + 10 * nOO;
    sd.gpart = gpart; sd.g = g; sd.anum = anum; sd.qs = anum + 64; sd.lossp = anum + 256; sd.w_iters = w; sd.w0 = w;
    time_it("corr2 plain", [&] { pt_launch_corr2(f, feat, sn, filt, spart, 0); }, 300, B);
    PtCorrFuse fz = {gpart, f.KSPL, w, 0.01f, g, anum, nullptr};
    time_it("corr2 fused-g", [&] { pt_launch_corr2(f, feat, sn, nullptr, spart, 0, &fz); }, 300, B);
    time_it("adj2 plain", [&] { pt_launch_adj2_plain(f, feat, sn, maps, gpart, 0); }, 300, B);
    time_it("adj2 sd t=1", [&] { pt_launch_adj2_sd(f, feat, sn, sd, 1, 0, 0); }, 300, B);
#ifdef PT_TRACE
    {
        unsigned long long* tb; const int NB = 4096;
        hipMalloc(&tb, NB * 8 * 8);
        auto trace = [&](const char* name, int nwg, int nst, auto fn) {
            for (int i = 0; i < 5; ++i) fn();
            hipDeviceSynchronize();
            hipMemset(tb, 0, NB * 8 * 8);
            hipMemcpyToSymbol(HIP_SYMBOL(pt_trace_buf), &tb, sizeof(tb));
            fn();
            hipDeviceSynchronize();
            unsigned long long* z = nullptr;
            hipMemcpyToSymbol(HIP_SYMBOL(pt_trace_buf), &z, sizeof(z));
            std::vector<unsigned long long> hb(NB * 8);
            hipMemcpy(hb.data(), tb, NB * 8 * 8, hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull;
            for (int w = 0; w < nwg; ++w) t0 = std::min(t0, hb[w * 8]);
            printf("%s: per-stamp min / mean / max over %d WGs [us since first WG start]\n", name, nwg);
            for (int k = 0; k < nst; ++k) {
                double mn = 1e30, mx = 0, sm = 0;
                for (int w = 0; w < nwg; ++w) { double v = (double)(hb[w * 8 + k] - t0) * 0.01; mn = std::min(mn, v); mx = std::max(mx, v); sm += v; }
                printf("   stamp %d: %7.2f %7.2f %7.2f\n", k, mn, sm / nwg, mx);
            }
        };
        trace("corr2 plain", 8 * n, 6, [&] { pt_launch_corr2(f, feat, sn, filt, spart, 0); });
        trace("corr2 fused", 8 * n, 6, [&] { pt_launch_corr2(f, feat, sn, nullptr, spart, 0, &fz); });
        trace("adj2 plain", f.CB * f.KSPL, 7, [&] { pt_launch_adj2_plain(f, feat, sn, maps, gpart, 0); });
        trace("adj2 sd", f.CB * f.KSPL, 7, [&] { pt_launch_adj2_sd(f, feat, sn, sd, 1, 0, 0); });
    }
#endif
    time_it("adj2 sd + corr2 fused alternating (per pair)", [&] { pt_launch_adj2_sd(f, feat, sn, sd, 1, 0, 0); pt_launch_corr2(f, feat, sn, nullptr, spart, 0, &fz); }, 300, 2 * B);
    return 0;
}
