// Diagnostic: where do the waves of the second-generation row pass spend their cycles?  (not product code)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/rowpass2_probe.hip -o tools/rowpass2_probe && tools/rowpass2_probe [T2] [wg per CU]
//   -DPROBE_NO_STAMPS: the kernel as it ships (launch times only); -DPROBE_KERNEL_SRC="\"file\"": another version of the source
#ifndef PROBE_NO_STAMPS
#define DMF_STAMPS 1
#endif
#ifdef PROBE_KERNEL_SRC
#include PROBE_KERNEL_SRC
#else
#include "../demethify_amd/csrc/dmf_kernels_rowpass2.hip"
#endif
#include <cstdio>
#include <random>
#include <vector>
using namespace dmf;
int main(int argc, char** argv) {
    const int64_t N = 1000000; const int S = 256, n_c = 12, n_u = 4, K = 16;
    const int T2 = argc > 1 ? atoi(argv[1]) : 20;
    const int per_cu = argc > 2 ? atoi(argv[2]) : 2;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0, 1);
    std::vector<double> hV((size_t)N * S), hR((size_t)N * n_c), hu((size_t)N * n_u), ha((size_t)K * S);
    std::vector<unsigned short> hD((size_t)N * S);
    for (auto& x : hV) x = U(rng); for (auto& x : hD) x = 1 + (int)(U(rng) * 80); for (auto& x : hR) x = U(rng);
    for (auto& x : hu) x = U(rng); for (auto& x : ha) x = U(rng) / K;
    double *V, *R, *u, *up, *a, *slab, *u2; unsigned short* D; SolverState* st; unsigned long long* stamps;
    hipMalloc(&V, hV.size() * 8); hipMalloc(&D, hD.size() * 2); hipMalloc(&R, hR.size() * 8); hipMalloc(&u, hu.size() * 8);
    hipMalloc(&up, hu.size() * 8); hipMalloc(&a, ha.size() * 8); hipMalloc(&u2, 8192 * 8); hipMalloc(&st, sizeof(SolverState));
    const int grid = 256 * per_cu;
    hipMalloc(&slab, (size_t)grid * n_u * S * 8); hipMalloc(&stamps, (size_t)grid * 4 * 16 * 8);
    hipMemcpy(V, hV.data(), hV.size() * 8, hipMemcpyHostToDevice); hipMemcpy(D, hD.data(), hD.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(R, hR.data(), hR.size() * 8, hipMemcpyHostToDevice); hipMemcpy(u, hu.data(), hu.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(up, hu.data(), hu.size() * 8, hipMemcpyHostToDevice); hipMemcpy(a, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
    SolverState h{}; h.a1 = 1; h.a2 = 1; h.l_w = 1e4; h.l_w_prev = 1e4; h.l_h = 1e6; h.l_h_prev = 1e6; h.dsq = 6400;
    hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice); hipMemset(stamps, 0, (size_t)grid * 4 * 16 * 8);
    const size_t lds = rowpass_v2_lds_bytes(S, n_u, T2);
    hipFuncSetAttribute((const void*)k_rowpass_v2<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < (argc > 3 ? atoi(argv[3]) : 3); ++rep) {
        hipEventRecord(e0);
#ifdef PROBE_NO_STAMPS
        hipLaunchKernelGGL((k_rowpass_v2<3, 4>), dim3(grid), dim3(256), lds, 0, V, D, 256, R, a, u, up, st, N, S, n_c, T2, 0, 1, slab, u2);
#else
        hipLaunchKernelGGL((k_rowpass_v2<3, 4>), dim3(grid), dim3(256), lds, 0, V, D, 256, R, a, u, up, st, N, S, n_c, T2, 0, 1, slab, u2, stamps);
#endif
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("launch %d: %.3f ms  (%s)\n", rep, ms, hipGetErrorString(hipGetLastError()));
    }
#ifdef PROBE_NO_STAMPS
    return 0;
#endif
    std::vector<unsigned long long> hs((size_t)grid * 4 * 16);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    const char* an[16] = {"tile store (vmcnt wait)", "phase A (MFMA)", "prefetch issue + partials", "wait X", "phase B / nothing", "wait Y", "phase C", "prefetch issue", "B: partial sums", "B: inner steps", "B: stores", "-", "-", "-", "-", "-"};
    double sum[16] = {0}; int cnt = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < 4; ++w) { for (int i = 0; i < 16; ++i) sum[i] += (double)hs[((size_t)b * 4 + w) * 16 + i]; ++cnt; }
    double tot = 0; for (int i = 0; i < 8; ++i) tot += sum[i];  // (8..15: sub-segments of phase B, already inside segment 4)
    const double steps = (N / 16.0) / grid;
    printf("T2 = %d, %d workgroups per CU: mean cycles per wave %.0f over the kernel, %.0f per block\n", T2, per_cu, tot / cnt, tot / cnt / steps);
    for (int i = 0; i < 11; ++i) if (an[i][0] != '-') printf("   %-28s %6.1f %%  (%.0f cycles per block)\n", an[i], 100 * sum[i] / tot, sum[i] / cnt / steps);
    return 0;
}
