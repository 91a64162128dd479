// Diagnostic: where do the waves of the integer Gram kernel spend their cycles?  (not product code)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gram_i8_probe.hip -o tools/gram_i8_probe && tools/gram_i8_probe
#ifndef NO_STAMPS  // -DNO_STAMPS: time the product kernel as it ships
#define DMF_STAMPS 1
#endif
#include "../demethify_amd/csrc/dmf_kernels_gram_i8.hip"
#include <cstdio>
#include <random>
#include <vector>
#ifndef PROBE_RING
#define PROBE_RING 8
#endif
using namespace dmf;
int main() {
    const int64_t N = 1000000; const int S = 256, n_c = 12, n_u = 4, SD = 256, ND = 1;
    const int NF = n_c * n_u + n_u * (n_u + 1) / 2;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0, 1);
    std::vector<double> hR((size_t)N * n_c), hu((size_t)N * n_u), hD((size_t)N * S);
    for (auto& x : hR) x = U(rng); for (auto& x : hu) x = U(rng); for (auto& x : hD) x = 1 + (int)(U(rng) * 80);
    std::vector<short> fa, fb;
    for (int l = n_c; l < n_c + n_u; ++l) for (int k = 0; k <= l; ++k) { fa.push_back((short)k); fb.push_back((short)l); }
    double *R, *u, *D; unsigned short* D16; signed char* Dt8; short *dfa, *dfb; long long* slab; unsigned long long* stamps;
    const int64_t N16 = (N + 15) / 16 * 16, plane = ((N + 31) / 32) * (SD / 32) * 1024;
    hipMalloc(&R, hR.size() * 8); hipMalloc(&u, hu.size() * 8 + 16); hipMalloc(&D, hD.size() * 8); hipMalloc(&D16, (size_t)N16 * SD * 2);
    hipMalloc(&Dt8, plane); hipMalloc(&dfa, NF * 2); hipMalloc(&dfb, NF * 2);
    hipMemcpy(R, hR.data(), hR.size() * 8, hipMemcpyHostToDevice); hipMemcpy(u, hu.data(), hu.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(D, hD.data(), hD.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dfa, fa.data(), NF * 2, hipMemcpyHostToDevice); hipMemcpy(dfb, fb.data(), NF * 2, hipMemcpyHostToDevice);
    launch_build_counts_int(D, N, S, ND, D16, N16, SD, Dt8, plane, 0);
    int nsh, ny; int64_t rpw; gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    const int64_t words = gram_i8_slab_words(N, SD, n_c, n_u);
    hipMalloc(&slab, words * 8); hipMalloc(&stamps, (size_t)nsh * ny * 8 * 8 * 8); hipMemset(stamps, 0, (size_t)nsh * ny * 8 * 8 * 8);
    const size_t lds = gram_i8_w8_lds_bytes(1, 1, PROBE_RING);
    hipFuncSetAttribute((const void*)k_gram_i8_w8<1, 1, PROBE_RING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_gram_i8_w8<1, 1, PROBE_RING>), dim3(nsh * ny), dim3(512), lds, 0, Dt8, plane, SD / 32, R, 12, u, N, n_c, n_u, dfa, dfb, NF, 0, 64, rpw, slab, SD, (const int*)nullptr
#ifdef DMF_STAMPS
                           , stamps
#endif
        );
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("launch %d: %.3f ms (%s)\n", rep, ms, hipGetErrorString(hipGetLastError()));
    }
#ifndef DMF_STAMPS
    return 0;
#endif
    std::vector<unsigned long long> hs((size_t)nsh * ny * 8 * 8);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    const char* nm[8] = {"-", "counted wait (vmcnt, lgkmcnt)", "barrier", "7 x (MFMA, conversion piece, LDS refills), drained",
                         "digit stores", "DMA issue (behind the MFMAs)", "-", "-"};
    double sum[8] = {0}; for (size_t i = 0; i < hs.size(); ++i) sum[i & 7] += (double)hs[i];
    const double waves = (double)nsh * ny * 8, blocks = (double)rpw / 32;
    double tot = 0; for (double x : sum) tot += x;
    printf("grid %d x %d, %.0f blocks per workgroup: %.0f cycles per block and wave\n", nsh, ny, blocks, tot / waves / blocks);
    for (int i = 0; i < 6; ++i) printf("   %-40s %5.1f %%  (%.0f cycles per block)\n", nm[i], 100 * sum[i] / tot, sum[i] / waves / blocks);
    return 0;
}
