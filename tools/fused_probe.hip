// Diagnostic: where do the waves of the fused row pass spend their cycles?  (not product code)
#define DMF_STAMPS 1
#include "../demethify_amd/csrc/dmf_kernels_fused.hip"
#include <cstdio>
#include <vector>
#include <random>
using namespace dmf;
int main(int argc, char** argv) {
    const int64_t N = 1000000; const int S = 256, n_c = 12, n_u = 4, K = 16, T2 = 20;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(0, 1);
    std::vector<double> hV((size_t)N * S), hD((size_t)N * S), hR((size_t)N * n_c), hu((size_t)N * n_u), ha((size_t)K * S);
    for (auto& x : hV) x = U(rng); for (auto& x : hD) x = 1 + (int)(U(rng) * 80); for (auto& x : hR) x = U(rng);
    for (auto& x : hu) x = U(rng); for (auto& x : ha) x = U(rng) / K;
    double *V, *D, *R, *u, *up, *a, *slab, *u2; SolverState* st; unsigned long long* stamps;
    hipMalloc(&V, hV.size() * 8); hipMalloc(&D, hD.size() * 8); hipMalloc(&R, hR.size() * 8); hipMalloc(&u, hu.size() * 8);
    hipMalloc(&up, hu.size() * 8); hipMalloc(&a, ha.size() * 8); hipMalloc(&u2, 8192); hipMalloc(&st, sizeof(SolverState));
    const int grid = rowpass_fused_grid(N, S);
    hipMalloc(&slab, (size_t)rowpass_fused_slab_doubles(N, S, n_c, n_u) * 8 * 2); hipMalloc(&stamps, (size_t)grid * 16 * 8 * 8);
    hipMemcpy(V, hV.data(), hV.size() * 8, hipMemcpyHostToDevice); hipMemcpy(D, hD.data(), hD.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(R, hR.data(), hR.size() * 8, hipMemcpyHostToDevice); hipMemcpy(u, hu.data(), hu.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(up, hu.data(), hu.size() * 8, hipMemcpyHostToDevice); hipMemcpy(a, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
    SolverState h{}; h.a1 = 1; h.a2 = 1; h.l_w = 1e4; h.l_w_prev = 1e4; h.l_h = 1e6; h.l_h_prev = 1e6; h.dsq = 6400;
    hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice); hipMemset(stamps, 0, (size_t)grid * 16 * 8 * 8);
    const int NW = (S + 63) / 64; const int waves = DMF_WAVES_PER_WG(NW);
    const size_t lds = fused_lds_bytes(S, 12, n_u, T2);
    hipFuncSetAttribute((const void*)k_rowpass_fused<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rowpass_fused<3, 4>), dim3(grid), dim3(waves * 64), lds, 0, V, D, R, a, u, up, st, N, S, n_c, T2, 0, slab, u2, stamps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("launch %d: %.3f ms  (%s)\n", rep, ms, hipGetErrorString(hipGetLastError()));
    }
    std::vector<unsigned long long> hs((size_t)grid * 16 * 8);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    const char* an[5] = {"store+prefetch+operands", "phase A (MFMA)", "wait X", "phase B / idle", "wait Y"};
    const char* cn[5] = {"rows first half", "rows second half", "wait X", "-", "wait Y"};
    for (int team = 0; team < 2; ++team) {
        double sum[8] = {0}; int cnt = 0;
        for (int b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) {
            const bool is_a = w < NW; if ((team == 0) != is_a) continue;
            for (int i = 0; i < 8; ++i) sum[i] += (double)hs[((size_t)b * 16 + w) * 8 + i]; ++cnt;
        }
        double tot = 0; for (int i = 0; i < 8; ++i) tot += sum[i];
        if (team == 0) printf("   %-26s %6.1f %%  (%.0f cycles per block-step)\n", "tile store (vmcnt waits)", 100 * sum[5] / tot, sum[5] / cnt / ((N / 16.0) / grid));
        printf("%s team: mean cycles per wave %.0f over the kernel\n", team == 0 ? "A" : "C", tot / cnt);
        for (int i = 0; i < 5; ++i) printf("   %-26s %6.1f %%  (%.0f cycles per block-step)\n", team == 0 ? an[i] : cn[i], 100 * sum[i] / tot, sum[i] / cnt / ((N / 16.0) / grid));
        if (team == 1) printf("   of the rows before X: first row %.0f, second %.0f, third %.0f cycles (the rest lands in 'rows first half')\n", sum[5] / cnt / ((N / 16.0) / grid), sum[6] / cnt / ((N / 16.0) / grid), sum[7] / cnt / ((N / 16.0) / grid));
    }
    return 0;
}
