// Diagnostic: latency of dependent FP64 VALU chains on gfx950, and of the phase-B row step (not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
template <int L>
__device__ __forceinline__ void fmac_bcast(double& acc, double x, double m) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(L));
}
template <int L>
__device__ __forceinline__ void fmac_bcast_nonop(double& acc, double x, double m) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(L));
}
__global__ void k(double* out, unsigned long long* cyc, int iters, int busy_waves) {
    const int wave = threadIdx.x >> 6;
    double x = out[threadIdx.x], y = 1.0000001, z = 0.25;
    if (wave > 0) {  // background waves: independent FP64 FMAs (like phase C)
        double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
        for (int i = 0; i < iters * 16; ++i) {
            a0 = fma(a0, y, z); a1 = fma(a1, y, z); a2 = fma(a2, y, z); a3 = fma(a3, y, z);
        }
        out[threadIdx.x] = a0 + a1 + a2 + a3;
        return;
    }
    unsigned long long t0 = stamp();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    }
    unsigned long long t1 = stamp();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(z));
    }
    unsigned long long t2 = stamp();
    double acc = x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { fmac_bcast<1>(acc, acc, y); }
    }
    unsigned long long t3 = stamp();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { fmac_bcast_nonop<1>(acc, x, y); }  // acc chain only (x not freshly written)
    }
    unsigned long long t4 = stamp();
    // the row step of phase B (NU = 4)
    double uv = x, upv = z, cj = 0.01, m0 = -0.1, m1 = -0.05, m2 = -0.02, m3 = -0.01, beta = 0.3;
    for (int i = 0; i < iters; ++i) {
        const double ut = fma(beta, uv - upv, uv);
        upv = uv;
        double r0 = ut + cj, r1 = 0.0;
        fmac_bcast<0>(r0, ut, m0); fmac_bcast<1>(r1, ut, m1); fmac_bcast<2>(r0, ut, m2); fmac_bcast<3>(r1, ut, m3);
        asm volatile("v_add_f64 %0, %1, %2 clamp" : "=v"(uv) : "v"(r0), "v"(r1));
    }
    unsigned long long t5 = stamp();
    out[threadIdx.x] = x + acc + uv + upv;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64); hipMemset(out, 0, 1024 * 8);
    const int iters = 2000;
    for (int waves : {1, 5, 9, 12}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters, waves - 1);
        unsigned long long h[5]; hipMemcpy(h, cyc, 40, hipMemcpyDeviceToHost);
        printf("waves/CU %2d (wave 0 measured; same SIMD shares with %d busy waves): dependent v_fma_f64 %.1f  v_add_f64 %.1f  fmac_dpp(self-dep, nop) %.1f  fmac_dpp(acc-dep) %.1f cycles/op;  row step %.1f cycles/iteration (s_memtime ticks at 100 MHz? see below)\n",
               waves, (waves - 1 + 3) / 4, h[0] / (8.0 * iters), h[1] / (8.0 * iters), h[2] / (8.0 * iters), h[3] / (8.0 * iters), h[4] / (double)iters);
    }
    return 0;
}
