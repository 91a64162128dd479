// Diagnostic: what the instructions of phase B's inner step cost one wave alone on its SIMD (not product code)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/step_probe.hip -o tools/step_probe && tools/step_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define REP8(X) X X X X X X X X
__global__ void k(double* out, unsigned long long* cyc, int iters) {
    double x = out[threadIdx.x], y = 1.0000001, z = 0.25, a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    unsigned long long t[12];
    int n = 0;
    t[n++] = stamp();
    for (int i = 0; i < iters; ++i) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
    t[n++] = stamp();  // 0: dependent FP64 FMA
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                     "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y), "v"(z));
    }
    t[n++] = stamp();  // 1: independent FP64 FMA (4 chains)
    int lo = __double2loint(x), hi = __double2hiint(x), l2 = lo + 1, h2 = hi + 1;
    for (int i = 0; i < iters; ++i) { REP8(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(lo));) }
    t[n++] = stamp();  // 2: dependent dpp mov
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %2, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %0, %4 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %2, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %5 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf"
                     : "=v"(lo), "=v"(hi), "=v"(l2), "=v"(h2) : "v"(lo), "v"(hi));
    }
    t[n++] = stamp();  // 3: independent dpp movs
    // 4: FP64 result -> dpp mov of its halves -> FP64 (the step's ut -> rotation -> fmac link)
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
             x = __hiloint2double(__double2hiint(x), __builtin_amdgcn_mov_dpp(__double2loint(x), 0x39, 0xF, 0xF, false));)
    }
    t[n++] = stamp();
    // 5: the step as the kernel has it
    double cur = x, prev = a0, cj = 0.01, m0 = -0.1, m1 = -0.05, m2 = -0.02, m3 = -0.01, beta = 0.3;
    for (int i = 0; i < iters; ++i) {
        double ut, acc;
        int r1l, r1h, r2l, r2h, r3l, r3h;
        asm volatile("v_add_f64 %0, %2, -%3\n\tv_fma_f64 %0, %4, %0, %2\n\tv_fma_f64 %1, %5, %0, %6" : "=&v"(ut), "=&v"(acc) : "v"(cur), "v"(prev), "v"(beta), "v"(m0), "v"(cj));
        asm volatile("v_mov_b32_dpp %0, %6 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %7 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %2, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp %4, %6 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %7 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xf"
                     : "=&v"(r1l), "=&v"(r1h), "=&v"(r2l), "=&v"(r2h), "=&v"(r3l), "=&v"(r3h) : "v"(__double2loint(ut)), "v"(__double2hiint(ut)));
        const double r1 = __hiloint2double(r1h, r1l), r2 = __hiloint2double(r2h, r2l), r3 = __hiloint2double(r3h, r3l);
        prev = cur;
        asm volatile("v_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %0, %4, %5\n\tv_fma_f64 %1, %6, %7, %0 clamp" : "+v"(acc), "=v"(cur) : "v"(m1), "v"(r1), "v"(m2), "v"(r2), "v"(m3), "v"(r3));
    }
    t[n++] = stamp();
    out[threadIdx.x] = x + a0 + a1 + a2 + a3 + lo + hi + l2 + h2 + cur + prev;
    if (threadIdx.x == 0) for (int i = 0; i + 1 < n; ++i) cyc[i] = t[i + 1] - t[i];
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 128); hipMemset(out, 0, 1024 * 8);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, iters);
        unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("s_memtime ticks per instruction (one wave alone; ticks are 100 MHz? compare the rows): dependent v_fma_f64 %.2f | independent v_fma_f64 %.2f | dependent dpp mov %.2f | independent dpp mov %.2f | fma_f64 + dpp mov of its result %.2f per pair | the 14-instruction step %.2f per step\n",
               h[0] / (8.0 * iters), h[1] / (8.0 * iters), h[2] / (8.0 * iters), h[3] / (8.0 * iters), h[4] / (8.0 * iters), h[5] / (double)iters);
    }
    return 0;
}
