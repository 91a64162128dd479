// Do FP64 MFMA and FP64 VALU FMA overlap on one SIMD of gfx950?  (diagnostic tool, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
// mode bit0: waves 0-3 run MFMA; bit1: waves 4-7 (or 4-11) run VALU FMA chains
__global__ __launch_bounds__(768) void k(double* out, int iters, int mode, int f32valu) {
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3, s = 0;
    if (wave < 4) {
        if (mode & 1) {
            v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            }
            s = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else if (mode & 2) {
        if (f32valu) {
            float f[16]; for (int j = 0; j < 16; ++j) f[j] = j; float fa = (float)a, fb = (float)b;
            for (int i = 0; i < iters * 16; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) f[j] = fmaf(fa, fb, f[j]);
            for (int j = 0; j < 16; ++j) s += f[j];
        } else {
            double f[16]; for (int j = 0; j < 16; ++j) f[j] = j;
            for (int i = 0; i < iters * 4; ++i)      // 64 FMAs per outer iteration ~ 4 MFMAs' worth of cycles at 4 cyc
#pragma unroll
                for (int j = 0; j < 16; ++j) f[j] = fma(a, b, f[j]);
            for (int j = 0; j < 16; ++j) s += f[j];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* out; hipMalloc(&out, 1024 * 768 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    struct { int threads, mode, f32; const char* name; } cfg[] = {
        {256, 1, 0, "MFMA f64 only (4 waves, 1/SIMD)"}, {512, 2, 0, "VALU f64 only (4 waves, 1/SIMD)"},
        {768, 2, 0, "VALU f64 only (8 waves, 2/SIMD)"}, {512, 3, 0, "MFMA + VALU f64 (1+1 per SIMD)"},
        {768, 3, 0, "MFMA + VALU f64 (1+2 per SIMD)"}, {512, 2, 1, "VALU f32 only (4 waves)"},
        {512, 3, 1, "MFMA f64 + VALU f32 (1+1 per SIMD)"}};
    for (auto& c : cfg) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k<<<256, c.threads>>>(out, iters, c.mode, c.f32);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-40s %.3f ms\n", c.name, best);
    }
    return 0;
}
