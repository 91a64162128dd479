// Probe of the f64 MFMA operand/result layouts and issue rates on gfx950 (diagnostic tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A, const double* B, double* C) {
    // A: 16x4 row-major, B: 4x16 row-major; hypothesis: a = A[l&15][l>>4], b = B[l>>4][l&15],
    // result reg r -> C[(l>>4) + 4r][l&15]
    const int l = threadIdx.x;
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[l * 4 + r] = c[r];
}

__global__ void k_layout44(const double* A, const double* B, double* C) {
    // 4x4x4 with 4 blocks: dump raw per-lane result for host-side layout search
    const int l = threadIdx.x;
    double c = 0.0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], c, 0, 0, 0);
    C[l] = c;
}

template <int MODE>
__global__ void k_rate(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double f0 = a, f1 = b, f2 = a + b, f3 = a - b, f4 = 1, f5 = 2, f6 = 3, f7 = 4;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else if (MODE == 1) {
            f0 = fma(a, b, f0); f1 = fma(a, b, f1); f2 = fma(a, b, f2); f3 = fma(a, b, f3);
            f4 = fma(a, b, f4); f5 = fma(a, b, f5); f6 = fma(a, b, f6); f7 = fma(a, b, f7);
        } else {
            f0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f1, 0, 0, 0);
            f2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f2, 0, 0, 0);
            f3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f3, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    std::vector<double> A(64), B(64), C(256), R(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 0.37 + k * 1.91;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 - k * 0.53 + j * 0.71 + (k * j) * 0.01;
    double *dA, *dB, *dC; long long* dcyc;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 65536 * 8 * 4); hipMalloc(&dcyc, 8);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) + 4 * r, col = l & 15; double ref = 0;
        for (int k = 0; k < 4; ++k) ref += A[row * 4 + k] * B[k * 16 + col];
        double e = fabs(ref - C[l * 4 + r]); if (e > maxerr) maxerr = e;
    }
    printf("16x16x4 f64 layout hypothesis max err = %g\n", maxerr);
    // 4x4x4: random operands, then search the layout on the host
    std::vector<double> a4(64), b4(64), c4(64);
    for (int l = 0; l < 64; ++l) { a4[l] = 1 + 0.013 * l * l + 0.7 * l; b4[l] = 3 - 0.011 * l * l + 0.3 * l; }
    hipMemcpy(dA, a4.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, b4.data(), 512, hipMemcpyHostToDevice);
    k_layout44<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(c4.data(), dC, 512, hipMemcpyDeviceToHost);
    // hypothesis family: block = l>>4?  A lane l: A[b][i=l&3][k=(l>>2)&3]; B lane l: B[b][k=(l>>2)&3][j=l&3]; D lane l: D[b][i=(l>>2)&3][j=l&3]
    for (int hyp = 0; hyp < 4; ++hyp) {
        double me = 0;
        for (int l = 0; l < 64; ++l) {
            int b = l >> 4, x = l & 3, y = (l >> 2) & 3; double ref = 0;
            for (int k = 0; k < 4; ++k) {
                // lane index holding A[b][i][k] / B[b][k][j] under each hypothesis
                int i = (hyp & 1) ? x : y, j = (hyp & 1) ? y : x;
                int la = (hyp & 2) ? (b * 16 + k * 4 + i) : (b * 16 + i * 4 + k);   // A[b][i][k] at lane
                int lb = (hyp & 2) ? (b * 16 + k * 4 + j) : (b * 16 + j * 4 + k);
                ref += a4[la] * b4[lb];
            }
            me = fmax(me, fabs(ref - c4[l]));
        }
        printf("4x4x4 hypothesis %d max err = %g\n", hyp, me);
    }
    {
        double me = 0;
        for (int l = 0; l < 64; ++l) {
            int b = (l >> 2) & 3, j = l & 3, i = l >> 4; double ref = 0;
            for (int k = 0; k < 4; ++k) ref += a4[(k << 4) + (b << 2) + i] * b4[(k << 4) + (b << 2) + j];
            me = fmax(me, fabs(ref - c4[l]));
        }
        printf("4x4x4 hypothesis 16x16-like (i=l>>4 out, blk=(l>>2)&3, A lane=(k<<4)+(b<<2)+i) max err = %g\n", me);
    }
    for (int l = 0; l < 64; l += 1) if (l < 8 || (l % 16) == 0) printf("c4[%d]=%.6f\n", l, c4[l]);
    // rates: one wave per SIMD (256 threads, 1 block) and chip-wide
    for (int mode = 0; mode < 3; ++mode) {
        const int iters = 20000; long long cyc = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k_rate<0><<<1024, 256>>>(dC, dcyc, iters);
            else if (mode == 1) k_rate<1><<<1024, 256>>>(dC, dcyc, iters);
            else k_rate<2><<<1024, 256>>>(dC, dcyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
        double fma_per_instr = mode == 0 ? 1024.0 : (mode == 1 ? 64.0 : 256.0);
        int per_iter = mode == 1 ? 8 : 4;
        double total_fma = 1024.0 * 4 * iters * per_iter * fma_per_instr;   // 4 waves per block
        printf("mode %d: %.3f ms, %.1f TFLOP/s f64, block0 cycles/instr = %.2f\n", mode, ms, 2 * total_fma / ms / 1e9,
               (double)cyc / (iters * per_iter));
    }
    return 0;
}
