// Diagnostic: do 16-byte buffer / global loads from 8-byte-aligned addresses return the right data on gfx950?
// (odd sample counts make every other row of V start 8 bytes off a 16-byte boundary)   not product code
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/align_probe.hip -o tools/align_probe && tools/align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ void k(const double* __restrict__ src, int n, double* __restrict__ out_buf, double* __restrict__ out_glb, int shift) {
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(n * 8), 0x00020000);
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)((2 * lane + shift) * 8), 0, 0);
    out_buf[2 * lane] = __hiloint2double((int)a.y, (int)a.x);
    out_buf[2 * lane + 1] = __hiloint2double((int)a.w, (int)a.z);
    const double* p = src + 2 * lane + shift;
    v2d g;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(p) : "memory");
    out_glb[2 * lane] = g.x;
    out_glb[2 * lane + 1] = g.y;
}
int main() {
    const int n = 200;
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000.0 + i;
    double *d, *ob, *og;
    hipMalloc(&d, n * 8); hipMalloc(&ob, 128 * 8); hipMalloc(&og, 128 * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 2; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, ob, og, shift);
        std::vector<double> rb(128), rg(128);
        hipMemcpy(rb.data(), ob, 128 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(rg.data(), og, 128 * 8, hipMemcpyDeviceToHost);
        int bad_b = 0, bad_g = 0;
        for (int i = 0; i < 128; ++i) {
            bad_b += rb[i] != 1000.0 + i + shift;
            bad_g += rg[i] != 1000.0 + i + shift;
        }
        printf("shift %d (byte offset %d mod 16): buffer_load_b128 wrong %d / 128, global_load_dwordx4 wrong %d / 128 (%s)\n", shift,
               8 * shift, bad_b, bad_g, hipGetErrorString(hipGetLastError()));
    }
    // the range check of a raw buffer is per dword?  last lane pair straddling the end of the buffer
    {
        const int m = 127;  // 127 doubles: lane 63 reads elements 126 (in) and 127 (out)
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, m, ob, og, 0);
        std::vector<double> rb(128);
        hipMemcpy(rb.data(), ob, 128 * 8, hipMemcpyDeviceToHost);
        printf("straddling the end: element 126 = %.1f (want 1126.0), element 127 = %.1f (want 0.0)\n", rb[126], rb[127]);
    }
    return 0;
}
