// Exact fixed-point digits for the integer matrix cores: a product z = xa * xb in [0, 1] as seven balanced base-256
// digits of rint(z 2^52).  Shared by dmf_kernels_gram_i8.hip (row features) and dmf_kernels_rowpass2.hip (the
// alpha_j alpha_l operand of the row pass's M product).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace dmf {

// Feature value z = xa * xb in [0, 1] -> the 64-bit integer rint(z 2^52) + bias: y = fma(xa, xb, 1.0) lies in [1, 2],
// so its bit pattern minus that of 1.0 IS rint(xa xb 2^52) (one rounding, of the exact product, by the FMA itself).
// bias = 0x0000808080808080: bytes 0..5 of the sum, XOR 0x80, are the balanced digits a_0..a_5, byte 6 is a_6
// (0 <= a_6 <= 17).
__device__ __forceinline__ void z_to_biased(double xa, double xb, unsigned int& lo, unsigned int& hi) {
    const double y = fma(xa, xb, 1.0);
    // one 64-bit add: bits(y) - bits(1.0) + bias
    const unsigned long long s = (unsigned long long)__double_as_longlong(y) + (0x0000808080808080ull - 0x3FF0000000000000ull);
    lo = (unsigned int)s;
    hi = (unsigned int)(s >> 32);
}

// 4 x 4 byte transpose: digits t = 0..3 of four 32-bit words w[0..3] (byte t of word r -> byte r of out[t])
__device__ __forceinline__ void transpose4(const unsigned int (&w)[4], unsigned int (&out)[4]) {
    const unsigned int a01 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), b01 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
    const unsigned int a23 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), b23 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
    out[0] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);
    out[1] = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
    out[2] = __builtin_amdgcn_perm(b23, b01, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
}

}  // namespace dmf
