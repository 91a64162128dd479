// Row-local inner iterations of the u phase (deconvolution.py:81-90) with lane = (row, unknown type): the cross-lane
// and clamp helpers shared by the fused row-pass kernels.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace dmf {

template <int CTRL>
__device__ __forceinline__ double f_dpp_quad(double x) {
    // mov_dpp, not update_dpp(0, ...): a quad permute has a source in every lane, and an "old" value would
    // cost a v_mov per half to initialise the destination (8 extra instructions per inner step at n_u = 4)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int NU, int L>
__device__ __forceinline__ double f_group_bcast(double x, int lane0) {
    if constexpr (NU == 1) return x;
    else if constexpr (NU == 2) return f_dpp_quad<(L) | (L << 2) | ((2 + L) << 4) | ((2 + L) << 6)>(x);
    else if constexpr (NU == 4) return f_dpp_quad<L | (L << 2) | (L << 4) | (L << 6)>(x);
    else return __shfl(x, lane0 + L, 64);
}

// sum_l base_l * Ms[l] over the NU lanes of a row group, as a balanced tree: phase B is one dependent
// chain on the workgroup's critical path, so its depth (not its instruction count) is what costs
template <int NU, int L0, int L1>
__device__ __forceinline__ double f_dot_tree(double base, const double (&Ms)[NU], int lane0) {
    if constexpr (L1 - L0 == 1) {
        return f_group_bcast<NU, L0>(base, lane0) * Ms[L0];
    } else {
        constexpr int MID = (L0 + L1) / 2;
        return f_dot_tree<NU, L0, MID>(base, Ms, lane0) + f_dot_tree<NU, MID, L1>(base, Ms, lane0);
    }
}

// clip(a - b, 0, 1) in one instruction: the VOP3 clamp modifier clamps an FP result to [0, 1]
// (np.clip(x, 0, 1) of deconvolution.py:88; a NaN would come out as 0 instead of NaN)
__device__ __forceinline__ double f_sub_clamp01(double a, double b) {
    double r;
    asm("v_add_f64 %0, %1, -%2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// clip(a + b, 0, 1) in one instruction (VOP3 clamp modifier; a NaN would come out as 0 instead of NaN)
__device__ __forceinline__ double f_add_clamp01(double a, double b) {
    double r;
    asm("v_add_f64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// clip(a * b + c, 0, 1) in one instruction: the VOP3 clamp modifier clamps an FP result to [0, 1]
// (np.clip(x, 0, 1) of deconvolution.py:88; a NaN would come out as 0 instead of NaN)
__device__ __forceinline__ double f_fma_clamp01(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// x of the lane R places further round this lane's row group (NU = 2 or 4 lanes, aligned to a quad)
template <int NU, int R>
__device__ __forceinline__ double f_group_rot(double x) {
    static_assert(NU == 2 || NU == 4, "row groups that tile a quad");
    if constexpr (NU == 2) return f_dpp_quad<1 | (0 << 2) | (3 << 4) | (2 << 6)>(x);
    else return f_dpp_quad<((0 + R) & 3) | (((1 + R) & 3) << 2) | (((2 + R) & 3) << 4) | (((3 + R) & 3) << 6)>(x);
}

// clip(seed + sum_l x_l * Mn[l], 0, 1) over the NU lanes of a row group as one FMA chain whose last link clamps.
// Under contention from the C waves of its SIMD the phase-B wave pays ~10 cycles per instruction issued,
// dependent or not, so the instruction count (NU FMAs here against NU multiplies + NU adds for a balanced tree
// and a separate clamp) matters more than the depth of the chain.
template <int NU, int L = 0>
__device__ __forceinline__ double f_step_chain(double acc, double x, const double (&Mn)[NU], int lane0) {
    if constexpr (L == NU - 1) {
        return f_fma_clamp01(f_group_bcast<NU, L>(x, lane0), Mn[L], acc);
    } else {
        return f_step_chain<NU, L + 1>(fma(f_group_bcast<NU, L>(x, lane0), Mn[L], acc), x, Mn, lane0);
    }
}

}  // namespace dmf
