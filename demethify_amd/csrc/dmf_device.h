// Device-side helpers: wave64 / block reductions used by every kernel file.
#pragma once
#include <hip/hip_runtime.h>

namespace dmf {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Sum over a block of NT threads (NT multiple of 64, <= 1024); result valid in thread 0.
// `red` is LDS scratch of >= NT/64 doubles. Fixed summation order -> deterministic.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot += red[w];
    }
    return tot;
}

template <int NT>
__device__ __forceinline__ double block_max(double v, double* red) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double tot = red[0];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) tot = fmax(tot, red[w]);
    }
    return tot;
}

// Momentum recurrence shared by both phases (demethify/deconvolution.py:83-85, :95-97).
__device__ __forceinline__ void momentum_step(double& a, double l_prev, double l_cur, double& beta) {
    const double a0 = a;
    a = (1.0 + sqrt(1.0 + 4.0 * a0 * a0)) / 2.0;
    beta = fmin((a0 - 1.0) / a, 0.9999 * sqrt(l_prev / l_cur));
}

}  // namespace dmf
