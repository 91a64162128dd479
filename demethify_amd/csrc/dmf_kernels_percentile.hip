// Percentiles over the replicate axis of a bootstrap stack (demethify/bootstrap.py:51-54 and :75-78:
// np.percentile(stack, q, axis=0) with numpy's default "linear" method), two percentiles per pass.
//
// x is [n replicates][m positions], replicate-major, so that for a fixed replicate consecutive positions are
// consecutive in memory: one thread per position streams its n values with coalesced loads.
//   k_percentile_tails  the usual confidence-interval case: both order statistics sit within KMAX of an end of
//                       the sorted column, so a thread keeps the KMAX smallest and KMAX largest values it has
//                       seen in registers (sorted insertion, fully unrolled) -- one HBM pass, no sort.
//   k_percentile_rank   any percentile: a workgroup loads a [n][P] tile into LDS and ranks every element by
//                       counting (ties broken by replicate index, so each rank is taken exactly once).
// The interpolation between the two neighbouring order statistics repeats numpy's _lerp operation by
// operation (no fused multiply-add), so the results are bit-identical to numpy's.
#include "dmf_internal.h"

namespace dmf {

// numpy/lib/_function_base_impl.py, _lerp(a, b, t): a + (b - a) t, or b - (b - a)(1 - t) once t >= 0.5
// (contraction off: numpy rounds the product before the addition; HIP's __dmul_rn is a plain `*` and would fuse)
__device__ __forceinline__ double np_lerp(double a, double b, double t) {
#pragma clang fp contract(off)
    const double diff = b - a;
    if (t >= 0.5) {
        const double prod = diff * (1.0 - t);
        return b - prod;
    }
    const double prod = diff * t;
    return a + prod;
}

template <int KMAX>
__device__ __forceinline__ double pick(const double (&buf)[KMAX], int idx) {
    double r = buf[0];
#pragma unroll
    for (int t = 1; t < KMAX; ++t) r = idx == t ? buf[t] : r;  // static indexing: buf stays in registers
    return r;
}

template <int KMAX>
__global__ __launch_bounds__(256) void k_percentile_tails(const double* __restrict__ x, int64_t n, int64_t m,
                                                          PercentilePlan p0, PercentilePlan p1,
                                                          double* __restrict__ out0, double* __restrict__ out1) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= m) return;
    double small[KMAX], large[KMAX];  // small ascending (the KMAX smallest), large descending (the KMAX largest)
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
        small[t] = __builtin_huge_val();
        large[t] = -__builtin_huge_val();
    }
    const double* __restrict__ col = x + p;
    auto take = [&](double v) {
        if (v < small[KMAX - 1]) {
            double w = v;
#pragma unroll
            for (int t = 0; t < KMAX; ++t) {
                const double s = small[t];
                const bool lt = w < s;
                small[t] = lt ? w : s;
                w = lt ? s : w;
            }
        }
        if (v > large[KMAX - 1]) {
            double w = v;
#pragma unroll
            for (int t = 0; t < KMAX; ++t) {
                const double s = large[t];
                const bool gt = w > s;
                large[t] = gt ? w : s;
                w = gt ? s : w;
            }
        }
    };
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {  // four loads in flight per thread
        const double v0 = col[i * m], v1 = col[(i + 1) * m], v2 = col[(i + 2) * m], v3 = col[(i + 3) * m];
        take(v0);
        take(v1);
        take(v2);
        take(v3);
    }
    for (; i < n; ++i) take(col[i * m]);
    // sorted index k counted from the top is n - 1 - k
    const double a0 = p0.from_top ? pick<KMAX>(large, (int)(n - 1 - p0.k_prev)) : pick<KMAX>(small, (int)p0.k_prev);
    const double b0 = p0.from_top ? pick<KMAX>(large, (int)(n - 1 - p0.k_next)) : pick<KMAX>(small, (int)p0.k_next);
    out0[p] = np_lerp(a0, b0, p0.gamma);
    if (out1 != nullptr) {
        const double a1 = p1.from_top ? pick<KMAX>(large, (int)(n - 1 - p1.k_prev)) : pick<KMAX>(small, (int)p1.k_prev);
        const double b1 = p1.from_top ? pick<KMAX>(large, (int)(n - 1 - p1.k_next)) : pick<KMAX>(small, (int)p1.k_next);
        out1[p] = np_lerp(a1, b1, p1.gamma);
    }
}

// P positions per workgroup (a power of two <= 8), tile[n][P] in dynamic LDS
__global__ __launch_bounds__(256) void k_percentile_rank(const double* __restrict__ x, int64_t n, int64_t m, int P,
                                                         PercentilePlan p0, PercentilePlan p1,
                                                         double* __restrict__ out0, double* __restrict__ out1) {
    extern __shared__ double tile[];
    __shared__ double sel[8][4];
    const int64_t pos0 = (int64_t)blockIdx.x * P;
    for (int64_t idx = threadIdx.x; idx < n * P; idx += 256) {
        const int64_t i = idx / P;
        const int pp = (int)(idx - i * P);
        tile[idx] = pos0 + pp < m ? x[i * m + pos0 + pp] : 0.0;
    }
    __syncthreads();
    const int pp = threadIdx.x % P, li = threadIdx.x / P, nl = 256 / P;
    for (int64_t i = li; i < n; i += nl) {
        const double a = tile[i * P + pp];
        int64_t rank = 0;
        for (int64_t j = 0; j < n; ++j) {
            const double b = tile[j * P + pp];
            rank += (b < a || (b == a && j < i)) ? 1 : 0;
        }
        if (rank == p0.k_prev) sel[pp][0] = a;
        if (rank == p0.k_next) sel[pp][1] = a;
        if (rank == p1.k_prev) sel[pp][2] = a;
        if (rank == p1.k_next) sel[pp][3] = a;
    }
    __syncthreads();
    if (threadIdx.x < P && pos0 + threadIdx.x < m) {
        out0[pos0 + threadIdx.x] = np_lerp(sel[threadIdx.x][0], sel[threadIdx.x][1], p0.gamma);
        if (out1 != nullptr) out1[pos0 + threadIdx.x] = np_lerp(sel[threadIdx.x][2], sel[threadIdx.x][3], p1.gamma);
    }
}

constexpr int kTailK = 32;  // the deepest register tail the tails kernel is built with

// how many values from an end of the sorted column the plan needs (0: more than kTailK, not a tail case)
static int plan_tail_depth(PercentilePlan& pl, int64_t n) {
    if (pl.k_next <= kTailK - 1) {
        pl.from_top = 0;
        return (int)pl.k_next + 1;
    }
    if (n - 1 - pl.k_prev <= kTailK - 1) {
        pl.from_top = 1;
        return (int)(n - pl.k_prev);
    }
    return 0;
}

int64_t percentile_max_replicates() { return (152 * 1024) / (int64_t)sizeof(double); }

hipError_t launch_percentile_pair(const double* x, int64_t n, int64_t m, PercentilePlan p0, PercentilePlan p1,
                                  double* out0, double* out1, hipStream_t st) {
    if (n < 1 || m < 1) return hipErrorInvalidValue;
    const int d0 = plan_tail_depth(p0, n), d1 = plan_tail_depth(p1, n);
    if (d0 > 0 && d1 > 0) {
        const int64_t grid = (m + 255) / 256;
        if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
        const int depth = d0 > d1 ? d0 : d1;  // the insertion cost grows with the tail depth: use the smallest build
        if (depth <= 8)
            hipLaunchKernelGGL(k_percentile_tails<8>, dim3((unsigned)grid), dim3(256), 0, st, x, n, m, p0, p1, out0, out1);
        else if (depth <= 16)
            hipLaunchKernelGGL(k_percentile_tails<16>, dim3((unsigned)grid), dim3(256), 0, st, x, n, m, p0, p1, out0, out1);
        else
            hipLaunchKernelGGL(k_percentile_tails<kTailK>, dim3((unsigned)grid), dim3(256), 0, st, x, n, m, p0, p1, out0,
                               out1);
        return hipGetLastError();
    }
    if (n > percentile_max_replicates()) return hipErrorInvalidValue;
    int P = 8;
    while (P > 1 && (size_t)n * P * sizeof(double) > 152 * 1024) P >>= 1;
    const size_t lds = (size_t)n * P * sizeof(double);
    static bool raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > 48 * 1024 && !raised[dev]) {
        // (the kernel also has 256 B of static LDS: dynamic + static must stay within the CU's 160 KB)
        hipError_t e = hipFuncSetAttribute((const void*)k_percentile_rank, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           152 * 1024);
        if (e != hipSuccess) return e;
        raised[dev] = true;
    }
    const int64_t grid = (m + P - 1) / P;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_percentile_rank, dim3((unsigned)grid), dim3(256), lds, st, x, n, m, P, p0, p1, out0, out1);
    return hipGetLastError();
}

}  // namespace dmf
