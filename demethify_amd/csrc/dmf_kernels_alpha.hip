// alpha phase on the packed per-sample Gram buffer, simplex projection, cost and the scalar
// bookkeeping of one outer iteration (demethify/deconvolution.py:93-102, :21-37, :212-221).
//
// gb[(K+1)(K+2)/2][S] is the packed upper triangle (tri(k,l), k <= l <= K) of the extended Gram
// matrix of x = (R, v) weighted by d: rows tri(k,l<K) are G_s = R^T diag(d_s) R, rows tri(k,K)
// are b_s = R^T (d_s * v_s), row tri(K,K) is v_s^T D_s v_s.  With them
//     R^T (d_s * (v_s - R a)) = b_s - G_s a,      cost_s = vDv_s - 2 a.b_s + a^T G_s a.
#include "dmf_device.h"
#include "dmf_internal.h"

namespace dmf {

// Compile-time bitonic network, descending; every index is a constant after unrolling so the
// array stays in registers.
template <int KMAX>
__device__ __forceinline__ void sort_desc(double (&x)[KMAX]) {
#pragma unroll
    for (int k = 2; k <= KMAX; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < KMAX; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const double lo = fmin(x[i], x[l]), hi = fmax(x[i], x[l]);
                    if ((i & k) == 0) { x[i] = hi; x[l] = lo; }
                    else { x[i] = lo; x[l] = hi; }
                }
            }
        }
    }
}

// projection_simplex_sort_2d for one column held in registers (deconvolution.py:25-35).
template <int KMAX>
__device__ __forceinline__ void project_column(double (&x)[KMAX], int K, double z) {
    double srt[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) srt[k] = k < K ? x[k] : -INFINITY;
    sort_desc<KMAX>(srt);
    // rho = last j with srt_j - (cumsum_j - z) / (j + 1) > 0, tested as srt_j (j + 1) - (cumsum_j - z) > 0
    // (same sign, no division in the scan); theta = (cumsum_rho - z) / (rho + 1) with ONE true division.
    double run = 0.0, theta_num = 0.0, theta_den = 0.0;
    double last_shift = 0.0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        if (j < K) {
            run += srt[j];
            const double shifted = run - z;
            if (fma(srt[j], (double)(j + 1), -shifted) > 0.0) {
                theta_num = shifted;
                theta_den = (double)(j + 1);
            }
            last_shift = shifted;
        }
    }
    // no j qualified (non-finite input only): upstream takes rho = -1 -> pi[-1] / 0
    const double theta = theta_den > 0.0 ? theta_num / theta_den : last_shift / 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) x[k] = k < K ? fmax(x[k] - theta, 0.0) : 0.0;
}

template <int KMAX>
__global__ __launch_bounds__(64) void k_alpha_phase(const double* __restrict__ gb,
                                                    double* __restrict__ alpha,
                                                    double* __restrict__ alpha_prev,
                                                    const SolverState* __restrict__ state, int S,
                                                    int K, int n_u, int n_iter2, int gb_in_lds,
                                                    double* __restrict__ partials) {
    extern __shared__ double lds_dyn[];
    if (state->done) return;
    const int tid = threadIdx.x;
    const int s = blockIdx.x * 64 + tid;
    const bool active = s < S;
    const int sc = active ? s : S - 1;
    const int nrows = (K + 1) * (K + 2) / 2;
    if (gb_in_lds) {
        for (int r = 0; r < nrows; ++r) lds_dyn[r * 64 + tid] = gb[(int64_t)r * S + sc];
    }
    const double* G = gb_in_lds ? (const double*)lds_dyn + tid : gb + sc;
    const int64_t gstride = gb_in_lds ? 64 : S;

    double a[KMAX], ap[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        a[k] = k < K ? alpha[(int64_t)k * S + sc] : 0.0;
        ap[k] = k < K ? alpha_prev[(int64_t)k * S + sc] : 0.0;
    }
    double a2 = state->a2, lh_prev = state->l_h_prev;
    const double lh = state->l_h;
    for (int t = 0; t < n_iter2; ++t) {
        double beta;
        momentum_step(a2, lh_prev, lh, beta);
        double at[KMAX], g[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            at[k] = a[k] + beta * (a[k] - ap[k]);
            ap[k] = a[k];
            g[k] = k < K ? G[tri(k, K) * gstride] : 0.0;
        }
#pragma unroll
        for (int l = 0; l < KMAX; ++l) {
            if (l < K) {
#pragma unroll
                for (int k = 0; k <= l; ++k) {
                    const double gkl = G[tri(k, l) * gstride];
                    g[k] = fma(-gkl, at[l], g[k]);
                    if (k != l) g[l] = fma(-gkl, at[k], g[l]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) a[k] = at[k] + g[k] / lh;
        project_column<KMAX>(a, K, 1.0);
        lh_prev = lh;
    }
    double cost = 0.0, n2 = 0.0;
    if (active) {
        cost = G[tri(K, K) * gstride];
        double lin = 0.0, quad = 0.0;
#pragma unroll
        for (int l = 0; l < KMAX; ++l) {
            if (l < K) {
                alpha[(int64_t)l * S + s] = a[l];
                alpha_prev[(int64_t)l * S + s] = ap[l];
                lin = fma(a[l], G[tri(l, K) * gstride], lin);
                double off = 0.0;
#pragma unroll
                for (int k = 0; k < l; ++k) off = fma(G[tri(k, l) * gstride], a[k], off);
                quad = fma(a[l], fma(2.0, off, G[tri(l, l) * gstride] * a[l]), quad);
                if (l >= K - n_u) n2 = fma(a[l], a[l], n2);
            }
        }
        cost = cost - 2.0 * lin + quad;
    }
    cost = wave_sum(cost);
    n2 = wave_sum(n2);
    if (tid == 0) {
        partials[2 * blockIdx.x] = cost;
        partials[2 * blockIdx.x + 1] = n2;
    }
}

// ---- runtime-K variants (16 < K <= 64): per-thread arrays live in scratch, loops are not unrolled.
// Rare path (model-selection sweeps with many unknown types); kept small to keep the build short.
__device__ __noinline__ void project_column_dyn(double* x, int K, double z) {
    double srt[kMaxK];
    for (int k = 0; k < K; ++k) {  // insertion sort, descending
        const double v = x[k];
        int j = k;
        while (j > 0 && srt[j - 1] < v) {
            srt[j] = srt[j - 1];
            --j;
        }
        srt[j] = v;
    }
    double run = 0.0, theta = 0.0, last_shift = 0.0;
    bool any = false;
    for (int j = 0; j < K; ++j) {
        run += srt[j];
        const double shifted = run - z;
        if (srt[j] - shifted / (double)(j + 1) > 0.0) {
            theta = shifted / (double)(j + 1);
            any = true;
        }
        last_shift = shifted;
    }
    if (!any) theta = last_shift / 0.0;
    for (int k = 0; k < K; ++k) x[k] = fmax(x[k] - theta, 0.0);
}

__global__ __launch_bounds__(64) void k_alpha_phase_dyn(const double* __restrict__ gb,
                                                        double* __restrict__ alpha,
                                                        double* __restrict__ alpha_prev,
                                                        const SolverState* __restrict__ state, int S,
                                                        int K, int n_u, int n_iter2,
                                                        double* __restrict__ partials) {
    if (state->done) return;
    const int tid = threadIdx.x;
    const int s = blockIdx.x * 64 + tid;
    const bool active = s < S;
    const int sc = active ? s : S - 1;
    const double* G = gb + sc;
    const int64_t gstride = S;
    double a[kMaxK], ap[kMaxK], at[kMaxK], g[kMaxK];
    for (int k = 0; k < K; ++k) {
        a[k] = alpha[(int64_t)k * S + sc];
        ap[k] = alpha_prev[(int64_t)k * S + sc];
    }
    double a2 = state->a2, lh_prev = state->l_h_prev;
    const double lh = state->l_h;
    for (int t = 0; t < n_iter2; ++t) {
        double beta;
        momentum_step(a2, lh_prev, lh, beta);
        for (int k = 0; k < K; ++k) {
            at[k] = a[k] + beta * (a[k] - ap[k]);
            ap[k] = a[k];
            g[k] = G[tri(k, K) * gstride];
        }
        for (int l = 0; l < K; ++l)
            for (int k = 0; k <= l; ++k) {
                const double gkl = G[tri(k, l) * gstride];
                g[k] = fma(-gkl, at[l], g[k]);
                if (k != l) g[l] = fma(-gkl, at[k], g[l]);
            }
        for (int k = 0; k < K; ++k) a[k] = at[k] + g[k] / lh;
        project_column_dyn(a, K, 1.0);
        lh_prev = lh;
    }
    double cost = 0.0, n2 = 0.0;
    if (active) {
        cost = G[tri(K, K) * gstride];
        double lin = 0.0, quad = 0.0;
        for (int l = 0; l < K; ++l) {
            alpha[(int64_t)l * S + s] = a[l];
            alpha_prev[(int64_t)l * S + s] = ap[l];
            lin = fma(a[l], G[tri(l, K) * gstride], lin);
            double off = 0.0;
            for (int k = 0; k < l; ++k) off = fma(G[tri(k, l) * gstride], a[k], off);
            quad = fma(a[l], fma(2.0, off, G[tri(l, l) * gstride] * a[l]), quad);
            if (l >= K - n_u) n2 = fma(a[l], a[l], n2);
        }
        cost = cost - 2.0 * lin + quad;
    }
    cost = wave_sum(cost);
    n2 = wave_sum(n2);
    if (tid == 0) {
        partials[2 * blockIdx.x] = cost;
        partials[2 * blockIdx.x + 1] = n2;
    }
}

__global__ __launch_bounds__(64) void k_project_dyn(const double* __restrict__ X, double* __restrict__ out,
                                                    int K, int S, double z) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= S) return;
    double x[kMaxK];
    for (int k = 0; k < K; ++k) x[k] = X[(int64_t)k * S + s];
    project_column_dyn(x, K, z);
    for (int k = 0; k < K; ++k) out[(int64_t)k * S + s] = x[k];
}

// Closes one outer iteration: sums the per-block partials and advances the scalar state
// (deconvolution.py:207, :216-221; a1/a2 advance exactly as the inner loops advanced them).  One wave.
// COHERENT: the partials were handed over by other workgroups of the SAME launch as atomic exchanges (the caller is
// the last workgroup to arrive): collected with atomics too.
template <bool COHERENT>
__device__ __forceinline__ void finish_iteration_body(const double* __restrict__ partials, int nb, SolverState* __restrict__ state,
                                                      int n_iter2) {
    double cost = 0.0, n2 = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) {
        if (COHERENT) {
            unsigned long long* pp = reinterpret_cast<unsigned long long*>(const_cast<double*>(partials)) + 2 * b;
            cost += __longlong_as_double((long long)atomicOr(pp, 0ull));
            n2 += __longlong_as_double((long long)atomicOr(pp + 1, 0ull));
        } else {
            cost += partials[2 * b];
            n2 += partials[2 * b + 1];
        }
    }
    cost = wave_sum(cost);
    n2 = wave_sum(n2);
    if (threadIdx.x == 0) {
        double a1 = state->a1, a2 = state->a2;
        if (const double* __restrict__ row = momentum_row(state, n_iter2)) {  // (run ahead by the host: SolverState)
            a1 = row[0];
            a2 = row[1];
            state->mom_i += 1;
        } else {
            for (int t = 0; t < n_iter2; ++t) {
                a1 = (1.0 + sqrt(1.0 + 4.0 * a1 * a1)) / 2.0;
                a2 = (1.0 + sqrt(1.0 + 4.0 * a2 * a2)) / 2.0;
            }
        }
        state->a1 = a1;
        state->a2 = a2;
        if (n_iter2 > 0) {
            state->l_w_prev = state->l_w;
            state->l_h_prev = state->l_h;
        }
        state->l_w = n2 * state->dsq;
        const double cf_prev = state->cf;
        state->cf_prev = cf_prev;
        state->cf = cost;
        state->iters += 1;
        state->arrive = 0;
        // deconvolution.py:220.  (A NaN cf_0 -- no cost before the loop was asked for -- compares false.)
        if (fabs(cost - cf_prev) < state->tol * state->band) state->done = state->band > 1.0 ? 2 : 1;
    }
}

__global__ __launch_bounds__(64) void k_finish_iteration(const double* __restrict__ partials, int nb,
                                                         SolverState* __restrict__ state,
                                                         int n_iter2) {
    if (state->done) return;
    finish_iteration_body<false>(partials, nb, state, n_iter2);
}

__global__ void k_set_lh(SolverState* state) {
    if (state->done) return;
    state->l_h = (state->rt_norm2 + state->u_norm2) * state->dsq;
}

hipError_t launch_set_lh(SolverState* state, hipStream_t st) {
    hipLaunchKernelGGL(k_set_lh, dim3(1), dim3(1), 0, st, state);
    return hipGetLastError();
}

template <int KMAX>
static hipError_t launch_alpha_t(const double* gb, double* alpha, double* alpha_prev,
                                 SolverState* state, int S, int K, int n_u, int n_iter2,
                                 double* partials, hipStream_t st) {
    const int nb = (S + 63) / 64;
    const size_t lds = (size_t)(K + 1) * (K + 2) / 2 * 64 * sizeof(double);
    const int in_lds = lds <= 150 * 1024;
    if (in_lds && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_alpha_phase<KMAX>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_alpha_phase<KMAX>, dim3(nb), dim3(64), in_lds ? lds : 0, st, gb, alpha,
                       alpha_prev, state, S, K, n_u, n_iter2, in_lds, partials);
    hipLaunchKernelGGL(k_finish_iteration, dim3(1), dim3(64), 0, st, partials, nb, state, n_iter2);
    return hipGetLastError();
}

// ---- lane-parallel alpha phase: G lanes per sample column (G = 4, 8, 16, 32, 64 >= K) --------------
// Lane k of a group owns row k of the sample's packed Gram matrix (registers), the extrapolated point
// is exchanged with group broadcasts, the simplex projection sorts across the group's lanes (bitonic
// network) and takes a parallel prefix sum.  ~10x shorter critical path than one thread per sample:
// this kernel sits between two row passes of every outer iteration.
template <int G>
__device__ __forceinline__ double group_get(double x, int src_lane) {
    return __shfl(x, src_lane, 64);
}

template <int G>
__global__ __launch_bounds__(64) void k_alpha_phase_lanes(const double* __restrict__ gb,
                                                          double* __restrict__ alpha,
                                                          double* __restrict__ alpha_prev,
                                                          const SolverState* __restrict__ state, int S,
                                                          int K, int n_u, int n_iter2,
                                                          double* __restrict__ partials) {
    if (state->done) return;
    constexpr int CPW = 64 / G;  // sample columns per wave
    const int lane = threadIdx.x;
    const int k = lane % G, grp = lane / G, base = grp * G;
    const int s = blockIdx.x * CPW + grp;
    const bool col_ok = s < S;
    const int sc = col_ok ? s : S - 1;
    const bool row_ok = k < K;
    const int kc = row_ok ? k : K - 1;

    double Grow[G];
#pragma unroll
    for (int l = 0; l < G; ++l) {
        const int lc = l < K ? l : K - 1;
        const int lo = kc < lc ? kc : lc, hi = kc < lc ? lc : kc;
        const double v = gb[(int64_t)tri(lo, hi) * S + sc];
        Grow[l] = (row_ok && l < K) ? v : 0.0;
    }
    const double bk = row_ok ? gb[(int64_t)tri(kc, K) * S + sc] : 0.0;
    double a = row_ok ? alpha[(int64_t)kc * S + sc] : 0.0;
    double ap = row_ok ? alpha_prev[(int64_t)kc * S + sc] : 0.0;

    double a2 = state->a2, lh_prev = state->l_h_prev;
    const double lh = state->l_h;
    const double rank1 = (double)(k + 1);
    for (int t = 0; t < n_iter2; ++t) {
        double beta;
        momentum_step(a2, lh_prev, lh, beta);
        const double at = a + beta * (a - ap);
        ap = a;
        double g = bk;
#pragma unroll
        for (int l = 0; l < G; ++l) g = fma(-Grow[l], group_get<G>(at, base + l), g);
        const double x = at + g / lh;  // deconvolution.py:100: alpha_temp + (...) / l_h
        // ---- projection onto the simplex (deconvolution.py:25-35), sorted copy across the lanes
        double srt = row_ok ? x : -INFINITY;
#pragma unroll
        for (int k2 = 2; k2 <= G; k2 <<= 1) {
#pragma unroll
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                const double other = __shfl_xor(srt, j, 64);
                const bool lower = (k & j) == 0;
                const bool desc = (k & k2) == 0;  // final pass (k2 == G): the whole group descending
                srt = (lower == desc) ? fmax(srt, other) : fmin(srt, other);
            }
        }
        double cum = row_ok ? srt : 0.0;  // padded lanes sort to the end (-inf) and add nothing
#pragma unroll
        for (int off = 1; off < G; off <<= 1) {
            const double up = __shfl_up(cum, off, G);
            if (k >= off) cum += up;
        }
        const double shifted = cum - 1.0;
        const bool cond = row_ok && fma(srt, rank1, -shifted) > 0.0;
        const unsigned long long ball = __ballot(cond);
        // rho = last lane of the group whose condition holds (lane 0 always does for finite input)
        int rho;
        if constexpr (G == 64) {
            rho = ball ? 63 - __clzll((long long)ball) : -1;
        } else {
            const unsigned int mine = (unsigned int)((ball >> base) & ((1ull << G) - 1ull));
            rho = mine ? 31 - __clz((int)mine) : -1;
        }
        const double num = group_get<G>(shifted, base + (rho >= 0 ? rho : K - 1));
        const double theta = rho >= 0 ? num / (double)(rho + 1) : num / 0.0;
        a = row_ok ? fmax(x - theta, 0.0) : 0.0;
        lh_prev = lh;
    }
    if (col_ok && row_ok) {
        alpha[(int64_t)k * S + s] = a;
        alpha_prev[(int64_t)k * S + s] = ap;
    }
    // cost_s = vDv - 2 a.b + a^T G a ; ||alpha_unknown||^2
    double ga = 0.0;
#pragma unroll
    for (int l = 0; l < G; ++l) ga = fma(Grow[l], group_get<G>(a, base + l), ga);
    double part = col_ok ? fma(a, ga, -2.0 * a * bk) : 0.0;
    if (col_ok && k == 0) part += gb[(int64_t)tri(K, K) * S + sc];
    double n2 = (col_ok && row_ok && k >= K - n_u) ? a * a : 0.0;
    part = wave_sum(part);
    n2 = wave_sum(n2);
    if (lane == 0) {
        partials[2 * blockIdx.x] = part;
        partials[2 * blockIdx.x + 1] = n2;
    }
}

// ---- K <= 16: one sample per 16-lane DPP row, every cross-lane step on DPP (no LDS round trips) ------------
// Same arithmetic as k_alpha_phase_lanes<16> except the summation order of G a (four interleaved chains).
// The inner iteration of this kernel is one long dependent chain run by 64 waves at the headline size, i.e.
// pure latency: a ds_bpermute round trip per shuffle (16 for the product, 10 sort stages, 4 scan steps) is what
// the older kernel spends most of its 2.2 us per inner iteration on.
template <int CTRL, bool ZERO_OOB>
__device__ __forceinline__ double dpp16(double x) {
    if constexpr (ZERO_OOB) {  // lanes without a source must read 0: needs the "old" operand
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    } else {  // permutations / values that are masked afterwards: no destination to initialise
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, false);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, false);
        return __hiloint2double(hi, lo);
    }
}

// value of lane (k ^ J) of the 16-lane row
template <int J>
__device__ __forceinline__ double row_xor(double x, int k) {
    if constexpr (J == 1) return dpp16<0xB1, false>(x);       // quad_perm [1, 0, 3, 2]
    else if constexpr (J == 2) return dpp16<0x4E, false>(x);  // quad_perm [2, 3, 0, 1]
    else if constexpr (J == 8) return dpp16<0x128, false>(x); // row_ror:8
    else {                                                    // J == 4: row_shl:4 for the lower half of an octet
        const double from_above = dpp16<0x104, false>(x), from_below = dpp16<0x114, false>(x);
        return (k & 4) ? from_below : from_above;
    }
}

// acc += x[lane L of the row] * m   (the s_nop covers the VALU-write -> DPP-read hazard inside the asm: FIRST = the first
// read of a freshly written x; the reads behind it need none)
template <int L, bool FIRST = true>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double x, double m) {
    if constexpr (FIRST)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(x), "v"(m), "n"(L));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(L));
}

template <int J>
__device__ __forceinline__ void bitonic_step(double& srt, int k, int k2) {
    const double other = row_xor<J>(srt, k);
    const bool lower = (k & J) == 0;
    const bool desc = (k & k2) == 0;  // final pass (k2 == 16): the whole row descending
    srt = (lower == desc) ? fmax(srt, other) : fmin(srt, other);
}

__global__ __launch_bounds__(64) void k_alpha_phase_row16(const double* __restrict__ gb, double* __restrict__ alpha,
                                                          double* __restrict__ alpha_prev, SolverState* state, int S, int K,
                                                          int n_u, int n_iter2, double* __restrict__ partials) {
    if (state->done) return;
    const int lane = threadIdx.x;
    const int k = lane & 15, grp = lane >> 4, base = grp * 16;
    const int s = blockIdx.x * 4 + grp;
    const bool col_ok = s < S;
    const int sc = col_ok ? s : S - 1;
    const bool row_ok = k < K;
    const int kc = row_ok ? k : K - 1;

    double Gneg[16];  // -G_s[k][l]
#pragma unroll
    for (int l = 0; l < 16; ++l) {
        const int lc = l < K ? l : K - 1;
        const int lo = kc < lc ? kc : lc, hi = kc < lc ? lc : kc;
        const double v = gb[(int64_t)tri(lo, hi) * S + sc];
        Gneg[l] = (row_ok && l < K) ? -v : 0.0;
    }
    const double bk = row_ok ? gb[(int64_t)tri(kc, K) * S + sc] : 0.0;
    double a = row_ok ? alpha[(int64_t)kc * S + sc] : 0.0;
    double ap = row_ok ? alpha_prev[(int64_t)kc * S + sc] : 0.0;

    double a2 = state->a2, lh_prev = state->l_h_prev;
    const double lh = state->l_h;
    const double rank1 = (double)(k + 1);
    // momentum coefficients: from the host's row when there is one (lane t holds beta_t, read with v_readlane), else the
    // recurrence itself, in every wave
    const double* __restrict__ mrow = n_iter2 <= 64 ? momentum_row(state, n_iter2) : nullptr;
    int b_lo = 0, b_hi = 0;
    if (mrow != nullptr) {
        const int tl = lane < n_iter2 ? lane : n_iter2 - 1;
        const double cap = lane == 0 ? 0.9999 * sqrt(lh_prev / lh) : 0.9999;  // l_h_ = l_h behind the first step (:101)
        const double bv = fmin(mrow[2 + n_iter2 + tl], cap);
        b_lo = __double2loint(bv);
        b_hi = __double2hiint(bv);
    }
    for (int t = 0; t < n_iter2; ++t) {
        double beta;
        if (mrow != nullptr) beta = __hiloint2double(__builtin_amdgcn_readlane(b_hi, t), __builtin_amdgcn_readlane(b_lo, t));
        else momentum_step(a2, lh_prev, lh, beta);
        const double at = a + beta * (a - ap);
        ap = a;
        // b - G at, four interleaved chains; the columns beyond K are zero and skipped in fours (K is wave-uniform)
        double g0 = bk, g1 = 0.0, g2 = 0.0, g3 = 0.0;
        fmac_rowbcast<0>(g0, at, Gneg[0]);          fmac_rowbcast<1, false>(g1, at, Gneg[1]);
        fmac_rowbcast<2, false>(g2, at, Gneg[2]);   fmac_rowbcast<3, false>(g3, at, Gneg[3]);
        if (K > 4) {
            fmac_rowbcast<4, false>(g0, at, Gneg[4]);   fmac_rowbcast<5, false>(g1, at, Gneg[5]);
            fmac_rowbcast<6, false>(g2, at, Gneg[6]);   fmac_rowbcast<7, false>(g3, at, Gneg[7]);
        }
        if (K > 8) {
            fmac_rowbcast<8, false>(g0, at, Gneg[8]);   fmac_rowbcast<9, false>(g1, at, Gneg[9]);
            fmac_rowbcast<10, false>(g2, at, Gneg[10]); fmac_rowbcast<11, false>(g3, at, Gneg[11]);
        }
        if (K > 12) {
            fmac_rowbcast<12, false>(g0, at, Gneg[12]); fmac_rowbcast<13, false>(g1, at, Gneg[13]);
            fmac_rowbcast<14, false>(g2, at, Gneg[14]); fmac_rowbcast<15, false>(g3, at, Gneg[15]);
        }
        const double g = (g0 + g1) + (g2 + g3);
        const double x = at + g / lh;  // deconvolution.py:100: alpha_temp + (...) / l_h
        // ---- projection onto the simplex (deconvolution.py:25-35): bitonic sort of the row, descending.  The lanes
        // beyond K hold -inf: with K <= 8 (4, 2) only the first 8 (4, 2) lanes need sorting -- the last stage then runs
        // "whole row descending" (k2 = 16) on the shorter network.
        double srt = row_ok ? x : -INFINITY;
        if (K > 8) {
            bitonic_step<1>(srt, k, 2);
            bitonic_step<2>(srt, k, 4);  bitonic_step<1>(srt, k, 4);
            bitonic_step<4>(srt, k, 8);  bitonic_step<2>(srt, k, 8);  bitonic_step<1>(srt, k, 8);
            bitonic_step<8>(srt, k, 16); bitonic_step<4>(srt, k, 16); bitonic_step<2>(srt, k, 16);
            bitonic_step<1>(srt, k, 16);
        } else if (K > 4) {
            bitonic_step<1>(srt, k, 2);
            bitonic_step<2>(srt, k, 4);  bitonic_step<1>(srt, k, 4);
            bitonic_step<4>(srt, k, 16); bitonic_step<2>(srt, k, 16); bitonic_step<1>(srt, k, 16);
        } else if (K > 2) {
            bitonic_step<1>(srt, k, 2);
            bitonic_step<2>(srt, k, 16); bitonic_step<1>(srt, k, 16);
        } else {
            bitonic_step<1>(srt, k, 16);
        }
        double cum = row_ok ? srt : 0.0;  // padded lanes sort to the end (-inf) and add nothing
        cum += dpp16<0x111, true>(cum);   // row_shr:1 .. 8, lanes without a source read 0: inclusive scan
        cum += dpp16<0x112, true>(cum);
        if (K > 4) cum += dpp16<0x114, true>(cum);
        if (K > 8) cum += dpp16<0x118, true>(cum);
        // theta = (cumsum_rho - 1) / (rho + 1) with rho the LAST position where u_rho - (cumsum_rho - 1) / (rho + 1) > 0
        // (:28-33).  t_j = (cumsum_j - 1) / (j + 1) grows exactly while that condition holds -- t_(j+1) - t_j =
        // (u_(j+1) - t_j) / (j + 2), and u_(j+1) > t_(j+1) <=> u_(j+1) > t_j -- and never again behind rho (u keeps falling,
        // t_j >= u_j from there on), so theta is the row's MAXIMUM of t_j: every lane divides once (one instruction
        // sequence for the wave, as the one division at rho was) and four DPP steps take the maximum, instead of a
        // ballot, a count of leading zeros and an LDS round trip (ds_bpermute) to fetch cumsum_rho on the critical chain.
        // Same value as the reference's up to the rounding of a near-tie between t_rho and t_(rho+1).
        double tj = row_ok ? (cum - 1.0) / rank1 : -INFINITY;
        tj = fmax(tj, row_xor<1>(tj, k));
        tj = fmax(tj, row_xor<2>(tj, k));
        if (K > 4) tj = fmax(tj, row_xor<4>(tj, k));
        if (K > 8) tj = fmax(tj, row_xor<8>(tj, k));
        a = row_ok ? fmax(x - tj, 0.0) : 0.0;
        lh_prev = lh;
    }
    if (col_ok && row_ok) {
        alpha[(int64_t)k * S + s] = a;
        alpha_prev[(int64_t)k * S + s] = ap;
    }
    // cost_s = vDv - 2 a.b + a^T G a ; ||alpha_unknown||^2.  (G a as one chain of DPP row broadcasts in column order: the
    // same sum, term by term, as sixteen ds_bpermute round trips gave -- fma(-x, y, -z) = -fma(x, y, z).)
    double nga = 0.0;
    fmac_rowbcast<0>(nga, a, Gneg[0]);          fmac_rowbcast<1, false>(nga, a, Gneg[1]);
    fmac_rowbcast<2, false>(nga, a, Gneg[2]);   fmac_rowbcast<3, false>(nga, a, Gneg[3]);
    fmac_rowbcast<4, false>(nga, a, Gneg[4]);   fmac_rowbcast<5, false>(nga, a, Gneg[5]);
    fmac_rowbcast<6, false>(nga, a, Gneg[6]);   fmac_rowbcast<7, false>(nga, a, Gneg[7]);
    fmac_rowbcast<8, false>(nga, a, Gneg[8]);   fmac_rowbcast<9, false>(nga, a, Gneg[9]);
    fmac_rowbcast<10, false>(nga, a, Gneg[10]); fmac_rowbcast<11, false>(nga, a, Gneg[11]);
    fmac_rowbcast<12, false>(nga, a, Gneg[12]); fmac_rowbcast<13, false>(nga, a, Gneg[13]);
    fmac_rowbcast<14, false>(nga, a, Gneg[14]); fmac_rowbcast<15, false>(nga, a, Gneg[15]);
    const double ga = -nga;
    double part = col_ok ? fma(a, ga, -2.0 * a * bk) : 0.0;
    if (col_ok && k == 0) part += gb[(int64_t)tri(K, K) * S + sc];
    double n2 = (col_ok && row_ok && k >= K - n_u) ? a * a : 0.0;
    part = wave_sum(part);
    n2 = wave_sum(n2);
    int last = 0;
    if (lane == 0) {
        // handed to the closing workgroup as returning atomic exchanges (see k_gram_v2_reduce: all atomics on an address
        // are performed in one place and a returned value means "performed"), then this workgroup's arrival is counted
        unsigned long long* __restrict__ pp = reinterpret_cast<unsigned long long*>(partials) + 2 * blockIdx.x;
        const unsigned long long r0 = atomicExch(pp, (unsigned long long)__double_as_longlong(part));
        const unsigned long long r1 = atomicExch(pp + 1, (unsigned long long)__double_as_longlong(n2));
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(r0), "v"(r1) : "memory");
        last = atomicAdd(&state->arrive, 1) == (int)gridDim.x - 1;
    }
    // The last workgroup to arrive closes the outer iteration (what k_finish_iteration does as a launch of its own
    // behind the other alpha kernels): every other workgroup has read the scalar state long before it incremented the
    // counter, so advancing the state here races with nobody.  Same summation order as k_finish_iteration.
    last = __shfl(last, 0, 64);
    if (last) finish_iteration_body<true>(partials, (int)gridDim.x, state, n_iter2);
}

static hipError_t launch_alpha_row16(const double* gb, double* alpha, double* alpha_prev, SolverState* state, int S,
                                     int K, int n_u, int n_iter2, double* partials, hipStream_t st) {
    const int nb = (S + 3) / 4;
    hipLaunchKernelGGL(k_alpha_phase_row16, dim3(nb), dim3(64), 0, st, gb, alpha, alpha_prev, state, S, K, n_u,
                       n_iter2, partials);  // (closes the outer iteration itself)
    return hipGetLastError();
}

template <int G>
static hipError_t launch_alpha_lanes_t(const double* gb, double* alpha, double* alpha_prev, SolverState* state,
                                       int S, int K, int n_u, int n_iter2, double* partials, hipStream_t st) {
    const int nb = (S * G + 63) / 64;
    hipLaunchKernelGGL(k_alpha_phase_lanes<G>, dim3(nb), dim3(64), 0, st, gb, alpha, alpha_prev, state, S, K, n_u,
                       n_iter2, partials);
    hipLaunchKernelGGL(k_finish_iteration, dim3(1), dim3(64), 0, st, partials, nb, state, n_iter2);
    return hipGetLastError();
}

hipError_t launch_alpha_phase(const double* gb, double* alpha, double* alpha_prev,
                              SolverState* state, int S, int n_c, int n_u, int n_iter2,
                              double* partials, bool thread_per_sample, hipStream_t st) {
    const int K = n_c + n_u;
    if (!thread_per_sample) {
        if (K <= 16) return launch_alpha_row16(gb, alpha, alpha_prev, state, S, K, n_u, n_iter2, partials, st);
        if (K <= 4) return launch_alpha_lanes_t<4>(gb, alpha, alpha_prev, state, S, K, n_u, n_iter2, partials, st);
        if (K <= 8) return launch_alpha_lanes_t<8>(gb, alpha, alpha_prev, state, S, K, n_u, n_iter2, partials, st);
        if (K <= 16) return launch_alpha_lanes_t<16>(gb, alpha, alpha_prev, state, S, K, n_u, n_iter2, partials, st);
        if (K <= 32) return launch_alpha_lanes_t<32>(gb, alpha, alpha_prev, state, S, K, n_u, n_iter2, partials, st);
        // (a wave per sample: the one-thread-per-sample kernel below took 8.3 ms at 128 samples and K = 41)
        if (K <= 64) return launch_alpha_lanes_t<64>(gb, alpha, alpha_prev, state, S, K, n_u, n_iter2, partials, st);
    } else {
        if (K <= 4) return launch_alpha_t<4>(gb, alpha, alpha_prev, state, S, K, n_u, n_iter2, partials, st);
        if (K <= 8) return launch_alpha_t<8>(gb, alpha, alpha_prev, state, S, K, n_u, n_iter2, partials, st);
        if (K <= 16) return launch_alpha_t<16>(gb, alpha, alpha_prev, state, S, K, n_u, n_iter2, partials, st);
    }
    if (K <= kMaxK) {
        const int nb = (S + 63) / 64;
        hipLaunchKernelGGL(k_alpha_phase_dyn, dim3(nb), dim3(64), 0, st, gb, alpha, alpha_prev, state, S, K,
                           n_u, n_iter2, partials);
        hipLaunchKernelGGL(k_finish_iteration, dim3(1), dim3(64), 0, st, partials, nb, state, n_iter2);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

// ---- purity-constrained alpha phase: Frank-Wolfe on the packed Gram data ----------------------------
// demethify/deconvolution.py:280-302 (`frank_wolfe_nmf`): per sample, the known block keeps mass
// purity[s] and the unknown block mass 1 - purity[s]; iteration k moves by 2 / (k + 2) towards the
// vertex with the smallest gradient entry in each block.  grad = G a - b (= -R^T (d * (v - R a))).
// One thread per sample; a[] in scratch for the runtime-K loop (secondary path, run time is dominated by
// the row pass).
__global__ __launch_bounds__(64) void k_alpha_frank_wolfe(const double* __restrict__ gb,
                                                          double* __restrict__ alpha,
                                                          const double* __restrict__ purity,
                                                          const SolverState* __restrict__ state, int S, int K,
                                                          int n_u, int max_iter, double* __restrict__ partials) {
    if (state->done) return;
    const int s = blockIdx.x * 64 + threadIdx.x;
    const bool active = s < S;
    const int sc = active ? s : S - 1;
    const int n_c = K - n_u;
    const double* __restrict__ G = gb + sc;
    const int64_t gs = S;
    const double pur = purity[sc];
    double a[kMaxK], grad[kMaxK];
    for (int k = 0; k < K; ++k) a[k] = alpha[(int64_t)k * S + sc];
    for (int it = 0; it < max_iter; ++it) {
        for (int k = 0; k < K; ++k) grad[k] = -G[tri(k, K) * gs];
        for (int l = 0; l < K; ++l)
            for (int k = 0; k <= l; ++k) {
                const double gkl = G[tri(k, l) * gs];
                grad[k] = fma(gkl, a[l], grad[k]);
                if (k != l) grad[l] = fma(gkl, a[k], grad[l]);
            }
        int i1 = 0, i2 = n_c;  // np.argmin: first index of the minimum
        for (int k = 1; k < n_c; ++k)
            if (grad[k] < grad[i1]) i1 = k;
        for (int k = n_c + 1; k < K; ++k)
            if (grad[k] < grad[i2]) i2 = k;
        const double gamma = 2.0 / (double)(it + 2);
        for (int k = 0; k < K; ++k) {
            double vertex = 0.0;
            if (k < n_c && k == i1) vertex = pur;
            if (k >= n_c && k == i2) vertex = 1.0 - pur;
            a[k] = (1.0 - gamma) * a[k] + gamma * vertex;
        }
    }
    double cost = 0.0, n2 = 0.0;
    if (active) {
        cost = G[tri(K, K) * gs];
        double lin = 0.0, quad = 0.0;
        for (int l = 0; l < K; ++l) {
            alpha[(int64_t)l * S + s] = a[l];
            lin = fma(a[l], G[tri(l, K) * gs], lin);
            double off = 0.0;
            for (int k = 0; k < l; ++k) off = fma(G[tri(k, l) * gs], a[k], off);
            quad = fma(a[l], fma(2.0, off, G[tri(l, l) * gs] * a[l]), quad);
            if (l >= n_c) n2 = fma(a[l], a[l], n2);
        }
        cost = cost - 2.0 * lin + quad;
    }
    cost = wave_sum(cost);
    n2 = wave_sum(n2);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = cost;
        partials[2 * blockIdx.x + 1] = n2;
    }
}

// K <= 16: one sample per 16-lane DPP row, as k_alpha_phase_row16 (the thread-per-sample kernel above keeps a[]
// and grad[] in scratch and needs ~40 us per Frank-Wolfe iteration; the CLI default with --purity is 500 of them
// per outer iteration).  Lane k owns row k of G_s; both block-wise argmins are butterfly reductions on DPP
// moves, lowest index first among equal minima (np.argmin).
template <int J>
__device__ __forceinline__ void argmin_step(double& v, int& idx, int k) {
    const double ov = row_xor<J>(v, k);
    int oi;
    if constexpr (J == 1) oi = __builtin_amdgcn_mov_dpp(idx, 0xB1, 0xF, 0xF, false);
    else if constexpr (J == 2) oi = __builtin_amdgcn_mov_dpp(idx, 0x4E, 0xF, 0xF, false);
    else if constexpr (J == 8) oi = __builtin_amdgcn_mov_dpp(idx, 0x128, 0xF, 0xF, false);
    else {
        const int up = __builtin_amdgcn_mov_dpp(idx, 0x104, 0xF, 0xF, false);
        const int dn = __builtin_amdgcn_mov_dpp(idx, 0x114, 0xF, 0xF, false);
        oi = (k & 4) ? dn : up;
    }
    const bool take = ov < v || (ov == v && oi < idx);
    v = take ? ov : v;
    idx = take ? oi : idx;
}

__global__ __launch_bounds__(64) void k_alpha_frank_wolfe_row16(const double* __restrict__ gb,
                                                                double* __restrict__ alpha,
                                                                const double* __restrict__ purity,
                                                                const SolverState* __restrict__ state, int S, int K,
                                                                int n_u, int max_iter, double* __restrict__ partials) {
    if (state->done) return;
    const int lane = threadIdx.x;
    const int k = lane & 15, grp = lane >> 4, base = grp * 16;
    const int s = blockIdx.x * 4 + grp;
    const bool col_ok = s < S;
    const int sc = col_ok ? s : S - 1;
    const bool row_ok = k < K;
    const int kc = row_ok ? k : K - 1;
    const int n_c = K - n_u;
    const bool known = k < n_c, unknown = row_ok && !known;

    double Grow[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) {
        const int lc = l < K ? l : K - 1;
        const int lo = kc < lc ? kc : lc, hi = kc < lc ? lc : kc;
        const double v = gb[(int64_t)tri(lo, hi) * S + sc];
        Grow[l] = (row_ok && l < K) ? v : 0.0;
    }
    const double bk = row_ok ? gb[(int64_t)tri(kc, K) * S + sc] : 0.0;
    const double pur = purity[sc];
    const double mass = known ? pur : 1.0 - pur;  // what this lane's vertex would put on row k
    double a = row_ok ? alpha[(int64_t)kc * S + sc] : 0.0;

    for (int it = 0; it < max_iter; ++it) {
        double g0 = -bk, g1 = 0.0, g2 = 0.0, g3 = 0.0;  // grad = G a - b (= -R^T (d * (v - R a)))
        fmac_rowbcast<0>(g0, a, Grow[0]);   fmac_rowbcast<1>(g1, a, Grow[1]);
        fmac_rowbcast<2>(g2, a, Grow[2]);   fmac_rowbcast<3>(g3, a, Grow[3]);
        fmac_rowbcast<4>(g0, a, Grow[4]);   fmac_rowbcast<5>(g1, a, Grow[5]);
        fmac_rowbcast<6>(g2, a, Grow[6]);   fmac_rowbcast<7>(g3, a, Grow[7]);
        fmac_rowbcast<8>(g0, a, Grow[8]);   fmac_rowbcast<9>(g1, a, Grow[9]);
        fmac_rowbcast<10>(g2, a, Grow[10]); fmac_rowbcast<11>(g3, a, Grow[11]);
        fmac_rowbcast<12>(g0, a, Grow[12]); fmac_rowbcast<13>(g1, a, Grow[13]);
        fmac_rowbcast<14>(g2, a, Grow[14]); fmac_rowbcast<15>(g3, a, Grow[15]);
        const double grad = (g0 + g1) + (g2 + g3);
        double v1 = known ? grad : INFINITY, v2 = unknown ? grad : INFINITY;
        int i1 = k, i2 = k;
        argmin_step<1>(v1, i1, k); argmin_step<2>(v1, i1, k); argmin_step<4>(v1, i1, k); argmin_step<8>(v1, i1, k);
        argmin_step<1>(v2, i2, k); argmin_step<2>(v2, i2, k); argmin_step<4>(v2, i2, k); argmin_step<8>(v2, i2, k);
        const double gamma = 2.0 / (double)(it + 2);
        const bool at_vertex = (known && k == i1) || (unknown && k == i2);
        const double vertex = at_vertex ? mass : 0.0;
        a = (1.0 - gamma) * a + gamma * vertex;
    }
    if (col_ok && row_ok) alpha[(int64_t)k * S + s] = a;
    // cost_s = vDv - 2 a.b + a^T G a ; ||alpha_unknown||^2
    double ga = 0.0;
#pragma unroll
    for (int l = 0; l < 16; ++l) ga = fma(Grow[l], __shfl(a, base + l, 64), ga);
    double part = col_ok ? fma(a, ga, -2.0 * a * bk) : 0.0;
    if (col_ok && k == 0) part += gb[(int64_t)tri(K, K) * S + sc];
    double n2 = (col_ok && unknown) ? a * a : 0.0;
    part = wave_sum(part);
    n2 = wave_sum(n2);
    if (lane == 0) {
        partials[2 * blockIdx.x] = part;
        partials[2 * blockIdx.x + 1] = n2;
    }
}

hipError_t launch_alpha_frank_wolfe(const double* gb, double* alpha, const double* purity, SolverState* state,
                                    int S, int n_c, int n_u, int max_iter, double* partials, hipStream_t st) {
    const int K = n_c + n_u;
    if (K > kMaxK) return hipErrorInvalidValue;
    if (K <= 16 && n_c >= 1) {
        const int nb4 = (S + 3) / 4;
        hipLaunchKernelGGL(k_alpha_frank_wolfe_row16, dim3(nb4), dim3(64), 0, st, gb, alpha, purity, state, S, K, n_u,
                           max_iter, partials);
        hipLaunchKernelGGL(k_finish_iteration, dim3(1), dim3(64), 0, st, partials, nb4, state, max_iter);
        return hipGetLastError();
    }
    const int nb = (S + 63) / 64;
    hipLaunchKernelGGL(k_alpha_frank_wolfe, dim3(nb), dim3(64), 0, st, gb, alpha, purity, state, S, K, n_u, max_iter,
                       partials);
    hipLaunchKernelGGL(k_finish_iteration, dim3(1), dim3(64), 0, st, partials, nb, state, max_iter);
    return hipGetLastError();
}

// ---- standalone projection (KAT entry point) ----------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(64) void k_project(const double* __restrict__ X, double* __restrict__ out,
                                                int K, int S, double z) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= S) return;
    double x[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) x[k] = k < K ? X[(int64_t)k * S + s] : 0.0;
    project_column<KMAX>(x, K, z);
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) out[(int64_t)k * S + s] = x[k];
}

hipError_t launch_project_simplex(const double* X, double* out, int K, int S, double z,
                                  hipStream_t st) {
    const dim3 grid((S + 63) / 64), block(64);
    if (K <= 4) hipLaunchKernelGGL(k_project<4>, grid, block, 0, st, X, out, K, S, z);
    else if (K <= 8) hipLaunchKernelGGL(k_project<8>, grid, block, 0, st, X, out, K, S, z);
    else if (K <= 16) hipLaunchKernelGGL(k_project<16>, grid, block, 0, st, X, out, K, S, z);
    else if (K <= kMaxK) hipLaunchKernelGGL(k_project_dyn, grid, block, 0, st, X, out, K, S, z);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- solver set-up helpers -------------------------------------------------------------------
// Copy the problem's known block (packed triangle over the n_c+1 extended indices (Rt, v)) into
// the solver's packed buffer over (Rt, u, v): index n_c ("v") moves to K.
__global__ __launch_bounds__(256) void k_scatter_known(const double* __restrict__ gb_known,
                                                       double* __restrict__ gb, int n_c, int K, int S) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    for (int l = 0; l <= n_c; ++l) {
        const int ld = l == n_c ? K : l;
        for (int k = 0; k <= l; ++k) {
            const int kd = k == n_c ? K : k;
            gb[(int64_t)tri(kd, ld) * S + s] = gb_known[(int64_t)tri(k, l) * S + s];
        }
    }
}

hipError_t launch_scatter_known_block(const double* gb_known, double* gb, int n_c, int K, int S,
                                      hipStream_t st) {
    hipLaunchKernelGGL(k_scatter_known, dim3((S + 255) / 256), dim3(256), 0, st, gb_known, gb, n_c, K, S);
    return hipGetLastError();
}

// State init (deconvolution.py:192-204).  Expects state->u_norm2 and state->cf already written
// by the sum-of-squares and cost launches that precede it on the stream.
__global__ __launch_bounds__(256) void k_init_state(SolverState* __restrict__ state,
                                                    const double* __restrict__ consts,
                                                    const double* __restrict__ alpha, int S, int n_c,
                                                    int n_u) {
    __shared__ double red[4];
    double acc = 0.0;
    const double* A2 = alpha + (int64_t)n_c * S;
    for (int i = threadIdx.x; i < n_u * S; i += 256) acc = fma(A2[i], A2[i], acc);
    const double n2 = block_sum<256>(acc, red);
    if (threadIdx.x == 0) {
        const double dsq = consts[0], rt2 = consts[1];
        state->a1 = 1.0;
        state->a2 = 1.0;
        state->dsq = dsq;
        state->rt_norm2 = rt2;
        state->l_w = n2 * dsq;
        state->l_w_prev = state->l_w;
        state->l_h = (rt2 + state->u_norm2) * dsq;
        state->l_h_prev = state->l_h;
        state->cf = __longlong_as_double(0x7FF8000000000000ll);  // not computed yet (dmf_solver_step, dmf_solver_get)
        state->cf_prev = state->cf;
        state->tol = 0.0;
        state->band = 1.0;
        state->mom = nullptr;
        state->mom_n = -1;
        state->iters = 0;
        state->done = 0;
        state->arrive = 0;
    }
}

hipError_t launch_init_state(SolverState* state, const double* consts, const double* alpha, int S,
                             int n_c, int n_u, hipStream_t st) {
    hipLaunchKernelGGL(k_init_state, dim3(1), dim3(256), 0, st, state, consts, alpha, S, n_c, n_u);
    return hipGetLastError();
}

}  // namespace dmf
