// u phase (demethify/deconvolution.py:81-90 and, for the unsupervised variant, :157-164).
//
// Gram form: alpha is fixed during the phase and rows are independent, so
// for row i the gradient at a point p (1 x n_u) is  c_i - p M_i  with
//     c_i = (d_i * (v_i - Rt_i alpha_known)) alpha_unk^T          (n_u)
//     M_i = alpha_unk diag(d_i) alpha_unk^T                        (n_u x n_u, symmetric)
// One pass over V and D builds c_i, M_i; the n_iter2 inner iterations are then row-local.
#include "dmf_device.h"
#include "dmf_internal.h"

namespace dmf {

template <int NU>
__global__ __launch_bounds__(256) void k_u_phase_gram(
    const double* __restrict__ V, const double* __restrict__ D, const double* __restrict__ Rt,
    const double* __restrict__ alpha, double* __restrict__ u, double* __restrict__ u_prev,
    const SolverState* __restrict__ state, int64_t N, int S, int n_c, int n_iter2, int mode,
    int alpha_in_lds) {
    constexpr int NP = NU * (NU + 1) / 2;
    constexpr int NV = NU + NP;
    extern __shared__ double lds_dyn[];
    __shared__ double cm[kRowsPerBlockU][NV + 1];
    if (state->done) return;

    const int K = n_c + NU;
    const double* A = alpha;
    if (alpha_in_lds) {
        for (int i = threadIdx.x; i < K * S; i += 256) lds_dyn[i] = alpha[i];
        __syncthreads();
        A = lds_dyn;
    }
    const double* A2 = A + (int64_t)n_c * S;  // unknown rows of alpha

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t row0 = (int64_t)blockIdx.x * kRowsPerBlockU;

    // ---- phase A: c_i, M_i for the block's rows (one wave per row, lanes over samples)
    for (int rr = 0; rr < kRowsPerBlockU / 4; ++rr) {
        const int il = wave * (kRowsPerBlockU / 4) + rr;
        const int64_t i = row0 + il;
        if (i >= N) break;
        double c[NU], M[NP];
#pragma unroll
        for (int j = 0; j < NU; ++j) c[j] = 0.0;
#pragma unroll
        for (int p = 0; p < NP; ++p) M[p] = 0.0;
        const double* rt_row = Rt + i * n_c;
        for (int s = lane; s < S; s += 64) {
            const double v = V[i * S + s];
            const double d = D[i * S + s];
            double pred = 0.0;
            for (int k = 0; k < n_c; ++k) pred = fma(rt_row[k], A[k * S + s], pred);
            const double e = v - pred;
            double a2[NU], da[NU];
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                a2[j] = A2[j * S + s];
                da[j] = d * a2[j];
                c[j] = fma(da[j], e, c[j]);
            }
#pragma unroll
            for (int l = 0; l < NU; ++l)
#pragma unroll
                for (int j = 0; j <= l; ++j) M[tri(j, l)] = fma(da[j], a2[l], M[tri(j, l)]);
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const double t = wave_sum(c[j]);
            if (lane == 0) cm[il][j] = t;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const double t = wave_sum(M[p]);
            if (lane == 0) cm[il][NU + p] = t;
        }
    }
    __syncthreads();

    // ---- phase B: n_iter2 accelerated projected-gradient steps, one lane per row
    if (threadIdx.x < kRowsPerBlockU) {
        const int64_t i = row0 + threadIdx.x;
        if (i < N) {
            double c[NU], M[NP], uu[NU], up[NU];
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                c[j] = cm[threadIdx.x][j];
                uu[j] = u[i * NU + j];
                up[j] = u_prev[i * NU + j];
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) M[p] = cm[threadIdx.x][NU + p];
            double a1 = state->a1, lw_prev = state->l_w_prev;
            const double lw = state->l_w;
            for (int t = 0; t < n_iter2; ++t) {
                double beta;
                momentum_step(a1, lw_prev, lw, beta);
                double ut[NU], base[NU];
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    ut[j] = uu[j] + beta * (uu[j] - up[j]);
                    base[j] = mode == 1 ? uu[j] : ut[j];  // deconvolution.py:163 vs :88
                    up[j] = uu[j];
                }
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    double g = c[j];
#pragma unroll
                    for (int l = 0; l < NU; ++l) g = fma(-base[l], M[l <= j ? tri(l, j) : tri(j, l)], g);
                    const double x = ut[j] + g / lw;
                    uu[j] = fmin(fmax(x, 0.0), 1.0);
                }
                lw_prev = lw;
            }
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                u[i * NU + j] = uu[j];
                u_prev[i * NU + j] = up[j];
            }
        }
    }
}

bool u_phase_gram_supported(int S, int n_c, int n_u) {
    (void)S;
    (void)n_c;
    return n_u >= 1 && n_u <= 16;  // 9..16: c and M (up to 152 doubles) still fit the register file
}

template <int NU>
static hipError_t launch_u_gram_t(const double* V, const double* D, const double* Rt,
                                  const double* alpha, double* u, double* u_prev,
                                  const SolverState* state, int64_t N, int S, int n_c, int n_iter2,
                                  int mode, hipStream_t st) {
    const size_t lds = (size_t)(n_c + NU) * S * sizeof(double);
    const int in_lds = lds <= 36 * 1024;
    const int64_t nb = (N + kRowsPerBlockU - 1) / kRowsPerBlockU;
    hipLaunchKernelGGL(k_u_phase_gram<NU>, dim3((unsigned)nb), dim3(256), in_lds ? lds : 0, st, V, D,
                       Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, in_lds);
    return hipGetLastError();
}

hipError_t launch_u_phase_gram(const double* V, const double* D, const double* Rt,
                               const double* alpha, double* u, double* u_prev,
                               const SolverState* state, int64_t N, int S, int n_c, int n_u,
                               int n_iter2, int mode, hipStream_t st) {
    switch (n_u) {
#define DMF_CASE(NU_) \
    case NU_: return launch_u_gram_t<NU_>(V, D, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, st);
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4) DMF_CASE(5) DMF_CASE(6) DMF_CASE(7) DMF_CASE(8)
        DMF_CASE(9) DMF_CASE(10) DMF_CASE(11) DMF_CASE(12) DMF_CASE(13) DMF_CASE(14) DMF_CASE(15) DMF_CASE(16)
#undef DMF_CASE
        default: return hipErrorInvalidValue;
    }
}

// ---- schedule-faithful fallback: one inner iteration per launch, any n_u <= 64 ------------
// u_next = clip(u_temp + ((D * (V - Rt a_known - base a_unk)) a_unk^T) / l_w, 0, 1)
__global__ __launch_bounds__(256) void k_u_step_direct(
    const double* __restrict__ V, const double* __restrict__ D, const double* __restrict__ Rt,
    const double* __restrict__ alpha, const double* __restrict__ u_cur,
    const double* __restrict__ u_prev, double* __restrict__ u_next,
    const SolverState* __restrict__ state, int64_t N, int S, int n_c, int n_u, int t, int mode) {
    extern __shared__ double lds_dyn[];  // per wave: resid[S], ut[64], base[64]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    double* resid = lds_dyn + (size_t)wave * (S + 128);
    double* ut = resid + S;
    double* base = ut + 64;

    double a1 = state->a1, lw_prev = state->l_w_prev, beta = 0.0;
    const double lw = state->l_w;
    for (int q = 0; q <= t; ++q) {
        momentum_step(a1, lw_prev, lw, beta);
        lw_prev = lw;
    }
    const double* A2 = alpha + (int64_t)n_c * S;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < N; i += (int64_t)gridDim.x * 4) {
        if (lane < n_u) {
            const double uc = u_cur[i * n_u + lane];
            const double x = uc + beta * (uc - u_prev[i * n_u + lane]);
            ut[lane] = x;
            base[lane] = mode == 1 ? uc : x;
        }
        __builtin_amdgcn_wave_barrier();
        const double* rt_row = Rt + i * n_c;
        for (int s = lane; s < S; s += 64) {
            double pred = 0.0;
            for (int k = 0; k < n_c; ++k) pred = fma(rt_row[k], alpha[k * S + s], pred);
            for (int j = 0; j < n_u; ++j) pred = fma(base[j], A2[j * S + s], pred);
            resid[s] = D[i * S + s] * (V[i * S + s] - pred);
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < n_u; ++j) {
            double part = 0.0;
            for (int s = lane; s < S; s += 64) part = fma(resid[s], A2[j * S + s], part);
            const double g = wave_sum(part);
            if (lane == 0) u_next[i * n_u + j] = fmin(fmax(ut[j] + g / lw, 0.0), 1.0);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

bool u_step_direct_supported(int S, int n_c, int n_u) {
    (void)n_c;
    return n_u >= 1 && n_u <= 64 && (size_t)4 * (S + 128) * sizeof(double) <= 60 * 1024;
}

hipError_t launch_u_step_direct(const double* V, const double* D, const double* Rt,
                                const double* alpha, const double* u_cur, const double* u_prev,
                                double* u_next, const SolverState* state, int64_t N, int S, int n_c,
                                int n_u, int t, int mode, hipStream_t st) {
    const size_t lds = (size_t)4 * (S + 128) * sizeof(double);
    int64_t nb = (N + 3) / 4;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_u_step_direct, dim3((unsigned)nb), dim3(256), lds, st, V, D, Rt, alpha, u_cur,
                       u_prev, u_next, state, N, S, n_c, n_u, t, mode);
    return hipGetLastError();
}

}  // namespace dmf
