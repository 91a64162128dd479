// u phase on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), gfx950.
//
// Same Gram form as dmf_kernels_rowpass.hip (c_i, M_i per row, then n_iter2 row-local steps) but the
// three contractions of the row pass run as MFMAs on 16-row x 16-sample tiles held in the
// "row-on-lane" layout  lane = (row = l & 15, q = l >> 4), register r <-> sample s0 + 4q + r:
//     E^T = V^T - alpha_known^T Rt^T        A = -alpha_known^T (m = sample, k = known type), B = Rt^T, C = V^T
//     c^T += alpha_unk (D*E)^T              A = alpha_unk      (m = unknown j, k = sample),  B = (D*E)^T
//     M^T += P D^T                          A = P (m = pair (j,l), k = sample), P = alpha_unk_j * alpha_unk_l
// (the m <-> sample permutation of the first product is folded into its A operand, so one 32-byte
// contiguous load per lane feeds all three).  The alpha-derived A operands are built once per wave and
// reused for every row block; workgroups are persistent over row blocks.
//
// Layout facts used (verified on hardware with tools/mfma_probe.hip):
//   A[i][k]: lane (i = l & 15, k = l >> 4);  B[k][j]: lane (k = l >> 4, j = l & 15);
//   C/D register r of lane l = C[(l >> 4) + 4 r][l & 15].
#include "dmf_device.h"
#include "dmf_internal.h"

namespace dmf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kStripsPerWave = 4;  // 16-sample strips owned by one wave (64 samples)
constexpr int kMfmaMaxWaves = 8;   // S <= 512 on this path

template <int NKC, int NU>
__global__ __launch_bounds__(512) void k_u_phase_mfma(
    const double* __restrict__ V, const double* __restrict__ D, const double* __restrict__ Rt,
    const double* __restrict__ alpha, double* __restrict__ u, double* __restrict__ u_prev,
    const SolverState* __restrict__ state, int64_t N, int S, int n_c, int n_iter2, int mode) {
    constexpr int NP = NU * (NU + 1) / 2;
    constexpr int NMT = (NP + 15) / 16;  // 16-row tiles of the pair matrix
    constexpr int NV = NU + NP;
    extern __shared__ double red[];      // [2][NW][NV][16]
    if (state->done) return;

    const int NW = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int m16 = lane & 15, q = lane >> 4;
    const double* __restrict__ A2 = alpha + (int64_t)n_c * S;

    // ---- per-wave constant A operands -------------------------------------------------------
    double a1op[kStripsPerWave][NKC > 0 ? NKC : 1];
    double a2op[kStripsPerWave][4];
    double pop[kStripsPerWave][NMT][4];
    bool strip_ok[kStripsPerWave];
#pragma unroll
    for (int t = 0; t < kStripsPerWave; ++t) {
        const int s0 = (wave * kStripsPerWave + t) * 16;
        strip_ok[t] = s0 < S;
        // first product: m <-> sample s0 + 4 (m & 3) + (m >> 2), k <-> known type 4 kc + q
        const int s_e = s0 + 4 * (m16 & 3) + (m16 >> 2);
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int kk = kc * 4 + q;
            a1op[t][kc] = (kk < n_c && s_e < S) ? -alpha[(int64_t)kk * S + s_e] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = s0 + 4 * q + r;  // k-step r: k = q <-> sample s0 + 4 q + r
            a2op[t][r] = (m16 < NU && s < S) ? A2[(int64_t)m16 * S + s] : 0.0;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int p = mt * 16 + m16;
                double val = 0.0;
                if (p < NP && s < S) {
                    int l = 0;
                    while ((l + 1) * (l + 2) / 2 <= p) ++l;
                    const int j = p - l * (l + 1) / 2;
                    val = A2[(int64_t)j * S + s] * A2[(int64_t)l * S + s];
                }
                pop[t][mt][r] = val;
            }
        }
    }

    const bool vec_ok = (S & 3) == 0;
    const int64_t nblk = (N + 15) / 16;
    double u2_acc = 0.0;
    int it = 0;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x, ++it) {
        const int64_t row0 = blk * 16;
        const int64_t row = row0 + m16;
        const bool row_ok = row < N;
        const int64_t rowc = row_ok ? row : N - 1;

        double rtop[NKC > 0 ? NKC : 1];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int kk = kc * 4 + q;
            rtop[kc] = (row_ok && kk < n_c) ? Rt[rowc * n_c + kk] : 0.0;
        }
        v4d cacc = {0.0, 0.0, 0.0, 0.0};
        v4d macc[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) macc[mt] = cacc;

#pragma unroll
        for (int t = 0; t < kStripsPerWave; ++t) {
            if (strip_ok[t]) {
                const int s0 = (wave * kStripsPerWave + t) * 16 + 4 * q;
                const double* __restrict__ vp = V + rowc * S + s0;
                const double* __restrict__ dp = D + rowc * S + s0;
                v4d e, d;
                if (vec_ok) {
                    if (s0 < S) {
                        const v2d v01 = *reinterpret_cast<const v2d*>(vp);
                        const v2d v23 = *reinterpret_cast<const v2d*>(vp + 2);
                        const v2d d01 = *reinterpret_cast<const v2d*>(dp);
                        const v2d d23 = *reinterpret_cast<const v2d*>(dp + 2);
                        e = v4d{v01.x, v01.y, v23.x, v23.y};
                        d = v4d{d01.x, d01.y, d23.x, d23.y};
                    } else {
                        e = v4d{0.0, 0.0, 0.0, 0.0};
                        d = e;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = s0 + r < S;
                        e[r] = ok ? vp[r] : 0.0;
                        d[r] = ok ? dp[r] : 0.0;
                    }
                }
                if (!row_ok) d = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc)
                    e = __builtin_amdgcn_mfma_f64_16x16x4f64(a1op[t][kc], rtop[kc], e, 0, 0, 0);
                const v4d w = d * e;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2op[t][r], w[r], cacc, 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt)
                        macc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(pop[t][mt][r], d[r], macc[mt], 0, 0, 0);
                }
            }
        }

        // ---- partial c / M of this wave's samples -> LDS (register r' of lane <-> m = q + 4 r')
        double* __restrict__ mine = red + ((size_t)((it & 1) * NW + wave) * NV) * 16;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int m = q + 4 * rr;
            if (m < NU) mine[m * 16 + m16] = cacc[rr];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const int p = mt * 16 + m;
                if (p < NP) mine[(NU + p) * 16 + m16] = macc[mt][rr];
            }
        }
        __syncthreads();

        // ---- row-local inner iterations by one wave (round robin), lane = (row, unknown j)
        if (wave == it % NW) {
            constexpr int RPW = 64 / NU;  // rows per pass
            const double* __restrict__ all = red + ((size_t)(it & 1) * NW * NV) * 16;
            const int rl = lane / NU, j = lane - rl * NU;
            for (int pass0 = 0; pass0 < 16; pass0 += RPW) {
                const int rloc = pass0 + rl;
                const bool ok = rl < RPW && rloc < 16 && row0 + rloc < N;
                const int rlc = rloc < 16 ? rloc : 15;
                double cj = 0.0, Mrow[NU];
#pragma unroll
                for (int l = 0; l < NU; ++l) Mrow[l] = 0.0;
                for (int w = 0; w < NW; ++w) {
                    const double* __restrict__ part = all + (size_t)w * NV * 16;
                    cj += part[j * 16 + rlc];
#pragma unroll
                    for (int l = 0; l < NU; ++l) {
                        const int p = l <= j ? tri(l, j) : tri(j, l);
                        Mrow[l] += part[(NU + p) * 16 + rlc];
                    }
                }
                const int64_t gi = ok ? (row0 + rloc) * NU + j : 0;
                double uu = ok ? u[gi] : 0.0;
                double up = ok ? u_prev[gi] : 0.0;
                double a1 = state->a1, lw_prev = state->l_w_prev;
                const double lw = state->l_w;
                const int lane0 = lane - j;
                for (int t2 = 0; t2 < n_iter2; ++t2) {
                    double beta;
                    momentum_step(a1, lw_prev, lw, beta);
                    const double ut = uu + beta * (uu - up);
                    const double base = mode == 1 ? uu : ut;  // deconvolution.py:163 vs :88
                    up = uu;
                    double g = cj;
#pragma unroll
                    for (int l = 0; l < NU; ++l) g = fma(-__shfl(base, lane0 + l, 64), Mrow[l], g);
                    uu = fmin(fmax(ut + g / lw, 0.0), 1.0);
                    lw_prev = lw;
                }
                if (ok) {
                    u[gi] = uu;
                    u_prev[gi] = up;
                    u2_acc = fma(uu, uu, u2_acc);
                }
            }
        }
    }
    (void)u2_acc;
}

bool u_phase_mfma_supported(int S, int n_c, int n_u) {
    return n_u >= 1 && n_u <= 8 && n_c <= 16 && S <= 16 * kStripsPerWave * kMfmaMaxWaves;
}

template <int NKC, int NU>
static hipError_t launch_u_mfma_t(const double* V, const double* D, const double* Rt, const double* alpha,
                                  double* u, double* u_prev, const SolverState* state, int64_t N, int S,
                                  int n_c, int n_iter2, int mode, hipStream_t st) {
    constexpr int NV = NU + NU * (NU + 1) / 2;
    const int nstrips = (S + 15) / 16;
    const int NW = (nstrips + kStripsPerWave - 1) / kStripsPerWave;
    const size_t lds = (size_t)2 * NW * NV * 16 * sizeof(double);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_u_phase_mfma<NKC, NU>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int64_t nblk = (N + 15) / 16;
    const int64_t grid = nblk < 768 ? nblk : 768;
    hipLaunchKernelGGL((k_u_phase_mfma<NKC, NU>), dim3((unsigned)grid), dim3(NW * 64), lds, st, V, D, Rt, alpha, u,
                       u_prev, state, N, S, n_c, n_iter2, mode);
    return hipGetLastError();
}

template <int NKC>
static hipError_t launch_u_mfma_nkc(int n_u, const double* V, const double* D, const double* Rt,
                                    const double* alpha, double* u, double* u_prev, const SolverState* state,
                                    int64_t N, int S, int n_c, int n_iter2, int mode, hipStream_t st) {
    switch (n_u) {
#define DMF_CASE(NU_) \
    case NU_: return launch_u_mfma_t<NKC, NU_>(V, D, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, st);
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4) DMF_CASE(5) DMF_CASE(6) DMF_CASE(7) DMF_CASE(8)
#undef DMF_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_u_phase_mfma(const double* V, const double* D, const double* Rt, const double* alpha,
                               double* u, double* u_prev, const SolverState* state, int64_t N, int S, int n_c,
                               int n_u, int n_iter2, int mode, hipStream_t st) {
    switch ((n_c + 3) / 4) {
        case 0: return launch_u_mfma_nkc<0>(n_u, V, D, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, st);
        case 1: return launch_u_mfma_nkc<1>(n_u, V, D, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, st);
        case 2: return launch_u_mfma_nkc<2>(n_u, V, D, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, st);
        case 3: return launch_u_mfma_nkc<3>(n_u, V, D, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, st);
        case 4: return launch_u_mfma_nkc<4>(n_u, V, D, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dmf
