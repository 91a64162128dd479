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
// `Rt` here is the problem's padded copy of R_trunc: row stride 4 * NKC doubles, zero pad columns.
#include <cstdlib>
#include <type_traits>

#include "dmf_device.h"
#include "dmf_internal.h"
#include "dmf_phaseb.h"

namespace dmf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kStripsPerWave = 4;  // 16-sample strips owned by one wave (64 samples)
constexpr int kMfmaMaxWaves = 8;   // S <= 512 on this path

// Broadcast of lane (group base + L)'s value inside aligned groups of NU lanes.  NU = 2 / 4 use a DPP
// quad permute (no LDS traffic); other group sizes go through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double x) {
    // (mov_dpp: every lane of a quad permute has a source, no "old" value to initialise)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int NU, int L>
__device__ __forceinline__ double group_bcast(double x, int lane0) {
    if constexpr (NU == 1) return x;
    else if constexpr (NU == 2) return dpp_quad<(L) | (L << 2) | ((2 + L) << 4) | ((2 + L) << 6)>(x);
    else if constexpr (NU == 4) return dpp_quad<L | (L << 2) | (L << 4) | (L << 6)>(x);
    else return __shfl(x, lane0 + L, 64);
}

template <int NU, int L = 0>
__device__ __forceinline__ double grad_row(double g, double base, const double (&Mrow)[NU], int lane0) {
    if constexpr (L < NU) {
        g = fma(-group_bcast<NU, L>(base, lane0), Mrow[L], g);
        return grad_row<NU, L + 1>(g, base, Mrow, lane0);
    } else {
        return g;
    }
}

// VEC: S % 4 == 0, so a lane's four samples are contiguous, 32-byte aligned and all in range.
// D16T (with VEC): the counts come from the problem's u16 copy (row stride SD) -- 8 instead of 32 bytes per lane and
// strip, and 8 instead of 32 staging registers for the prefetched strips (with 7 or 8 unknowns the f64 form spills).
template <int NKC, int NU, bool VEC, bool D16T>
__global__ __launch_bounds__(512) void k_u_phase_mfma(
    const double* __restrict__ V, const double* __restrict__ D, const unsigned short* __restrict__ D16, int SD,
    const double* __restrict__ Rt,
    const double* __restrict__ alpha, double* __restrict__ u, double* __restrict__ u_prev,
    const SolverState* __restrict__ state, int64_t N, int S, int n_c, int n_iter2, int mode,
    double* __restrict__ cm_out) {
    // cm_out != nullptr: "split" mode for many inner steps -- the per-row c_i / M_i go to cm_out[row][NU + NP]
    // and k_u_inner_rows runs the inner iterations with every lane of the chip busy, instead of one wave per
    // workgroup doing them here while the others wait.
    constexpr int NP = NU * (NU + 1) / 2;
    constexpr int NMT = (NP + 15) / 16;  // 16-row tiles of the pair matrix
    constexpr int NV = NU + NP;
    extern __shared__ double lds_dyn[];  // beta[n_iter2] then red[2][NW][NV][16]
    if (state->done) return;

    const int NW = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int m16 = lane & 15, q = lane >> 4;
    const double* __restrict__ A2 = alpha + (int64_t)n_c * S;
    double* __restrict__ beta_tab = lds_dyn;
    double* __restrict__ red = lds_dyn + (cm_out ? 0 : ((n_iter2 + 1) & ~1));

    // momentum coefficients of the n_iter2 inner steps (deconvolution.py:83-85): same for every row
    if (threadIdx.x == 0 && cm_out == nullptr) {
        double a1 = state->a1, lw_prev = state->l_w_prev;
        const double lw = state->l_w;
        for (int t2 = 0; t2 < n_iter2; ++t2) {
            double beta;
            momentum_step(a1, lw_prev, lw, beta);
            beta_tab[t2] = beta;
            lw_prev = lw;
        }
    }
    const double inv_lw = 1.0 / state->l_w;  // x / l_w as x * (1 / l_w): <= 1 ulp from the division

    // ---- per-wave constant A operands (zero for samples >= S and types >= n_c / n_u) ----------
    double a1op[kStripsPerWave][NKC > 0 ? NKC : 1];
    double a2op[kStripsPerWave][4];
    double pop[kStripsPerWave][NMT][4];
    int col0[kStripsPerWave];  // first of this lane's four samples in strip t (clamped into range)
#pragma unroll
    for (int t = 0; t < kStripsPerWave; ++t) {
        const int s0 = (wave * kStripsPerWave + t) * 16;
        // first product: m <-> sample s0 + 4 (m & 3) + (m >> 2), k <-> known type 4 kc + q
        const int s_e = s0 + 4 * (m16 & 3) + (m16 >> 2);
        const int s_ec = s_e < S ? s_e : S - 1;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int kk = kc * 4 + q;
            const double keep = (kk < n_c && s_e < S) ? -1.0 : 0.0;
            a1op[t][kc] = keep * alpha[(int64_t)(kk < n_c ? kk : 0) * S + s_ec];
        }
        int jp[NMT], lp[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            const int p = mt * 16 + m16;
            int l = 0;
            while ((l + 1) * (l + 2) / 2 <= p) ++l;
            jp[mt] = p < NP ? p - l * (l + 1) / 2 : 0;
            lp[mt] = p < NP ? l : 0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = s0 + 4 * q + r;  // k-step r: k = q <-> sample s0 + 4 q + r
            const int sc = s < S ? s : S - 1;
            const double keep2 = (m16 < NU && s < S) ? 1.0 : 0.0;
            a2op[t][r] = keep2 * A2[(int64_t)(m16 < NU ? m16 : 0) * S + sc];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const double keepp = (mt * 16 + m16 < NP && s < S) ? 1.0 : 0.0;
                pop[t][mt][r] = keepp * (A2[(int64_t)jp[mt] * S + sc] * A2[(int64_t)lp[mt] * S + sc]);
            }
        }
        const int c = s0 + 4 * q;
        col0[t] = VEC ? (c < S ? c : 0) : c;
    }
    __syncthreads();  // beta_tab visible

    const int64_t nblk = (N + 15) / 16;
    // Loads are unconditional from clamped addresses: out-of-range samples meet zero A operands and
    // out-of-range rows only feed output columns that are never read, so no masking is needed.
    using DStage = std::conditional_t<D16T, unsigned long long, v4d>;  // a strip's four counts as staged for the next block
    auto load_strip = [&](int64_t rowc, int t, v4d& e, DStage& d) {
        const double* __restrict__ vp = V + rowc * S;
        if constexpr (D16T) {
            static_assert(!D16T || VEC, "the u16 path reads four samples with one 8-byte load");
            const v2d v01 = *reinterpret_cast<const v2d*>(vp + col0[t]);
            const v2d v23 = *reinterpret_cast<const v2d*>(vp + col0[t] + 2);
            e = v4d{v01.x, v01.y, v23.x, v23.y};
            d = *reinterpret_cast<const unsigned long long*>(D16 + rowc * SD + col0[t]);
        } else {
            const double* __restrict__ dp = D + rowc * S;
            if constexpr (VEC) {
                const v2d v01 = *reinterpret_cast<const v2d*>(vp + col0[t]);
                const v2d v23 = *reinterpret_cast<const v2d*>(vp + col0[t] + 2);
                const v2d d01 = *reinterpret_cast<const v2d*>(dp + col0[t]);
                const v2d d23 = *reinterpret_cast<const v2d*>(dp + col0[t] + 2);
                e = v4d{v01.x, v01.y, v23.x, v23.y};
                d = v4d{d01.x, d01.y, d23.x, d23.y};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = col0[t] + r < S ? col0[t] + r : S - 1;
                    e[r] = vp[c];
                    d[r] = dp[c];
                }
            }
        }
    };
    auto counts_of = [&](const DStage& d) -> v4d {
        if constexpr (D16T) {
            const unsigned int lo = (unsigned int)d, hi = (unsigned int)(d >> 32);
            return v4d{(double)(lo & 0xFFFFu), (double)(lo >> 16), (double)(hi & 0xFFFFu), (double)(hi >> 16)};
        } else {
            return d;
        }
    };
    auto row_of = [&](int64_t blk) {
        const int64_t row = blk * 16 + m16;
        return row < N ? row : N - 1;
    };

    v4d nv[kStripsPerWave];
    DStage nd[kStripsPerWave];
    double nrt[NKC > 0 ? NKC : 1];
    {
        const int64_t rowc = row_of(blockIdx.x);
#pragma unroll
        for (int t = 0; t < kStripsPerWave; ++t) load_strip(rowc, t, nv[t], nd[t]);
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) nrt[kc] = Rt[rowc * (4 * NKC) + kc * 4 + q];
    }

    int it = 0;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x, ++it) {
        const int64_t row0 = blk * 16;
        const int64_t nxt = blk + gridDim.x < nblk ? blk + gridDim.x : blk;
        const int64_t rowc_n = row_of(nxt);

        // the wave that will run this block's inner iterations fetches its u / u_ now (first pass)
        constexpr int RPW = 64 / NU;  // rows per pass of the inner-iteration phase
        const int rl = lane / NU, j = lane - rl * NU;
        const bool my_turn = cm_out == nullptr && wave == it % NW;
        const bool ok0 = rl < RPW && rl < 16 && row0 + rl < N;
        double uu0 = 0.0, up0 = 0.0;
        if (my_turn) {
            const int64_t gi0 = ok0 ? (row0 + rl) * NU + j : 0;
            uu0 = u[gi0];
            up0 = u_prev[gi0];
        }

        double rtop[NKC > 0 ? NKC : 1];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            rtop[kc] = nrt[kc];  // B operand of the first product: Rt^T[k = 4 kc + q][n = row]
            nrt[kc] = Rt[rowc_n * (4 * NKC) + kc * 4 + q];
        }
        v4d cacc = {0.0, 0.0, 0.0, 0.0};
        v4d macc[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) macc[mt] = cacc;

#pragma unroll
        for (int t = 0; t < kStripsPerWave; ++t) {
            v4d e = nv[t];
            const v4d d = counts_of(nd[t]);
            load_strip(rowc_n, t, nv[t], nd[t]);  // prefetch the next row block's strip t ...
            __builtin_amdgcn_sched_barrier(0);    // ... and keep it in front of this strip's MFMAs
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

        if (cm_out != nullptr) {  // split mode: fixed-order sum over the waves, one (value, row) per thread
            const double* __restrict__ all = red + ((size_t)(it & 1) * NW * NV) * 16;
            for (int e = threadIdx.x; e < NV * 16; e += blockDim.x) {
                const int v = e >> 4, r = e & 15;
                double tot = 0.0;
                for (int w = 0; w < NW; ++w) tot += all[((size_t)w * NV + v) * 16 + r];
                if (row0 + r < N) cm_out[(row0 + r) * NV + v] = tot;
            }
        }
        // ---- row-local inner iterations by one wave (round robin), lane = (row, unknown j)
        if (my_turn) {
            const double* __restrict__ all = red + ((size_t)(it & 1) * NW * NV) * 16;
            const int lane0 = lane - j;
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
                double uu = pass0 == 0 ? uu0 : u[gi];
                double up = pass0 == 0 ? up0 : u_prev[gi];
                for (int t2 = 0; t2 < n_iter2; ++t2) {
                    const double beta = beta_tab[t2];
                    const double ut = uu + beta * (uu - up);
                    const double base = mode == 1 ? uu : ut;  // deconvolution.py:163 vs :88
                    up = uu;
                    const double g = grad_row<NU>(cj, base, Mrow, lane0);
                    uu = fmin(fmax(fma(g, inv_lw, ut), 0.0), 1.0);
                }
                if (ok) {
                    u[gi] = uu;
                    u_prev[gi] = up;
                }
            }
        }
    }
}

bool u_phase_mfma_supported(int S, int n_c, int n_u) {
    return n_u >= 1 && n_u <= 8 && n_c <= 16 && S <= 16 * kStripsPerWave * kMfmaMaxWaves;
}

template <int NKC, int NU>
static hipError_t launch_u_mfma_t(const double* V, const double* D, const unsigned short* D16, int SD, const double* Rt,
                                  const double* alpha, double* u, double* u_prev, const SolverState* state, int64_t N,
                                  int S, int n_c, int n_iter2, int mode, double* cm_out, hipStream_t st) {
    constexpr int NV = NU + NU * (NU + 1) / 2;
    const int nstrips = (S + 15) / 16;
    const int NW = (nstrips + kStripsPerWave - 1) / kStripsPerWave;
    // (split mode keeps no momentum table in LDS)
    const size_t lds = ((size_t)(cm_out ? 0 : ((n_iter2 + 1) & ~1)) + (size_t)2 * NW * NV * 16) * sizeof(double);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    const bool vec = (S & 3) == 0;
    const bool d16 = vec && D16 != nullptr && (SD & 3) == 0;
    const void* fn = d16   ? (const void*)k_u_phase_mfma<NKC, NU, true, true>
                     : vec ? (const void*)k_u_phase_mfma<NKC, NU, true, false>
                           : (const void*)k_u_phase_mfma<NKC, NU, false, false>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int64_t nblk = (N + 15) / 16;
    // persistent workgroups: as many as fit two waves per SIMD (the kernel needs ~250 registers) -- 3 per CU at
    // NW = 2 left a quarter of the wave slots empty
    int per_cu = NW >= 8 ? 1 : 8 / NW;
#ifdef DMF_EXPERIMENT  // (an experiment build only: DMF_EXPERIMENT=1 python -m demethify_amd._build)
    if (const char* v = getenv("DMF_UMFMA_PER_CU")) per_cu = atoi(v) > 0 ? atoi(v) : per_cu;  // (experiments)
#endif
    const int64_t cap = (int64_t)256 * per_cu;
    const int64_t grid = nblk < cap ? nblk : cap;
    if (d16)
        hipLaunchKernelGGL((k_u_phase_mfma<NKC, NU, true, true>), dim3((unsigned)grid), dim3(NW * 64), lds, st, V, D, D16, SD,
                           Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, cm_out);
    else if (vec)
        hipLaunchKernelGGL((k_u_phase_mfma<NKC, NU, true, false>), dim3((unsigned)grid), dim3(NW * 64), lds, st, V, D, D16, SD,
                           Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, cm_out);
    else
        hipLaunchKernelGGL((k_u_phase_mfma<NKC, NU, false, false>), dim3((unsigned)grid), dim3(NW * 64), lds, st, V, D, D16,
                           SD, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, cm_out);
    return hipGetLastError();
}

template <int NKC>
static hipError_t launch_u_mfma_nkc(int n_u, const double* V, const double* D, const unsigned short* D16, int SD,
                                    const double* Rt, const double* alpha, double* u, double* u_prev,
                                    const SolverState* state, int64_t N, int S, int n_c, int n_iter2, int mode,
                                    double* cm_out, hipStream_t st) {
    switch (n_u) {
#define DMF_CASE(NU_) \
    case NU_: return launch_u_mfma_t<NKC, NU_>(V, D, D16, SD, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, cm_out, st);
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4) DMF_CASE(5) DMF_CASE(6) DMF_CASE(7) DMF_CASE(8)
#undef DMF_CASE
        default: return hipErrorInvalidValue;
    }
}

static hipError_t launch_u_phase_mfma_impl(const double* V, const double* D, const unsigned short* D16, int SD,
                                           const double* Rt, const double* alpha, double* u, double* u_prev,
                                           const SolverState* state, int64_t N, int S, int n_c, int n_u, int n_iter2,
                                           int mode, double* cm_out, hipStream_t st) {
    switch ((n_c + 3) / 4) {
        case 0: return launch_u_mfma_nkc<0>(n_u, V, D, D16, SD, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, cm_out, st);
        case 1: return launch_u_mfma_nkc<1>(n_u, V, D, D16, SD, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, cm_out, st);
        case 2: return launch_u_mfma_nkc<2>(n_u, V, D, D16, SD, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, cm_out, st);
        case 3: return launch_u_mfma_nkc<3>(n_u, V, D, D16, SD, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, cm_out, st);
        case 4: return launch_u_mfma_nkc<4>(n_u, V, D, D16, SD, Rt, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, cm_out, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_u_phase_mfma(const double* V, const double* D, const unsigned short* D16, int SD, const double* Rt,
                               const double* alpha, double* u, double* u_prev, const SolverState* state, int64_t N, int S,
                               int n_c, int n_u, int n_iter2, int mode, hipStream_t st) {
    return launch_u_phase_mfma_impl(V, D, D16, SD, Rt, alpha, u, u_prev, state, N, S, n_c, n_u, n_iter2, mode, nullptr, st);
}

// ---- split mode for many inner steps (the CLI default under --purity is 500): c_i / M_i per row through HBM
// (N x (n_u + n_u (n_u + 1) / 2) doubles, small next to V and D), then the inner iterations with lane = (row, j)
// over the whole chip.
__global__ void k_beta_table(const SolverState* __restrict__ state, int n_iter2, double* __restrict__ beta_out) {
    if (state->done) return;
    double a1 = state->a1, lw_prev = state->l_w_prev;
    const double lw = state->l_w;
    for (int t2 = 0; t2 < n_iter2; ++t2) {  // deconvolution.py:83-85
        double beta;
        momentum_step(a1, lw_prev, lw, beta);
        beta_out[t2] = beta;
        lw_prev = lw;
    }
}

constexpr int kBetaChunk = 6144;  // momentum coefficients held in LDS at a time (48 KB)

template <int NU>
__global__ __launch_bounds__(256) void k_u_inner_rows(const double* __restrict__ cm, const double* __restrict__ beta_g,
                                                      double* __restrict__ u, double* __restrict__ u_prev,
                                                      const SolverState* __restrict__ state, int64_t N, int n_iter2,
                                                      int mode) {
    constexpr int NP = NU * (NU + 1) / 2, NV = NU + NP;
    constexpr int RPW = 64 / NU;  // rows per wave
    extern __shared__ double beta_tab[];
    if (state->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rl = lane / NU, j = lane - rl * NU, lane0 = lane - j;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + rl;
    const bool ok = rl < RPW && row < N;
    const int64_t rowc = ok ? row : 0;
    const double inv_lw = 1.0 / state->l_w;  // as in k_u_phase_mfma
    const double* __restrict__ mine = cm + rowc * NV;
    // c_j / l_w and -M_jl / l_w: the step is then one multiply-add chain whose last link clamps to [0, 1] (VOP3 clamp),
    // the form of the row pass's phase B (15 instead of 18 vector instructions per step at four unknowns; at the purity
    // mode's 500 steps and 1e6 rows the kernel takes 1.15 ms either way).
    const double cjs = mine[j] * inv_lw;
    double Ms[NU];
#pragma unroll
    for (int l = 0; l < NU; ++l) Ms[l] = -inv_lw * mine[NU + (l <= j ? tri(l, j) : tri(j, l))];
    const int64_t gi = rowc * NU + j;
    double uu = u[gi], up = u_prev[gi];
    // the momentum coefficients pass through LDS kBetaChunk at a time: any n_iter2 runs (the reference has no limit)
    for (int t0 = 0; t0 < n_iter2; t0 += kBetaChunk) {
        const int nt = n_iter2 - t0 < kBetaChunk ? n_iter2 - t0 : kBetaChunk;
        if (t0 > 0) __syncthreads();  // the previous chunk has been consumed by every wave
        for (int t = threadIdx.x; t < nt; t += 256) beta_tab[t] = beta_g[t0 + t];
        __syncthreads();
        for (int t2 = 0; t2 < nt; ++t2) {
            const double beta = beta_tab[t2];
            const double ut = uu + beta * (uu - up);
            const double base = mode == 1 ? uu : ut;  // deconvolution.py:163 vs :88
            up = uu;
            uu = f_step_chain<NU>(ut + cjs, base, Ms, lane0);  // clip(ut + (c_j - sum_l M_jl x_l) / l_w, 0, 1)
        }
    }
    if (ok) {
        u[gi] = uu;
        u_prev[gi] = up;
    }
}

// The same inner iterations with ONE CpG row per 16-lane DPP row (lane j < NU of the row holds unknown j): the gradient
// needs lane l's value in every lane of the row, which v_fmac_f64_dpp row_newbcast:l delivers inside the multiply-add --
// NU instructions per step where the group form above pays 2 NU ds_bpermute round trips (k_u_inner_rows<8>: 239 us at
// 5e5 rows and 20 steps; this form: see DESIGN.md).  Same per-row arithmetic order: g = c_j - sum_l M_jl x_l, l ascending.
// (a DPP source written by the previous vector instruction needs two wait states: only the first multiply-add of a
// gradient follows the instruction that produced x)
template <int L>
__device__ __forceinline__ void fmac_row16(double& acc, double x, double m) {
    if constexpr (L == 0)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(x), "v"(m), "n"(L));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(x), "v"(m), "n"(L));
}
template <int NU, int L = 0>
__device__ __forceinline__ void grad_row16(double& g, double base, const double (&Mneg)[NU]) {
    if constexpr (L < NU) {
        fmac_row16<L>(g, base, Mneg[L]);
        grad_row16<NU, L + 1>(g, base, Mneg);
    }
}

template <int NU>
__global__ __launch_bounds__(256) void k_u_inner_rows16(const double* __restrict__ cm, const double* __restrict__ beta_g,
                                                        double* __restrict__ u, double* __restrict__ u_prev,
                                                        const SolverState* __restrict__ state, int64_t N, int n_iter2,
                                                        int mode) {
    static_assert(NU >= 1 && NU <= 16, "one row per DPP row");
    constexpr int NP = NU * (NU + 1) / 2, NV = NU + NP;
    extern __shared__ double beta_tab[];
    if (state->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 4 + (lane >> 4);
    const bool ok = j < NU && row < N;
    const int64_t rowc = row < N ? row : 0;
    const int jc = j < NU ? j : 0;
    const double inv_lw = 1.0 / state->l_w;  // as in k_u_phase_mfma
    const double* __restrict__ mine = cm + rowc * NV;
    const double cj = mine[jc];
    double Mneg[NU];
#pragma unroll
    for (int l = 0; l < NU; ++l) Mneg[l] = -mine[NU + (l <= jc ? tri(l, jc) : tri(jc, l))];
    const int64_t gi = rowc * NU + jc;
    double uu = ok ? u[gi] : 0.0, up = ok ? u_prev[gi] : 0.0;
    for (int t0 = 0; t0 < n_iter2; t0 += kBetaChunk) {
        const int nt = n_iter2 - t0 < kBetaChunk ? n_iter2 - t0 : kBetaChunk;
        if (t0 > 0) __syncthreads();  // the previous chunk has been consumed by every wave
        for (int t = threadIdx.x; t < nt; t += 256) beta_tab[t] = beta_g[t0 + t];
        __syncthreads();
        for (int t2 = 0; t2 < nt; ++t2) {
            const double beta = beta_tab[t2];
            const double ut = uu + beta * (uu - up);
            const double base = mode == 1 ? uu : ut;  // deconvolution.py:163 vs :88
            up = uu;
            double g = cj;
            grad_row16<NU>(g, base, Mneg);
            uu = fmin(fmax(fma(g, inv_lw, ut), 0.0), 1.0);
        }
    }
    if (ok) {
        u[gi] = uu;
        u_prev[gi] = up;
    }
}

// More than 16 unknowns: one CpG row per 32 lanes = two DPP rows (lane j < 16 of the first holds unknown j, of the second
// unknown 16 + j).  Each lane keeps its own iterate value and, exchanged once per step, its partner's 16 lanes away; the
// gradient's first 16 terms broadcast from the half that holds unknowns 0..15 (for the first DPP row that is the lane's own
// value, for the second the partner's), the rest from the other -- row_newbcast within each DPP row, as above.
template <int NU, int L = 0>
__device__ __forceinline__ void grad_row32_lo(double& g, double x, const double (&Mneg)[NU]) {
    if constexpr (L < 16) {
        fmac_row16<L>(g, x, Mneg[L]);
        grad_row32_lo<NU, L + 1>(g, x, Mneg);
    }
}
template <int NU, int L = 16>
__device__ __forceinline__ void grad_row32_hi(double& g, double x, const double (&Mneg)[NU]) {
    if constexpr (L < NU) {
        // (the first of these follows the instruction that selected x: fmac_row16<0> carries the wait states)
        if constexpr (L == 16) fmac_row16<0>(g, x, Mneg[L]);
        else fmac_row16<L - 16>(g, x, Mneg[L]);
        grad_row32_hi<NU, L + 1>(g, x, Mneg);
    }
}

template <int NU>
__global__ __launch_bounds__(256) void k_u_inner_rows32(const double* __restrict__ cm, const double* __restrict__ beta_g,
                                                        double* __restrict__ u, double* __restrict__ u_prev,
                                                        const SolverState* __restrict__ state, int64_t N, int n_iter2,
                                                        int mode) {
    static_assert(NU > 16 && NU <= 32, "one row per two DPP rows");
    constexpr int NP = NU * (NU + 1) / 2, NV = NU + NP;
    extern __shared__ double beta_tab[];
    if (state->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31;            // unknown of this lane
    const bool upper = (lane & 16) != 0;  // second DPP row of the CpG row: unknowns 16..31
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 2 + (lane >> 5);
    const bool ok = j < NU && row < N;
    const int64_t rowc = row < N ? row : 0;
    const int jc = j < NU ? j : 0;
    const double inv_lw = 1.0 / state->l_w;  // as in k_u_phase_mfma
    const double* __restrict__ mine = cm + rowc * NV;
    const double cj = mine[jc];
    double Mneg[NU];
#pragma unroll
    for (int l = 0; l < NU; ++l) Mneg[l] = j < NU ? -mine[NU + (l <= jc ? tri(l, jc) : tri(jc, l))] : 0.0;
    const int64_t gi = rowc * NU + jc;
    double uu = ok ? u[gi] : 0.0, up = ok ? u_prev[gi] : 0.0;
    for (int t0 = 0; t0 < n_iter2; t0 += kBetaChunk) {
        const int nt = n_iter2 - t0 < kBetaChunk ? n_iter2 - t0 : kBetaChunk;
        if (t0 > 0) __syncthreads();  // the previous chunk has been consumed by every wave
        for (int t = threadIdx.x; t < nt; t += 256) beta_tab[t] = beta_g[t0 + t];
        __syncthreads();
        for (int t2 = 0; t2 < nt; ++t2) {
            const double beta = beta_tab[t2];
            const double ut = uu + beta * (uu - up);
            const double base = mode == 1 ? uu : ut;  // deconvolution.py:163 vs :88
            up = uu;
            const double other = __shfl_xor(base, 16, 64);  // the partner lane's value (lanes >= NU hold 0)
            const double x_lo = upper ? other : base, x_hi = upper ? base : other;
            double g = cj;
            grad_row32_lo<NU>(g, x_lo, Mneg);
            grad_row32_hi<NU>(g, x_hi, Mneg);
            uu = fmin(fmax(fma(g, inv_lw, ut), 0.0), 1.0);
            if (j >= NU) uu = 0.0;
        }
    }
    if (ok) {
        u[gi] = uu;
        u_prev[gi] = up;
    }
}

// The inner iterations AND b_u = u^T (D * V) of the rows just finished, in one launch (wide row groups on u16 counts; the
// integer Gram route needs b_u from a stream over V and the counts, k_bu_cols2 as a kernel of its own).  The inner
// iterations are a chain of dependent FP64 instructions with next to no memory traffic, the b_u stream is all memory
// traffic: a workgroup alternates between them on chunks of 16 NSG CpG rows --
//   * all loads of the chunk are issued up front: the rows' c / M (lane = (row, unknown), one row per DPP row as in
//     k_u_inner_rows16, four rows per wave), then the V / count pieces of the b_u stream (lane = two adjacent samples,
//     4 NSG rows per wave), which land while the chains run;
//   * the finished rows go to HBM and into an LDS tile (double buffered: one barrier per chunk);
//   * b_u accumulates per lane over all chunks of the (persistent) workgroup, is summed over the waves at the end and
//     written as one slab per workgroup (layout of k_bu_cols2); the workgroup's share of ||u||_F^2 goes to u2_partials.
// NSG = 128-sample groups (S <= 128 NSG); a workgroup has 4 NSG waves.  Small chunks keep the register count low
// (n_u = 12: ~130): what hides the chains and the load latency is the number of resident workgroups.
constexpr int kInnerBuMaxSteps = 1024;

template <int NU, int NSG, bool ODD>
__global__ __launch_bounds__(256 * NSG) void k_inner_bu(const double* __restrict__ cm, const double* __restrict__ beta_g,
                                                        double* __restrict__ u, double* __restrict__ u_prev,
                                                        const SolverState* __restrict__ state,
                                                        const double* __restrict__ V,
                                                        const unsigned short* __restrict__ D16, int SD, int64_t N, int S,
                                                        int n_iter2, int mode, double* __restrict__ slab,
                                                        double* __restrict__ u2_partials) {
    static_assert(NU >= 1 && NU <= 16 && (NSG == 1 || NSG == 2), "one row per DPP row; S <= 256");
    constexpr int NP = NU * (NU + 1) / 2, NV = NU + NP;
    constexpr int NWV = 4 * NSG, kChunk = 16 * NSG, kRows = 4 * NSG;
    constexpr int US = NU + (NU & 1);  // row stride of the LDS tile of finished rows (even: 16-byte reads of pairs)
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    extern __shared__ double lds_ib[];
    double* __restrict__ beta_tab = lds_ib;
    double* __restrict__ u_s = lds_ib + ((n_iter2 + 1) & ~1);  // [2][kChunk][US]
    double* __restrict__ red = u_s + 2 * kChunk * US;          // [NSG][NU][2][64], then [NWV] for ||u||^2
    if (state->done) return;
    for (int t = threadIdx.x; t < n_iter2; t += 256 * NSG) beta_tab[t] = beta_g[t];
    const double inv_lw = 1.0 / state->l_w;  // as in k_u_phase_mfma
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane & 15, sub = lane >> 4, jc = j < NU ? j : 0;
    const int sg = wave >> 2, wr = wave & 3;
    const int s = sg * 128 + 2 * lane;
    const bool active = s < S;
    // odd S: the row's last sample sits alone in its lane -- its V comes as the upper half of the pair one element lower
    // (never past the end of the row), its partner's count is zero padding; 16-byte loads from 8-byte-aligned addresses:
    // tools/align_probe.hip
    const bool lone = ODD && s == S - 1;
    const int sc = ODD ? (lone ? S - 2 : (active ? s : 0)) : (active ? s : S - 2);
    const int sd = ODD ? (active ? s : 0) : sc;
    typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));
    double acc[NU][2];
#pragma unroll
    for (int l = 0; l < NU; ++l) acc[l][0] = acc[l][1] = 0.0;
    double u2 = 0.0;
    __syncthreads();

    const int64_t nchunks = (N + kChunk - 1) / kChunk;
    int it = 0;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x, ++it) {
        const int64_t row0 = chunk * kChunk;
        // ---- loads: the chain's operands first (waited for first), then the b_u stream's pieces
        const int64_t row = row0 + wave * 4 + sub;
        const bool ok = j < NU && row < N;
        const int64_t rowc = row < N ? row : 0;
        const double* __restrict__ mine = cm + rowc * NV;
        const double cj = mine[jc];
        double Mneg[NU];
#pragma unroll
        for (int l = 0; l < NU; ++l) Mneg[l] = -mine[NU + (l <= jc ? tri(l, jc) : tri(jc, l))];
        const int64_t gi = rowc * NU + jc;
        double uu = ok ? u[gi] : 0.0, up = ok ? u_prev[gi] : 0.0;
        v2d_t vv[kRows];
        unsigned int dd[kRows];
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const int64_t r = row0 + kRows * wr + x;
            const int64_t rc = r < N ? r : N - 1;
            if constexpr (ODD) {
                dd[x] = r < N && active ? *reinterpret_cast<const unsigned int*>(D16 + rc * SD + sd) : 0u;
                const v2d_u vl = *reinterpret_cast<const v2d_u*>(V + rc * S + sc);
                vv[x] = v2d_t{lone ? vl.y : vl.x, vl.y};
            } else {  // (lanes past S accumulate sums nobody reads)
                dd[x] = r < N ? *reinterpret_cast<const unsigned int*>(D16 + rc * SD + sd) : 0u;
                vv[x] = *reinterpret_cast<const v2d_t*>(V + rc * S + sc);
            }
        }
        // ---- the chunk's inner iterations (same arithmetic as k_u_inner_rows16)
        for (int t2 = 0; t2 < n_iter2; ++t2) {
            const double beta = beta_tab[t2];
            const double ut = uu + beta * (uu - up);
            const double base = mode == 1 ? uu : ut;  // deconvolution.py:163 vs :88
            up = uu;
            double g = cj;
            grad_row16<NU>(g, base, Mneg);
            uu = fmin(fmax(fma(g, inv_lw, ut), 0.0), 1.0);
        }
        double* __restrict__ tile = u_s + (it & 1) * kChunk * US;
        if (ok) {
            u[gi] = uu;
            u_prev[gi] = up;
            u2 = fma(uu, uu, u2);
        }
        if (j < NU) tile[(wave * 4 + sub) * US + j] = ok ? uu : 0.0;
        __syncthreads();  // the tile is complete (and, double buffered, not rewritten before every wave has read it)
        // ---- b_u of this wave's rows
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const double t0 = (double)(dd[x] & 0xFFFFu) * vv[x].x;
            const double t1 = (double)(dd[x] >> 16) * vv[x].y;
            const double* __restrict__ ur = tile + (kRows * wr + x) * US;
#pragma unroll
            for (int l = 0; l + 1 < NU; l += 2) {
                const v2d_t u01 = *reinterpret_cast<const v2d_t*>(ur + l);
                acc[l][0] = fma(t0, u01.x, acc[l][0]);
                acc[l][1] = fma(t1, u01.x, acc[l][1]);
                acc[l + 1][0] = fma(t0, u01.y, acc[l + 1][0]);
                acc[l + 1][1] = fma(t1, u01.y, acc[l + 1][1]);
            }
            if constexpr (NU & 1) {
                const double ul = ur[NU - 1];
                acc[NU - 1][0] = fma(t0, ul, acc[NU - 1][0]);
                acc[NU - 1][1] = fma(t1, ul, acc[NU - 1][1]);
            }
        }
    }
    // ---- the workgroup's slab: waves of a sample group summed in wave order
    double* __restrict__ part = red + (size_t)sg * NU * 2 * 64;
    for (int r = 1; r < 4; ++r) {
        __syncthreads();
        if (wr == r) {
#pragma unroll
            for (int l = 0; l < NU; ++l) {
                part[(l * 2 + 0) * 64 + lane] = acc[l][0];
                part[(l * 2 + 1) * 64 + lane] = acc[l][1];
            }
        }
        __syncthreads();
        if (wr == 0) {
#pragma unroll
            for (int l = 0; l < NU; ++l) {
                acc[l][0] += part[(l * 2 + 0) * 64 + lane];
                acc[l][1] += part[(l * 2 + 1) * 64 + lane];
            }
        }
    }
    if (wr == 0 && active) {
#pragma unroll
        for (int l = 0; l < NU; ++l) {
            double* __restrict__ out = slab + ((int64_t)blockIdx.x * NU + l) * S + s;
            out[0] = acc[l][0];
            if (!lone) out[1] = acc[l][1];
        }
    }
    __syncthreads();
    u2 = wave_sum(u2);
    double* __restrict__ u2w = red;  // (the b_u sums have been consumed)
    if (lane == 0) u2w[wave] = u2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) tot += u2w[w];
        u2_partials[blockIdx.x] = tot;
    }
}

bool u_inner_bu_supported(const double* V, int S, int SD, int n_u, int n_iter2) {
    return n_u >= 1 && n_u <= 16 && S >= 2 && S <= 256 && (SD & 1) == 0 && n_iter2 <= kInnerBuMaxSteps &&
           (reinterpret_cast<uintptr_t>(V) & 7) == 0;
}

int u_inner_bu_grid(int64_t N, int S) {
    const int64_t nchunks = S <= 128 ? (N + 15) / 16 : (N + 31) / 32;
    const int64_t cap = S <= 128 ? 2048 : 1024;  // up to eight (four) workgroups of four (eight) waves per CU
    return (int)(nchunks < cap ? (nchunks < 1 ? 1 : nchunks) : cap);
}

// cm + beta as launch_u_inner; slab: u_inner_bu_grid(N, S) x n_u x S doubles; u2_partials: one double per workgroup
static hipError_t launch_u_inner_bu(const double* cm, double* beta, double* u, double* u_prev, const SolverState* state,
                                    const double* V, const unsigned short* D16, int SD, int64_t N, int S, int n_u,
                                    int n_iter2, int mode, double* slab, double* u2_partials, int* grid_out,
                                    hipStream_t st) {
    if (!u_inner_bu_supported(V, S, SD, n_u, n_iter2)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_beta_table, dim3(1), dim3(1), 0, st, state, n_iter2, beta);
    const int grid = u_inner_bu_grid(N, S);
    *grid_out = grid;
    const int nsg = S <= 128 ? 1 : 2;
    const int us = n_u + (n_u & 1);
    const size_t lds = ((size_t)((n_iter2 + 1) & ~1) + (size_t)2 * 16 * nsg * us + (size_t)nsg * n_u * 2 * 64) * sizeof(double);
#define DMF_LAUNCH(NU_, NSG_, ODD_)                                                                                      \
    hipLaunchKernelGGL((k_inner_bu<NU_, NSG_, ODD_>), dim3((unsigned)grid), dim3(256 * NSG_), lds, st, cm, beta, u, u_prev, \
                       state, V, D16, SD, N, S, n_iter2, mode, slab, u2_partials)
#define DMF_CASE(NU_)                                     \
    case NU_:                                             \
        if (nsg == 1) {                                   \
            if (S & 1) DMF_LAUNCH(NU_, 1, true);          \
            else DMF_LAUNCH(NU_, 1, false);               \
        } else {                                          \
            if (S & 1) DMF_LAUNCH(NU_, 2, true);          \
            else DMF_LAUNCH(NU_, 2, false);               \
        }                                                 \
        break;
    switch (n_u) {
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4)  // (narrow row groups behind the producer: more than 16 known types)
        DMF_CASE(5) DMF_CASE(6) DMF_CASE(7) DMF_CASE(8) DMF_CASE(9) DMF_CASE(10) DMF_CASE(11) DMF_CASE(12) DMF_CASE(13)
        DMF_CASE(14) DMF_CASE(15) DMF_CASE(16)
        default: return hipErrorInvalidValue;
    }
#undef DMF_CASE
#undef DMF_LAUNCH
    return hipGetLastError();
}

int64_t u_phase_split_cm_doubles(int64_t N, int n_u) { return N * (n_u + (int64_t)n_u * (n_u + 1) / 2); }

// the inner iterations from cm (N x (n_u + NP) doubles); beta: n_iter2 doubles of device scratch
static hipError_t launch_u_inner(const double* cm, double* beta, double* u, double* u_prev, const SolverState* state,
                                 int64_t N, int n_u, int n_iter2, int mode, hipStream_t st) {
    hipLaunchKernelGGL(k_beta_table, dim3(1), dim3(1), 0, st, state, n_iter2, beta);
    const size_t lds = (size_t)(n_iter2 < kBetaChunk ? n_iter2 : kBetaChunk) * sizeof(double);
#define DMF_CASE(NU_)                                                                                          \
    case NU_: {                                                                                                \
        const int64_t rows_per_block = 4 * (64 / NU_);                                                         \
        const int64_t grid = (N + rows_per_block - 1) / rows_per_block;                                        \
        hipLaunchKernelGGL((k_u_inner_rows<NU_>), dim3((unsigned)grid), dim3(256), lds, st, cm, beta, u, u_prev, \
                           state, N, n_iter2, mode);                                                           \
        break;                                                                                                 \
    }
#define DMF_CASE16(NU_)                                                                                          \
    case NU_: {                                                                                                  \
        const int64_t grid = (N + 15) / 16; /* 4 waves x 4 rows */                                               \
        hipLaunchKernelGGL((k_u_inner_rows16<NU_>), dim3((unsigned)grid), dim3(256), lds, st, cm, beta, u, u_prev, \
                           state, N, n_iter2, mode);                                                             \
        break;                                                                                                   \
    }
    switch (n_u) {
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4) DMF_CASE16(5) DMF_CASE16(6) DMF_CASE16(7) DMF_CASE16(8)
        DMF_CASE16(9) DMF_CASE16(10) DMF_CASE16(11) DMF_CASE16(12) DMF_CASE16(13) DMF_CASE16(14) DMF_CASE16(15) DMF_CASE16(16)
#define DMF_CASE32(NU_)                                                                                          \
    case NU_: {                                                                                                  \
        const int64_t grid = (N + 7) / 8; /* 4 waves x 2 rows */                                                 \
        hipLaunchKernelGGL((k_u_inner_rows32<NU_>), dim3((unsigned)grid), dim3(256), lds, st, cm, beta, u, u_prev, \
                           state, N, n_iter2, mode);                                                             \
        break;                                                                                                   \
    }
        DMF_CASE32(17) DMF_CASE32(18) DMF_CASE32(19) DMF_CASE32(20) DMF_CASE32(21) DMF_CASE32(22) DMF_CASE32(23) DMF_CASE32(24)
        DMF_CASE32(25) DMF_CASE32(26) DMF_CASE32(27) DMF_CASE32(28) DMF_CASE32(29) DMF_CASE32(30) DMF_CASE32(31) DMF_CASE32(32)
#undef DMF_CASE32
        default: return hipErrorInvalidValue;
    }
#undef DMF_CASE
#undef DMF_CASE16
    return hipGetLastError();
}

// cm: N x (n_u + NP) doubles, beta: n_iter2 doubles (both device scratch owned by the caller)
hipError_t launch_u_phase_split(const double* V, const double* D, const unsigned short* D16, int SD, const double* Rt,
                                const double* alpha, double* u, double* u_prev, const SolverState* state, int64_t N, int S,
                                int n_c, int n_u, int n_iter2, int mode, double* cm, double* beta, hipStream_t st) {
    if (cm == nullptr || beta == nullptr) return hipErrorInvalidValue;
    hipError_t e = launch_u_phase_mfma_impl(V, D, D16, SD, Rt, alpha, u, u_prev, state, N, S, n_c, n_u, n_iter2, mode, cm, st);
    if (e != hipSuccess) return e;
    return launch_u_inner(cm, beta, u, u_prev, state, N, n_u, n_iter2, mode, st);
}

// the same with the integer-matrix-core producer of dmf_kernels_cm_i8.hip (n_u <= 16; its preconditions are the caller's)
hipError_t launch_u_phase_split_i8(const double* V, const unsigned short* D16, int SD, int ND, const double* Rt,
                                   const double* alpha, double* u, double* u_prev, const SolverState* state, int64_t N,
                                   int S, int n_c, int n_u, int n_iter2, int mode, double* cm, double* beta,
                                   hipStream_t st) {
    if (cm == nullptr || beta == nullptr) return hipErrorInvalidValue;
    hipError_t e = launch_cm_i8(V, D16, SD, ND, Rt, alpha, state, N, S, n_c, n_u, cm, st);
    if (e != hipSuccess) return e;
    return launch_u_inner(cm, beta, u, u_prev, state, N, n_u, n_iter2, mode, st);
}

// producer of dmf_kernels_cm_i8.hip, then the inner iterations fused with the b_u stream (k_inner_bu): the whole u phase
// plus b_u and ||u||^2 of the integer Gram route in two launches (+ the momentum table)
hipError_t launch_u_phase_split_i8_bu(const double* V, const unsigned short* D16, int SD, int ND, const double* Rt,
                                      const double* alpha, double* u, double* u_prev, const SolverState* state, int64_t N,
                                      int S, int n_c, int n_u, int n_iter2, int mode, double* cm, double* beta,
                                      double* slab, double* u2_partials, int* grid_out, hipStream_t st) {
    if (cm == nullptr || beta == nullptr || slab == nullptr || u2_partials == nullptr) return hipErrorInvalidValue;
    hipError_t e = launch_cm_i8(V, D16, SD, ND, Rt, alpha, state, N, S, n_c, n_u, cm, st);
    if (e != hipSuccess) return e;
    return launch_u_inner_bu(cm, beta, u, u_prev, state, V, D16, SD, N, S, n_u, n_iter2, mode, slab, u2_partials, grid_out, st);
}

}  // namespace dmf
