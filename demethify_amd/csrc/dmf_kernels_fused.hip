// Fused row pass: ONE read of V and D per outer iteration.
//
// A persistent workgroup walks 16-row blocks.  Per block:
//   load    coalesced global loads (16 B per lane, two rows per instruction) -> registers -> LDS tile
//   phase A read the tile in the row-on-lane layout and run the three FP64-MFMA contractions of
//           dmf_kernels_rowpass_mfma.hip (E = V - Rt a_known, c = a_unk (D*E)^T, M = P D^T)
//   phase B n_iter2 row-local accelerated projected-gradient steps (one wave, round robin); the new u
//           rows go to global memory and to LDS
//   phase C read the tile sample-on-lane and accumulate, in registers across all row blocks, the
//           u-dependent entries of the per-sample Gram matrices for the alpha phase
//           (cross[k][j] += d Rt_k u_j, uu[j<=l] += d u_j u_l, bu[j] += d v u_j), as dmf_kernels_gram.hip
// Phases A/B and phase C run on two wave teams of the same workgroup, one block apart, so that each SIMD
// always has an MFMA-bound wave and a VALU-bound wave resident (see the kernel's comment).
// At the end every workgroup stores its accumulators as one slab (job order of the solver's table) and
// its share of ||u||_F^2; k_gram_reduce sums the slabs in fixed order.
//
// Preconditions (checked by the launcher): S % 4 == 0, S <= 256, n_c <= 16, n_u <= 8; `Rtp` is the
// problem's zero-padded copy of R_trunc (row stride 4 NKC).
#include "dmf_device.h"
#include "dmf_internal.h"

namespace dmf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kTileRowDoubles = 66;  // 64 samples + 16 B pad: conflict-free b128 stores, b64 row reads
constexpr int kTileDoubles = 16 * kTileRowDoubles;

template <int CTRL>
__device__ __forceinline__ double f_dpp_quad(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int NU, int L>
__device__ __forceinline__ double f_group_bcast(double x, int lane0) {
    if constexpr (NU == 1) return x;
    else if constexpr (NU == 2) return f_dpp_quad<(L) | (L << 2) | ((2 + L) << 4) | ((2 + L) << 6)>(x);
    else if constexpr (NU == 4) return f_dpp_quad<L | (L << 2) | (L << 4) | (L << 6)>(x);
    else return __shfl(x, lane0 + L, 64);
}

template <int NU, int L = 0>
__device__ __forceinline__ double f_grad_row(double g, double base, const double (&Mrow)[NU], int lane0) {
    if constexpr (L < NU) {
        g = fma(-f_group_bcast<NU, L>(base, lane0), Mrow[L], g);
        return f_grad_row<NU, L + 1>(g, base, Mrow, lane0);
    } else {
        return g;
    }
}

// Team layout: a workgroup has 2 * NW waves (NW = ceil(S / 64)).  Waves [0, NW) form the A team,
// waves [NW, 2 NW) the C team; A wave w and C wave NW + w own sample columns [64 w, 64 w + 64).
// Step s of a workgroup overlaps, on every SIMD, the MFMA work of block s (A team) with the VALU
// work of block s - 1 (C team):
//   A team  write the prefetched V / D tile of block s to LDS buffer s & 1, issue the global loads of
//           block s + 1 into registers, phase A (MFMA) -> partial c / M in LDS
//           -- barrier X --   phase B (one A wave, round robin) -> u rows to global + LDS   -- barrier Y --
//   C team  phase C rows 0..7 of block s - 1 (buffer (s - 1) & 1)   -- X --   rows 8..15   -- Y --
template <int NKC, int NU>
__global__ __launch_bounds__(512) void k_rowpass_fused(
    const double* __restrict__ V, const double* __restrict__ D, const double* __restrict__ Rtp,
    const double* __restrict__ alpha, double* __restrict__ u, double* __restrict__ u_prev,
    const SolverState* __restrict__ state, int64_t N, int S, int n_c, int n_iter2, int mode,
    double* __restrict__ slab, double* __restrict__ u2_partials) {
    constexpr int NCT = 4 * NKC;
    constexpr int NCTL = NCT > 0 ? NCT : 1;
    constexpr int NP = NU * (NU + 1) / 2;
    constexpr int NMT = (NP + 15) / 16;
    constexpr int NV = NU + NP;
    constexpr int NACC = NCT * NU + NP + NU;
    extern __shared__ double lds_dyn[];
    if (state->done) return;

    const int NW = blockDim.x >> 7;  // waves per team
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool a_team = wave < NW;
    const int cw = a_team ? wave : wave - NW;  // column group of this wave
    const int wcol0 = cw * 64;

    // LDS carve-up: beta[n_iter2 (even)] | ubuf[2][16][NU] | rtbuf[2][16][NCT] | red[NW][NV][16] |
    //               tiles[2 buffers][NW column groups][V, D][16][kTileRowDoubles]
    double* __restrict__ beta_tab = lds_dyn;
    double* __restrict__ ubuf = beta_tab + ((n_iter2 + 1) & ~1);
    double* __restrict__ rtbuf = ubuf + 2 * 16 * NU;
    double* __restrict__ red = rtbuf + 2 * 16 * NCTL;
    double* __restrict__ tiles = red + NW * NV * 16;
    auto tile_of = [&](int buf) { return tiles + ((size_t)(buf * NW + cw) * 2) * kTileDoubles; };

    if (threadIdx.x == 0) {
        double a1 = state->a1, lw_prev = state->l_w_prev;
        const double lw = state->l_w;
        for (int t2 = 0; t2 < n_iter2; ++t2) {
            double beta;
            momentum_step(a1, lw_prev, lw, beta);
            beta_tab[t2] = beta;
            lw_prev = lw;
        }
    }
    __syncthreads();  // beta_tab visible

    const int64_t nblk = (N + 15) / 16;
    const int nk = (int)((nblk - blockIdx.x + gridDim.x - 1) / gridDim.x);  // blocks of this workgroup

    if (a_team) {
        // =========================== A team: phases A and B ===================================
        const int m16 = lane & 15, q = lane >> 4;
        const double* __restrict__ A2 = alpha + (int64_t)n_c * S;
        const double inv_lw = 1.0 / state->l_w;  // x / l_w as x * (1 / l_w): <= 1 ulp from the division
        double a1op[4][NKC > 0 ? NKC : 1];
        double a2op[4][4];
        double pop[4][NMT][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int s0 = wcol0 + t * 16;
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
                const int s = s0 + 4 * q + r;
                const int sc = s < S ? s : S - 1;
                const double keep2 = (m16 < NU && s < S) ? 1.0 : 0.0;
                a2op[t][r] = keep2 * A2[(int64_t)(m16 < NU ? m16 : 0) * S + sc];
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    const double keepp = (mt * 16 + m16 < NP && s < S) ? 1.0 : 0.0;
                    pop[t][mt][r] = keepp * (A2[(int64_t)jp[mt] * S + sc] * A2[(int64_t)lp[mt] * S + sc]);
                }
            }
        }
        // global -> register staging geometry: load i covers rows 2i, 2i+1; lane -> (row half, 2 samples)
        const int ld_row = lane >> 5;
        const int ld_col = (lane & 31) * 2;
        int ld_gcol = wcol0 + ld_col;
        if (ld_gcol > S - 2) ld_gcol = S - 2;  // ragged last column group: clamped samples are never consumed
        v2d pv[8], pd[8];
        auto prefetch = [&](int64_t blk) {
            const int64_t r0 = blk * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t gr = r0 + 2 * i + ld_row < N ? r0 + 2 * i + ld_row : N - 1;
                pv[i] = *reinterpret_cast<const v2d*>(V + gr * S + ld_gcol);
                pd[i] = *reinterpret_cast<const v2d*>(D + gr * S + ld_gcol);
            }
        };
        prefetch(blockIdx.x);
        double u2_acc = 0.0;

        for (int s = 0; s <= nk; ++s) {
            if (s < nk) {
                const int64_t blk = blockIdx.x + (int64_t)s * gridDim.x;
                const int64_t row0 = blk * 16;
                const int nvalid = N - row0 < 16 ? (int)(N - row0) : 16;
                double* __restrict__ tileV = tile_of(s & 1);
                double* __restrict__ tileD = tileV + kTileDoubles;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    *reinterpret_cast<v2d*>(tileV + (2 * i + ld_row) * kTileRowDoubles + ld_col) = pv[i];
                    *reinterpret_cast<v2d*>(tileD + (2 * i + ld_row) * kTileRowDoubles + ld_col) = pd[i];
                }
                if (s + 1 < nk) prefetch(blk + gridDim.x);
                // the wave that will run this block's inner iterations fetches its u / u_ now (first pass)
                constexpr int RPW = 64 / NU;
                const int rl = lane / NU, j = lane - rl * NU;
                const bool my_turn = wave == s % NW;
                double uu0 = 0.0, up0 = 0.0;
                if (my_turn) {
                    const bool ok0 = rl < RPW && rl < nvalid;
                    const int64_t gi0 = ok0 ? (row0 + rl) * NU + j : 0;
                    uu0 = u[gi0];
                    up0 = u_prev[gi0];
                }
                double rtop[NKC > 0 ? NKC : 1];
                {
                    const int64_t rowc = row0 + m16 < N ? row0 + m16 : N - 1;
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        rtop[kc] = Rtp[rowc * NCT + kc * 4 + q];
                        if (wave == 0) rtbuf[((s & 1) * 16 + m16) * NCTL + kc * 4 + q] = rtop[kc];
                    }
                }
                __builtin_amdgcn_wave_barrier();

                // ---- phase A: MFMA contractions on the tile, row-on-lane layout
                v4d cacc = {0.0, 0.0, 0.0, 0.0};
                v4d macc[NMT];
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) macc[mt] = cacc;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double* __restrict__ tv = tileV + m16 * kTileRowDoubles + t * 16 + 4 * q;
                    const double* __restrict__ td = tileD + m16 * kTileRowDoubles + t * 16 + 4 * q;
                    const v2d v01 = *reinterpret_cast<const v2d*>(tv), v23 = *reinterpret_cast<const v2d*>(tv + 2);
                    const v2d d01 = *reinterpret_cast<const v2d*>(td), d23 = *reinterpret_cast<const v2d*>(td + 2);
                    v4d e = {v01.x, v01.y, v23.x, v23.y};
                    const v4d d = {d01.x, d01.y, d23.x, d23.y};
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
                double* __restrict__ mine = red + (size_t)wave * NV * 16;
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
                __syncthreads();  // ---- barrier X

                // ---- phase B: row-local inner iterations, lane = (row, unknown j)
                if (my_turn) {
                    const int lane0 = lane - j;
                    double* __restrict__ ub = ubuf + (s & 1) * 16 * NU;
                    for (int pass0 = 0; pass0 < 16; pass0 += RPW) {
                        const int rloc = pass0 + rl;
                        const bool ok = rl < RPW && rloc < nvalid;
                        const int rlc = rloc < 16 ? rloc : 15;
                        double cj = 0.0, Mrow[NU];
#pragma unroll
                        for (int l = 0; l < NU; ++l) Mrow[l] = 0.0;
                        for (int w = 0; w < NW; ++w) {
                            const double* __restrict__ part = red + (size_t)w * NV * 16;
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
                            const double g = f_grad_row<NU>(cj, base, Mrow, lane0);
                            uu = fmin(fmax(fma(g, inv_lw, ut), 0.0), 1.0);
                        }
                        if (ok) {
                            u[gi] = uu;
                            u_prev[gi] = up;
                            ub[rloc * NU + j] = uu;
                            u2_acc = fma(uu, uu, u2_acc);
                        }
                    }
                }
                __syncthreads();  // ---- barrier Y
            } else {
                __syncthreads();  // X: the C team is finishing the last block
                __syncthreads();  // Y
            }
        }
        // share of ||u||^2 (red is free again: the last step's phase B is behind barrier Y)
        const double w2 = wave_sum(u2_acc);
        if (lane == 0) red[wave] = w2;
    } else {
        // =========================== C team: phase C ==========================================
        double acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = 0.0;
        const int sC = wcol0 + lane;  // lane = sample column

        auto accum_rows = [&](int buf, int r_begin, int nvalid) {
            const double* __restrict__ tileV = tile_of(buf);
            const double* __restrict__ tileD = tileV + kTileDoubles;
            const double* __restrict__ ub = ubuf + buf * 16 * NU;
            const double* __restrict__ rb = rtbuf + buf * 16 * NCTL;
#pragma unroll 2
            for (int r = r_begin; r < r_begin + 8; ++r) {
                if (r < nvalid) {
                    const double d = tileD[r * kTileRowDoubles + lane];
                    const double v = tileV[r * kTileRowDoubles + lane];
                    double uj[NU], t[NU];
#pragma unroll
                    for (int jj = 0; jj < NU; ++jj) {
                        uj[jj] = ub[r * NU + jj];
                        t[jj] = d * uj[jj];
                    }
#pragma unroll
                    for (int k = 0; k < NCT; ++k) {
                        const double rk = rb[r * NCTL + k];
#pragma unroll
                        for (int jj = 0; jj < NU; ++jj) acc[k * NU + jj] = fma(rk, t[jj], acc[k * NU + jj]);
                    }
#pragma unroll
                    for (int l = 0; l < NU; ++l)
#pragma unroll
                        for (int jj = 0; jj <= l; ++jj)
                            acc[NCT * NU + tri(jj, l)] = fma(t[jj], uj[l], acc[NCT * NU + tri(jj, l)]);
#pragma unroll
                    for (int jj = 0; jj < NU; ++jj)
                        acc[NCT * NU + NP + jj] = fma(t[jj], v, acc[NCT * NU + NP + jj]);
                }
            }
        };

        for (int s = 0; s <= nk; ++s) {
            int nvalid = 0, buf = 0;
            if (s >= 1) {
                const int64_t row0 = (blockIdx.x + (int64_t)(s - 1) * gridDim.x) * 16;
                nvalid = N - row0 < 16 ? (int)(N - row0) : 16;
                buf = (s - 1) & 1;
                accum_rows(buf, 0, nvalid);
            }
            __syncthreads();  // ---- barrier X
            if (s >= 1) accum_rows(buf, 8, nvalid);
            __syncthreads();  // ---- barrier Y
        }

        // ---- slab of this workgroup (job order of the solver's table)
        if (sC < S) {
            const int n_jobs = n_c * NU + NP + NU;
            double* __restrict__ out = slab + (int64_t)blockIdx.x * n_jobs * S + sC;
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                int job = -1;
                if (a < NCT * NU) {
                    const int k = a / NU, jj = a % NU;
                    if (k < n_c) job = jj * n_c + jj * (jj + 1) / 2 + k;
                } else if (a < NCT * NU + NP) {
                    const int qq = a - NCT * NU;
                    int l = 0;
                    while ((l + 1) * (l + 2) / 2 <= qq) ++l;
                    const int jj = qq - l * (l + 1) / 2;
                    job = l * n_c + l * (l + 1) / 2 + n_c + jj;
                } else {
                    job = NU * n_c + NP + (a - NCT * NU - NP);
                }
                if (job >= 0) out[(int64_t)job * S] = acc[a];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += red[w];
        u2_partials[blockIdx.x] = tot;
    }
}

// sum of the per-workgroup ||u||^2 shares -> state->u_norm2, then l_h (deconvolution.py:212)
__global__ __launch_bounds__(256) void k_finish_u_norm(const double* __restrict__ partials, int n,
                                                       SolverState* __restrict__ state) {
    __shared__ double red[4];
    if (state->done) return;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
    const double tot = block_sum<256>(acc, red);
    if (threadIdx.x == 0) {
        state->u_norm2 = tot;
        state->l_h = (state->rt_norm2 + tot) * state->dsq;
    }
}

static size_t fused_lds_bytes(int S, int nct, int n_u, int n_iter2) {
    const int NW = (S + 63) / 64;
    const int nv = n_u + n_u * (n_u + 1) / 2;
    return ((size_t)((n_iter2 + 1) & ~1) + 2 * 16 * n_u + 2 * 16 * (nct > 0 ? nct : 1) + (size_t)NW * nv * 16 +
            (size_t)2 * NW * 2 * kTileDoubles) * sizeof(double);
}

bool rowpass_fused_supported(int S, int n_c, int n_u) {
    if ((S & 3) != 0 || S > 256 || n_c > 16 || n_u < 1 || n_u > 8) return false;
    const int nct = (n_c + 3) / 4 * 4;
    return nct * n_u + n_u * (n_u + 1) / 2 + n_u <= 80;
}

int rowpass_fused_grid(int64_t N, int S) {
    const int NW = (S + 63) / 64;
    const int per_cu = NW <= 2 ? 2 : 1;  // LDS: two tile buffers per column group
    const int64_t nblk = (N + 15) / 16;
    const int64_t g = 256 * per_cu;
    return (int)(nblk < g ? nblk : g);
}

int64_t rowpass_fused_slab_doubles(int64_t N, int S, int n_c, int n_u) {
    return (int64_t)rowpass_fused_grid(N, S) * (n_c * n_u + n_u * (n_u + 1) / 2 + n_u) * S;
}

template <int NKC, int NU>
static hipError_t launch_fused_t(const double* V, const double* D, const double* Rtp, const double* alpha,
                                 double* u, double* u_prev, SolverState* state, int64_t N, int S, int n_c,
                                 int n_iter2, int mode, double* slab, double* u2_partials, int* grid_out,
                                 hipStream_t st) {
    if constexpr (4 * NKC * NU + NU * (NU + 1) / 2 + NU > 80) {
        return hipErrorInvalidValue;
    } else {
        const int NW = (S + 63) / 64;
        const size_t lds = fused_lds_bytes(S, 4 * NKC, NU, n_iter2);
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)k_rowpass_fused<NKC, NU>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const int grid = rowpass_fused_grid(N, S);
        *grid_out = grid;
        hipLaunchKernelGGL((k_rowpass_fused<NKC, NU>), dim3(grid), dim3(2 * NW * 64), lds, st, V, D, Rtp, alpha, u,
                           u_prev, state, N, S, n_c, n_iter2, mode, slab, u2_partials);
        hipLaunchKernelGGL(k_finish_u_norm, dim3(1), dim3(256), 0, st, u2_partials, grid, state);
        return hipGetLastError();
    }
}

template <int NKC>
static hipError_t launch_fused_nkc(int n_u, const double* V, const double* D, const double* Rtp,
                                   const double* alpha, double* u, double* u_prev, SolverState* state,
                                   int64_t N, int S, int n_c, int n_iter2, int mode, double* slab,
                                   double* u2_partials, int* grid_out, hipStream_t st) {
    switch (n_u) {
#define DMF_CASE(NU_)                                                                                      \
    case NU_:                                                                                              \
        return launch_fused_t<NKC, NU_>(V, D, Rtp, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, slab, \
                                        u2_partials, grid_out, st);
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4) DMF_CASE(5) DMF_CASE(6) DMF_CASE(7) DMF_CASE(8)
#undef DMF_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_rowpass_fused(const double* V, const double* D, const double* Rtp, const double* alpha,
                                double* u, double* u_prev, SolverState* state, int64_t N, int S, int n_c,
                                int n_u, int n_iter2, int mode, double* slab, double* u2_partials,
                                int* grid_out, hipStream_t st) {
    switch ((n_c + 3) / 4) {
#define DMF_NKC(X)                                                                                          \
    case X:                                                                                                 \
        return launch_fused_nkc<X>(n_u, V, D, Rtp, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, slab, \
                                   u2_partials, grid_out, st);
        DMF_NKC(0) DMF_NKC(1) DMF_NKC(2) DMF_NKC(3) DMF_NKC(4)
#undef DMF_NKC
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dmf
