// u phase for many unknown cell types (9 <= n_u <= 26; the --ic sweep of ic.py:171 goes to 25) on the FP64
// matrix cores.  Same Gram form as dmf_kernels_rowpass_mfma.hip: per row i
//     c_i = alpha_unk (d_i * (v_i - Rt_i alpha_known))^T       (n_u values)
//     M_i = alpha_unk diag(d_i) alpha_unk^T                    (n_u (n_u + 1) / 2 values, up to 351)
// then n_iter2 row-local accelerated projected-gradient steps  u <- clip(ut + (c - M x) / l_w, 0, 1).
// With this many pair rows the M accumulators of a 64-sample wave no longer fit its registers, so the work is
// split the other way round: a workgroup takes a block of 16 rows, each wave OWNS up to three 16-row tiles
// of the [c ; M] output and walks over ALL 16-sample strips of the block (v, d and the E = V - Rt alpha_known
// product are re-derived per wave: a few MFMAs against 12 per strip for the owned tiles).  No cross-wave
// reduction, fixed summation order, results land in LDS:
//     M stage  per strip: E chain (NKC MFMAs), then per owned tile 4 MFMAs; A operands alpha_j alpha_l are formed
//              on the fly from an LDS copy of alpha (row stride S16 + 2)
//     B stage  lane = (row, j) in groups of GS = 16 or 32 lanes; each lane keeps row j of M_i in registers
//              and the current gradient point of its row is exchanged through LDS
// Layouts (tools/mfma_probe.hip): A[i][k]: lane (i = l & 15, k = l >> 4); B[k][j]: lane (k = l >> 4, j = l & 15);
// C register r of lane l = C[(l >> 4) + 4 r][l & 15]; "row-on-lane": lane = (row = l & 15, q = l >> 4),
// register r <-> sample s0 + 4 q + r.
#include <cstdlib>

#include "dmf_device.h"
#include "dmf_internal.h"

namespace dmf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kBigTilesPerWave = 3;
constexpr int kBigMaxNu = 26;  // 2 + 22 tiles = 8 waves x 3

struct BigLayout {  // dynamic LDS, in doubles
    int AS;      // alpha row stride (S rounded up to 16, + 2)
    int NPS;     // M row stride (NP rounded up to an odd number)
    int off_alpha, off_c, off_m, off_p, total;
};

__host__ __device__ inline BigLayout big_layout(int S, int n_c, int n_u, int n_iter2, int GS) {
    BigLayout L;
    const int K = n_c + n_u, NP = n_u * (n_u + 1) / 2;
    L.AS = (S + 15) / 16 * 16 + 2;
    L.NPS = NP | 1;
    L.off_alpha = (n_iter2 + 1) & ~1;
    L.off_c = L.off_alpha + K * L.AS;
    L.off_m = L.off_c + 16 * (GS + 1);      // c[row][j], row stride GS + 1
    L.off_p = L.off_m + 16 * L.NPS;         // M[row][pair]
    L.total = L.off_p + 16 * GS;            // gradient point x[row][j]
    return L;
}

template <int NKC, int GS>
__global__ __launch_bounds__(GS == 16 ? 256 : 512) void k_u_phase_big(
    const double* __restrict__ V, const double* __restrict__ D, const double* __restrict__ Rtp,
    const double* __restrict__ alpha, double* __restrict__ u, double* __restrict__ u_prev,
    const SolverState* __restrict__ state, int64_t N, int S, int n_c, int n_u, int n_iter2, int mode) {
    extern __shared__ double lds_dyn[];
    if (state->done) return;
    constexpr int NWV = GS == 16 ? 4 : 8;       // waves per workgroup
    constexpr int RPWV = 64 / GS;               // rows per wave in the B stage (NWV * RPWV = 16)
    constexpr int kBigPrefetch = GS == 16 ? 1 : 2;  // strips in flight; measured: occupancy beats a deeper prefetch
    const int K = n_c + n_u, NP = n_u * (n_u + 1) / 2;
    const int CT = (n_u + 15) / 16, MT = (NP + 15) / 16;  // 16-row tiles of c and of M
    const BigLayout L = big_layout(S, n_c, n_u, n_iter2, GS);
    double* __restrict__ beta_tab = lds_dyn;
    double* __restrict__ alds = lds_dyn + L.off_alpha;  // rows 0..n_c-1: -alpha_known, then alpha_unk; 0 past S
    double* __restrict__ cbuf = lds_dyn + L.off_c;
    double* __restrict__ mbuf = lds_dyn + L.off_m;
    double* __restrict__ pbuf = lds_dyn + L.off_p;
    const int AS = L.AS, NPS = L.NPS;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int m16 = lane & 15, q = lane >> 4;

    if (threadIdx.x == 0) {  // momentum coefficients of the n_iter2 inner steps (deconvolution.py:83-85)
        double a1 = state->a1, lw_prev = state->l_w_prev;
        const double lw = state->l_w;
        for (int t2 = 0; t2 < n_iter2; ++t2) {
            double beta;
            momentum_step(a1, lw_prev, lw, beta);
            beta_tab[t2] = beta;
            lw_prev = lw;
        }
    }
    for (int i = threadIdx.x; i < K * AS; i += NWV * 64) {
        const int r = i / AS, c = i - r * AS;
        double val = 0.0;
        if (c < S) val = r < n_c ? -alpha[(int64_t)r * S + c] : alpha[(int64_t)r * S + c];
        alds[i] = val;
    }
    const double inv_lw = 1.0 / state->l_w;  // x / l_w as x * (1 / l_w): <= 1 ulp from the division

    // ---- the tiles this wave owns: tile t < CT is rows 16 t.. of c, tile CT + t is pairs 16 t.. of M
    int n_own = 0;
    int row_a[kBigTilesPerWave], row_b[kBigTilesPerWave];  // alds rows whose product is this lane's A operand
    bool is_c[kBigTilesPerWave], live[kBigTilesPerWave];
    int out_base[kBigTilesPerWave];
#pragma unroll
    for (int x = 0; x < kBigTilesPerWave; ++x) {
        const int tile = x * NWV + wave;  // round robin: 12 tiles over 8 waves are 2,2,2,2,1,1,1,1, not 3,3,3,3,0,0,0,0
        is_c[x] = tile < CT;
        live[x] = false;
        row_a[x] = row_b[x] = 0;
        out_base[x] = 0;
        if (tile < CT + MT) {
            n_own = x + 1;
            if (is_c[x]) {
                const int j = tile * 16 + m16;
                live[x] = j < n_u;
                row_a[x] = n_c + (live[x] ? j : 0);
                out_base[x] = tile * 16;
            } else {
                const int p = (tile - CT) * 16 + m16;
                live[x] = p < NP;
                int l = 0;
                while ((l + 1) * (l + 2) / 2 <= p) ++l;
                row_a[x] = n_c + (live[x] ? p - l * (l + 1) / 2 : 0);
                row_b[x] = n_c + (live[x] ? l : 0);
                out_base[x] = (tile - CT) * 16;
            }
        }
    }
    __syncthreads();

    const int nstrips = (S + 15) / 16;
    const int64_t nblk = (N + 15) / 16;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t row0 = blk * 16;
        const int64_t rowc = row0 + m16 < N ? row0 + m16 : N - 1;  // clamped: rows past N are never stored
        double rtop[NKC > 0 ? NKC : 1];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) rtop[kc] = Rtp[rowc * (4 * NKC) + kc * 4 + q];
        v4d acc[kBigTilesPerWave];
#pragma unroll
        for (int x = 0; x < kBigTilesPerWave; ++x) acc[x] = v4d{0.0, 0.0, 0.0, 0.0};

        // ================= M stage: this wave's tiles over all the strips =================
        const double* __restrict__ vrow = V + rowc * S;
        const double* __restrict__ drow = D + rowc * S;
        // kBigPrefetch strips in flight per wave: one strip is only 12 MFMAs (~0.4 us) of work, far less than an
        // HBM round trip, and with up to 200 VGPRs only two waves share a SIMD
        v4d nv[kBigPrefetch], nd[kBigPrefetch];
        const bool vec = (S & 3) == 0;  // a lane's four samples are then contiguous, 32-byte aligned and in range
        auto load_strip = [&](int t, v4d& e, v4d& d) {
            if (vec) {
                int c = 16 * t + 4 * q;
                c = c < S ? c : S - 4;
                const v2d v01 = *reinterpret_cast<const v2d*>(vrow + c);
                const v2d v23 = *reinterpret_cast<const v2d*>(vrow + c + 2);
                const v2d d01 = *reinterpret_cast<const v2d*>(drow + c);
                const v2d d23 = *reinterpret_cast<const v2d*>(drow + c + 2);
                e = v4d{v01.x, v01.y, v23.x, v23.y};
                d = v4d{d01.x, d01.y, d23.x, d23.y};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int c = 16 * t + 4 * q + r;
                    c = c < S ? c : S - 1;  // clamped: columns past S meet zero alpha
                    e[r] = vrow[c];
                    d[r] = drow[c];
                }
            }
        };
        auto run_strip = [&](int t, v4d e, const v4d d) {
            // E^T = V^T - alpha_known^T Rt^T: A operand row m <-> sample 16 t + 4 (m & 3) + (m >> 2), k = q
            const int s_e = 16 * t + 4 * (m16 & 3) + (m16 >> 2);
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int kk = kc * 4 + q;
                const double a1 = kk < n_c ? alds[kk * AS + s_e] : 0.0;
                e = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, rtop[kc], e, 0, 0, 0);
            }
            const v4d w = d * e;
            const int col = 16 * t + 4 * q;  // k-step r <-> sample col + r
#pragma unroll
            for (int x = 0; x < kBigTilesPerWave; ++x) {
                if (x < n_own) {  // wave-uniform
                    const v2d a01 = *reinterpret_cast<const v2d*>(alds + row_a[x] * AS + col);
                    const v2d a23 = *reinterpret_cast<const v2d*>(alds + row_a[x] * AS + col + 2);
                    v4d a = {a01.x, a01.y, a23.x, a23.y};
                    if (!is_c[x]) {
                        const v2d b01 = *reinterpret_cast<const v2d*>(alds + row_b[x] * AS + col);
                        const v2d b23 = *reinterpret_cast<const v2d*>(alds + row_b[x] * AS + col + 2);
                        a = a * v4d{b01.x, b01.y, b23.x, b23.y};
                    }
                    if (!live[x]) a = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], is_c[x] ? w[r] : d[r], acc[x], 0, 0, 0);
                }
            }
        };
#pragma unroll
        for (int i = 0; i < kBigPrefetch; ++i)
            if (i < nstrips) load_strip(i, nv[i], nd[i]);
        for (int t0 = 0; t0 < nstrips; t0 += kBigPrefetch) {
#pragma unroll
            for (int i = 0; i < kBigPrefetch; ++i) {
                if (t0 + i < nstrips) {  // wave-uniform
                    const v4d e = nv[i], d = nd[i];
                    if (t0 + i + kBigPrefetch < nstrips) load_strip(t0 + i + kBigPrefetch, nv[i], nd[i]);
                    run_strip(t0 + i, e, d);
                }
            }
        }
        // C register r of lane l: output row (q + 4 r) of the tile, column m16 = block row
#pragma unroll
        for (int x = 0; x < kBigTilesPerWave; ++x) {
            if (x < n_own) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = out_base[x] + q + 4 * r;
                    if (is_c[x]) {
                        if (o < n_u) cbuf[m16 * (GS + 1) + o] = acc[x][r];
                    } else {
                        if (o < NP) mbuf[m16 * NPS + o] = acc[x][r];
                    }
                }
            }
        }
        __syncthreads();

        // ================= B stage: n_iter2 row-local steps, lane = (row, j) =================
        const int rloc = wave * RPWV + lane / GS, j = lane % GS;
        const bool jlive = j < n_u;
        const int jc = jlive ? j : 0;
        const int64_t grow = row0 + rloc;
        const bool store = jlive && grow < N;
        const int64_t gi = (grow < N ? grow : N - 1) * n_u + jc;
        double uv = u[gi], upv = u_prev[gi];
        double mrow[GS];  // -M_i[j][l] / l_w
#pragma unroll
        for (int l = 0; l < GS; ++l) {
            const int lc = l < n_u ? l : 0;
            const int p = lc <= jc ? tri(lc, jc) : tri(jc, lc);
            mrow[l] = l < n_u ? -inv_lw * mbuf[rloc * NPS + p] : 0.0;
        }
        const double cj = inv_lw * cbuf[rloc * (GS + 1) + jc];
        double* __restrict__ xrow = pbuf + rloc * GS;
        for (int t2 = 0; t2 < n_iter2; ++t2) {
            const double beta = beta_tab[t2];
            const double ut = fma(beta, uv - upv, uv);
            xrow[j] = mode == 1 ? uv : ut;  // deconvolution.py:163 (previous iterate) vs :88 (extrapolated point)
            upv = uv;
            // (the row's lanes sit in one wave: LDS executes a wave's accesses in order, no barrier needed)
            double g0 = ut + cj, g1 = 0.0, g2 = 0.0, g3 = 0.0;
#pragma unroll
            for (int l = 0; l < GS; l += 4) {
                const v2d x01 = *reinterpret_cast<const v2d*>(xrow + l);
                const v2d x23 = *reinterpret_cast<const v2d*>(xrow + l + 2);
                g0 = fma(mrow[l], x01.x, g0);
                g1 = fma(mrow[l + 1], x01.y, g1);
                g2 = fma(mrow[l + 2], x23.x, g2);
                g3 = fma(mrow[l + 3], x23.y, g3);
            }
            uv = fmin(fmax((g0 + g1) + (g2 + g3), 0.0), 1.0);
        }
        if (store) {
            u[gi] = uv;
            u_prev[gi] = upv;
        }
        __syncthreads();  // cbuf / mbuf / pbuf are rewritten by the next block
    }
}

static int big_group_size(int n_u) { return n_u <= 16 ? 16 : 32; }

bool u_phase_big_supported(int S, int n_c, int n_u, int n_iter2) {
    if (n_u < 9 || n_u > kBigMaxNu || n_c > 16 || S < 1) return false;
    const int GS = big_group_size(n_u);
    const int NP = n_u * (n_u + 1) / 2;
    if ((n_u + 15) / 16 + (NP + 15) / 16 > (GS == 16 ? 4 : 8) * kBigTilesPerWave) return false;
    return (size_t)big_layout(S, n_c, n_u, n_iter2, GS).total * sizeof(double) <= 160 * 1024;
}

template <int NKC, int GS>
static hipError_t launch_big_t(const double* V, const double* D, const double* Rtp, const double* alpha, double* u,
                               double* u_prev, const SolverState* state, int64_t N, int S, int n_c, int n_u,
                               int n_iter2, int mode, hipStream_t st) {
    const size_t lds = (size_t)big_layout(S, n_c, n_u, n_iter2, GS).total * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > 48 * 1024 && !raised[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)k_u_phase_big<NKC, GS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised[dev] = true;
    }
    const int64_t nblk = (N + 15) / 16;
    int per_cu = lds <= 78 * 1024 ? 2 : 1;
#ifdef DMF_EXPERIMENT  // (an experiment build only: DMF_EXPERIMENT=1 python -m demethify_amd._build)
    if (const char* v = getenv("DMF_UBIG_PER_CU")) per_cu = atoi(v) > 0 ? atoi(v) : per_cu;  // (experiments)
#endif
    const int64_t want = 256 * per_cu;
    const int64_t grid = nblk < want ? nblk : want;
    hipLaunchKernelGGL((k_u_phase_big<NKC, GS>), dim3((unsigned)grid), dim3(GS == 16 ? 256 : 512), lds, st, V, D, Rtp,
                       alpha, u, u_prev, state, N, S, n_c, n_u, n_iter2, mode);
    return hipGetLastError();
}

hipError_t launch_u_phase_big(const double* V, const double* D, const double* Rtp, const double* alpha, double* u,
                              double* u_prev, const SolverState* state, int64_t N, int S, int n_c, int n_u,
                              int n_iter2, int mode, hipStream_t st) {
    const int gs = big_group_size(n_u);
#define DMF_CASE(X)                                                                                              \
    case X:                                                                                                      \
        return gs == 16 ? launch_big_t<X, 16>(V, D, Rtp, alpha, u, u_prev, state, N, S, n_c, n_u, n_iter2, mode, st) \
                        : launch_big_t<X, 32>(V, D, Rtp, alpha, u, u_prev, state, N, S, n_c, n_u, n_iter2, mode, st);
    switch ((n_c + 3) / 4) {
        DMF_CASE(0) DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4)
        default: return hipErrorInvalidValue;
    }
#undef DMF_CASE
}

}  // namespace dmf
