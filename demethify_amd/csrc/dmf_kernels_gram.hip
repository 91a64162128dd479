// Shape-specialised per-sample weighted Gram accumulation for the alpha phase (the part of
// G_s = R^T diag(d_s) R and b_s = R^T (d_s * v_s) that involves the unknown profiles u).
//
// One pass over V and D per outer iteration.  Lane = sample column (coalesced 512 B per wave-row),
// the row's R_trunc (padded copy Rtp) / u values are wave-uniform (scalar loads feeding v_fma_f64 directly), every
// accumulator lives in registers for the whole row chunk:
//     cross[k][j] += d * Rt_ik * u_ij        (NCT x NU)
//     uu[j<=l]    += d * u_ij * u_il         (NU (NU+1) / 2)
//     bu[j]       += d * v * u_ij            (NU)
// Job order of the slab = the solver's job table (dmf_api.hip: l = n_c..K, k <= l).
#include "dmf_device.h"
#include "dmf_internal.h"

namespace dmf {

constexpr int kGramRedChunk = 16;

template <int NCT, int NU>
__global__ __launch_bounds__(256) void k_gram_u(const double* __restrict__ V, const double* __restrict__ D,
                                                const double* __restrict__ Rtp, const double* __restrict__ u,
                                                int64_t N, int S, int n_c, int64_t rows_per_chunk,
                                                double* __restrict__ slab, const int* __restrict__ done_flag) {
    constexpr int NP = NU * (NU + 1) / 2;
    constexpr int NACC = NCT * NU + NP + NU;
    __shared__ double red[3][kGramRedChunk][64];
    if (done_flag != nullptr && *done_flag) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 64 + lane;
    const bool active = s < S;
    const int sc = active ? s : S - 1;

    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = 0.0;

    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < N ? r0 + rows_per_chunk : N;

    // Rtp is R_trunc with rows padded to NCT doubles (zeros): unconditional, 32-byte aligned scalar
    // loads, and the padded accumulators simply stay zero.
    auto accum_row = [&](int64_t i, double d, double v) {
        const double* __restrict__ rt_row = Rtp + i * NCT;
        const double* __restrict__ u_row = u + i * NU;
        double uj[NU], t[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            uj[j] = u_row[j];
            t[j] = d * uj[j];
        }
#pragma unroll
        for (int k = 0; k < NCT; ++k) {
            const double rk = rt_row[k];
#pragma unroll
            for (int j = 0; j < NU; ++j) acc[k * NU + j] = fma(rk, t[j], acc[k * NU + j]);
        }
#pragma unroll
        for (int l = 0; l < NU; ++l)
#pragma unroll
            for (int j = 0; j <= l; ++j) acc[NCT * NU + tri(j, l)] = fma(t[j], uj[l], acc[NCT * NU + tri(j, l)]);
#pragma unroll
        for (int j = 0; j < NU; ++j) acc[NCT * NU + NP + j] = fma(t[j], v, acc[NCT * NU + NP + j]);
    };

    // kRowUnroll rows in flight per wave: their V / D loads are issued together before the FMA work of
    // the first one starts.  No predicates in the main loop (inactive lanes read a clamped column and
    // never store); the ragged end of the chunk goes through the one-row tail loop.
    constexpr int kRowUnroll = 4;
    int64_t ib = r0 + wave;
    for (; ib + 4 * (kRowUnroll - 1) < r1; ib += 4 * kRowUnroll) {
        double dd[kRowUnroll], vv[kRowUnroll];
#pragma unroll
        for (int x = 0; x < kRowUnroll; ++x) {
            dd[x] = D[(ib + 4 * x) * S + sc];
            vv[x] = V[(ib + 4 * x) * S + sc];
        }
#pragma unroll
        for (int x = 0; x < kRowUnroll; ++x) accum_row(ib + 4 * x, dd[x], vv[x]);
    }
    for (; ib < r1; ib += 4) accum_row(ib, D[ib * S + sc], V[ib * S + sc]);

    // cross-wave sum in fixed order, kGramRedChunk accumulators at a time, then scatter to the
    // slab in the solver's job order
    const int n_jobs = n_c * NU + NP + NU;
#pragma unroll
    for (int c0 = 0; c0 < NACC; c0 += kGramRedChunk) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int p = 0; p < kGramRedChunk; ++p)
                if (c0 + p < NACC) red[wave - 1][p][lane] = acc[c0 + p];
        }
        __syncthreads();
        if (wave == 0 && active) {
#pragma unroll
            for (int p = 0; p < kGramRedChunk; ++p) {
                const int a = c0 + p;
                if (a < NACC) {
                    // accumulator a -> (k, l) -> job index
                    int job = -1;
                    if (a < NCT * NU) {
                        const int k = a / NU, j = a % NU;
                        if (k < n_c) job = j * n_c + j * (j + 1) / 2 + k;
                    } else if (a < NCT * NU + NP) {
                        // packed tri(j, l) with j <= l: recover l then j
                        const int q = a - NCT * NU;
                        int l = 0;
                        while ((l + 1) * (l + 2) / 2 <= q) ++l;
                        const int j = q - l * (l + 1) / 2;
                        job = l * n_c + l * (l + 1) / 2 + n_c + j;
                    } else {
                        const int j = a - NCT * NU - NP;
                        job = NU * n_c + NP + j;
                    }
                    if (job >= 0) {
                        const double tot = ((acc[a] + red[0][p][lane]) + red[1][p][lane]) + red[2][p][lane];
                        slab[((int64_t)blockIdx.y * n_jobs + job) * S + s] = tot;
                    }
                }
            }
        }
    }
}

// geometry shared with the generic path's reducer (k_gram_reduce in dmf_kernels_stream.hip)
static void gram_u_geometry(int64_t N, int S, int* nsx, int* ny, int64_t* rows_per_chunk) {
    *nsx = (S + 63) / 64;
    int64_t want = 1024 / (*nsx);
    if (want < 1) want = 1;
    int64_t rpc = (N + want - 1) / want;
    if (rpc < 64) rpc = 64;
    *rows_per_chunk = rpc;
    *ny = (int)((N + rpc - 1) / rpc);
}

bool gram_u_supported(int n_c, int n_u) {
    if (n_u < 1 || n_u > 13 || n_c > 16) return false;  // (13: the register budget below with n_c = 0)
    const int nct = (n_c + 3) / 4 * 4;
    return nct * n_u + n_u * (n_u + 1) / 2 + n_u <= 112;
}

int64_t gram_u_slab_doubles(int64_t N, int S, int n_c, int n_u) {
    int nsx, ny;
    int64_t rpc;
    gram_u_geometry(N, S, &nsx, &ny, &rpc);
    return (int64_t)ny * (n_c * n_u + n_u * (n_u + 1) / 2 + n_u) * S;
}

template <int NCT, int NU>
static hipError_t launch_gram_u_t(const double* V, const double* D, const double* Rt, const double* u,
                                  int64_t N, int S, int n_c, double* slab, const int* done_flag,
                                  int* ny_out, hipStream_t st) {
    int nsx, ny;
    int64_t rpc;
    gram_u_geometry(N, S, &nsx, &ny, &rpc);
    *ny_out = ny;
    hipLaunchKernelGGL((k_gram_u<NCT, NU>), dim3(nsx, ny), dim3(256), 0, st, V, D, Rt, u, N, S, n_c, rpc, slab,
                       done_flag);
    return hipGetLastError();
}

template <int NCT>
static hipError_t launch_gram_u_nct(int n_u, const double* V, const double* D, const double* Rt,
                                    const double* u, int64_t N, int S, int n_c, double* slab,
                                    const int* done_flag, int* ny_out, hipStream_t st) {
    switch (n_u) {
#define DMF_CASE(NU_)                                                                       \
    case NU_:                                                                               \
        if constexpr (NCT * NU_ + NU_ * (NU_ + 1) / 2 + NU_ <= 112)                          \
            return launch_gram_u_t<NCT, NU_>(V, D, Rt, u, N, S, n_c, slab, done_flag, ny_out, st); \
        else                                                                                \
            return hipErrorInvalidValue;
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4) DMF_CASE(5) DMF_CASE(6) DMF_CASE(7) DMF_CASE(8)
        DMF_CASE(9) DMF_CASE(10) DMF_CASE(11) DMF_CASE(12) DMF_CASE(13)
#undef DMF_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gram_u(const double* V, const double* D, const double* Rt, const double* u, int64_t N,
                         int S, int n_c, int n_u, double* slab, const int* done_flag, int* ny_out,
                         hipStream_t st) {
    const int nct = (n_c + 3) / 4 * 4;
    switch (nct) {
        case 0: return launch_gram_u_nct<0>(n_u, V, D, Rt, u, N, S, n_c, slab, done_flag, ny_out, st);
        case 4: return launch_gram_u_nct<4>(n_u, V, D, Rt, u, N, S, n_c, slab, done_flag, ny_out, st);
        case 8: return launch_gram_u_nct<8>(n_u, V, D, Rt, u, N, S, n_c, slab, done_flag, ny_out, st);
        case 12: return launch_gram_u_nct<12>(n_u, V, D, Rt, u, N, S, n_c, slab, done_flag, ny_out, st);
        case 16: return launch_gram_u_nct<16>(n_u, V, D, Rt, u, N, S, n_c, slab, done_flag, ny_out, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dmf
