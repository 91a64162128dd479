// Per-row c_i / M_i of the split u phase for wide row groups (5 <= n_u <= 16), gfx950:
//     c_i = alpha_unk (d_i * (v_i - Rt_i alpha_known))^T       FP64 matrix cores (as dmf_kernels_rowpass_mfma.hip)
//     M_i[pair (j,l)] = sum_s (alpha_js alpha_ls) d_is          INTEGER matrix cores, exactly (as dmf_kernels_rowpass2.hip):
// the counts are one or two balanced 8-bit digits, P = alpha_j alpha_l in [0, 1] seven digits of rint(P 2^52)
// (dmf_fixedpoint.h); one v_mfma_i32_16x16x64_i8 per digit covers 16 rows x 64 samples x 16 pairs.  With n_u unknowns
// there are n_u (n_u + 1) / 2 pairs (78 at 12): on the FP64 cores that product is 64 cycles per 16 x 4 x 16 piece and
// was nine tenths of k_u_phase_mfma / k_u_phase_big; here it is 16 cycles per 16 x 64 x 16 piece and digit.
//
// Work split: a WAVE owns 32 CpG rows (two 16-row halves) and walks over ALL samples; nothing is exchanged between
// waves after set-up (no barrier in the row loop, fixed summation order).  Per 16-row x 64-sample unit the wave
//   * takes V (32 B per lane and 16-sample strip) and the u16 counts (8 B) in the "row-on-lane" layout
//     lane = (row = l & 15, q = l >> 4), register r <-> sample s0 + 4 q + r, prefetched one unit ahead;
//   * E^T = V^T - alpha_known^T Rt^T, c^T += alpha_unk (D*E)^T on v_mfma_f64_16x16x4 (A operands read from an LDS
//     copy of alpha: a wave now meets every column group, so they no longer fit its registers);
//   * packs the strip's four counts into one dword per digit plane -- the lane's 16 samples of the unit are exactly
//     the 16 bytes of an i8 MFMA A operand (k <-> (q, strip, r)), no LDS round trip.
// Then, per 16-pair tile: 7 (or 14) integer MFMAs per column group and row half against the P digits (B operands,
// built once per workgroup in LDS in the same k order), the exact combine of dmf_kernels_rowpass2.hip, and the
// row's M values go to HBM.  k_u_inner_rows16 runs the inner iterations from there.
//
// Layout facts used (tools/mfma_probe.hip): FP64 A[i][k]: lane (i = l & 15, k = l >> 4); B[k][j]: lane (k = l >> 4,
// j = l & 15); C register r of lane l = C[(l >> 4) + 4 r][l & 15].  i8 16x16x64: A lane (row l & 15), B lane (column
// l & 15), byte b of register g of lane-group l >> 4 is the same k on both sides; C register r = row 4 (l >> 4) + r.
#include <cstdlib>

#include "dmf_device.h"
#include "dmf_fixedpoint.h"
#include "dmf_internal.h"

namespace dmf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kCmWaves = 8;  // waves per workgroup (they only share the LDS tables)
constexpr int kCmNkcWide = 12;  // chain links of the instantiation for more than 16 known types (n_c <= 48)

struct CmLayout {
    int AS;          // row stride of the alpha copy in doubles (64 ncg + 4)
    int n_rows;      // 4 ceil(n_c / 4) (negated known rows, zero padded) + n_u
    int pd_dwords4;  // P digit table: [ncg][nmt][7][64] x 16 bytes
    size_t bytes;
};

// (n_tiles: 16-pair tiles of M_i held by one launch; <= 0: all of them)
__host__ __device__ inline CmLayout cm_layout(int S, int n_c, int n_u, int n_tiles = 0) {
    CmLayout L;
    const int ncg = (S + 63) / 64, nmt = n_tiles > 0 ? n_tiles : (n_u * (n_u + 1) / 2 + 15) / 16;
    L.AS = 64 * ncg + 4;
    L.n_rows = (n_c + 3) / 4 * 4 + n_u;
    L.pd_dwords4 = ncg * nmt * 7 * 64;
    L.bytes = (size_t)L.pd_dwords4 * 16 + (size_t)L.n_rows * L.AS * sizeof(double);
    return L;
}

// NKC = ceil(n_c / 4); ND = count digit planes (1: every count <= 127, 2: <= 32639); NCGX >= number of 64-sample
// column groups (the per-group loops are unrolled NCGX times behind wave-uniform guards); S4: S % 4 == 0 (the plain
// strip loads; any other S >= 2 takes the form with per-pair offsets).
template <int NKC, int ND, int NCGX, bool S4>
__global__ __launch_bounds__(512) void k_cm_i8(const double* __restrict__ V, const unsigned short* __restrict__ D16, int SD,
                                               const double* __restrict__ Rtp, const double* __restrict__ alpha,
                                               const SolverState* __restrict__ state, int64_t N, int S, int n_c, int n_u,
                                               double* __restrict__ cm_out, int col0, int Sp, int accumulate, int c_tile,
                                               int c_store, int mt0, int mt1) {
    // More than 16 unknowns: a launch computes ONE 16-row tile of c (unknowns 16 c_tile ..; stored if c_store) and the
    // pair tiles mt0 .. mt1 - 1 of M_i -- the digit table of all of them (21 tiles at 25 unknowns) does not fit the LDS.
    // A launch covers the PANEL of samples col0 .. col0 + Sp - 1 (Sp <= 256) of rows that are S samples long; beyond 256
    // samples the launcher walks the panels and every launch but the first adds to what is in cm_out (a wave owns its
    // rows, the panels follow each other on the stream: fixed summation order).
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (state->done) return;
    constexpr int NWT = 7 + ND - 1;  // digit weights 256^0 .. 256^(NWT-1)
    const int NP = n_u * (n_u + 1) / 2, NV = n_u + NP;
    const int ncg = (Sp + 63) / 64;
    // NKC <= 4: exactly ceil(n_c / 4) chain links.  NKC = kCmNkcWide: up to that many (more than 16 known types), the
    // links behind wave-uniform guards, and the block's R_trunc rows fetched at the block (not a block ahead: registers).
    const int nkc = NKC <= 4 ? NKC : (n_c + 3) / 4;
    const int nct = 4 * nkc;
    constexpr bool kRtAhead = NKC <= 4;
    const int ntl = mt1 - mt0;  // pair tiles of this launch
    const CmLayout L = cm_layout(Sp, n_c, n_u, ntl);
    v4i* __restrict__ pd = reinterpret_cast<v4i*>(lds_raw);
    double* __restrict__ alds = reinterpret_cast<double*>(lds_raw + (size_t)L.pd_dwords4 * 16);
    const int AS = L.AS;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int m16 = lane & 15, q = lane >> 4;

    // ---- workgroup tables -------------------------------------------------------------------------------------
    // alpha copy: rows 0 .. 4 NKC - 1 = -alpha_known (zero rows past n_c), then alpha_unk; zero past the panel
    for (int i = threadIdx.x; i < L.n_rows * AS; i += kCmWaves * 64) {
        const int r = i / AS, c = i - r * AS;
        double val = 0.0;
        if (c < Sp) {
            if (r < nct) val = r < n_c ? -alpha[(int64_t)r * S + col0 + c] : 0.0;
            else val = alpha[(int64_t)(n_c + r - nct) * S + col0 + c];
        }
        alds[i] = val;
    }
    // P digits: item (column group, pair tile) per wave; lane (pair m16, q), register g = strip, byte i:
    // sample 64 cg + 16 g + 4 q + i  (the k order in which the row loop packs the counts)
    for (int item = wave; item < ncg * ntl; item += kCmWaves) {
        const int cg = item / ntl, mt = mt0 + (item - cg * ntl);
        const int p = mt * 16 + m16;
        int pl = 0;
        while ((pl + 1) * (pl + 2) / 2 <= p) ++pl;
        const int pj = p - pl * (pl + 1) / 2;
        const bool pair_ok = p < NP;
        v4i dg[7];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned int lo[4], hi[4], tl[4], th[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = 64 * cg + 16 * g + 4 * q + i;
                const bool in = pair_ok && col < Sp;
                const double aj = in ? alpha[(int64_t)(n_c + pj) * S + col0 + col] : 0.0;
                const double al = in ? alpha[(int64_t)(n_c + pl) * S + col0 + col] : 0.0;
                z_to_biased(aj, al, lo[i], hi[i]);
            }
            transpose4(lo, tl);
            transpose4(hi, th);
            dg[0][g] = (int)(tl[0] ^ 0x80808080u);
            dg[1][g] = (int)(tl[1] ^ 0x80808080u);
            dg[2][g] = (int)(tl[2] ^ 0x80808080u);
            dg[3][g] = (int)(tl[3] ^ 0x80808080u);
            dg[4][g] = (int)(th[0] ^ 0x80808080u);
            dg[5][g] = (int)(th[1] ^ 0x80808080u);
            dg[6][g] = (int)th[2];
        }
#pragma unroll
        for (int t = 0; t < 7; ++t) pd[(item * 7 + t) * 64 + lane] = dg[t];
    }
    __syncthreads();

    const int64_t nblk = (N + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * kCmWaves;
    int64_t blk = (int64_t)blockIdx.x * kCmWaves + wave;
    if (blk >= nblk) return;  // (no barrier below)

    // row-on-lane addresses of a 16-row half: rows past N are clamped (their outputs are never stored)
    auto row_of = [&](int64_t b, int h) {
        const int64_t row = b * 32 + 16 * h + m16;
        return row < N ? row : N - 1;
    };
    // strip t of column group cg: a lane's four samples start at c = 64 cg + 16 t + 4 q.  Samples at or beyond S meet
    // zero-padded counts, so their V only has to be finite: a group out of range reads the row's first group, and a pair
    // with one sample in range is fetched one element lower (never past the end of the row: the last row of V ends
    // the array).  With odd S rows start 8 bytes off a 16-byte boundary every other time; 16-byte global loads take that
    // (tools/align_probe.hip), the type says so.
    typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));
    auto load_strip = [&](int64_t rowc, int cg, int t, v4d& e, unsigned long long& d) {
        int c = 64 * cg + 16 * t + 4 * q;
        if constexpr (S4) {  // panel width % 4 == 0 (and S even): a lane's four samples are all in range or all out
            if constexpr (NCGX > 2) asm volatile("" : "+v"(c));  // (sixteen hoisted strip addresses spill; see below)
            const double* __restrict__ vp = V + rowc * S + col0 + (c < Sp ? c : 0);
            const v2d v01 = *reinterpret_cast<const v2d*>(vp);
            const v2d v23 = *reinterpret_cast<const v2d*>(vp + 2);
            e = v4d{v01.x, v01.y, v23.x, v23.y};
        } else {
            // (recomputed at every use: hoisted out of the row loop, the per-strip lane constants below cost the kernel 30
            // to 60 registers and, with four column groups, spills)
            asm volatile("" : "+v"(c));
            const int left = Sp - c;
            const int cb = left > 0 ? c : 0;
            const int nvb = left > 0 ? left : Sp;  // samples of the group read that exist (>= 1)
            const int o01 = nvb >= 2 ? 0 : -1;     // one sample: the pair (cb - 1, cb) -- inside the row: S >= 2, and a
                                                   // one-sample panel is never the row's first
            const int o23 = nvb >= 4 ? 2 : (nvb == 3 ? 1 : o01);  // three: (cb + 1, cb + 2); fewer: the first pair again
            const double* __restrict__ vp = V + rowc * S + col0 + cb;
            const v2d_u v01 = *reinterpret_cast<const v2d_u*>(vp + o01);
            const v2d_u v23 = *reinterpret_cast<const v2d_u*>(vp + o23);
            e = v4d{nvb >= 2 ? v01.x : v01.y, v01.y, nvb == 3 ? v23.y : v23.x, v23.y};
        }
        d = *reinterpret_cast<const unsigned long long*>(D16 + rowc * SD + col0 + (64 * cg + 16 * t + 4 * q));
    };

    v4d nv[4];
    unsigned long long nd[4];
    double nrt[2][kRtAhead && NKC > 0 ? NKC : 1];
    {
        const int64_t r0 = row_of(blk, 0), r1 = row_of(blk, 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) load_strip(r0, 0, t, nv[t], nd[t]);
        if constexpr (kRtAhead) {
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                nrt[0][kc] = Rtp[r0 * nct + kc * 4 + q];
                nrt[1][kc] = Rtp[r1 * nct + kc * 4 + q];
            }
        }
    }
    const int m16c = 16 * c_tile + m16 < n_u ? 16 * c_tile + m16 : 0;  // (rows >= n_u of the c tile are never stored: any finite operand will do)
    const double* __restrict__ a2row = alds + (nct + m16c) * AS + 4 * q;
    const double* __restrict__ a1row = alds + q * AS + 4 * (m16 & 3) + (m16 >> 2);  // + 4 kc AS + 64 cg + 16 t

    for (; blk < nblk; blk += stride) {
        const int64_t nxt = blk + stride < nblk ? blk + stride : blk;
        const int64_t rowh[2] = {row_of(blk, 0), row_of(blk, 1)};
        const int64_t rown0 = row_of(nxt, 0);
        double rtop[2][NKC > 0 ? NKC : 1];
        if constexpr (kRtAhead) {
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                rtop[0][kc] = nrt[0][kc];
                rtop[1][kc] = nrt[1][kc];
            }
            if (NKC > 0) {
                const int64_t rown1 = row_of(nxt, 1);
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) {
                    nrt[0][kc] = Rtp[rown0 * nct + kc * 4 + q];
                    nrt[1][kc] = Rtp[rown1 * nct + kc * 4 + q];
                }
            }
        } else {
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int kk = kc < nkc ? kc : 0;  // (links past ceil(n_c / 4) are skipped below)
                rtop[0][kc] = Rtp[rowh[0] * nct + kk * 4 + q];
                rtop[1][kc] = Rtp[rowh[1] * nct + kk * 4 + q];
            }
        }
        v4d cacc[2] = {v4d{0.0, 0.0, 0.0, 0.0}, v4d{0.0, 0.0, 0.0, 0.0}};
        v4i cdig[2][NCGX][ND];  // count digit planes of this block: [row half][column group][plane], register = strip

#pragma unroll
        for (int cg = 0; cg < NCGX; ++cg) {
            if (cg < ncg) {  // wave-uniform
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // the unit after this one (same block: other half / next column group; else the next block's first)
                    int64_t nrow;
                    int ncgi;
                    if (h == 0) {
                        nrow = rowh[1];
                        ncgi = cg;
                    } else if (cg + 1 < ncg) {
                        nrow = rowh[0];
                        ncgi = cg + 1;
                    } else {
                        nrow = rown0;
                        ncgi = 0;
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        v4d e = nv[t];
                        const unsigned long long dd = nd[t];
                        load_strip(nrow, ncgi, t, nv[t], nd[t]);  // prefetch ...
                        __builtin_amdgcn_sched_barrier(0);        // ... in front of this strip's MFMAs
#pragma unroll
                        for (int kc = 0; kc < NKC; ++kc) {
                            if (NKC <= 4 || kc < nkc) {  // (wave-uniform)
                                const double a1 = a1row[4 * kc * AS + 64 * cg + 16 * t];
                                e = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, rtop[h][kc], e, 0, 0, 0);
                            }
                        }
                        const unsigned int dl = (unsigned int)dd, dh = (unsigned int)(dd >> 32);
                        const v4d d = {(double)(dl & 0xFFFFu), (double)(dl >> 16), (double)(dh & 0xFFFFu), (double)(dh >> 16)};
                        const v4d w = d * e;
                        const v2d a01 = *reinterpret_cast<const v2d*>(a2row + 64 * cg + 16 * t);
                        const v2d a23 = *reinterpret_cast<const v2d*>(a2row + 64 * cg + 16 * t + 2);
                        cacc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, w[0], cacc[h], 0, 0, 0);
                        cacc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, w[1], cacc[h], 0, 0, 0);
                        cacc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, w[2], cacc[h], 0, 0, 0);
                        cacc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, w[3], cacc[h], 0, 0, 0);
                        // the strip's four counts as bytes r = 0..3 of one dword per digit plane
                        if constexpr (ND == 1) {
                            cdig[h][cg][0][t] = (int)__builtin_amdgcn_perm(dh, dl, 0x06040200u);
                        } else {
                            // d = lo + 256 hi with lo in [-128, 127]: d + 128 = 256 hi + (lo + 128)
                            const unsigned int xl = dl + 0x00800080u, xh = dh + 0x00800080u;
                            cdig[h][cg][0][t] = (int)(__builtin_amdgcn_perm(xh, xl, 0x06040200u) ^ 0x80808080u);
                            cdig[h][cg][1][t] = (int)__builtin_amdgcn_perm(xh, xl, 0x07050301u);
                        }
                    }
                }
            }
        }

        // ---- c: C register r of lane l = c^T[unknown q + 4 r][row m16]
        const int64_t row00 = blk * 32;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t row = row00 + 16 * h + m16;
            if (row < N) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ju = 16 * c_tile + q + 4 * r;
                    if (c_store && ju < n_u) {
                        double* __restrict__ dst = cm_out + row * NV + ju;
                        *dst = accumulate ? *dst + cacc[h][r] : cacc[h][r];
                    }
                }
            }
        }

        // ---- M: per 16-pair tile, digit weight w = P digit + count plane; lane (pair m16, q) gets rows 4 q + r
        // (HJ row halves share one read of the B operand; with four column groups the count planes leave registers
        // for one half's accumulators only)
        constexpr int HJ = NCGX <= 2 ? 2 : 1;
        for (int mt = mt0; mt < mt1; ++mt) {
            const int p = mt * 16 + m16;
#pragma unroll
            for (int h0 = 0; h0 < 2; h0 += HJ) {
                v4i acc[HJ][NWT];
#pragma unroll
                for (int h = 0; h < HJ; ++h)
#pragma unroll
                    for (int w = 0; w < NWT; ++w) acc[h][w] = v4i{0, 0, 0, 0};
#pragma unroll
                for (int cg = 0; cg < NCGX; ++cg) {
                    if (cg < ncg) {
                        const v4i* __restrict__ bp = pd + ((cg * ntl + (mt - mt0)) * 7) * 64 + lane;
#pragma unroll
                        for (int t = 0; t < 7; ++t) {
                            const v4i b = bp[t * 64];
#pragma unroll
                            for (int h = 0; h < HJ; ++h) {
                                acc[h][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cdig[h0 + h][cg][0], b, acc[h][t], 0, 0, 0);
                                if constexpr (ND == 2)
                                    acc[h][t + 1] =
                                        __builtin_amdgcn_mfma_i32_16x16x64_i8(cdig[h0 + h][cg][1], b, acc[h][t + 1], 0, 0, 0);
                            }
                        }
                    }
                }
                // exact integer sum_w 256^w acc[w] in two halves that fit a double without rounding (|acc| < 2^24 per
                // digit product sum at S <= 256 with two count planes), one rounding when they are joined
#pragma unroll
                for (int h = 0; h < HJ; ++h) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const double lo = fma(fma(fma((double)acc[h][3][rr], 256.0, (double)acc[h][2][rr]), 256.0, (double)acc[h][1][rr]),
                                              256.0, (double)acc[h][0][rr]);
                        double hi = fma(fma((double)acc[h][6][rr], 256.0, (double)acc[h][5][rr]), 256.0, (double)acc[h][4][rr]);
                        if constexpr (ND == 2) hi = fma((double)acc[h][7][rr], 16777216.0, hi);
                        const double m = fma(hi, 0x1p32, lo) * 0x1p-52;
                        const int64_t row = row00 + 16 * (h0 + h) + 4 * q + rr;
                        if (p < NP && row < N) {
                            double* __restrict__ dst = cm_out + row * NV + n_u + p;
                            *dst = accumulate ? *dst + m : m;
                        }
                    }
                }
            }
        }
    }
}

constexpr int kCmPanel = 256;  // samples per launch

bool cm_i8_supported(const double* V, int S, int n_c, int n_u, int ND, int SD) {
    if (n_u < 1 || n_u > 32 || n_c < 0 || n_c > 4 * kCmNkcWide || n_c + n_u > 64) return false;
    if (S < 2 || S > 2048 || (SD & 3) != 0 || SD < (S + 63) / 64 * 64) return false;
    if (ND != 1 && ND != 2) return false;
    if ((reinterpret_cast<uintptr_t>(V) & 7) != 0) return false;
    return cm_layout(S < kCmPanel ? S : kCmPanel, n_c, n_u, 1).bytes <= 160 * 1024;  // (at least one pair tile per launch)
}

template <int NKC, int ND, int NCGX, bool S4>
static hipError_t launch_cm_t(const double* V, const unsigned short* D16, int SD, const double* Rtp, const double* alpha,
                              const SolverState* state, int64_t N, int S, int n_c, int n_u, double* cm, int col0, int Sp,
                              int c_tile, int c_store, int mt0, int mt1, hipStream_t st) {
    const size_t lds = cm_layout(Sp, n_c, n_u, mt1 - mt0).bytes;
    static bool raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!raised[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)k_cm_i8<NKC, ND, NCGX, S4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        raised[dev] = true;
    }
    const int64_t nblk = (N + 31) / 32;
    const int64_t want = (nblk + kCmWaves - 1) / kCmWaves;
    // one workgroup (two waves per SIMD at ~200 registers) per CU, two where the tables leave room
    int per_cu = lds <= 76 * 1024 ? 2 : 1;
#ifdef DMF_EXPERIMENT  // (an experiment build only: DMF_EXPERIMENT=1 python -m demethify_amd._build)
    if (const char* v = getenv("DMF_CM_PER_CU")) per_cu = atoi(v) > 0 ? atoi(v) : per_cu;  // (experiments)
#endif
    const int64_t cap = (int64_t)256 * per_cu;
    const int64_t grid = want < cap ? want : cap;
    hipLaunchKernelGGL((k_cm_i8<NKC, ND, NCGX, S4>), dim3((unsigned)grid), dim3(kCmWaves * 64), lds, st, V, D16, SD, Rtp, alpha,
                       state, N, S, n_c, n_u, cm, col0, Sp, col0 > 0 ? 1 : 0, c_tile, c_store, mt0, mt1);
    return hipGetLastError();
}

template <int NKC>
static hipError_t launch_cm_nkc(const double* V, const unsigned short* D16, int SD, int ND, const double* Rtp,
                                const double* alpha, const SolverState* state, int64_t N, int S, int n_c, int n_u,
                                double* cm, int col0, int Sp, int c_tile, int c_store, int mt0, int mt1, hipStream_t st) {
    const bool wide = Sp > 128;
    const bool s4 = (Sp & 3) == 0 && (S & 1) == 0;  // (odd S: rows of V are 8-byte aligned only)
#define DMF_CM(ND_, NCGX_)                                                                                                  \
    return s4 ? launch_cm_t<NKC, ND_, NCGX_, true>(V, D16, SD, Rtp, alpha, state, N, S, n_c, n_u, cm, col0, Sp, c_tile,      \
                                                   c_store, mt0, mt1, st)                                                   \
              : launch_cm_t<NKC, ND_, NCGX_, false>(V, D16, SD, Rtp, alpha, state, N, S, n_c, n_u, cm, col0, Sp, c_tile,     \
                                                    c_store, mt0, mt1, st)
    if (ND == 1) {
        if (wide) DMF_CM(1, 4);
        DMF_CM(1, 2);
    }
    if (wide) DMF_CM(2, 4);
    DMF_CM(2, 2);
#undef DMF_CM
}

// cm: N x (n_u + n_u (n_u + 1) / 2) doubles.  Preconditions (cm_i8_supported + the caller): counts integral in
// [0, 127] (ND = 1) or [0, 32639] (ND = 2) in D16 (row stride SD, zero padded to a multiple of 64 samples, rows to a
// multiple of 16), alpha in [0, 1], Rtp = padded R_trunc (row stride 4 ceil(n_c / 4)).
hipError_t launch_cm_i8(const double* V, const unsigned short* D16, int SD, int ND, const double* Rtp, const double* alpha,
                        const SolverState* state, int64_t N, int S, int n_c, int n_u, double* cm, hipStream_t st) {
    if (!cm_i8_supported(V, S, n_c, n_u, ND, SD) || cm == nullptr || D16 == nullptr) return hipErrorInvalidValue;
    const int nmt = (n_u * (n_u + 1) / 2 + 15) / 16, n_ct = (n_u + 15) / 16;
    for (int col0 = 0; col0 < S; col0 += kCmPanel) {  // panels of 256 samples; the second and later ones add to cm
        const int Sp = S - col0 < kCmPanel ? S - col0 : kCmPanel;
        // pair tiles per launch: what the LDS holds beside the alpha copy; at least one launch per c tile
        int tpl = nmt;
        while (tpl > 1 && cm_layout(Sp, n_c, n_u, tpl).bytes > (size_t)160 * 1024) --tpl;
        int n_launch = (nmt + tpl - 1) / tpl;
        if (n_launch < n_ct) n_launch = n_ct;
        tpl = (nmt + n_launch - 1) / n_launch;
        for (int li = 0; li < n_launch; ++li) {
            const int mt0 = li * tpl < nmt ? li * tpl : nmt, mt1 = mt0 + tpl < nmt ? mt0 + tpl : nmt;
            const int c_tile = li < n_ct ? li : n_ct - 1, c_store = li < n_ct ? 1 : 0;
            hipError_t e;
            switch ((n_c + 3) / 4) {
#define DMF_NKC(X_) e = launch_cm_nkc<X_>(V, D16, SD, ND, Rtp, alpha, state, N, S, n_c, n_u, cm, col0, Sp, c_tile, c_store, mt0, mt1, st)
                case 0: DMF_NKC(0); break;
                case 1: DMF_NKC(1); break;
                case 2: DMF_NKC(2); break;
                case 3: DMF_NKC(3); break;
                case 4: DMF_NKC(4); break;
                default: DMF_NKC(kCmNkcWide); break;
#undef DMF_NKC
            }
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace dmf
