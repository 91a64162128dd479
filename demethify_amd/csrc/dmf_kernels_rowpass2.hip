// Row pass, second generation: the u phase of one outer iteration (deconvolution.py:81-90 / :157-164) plus the
// part of the alpha phase's right-hand side that needs V (b_u = u^T (D * V)) in ONE read of V (f64) and of the
// counts stored as u16 (D16: exact, 2 bytes per element instead of 8; built once per problem).  The u-dependent
// Gram entries that need no V (cross = Rt^T diag(d) u, uu = u^T diag(d) u) are left to the integer matrix-core
// kernel of dmf_kernels_gram_i8.hip, which reads the counts as 8-bit planes.
//
// Against the first-generation kernel (dmf_kernels_fused.hip) the FP64 work per element drops from 92 to 33 FMA
// (phase C's 58 cross / uu products are gone), which is what that kernel was bound by; a workgroup is one wave per
// 64-sample column group, holds its alpha-derived MFMA operands in registers instead of a 33 KB LDS copy, and needs
// one 16-row tile buffer only, so several workgroups share a CU: while one of them runs the row-local inner iterations
// (one wave, a dependent chain), the others keep the FP64 pipe busy with their contractions.
//
// Per 16-row block, per workgroup (waves = column groups):
//   tile    prefetched global loads (V: 16 B per lane, two rows per instruction; D16: 16 B = 8 counts per lane)
//           -> LDS tile of the wave's own column group (V f64, the counts as the u16 they arrive as)
//   phase A FP64-MFMA contractions on the tile in the row-on-lane layout (as dmf_kernels_rowpass_mfma.hip):
//           E = V - Rt a_known (16x16x4), c = a_unk (D*E)^T (4x4x4, 4 blocks); M = D P^T exactly on the i8 MFMA
//           (16x16x64, counts and P = alpha_j alpha_l as balanced 8-bit digits) -> partial c / M
//   -- barrier X --
//   phase B (one wave, round robin) sums the partials and runs the n_iter2 accelerated projected-gradient steps,
//           lane = (row, unknown); u / u_ go to HBM and u to LDS
//   -- barrier Y --
//   phase C b_u[j][s] += sum_rows u_j (d v) on the 4x4x4 FP64 MFMA, accumulators in registers across all blocks
// Rows beyond N in the last block read a clamped V row and zero counts (D16 is zero-padded to a multiple of 16 rows
// and of 64 columns), so they add nothing; their u is never stored.
//
// Preconditions (checked by the launcher / the solver): 2 <= S <= 256 (odd S: see the tile prefetch), n_c <= 16, n_u <= 4, counts integral and
// <= 32639 (nd = 1: <= 127), alpha within [0, 1] (true of every iterate: columns on the simplex).
#include "dmf_device.h"
#include "dmf_internal.h"
#include "dmf_phaseb.h"
#include "dmf_fixedpoint.h"
#include <cstdlib>

namespace dmf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

// Diagnostic build only (tools/rowpass2_probe.hip defines DMF_STAMPS): per-wave cycle sums of the kernel's segments
// go to a debug buffer of their own; the product build compiles none of it.
#ifdef DMF_STAMPS
#define DMF2_STAMP_DECL unsigned long long st_last = dmf2_stamp(), st_seg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DMF2_STAMP(i) { const unsigned long long st_now = dmf2_stamp(); st_seg[i] += st_now - st_last; st_last = st_now; }
#define DMF2_STAMP_FLUSH if (lane == 0) for (int i_ = 0; i_ < 16; ++i_) stamps_out[((size_t)blockIdx.x * 4 + wave) * 16 + i_] = st_seg[i_];
__device__ __forceinline__ unsigned long long dmf2_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DMF2_SUB_BEGIN unsigned long long st_sub = dmf2_stamp();
#define DMF2_SUB(i) { const unsigned long long st_now = dmf2_stamp(); st_seg[i] += st_now - st_sub; st_sub = st_now; }
#else
#define DMF2_STAMP_DECL
#define DMF2_STAMP(i)
#define DMF2_STAMP_FLUSH
#define DMF2_SUB_BEGIN
#define DMF2_SUB(i)
#endif

#ifndef DMF_V2_STAGGER
#define DMF_V2_STAGGER 12  // x 64 cycles: how long the waves that do not run phase B hold back their prefetch
#endif

namespace {
constexpr int kRowV = 66;  // V tile row: 64 samples + 16 B pad (f64)
constexpr int kRowD = 72;  // count tile row: 64 samples as u16 + 16 B pad (36 dwords: rows 4, 8, 12 apart start 16, 32, 48 banks apart)
constexpr int kTileVBytes2 = 16 * kRowV * 8;
constexpr int kTileDBytes2 = 16 * kRowD * 2;
constexpr int kTileBytes2 = kTileVBytes2 + kTileDBytes2;  // one column group: V (f64), counts (u16, as they arrive)
}  // namespace

// One accelerated projected-gradient step of a row group (deconvolution.py:83-88): (cur, prev) = (u, u_) in,
// prev = the new u out (cur is then u_).  AT_PREV: gradient at the previous iterate (deconvolution.py:163) instead
// of the extrapolated point (:88).  c and M arrive pre-scaled by 1 / l_w (M negated).
// Ms: this lane's row of -M / l_w in ROTATED order -- Ms[r] = -M[j][(j + r) % NU] / l_w -- with the step's own "+ ut"
// folded into Ms[0] at the extrapolated point (deconvolution.py:88; not at :163, where the gradient point differs).
// A wave alone on its SIMD issues an FP64 instruction every 8 cycles and a 32-bit one every 4 (tools/rowpass2_probe:
// 145 cycles per step for 11 FP64 + 10 other instructions, with or without a second workgroup on the CU), so phase B
// costs what it issues: NU - 1 quad rotations (instead of NU broadcasts) and one FMA chain whose last link clamps --
// 6 FP64 + 8 other instructions per step at NU = 4.
template <int NU, bool AT_PREV>
__device__ __forceinline__ void inner_step(double cur, double& prev, double cj, const double (&Ms)[NU], int b_lo, int b_hi,
                                           int t2, int lane0) {
    const double beta = __hiloint2double(__builtin_amdgcn_readlane(b_hi, t2), __builtin_amdgcn_readlane(b_lo, t2));
    const double ut = fma(beta, cur - prev, cur);
    const double x = AT_PREV ? cur : ut;
    if constexpr (NU == 3) {  // (three lanes per row do not tile a quad: shuffles, in the same rotated order)
        const int jb = (threadIdx.x & 63) - lane0;
        double acc = fma(Ms[0], x, AT_PREV ? ut + cj : cj);
        acc = fma(Ms[1], __shfl(x, lane0 + (jb + 1) % 3, 64), acc);
        prev = f_fma_clamp01(Ms[2], __shfl(x, lane0 + (jb + 2) % 3, 64), acc);
    } else if constexpr (NU == 1) {
        prev = f_fma_clamp01(Ms[0], x, AT_PREV ? ut + cj : cj);
    } else {
        double acc = fma(Ms[0], x, AT_PREV ? ut + cj : cj);
#pragma unroll
        for (int r = 1; r < NU - 1; ++r) acc = fma(Ms[r], r == 1 ? f_group_rot<NU, 1>(x) : f_group_rot<NU, 2>(x), acc);
        prev = f_fma_clamp01(Ms[NU - 1], NU == 2 ? f_group_rot<NU, 1>(x) : f_group_rot<NU, 3>(x), acc);
    }
}

template <int NU, bool AT_PREV>
__device__ __forceinline__ void inner_steps(double& uu, double& up, double cj, const double (&Ms)[NU], int b_lo, int b_hi,
                                            int t_end, int lane0) {
    int t2 = 0;
    for (; t2 + 1 < t_end; t2 += 2) {
        inner_step<NU, AT_PREV>(uu, up, cj, Ms, b_lo, b_hi, t2, lane0);      // new u in `up`, previous u in `uu`
        inner_step<NU, AT_PREV>(up, uu, cj, Ms, b_lo, b_hi, t2 + 1, lane0);  // and back
    }
    if (t2 < t_end) {
        inner_step<NU, AT_PREV>(uu, up, cj, Ms, b_lo, b_hi, t2, lane0);
        const double tmp = uu;
        uu = up;
        up = tmp;
    }
}

// MAXW: most waves (64-sample column groups) a workgroup may have.  4: S <= 256, two workgroups per CU.  8: S <= 512, ONE
// workgroup of up to eight waves per CU -- nothing hides its phase B, but a block then carries twice the samples, so the
// time per element is that of the four-wave form (measured: DESIGN.md section 5).
template <int NKC, int NU, int MAXW = 4>
__global__ __launch_bounds__(64 * MAXW, MAXW == 4 ? 2 : 1) void k_rowpass_v2(
    const double* __restrict__ V, const unsigned short* __restrict__ D16, int SD, const double* __restrict__ Rtp,
    const double* __restrict__ alpha, double* __restrict__ u, double* __restrict__ u_prev,
    const SolverState* __restrict__ state, int64_t N, int S, int n_c, int n_iter2, int mode, int nd,
    double* __restrict__ slab, double* __restrict__ u2_partials
#ifdef DMF_STAMPS
    , unsigned long long* __restrict__ stamps_out
#endif
    ) {
    static_assert(NU >= 1 && NU <= 4, "one phase-B pass per block, 4x4x4 MFMA for the c product");
    constexpr int NCT = 4 * NKC;
    constexpr int NP = NU * (NU + 1) / 2;
    // a (row, unknown j) slot of a wave's partial sums: { c_j, M_jj, M_j(j+1), .. (round the row) } pre-scaled by 1 / l_w (M negated),
    // padded to whole 16-byte pieces -- phase B fetches its lane's slot with SLOT / 2 ds_read_b128 per column group
    constexpr int SLOT = (NU + 2) & ~1;
    constexpr int NV = NU * SLOT;  // doubles per row
    extern __shared__ double lds_dyn[];
    if (state->done) return;

    const int NW = blockDim.x >> 6;  // column groups = waves
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int wcol0 = wave * 64;

    // LDS carve-up (doubles unless noted): beta[n_iter2 (even)] | ubuf[16][NU] | red[MAXW][16][NU][SLOT] | u2[MAXW] |
    //   tiles[NW]{ V f64 [16][66], counts u16 [16][72] }
    double* __restrict__ beta_tab = lds_dyn;
    double* __restrict__ ubuf = beta_tab + ((n_iter2 + 1) & ~1);
    double* __restrict__ red = ubuf + 16 * NU;
    double* __restrict__ u2red = red + MAXW * NV * 16;
    char* __restrict__ tile0 = reinterpret_cast<char*>(u2red + MAXW);
    char* __restrict__ tile = tile0 + (size_t)wave * kTileBytes2;
    double* __restrict__ tileV = reinterpret_cast<double*>(tile);
    // the counts stay u16 in the tile (LINEAR rows: no piece swap): phase A and phase C convert what they read, the integer
    // product builds its balanced byte digits from 32 bytes of a row -- converting at the tile store (f32 copy + two digit
    // planes: 52 vector instructions and 8 LDS writes per block and wave, on every wave's critical path) cost 9 % of the kernel
    unsigned short* __restrict__ tileD = reinterpret_cast<unsigned short*>(tile + kTileVBytes2);

    // momentum coefficients of the inner steps (deconvolution.py:83-85): from the host's row of ratios when there is one
    // (SolverState), else the recurrence itself by one thread -- n_iter2 square roots and divisions in a row, ~5 us
    if (const double* __restrict__ mrow = momentum_row(state, n_iter2)) {
        const double lw_prev = state->l_w_prev, lw = state->l_w;
        for (int t2 = threadIdx.x; t2 < n_iter2; t2 += blockDim.x)
            beta_tab[t2] = fmin(mrow[2 + t2], t2 == 0 ? 0.9999 * sqrt(lw_prev / lw) : 0.9999);  // l_w_ = l_w behind step 0 (:89)
    } else if (threadIdx.x == 0) {
        double a1 = state->a1, lw_prev = state->l_w_prev;
        const double lw = state->l_w;
        for (int t2 = 0; t2 < n_iter2; ++t2) {
            double beta;
            momentum_step(a1, lw_prev, lw, beta);
            beta_tab[t2] = beta;
            lw_prev = lw;
        }
    }
    // the partial-sum slots of column groups this workgroup does not have stay zero (phase B adds all MAXW of them,
    // unrolled, so that its LDS reads go out in one batch)
    for (int i = threadIdx.x; i < MAXW * NV * 16; i += blockDim.x) red[i] = 0.0;
    // x / l_w as x * (1 / l_w), folded into the operands that produce c and M: <= 1 ulp from the division
    const double inv_lw = 1.0 / state->l_w;
    // M arrives as an exact integer multiple of 2^-52; phase B adds -M x / l_w.  (Wave-uniform: kept on the scalar side.)
    const double m_scale_v = -inv_lw * 0x1p-52;
    const double m_scale = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(m_scale_v)),
                                            __builtin_amdgcn_readfirstlane(__double2loint(m_scale_v)));

    // ---- alpha-derived MFMA A operands of this wave's four 16-sample strips, in registers for the whole launch
    const int m16 = lane & 15, q = lane >> 4;
    // Tile swizzle.  A 16-byte LDS read is served in four groups of 16 lanes that are NOT the four rows of 16 lanes
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same + 32): with lane = (row m16, sample piece q) a group holds
    // rows 0-3 / 12-15 at piece q0 and rows 4-11 at piece q0 + 1 (or the other way round), and in any LINEAR layout two of
    // those 16 lanes share a bank.  So rows 4..11 of every tile keep their 16-byte pieces pairwise swapped (a sample s of
    // such a row sits where s ^ 4 would; V only: the u16 counts are read in 8-byte and 2-byte pieces, which a linear row with a
    // 36-dword stride serves without conflicts) -- then each group
    // reads 16 different rows at one piece offset, and the odd piece stride (33) makes that conflict-free.
    const int sw_row = (m16 >= 4 && m16 < 12) ? 1 : 0;  // is row m16 a swapped row (phase A: lane = (row, piece))
    const int qs = q ^ sw_row;                           // where this lane's piece q sits in row m16
    // Phase C reads with lane = (row quad member q, sample m16): rows R + 8 (q & 1) + 4 (q >> 1), R = 0..3, so that the two
    // rows a 32-lane group of a ds_read_b64 touches lie 8 apart -- 8 x 132 dwords = 32 banks: the halves of the 64
    const int c_row = 8 * (q & 1) + 4 * (q >> 1);
    const int mC = m16 ^ ((q == 1 || q == 2) ? 4 : 0);   // (rows 8..11 and 4..7 are swapped rows)
    const int e_col = wcol0 + 4 * (m16 & 3) + (m16 >> 2);  // + 16 t: first product, m <-> sample
    const int k_col = wcol0 + 4 * q;                        // + 16 t + r: k-step r, k = q <-> sample
    double a1r[4][NKC > 0 ? NKC : 1];  // -alpha_known[k = 4 kc + q][sample]
    double a2r[4][4];                  // alpha_unk[m16 & 3][sample] (c product, 4x4x4: block = 4 rows, i = unknown)
    // M product (M_i[pair] = sum_s alpha_j alpha_l d_is) on the INTEGER matrix cores, exactly: the counts are one or two
    // balanced 8-bit digits, P = alpha_j alpha_l in [0, 1] seven digits of rint(P 2^52) (dmf_fixedpoint.h), one
    // v_mfma_i32_16x16x64_i8 per digit covers the wave's 64 samples (A = count digits [row][sample], B = digit t of
    // P [sample][pair]).  An FP64 MFMA holds the SIMD's FP64 pipe for 64 cycles, and this product was 16 of the 28 per
    // block and wave -- time during which the other workgroup's phases B and C on the same SIMD could not issue.
    v4i pdg[7];  // B operands: digit t of P[pair = m16][samples 16 q .. 16 q + 15 of this column group]
    int pj = 0, pl = 0;  // pair m16 = (pj <= pl)
    while ((pl + 1) * (pl + 2) / 2 <= m16) ++pl;
    pj = m16 - pl * (pl + 1) / 2;
    {
        const bool pair_ok = m16 < NP;
        const bool a2_ok = (m16 & 3) < NU;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int row = kc * 4 + q, col = e_col + 16 * t;
                a1r[t][kc] = (row < n_c && col < S) ? -alpha[(int64_t)row * S + col] : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = k_col + 16 * t + r;
                a2r[t][r] = (col < S && a2_ok) ? alpha[(int64_t)(n_c + (m16 & 3)) * S + col] * inv_lw : 0.0;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // four samples per dword of each digit
            unsigned int lo[4], hi[4], tl[4], th[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = wcol0 + 16 * q + 4 * g + i;
                const bool in = pair_ok && col < S;
                const double aj = in ? alpha[(int64_t)(n_c + pj) * S + col] : 0.0;
                const double al = in ? alpha[(int64_t)(n_c + pl) * S + col] : 0.0;
                z_to_biased(aj, al, lo[i], hi[i]);
            }
            transpose4(lo, tl);
            transpose4(hi, th);
            pdg[0][g] = (int)(tl[0] ^ 0x80808080u);
            pdg[1][g] = (int)(tl[1] ^ 0x80808080u);
            pdg[2][g] = (int)(tl[2] ^ 0x80808080u);
            pdg[3][g] = (int)(tl[3] ^ 0x80808080u);
            pdg[4][g] = (int)(th[0] ^ 0x80808080u);
            pdg[5][g] = (int)(th[1] ^ 0x80808080u);
            pdg[6][g] = (int)th[2];
        }
    }
    __syncthreads();  // beta_tab, the zeroed slots

    const int64_t nblk = (N + 15) / 16;
    const int nk = (int)((nblk - blockIdx.x + gridDim.x - 1) / gridDim.x);  // blocks of this workgroup

    // global -> register staging: V load i covers rows 2i, 2i+1 (lane -> row half, 2 samples);
    // D16 load i covers rows 8i .. 8i+7 (lane -> row lane >> 3, 8 samples)
    const int ld_row = lane >> 5;
    const int ld_col = (lane & 31) * 2;
    int ld_gcol = wcol0 + ld_col;
    // ragged last column group: clamped samples meet zero counts.  Odd S: the row's last sample shares its pair with the
    // next row's first element (or, on the last row, with a zero from the descriptor's range check) -- a finite value
    // against a zero count; rows then start 8 bytes off a 16-byte boundary every other time, which 16-byte buffer loads
    // take (tools/align_probe.hip).
    const int last_pair = (S - 1) & ~1;
    if (ld_gcol > last_pair) ld_gcol = last_pair;
    const int d_row = lane >> 3, d_col = (lane & 7) * 8;
    // (tile swizzle at the stores: V rows 2 i, 2 i + 1 are swapped rows for i = 2..5)
    const int ld_col_sw = ((lane & 31) ^ 2) * 2;
    v2d pv[8];
    v4u pd[2];
    double nrt[NKC > 0 ? NKC : 1];
    double pu, pup;  // u / u_ of lane (row, unknown) for the wave that runs the block's inner iterations
    constexpr int RPWB = 64 / NU;  // >= 16 rows per wave: one phase-B pass covers the block
    const int rl = lane / NU, jb = lane - rl * NU;
    const bool b_lane = rl < 16 && rl < RPWB;
    // The block's loads are BUFFER loads: address = descriptor base (scalar: array + block offset, recomputed per block on
    // the scalar unit) + a loop-invariant per-lane byte offset + a scalar / immediate offset per instruction -- no vector
    // address arithmetic at all (the flat-address form of this prefetch cost ~80 vector instructions per block and wave,
    // and the row pass is bound by the SIMD's issue slots).  The descriptor's range check replaces the clamps of the
    // last, partial block: rows at or beyond N read as zero, and their counts are zero.
    const unsigned int v_off = (unsigned int)(ld_row * S + ld_gcol) * 8u;               // V: row ld_row of a row pair
    const unsigned int d_off = (unsigned int)(d_row * SD + wcol0 + d_col) * 2u;         // counts: row d_row of eight
    const unsigned int r_off = (unsigned int)(m16 * NCT + q) * 8u;                      // R_trunc (padded copy): row m16
    const unsigned int u_off = b_lane ? (unsigned int)lane * 8u : 0xFFFFFFF0u;          // u / u_: (row, unknown) = lane
    auto span = [](int64_t bytes) { return (unsigned int)(bytes < 0 ? 0 : (bytes > 0x7FFFFFFF ? 0x7FFFFFFF : bytes)); };
    auto prefetch = [&](int64_t blk) {
        const int64_t r0 = blk * 16;
        const int64_t left = N - r0;  // rows of the arrays from this block on (> 0)
        {
            // FIRST in the batch: the tile store's wait for the (younger) tile loads then covers them, and phase B
            // finds its u / u_ complete without a vmcnt wait of its own -- a wait there would also cover the NEXT
            // block's prefetch, issued just before barrier X, and put a whole HBM round trip on the critical path.
            const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)(u + r0 * NU), 0, span(left * NU * 8), 0x00020000);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(u_prev + r0 * NU), 0, span(left * NU * 8), 0x00020000);
            const v2u a = __builtin_amdgcn_raw_buffer_load_b64(ru, u_off, 0, 0);
            const v2u b = __builtin_amdgcn_raw_buffer_load_b64(rp, u_off, 0, 0);
            pu = __hiloint2double((int)a.y, (int)a.x);
            pup = __hiloint2double((int)b.y, (int)b.x);
        }
        if (NKC > 0) {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(Rtp + r0 * NCT), 0, span(left * NCT * 8), 0x00020000);
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const v2u a = __builtin_amdgcn_raw_buffer_load_b64(rr, r_off, kc * 32, 0);
                nrt[kc] = __hiloint2double((int)a.y, (int)a.x);
            }
        }
        {
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(V + r0 * S), 0, span(left * S * 8), 0x00020000);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const v4u a = __builtin_amdgcn_raw_buffer_load_b128(rv, v_off, 2 * i * S * 8, 0);
                pv[i] = v2d{__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)a.w, (int)a.z)};
            }
        }
        {
            // (the count copy is zero-padded to whole blocks of 16 rows: always in range)
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(D16 + r0 * SD), 0, span((int64_t)16 * SD * 2), 0x00020000);
            pd[0] = __builtin_amdgcn_raw_buffer_load_b128(rd, d_off, 0, 0);
            pd[1] = __builtin_amdgcn_raw_buffer_load_b128(rd, d_off, 8 * SD * 2, 0);
        }
    };
    if (nk > 0) prefetch(blockIdx.x);

    double bu[4] = {0.0, 0.0, 0.0, 0.0};  // b_u[unknown q][sample 16 t + m16] of this wave's column group, t = 0..3
    // this lane's places in its wave's partial-sum slots: c[unknown q][row m16]; M[pair m16 = (pj, pl)][rows 4 q ..] twice
    // (M is symmetric), a slot's M entries in ROTATED order -- slot (row, j) entry r is M[j][(j + r) % NU], the order in
    // which phase B's quad rotations deliver the iterate; rows 4 q + rr at immediate offsets
    double* __restrict__ const mine_c = red + (size_t)wave * NV * 16 + (m16 * NU + q) * SLOT;
    double* __restrict__ const mine_m1 = red + (size_t)wave * NV * 16 + (4 * q * NU + pl) * SLOT + 1 + (pj - pl + NU) % NU;
    double* __restrict__ const mine_m2 = red + (size_t)wave * NV * 16 + (4 * q * NU + pj) * SLOT + 1 + (pl - pj) % NU;
    double u2_acc = 0.0;

    DMF2_STAMP_DECL
    for (int s = 0; s < nk; ++s) {
        const int64_t blk = blockIdx.x + (int64_t)s * gridDim.x;
        const int64_t row0 = blk * 16;
        // ---- tile store (waits for the prefetched loads)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<v2d*>(tileV + (2 * i + ld_row) * kRowV + ((i >= 2 && i < 6) ? ld_col_sw : ld_col)) = pv[i];
#ifndef DMF_ABLATE_DSTORE
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<v4u*>(tileD + (8 * i + d_row) * kRowD + d_col) = pd[i];
#endif
        double rtop[NKC > 0 ? NKC : 1];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) rtop[kc] = nrt[kc];  // B operand of the first product: Rt^T[k = 4 kc + q][n = row]
        // the block's u / u_ arrived with the tile (the next prefetch reuses pu / pup before phase B runs)
        const bool my_turn = wave == s % NW;
        const bool ok = b_lane && row0 + rl < N;
        const double uu0 = pu;
        const double up0 = pup;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(0);
        DMF2_STAMP(0)  // tile store (vmcnt wait for the prefetch)

        // ---- phase A: strips of 16 samples; the LDS reads of strip t + 1 are issued before the MFMAs of strip t,
        // and the E chain of strip t + 1 is slotted between the c / M MFMAs of strip t (a dependent FP64 MFMA
        // stalls behind its producer)
        struct Strip {
            v2d v01, v23;
            v2u dw;  // the four counts of the piece, u16
        };
        auto load_strip = [&](int t, Strip& R) {
            const double* __restrict__ tv = tileV + m16 * kRowV + t * 16 + 4 * qs;
            R.v01 = *reinterpret_cast<const v2d*>(tv);
            R.v23 = *reinterpret_cast<const v2d*>(tv + 2);
            R.dw = *reinterpret_cast<const v2u*>(tileD + m16 * kRowD + t * 16 + 4 * q);
        };
        double csm0 = 0.0, csm1 = 0.0;  // c[unknown q][row m16], one double per lane
        auto e_init = [&](const Strip& R) { return v4d{R.v01.x, R.v01.y, R.v23.x, R.v23.y}; };
        auto run_strip = [&](const Strip& R, v4d e, const Strip& Rn, const double (&a1n)[NKC > 0 ? NKC : 1],
                             const double (&a2)[4], bool has_next) {
            const v4d d = {(double)(R.dw.x & 0xFFFFu), (double)(R.dw.x >> 16), (double)(R.dw.y & 0xFFFFu), (double)(R.dw.y >> 16)};
            const v4d w = d * e;
            v4d en = e_init(Rn);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r & 1) csm1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[r], w[r], csm1, 0, 0, 0);
                else csm0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[r], w[r], csm0, 0, 0, 0);
                if (has_next && r < NKC) en = __builtin_amdgcn_mfma_f64_16x16x4f64(a1n[r], rtop[r], en, 0, 0, 0);
            }
            return en;
        };
#ifdef DMF_ABLATE_E
        const double csm = tileV[m16 * kRowV + 4 * qs];
#else
        Strip sa, sb;
        load_strip(0, sa);
        load_strip(1, sb);
        __builtin_amdgcn_sched_barrier(0);
        v4d e0 = e_init(sa);
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1r[0][kc], rtop[kc], e0, 0, 0, 0);
        const v4d e1 = run_strip(sa, e0, sb, a1r[1], a2r[0], true);
        load_strip(2, sa);
        __builtin_amdgcn_sched_barrier(0);
        const v4d e2 = run_strip(sb, e1, sa, a1r[2], a2r[1], true);
        load_strip(3, sb);
        __builtin_amdgcn_sched_barrier(0);
        const v4d e3 = run_strip(sa, e2, sb, a1r[3], a2r[2], true);
        (void)run_strip(sb, e3, sb, a1r[3], a2r[3], false);
        const double csm = csm0 + csm1;
#endif
        // M on the integer matrix cores: digit weights 256^0 .. 256^7 (count digit d + P digit t -> weight t + d)
        v4i mw[8];
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) mw[w8] = v4i{0, 0, 0, 0};
#ifndef DMF_ABLATE_M  // (diagnostic builds of tools/rowpass2_probe.hip leave pieces out to see what they cost)
        {
            // A: counts [row m16][samples 16 q .. 16 q + 15] as balanced digits: d + 128 = b0 + 256 b1, digit 0 = b0 - 128
            // (b0 ^ 0x80 as i8), digit 1 = b1
            const v4u wa = *reinterpret_cast<const v4u*>(tileD + m16 * kRowD + 16 * q);
            const v4u wb = *reinterpret_cast<const v4u*>(tileD + m16 * kRowD + 16 * q + 8);
            const unsigned int e0 = wa.x + 0x00800080u, e1 = wa.y + 0x00800080u, e2 = wa.z + 0x00800080u, e3 = wa.w + 0x00800080u;
            const unsigned int e4 = wb.x + 0x00800080u, e5 = wb.y + 0x00800080u, e6 = wb.z + 0x00800080u, e7 = wb.w + 0x00800080u;
            const v4i c0 = {(int)(__builtin_amdgcn_perm(e1, e0, 0x06040200u) ^ 0x80808080u),
                            (int)(__builtin_amdgcn_perm(e3, e2, 0x06040200u) ^ 0x80808080u),
                            (int)(__builtin_amdgcn_perm(e5, e4, 0x06040200u) ^ 0x80808080u),
                            (int)(__builtin_amdgcn_perm(e7, e6, 0x06040200u) ^ 0x80808080u)};
#pragma unroll
            for (int t = 0; t < 7; ++t) mw[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(c0, pdg[t], mw[t], 0, 0, 0);
            if (nd == 2) {
                const v4i c1 = {(int)__builtin_amdgcn_perm(e1, e0, 0x07050301u), (int)__builtin_amdgcn_perm(e3, e2, 0x07050301u),
                                (int)__builtin_amdgcn_perm(e5, e4, 0x07050301u), (int)__builtin_amdgcn_perm(e7, e6, 0x07050301u)};
#pragma unroll
                for (int t = 0; t < 7; ++t) mw[t + 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(c1, pdg[t], mw[t + 1], 0, 0, 0);
            }
        }
#endif
        // lane (pair m16, q) holds rows 4 q + reg: the exact integer sum_w 256^w mw[w] in two halves that fit a double
        // without rounding (|mw| < 2^21 per digit product sum), one rounding when they are joined
        double mrow[4];
#ifdef DMF_ABLATE_M
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) mrow[rr] = 0.25 * (rr == (m16 & 3));
#else
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const double lo = fma(fma(fma((double)mw[3][rr], 256.0, (double)mw[2][rr]), 256.0, (double)mw[1][rr]), 256.0, (double)mw[0][rr]);
            const double hi = fma(fma(fma((double)mw[7][rr], 256.0, (double)mw[6][rr]), 256.0, (double)mw[5][rr]), 256.0, (double)mw[4][rr]);
            mrow[rr] = fma(hi, 0x1p32, lo) * m_scale;
        }
#endif
        DMF2_STAMP(1)  // phase A
        // The next block's global loads (their staging registers were free during phase A).  A workgroup's waves reach
        // this point together, and 42 KB of loads take the CU's one address unit ~700 cycles: only the wave that runs
        // phase B issues its loads here; the others issue theirs behind barrier X, while they wait for phase B anyway.
        {
            if (q < NU) *mine_c = csm;  // c[unknown q][row m16]
            if (m16 < NP) {  // C layout of the 16x16x64 tile: col = pair m16 = (pj, pl), rows 4 q + reg; M is symmetric
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) mine_m1[rr * NU * SLOT] = mrow[rr];
                if (pj != pl) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) mine_m2[rr * NU * SLOT] = mrow[rr];
                }
            }
        }
        DMF2_STAMP(2)  // partials
        // (a bare barrier behind an LDS-only wait: __syncthreads() would also drain vmcnt)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ---- barrier X
        DMF2_STAMP(3)  // wait X
        // The next block's global loads (their staging registers were free during phase A).  42 KB of loads take the CU's
        // one address unit ~700 cycles when a workgroup's waves issue them together: the wave that runs phase B goes
        // first, the others -- who wait for phase B anyway -- a little later.  (ONE place for all waves: with the loads in
        // two branches the register allocator gave them different destinations and put copies -- and the waits for the
        // data -- behind one of them.)
#ifndef DMF_V2_NO_STAGGER
        if (!my_turn) __builtin_amdgcn_s_sleep(DMF_V2_STAGGER);
#endif
        if (s + 1 < nk) prefetch(blk + gridDim.x);
        DMF2_STAMP(7)  // prefetch issue

        // ---- phase B: row-local inner iterations, lane = (row, unknown j); c and M pre-scaled by 1 / l_w
        if (my_turn) {
            __builtin_amdgcn_s_setprio(3);  // a dependent chain on the workgroup's critical path
            DMF2_SUB_BEGIN
            const int lane0 = lane - jb;
            const int rlc = rl < 16 ? rl : 15;
            double cj = 0.0, Ms[NU];
#pragma unroll
            for (int l = 0; l < NU; ++l) Ms[l] = 0.0;
            {
                // this lane's slot of every column group, in column-group order (a fixed summation order); the values
                // are already c / l_w and -M / l_w
                const double* __restrict__ slot = red + (rlc * NU + jb) * SLOT;
#pragma unroll 1
                for (int w0 = 0; w0 < MAXW; w0 += 4) {  // (batches of four column groups: registers)
                    v2d part[4][SLOT / 2];
#pragma unroll
                    for (int w = 0; w < 4; ++w)
#pragma unroll
                        for (int h = 0; h < SLOT / 2; ++h)
                            part[w][h] = *reinterpret_cast<const v2d*>(slot + (w0 + w) * NV * 16 + 2 * h);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        cj += part[w][0].x;
#pragma unroll
                        for (int l = 0; l < NU; ++l) Ms[l] += (l & 1) ? part[w][(l + 1) / 2].x : part[w][l / 2].y;
                    }
                }
            }
            double uu = uu0, up = up0;
            DMF2_SUB(8)
            // The momentum coefficients ride in a VGPR (lane t holds beta_t) and reach the loop through
            // v_readlane: an LDS read here would sit on the dependent chain every step.
            for (int t0 = 0; t0 < n_iter2; t0 += 64) {
                const int tl = t0 + lane < n_iter2 ? t0 + lane : n_iter2 - 1;
                const double bvec = beta_tab[tl];  // (lane t: beta_t; the first read goes out with the partial-sum reads)
                const int b_lo = __double2loint(bvec), b_hi = __double2hiint(bvec);
                const int t_end = n_iter2 - t0 < 64 ? n_iter2 - t0 : 64;
                // (u, u_) swap roles every step: written out in pairs so that no register copies sit on the chain
                // (the loop holds v_readlane, a convergent operation the unroller will not split by itself)
                if (mode == 1) {
                    inner_steps<NU, true>(uu, up, cj, Ms, b_lo, b_hi, t_end, lane0);
                } else {
                    if (t0 == 0) Ms[0] += 1.0;  // the step's own "+ ut" (deconvolution.py:88), folded into the diagonal
                    inner_steps<NU, false>(uu, up, cj, Ms, b_lo, b_hi, t_end, lane0);
                }
            }
            DMF2_SUB(9)
            if (b_lane) ubuf[rl * NU + jb] = ok ? uu : 0.0;  // rows beyond N: phase C multiplies them by zero counts
            if (ok) {
                // (buffer stores, like the loads: scalar base of the block + this lane's constant offset)
                const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)(u + row0 * NU), 0, span((N - row0) * NU * 8), 0x00020000);
                const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(u_prev + row0 * NU), 0, span((N - row0) * NU * 8), 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b64(v2u{(unsigned int)__double2loint(uu), (unsigned int)__double2hiint(uu)}, ru, u_off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(v2u{(unsigned int)__double2loint(up), (unsigned int)__double2hiint(up)}, rp, u_off, 0, 0);
                u2_acc = fma(uu, uu, u2_acc);
            }
            DMF2_SUB(10)
            __builtin_amdgcn_s_setprio(0);
        }
        DMF2_STAMP(4)  // phase B (or nothing)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ---- barrier Y
        DMF2_STAMP(5)  // wait Y

        // ---- phase C: b_u[j][s] += sum_rows u[row][j] (d v)[row][s] on the 4x4x4 (4 blocks) FP64 MFMA: block = sample
        // quad of a 16-sample strip, i = unknown, j = sample in the quad, k = row in a quad of rows (A[b][i][k] sits
        // at lane 16 k + 4 b + i, B[b][k][j] at lane 16 k + 4 b + j, the result D[b][i][j] at lane 16 i + 4 b + j:
        // tools/mfma_probe.hip).  k may number the rows of a quad in any order: lane (q, m16) reads (d v) of row
        // R + c_row(q), sample 16 t + m16 (at its swizzled place) and u of that row, unknown m16 & 3; the 32 tile reads of
        // the block are independent and go out in two batches (a lane = sample loop with per-row broadcast reads of u
        // spent ~1.6k cycles per block on LDS round trips).
#ifndef DMF_ABLATE_C
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            double vv[2][4], ua[2];
            unsigned short dd[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = 2 * half + rr + c_row;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    vv[rr][t] = tileV[row * kRowV + 16 * t + mC];
                    dd[rr][t] = tileD[row * kRowD + 16 * t + m16];
                }
                ua[rr] = (m16 & 3) < NU ? ubuf[row * NU + (m16 & 3)] : 0.0;
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    bu[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(ua[rr], (double)dd[rr][t] * vv[rr][t], bu[t], 0, 0, 0);
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        DMF2_STAMP(6)  // phase C
        // (the next iteration's tile store touches this wave's own tile only; ubuf and red are rewritten behind
        // the next barrier X / by phase A after this wave's own phase C)
    }

    DMF2_STAMP_FLUSH
    const double w2 = wave_sum(u2_acc);
    if (lane == 0) u2red[wave] = w2;
    __syncthreads();
    // ---- b_u slab of this workgroup [NU][S] and its share of ||u||_F^2
    if (q < NU) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int sC = wcol0 + 16 * t + m16;
            if (sC < S) slab[((int64_t)blockIdx.x * NU + q) * S + sC] = bu[t];
        }
    }
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += u2red[w];
        u2_partials[blockIdx.x] = tot;
    }
}

size_t rowpass_v2_lds_bytes(int S, int n_u, int n_iter2) {
    const int NW = (S + 63) / 64;
    const int maxw = NW <= 4 ? 4 : 8;
    const int nv = n_u * ((n_u + 2) & ~1);  // a row's partial-sum slots (k_rowpass_v2: NV)
    const size_t doubles = (size_t)((n_iter2 + 1) & ~1) + 16 * n_u + (size_t)maxw * nv * 16 + maxw;
    return doubles * sizeof(double) + (size_t)NW * kTileBytes2;
}

bool rowpass_v2_supported(int S, int n_c, int n_u, int n_iter2) {
    if (S < 2 || S > 512 || n_c > 16 || n_u < 1 || n_u > 4) return false;
    // up to 256 samples: two workgroups per CU within 160 KB; beyond: one workgroup of up to eight waves
    return rowpass_v2_lds_bytes(S, n_u, n_iter2) <= (size_t)(S <= 256 ? 80 : 160) * 1024;
}

int rowpass_v2_grid(int64_t N, int S) {
    const int NW = (S + 63) / 64;
    int per_cu = NW > 4 ? 1 : 8 / NW;  // two waves per SIMD: NW = 5..8 -> 1, 4 -> 2, 3 -> 2, 2 -> 4, 1 -> 8 workgroups per CU
#ifdef DMF_EXPERIMENT  // (an experiment build only: DMF_EXPERIMENT=1 python -m demethify_amd._build)
    if (const char* v = getenv("DMF_V2_PER_CU")) per_cu = atoi(v) > 0 ? atoi(v) : per_cu;  // (experiments)
#endif
    const int64_t nblk = (N + 15) / 16;
    const int64_t g = 256 * per_cu;
    return (int)(nblk < g ? nblk : g);
}

template <int NKC, int NU, int MAXW>
static hipError_t launch_v2_t(const double* V, const unsigned short* D16, int SD, const double* Rtp, const double* alpha,
                              double* u, double* u_prev, SolverState* state, int64_t N, int S, int n_c, int n_iter2,
                              int mode, int nd, double* slab, double* u2_partials, int* grid_out, hipStream_t st) {
    const int NW = (S + 63) / 64;
    const size_t lds = rowpass_v2_lds_bytes(S, NU, n_iter2);
    constexpr size_t kLdsCap = (size_t)(MAXW == 4 ? 80 : 160) * 1024;
    if (NW > MAXW || (MAXW == 8 && NW <= 4) || lds > kLdsCap || N < 1 || SD < NW * 64 || (SD & 7) != 0 || nd < 1 || nd > 2)
        return hipErrorInvalidValue;
    static bool lds_limit_raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > 48 * 1024 && !lds_limit_raised[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)k_rowpass_v2<NKC, NU, MAXW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCap);
        if (e != hipSuccess) return e;
        lds_limit_raised[dev] = true;
    }
    const int grid = rowpass_v2_grid(N, S);
    *grid_out = grid;
    hipLaunchKernelGGL((k_rowpass_v2<NKC, NU, MAXW>), dim3(grid), dim3(NW * 64), lds, st, V, D16, SD, Rtp, alpha, u, u_prev,
                       state, N, S, n_c, n_iter2, mode, nd, slab, u2_partials
#ifdef DMF_STAMPS
                       , (unsigned long long*)nullptr
#endif
                       );
    return hipGetLastError();
}

template <int NKC>
static hipError_t launch_v2_nkc(int n_u, const double* V, const unsigned short* D16, int SD, const double* Rtp,
                                const double* alpha, double* u, double* u_prev, SolverState* state, int64_t N, int S,
                                int n_c, int n_iter2, int mode, int nd, double* slab, double* u2_partials, int* grid_out,
                                hipStream_t st) {
    switch (n_u) {
#define DMF_CASE(NU_)                                                                                                   \
    case NU_:                                                                                                           \
        return S <= 256 ? launch_v2_t<NKC, NU_, 4>(V, D16, SD, Rtp, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, nd, \
                                                   slab, u2_partials, grid_out, st)                                     \
                        : launch_v2_t<NKC, NU_, 8>(V, D16, SD, Rtp, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, nd, \
                                                   slab, u2_partials, grid_out, st);
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4)
#undef DMF_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_rowpass_v2(const double* V, const unsigned short* D16, int SD, const double* Rtp, const double* alpha,
                             double* u, double* u_prev, SolverState* state, int64_t N, int S, int n_c, int n_u,
                             int n_iter2, int mode, int nd, double* slab, double* u2_partials, int* grid_out, hipStream_t st) {
    switch ((n_c + 3) / 4) {
#define DMF_NKC(X)                                                                                                \
    case X:                                                                                                       \
        return launch_v2_nkc<X>(n_u, V, D16, SD, Rtp, alpha, u, u_prev, state, N, S, n_c, n_iter2, mode, nd, slab, \
                                u2_partials, grid_out, st);
        DMF_NKC(0) DMF_NKC(1) DMF_NKC(2) DMF_NKC(3) DMF_NKC(4)
#undef DMF_NKC
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dmf
