// Fused row pass: ONE read of V and D per outer iteration.
//
// A persistent workgroup walks 16-row blocks.  Per block:
//   load    coalesced global loads (16 B per lane, two rows per instruction) -> registers -> LDS tile
//   phase A read the tile in the row-on-lane layout and run the three FP64-MFMA contractions of
//           dmf_kernels_rowpass_mfma.hip (E = V - Rt a_known, c = a_unk (D*E)^T, M = P D^T)
//   phase B n_iter2 row-local accelerated projected-gradient steps (one wave, round robin); the new u / u_
//           rows go to LDS (a wave of the other team stores them to HBM one step later)
//   phase C read the tile sample-on-lane and accumulate, in registers across all row blocks, the
//           u-dependent entries of the per-sample Gram matrices for the alpha phase
//           (cross[k][j] += d Rt_k u_j, uu[j<=l] += d u_j u_l, bu[j] += d v u_j), as dmf_kernels_gram.hip
// Phases A/B and phase C run on two wave teams of the same workgroup, one block apart (see the kernel's
// comment): FP64 MFMA and FP64 VALU share one pipe on gfx950, so the point of the split is not co-issue but
// keeping that pipe fed while the other team waits on LDS, a barrier or phase B's dependent chain.
// At the end every workgroup stores its accumulators as one slab (job order of the solver's table) and
// its share of ||u||_F^2; k_gram_reduce sums the slabs in fixed order.
//
// Preconditions (checked by the launcher): N % 16 == 0 (the caller runs the ragged tail, < 16 rows, through
// the unfused kernels), S % 4 == 0, S <= 256, n_c <= 16, n_u <= 4, counts exactly representable in f32;
// `Rtp` is the problem's zero-padded copy of R_trunc (row stride 4 NKC).
#include "dmf_device.h"
#include "dmf_internal.h"
#include "dmf_phaseb.h"

namespace dmf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// Diagnostic build only (tools/fused_probe.hip defines DMF_STAMPS): per-wave cycle sums of the
// kernel's segments go to a debug buffer of their own; the product build compiles none of it.
#ifdef DMF_STAMPS
#define DMF_STAMP_DECL unsigned long long st_last = dmf_stamp(), st_seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DMF_STAMP(i) { const unsigned long long st_now = dmf_stamp(); st_seg[i] += st_now - st_last; st_last = st_now; }
#define DMF_ROW_STAMP(rr) if (r_begin == 8 * half && (rr) < 3) DMF_STAMP(5 + (rr))
#define DMF_STAMP_FLUSH if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) stamps_out[((size_t)blockIdx.x * 16 + wave) * 8 + i_] = st_seg[i_];
__device__ __forceinline__ unsigned long long dmf_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#else
#define DMF_STAMP_DECL
#define DMF_ROW_STAMP(rr)
#define DMF_STAMP(i)
#define DMF_STAMP_FLUSH
#endif

#define DMF_WAVES_PER_WG(NW) (3 * (NW))  // A team (NW waves) + C team (2 NW waves)
#ifndef DMF_C_ROWS_BEFORE_X
#define DMF_C_ROWS_BEFORE_X 4
#endif
constexpr int kCRowsBeforeX = DMF_C_ROWS_BEFORE_X;  // of a C wave's 8 rows per block, how many run before barrier X

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTileRowDoubles = 66;  // V tile row: 64 samples + 16 B pad (f64)
constexpr int kTileRowFloats = 68;   // D tile row: 64 samples + 16 B pad (f32: counts < 2^24 are exact)
constexpr int kTileVBytes = 16 * kTileRowDoubles * 8;
constexpr int kTileBytes = kTileVBytes + 16 * kTileRowFloats * 4;  // one column group, one buffer

// Team layout: a workgroup has 3 NW waves (NW = ceil(S / 64) column groups of 64 samples).
//   A team  waves [0, NW): A wave w owns column group w for phase A (MFMA) and takes turns at phase B
//   C team  waves [NW, 3 NW): C wave (g, h) owns column group g and rows [8h, 8h + 8) of every block
// so each SIMD holds one MFMA-bound wave and two VALU-bound waves (a lone wave issues FP64 VALU at about
// two thirds of the rate two waves reach).  Step s overlaps the A team's work on block s with the C team's work on block s - 1:
//   A team  store the prefetched V / D tile of block s into LDS buffer s & 1, phase A -> partial c / M in
//           LDS, then issue the global loads of block s + 1 into registers
//           -- barrier X --   phase B (one A wave, round robin) -> u / u_ rows to LDS   -- barrier Y --
//   C team  store block s - 1's u / u_ rows to HBM, phase C on the first 4 of its rows of block s - 1
//           -- X --   last 4 rows   -- Y --
// The alpha-derived MFMA A operands are re-read from an LDS copy of alpha for every strip (cheap) instead
// of living in ~90 VGPRs, which is what lets three waves per SIMD fit.
template <int NKC, int NU>
__global__ __launch_bounds__(768) void k_rowpass_fused(
    const double* __restrict__ V, const double* __restrict__ D, const double* __restrict__ Rtp,
    const double* __restrict__ alpha, double* __restrict__ u, double* __restrict__ u_prev,
    const SolverState* __restrict__ state, int64_t N, int S, int n_c, int n_iter2, int mode,
    double* __restrict__ slab, double* __restrict__ u2_partials
#ifdef DMF_STAMPS
    , unsigned long long* __restrict__ stamps_out
#endif
    ) {
    constexpr int NCT = 4 * NKC;
    constexpr int NCTL = NCT > 0 ? NCT : 1;
    constexpr int NP = NU * (NU + 1) / 2;
    constexpr int NMT = (NP + 15) / 16;
    constexpr int NV = NU + NP;
    constexpr int NACC = NCT * NU + NP + NU;
    extern __shared__ double lds_dyn[];
    if (state->done) return;

    const int NW = blockDim.x / 192;  // column groups
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool a_team = wave < NW;
    const int cidx = a_team ? 0 : wave - NW;
    const int cg = a_team ? wave : cidx % NW;  // column group of this wave
    const int wcol0 = cg * 64;

    // LDS carve-up (doubles unless noted):
    //   beta[n_iter2 (even)] | ubuf[2][16][NU] | upbuf[2][16][NU] | rtbuf[2][16][NCT] | red[NW][NV][16] |
    //   alds[(NCT + NU + 1) rows][AS]: -alpha_known (zero-padded to NCT rows), alpha_unk, one zero row |
    //   tiles[2 buffers][NW groups]{ V f64 [16][66], D f32 [16][68] }
    const int AS = NW * 64 + 2;  // alpha row stride in LDS
    double* __restrict__ beta_tab = lds_dyn;
    double* __restrict__ ubuf = beta_tab + ((n_iter2 + 1) & ~1);
    double* __restrict__ upbuf = ubuf + 2 * 16 * NU;  // u_ rows, stored to HBM by a C wave
    double* __restrict__ rtbuf = upbuf + 2 * 16 * NU;
    double* __restrict__ red = rtbuf + 2 * 16 * NCTL;
    double* __restrict__ alds = red + NW * NV * 16;
    char* __restrict__ tiles = reinterpret_cast<char*>(alds + (size_t)(NCT + NU + 1) * AS);
    auto tile_of = [&](int buf) { return tiles + (size_t)(buf * NW + cg) * kTileBytes; };
    constexpr int ZROW = NCT + NU;  // index of the all-zero row of alds

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
    for (int i = threadIdx.x; i < (NCT + NU + 1) * AS; i += blockDim.x) {
        const int r = i / AS, c = i - r * AS;
        double val = 0.0;
        if (c < S) {
            if (r < n_c) val = -alpha[(int64_t)r * S + c];
            else if (r >= NCT && r < NCT + NU) val = alpha[(int64_t)(n_c + r - NCT) * S + c];
        }
        alds[i] = val;
    }
    __syncthreads();

    const int64_t nblk = N / 16;  // N % 16 == 0 (launcher's contract)
    const int nk = (int)((nblk - blockIdx.x + gridDim.x - 1) / gridDim.x);  // blocks of this workgroup

    if (a_team) {
        // =========================== A team: phases A and B ===================================
        const int m16 = lane & 15, q = lane >> 4;
        const double inv_lw = 1.0 / state->l_w;  // x / l_w as x * (1 / l_w): <= 1 ulp from the division
        // LDS rows / columns this lane reads to form the MFMA A operands of a strip
        const int e_col = wcol0 + 4 * (m16 & 3) + (m16 >> 2);   // + 16 t : first product, m <-> sample
        const int k_col = wcol0 + 4 * q;                         // + 16 t + r : k-step r, k = q <-> sample
        // c = a_unk (D*E)^T: with at most 4 unknown types the 4x4x4 (4 blocks) MFMA does it with every lane
        // useful (blocks = groups of 4 rows; A[b][i][k] sits at lane 16 k + 4 b + i, D[b][i][j] at lane
        // 16 i + 4 b + j, verified in tools/mfma_probe.hip): 18 cycles instead of 64 per k-step.
        constexpr bool kSmallC = NU <= 4;
        const int a2_row = kSmallC ? ((m16 & 3) < NU ? NCT + (m16 & 3) : ZROW) : (m16 < NU ? NCT + m16 : ZROW);
        int jp_row[NMT], lp_row[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            const int p = mt * 16 + m16;
            int l = 0;
            while ((l + 1) * (l + 2) / 2 <= p) ++l;
            jp_row[mt] = p < NP ? NCT + p - l * (l + 1) / 2 : ZROW;
            lp_row[mt] = p < NP ? NCT + l : ZROW;
        }
        // global -> register staging geometry: load i covers rows 2i, 2i+1; lane -> (row half, 2 samples)
        const int ld_row = lane >> 5;
        const int ld_col = (lane & 31) * 2;
        int ld_gcol = wcol0 + ld_col;
        if (ld_gcol > S - 2) ld_gcol = S - 2;  // ragged last column group: clamped samples meet zero operands
        v2d pv[8], pd[8];  // D stays f64 in flight: converting here would wait for the load right away
        double nrt[NKC > 0 ? NKC : 1];
        // (small loads first: vmcnt retires in order, so nothing issued after the 16 tile loads may be
        // waited on before the next step)
        // N is a multiple of 16 here (the launcher hands the ragged tail to the unfused kernels), so every
        // address is a wave-uniform 64-bit base plus a per-lane 32-bit offset: no clamps, no 64-bit multiplies.
        const int lane_tile_off = ld_row * S + ld_gcol;
        const int lane_rt_off = m16 * NCT + q;
        auto prefetch = [&](int64_t blk) {
            const int64_t r0 = blk * 16;
            const double* __restrict__ rb = Rtp + r0 * NCT;
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) nrt[kc] = rb[lane_rt_off + kc * 4];
            const double* __restrict__ vb = V + r0 * S;
            const double* __restrict__ db = D + r0 * S;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                pv[i] = *reinterpret_cast<const v2d*>(vb + (2 * i) * S + lane_tile_off);
                pd[i] = *reinterpret_cast<const v2d*>(db + (2 * i) * S + lane_tile_off);
            }
        };
        prefetch(blockIdx.x);
        double u2_acc = 0.0;
        DMF_STAMP_DECL

        for (int s = 0; s <= nk; ++s) {
            if (s < nk) {
                const int64_t blk = blockIdx.x + (int64_t)s * gridDim.x;
                const int64_t row0 = blk * 16;
                char* __restrict__ tile = tile_of(s & 1);
                double* __restrict__ tileV = reinterpret_cast<double*>(tile);
                float* __restrict__ tileD = reinterpret_cast<float*>(tile + kTileVBytes);
                // the A team's path to barrier X (then phase B) is the workgroup's critical path; the C waves
                // that share this SIMD's FP64 pipe fill whatever issue slots are left
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    *reinterpret_cast<v2d*>(tileV + (2 * i + ld_row) * kTileRowDoubles + ld_col) = pv[i];
                    *reinterpret_cast<v2f*>(tileD + (2 * i + ld_row) * kTileRowFloats + ld_col) =
                        v2f{(float)pd[i].x, (float)pd[i].y};
                }
                DMF_STAMP(5)  // tile store (waits for the prefetched loads)
                double rtop[NKC > 0 ? NKC : 1];
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) {
                    rtop[kc] = nrt[kc];  // B operand of the first product: Rt^T[k = 4 kc + q][n = row]
                    if (wave == 0) rtbuf[((s & 1) * 16 + m16) * NCTL + kc * 4 + q] = rtop[kc];
                }
                // the wave that will run this block's inner iterations fetches its u / u_ now (first pass)
                constexpr int RPW = 64 / NU;
                const int rl = lane / NU, j = lane - rl * NU;
                const bool my_turn = wave == s % NW;
                // (unconditional on purpose: loads under a branch make the compiler's s_waitcnt placement
                // pessimistic at the join and stall on the fresh prefetch)
                const bool ok0 = rl < RPW && rl < 16;
                const int64_t gi0 = ok0 ? (row0 + rl) * NU + j : 0;
                const double uu0 = u[gi0];
                const double up0 = u_prev[gi0];
                __builtin_amdgcn_wave_barrier();
                DMF_STAMP(0)  // tile store

                // ---- phase A: MFMA contractions on the tile, row-on-lane layout.  The LDS reads of strip
                // t + 1 (tile values and alpha operands) are issued before the MFMAs of strip t, so that one
                // A wave per SIMD does not expose an LDS round trip in front of every MFMA group.
                struct Strip {
                    v2d v01, v23;
                    v4f df;
                    double a1v[NKC > 0 ? NKC : 1];
                    v2d a2lo, a2hi;
                    v2d jlo[NMT], jhi[NMT], llo[NMT], lhi[NMT];
                };
                auto load_strip = [&](int t, Strip& R) {
                    const double* __restrict__ tv = tileV + m16 * kTileRowDoubles + t * 16 + 4 * q;
                    R.v01 = *reinterpret_cast<const v2d*>(tv);
                    R.v23 = *reinterpret_cast<const v2d*>(tv + 2);
                    R.df = *reinterpret_cast<const v4f*>(tileD + m16 * kTileRowFloats + t * 16 + 4 * q);
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) R.a1v[kc] = alds[(kc * 4 + q) * AS + e_col + 16 * t];
                    // operand rows: this lane's four samples are contiguous and 16-B aligned in alds
                    const int col = k_col + 16 * t;
                    R.a2lo = *reinterpret_cast<const v2d*>(alds + a2_row * AS + col);
                    R.a2hi = *reinterpret_cast<const v2d*>(alds + a2_row * AS + col + 2);
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt) {
                        R.jlo[mt] = *reinterpret_cast<const v2d*>(alds + jp_row[mt] * AS + col);
                        R.jhi[mt] = *reinterpret_cast<const v2d*>(alds + jp_row[mt] * AS + col + 2);
                        R.llo[mt] = *reinterpret_cast<const v2d*>(alds + lp_row[mt] * AS + col);
                        R.lhi[mt] = *reinterpret_cast<const v2d*>(alds + lp_row[mt] * AS + col + 2);
                    }
                };
                // Two accumulator sets (even / odd k-steps): a dependent FP64 MFMA issued right behind its
                // producer stalls for most of the producer's latency, four-plus independent chains do not.
                v4d cacc = {0.0, 0.0, 0.0, 0.0}, cacc1 = cacc;
                double csm0 = 0.0, csm1 = 0.0;  // kSmallC: c[unknown q][row m16], one double per lane
                v4d macc[NMT], macc1[NMT];
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) macc[mt] = macc1[mt] = cacc;
                // first product of a strip: E = V - Rt a_known (a chain of NKC dependent MFMAs)
                auto e_init = [&](const Strip& R) { return v4d{R.v01.x, R.v01.y, R.v23.x, R.v23.y}; };
                // c / M products of strip R (8 MFMAs on two accumulators) with the E chain of the NEXT strip
                // slotted between them, so that no MFMA waits on the result of the one issued just before it
                auto run_strip = [&](const Strip& R, v4d e, const Strip& Rn, bool has_next) {
                    const v4d d = {(double)R.df.x, (double)R.df.y, (double)R.df.z, (double)R.df.w};
                    const v4d w = d * e;
                    const double a2v[4] = {R.a2lo.x, R.a2lo.y, R.a2hi.x, R.a2hi.y};
                    v4d en = e_init(Rn);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (kSmallC) {
                            if (r & 1) csm1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2v[r], w[r], csm1, 0, 0, 0);
                            else csm0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2v[r], w[r], csm0, 0, 0, 0);
                        } else {
                            if (r & 1) cacc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2v[r], w[r], cacc1, 0, 0, 0);
                            else cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2v[r], w[r], cacc, 0, 0, 0);
                        }
#pragma unroll
                        for (int mt = 0; mt < NMT; ++mt) {
                            const double pj = r == 0 ? R.jlo[mt].x : r == 1 ? R.jlo[mt].y : r == 2 ? R.jhi[mt].x : R.jhi[mt].y;
                            const double pl = r == 0 ? R.llo[mt].x : r == 1 ? R.llo[mt].y : r == 2 ? R.lhi[mt].x : R.lhi[mt].y;
                            if (r & 1) macc1[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj * pl, d[r], macc1[mt], 0, 0, 0);
                            else macc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj * pl, d[r], macc[mt], 0, 0, 0);
                        }
                        if (has_next && r < NKC) en = __builtin_amdgcn_mfma_f64_16x16x4f64(Rn.a1v[r], rtop[r], en, 0, 0, 0);
                    }
                    if (has_next) {
#pragma unroll
                        for (int kc = 4; kc < NKC; ++kc)  // NKC <= 4: never runs; kept for clarity
                            en = __builtin_amdgcn_mfma_f64_16x16x4f64(Rn.a1v[kc], rtop[kc], en, 0, 0, 0);
                    }
                    return en;
                };
                Strip sa, sb;
                load_strip(0, sa);
                load_strip(1, sb);
                __builtin_amdgcn_sched_barrier(0);
                v4d e0 = e_init(sa);
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa.a1v[kc], rtop[kc], e0, 0, 0, 0);
                const v4d e1 = run_strip(sa, e0, sb, true);
                load_strip(2, sa);
                __builtin_amdgcn_sched_barrier(0);
                const v4d e2 = run_strip(sb, e1, sa, true);
                load_strip(3, sb);
                __builtin_amdgcn_sched_barrier(0);
                const v4d e3 = run_strip(sa, e2, sb, true);
                (void)run_strip(sb, e3, sb, false);
                cacc += cacc1;
                if constexpr (kSmallC) cacc[0] = csm0 + csm1;  // same place as register 0 of the 16x16 tile
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) macc[mt] += macc1[mt];
                // the next block's global loads go out only now: their 64 staging VGPRs are dead during
                // phase A (that is what pays for the second strip register set) and the loads still have
                // all of phase B to land
                prefetch(s + 1 < nk ? blk + gridDim.x : blk);
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
                __builtin_amdgcn_s_setprio(0);
                DMF_STAMP(1)  // phase A
                __syncthreads();  // ---- barrier X
                DMF_STAMP(2)  // wait at X

                // ---- phase B: row-local inner iterations, lane = (row, unknown j); c and M pre-scaled by 1/l_w
                if (my_turn) {
                    // the other waves of this SIMD keep the FP64 pipe busy with independent work; this
                    // dependent chain sits on the workgroup's critical path, so it goes first
                    __builtin_amdgcn_s_setprio(3);
                    const int lane0 = lane - j;
                    double* __restrict__ ub = ubuf + (s & 1) * 16 * NU;
                    double* __restrict__ upb = upbuf + (s & 1) * 16 * NU;
                    for (int pass0 = 0; pass0 < 16; pass0 += RPW) {
                        const int rloc = pass0 + rl;
                        const bool ok = rl < RPW && rloc < 16;
                        const int rlc = rloc < 16 ? rloc : 15;
                        double cj = 0.0, Ms[NU];
#pragma unroll
                        for (int l = 0; l < NU; ++l) Ms[l] = 0.0;
                        for (int w = 0; w < NW; ++w) {
                            const double* __restrict__ part = red + (size_t)w * NV * 16;
                            cj += part[j * 16 + rlc];
#pragma unroll
                            for (int l = 0; l < NU; ++l) {
                                const int p = l <= j ? tri(l, j) : tri(j, l);
                                Ms[l] += part[(NU + p) * 16 + rlc];
                            }
                        }
                        cj *= inv_lw;
#pragma unroll
                        for (int l = 0; l < NU; ++l) Ms[l] *= -inv_lw;  // the chain below adds -M x
                        const int64_t gi = ok ? (row0 + rloc) * NU + j : 0;
                        double uu = pass0 == 0 ? uu0 : u[gi];
                        double up = pass0 == 0 ? up0 : u_prev[gi];
                        // The momentum coefficients ride in a VGPR (lane t holds beta_t) and reach the loop
                        // through v_readlane: an LDS read here would sit on the dependent chain every step.
                        for (int t0 = 0; t0 < n_iter2; t0 += 64) {
                            const int tl = t0 + lane < n_iter2 ? t0 + lane : n_iter2 - 1;
                            const double bvec = beta_tab[tl];
                            const int b_lo = __double2loint(bvec), b_hi = __double2hiint(bvec);
                            const int t_end = n_iter2 - t0 < 64 ? n_iter2 - t0 : 64;
                            if (mode == 1) {  // deconvolution.py:163: gradient at the previous iterate
#pragma unroll 2  // (u, u_) swap roles every step: unrolled by two, the copies between them disappear
                                for (int t2 = 0; t2 < t_end; ++t2) {
                                    const double beta = __hiloint2double(__builtin_amdgcn_readlane(b_hi, t2),
                                                                         __builtin_amdgcn_readlane(b_lo, t2));
                                    const double ut = fma(beta, uu - up, uu);
                                    up = uu;
                                    uu = f_step_chain<NU>(ut + cj, up, Ms, lane0);
                                }
                            } else {          // deconvolution.py:88: gradient at the extrapolated point
#pragma unroll 2  // (u, u_) swap roles every step: unrolled by two, the copies between them disappear
                                for (int t2 = 0; t2 < t_end; ++t2) {
                                    const double beta = __hiloint2double(__builtin_amdgcn_readlane(b_hi, t2),
                                                                         __builtin_amdgcn_readlane(b_lo, t2));
                                    const double ut = fma(beta, uu - up, uu);
                                    up = uu;
                                    uu = f_step_chain<NU>(ut + cj, ut, Ms, lane0);
                                }
                            }
                        }
                        if (ok) {
                            // to LDS only: a C wave stores the rows to HBM next step, so that no A wave ever
                            // has a store in flight when its next tile write waits on vmcnt
                            ub[rloc * NU + j] = uu;
                            upb[rloc * NU + j] = up;
                            u2_acc = fma(uu, uu, u2_acc);
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
                DMF_STAMP(3)  // phase B (or idle)
                __syncthreads();  // ---- barrier Y
                DMF_STAMP(4)  // wait at Y
            } else {
                __syncthreads();  // X: the C team is finishing the last block
                __syncthreads();  // Y
            }
        }
        // share of ||u||^2 (red is free again: the last step's phase B is behind barrier Y)
        const double w2 = wave_sum(u2_acc);
        if (lane == 0) red[wave] = w2;
        DMF_STAMP_FLUSH
    } else {
        // =========================== C team: phase C ==========================================
        const int half = cidx / NW;  // rows [8 half, 8 half + 8) of every block
        double acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = 0.0;
        const int sC = wcol0 + lane;  // lane = sample column

        DMF_STAMP_DECL
        auto accum_rows = [&](int buf, int r_begin, int n_rows) {
            const char* __restrict__ tile = tile_of(buf);
            const double* __restrict__ tileV = reinterpret_cast<const double*>(tile);
            const float* __restrict__ tileD = reinterpret_cast<const float*>(tile + kTileVBytes);
            const double* __restrict__ ub = ubuf + buf * 16 * NU;
            const double* __restrict__ rb = rtbuf + buf * 16 * NCTL;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                if (rr >= n_rows) break;
                const int r = r_begin + rr;
                const double d = (double)tileD[r * kTileRowFloats + lane];
                const double v = tileV[r * kTileRowDoubles + lane];
                double uj[NU], t[NU];
                if constexpr ((NU & 1) == 0) {
#pragma unroll
                    for (int jj = 0; jj < NU; jj += 2) {  // 16-B broadcast reads
                        const v2d two = *reinterpret_cast<const v2d*>(ub + r * NU + jj);
                        uj[jj] = two.x;
                        uj[jj + 1] = two.y;
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < NU; ++jj) uj[jj] = ub[r * NU + jj];
                }
#pragma unroll
                for (int jj = 0; jj < NU; ++jj) t[jj] = d * uj[jj];
#pragma unroll
                for (int k = 0; k < NCT; k += 2) {
                    const v2d rk2 = *reinterpret_cast<const v2d*>(rb + r * NCTL + k);
#pragma unroll
                    for (int jj = 0; jj < NU; ++jj) {
                        acc[k * NU + jj] = fma(rk2.x, t[jj], acc[k * NU + jj]);
                        acc[(k + 1) * NU + jj] = fma(rk2.y, t[jj], acc[(k + 1) * NU + jj]);
                    }
                }
#pragma unroll
                for (int l = 0; l < NU; ++l)
#pragma unroll
                    for (int jj = 0; jj <= l; ++jj)
                        acc[NCT * NU + tri(jj, l)] = fma(t[jj], uj[l], acc[NCT * NU + tri(jj, l)]);
#pragma unroll
                for (int jj = 0; jj < NU; ++jj)
                    acc[NCT * NU + NP + jj] = fma(t[jj], v, acc[NCT * NU + NP + jj]);
                DMF_ROW_STAMP(rr)
            }
        };

        for (int s = 0; s <= nk; ++s) {
            int buf = 0;
            if (s >= 1) {
                const int64_t row0 = (blockIdx.x + (int64_t)(s - 1) * gridDim.x) * 16;
                buf = (s - 1) & 1;
                if (cidx == 0) {  // the block's rows are contiguous in u: coalesced stores
                    for (int e = lane; e < 16 * NU; e += 64) {
                        u[row0 * NU + e] = ubuf[buf * 16 * NU + e];
                        u_prev[row0 * NU + e] = upbuf[buf * 16 * NU + e];
                    }
                }
                if (kCRowsBeforeX > 0) accum_rows(buf, 8 * half, kCRowsBeforeX);
            }
            DMF_STAMP(0)  // rows before X
            __syncthreads();  // ---- barrier X
            DMF_STAMP(2)  // wait at X
            if (s >= 1) accum_rows(buf, 8 * half + kCRowsBeforeX, 8 - kCRowsBeforeX);
            DMF_STAMP(1)  // last 4 rows
            __syncthreads();  // ---- barrier Y
            DMF_STAMP(4)  // wait at Y
        }
        DMF_STAMP_FLUSH

        // ---- slab of this wave's half of the workgroup (job order of the solver's table)
        if (sC < S) {
            const int n_jobs = n_c * NU + NP + NU;
            double* __restrict__ out = slab + ((int64_t)blockIdx.x * 2 + half) * n_jobs * S + sC;
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

size_t fused_lds_bytes(int S, int nct, int n_u, int n_iter2) {
    const int NW = (S + 63) / 64;
    const int nv = n_u + n_u * (n_u + 1) / 2;
    const size_t doubles = (size_t)((n_iter2 + 1) & ~1) + 4 * 16 * n_u + 2 * 16 * (nct > 0 ? nct : 1) +
                           (size_t)NW * nv * 16 + (size_t)(nct + n_u + 1) * (NW * 64 + 2);
    return doubles * sizeof(double) + (size_t)2 * NW * kTileBytes;
}

hipError_t launch_finish_u_norm(const double* u2_partials, int n, SolverState* state, hipStream_t st) {
    hipLaunchKernelGGL(k_finish_u_norm, dim3(1), dim3(256), 0, st, u2_partials, n, state);
    return hipGetLastError();
}

bool rowpass_fused_supported(int S, int n_c, int n_u) {
    // n_u <= 4: beyond that phase B needs two passes per block and the accumulators spill; measured at
    // 5e5 x 128, n_c = 0: n_u = 5 / 6 / 8 run 0.96 / 1.42 / 2.1 ms fused against 0.62 / 0.68 / 1.18 ms unfused
    if ((S & 3) != 0 || S > 256 || n_c > 16 || n_u < 1 || n_u > 4) return false;
    const int nct = (n_c + 3) / 4 * 4;
    return nct * n_u + n_u * (n_u + 1) / 2 + n_u <= 80;
}

int rowpass_fused_grid(int64_t N, int S) {
    const int NW = (S + 63) / 64;
    const int per_cu = NW <= 2 ? 2 : 1;  // LDS: two tile buffers per column group
    const int64_t nblk = N / 16;
    const int64_t g = 256 * per_cu;
    return (int)(nblk < g ? nblk : g);
}

int64_t rowpass_fused_slab_doubles(int64_t N, int S, int n_c, int n_u) {
    return (int64_t)2 * rowpass_fused_grid(N, S) * (n_c * n_u + n_u * (n_u + 1) / 2 + n_u) * S;
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
        if (lds > 160 * 1024 || (N & 15) != 0 || N < 16) return hipErrorInvalidValue;
        // raise the dynamic-LDS limit once per (instantiation, device): this launch sits in the per-iteration loop
        static bool lds_limit_raised[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (lds > 48 * 1024 && !lds_limit_raised[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)k_rowpass_fused<NKC, NU>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            lds_limit_raised[dev] = true;
        }
        const int grid = rowpass_fused_grid(N, S);
        *grid_out = grid;
        hipLaunchKernelGGL((k_rowpass_fused<NKC, NU>), dim3(grid), dim3(DMF_WAVES_PER_WG(NW) * 64), lds, st, V, D, Rtp, alpha, u,
                           u_prev, state, N, S, n_c, n_iter2, mode, slab, u2_partials
#ifdef DMF_STAMPS
                           , (unsigned long long*)nullptr
#endif
                           );
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
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4)
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
