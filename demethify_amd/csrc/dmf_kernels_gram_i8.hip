// u-dependent Gram entries of the alpha phase on the INTEGER matrix cores, exactly.
//
// The alpha phase (deconvolution.py:93-102) needs, per sample s, G_s = R^T diag(d_s) R and b_s = R^T (d_s * v_s).
// The entries that involve the unknown profiles u and no V are sums over the CpG rows of a per-row feature times a
// count:   cross[k][j][s] = sum_i (Rt_ik u_ij) d_is      uu[j<=l][s] = sum_i (u_ij u_il) d_is
// i.e. one GEMM  G[p][s] = sum_i Z[i][p] D[i][s]  with Z (N x NF) formed on the fly and D the counts.  D is an
// exact small integer and Z lies in [0, 1], so the GEMM is evaluated WITHOUT rounding: Z is written in fixed point,
// z ~ rint(z 2^52) = sum_t a_t 256^t with seven balanced 8-bit digits a_t in [-128, 127], the counts likewise as
// one (d <= 127) or two (d <= 32639) balanced digits, every digit product is accumulated exactly in i32 by
// v_mfma_i32_32x32x32_i8 and the per-workgroup sums are combined in 64-bit integer arithmetic (k_gram_v2_reduce): one
// conversion to f64 at the very end.  The only rounding against exact arithmetic is rint(z 2^52) of the exact product
// z = Rt_ik u_ij: |error| <= 2^-53 per feature value, an ulp of 1.0 -- far below a single rounding of an f64
// accumulation of the same sum (2^-53 RELATIVE to a running sum of ~N d z / 2).  (FP64 needs one FMA per
// (row, feature, sample); here the same product costs 7 i8 MACs at 64x the FP64 rate.)
//
// Layouts.  B operand = counts as 8-bit digit planes Dt8[plane][row block of 32][sample block of 32][n 32][k 32]
// (built once per problem, k_build_dt8): lane (n = l & 31, h = l >> 5) loads its 16 bytes k = 16 h .. 16 h + 15 with
// one 16-B request, the wave 1 KB contiguous.  A operand = digit t of feature p for the block's 32 rows, generated
// by the workgroup into LDS as Atile[h][m][16 B] (m = 64 t + p): conflict-free ds_read_b128 per (m, h).
// Any bijection (h, byte) -> k the hardware applies is the same for A and B, so only "16 bytes per lane = 16
// consecutive rows" matters; the C layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) is the
// dtype-independent 32x32 map.  A workgroup = 8 waves = 4 sample blocks (128 samples) x 2 feature halves x one row
// range, accumulators in registers over the whole range (7 or 8 tiles of 16 VGPRs per wave, two waves per SIMD).
#include <cstdint>
#include <cstdlib>

#include "dmf_device.h"
#include "dmf_internal.h"
#include "dmf_fixedpoint.h"

namespace dmf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

namespace {
constexpr int kNSL = 7;       // balanced base-256 digits of rint(z 2^52)
constexpr int kRing = 8;      // LDS-DMA ring: block slots
constexpr int kMaxFeat = 576;  // (64 per launch; 32 unknowns without known types: 528)
}  // namespace

// ------------------------------------------------------------------------------------------------ builders
// counts (f64, integral, 0 <= d <= 65535) -> D16[N16][SD], zero padded (rows to a multiple of 16, columns to SD)
__global__ __launch_bounds__(256) void k_build_d16(const double* __restrict__ D, int64_t N, int S,
                                                   unsigned short* __restrict__ D16, int64_t N16, int SD) {
    const int64_t chunks = N16 * (SD / 8);
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * 256) {
        const int64_t row = c / (SD / 8);
        const int col0 = (int)(c - row * (SD / 8)) * 8;
        unsigned int w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned int lo = 0, hi = 0;
            if (row < N) {
                if (col0 + 2 * e < S) lo = (unsigned int)D[row * S + col0 + 2 * e];
                if (col0 + 2 * e + 1 < S) hi = (unsigned int)D[row * S + col0 + 2 * e + 1];
            }
            w[e] = lo | (hi << 16);
        }
        *reinterpret_cast<v4u*>(D16 + row * SD + col0) = v4u{w[0], w[1], w[2], w[3]};
    }
}

// D16 -> balanced 8-bit digit planes in the MFMA B layout.  One wave per (row block, sample block) tile.
// ND = 1: plane 0 = d (<= 127).  ND = 2: d + 128 = b0 + 256 b1, plane 0 = b0 - 128 (as i8: b0 ^ 0x80), plane 1 = b1
// (<= 127 for d <= 32639), so that d = plane0 + 256 plane1 exactly.
__global__ __launch_bounds__(256) void k_build_dt8(const unsigned short* __restrict__ D16, int64_t N, int SD, int ND,
                                                   signed char* __restrict__ Dt8, int64_t plane_stride, int64_t n_tiles) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 31, h = lane >> 5;
    const int SB = SD / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t rb = tile / SB;
        const int sb = (int)(tile - rb * SB);
        unsigned int p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t row = rb * 32 + 16 * h + j;
            unsigned int d = row < N ? D16[row * SD + sb * 32 + n] : 0u;
            unsigned int b0 = d, b1 = 0;
            if (ND == 2) {
                d += 128u;
                b0 = (d & 0xFFu) ^ 0x80u;
                b1 = d >> 8;
            }
            p0[j >> 2] |= (b0 & 0xFFu) << (8 * (j & 3));
            p1[j >> 2] |= (b1 & 0xFFu) << (8 * (j & 3));
        }
        signed char* __restrict__ dst = Dt8 + tile * 1024 + n * 32 + h * 16;
        *reinterpret_cast<v4u*>(dst) = v4u{p0[0], p0[1], p0[2], p0[3]};
        if (ND == 2) *reinterpret_cast<v4u*>(dst + plane_stride) = v4u{p1[0], p1[1], p1[2], p1[3]};
    }
}

hipError_t launch_build_counts_int(const double* D, int64_t N, int S, int ND, unsigned short* D16, int64_t N16, int SD,
                                   signed char* Dt8, int64_t plane_stride, hipStream_t st) {
    const int64_t chunks = N16 * (SD / 8);
    int64_t g = (chunks + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_build_d16, dim3((unsigned)g), dim3(256), 0, st, D, N, S, D16, N16, SD);
    if (Dt8 != nullptr) {
        const int64_t n_tiles = ((N + 31) / 32) * (SD / 32);
        int64_t gt = (n_tiles + 3) / 4;
        if (gt > 16384) gt = 16384;
        hipLaunchKernelGGL(k_build_dt8, dim3((unsigned)gt), dim3(256), 0, st, D16, N, SD, ND, Dt8, plane_stride, n_tiles);
    }
    return hipGetLastError();
}

// Row-resampled copy of the u16 counts (a bootstrap replicate, bootstrap.py:28): dst[r][:] = src[idx[r]][:] for r < n_idx,
// zero rows up to N16; the largest count of the copy goes to *max_out (atomicMax; the caller zeroes it) -- max(D) of the
// resampled counts is what the reference's d = max(D)^2 is taken from.  One wave per destination row.
__global__ __launch_bounds__(256) void k_gather_rows_u16(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                         const long long* __restrict__ idx, int64_t n_idx, int64_t N16, int SD,
                                                         unsigned int* __restrict__ max_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned int mx = 0u;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < N16; r += (int64_t)gridDim.x * 4) {
        const unsigned short* __restrict__ s = r < n_idx ? src + idx[r] * SD : nullptr;
        for (int c = lane * 8; c < SD; c += 512) {
            v4u w = v4u{0u, 0u, 0u, 0u};
            if (s != nullptr) w = *reinterpret_cast<const v4u*>(s + c);
            *reinterpret_cast<v4u*>(dst + r * SD + c) = w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned int lo = w[e] & 0xFFFFu, hi = w[e] >> 16;
                mx = mx > lo ? mx : lo;
                mx = mx > hi ? mx : hi;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)mx, off, 64);
        mx = mx > o ? mx : o;
    }
    // one atomic per workgroup (a quarter of a million waves on one address took 2.7 ms of a 0.3 ms copy)
    __shared__ unsigned int wmax[4];
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int m = wmax[0];
        for (int w = 1; w < 4; ++w) m = m > wmax[w] ? m : wmax[w];
        if (m > 0u) atomicMax(max_out, m);
    }
}

hipError_t launch_gather_counts_int(const unsigned short* src16, const long long* idx, int64_t n_idx, int SD, int ND,
                                    unsigned short* D16, int64_t N16, signed char* Dt8, int64_t plane_stride,
                                    unsigned int* max_out, hipStream_t st) {
    int64_t g = (N16 + 3) / 4;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipError_t e = hipMemsetAsync(max_out, 0, sizeof(unsigned int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gather_rows_u16, dim3((unsigned)g), dim3(256), 0, st, src16, D16, idx, n_idx, N16, SD, max_out);
    if (Dt8 != nullptr) {
        const int64_t n_tiles = ((n_idx + 31) / 32) * (SD / 32);
        int64_t gt = (n_tiles + 3) / 4;
        if (gt > 16384) gt = 16384;
        hipLaunchKernelGGL(k_build_dt8, dim3((unsigned)gt), dim3(256), 0, st, D16, n_idx, SD, ND, Dt8, plane_stride, n_tiles);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ the GEMM (eight waves)
typedef __attribute__((address_space(1))) const void gmem_void;
typedef __attribute__((address_space(3))) int lds_int;

// Diagnostic build only (tools/gram_i8_probe.hip defines DMF_STAMPS): per-wave cycle sums of the block loop's segments.
#ifdef DMF_STAMPS
#define DMFG_STAMP_DECL unsigned long long st_last = dmfg_stamp(), st_seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DMFG_STAMP(i) { const unsigned long long st_now = dmfg_stamp(); st_seg[i] += st_now - st_last; st_last = st_now; }
#define DMFG_STAMP_FLUSH if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) stamps_out[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 + i_] = st_seg[i_];
__device__ __forceinline__ unsigned long long dmfg_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#else
#define DMFG_STAMP_DECL
#define DMFG_STAMP(i)
#define DMFG_STAMP_FLUSH
#endif

// Per 32-row block the workgroup needs 1 KB of count digits per sample group and plane (B operand) and the block's rows
// of R_trunc and u.  Both arrive by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, requests stay in flight across
// barriers) into a ring of RING block slots; "landed" is a counted s_waitcnt vmcnt(n) followed by the block's one barrier
// (the compiler does not order LDS reads behind LDS-DMA: the wait + barrier pair does).
//
// Eight waves, two per SIMD.  The first version of this kernel had four waves with all 64 features' accumulators each
// (~480 registers, one wave per SIMD); in-kernel stamps showed its 2.0 k cycles per 32-row block to be one wave's
// serial instruction stream -- DMA issue 0.35 k, LDS round trips 0.5 k + 0.4 k, digit arithmetic 0.6 k, 14 MFMAs 0.45 k of
// matrix pipe -- and reordering that stream moved nothing: nobody to issue while the wave waits.  Here a wave keeps ONE
// feature half (32 features x 32 samples x 7 or 8 digit weights = 112 / 128 accumulator registers), wave = (sample group
// w & 3, feature half w >> 2), so that two waves share a SIMD and each other's LDS / DMA / barrier waits:
//   * DMA per block: waves 0..3 fetch the ND count planes of their sample group, waves 4..7 the XL pieces of the x image;
//   * digits: wave w converts 16 features x 16 rows (lane = feature l & 15, row quad l >> 4), one dword per digit;
//   * three A tiles: iteration b multiplies block b from operands read during iteration b - 1, converts block b + 2 and
//     reads the operands of block b + 1.
// It is used at every feature count (it beat the four-wave form, since removed, also at 10..15 features).
template <int XL, int ND, int RING>
__global__ __launch_bounds__(512) void k_gram_i8_w8(const signed char* __restrict__ Dt8, int64_t plane_stride, int SB,
                                                    const double* __restrict__ Rtp, int nct, const double* __restrict__ u,
                                                    int64_t N, int n_c, int n_u, const short* __restrict__ feat_a,
                                                    const short* __restrict__ feat_b, int NF, int p0, int MFtot,
                                                    int64_t rows_per_wg, long long* __restrict__ slab, int SDs,
                                                    const int* __restrict__ done_flag
#ifdef DMF_STAMPS
                                                    , unsigned long long* __restrict__ stamps_out
#endif
                                                    ) {
    constexpr int MF = 64, MA = kNSL * MF;  // feature slots per digit (lane = feature), rows of the A tile
    constexpr int NWT = kNSL + ND - 1;  // digit weights 256^0 .. 256^(NWT-1): count digit d + feature digit t -> t + d
    constexpr int NDMA = ND > XL ? ND : XL;  // DMA descriptors a lane may need (count waves ND, row waves XL)
    constexpr int kSlotB = 4 * ND * 1024, kSlotX = XL * 4096, kSlot = kSlotB + kSlotX;  // B: [sample group][plane][1 KB]
    constexpr int kAtile = 2 * MA * 4;  // dwords of one block's A tile [2 (h)][MA][4]
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char* __restrict__ ring = lds_raw;                                                            // [RING][kSlot]
    unsigned int* __restrict__ atile = reinterpret_cast<unsigned int*>(lds_raw + RING * kSlot);  // [3][2 (h)][MA][4]
    if (done_flag != nullptr && *done_flag) return;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int sg = wave & 3, fh = wave >> 2;
    // Workgroups of one row range (they fetch the same rows of R_trunc and u) onto ONE XCD, so that the second fetch
    // finds the rows in that XCD's L2: consecutive workgroup ids go round the 8 XCDs, hence range y = 8 k + xcd and
    // sample quarter x take id = 8 (k nsh + x) + xcd.  (1-D launch; any other grid shape keeps the natural order.)
    const int nsh = (SB + 3) / 4, n_wg = (int)gridDim.x, ny = n_wg / nsh;
    int bx, by;
    if (ny % 8 == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bx = slot % nsh;
        by = (slot / nsh) * 8 + xcd;
    } else {
        bx = blockIdx.x % nsh;
        by = blockIdx.x / nsh;
    }
    const int sb = bx * 4 + sg;  // this wave's block of 32 samples
    const bool wave_on = sb < SB;
    const int sbc = wave_on ? sb : SB - 1;
    const int64_t r_begin = (int64_t)by * rows_per_wg;  // multiple of 32
    int64_t r_end = r_begin + rows_per_wg;
    if (r_end > N) r_end = N;
    const int nb = r_end > r_begin ? (int)((r_end - r_begin + 31) / 32) : 0;
    if (nb == 0) return;

    for (int i = tid; i < 3 * kAtile; i += 512) atile[i] = 0u;

    // this lane's feature: byte offsets of its two factors inside a block's x image
    // [R_trunc rows: 32 x nct doubles (the padded copy)][u rows: 32 x n_u doubles]
    // digit conversion: wave w = features 16 (w & 3) .. + 15 x rows 16 (w >> 2) .. + 15, lane = (feature l & 15, row
    // quad l >> 4): a lane's four rows are one dword of the A tile, and a 32-lane group (16 features x 2 quads) writes
    // banks 4 f + q -- two-way, which a ds_write_b32 hides -- and reads rows 4 apart (conflict-free: 96-B rows)
    const int feat = 16 * (wave & 3) + (lane & 15), quad = lane >> 4, row_first = 16 * (wave >> 2) + 4 * quad;
    const bool feat_on = feat < NF;
    int offA = 0, strideA = 0, offB = 0, strideB = 0;
    if (feat_on) {
        const int ia = feat_a[p0 + feat], ib = feat_b[p0 + feat];
        offA = ia < n_c ? ia * 8 : 256 * nct + (ia - n_c) * 8;
        strideA = ia < n_c ? nct * 8 : n_u * 8;
        offB = ib < n_c ? ib * 8 : 256 * nct + (ib - n_c) * 8;
        strideB = ib < n_c ? nct * 8 : n_u * 8;
    }
    offA += row_first * strideA;
    offB += row_first * strideB;

    v16i acc[NWT];
#pragma unroll
    for (int a = 0; a < NWT; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0;

    // One DMA descriptor per lane and piece: source = base + min(block * step, limit).  Waves 0..3 fetch the ND count
    // planes of their sample group, waves 4..7 the XL pieces of the x image: 16-B chunk c of
    // [32 rows of the padded R_trunc copy][32 rows of u]; the limit keeps the last block's chunks inside the arrays
    // (rows of the padded copy are multiples of 32 B; u may end on an odd double, the solver rounds its allocation up).
    const char* gbase[NDMA];
    int64_t glim[NDMA];
    int gstep[NDMA], lds_dst[NDMA];
    const int n_dma = wave < 4 ? ND : XL;  // (wave-uniform)
#pragma unroll
    for (int x = 0; x < NDMA; ++x) {
        if (wave < 4) {  // count planes of this wave's sample group
            const int xp = x < ND ? x : ND - 1;
            gbase[x] = reinterpret_cast<const char*>(Dt8) + (int64_t)xp * plane_stride +
                       (((r_begin >> 5) * SB + sbc) * 1024 + (lane & 31) * 32 + (lane >> 5) * 16);
            gstep[x] = SB * 1024;
            glim[x] = (int64_t)1 << 62;
            lds_dst[x] = (sg * ND + xp) * 1024;
        } else {
            const int n_chunk_rt = 16 * nct, n_chunks = 16 * (nct + n_u);
            const int xx = x < XL ? x : XL - 1;
            int c = (tid - 256) + 256 * xx;
            if (c >= n_chunks) c = n_chunks - 1;  // lands in the slot's padding
            if (c < n_chunk_rt) {
                const int64_t off0 = r_begin * nct * 8 + (int64_t)c * 16;
                gbase[x] = reinterpret_cast<const char*>(Rtp) + off0;
                gstep[x] = 32 * nct * 8;
                glim[x] = N * nct * 8 - 16 - off0;
            } else {
                const int64_t off0 = r_begin * n_u * 8 + (int64_t)(c - n_chunk_rt) * 16;
                gbase[x] = reinterpret_cast<const char*>(u) + off0;
                gstep[x] = 32 * n_u * 8;
                glim[x] = ((N * n_u * 8 - 8) & ~(int64_t)15) - off0;
            }
            lds_dst[x] = kSlotB + (xx * 4 + (wave - 4)) * 1024;
        }
    }
    const bool tail_clamp = r_end == N && (N & 31) != 0;  // only the last block of the last row range can run past N
    auto issue = [&](int j) {
        const int jc = j < nb ? j : nb - 1;  // beyond the range: a repeat of the last block keeps the DMA count uniform
        char* __restrict__ slot = ring + (j % RING) * kSlot;
#pragma unroll
        for (int x = 0; x < NDMA; ++x) {
            if (x >= n_dma) break;
            int64_t off = (int64_t)jc * gstep[x];
            if (tail_clamp && jc == nb - 1) off = off < glim[x] ? off : glim[x];
            __builtin_amdgcn_global_load_lds((gmem_void*)(gbase[x] + off), (lds_int*)(slot + lds_dst[x]), 16, 0, 0);
        }
    };
    // rows of block j's two factors (this lane's feature, its four rows) into registers
    double xa[4], xv[4];
    auto read_rows = [&](int j) {
        const char* __restrict__ xb = ring + (j % RING) * kSlot + kSlotB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xa[r] = *reinterpret_cast<const double*>(xb + offA + r * strideA);  // (offset 0 without a feature)
            xv[r] = *reinterpret_cast<const double*>(xb + offB + r * strideB);
        }
    };
    // ... and from there, as seven digit dwords, into A tile `buf`
    unsigned int lo[4], hi[4], tl[4], th[4];
    // (lanes without a feature convert whatever row 0 / column 0 holds: their digits land in A-tile rows whose
    // accumulators are never stored -- no select)
    auto convert_row = [&](int r) { z_to_biased(xa[r], xv[r], lo[r], hi[r]); };
    auto store_digits = [&](int buf) {
        unsigned int* __restrict__ at = atile + buf * kAtile + (((wave >> 2) * MA + feat) << 2) + quad;
#pragma unroll
        for (int t = 0; t < 4; ++t) at[(t * MF) << 2] = tl[t] ^ 0x80808080u;
        at[(4 * MF) << 2] = th[0] ^ 0x80808080u;
        at[(5 * MF) << 2] = th[1] ^ 0x80808080u;
        at[(6 * MF) << 2] = th[2];
    };
    auto generate = [&](int buf) {  // (prologue: the whole conversion in one piece)
#pragma unroll
        for (int r = 0; r < 4; ++r) convert_row(r);
        transpose4(lo, tl);
        transpose4(hi, th);
        store_digits(buf);
    };
    v4i bq[ND], aop[kNSL];
    auto load_operands = [&](int j, int abuf) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
            bq[d] = *reinterpret_cast<const v4i*>(ring + (j % RING) * kSlot + (sg * ND + d) * 1024 + lane * 16);
        const unsigned int* __restrict__ at = atile + abuf * kAtile + (((lane >> 5) * MA + 32 * fh + (lane & 31)) << 2);
#pragma unroll
        for (int t = 0; t < kNSL; ++t) aop[t] = *reinterpret_cast<const v4i*>(at + ((t * MF) << 2));
    };

    // prologue: RING - 1 blocks in flight; digits of blocks 0 and 1; operands of block 0; rows of block 2
#pragma unroll 1
    for (int j = 0; j < RING - 1; ++j) issue(j);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_rows(0);
    generate(0);
    read_rows(nb > 1 ? 1 : 0);
    generate(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_operands(0, 0);
    read_rows(nb > 2 ? 2 : nb - 1);

    DMFG_STAMP_DECL
    int a_cur = 0;  // A tile of block b (b % 3 without the division)
    // The block loop, unrolled over the ring: slot numbers are then constants and every LDS address of the body is a
    // loop-invariant register plus an immediate (the issue slots of the SIMD, ~4 cycles per vector or LDS instruction
    // of either wave, are what bounds this kernel -- stamps: the matrix phase takes 2 x (instructions) x 4 cycles).
    // Reads for blocks beyond the range (rows of b + 3, operands of b + 1) go to whatever the slot holds: their
    // registers are never consumed.
    const char* __restrict__ rowA = ring + kSlotB + offA;
    const char* __restrict__ rowB = ring + kSlotB + offB;
    const char* __restrict__ bsrc = ring + sg * ND * 1024 + lane * 16;
#pragma unroll 1
    for (int b0 = 0; b0 < nb; b0 += RING) {
#pragma unroll
        for (int uu = 0; uu < RING; ++uu) {
            const int b = b0 + uu;
            if (b >= nb) break;
            const int a_next = a_cur == 2 ? 0 : a_cur + 1, a_gen = a_next == 2 ? 0 : a_next + 1;
            DMFG_STAMP(0)
            // own DMA of blocks <= b + 3 landed, own LDS traffic (digit writes, operand and row reads) done; then
            // everyone's (waves 0..3 have ND DMAs per block in flight, waves 4..7 XL)
            // (the DMA of block b + RING - 1 is issued further down, in the matrix phase: blocks b + 4 .. b + RING - 2
            // may still be in flight here)
            if (XL == ND || wave < 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 5) * ND) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 5) * XL) : "memory");
            DMFG_STAMP(1)
#ifndef DMF_GABL_BARRIER
            __builtin_amdgcn_s_barrier();
#endif
            DMFG_STAMP(2)
            // Block b's seven MFMAs (operands in registers since the last iteration), one per piece of the conversion of
            // block b + 2 (rows in registers likewise), in this order and no other.  Each register is refilled from LDS
            // as soon as its consumer has issued -- A operand t of block b + 1 behind MFMA t, row r of block b + 3
            // behind its conversion, the count tile behind the last MFMA -- so that the LDS array works during the
            // matrix phase and only the seven digit stores remain behind it.  No branches: a wave beyond the last
            // sample block multiplies a repeat tile into accumulators that are never stored.
            const unsigned int* __restrict__ at_next = atile + a_next * kAtile + (((lane >> 5) * MA + 32 * fh + (lane & 31)) << 2);
            const int row_off = ((uu + 3) % RING) * kSlot, op_off = ((uu + 1) % RING) * kSlot;  // (constants: uu is unrolled)
#pragma unroll
            for (int t = 0; t < kNSL; ++t) {
#pragma unroll
                for (int d = 0; d < ND; ++d) {
#ifndef DMF_GABL_MFMA
                    acc[t + d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aop[t], bq[d], acc[t + d], 0, 0, 0);
#endif
                    // (the conversion pieces sit behind the first MFMAs of the block: one per MFMA)
                    const int slot = t * ND + d;
#ifndef DMF_GABL_CONV
                    if (slot < 4) convert_row(slot);
                    else if (slot == 4) transpose4(lo, tl);
                    else if (slot == 5) transpose4(hi, th);
#endif
                    __builtin_amdgcn_sched_barrier(0);
#ifndef DMF_GABL_ROWS
                    if (slot < 4) {
                        xa[slot] = *reinterpret_cast<const double*>(rowA + slot * strideA + row_off);
                        xv[slot] = *reinterpret_cast<const double*>(rowB + slot * strideB + row_off);
                    }
#endif
#ifndef DMF_GABL_AOP
                    if (d == ND - 1) aop[t] = *reinterpret_cast<const v4i*>(at_next + ((t * MF) << 2));
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#ifndef DMF_GABL_BQ
#pragma unroll
            for (int d = 0; d < ND; ++d) bq[d] = *reinterpret_cast<const v4i*>(bsrc + d * 1024 + op_off);
#endif
            DMFG_STAMP(3)
            {   // DMA of block b + RING - 1 into the slot of block b - 1 (its count tile and rows went to registers long
                // ago), here rather than at the head of the iteration: while this wave waits for the load path to take
                // the request, the other wave of the SIMD has matrix work to issue
                const int j = b + RING - 1, jc = j < nb ? j : nb - 1;  // beyond the range: a repeat keeps the DMA count uniform
                char* __restrict__ slot = ring + ((uu + RING - 1) % RING) * kSlot;
#pragma unroll
                for (int x = 0; x < NDMA; ++x) {
                    if (x >= n_dma) break;
                    int64_t off = (int64_t)jc * gstep[x];
                    if (tail_clamp && jc == nb - 1) off = off < glim[x] ? off : glim[x];
#ifndef DMF_GABL_DMA
                    __builtin_amdgcn_global_load_lds((gmem_void*)(gbase[x] + off), (lds_int*)(slot + lds_dst[x]), 16, 0, 0);
#endif
                }
            }
            DMFG_STAMP(5)
#ifndef DMF_GABL_STORE
            store_digits(a_gen);
#endif
            a_cur = a_next;
            DMFG_STAMP(4)
        }
    }
    DMFG_STAMP_FLUSH
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (repeat) DMAs must not outlive the workgroup's LDS

    // slab[y][half][feature slot][sample] (i64), as the four-wave kernel writes it
    if (wave_on) {
        const int n = lane & 31, h = lane >> 5;
        long long* __restrict__ out = slab + ((int64_t)by * 2 * MFtot + p0) * SDs + sb * 32 + n;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
            const int p = 32 * fh + m;
            long long lo = 0, hi = 0;
#pragma unroll
            for (int wt = 0; wt < NWT; ++wt) {
                const long long v = (long long)acc[wt][e];
                if (wt < 4) lo += v << (8 * wt);
                else hi += v << (8 * (wt - 4));
            }
            if (p < NF) {
                out[(int64_t)p * SDs] = lo;
                out[(int64_t)(MFtot + p) * SDs] = hi;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ b_u alone
// b_u[j][s] = sum_i u_ij d_is v_is for the solver paths whose u phase is a kernel of its own (n_u 5..20: the integer
// GEMM above then takes all the V-free Gram entries and this stream kernel the one right-hand side that needs V).
// Lane = sample, a row's u values are wave-uniform scalar loads, eight rows of V / D16 in flight per wave; the four
// waves of a workgroup are summed in fixed order into one slab [NU][S] per workgroup.  HBM-bound: V (f64) + counts (u16).
template <int NU>
__global__ __launch_bounds__(256) void k_bu_cols(const double* __restrict__ V, const unsigned short* __restrict__ D16, int SD,
                                                 const double* __restrict__ u, int64_t N, int S, double* __restrict__ slab,
                                                 const int* __restrict__ done_flag) {
    constexpr int kRows = 8;
    __shared__ double red[3][NU][64];
    if (done_flag != nullptr && *done_flag) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int s = blockIdx.y * 64 + lane;
    const bool active = s < S;
    const int sc = active ? s : S - 1;
    double acc[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) acc[j] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i0 = (int64_t)blockIdx.x * 4 + wave; i0 < N; i0 += kRows * stride) {
        double t[kRows];
        int64_t row[kRows];
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const int64_t i = i0 + x * stride;
            row[x] = i < N ? i : N - 1;
            const double d = i < N ? (double)D16[row[x] * SD + sc] : 0.0;
            t[x] = d * V[row[x] * S + sc];
        }
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const double* __restrict__ u_row = u + row[x] * NU;
#pragma unroll
            for (int j = 0; j < NU; ++j) acc[j] = fma(t[x], u_row[j], acc[j]);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < NU; ++j) red[wave - 1][j][lane] = acc[j];
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
        for (int j = 0; j < NU; ++j)
            slab[((int64_t)blockIdx.x * NU + j) * S + s] = ((acc[j] + red[0][j][lane]) + red[1][j][lane]) + red[2][j][lane];
    }
}

// The same stream with TWO adjacent samples per lane (odd S: see `lone`): one 16-byte load of V and one 4-byte
// load of the counts per row and lane -- half the load instructions per byte (the 2-byte-per-lane count load of the form
// above is the worst shape for the load path).  Same per-element arithmetic; slab layout unchanged.
template <int NU>
__global__ __launch_bounds__(256) void k_bu_cols2(const double* __restrict__ V, const unsigned short* __restrict__ D16, int SD,
                                                  const double* __restrict__ u, int64_t N, int S, double* __restrict__ slab,
                                                  const int* __restrict__ done_flag, int with_vdv) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int kRows = 8;
    __shared__ double red[3][NU][2][64];
    if (done_flag != nullptr && *done_flag) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int s = blockIdx.y * 128 + 2 * lane;
    const bool active = s < S;
    // odd S: the row's lone last sample takes the upper half of the pair one element lower; its partner's count is zero
    // padding (as in k_cost_cols2)
    const bool lone = s == S - 1;
    const int sc = lone ? S - 2 : (active ? s : 0);
    const int sd = active ? s : 0;
    typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));
    double acc[NU][2];
#pragma unroll
    for (int j = 0; j < NU; ++j) acc[j][0] = acc[j][1] = 0.0;
    // with_vdv: one more slab row per workgroup, sum_i d v^2 of the same stream (the v^T D v entry of a problem's known
    // block: problem_finalize) -- two more FMAs per row pair on a kernel that waits for HBM
    double vdv0 = 0.0, vdv1 = 0.0;
    const int ncol = NU + (with_vdv ? 1 : 0);
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i0 = (int64_t)blockIdx.x * 4 + wave; i0 < N; i0 += kRows * stride) {
        double t0[kRows], t1[kRows];
        int64_t row[kRows];
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const int64_t i = i0 + x * stride;
            row[x] = i < N ? i : N - 1;
            const unsigned int dd = i < N ? *reinterpret_cast<const unsigned int*>(D16 + row[x] * SD + sd) : 0u;
            const v2d_u v = *reinterpret_cast<const v2d_u*>(V + row[x] * S + sc);
            t0[x] = (double)(dd & 0xFFFFu) * (lone ? v.y : v.x);
            t1[x] = (double)(dd >> 16) * v.y;
            vdv0 = fma(t0[x], lone ? v.y : v.x, vdv0);
            vdv1 = fma(t1[x], v.y, vdv1);
        }
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const double* __restrict__ u_row = u + row[x] * NU;
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const double uj = u_row[j];
                acc[j][0] = fma(t0[x], uj, acc[j][0]);
                acc[j][1] = fma(t1[x], uj, acc[j][1]);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            red[wave - 1][j][0][lane] = acc[j][0];
            red[wave - 1][j][1][lane] = acc[j][1];
        }
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            double* __restrict__ out = slab + ((int64_t)blockIdx.x * ncol + j) * S + s;
            out[0] = ((acc[j][0] + red[0][j][0][lane]) + red[1][j][0][lane]) + red[2][j][0][lane];
            if (!lone) out[1] = ((acc[j][1] + red[0][j][1][lane]) + red[1][j][1][lane]) + red[2][j][1][lane];
        }
    }
    if (with_vdv) {  // (wave-uniform; the cross-wave buffer is free again behind a second barrier)
        __syncthreads();
        if (wave > 0) {
            red[wave - 1][0][0][lane] = vdv0;
            red[wave - 1][0][1][lane] = vdv1;
        }
        __syncthreads();
        if (wave == 0 && active) {
            double* __restrict__ out = slab + ((int64_t)blockIdx.x * ncol + NU) * S + s;
            out[0] = ((vdv0 + red[0][0][0][lane]) + red[1][0][0][lane]) + red[2][0][0][lane];
            if (!lone) out[1] = ((vdv1 + red[0][0][1][lane]) + red[1][0][1][lane]) + red[2][0][1][lane];
        }
    }
}

int bu_cols_grid(int64_t N) {
    int64_t want = (N + 4 * 8 - 1) / (4 * 8);
    if (want > 512) want = 512;
    return (int)(want < 1 ? 1 : want);
}

// with_vdv (may be null): in: the caller would like sum_i d v^2 as one more slab row per workgroup (slab rows n_u + 1);
// out: whether the kernel chosen for this shape delivers it.
hipError_t launch_bu_cols(const double* V, const unsigned short* D16, int SD, const double* u, int64_t N, int S, int n_u,
                          double* slab, const int* done_flag, int* n_slabs_out, hipStream_t st, bool* with_vdv) {
    const int nbx = bu_cols_grid(N);
    *n_slabs_out = nbx;
    const dim3 block(256);
    const bool two = (SD & 1) == 0 && S >= 128 && n_u <= 16 && (reinterpret_cast<uintptr_t>(V) & 7) == 0;
    const int vdv = (with_vdv != nullptr && *with_vdv && two) ? 1 : 0;
    if (with_vdv != nullptr) *with_vdv = vdv != 0;
    if (two) {
        const dim3 grid2(nbx, (S + 127) / 128);
        switch (n_u) {  // (LDS for the cross-wave sum: 3 x NU x 2 x 64 doubles = 48 KB at sixteen unknowns)
#define DMF_CASE2(NU_) \
    case NU_: hipLaunchKernelGGL((k_bu_cols2<NU_>), grid2, block, 0, st, V, D16, SD, u, N, S, slab, done_flag, vdv); return hipGetLastError();
            DMF_CASE2(1) DMF_CASE2(2) DMF_CASE2(3) DMF_CASE2(4) DMF_CASE2(5) DMF_CASE2(6) DMF_CASE2(7) DMF_CASE2(8) DMF_CASE2(9) DMF_CASE2(10)
            DMF_CASE2(11) DMF_CASE2(12) DMF_CASE2(13) DMF_CASE2(14) DMF_CASE2(15) DMF_CASE2(16)
#undef DMF_CASE2
            default: break;
        }
    }
    const dim3 grid(nbx, (S + 63) / 64);
    switch (n_u) {
#define DMF_CASE(NU_) \
    case NU_: hipLaunchKernelGGL((k_bu_cols<NU_>), grid, block, 0, st, V, D16, SD, u, N, S, slab, done_flag); break;
        DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4) DMF_CASE(5) DMF_CASE(6) DMF_CASE(7) DMF_CASE(8) DMF_CASE(9) DMF_CASE(10)
        DMF_CASE(11) DMF_CASE(12) DMF_CASE(13) DMF_CASE(14) DMF_CASE(15) DMF_CASE(16) DMF_CASE(17) DMF_CASE(18) DMF_CASE(19) DMF_CASE(20)
        DMF_CASE(21) DMF_CASE(22) DMF_CASE(23) DMF_CASE(24) DMF_CASE(25) DMF_CASE(26) DMF_CASE(27) DMF_CASE(28) DMF_CASE(29) DMF_CASE(30)
        DMF_CASE(31) DMF_CASE(32)
#undef DMF_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ reduce
constexpr int kRedChunks = 16;

// Grid (sample blocks, jobs, kRedChunks).  Jobs < n_feat add their chunk of the i64 slabs into acc64 with 64-bit integer
// atomics (exact, so the order does not matter); the b_u jobs store their chunk's sum (slab order) into bu_part;
// k_gram_v2_finish turns the complete sums into the solver's packed Gram rows.
// Loads are issued four slabs ahead of their use (the loop is a chain of HBM round trips otherwise).
__global__ __launch_bounds__(256) void k_gram_v2_reduce(const long long* __restrict__ slab_i8, int ny, int MF, int SDs,
                                                        const double* __restrict__ slab_bu, int n_bu_slabs, int n_u,
                                                        int n_feat, int S, unsigned long long* __restrict__ acc64,
                                                        double* __restrict__ bu_part,
                                                        const int* __restrict__ done_flag,
                                                        const double* __restrict__ u2_partials, int n_u2,
                                                        SolverState* __restrict__ state) {
    __shared__ long long part[3][2][64];
    __shared__ double partd[3][64];
    if (done_flag != nullptr && *done_flag) return;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + lane;
    const int job = blockIdx.y, chunk = blockIdx.z;
    const bool active = s < S;
    const int sc = active ? s : S - 1;
    if (u2_partials != nullptr && blockIdx.x == 0 && job == 0 && chunk == 0 && grp == 3) {
        // the row pass's per-workgroup shares of ||u||_F^2 -> state->u_norm2 and l_h (deconvolution.py:212), summed in
        // workgroup order (needs nothing from this launch: any one wave can do it)
        double a = 0.0;
        for (int i = lane; i < n_u2; i += 64) a += u2_partials[i];
        a = wave_sum(a);
        if (lane == 0) {
            state->u_norm2 = a;
            state->l_h = (state->rt_norm2 + a) * state->dsq;
        }
    }
    if (job < n_feat) {
        const int per = (ny + kRedChunks - 1) / kRedChunks;
        const int y0 = chunk * per, y1 = y0 + per < ny ? y0 + per : ny;
        const int64_t hi_off = (int64_t)MF * SDs, y_stride = (int64_t)2 * MF * SDs;
        const long long* __restrict__ base = slab_i8 + (int64_t)job * SDs + sc;
        long long lo = 0, hi = 0;
        for (int y = y0 + grp; y < y1; y += 16) {
            long long l[4], h[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = y + 4 * q;
                const long long* __restrict__ b = base + (int64_t)(yy < y1 ? yy : y) * y_stride;
                l[q] = b[0];
                h[q] = b[hi_off];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (y + 4 * q < y1) {
                    lo += l[q];
                    hi += h[q];
                }
        }
        if (grp > 0) {
            part[grp - 1][0][lane] = lo;
            part[grp - 1][1][lane] = hi;
        }
        __syncthreads();
        if (grp == 0 && active && y0 < y1) {
            lo += part[0][0][lane] + part[1][0][lane] + part[2][0][lane];
            hi += part[0][1][lane] + part[1][1][lane] + part[2][1][lane];
            atomicAdd(acc64 + (int64_t)job * S + s, (unsigned long long)lo);
            atomicAdd(acc64 + ((int64_t)n_feat + job) * S + s, (unsigned long long)hi);
        }
    } else {
        const int j = job - n_feat;
        const int per = (n_bu_slabs + kRedChunks - 1) / kRedChunks;  // slabs of this chunk, split over the 4 waves
        const int c0 = chunk * per, c1 = c0 + per < n_bu_slabs ? c0 + per : n_bu_slabs;
        const int perw = (per + 3) / 4;
        const int g0 = c0 + grp * perw, g1 = g0 + perw < c1 ? g0 + perw : c1;
        const double* __restrict__ base = slab_bu + (int64_t)j * S + sc;
        const int64_t g_stride = (int64_t)n_u * S;
        double acc = 0.0;
        for (int g = g0; g < g1; g += 4) {  // slab order: reproducible
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = base[(int64_t)(g + q < g1 ? g + q : g) * g_stride];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (g + q < g1) acc += v[q];
        }
        if (grp > 0) partd[grp - 1][lane] = acc;
        __syncthreads();
        if (grp == 0 && active) bu_part[((int64_t)chunk * n_u + j) * S + s] = ((acc + partd[0][lane]) + partd[1][lane]) + partd[2][lane];
    }
}

// The finishing pass: the complete sums -> the solver's packed Gram rows, and the scratch cleared for the next outer
// iteration.  A launch of its own: the kernel boundary is what orders every workgroup's contributions before these reads.
// (Round 2 let the last workgroup of a column to arrive do this inside k_gram_v2_reduce, on relaxed atomics -- 3 us
// cheaper per outer iteration, and correct on gfx950 in every test, but resting on "a returned atomic has been
// performed", which the memory model does not promise.  Making the arrival an agent-scope acquire-release operation
// instead costs +85 us per launch at the headline shape: the release writes back an L2 that the integer Gram kernel has
// just filled with 34 MB of slabs.)
__global__ __launch_bounds__(64) void k_gram_v2_finish(int n_u, int n_feat, int S, unsigned long long* __restrict__ acc64,
                                                       double* __restrict__ bu_part, const int* __restrict__ dst_row,
                                                       double* __restrict__ gb, const int* __restrict__ done_flag) {
    if (done_flag != nullptr && *done_flag) return;
    const int s = blockIdx.x * 64 + threadIdx.x;
    const int job = blockIdx.y;
    if (s >= S) return;
    if (job < n_feat) {
        unsigned long long* __restrict__ plo = acc64 + (int64_t)job * S + s;
        unsigned long long* __restrict__ phi = acc64 + ((int64_t)n_feat + job) * S + s;
        const long long lo = (long long)*plo, hi = (long long)*phi;
        *plo = 0ull;
        *phi = 0ull;
        // sum = lo + 2^32 hi (an integer of up to ~95 bits) -> f64: the conversions and the FMA round at 2^-53 relative
        gb[(int64_t)dst_row[job] * S + s] = fma((double)hi, 0x1p32, (double)lo) * 0x1p-52;
    } else {
        const int j = job - n_feat;
        double acc = 0.0;
        for (int c = 0; c < kRedChunks; ++c) acc += bu_part[((int64_t)c * n_u + j) * S + s];  // chunk order: reproducible
        gb[(int64_t)dst_row[job] * S + s] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
void gram_i8_geometry(int64_t N, int SD, int* nsh, int* ny, int64_t* rows_per_wg) {
    const int SB = SD / 32;
    *nsh = (SB + 3) / 4;
    int want = 256 / *nsh;  // one workgroup per CU
    if (want < 1) want = 1;
    int64_t rpw = ((N + want - 1) / want + 31) / 32 * 32;
    if (rpw < 32) rpw = 32;
    *rows_per_wg = rpw;
    *ny = (int)((N + rpw - 1) / rpw);
}

bool gram_i8_supported(int n_c, int n_u, int ND, int64_t N, int SD) {
    const int nf = n_c * n_u + n_u * (n_u + 1) / 2;
    // (a block's x image -- 32 rows of the padded R_trunc copy and of u -- is fetched as at most two 4-KB pieces)
    if ((n_c + 3) / 4 * 4 + n_u > 32 || nf < 1 || nf > kMaxFeat || ND < 1 || ND > 2) return false;
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    return rpw * 128 * 128 * ND < (int64_t)1 << 31;  // i32 accumulators cannot overflow within a row range
}

// The known block of a problem's packed Gram (dmf_api.hip, problem_finalize) through the same kernels: its dense pairs
// (R_trunc column k x column l) are features whose two factors both come from the R_trunc image (n_u = 0), its
// right-hand sides sum_i Rt_ik d_is v_is are k_bu_cols with R_trunc in the place of u.
bool gram_i8_known_supported(int n_c, int ND, int64_t N, int SD) {
    const int nf = n_c * (n_c + 1) / 2;
    if (n_c < 1 || (n_c + 3) / 4 * 4 > 32 || nf > kMaxFeat || ND < 1 || ND > 2) return false;
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    return rpw * 128 * 128 * ND < (int64_t)1 << 31;
}
int64_t gram_i8_slab_words_nf(int64_t N, int SD, int nf) {
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    return (int64_t)ny * 2 * ((nf + 31) / 32 * 32) * SD;
}
int64_t gram_i8_acc_words_nf(int S, int nf, int n_bu) { return (int64_t)2 * nf * S + (int64_t)kRedChunks * n_bu * S; }

// i64 words: [ny][2][slots][SD] slab + acc64[2][nf][S] + (as doubles) bu_part[kRedChunks][n_u][S]
int64_t gram_i8_slab_words(int64_t N, int SD, int n_c, int n_u) {
    const int nf = n_c * n_u + n_u * (n_u + 1) / 2;
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    return (int64_t)ny * 2 * ((nf + 31) / 32 * 32) * SD;
}
int64_t gram_i8_acc_words(int S, int n_c, int n_u) {
    const int nf = n_c * n_u + n_u * (n_u + 1) / 2;
    return (int64_t)2 * nf * S + (int64_t)kRedChunks * n_u * S;
}

size_t gram_i8_w8_lds_bytes(int xl, int nd, int ring) { return (size_t)ring * (4096 * nd + xl * 4096) + (size_t)3 * 2 * (kNSL * 64) * 16; }

template <int XL, int ND, int RING = kRing>
static hipError_t launch_gram_i8_w8_t(const signed char* Dt8, int64_t plane_stride, int SD, const double* Rtp, const double* u,
                                      int64_t N, int n_c, int n_u, const short* fa, const short* fb, int NF, int p0, int MFtot,
                                      long long* slab, const int* done_flag, hipStream_t st) {
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    const size_t lds = gram_i8_w8_lds_bytes(XL, ND, RING);
    static bool lds_limit_raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!lds_limit_raised[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gram_i8_w8<XL, ND, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_limit_raised[dev] = true;
    }
    hipLaunchKernelGGL((k_gram_i8_w8<XL, ND, RING>), dim3(nsh * ny), dim3(512), lds, st, Dt8, plane_stride, SD / 32, Rtp,
                       (n_c + 3) / 4 * 4, u, N, n_c, n_u, fa, fb, NF, p0, MFtot, rpw, slab, SD, done_flag
#ifdef DMF_STAMPS
                       , (unsigned long long*)nullptr
#endif
                       );
    return hipGetLastError();
}

hipError_t launch_gram_i8(const signed char* Dt8, int64_t plane_stride, int SD, int ND, const double* Rtp, const double* u,
                          int64_t N, int n_c, int n_u, const short* fa, const short* fb, int NF, long long* slab,
                          int64_t slab_words, const int* done_flag, int* ny_out, hipStream_t st) {
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    *ny_out = ny;
    const int MFtot = (NF + 31) / 32 * 32;
    // every workgroup (row range) writes its own [2][MFtot][SD] slab: the buffer must hold all of them
    if ((int64_t)ny * 2 * MFtot * SD > slab_words || rpw * 128 * 128 * ND >= (int64_t)1 << 31) return hipErrorInvalidValue;
    // more than 64 features = more launches over the (small) 8-bit planes
    const int chunk = 64;  // features per launch (lane = feature slot of a 64-lane wave; two feature halves of 32)
    const bool wide = (n_c + 3) / 4 * 4 + n_u > 16;  // x image of a block beyond 4 KB: two DMA pieces per thread
    for (int p0 = 0; p0 < NF; p0 += chunk) {
        const int nf = NF - p0 < chunk ? NF - p0 : chunk;
        hipError_t e;
        if (ND == 1) {
            if (wide) e = launch_gram_i8_w8_t<2, 1>(Dt8, plane_stride, SD, Rtp, u, N, n_c, n_u, fa, fb, nf, p0, MFtot, slab, done_flag, st);
            else e = launch_gram_i8_w8_t<1, 1>(Dt8, plane_stride, SD, Rtp, u, N, n_c, n_u, fa, fb, nf, p0, MFtot, slab, done_flag, st);
        } else {  // two count digits (some count above 127: what sequencing data looks like)
            if (wide)  // (16-KB block slots: a ring of six is what the LDS holds beside the three A tiles)
                e = launch_gram_i8_w8_t<2, 2, 6>(Dt8, plane_stride, SD, Rtp, u, N, n_c, n_u, fa, fb, nf, p0, MFtot, slab, done_flag, st);
            else
                e = launch_gram_i8_w8_t<1, 2>(Dt8, plane_stride, SD, Rtp, u, N, n_c, n_u, fa, fb, nf, p0, MFtot, slab, done_flag, st);
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_gram_v2_reduce(const long long* slab_i8, int ny, int NF, int SD, const double* slab_bu, int n_bu_slabs,
                                 int n_u, int S, long long* acc_words, const int* dst_row, double* gb, const int* done_flag,
                                 const double* u2_partials, int n_u2, SolverState* state, hipStream_t st) {
    const int MF = (NF + 31) / 32 * 32;
    unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(acc_words);
    double* bu_part = reinterpret_cast<double*>(acc_words + (int64_t)2 * NF * S);
    hipLaunchKernelGGL(k_gram_v2_reduce, dim3((S + 63) / 64, NF + n_u, kRedChunks), dim3(256), 0, st, slab_i8, ny, MF, SD,
                       slab_bu, n_bu_slabs, n_u, NF, S, acc64, bu_part, done_flag, u2_partials, n_u2, state);
    hipLaunchKernelGGL(k_gram_v2_finish, dim3((S + 63) / 64, NF + n_u), dim3(64), 0, st, n_u, NF, S, acc64, bu_part, dst_row,
                       gb, done_flag);
    return hipGetLastError();
}

}  // namespace dmf
