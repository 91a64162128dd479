// u-dependent Gram entries of the alpha phase on the INTEGER matrix cores, exactly.
//
// The alpha phase (deconvolution.py:93-102) needs, per sample s, G_s = R^T diag(d_s) R and b_s = R^T (d_s * v_s).
// The entries that involve the unknown profiles u and no V are sums over the CpG rows of a per-row feature times a
// count:   cross[k][j][s] = sum_i (Rt_ik u_ij) d_is      uu[j<=l][s] = sum_i (u_ij u_il) d_is
// i.e. one GEMM  G[p][s] = sum_i Z[i][p] D[i][s]  with Z (N x NF) formed on the fly and D the counts.  D is an
// exact small integer and Z lies in [0, 1], so the GEMM is evaluated WITHOUT rounding: Z is written in fixed point,
// z ~ rint(z 2^54) = sum_t a_t 256^t with seven balanced 8-bit digits a_t in [-128, 127], the counts likewise as
// one (d <= 127) or two (d <= 32639) balanced digits, every digit product is accumulated exactly in i32 by
// v_mfma_i32_32x32x32_i8, the per-workgroup i32 sums are added as i64 and the digit weights are applied once at
// the end (k_gram_v2_reduce).  The only rounding against exact arithmetic is rint(z 2^54): |error| <= 2^-55 per
// feature value, zero for z >= 1/4 -- tighter than ANY f64 accumulation of the same sum, whose every addition
// rounds at 2^-53 relative.  (FP64 needs one FMA per (row, feature, sample); here the same product costs 7 i8 MACs
// at 64x the FP64 rate.)
//
// Layouts.  B operand = counts as 8-bit digit planes Dt8[plane][row block of 32][sample block of 32][n 32][k 32]
// (built once per problem, k_build_dt8): lane (n = l & 31, h = l >> 5) loads its 16 bytes k = 16 h .. 16 h + 15 with
// one 16-B global load, the wave 1 KB contiguous.  A operand = digit t of feature p for the block's 32 rows, generated
// by the workgroup into LDS as Atile[h][m][16 B] (m = t * 32 NFT + p): conflict-free ds_read_b128 per (m, h).
// Any bijection (h, byte) -> k the hardware applies is the same for A and B, so only "16 bytes per lane = 16
// consecutive rows" matters; the C layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) is the
// dtype-independent 32x32 map.  A workgroup = 4 waves = 4 sample blocks (128 samples) x one row range, accumulators
// in registers over the whole range (NWT * NFT tiles of 16 VGPRs per wave, one wave per SIMD).
#include "dmf_device.h"
#include "dmf_internal.h"

namespace dmf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

namespace {
constexpr int kNSL = 7;       // balanced base-256 digits of rint(z 2^54)
constexpr int kXStride = 21;  // doubles per staged row of x = (Rt, u): K <= 20, odd stride against bank conflicts
constexpr int kMaxFeat = 96;
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

// ------------------------------------------------------------------------------------------------ the GEMM
// z -> the 64-bit integer rint(z 2^54) + bias, bias = 0x0000808080808080: bytes 0..5 of the sum, XOR 0x80, are the
// balanced digits a_0..a_5, byte 6 is a_6 (0 <= a_6 <= 65: z <= 1)
__device__ __forceinline__ void z_to_biased(double z, unsigned int& lo, unsigned int& hi) {
    const double t = rint(z * 0x1p54);
    const unsigned int h = (unsigned int)(t * 0x1p-32);             // trunc: t >= 0
    const unsigned int l = (unsigned int)fma(-0x1p32, (double)h, t);  // exact remainder
    const unsigned int l2 = l + 0x80808080u;
    lo = l2;
    hi = h + 0x00008080u + (l2 < l ? 1u : 0u);
}

template <int NFT, int ND>
__global__ __launch_bounds__(256) void k_gram_i8(const signed char* __restrict__ Dt8, int64_t plane_stride, int SB,
                                                 const double* __restrict__ Rt, const double* __restrict__ u, int64_t N,
                                                 int n_c, int n_u, const short* __restrict__ feat_a,
                                                 const short* __restrict__ feat_b, int NF, int p0, int MFtot,
                                                 int64_t rows_per_wg, int* __restrict__ slab, int SDs,
                                                 const int* __restrict__ done_flag) {
    // features [p0, p0 + NF) of the table, NF <= 32 NFT: one launch per chunk of features (accumulator registers);
    // MFtot = feature slots of the whole slab
    constexpr int NWT = kNSL + ND - 1;   // digit weights 256^0 .. 256^(NWT-1)
    constexpr int MF = 32 * NFT;         // feature slots per digit
    constexpr int MA = kNSL * MF;        // rows of the A tile
    __shared__ __attribute__((aligned(16))) unsigned int atile[2][2 * MA * 4];  // [buf][h][m][4 dwords]
    __shared__ double xs[2][32 * kXStride];
    __shared__ short fa_s[kMaxFeat], fb_s[kMaxFeat];
    if (done_flag != nullptr && *done_flag) return;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int K = n_c + n_u;
    const int sb = blockIdx.x * 4 + wave;  // this wave's block of 32 samples
    const bool wave_on = sb < SB;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_wg;
    int64_t r_end = r_begin + rows_per_wg;
    if (r_end > N) r_end = N;
    const int nb = r_end > r_begin ? (int)((r_end - r_begin + 31) / 32) : 0;

    for (int i = tid; i < 2 * 2 * MA * 4; i += 256) (&atile[0][0])[i] = 0u;  // feature slots >= NF stay zero digits
    for (int i = tid; i < NF; i += 256) {
        fa_s[i] = feat_a[p0 + i];
        fb_s[i] = feat_b[p0 + i];
    }

    v16i acc[NWT * NFT];
#pragma unroll
    for (int a = 0; a < NWT * NFT; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0;

    // x rows of a block: 32 x K doubles, element e = row * K + col -> thread e, e + 256, e + 512 (K <= 20: < 768)
    int xrow[3], xcol[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const int e = tid + 256 * x;
        xrow[x] = e < 32 * K ? e / K : -1;
        xcol[x] = e < 32 * K ? e - xrow[x] * K : 0;
    }
    auto load_x = [&](int b, double (&xr)[3]) {
        const int64_t row0 = r_begin + (int64_t)b * 32;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            double val = 0.0;
            if (xrow[x] >= 0 && row0 + xrow[x] < r_end)
                val = xcol[x] < n_c ? Rt[(row0 + xrow[x]) * n_c + xcol[x]] : u[(row0 + xrow[x]) * n_u + (xcol[x] - n_c)];
            xr[x] = val;
        }
    };
    auto store_x = [&](int buf, const double (&xr)[3]) {
#pragma unroll
        for (int x = 0; x < 3; ++x)
            if (xrow[x] >= 0) xs[buf][xrow[x] * kXStride + xcol[x]] = xr[x];
    };
    // digits of the block's features: task = (kq = 4 rows, p = feature); 4 x 4 byte transposes turn the rows'
    // 64-bit fixed-point values into one dword (4 rows) per digit
    auto generate = [&](int buf, int xbuf) {
        unsigned int* __restrict__ at = atile[buf];
        const double* __restrict__ xb = xs[xbuf];
        const int kq = tid & 7;
        for (int p = tid >> 3; p < NF; p += 32) {
            const int ia = fa_s[p], ib = fb_s[p];
            unsigned int lo[4], hi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double* __restrict__ xrow = xb + (4 * kq + r) * kXStride;
                z_to_biased(xrow[ia] * xrow[ib], lo[r], hi[r]);
            }
            const unsigned int l01a = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400u), l01b = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602u);
            const unsigned int l23a = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400u), l23b = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602u);
            const unsigned int h01a = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400u), h01b = __builtin_amdgcn_perm(hi[1], hi[0], 0x07030602u);
            const unsigned int h23a = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400u), h23b = __builtin_amdgcn_perm(hi[3], hi[2], 0x07030602u);
            unsigned int dg[kNSL];
            dg[0] = __builtin_amdgcn_perm(l23a, l01a, 0x05040100u) ^ 0x80808080u;
            dg[1] = __builtin_amdgcn_perm(l23a, l01a, 0x07060302u) ^ 0x80808080u;
            dg[2] = __builtin_amdgcn_perm(l23b, l01b, 0x05040100u) ^ 0x80808080u;
            dg[3] = __builtin_amdgcn_perm(l23b, l01b, 0x07060302u) ^ 0x80808080u;
            dg[4] = __builtin_amdgcn_perm(h23a, h01a, 0x05040100u) ^ 0x80808080u;
            dg[5] = __builtin_amdgcn_perm(h23a, h01a, 0x07060302u) ^ 0x80808080u;
            dg[6] = __builtin_amdgcn_perm(h23b, h01b, 0x05040100u);
            const int h = kq >> 2, dw = kq & 3;
#pragma unroll
            for (int t = 0; t < kNSL; ++t) at[((h * MA + t * MF + p) << 2) + dw] = dg[t];
        }
    };
    auto load_b = [&](int b, v4i (&bq)[ND]) {
        const int64_t rb = (r_begin >> 5) + b;  // r_begin is a multiple of 32
        const signed char* __restrict__ src = Dt8 + (rb * SB + (wave_on ? sb : 0)) * 1024 + (lane & 31) * 32 + (lane >> 5) * 16;
#pragma unroll
        for (int d = 0; d < ND; ++d) bq[d] = *reinterpret_cast<const v4i*>(src + d * plane_stride);
    };

    double xr[3];
    v4i bcur[ND], bnext[ND];
    __syncthreads();  // zeroed tiles, feature table
    if (nb > 0) {
        load_x(0, xr);
        store_x(0, xr);
        load_b(0, bcur);
        __syncthreads();
        generate(0, 0);
        if (nb > 1) {
            load_x(1, xr);
            store_x(1, xr);
        }
        __syncthreads();
    }
    for (int b = 0; b < nb; ++b) {
        const int cur = b & 1, nxt = cur ^ 1;
        if (b + 1 < nb) load_b(b + 1, bnext);
        if (b + 2 < nb) load_x(b + 2, xr);
        if (wave_on) {
            const unsigned int* __restrict__ at = atile[cur] + (((lane >> 5) * MA + (lane & 31)) << 2);
#pragma unroll
            for (int t = 0; t < kNSL; ++t)
#pragma unroll
                for (int f = 0; f < NFT; ++f) {
                    const v4i a = *reinterpret_cast<const v4i*>(at + ((t * MF + 32 * f) << 2));
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        acc[(t + d) * NFT + f] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bcur[d], acc[(t + d) * NFT + f], 0, 0, 0);
                }
        }
        if (b + 1 < nb) generate(nxt, nxt);  // xs[nxt] holds block b + 1
        if (b + 2 < nb) store_x(cur, xr);    // xs[cur] (block b) was consumed when A[cur] was generated
        __syncthreads();
#pragma unroll
        for (int d = 0; d < ND; ++d) bcur[d] = bnext[d];
    }

    // ---- slab[y][weight][feature slot][sample] (i32): exact partial sums of this row range
    if (wave_on) {
        const int n = lane & 31, h = lane >> 5;
        int* __restrict__ out = slab + ((int64_t)blockIdx.y * NWT * MFtot + p0) * SDs + sb * 32 + n;
#pragma unroll
        for (int wt = 0; wt < NWT; ++wt)
#pragma unroll
            for (int f = 0; f < NFT; ++f)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int p = 32 * f + m;
                    if (p < NF) out[(int64_t)(wt * MFtot + p) * SDs] = acc[wt * NFT + f][e];
                }
    }
}

// ------------------------------------------------------------------------------------------------ reduce
// gb[dst_row[job]][s] for every u-dependent job of the solver's table:
//   jobs < n_feat (cross / uu):  2^-54 sum_w 256^w (sum_y slab_i8[y][w][job][s])      (i64 sums, weights applied once)
//   jobs >= n_feat (b_u[j]):     sum_g slab_bu[g][j][s] in workgroup order (fixed order: deterministic)
__global__ __launch_bounds__(256) void k_gram_v2_reduce(const int* __restrict__ slab_i8, int ny, int NWT, int MF, int SDs,
                                                        const double* __restrict__ slab_bu, int n_bu_slabs, int n_u,
                                                        int n_feat, int S, const int* __restrict__ dst_row,
                                                        double* __restrict__ gb, const int* __restrict__ done_flag) {
    __shared__ long long part[3][8][64];
    __shared__ double partd[3][64];
    if (done_flag != nullptr && *done_flag) return;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;  // 4 groups split the slabs
    const int s = blockIdx.x * 64 + lane;
    const int job = blockIdx.y;
    const bool active = s < S;
    const int sc = active ? s : S - 1;
    if (job < n_feat) {
        long long tw[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) tw[w] = 0;
        for (int y = grp; y < ny; y += 4) {
            const int* __restrict__ base = slab_i8 + ((int64_t)y * NWT * MF + job) * SDs + sc;
#pragma unroll
            for (int w = 0; w < 8; ++w)
                if (w < NWT) tw[w] += base[(int64_t)w * MF * SDs];
        }
        if (grp > 0) {
#pragma unroll
            for (int w = 0; w < 8; ++w) part[grp - 1][w][lane] = tw[w];
        }
        __syncthreads();
        if (grp == 0 && active) {
            double r = 0.0;
#pragma unroll
            for (int w = 7; w >= 0; --w) {
                if (w < NWT) {
                    const long long t = ((tw[w] + part[0][w][lane]) + part[1][w][lane]) + part[2][w][lane];
                    r = fma(r, 256.0, (double)t);  // Horner from the heaviest digit: <= 1 rounding per step
                }
            }
            gb[(int64_t)dst_row[job] * S + s] = r * 0x1p-54;
        }
    } else {
        const int j = job - n_feat;
        double acc = 0.0;
        // fixed partition and order of the partial sums: bitwise reproducible
        const int per = (n_bu_slabs + 3) / 4;
        const int g0 = grp * per, g1 = g0 + per < n_bu_slabs ? g0 + per : n_bu_slabs;
        for (int g = g0; g < g1; ++g) acc += slab_bu[((int64_t)g * n_u + j) * S + sc];
        if (grp > 0) partd[grp - 1][lane] = acc;
        __syncthreads();
        if (grp == 0 && active) gb[(int64_t)dst_row[job] * S + s] = ((acc + partd[0][lane]) + partd[1][lane]) + partd[2][lane];
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

int gram_i8_weights(int ND) { return kNSL + ND - 1; }

bool gram_i8_supported(int n_c, int n_u, int ND, int64_t N, int SD) {
    const int nf = n_c * n_u + n_u * (n_u + 1) / 2;
    if (n_c + n_u > 20 || nf < 1 || nf > kMaxFeat || ND < 1 || ND > 2) return false;
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    return rpw * 128 * 128 * ND < (int64_t)1 << 31;  // i32 accumulators cannot overflow within a row range
}

int64_t gram_i8_slab_ints(int64_t N, int SD, int n_c, int n_u, int ND) {
    const int nf = n_c * n_u + n_u * (n_u + 1) / 2;
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    return (int64_t)ny * gram_i8_weights(ND) * ((nf + 31) / 32 * 32) * SD;
}

template <int NFT, int ND>
static hipError_t launch_gram_i8_t(const signed char* Dt8, int64_t plane_stride, int SD, const double* Rt, const double* u,
                                   int64_t N, int n_c, int n_u, const short* fa, const short* fb, int NF, int p0, int MFtot,
                                   int* slab, const int* done_flag, hipStream_t st) {
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    hipLaunchKernelGGL((k_gram_i8<NFT, ND>), dim3(nsh, ny), dim3(256), 0, st, Dt8, plane_stride, SD / 32, Rt, u, N, n_c,
                       n_u, fa, fb, NF, p0, MFtot, rpw, slab, SD, done_flag);
    return hipGetLastError();
}

hipError_t launch_gram_i8(const signed char* Dt8, int64_t plane_stride, int SD, int ND, const double* Rt, const double* u,
                          int64_t N, int n_c, int n_u, const short* fa, const short* fb, int NF, int* slab,
                          const int* done_flag, int* ny_out, hipStream_t st) {
    int nsh, ny;
    int64_t rpw;
    gram_i8_geometry(N, SD, &nsh, &ny, &rpw);
    *ny_out = ny;
    const int MFtot = (NF + 31) / 32 * 32;
    // accumulator registers: (7 + ND - 1) * NFT tiles of 16 per wave -> 64 features per launch with one count digit,
    // 32 with two; more features = more launches over the (small) 8-bit planes
    const int chunk = ND == 1 ? 64 : 32;
    for (int p0 = 0; p0 < NF; p0 += chunk) {
        const int nf = NF - p0 < chunk ? NF - p0 : chunk;
        hipError_t e;
        if (ND == 1 && nf > 32) e = launch_gram_i8_t<2, 1>(Dt8, plane_stride, SD, Rt, u, N, n_c, n_u, fa, fb, nf, p0, MFtot, slab, done_flag, st);
        else if (ND == 1) e = launch_gram_i8_t<1, 1>(Dt8, plane_stride, SD, Rt, u, N, n_c, n_u, fa, fb, nf, p0, MFtot, slab, done_flag, st);
        else e = launch_gram_i8_t<1, 2>(Dt8, plane_stride, SD, Rt, u, N, n_c, n_u, fa, fb, nf, p0, MFtot, slab, done_flag, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_gram_v2_reduce(const int* slab_i8, int ny, int ND, int NF, int SD, const double* slab_bu, int n_bu_slabs,
                                 int n_u, int S, const int* dst_row, double* gb, const int* done_flag, hipStream_t st) {
    const int MF = (NF + 31) / 32 * 32;
    hipLaunchKernelGGL(k_gram_v2_reduce, dim3((S + 63) / 64, NF + n_u), dim3(256), 0, st, slab_i8, ny, gram_i8_weights(ND),
                       MF, SD, slab_bu, n_bu_slabs, n_u, NF, S, dst_row, gb, done_flag);
    return hipGetLastError();
}

}  // namespace dmf
