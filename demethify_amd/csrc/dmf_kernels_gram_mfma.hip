// Any-shape per-sample weighted Gram accumulation on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), for the
// shapes whose accumulators do not fit the registers of the lane-per-sample kernel (dmf_kernels_gram.hip):
// many unknown types (the --ic sweep goes to n_u = 25, ic.py:171) or many known x unknown pairs.
//
// Every job of the solver's table is one row of the per-sample packed Gram that involves u:
//     G[(k, l)][s] = sum_r f_k(r) f_l(r) d[r][s]        f_x(r) = R_trunc[r][x] (x < n_c) or u[r][x - n_c]
//     b[k][s]      = sum_r f_k(r) (d[r][s] v[r][s])     (jobs with l = K, the "v" column)
// i.e. a GEMM  [jobs x rows] x [rows x samples]  whose A operand is a product of two row features (formed on
// the fly: one multiply per operand) and whose B operand is the D tile itself, or D * V for the b rows: those
// tiles go to waves of their own (a wave never mixes the two kinds), so only these waves load V as well.
//
// Workgroup = 8 waves on one 64-sample column group and one chunk of rows.  Wave w owns MTW consecutive
// 16-job tiles and all four 16-sample tiles of the group: 16 MTW accumulators per lane, live over the whole
// chunk.  Per 16-row block the 16 x K row features are staged in LDS once for the workgroup; the D operands of
// the next block are prefetched into registers while the MFMAs of the current one run.
// Layouts (tools/mfma_probe.hip): A[i][k]: lane (i = l & 15, k = l >> 4); B[k][j]: lane (k = l >> 4, j = l & 15);
// C register r of lane l = C[(l >> 4) + 4 r][l & 15].
// Output: slab[y][job][s] per row chunk y, summed in fixed order by k_gram_reduce: deterministic.
#include "dmf_internal.h"

namespace dmf {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kGramMfmaWaves = 8;
constexpr int kGramMfmaMaxTilesPerWave = 4;  // 16 accumulators per tile: 5 and more spill

// Waves [0, v_wave0) take the jobs [job_begin, job_end) (B = D), waves [v_wave0, 8) the "v" jobs
// [vjob_begin, vjob_end) (B = D * V); MTW consecutive 16-job tiles per wave.
typedef __attribute__((address_space(1))) const void gmem_void;
typedef __attribute__((address_space(3))) int lds_int;

// DMA: S is even, so a block's 16 x 64 tile of D (and of V, if some wave owns b rows) is fetched once per
// workgroup straight into LDS with global_load_lds_dwordx4 (one 16-B request per thread and tile, no VGPR
// destination, a whole block ahead) and every wave reads its B operands from there; otherwise (odd S: 16-B
// requests would straddle row ends) each wave loads its own B operands into registers, one block ahead.
template <int MTW, bool DMA>
__global__ __launch_bounds__(512) void k_gram_mfma(const double* __restrict__ V, const double* __restrict__ D,
                                                   const double* __restrict__ Rt, const double* __restrict__ u,
                                                   int64_t N, int S, int n_c, int n_u,
                                                   const short* __restrict__ job_k, const short* __restrict__ job_l,
                                                   int n_jobs, int job_begin, int job_end, int vjob_begin,
                                                   int vjob_end, int v_wave0, int64_t rows_per_chunk,
                                                   double* __restrict__ slab, const int* __restrict__ done_flag) {
    __shared__ double feat[2][16 * (kMaxK + 1)];  // [buffer][row][feature], row stride K + 1
    __shared__ __attribute__((aligned(16))) double dtile[DMA ? 2 : 1][DMA ? 16 * 64 : 2];
    __shared__ __attribute__((aligned(16))) double vtile[DMA ? 2 : 1][DMA ? 16 * 64 : 2];
    if (done_flag != nullptr && *done_flag) return;
    const int K = n_c + n_u, FS = K + 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int i16 = lane & 15, kq = lane >> 4;
    const int s0 = blockIdx.x * 64;
    const bool VJOBS = wave >= v_wave0;  // wave-uniform
    const int tile0 = VJOBS ? (wave - v_wave0) * MTW : wave * MTW;
    const int jb = VJOBS ? vjob_begin : job_begin, je = VJOBS ? vjob_end : job_end;

    // this lane's job in each of the wave's tiles (A operand row i16), and its two features
    int fk[MTW], fl[MTW];
    double keep[MTW];
#pragma unroll
    for (int x = 0; x < MTW; ++x) {
        const int job = jb + (tile0 + x) * 16 + i16;
        const bool ok = job < je;
        fk[x] = ok ? job_k[job] : 0;
        fl[x] = ok ? job_l[job] : 0;
        if (VJOBS) fl[x] = fk[x];  // the "v" column is not a row feature: A = f_k alone
        keep[x] = ok ? 1.0 : 0.0;
    }
    // this lane's B-operand columns (clamped: out-of-range samples are computed and never stored)
    int col[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c = s0 + 16 * nt + i16;
        col[nt] = c < S ? c : S - 1;
    }

    v4d acc[MTW][4];
#pragma unroll
    for (int x = 0; x < MTW; ++x)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[x][nt] = v4d{0.0, 0.0, 0.0, 0.0};

    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < N ? r0 + rows_per_chunk : N;
    if (r0 >= r1) return;  // (whole workgroup: no barrier is skipped by part of it)

    // Row features travel global -> registers -> LDS in two steps one block apart (a thread carries at most
    // 16 K / 512 = 2 of them), so that no wave ever waits for a load it has just issued.
    auto fetch_feat = [&](int64_t row0, double (&fr)[2]) {  // 16 x K features of rows row0 .. row0 + 15 (zero past N)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = threadIdx.x + 512 * h;
            const int r = idx / K, x = idx - r * K;
            const int64_t row = row0 + r;
            double f = 0.0;
            if (idx < 16 * K && row < N) f = x < n_c ? Rt[row * n_c + x] : u[row * n_u + (x - n_c)];
            fr[h] = f;
        }
    };
    auto park_feat = [&](int buf, const double (&fr)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = threadIdx.x + 512 * h;
            const int r = idx / K, x = idx - r * K;
            if (idx < 16 * K) feat[buf][r * FS + x] = fr[h];
        }
    };
    auto load_b = [&](int64_t row0, double (&b)[4][4]) {  // B operands of the 4 k-steps x 4 sample tiles
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            int64_t row = row0 + 4 * st + kq;
            row = row < N ? row : N - 1;  // rows past N meet zero features
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) b[st][nt] = D[row * S + col[nt]];
            if (VJOBS) {  // wave-uniform: only the waves that own b rows read V
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) b[st][nt] *= V[row * S + col[nt]];
            }
        }
    };

    const bool any_v = v_wave0 < kGramMfmaWaves;  // some wave owns b rows: the V tile is needed too
    // DMA: thread t fetches the pair of samples 2 (t % 32), + 1 of tile row t / 32 (clamped into the matrix:
    // clamped rows meet zero features, clamped columns are computed and never stored)
    auto fetch_tile = [&](int tb, int64_t row0) {
        if constexpr (DMA) {
            int64_t row = row0 + (threadIdx.x >> 5);
            row = row < N ? row : N - 1;
            int c = s0 + 2 * (threadIdx.x & 31);
            c = c < S ? c : S - 2;
            const int64_t off = row * S + c;
            // one wave-wide request writes 64 x 16 B = 1 KB of contiguous LDS: wave w covers tile rows 2 w, 2 w + 1
            lds_int* dst_d = (lds_int*)(&dtile[tb][0]) + (threadIdx.x >> 6) * 256;
            __builtin_amdgcn_global_load_lds((gmem_void*)(D + off), dst_d, 16, 0, 0);
            if (any_v) {
                lds_int* dst_v = (lds_int*)(&vtile[tb][0]) + (threadIdx.x >> 6) * 256;
                __builtin_amdgcn_global_load_lds((gmem_void*)(V + off), dst_v, 16, 0, 0);
            }
        }
    };

    // software pipeline over 16-row blocks: while block i runs on the matrix cores, the D operands of block
    // i + 1 and the row features of block i + 2 are in flight
    double bcur[DMA ? 1 : 4][4], bnext[DMA ? 1 : 4][4], f_next[2];
    {
        double f0[2];
        fetch_feat(r0, f0);
        if constexpr (DMA) fetch_tile(0, r0);
        else load_b(r0, bcur);
        park_feat(0, f0);
    }
    fetch_feat(r0 + 16, f_next);  // (rows past r1 are simply not used)
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int64_t row0 = r0; row0 < r1; row0 += 16, buf ^= 1) {
        const bool more = row0 + 16 < r1;
        double f_after[2] = {0.0, 0.0};
        if (more) {
            park_feat(buf ^ 1, f_next);  // fetched one block ago; the buffer's readers passed the last barrier
            if constexpr (DMA) fetch_tile(buf ^ 1, row0 + 16);
            else load_b(row0 + 16, bnext);
            fetch_feat(row0 + 32, f_after);
        }
        const double* __restrict__ f = feat[buf];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            double a[MTW], b[4];
#pragma unroll
            for (int x = 0; x < MTW; ++x) {
                const double fa = f[(4 * st + kq) * FS + fk[x]];
                const double fb = VJOBS ? keep[x] : keep[x] * f[(4 * st + kq) * FS + fl[x]];
                a[x] = fa * fb;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if constexpr (DMA) {
                    b[nt] = dtile[buf][(4 * st + kq) * 64 + 16 * nt + i16];
                    if (VJOBS) b[nt] *= vtile[buf][(4 * st + kq) * 64 + 16 * nt + i16];
                } else {
                    b[nt] = bcur[st][nt];
                }
            }
#pragma unroll
            for (int x = 0; x < MTW; ++x)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[x][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[nt], acc[x][nt], 0, 0, 0);
        }
        if (more) {
            if constexpr (!DMA) {
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) bcur[st][nt] = bnext[st][nt];
            }
            f_next[0] = f_after[0];
            f_next[1] = f_after[1];
        }
        // the tile fetch issued at the top of this block has landed before anyone passes the barrier (so has
        // the feature fetch of block i + 2: both had the whole block to arrive)
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // C register r of lane l: job row (l >> 4) + 4 r of the tile, sample column l & 15
#pragma unroll
    for (int x = 0; x < MTW; ++x) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int job = jb + (tile0 + x) * 16 + kq + 4 * r;
            if (job < je) {
                double* __restrict__ out = slab + ((int64_t)blockIdx.y * n_jobs + job) * S;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int s = s0 + 16 * nt + i16;
                    if (s < S) out[s] = acc[x][nt][r];
                }
            }
        }
    }
}

static void gram_mfma_geometry(int64_t N, int S, int* nsx, int* ny, int64_t* rows_per_chunk) {
    *nsx = (S + 63) / 64;
    int64_t want = 512 / (*nsx);  // two 8-wave workgroups per CU when the registers allow (<= 2 tiles per wave)
    if (want < 1) want = 1;
    int64_t rpc = (N + want - 1) / want;
    rpc = (rpc + 15) / 16 * 16;
    if (rpc < 64) rpc = 64;
    *rows_per_chunk = rpc;
    *ny = (int)((N + rpc - 1) / rpc);
}

int64_t gram_mfma_slab_doubles(int64_t N, int S, int n_jobs) {
    int nsx, ny;
    int64_t rpc;
    gram_mfma_geometry(N, S, &nsx, &ny, &rpc);
    return (int64_t)ny * n_jobs * S;
}

// jobs [0, n_dense) have B = D, jobs [n_dense, count) are the "v" column (B = D * V); ny_out = slab rows per job
hipError_t launch_gram_mfma(const double* V, const double* D, const double* Rt, const double* u, int64_t N, int S,
                            int n_c, int n_u, GramJobTable jobs, int n_dense, double* slab, int64_t slab_doubles,
                            const int* done_flag, int* ny_out, hipStream_t st) {
    if (jobs.count <= 0 || n_dense < 0 || n_dense > jobs.count || n_c + n_u > kMaxK) return hipErrorInvalidValue;
    int nsx, ny;
    int64_t rpc;
    gram_mfma_geometry(N, S, &nsx, &ny, &rpc);
    if ((int64_t)ny * jobs.count * S > slab_doubles) return hipErrorInvalidValue;
    *ny_out = ny;
    const dim3 grid(nsx, ny), block(kGramMfmaWaves * 64);
    const bool dma = (S & 1) == 0 && S >= 2 && ((uintptr_t)D & 15) == 0 && ((uintptr_t)V & 15) == 0;
    int d_begin = 0, v_begin = n_dense;
    while (d_begin < n_dense || v_begin < jobs.count) {
        const int tiles_d = (n_dense - d_begin + 15) / 16, tiles_v = (jobs.count - v_begin + 15) / 16;
        // smallest tiles-per-wave that fits both kinds on 8 waves; what does not fit waits for the next launch
        int mtw = 1;
        while (mtw < kGramMfmaMaxTilesPerWave && (tiles_d + mtw - 1) / mtw + (tiles_v + mtw - 1) / mtw > kGramMfmaWaves)
            ++mtw;
        int waves_v = (tiles_v + mtw - 1) / mtw;
        if (waves_v > kGramMfmaWaves / 2) waves_v = kGramMfmaWaves / 2;
        int waves_d = (tiles_d + mtw - 1) / mtw;
        if (waves_d > kGramMfmaWaves - waves_v) waves_d = kGramMfmaWaves - waves_v;
        int d_end = d_begin + waves_d * mtw * 16, v_end = v_begin + waves_v * mtw * 16;
        if (d_end > n_dense) d_end = n_dense;
        if (v_end > jobs.count) v_end = jobs.count;
        const int v_wave0 = kGramMfmaWaves - waves_v;
#define DMF_CASE(M_)                                                                                          \
    case M_:                                                                                                  \
        if (dma)                                                                                              \
            hipLaunchKernelGGL((k_gram_mfma<M_, true>), grid, block, 0, st, V, D, Rt, u, N, S, n_c, n_u,      \
                               jobs.k_idx, jobs.l_idx, jobs.count, d_begin, d_end, v_begin, v_end, v_wave0, \
                               rpc, slab, done_flag);                                                         \
        else                                                                                                  \
            hipLaunchKernelGGL((k_gram_mfma<M_, false>), grid, block, 0, st, V, D, Rt, u, N, S, n_c, n_u,     \
                               jobs.k_idx, jobs.l_idx, jobs.count, d_begin, d_end, v_begin, v_end, v_wave0, \
                               rpc, slab, done_flag);                                                         \
        break;
        switch (mtw) {
            DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4)
            default: return hipErrorInvalidValue;
        }
#undef DMF_CASE
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        d_begin = d_end;
        v_begin = v_end;
    }
    return hipSuccess;
}

}  // namespace dmf
