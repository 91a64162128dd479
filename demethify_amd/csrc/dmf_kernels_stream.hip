// Streaming kernels: count conversion, reductions, row gather, the direct weighted cost
// (demethify/deconvolution.py:15-17) and the generic per-sample weighted Gram accumulation
// that feeds the alpha phase (G_s = R^T diag(d_s) R, b_s = R^T (d_s * v_s)).
#include "dmf_device.h"
#include "dmf_internal.h"

namespace dmf {

// ------------------------------------------------------------------ small utilities
__global__ __launch_bounds__(256) void k_convert_counts(const long long* __restrict__ src,
                                                        double* __restrict__ dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) dst[i] = (double)src[i];
}

hipError_t launch_convert_counts(const long long* src, double* dst, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_convert_counts, dim3(blocks), dim3(256), 0, st, src, dst, n);
    return hipGetLastError();
}

// MODE 0: max(x)   MODE 1: sum(x^2)   MODE 2: max |x - (double)(float)x|  (is x exact in f32?)
// MODE 3: max(x) if every x is a non-negative integer, +inf otherwise   MODE 4: 0 if every x lies in [0, 1], else 1
template <int MODE>
__global__ __launch_bounds__(256) void k_reduce_partial(const double* __restrict__ x, int64_t n,
                                                        double* __restrict__ partial,
                                                        const int* __restrict__ done_flag) {
    __shared__ double red[4];
    if (done_flag != nullptr && *done_flag) return;
    double acc = MODE == 1 ? 0.0 : -INFINITY;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const double v = x[i];
        if (MODE == 1) acc = fma(v, v, acc);
        else if (MODE == 0) acc = fmax(acc, v);
        else if (MODE == 2) acc = fmax(acc, fabs(v - (double)(float)v));
        else if (MODE == 3) acc = fmax(acc, (v >= 0.0 && v == rint(v)) ? v : INFINITY);
        else acc = fmax(acc, (v >= 0.0 && v <= 1.0) ? 0.0 : 1.0);
    }
    const double tot = MODE == 1 ? block_sum<256>(acc, red) : block_max<256>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_reduce_final(const double* __restrict__ partial, int n,
                                                      double* __restrict__ out,
                                                      const int* __restrict__ done_flag) {
    __shared__ double red[4];
    if (done_flag != nullptr && *done_flag) return;
    double acc = MODE == 1 ? 0.0 : -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (MODE == 1) acc += partial[i];
        else acc = fmax(acc, partial[i]);
    }
    const double tot = MODE == 1 ? block_sum<256>(acc, red) : block_max<256>(acc, red);
    if (threadIdx.x == 0) *out = tot;
}

static inline int reduce_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

hipError_t launch_max_f64(const double* x, int64_t n, double* scratch, double* out, hipStream_t st) {
    const int nb = reduce_blocks(n);
    hipLaunchKernelGGL(k_reduce_partial<0>, dim3(nb), dim3(256), 0, st, x, n, scratch, (const int*)nullptr);
    hipLaunchKernelGGL(k_reduce_final<0>, dim3(1), dim3(256), 0, st, scratch, nb, out, (const int*)nullptr);
    return hipGetLastError();
}

hipError_t launch_f32_residual_max(const double* x, int64_t n, double* scratch, double* out, hipStream_t st) {
    const int nb = reduce_blocks(n);
    hipLaunchKernelGGL(k_reduce_partial<2>, dim3(nb), dim3(256), 0, st, x, n, scratch, (const int*)nullptr);
    hipLaunchKernelGGL(k_reduce_final<0>, dim3(1), dim3(256), 0, st, scratch, nb, out, (const int*)nullptr);
    return hipGetLastError();
}

hipError_t launch_int_count_max(const double* x, int64_t n, double* scratch, double* out, hipStream_t st) {
    const int nb = reduce_blocks(n);
    hipLaunchKernelGGL(k_reduce_partial<3>, dim3(nb), dim3(256), 0, st, x, n, scratch, (const int*)nullptr);
    hipLaunchKernelGGL(k_reduce_final<0>, dim3(1), dim3(256), 0, st, scratch, nb, out, (const int*)nullptr);
    return hipGetLastError();
}

hipError_t launch_unit_range_check(const double* x, int64_t n, double* scratch, double* out, hipStream_t st) {
    const int nb = reduce_blocks(n);
    hipLaunchKernelGGL(k_reduce_partial<4>, dim3(nb), dim3(256), 0, st, x, n, scratch, (const int*)nullptr);
    hipLaunchKernelGGL(k_reduce_final<0>, dim3(1), dim3(256), 0, st, scratch, nb, out, (const int*)nullptr);
    return hipGetLastError();
}

hipError_t launch_sumsq_f64(const double* x, int64_t n, double* scratch, double* out,
                            const int* done_flag, hipStream_t st) {
    const int nb = reduce_blocks(n);
    hipLaunchKernelGGL(k_reduce_partial<1>, dim3(nb), dim3(256), 0, st, x, n, scratch, done_flag);
    hipLaunchKernelGGL(k_reduce_final<1>, dim3(1), dim3(256), 0, st, scratch, nb, out, done_flag);
    return hipGetLastError();
}

// dst[r][:] = src[idx[r]][:]; one wave per destination row chunk (bootstrap.py:28)
__global__ __launch_bounds__(256) void k_gather_rows(const double* __restrict__ src,
                                                     double* __restrict__ dst,
                                                     const long long* __restrict__ idx,
                                                     int64_t n_idx, int64_t width) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t r = (int64_t)blockIdx.x * 4 + wave;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (; r < n_idx; r += stride) {
        const double* s = src + idx[r] * width;
        double* d = dst + r * width;
        for (int64_t c = lane; c < width; c += 64) d[c] = s[c];
    }
}

hipError_t launch_gather_rows(const double* src, double* dst, const long long* idx, int64_t n_idx,
                              int64_t width, hipStream_t st) {
    if (n_idx <= 0 || width <= 0) return hipSuccess;
    int64_t b = (n_idx + 3) / 4;
    if (b > 65536) b = 65536;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)b), dim3(256), 0, st, src, dst, idx, n_idx, width);
    return hipGetLastError();
}

// flag |= 1 if any idx[r] lies outside [0, n_src) (dmf_problem_gather_device: the range check of a device index array)
__global__ __launch_bounds__(256) void k_index_range_check(const long long* __restrict__ idx, int64_t n_idx, int64_t n_src,
                                                           unsigned int* __restrict__ flag) {
    bool bad = false;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_idx; r += (int64_t)gridDim.x * 256) {
        const long long v = idx[r];
        bad = bad || v < 0 || v >= n_src;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

hipError_t launch_index_range_check(const long long* idx, int64_t n_idx, int64_t n_src, unsigned int* flag, hipStream_t st) {
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(unsigned int), st);
    if (e != hipSuccess) return e;
    int64_t b = (n_idx + 255) / 256;
    if (b > 1024) b = 1024;
    hipLaunchKernelGGL(k_index_range_check, dim3((unsigned)(b < 1 ? 1 : b)), dim3(256), 0, st, idx, n_idx, n_src, flag);
    return hipGetLastError();
}

// dst[r][0..width_dst) = src[r][0..width_src) followed by zeros (row-padded copy of R_trunc)
__global__ __launch_bounds__(256) void k_pad_rows(const double* __restrict__ src, double* __restrict__ dst,
                                                  int64_t n_rows, int width_src, int width_dst) {
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = n_rows * width_dst, stride = (int64_t)gridDim.x * 256;
    for (; e < total; e += stride) {
        const int64_t r = e / width_dst;
        const int c = (int)(e - r * width_dst);
        dst[e] = c < width_src ? src[r * width_src + c] : 0.0;
    }
}

hipError_t launch_pad_rows(const double* src, double* dst, int64_t n_rows, int width_src, int width_dst,
                           hipStream_t st) {
    if (n_rows <= 0 || width_dst <= 0) return hipSuccess;
    int64_t b = (n_rows * width_dst + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)b), dim3(256), 0, st, src, dst, n_rows, width_src, width_dst);
    return hipGetLastError();
}

// ------------------------------------------------------------------ direct weighted cost
// sum_{i,s} d_is (v_is - sum_k R_ik alpha_ks)^2 with R = [Rt | u]; alpha staged in LDS when it fits.
__global__ __launch_bounds__(256) void k_cost(const double* __restrict__ V, const double* __restrict__ D,
                                              const double* __restrict__ Rt, const double* __restrict__ u,
                                              const double* __restrict__ alpha, int64_t N, int S,
                                              int n_c, int n_u, int alpha_in_lds,
                                              double* __restrict__ partial) {
    extern __shared__ double lds_dyn[];
    __shared__ double red[4];
    const int K = n_c + n_u;
    const double* A = alpha;
    if (alpha_in_lds) {
        for (int i = threadIdx.x; i < K * S; i += 256) lds_dyn[i] = alpha[i];
        __syncthreads();
        A = lds_dyn;
    }
    const int tpr = S < 256 ? S : 256;      // threads per row
    const int rows_per_tile = 256 / tpr;
    const int r = threadIdx.x / tpr, c = threadIdx.x - r * tpr;
    double acc = 0.0;
    if (r < rows_per_tile) {
        // four rows in flight per thread (one row at a time leaves a single V / D load outstanding per lane:
        // 5.8 ms at 1e6 x 256 against ~1 ms for the stream); rows past N are clamped and weigh 0
        const int64_t stride = (int64_t)gridDim.x * rows_per_tile;
        for (int64_t i0 = (int64_t)blockIdx.x * rows_per_tile + r; i0 < N; i0 += 4 * stride) {
            int64_t row[4];
            double w[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int64_t i = i0 + x * stride;
                w[x] = i < N ? 1.0 : 0.0;
                row[x] = i < N ? i : N - 1;
            }
            for (int s = c; s < S; s += tpr) {
                double v[4], d[4], pred[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    v[x] = V[row[x] * S + s];
                    d[x] = D[row[x] * S + s];
                    pred[x] = 0.0;
                }
                for (int k = 0; k < n_c; ++k) {
                    const double ak = A[k * S + s];
#pragma unroll
                    for (int x = 0; x < 4; ++x) pred[x] = fma(Rt[row[x] * n_c + k], ak, pred[x]);
                }
                for (int j = 0; j < n_u; ++j) {
                    const double aj = A[(n_c + j) * S + s];
#pragma unroll
                    for (int x = 0; x < 4; ++x) pred[x] = fma(u[row[x] * n_u + j], aj, pred[x]);
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) {  // same per-row arithmetic as before; rows are summed in index order
                    const double e = v[x] - pred[x];
                    acc = fma(w[x] * d[x] * e, e, acc);
                }
            }
        }
    }
    const double tot = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// Column-resident variant for the shapes of the fast solver paths (n_c <= 16, n_u <= 4): lane = sample, the lane's
// alpha column lives in registers, a row's profile values (R_trunc padded copy, u) are wave-uniform scalar loads that
// feed v_fma_f64 directly, eight rows of V / D are in flight per wave.  HBM-bound: one read of V and of the counts
// (u16 copy when the problem has one: 10 instead of 16 bytes per element).  Same per-element arithmetic as k_cost.
template <int NKC, int NU, bool D16>
__global__ __launch_bounds__(256) void k_cost_cols(const double* __restrict__ V, const void* __restrict__ Dv, int SD,
                                                   const double* __restrict__ Rtp, const double* __restrict__ u,
                                                   const double* __restrict__ alpha, int64_t N, int S, int n_c,
                                                   double* __restrict__ partial) {
    constexpr int NCT = 4 * NKC;
    constexpr int kRows = 8;
    __shared__ double red[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int s = blockIdx.y * 64 + lane;
    const bool active = s < S;
    const int sc = active ? s : S - 1;
    double ak[NCT > 0 ? NCT : 1], aj[NU > 0 ? NU : 1];
#pragma unroll
    for (int k = 0; k < NCT; ++k) ak[k] = k < n_c ? alpha[(int64_t)k * S + sc] : 0.0;
#pragma unroll
    for (int j = 0; j < NU; ++j) aj[j] = alpha[(int64_t)(n_c + j) * S + sc];
    const double* __restrict__ Df = reinterpret_cast<const double*>(Dv);
    const unsigned short* __restrict__ Dh = reinterpret_cast<const unsigned short*>(Dv);
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i0 = (int64_t)blockIdx.x * 4 + wave; i0 < N; i0 += kRows * stride) {
        double v[kRows], d[kRows];
        int64_t row[kRows];
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const int64_t i = i0 + x * stride;
            row[x] = i < N ? i : N - 1;
            v[x] = V[row[x] * S + sc];
            if constexpr (D16) d[x] = (double)Dh[row[x] * SD + sc];
            else d[x] = Df[row[x] * S + sc];
            if (i >= N) d[x] = 0.0;  // rows past N weigh nothing
        }
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const double* __restrict__ rt_row = Rtp + row[x] * NCT;
            const double* __restrict__ u_row = u + row[x] * NU;
            double pred = 0.0;
#pragma unroll
            for (int k = 0; k < NCT; ++k) pred = fma(rt_row[k], ak[k], pred);
#pragma unroll
            for (int j = 0; j < NU; ++j) pred = fma(u_row[j], aj[j], pred);
            const double e = v[x] - pred;
            acc = fma(d[x] * e, e, acc);
        }
    }
    if (!active) acc = 0.0;
    const double tot = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// The same kernel with TWO adjacent samples per lane (the u16 count copy; odd S: see `lone`): one 16-byte load of V
// and one 4-byte load of the counts per row and lane instead of 8 + 2 bytes -- half the load instructions per byte.
// The headline shape streams 2.56 GB per evaluation; the one-sample form reached 3.9 TB/s, 62 % of what a plain copy
// gets on this part.  Per-element arithmetic as above; a lane's two samples have an accumulator each.
template <int NKC, int NU, bool ODD>
__global__ __launch_bounds__(256) void k_cost_cols2(const double* __restrict__ V, const unsigned short* __restrict__ Dh, int SD,
                                                    const double* __restrict__ Rtp, const double* __restrict__ u,
                                                    const double* __restrict__ alpha, int64_t N, int S, int n_c,
                                                    double* __restrict__ partial) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int NCT = 4 * NKC;
    constexpr int kRows = 8;
    __shared__ double red[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int s = blockIdx.y * 128 + 2 * lane;
    const bool active = s < S;
    // odd S: the row's last sample sits alone in its lane -- it takes the upper half of the pair one element lower (never
    // past the end of a row), its partner's count is the zero padding of the u16 copy; rows of V then start 8 bytes off a
    // 16-byte boundary every other time, which 16-byte loads take (tools/align_probe.hip)
    const bool lone = ODD && s == S - 1;  // (ODD: S is odd -- the even form carries none of this)
    const int sc = lone ? S - 2 : (active ? s : 0);  // V / alpha column of the pair fetched
    const int sd = active ? s : 0;                   // count pair (4-byte aligned: s is even)
    typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));
    double ak[NCT > 0 ? NCT : 1][2], aj[NU > 0 ? NU : 1][2];
#pragma unroll
    for (int k = 0; k < NCT; ++k) {
#pragma unroll
        for (int h = 0; h < 2; ++h) ak[k][h] = k < n_c ? alpha[(int64_t)k * S + sc + h] : 0.0;
        if (lone) ak[k][0] = ak[k][1];
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
#pragma unroll
        for (int h = 0; h < 2; ++h) aj[j][h] = alpha[(int64_t)(n_c + j) * S + sc + h];
        if (lone) aj[j][0] = aj[j][1];
    }
    double acc0 = 0.0, acc1 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i0 = (int64_t)blockIdx.x * 4 + wave; i0 < N; i0 += kRows * stride) {
        v2d v[kRows];
        unsigned int dd[kRows];
        int64_t row[kRows];
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const int64_t i = i0 + x * stride;
            row[x] = i < N ? i : N - 1;
            const v2d_u vl = *reinterpret_cast<const v2d_u*>(V + row[x] * S + sc);
            v[x] = v2d{lone ? vl.y : vl.x, vl.y};
            dd[x] = *reinterpret_cast<const unsigned int*>(Dh + row[x] * SD + sd);
            if (i >= N) dd[x] = 0u;  // rows past N weigh nothing
        }
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const double* __restrict__ rt_row = Rtp + row[x] * NCT;
            const double* __restrict__ u_row = u + row[x] * NU;
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int k = 0; k < NCT; ++k) {
                const double r = rt_row[k];
                p0 = fma(r, ak[k][0], p0);
                p1 = fma(r, ak[k][1], p1);
            }
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const double r = u_row[j];
                p0 = fma(r, aj[j][0], p0);
                p1 = fma(r, aj[j][1], p1);
            }
            const double e0 = v[x].x - p0, e1 = v[x].y - p1;
            acc0 = fma((double)(dd[x] & 0xFFFFu) * e0, e0, acc0);
            acc1 = fma((double)(dd[x] >> 16) * e1, e1, acc1);
        }
    }
    double acc = acc0 + acc1;
    if (!active) acc = 0.0;
    const double tot = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// v^T D v per sample column (the constant term of the Gram-form cost, one entry of the packed Gram's known block):
// gb[s] = sum_i d_is v_is^2.  Lane = sample, eight rows in flight per wave, workgroup partials summed in workgroup order
// by the last launch.  The generic Gram kernel spent 2.7 ms on this one job at the headline shape (every bootstrap
// replicate builds a problem); this stream takes the time of one read of V and the counts.
template <bool D16>
__global__ __launch_bounds__(256) void k_vdv_cols(const double* __restrict__ V, const void* __restrict__ Dv, int SD, int64_t N,
                                                  int S, double* __restrict__ slab) {
    constexpr int kRows = 8;
    __shared__ double red[3][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int s = blockIdx.y * 64 + lane;
    const bool active = s < S;
    const int sc = active ? s : S - 1;
    const double* __restrict__ Df = reinterpret_cast<const double*>(Dv);
    const unsigned short* __restrict__ Dh = reinterpret_cast<const unsigned short*>(Dv);
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i0 = (int64_t)blockIdx.x * 4 + wave; i0 < N; i0 += kRows * stride) {
        double v[kRows], d[kRows];
#pragma unroll
        for (int x = 0; x < kRows; ++x) {
            const int64_t i = i0 + x * stride;
            const int64_t row = i < N ? i : N - 1;
            v[x] = V[row * S + sc];
            if constexpr (D16) d[x] = (double)Dh[row * SD + sc];
            else d[x] = Df[row * S + sc];
            if (i >= N) d[x] = 0.0;
        }
#pragma unroll
        for (int x = 0; x < kRows; ++x) acc = fma(d[x] * v[x], v[x], acc);
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && active) slab[(int64_t)blockIdx.x * S + s] = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];
}

__global__ __launch_bounds__(64) void k_vdv_finish(const double* __restrict__ slab, int nb, int S, double* __restrict__ out) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= S) return;
    double acc = 0.0;
    for (int b = 0; b < nb; b += 8) {  // (eight loads in flight, the same order of additions: 256 round trips were 0.1 ms)
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = slab[(int64_t)(b + q < nb ? b + q : b) * S + s];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (b + q < nb) acc += v[q];
    }
    out[s] = acc;
}

int vdv_cols_grid(int64_t N) {
    int64_t want = (N + 4 * 8 - 1) / (4 * 8);
    if (want > 256) want = 256;
    return (int)(want < 1 ? 1 : want);
}

// slab: vdv_cols_grid(N) * S doubles
hipError_t launch_vdv_cols(const double* V, const double* D, const unsigned short* D16, int SD, int64_t N, int S, double* slab,
                           double* out, hipStream_t st) {
    const int nbx = vdv_cols_grid(N);
    const dim3 grid(nbx, (S + 63) / 64);
    if (D16 != nullptr) hipLaunchKernelGGL((k_vdv_cols<true>), grid, dim3(256), 0, st, V, (const void*)D16, SD, N, S, slab);
    else hipLaunchKernelGGL((k_vdv_cols<false>), grid, dim3(256), 0, st, V, (const void*)D, S, N, S, slab);
    hipLaunchKernelGGL(k_vdv_finish, dim3((S + 63) / 64), dim3(64), 0, st, slab, nbx, S, out);
    return hipGetLastError();
}

bool cost_cols_supported(int S, int n_c, int n_u) { return n_c <= 16 && n_u >= 0 && n_u <= 4 && n_c + n_u >= 1; }

template <int NKC, int NU>
static hipError_t launch_cost_cols_t(const double* V, const double* D, const unsigned short* D16, int SD, const double* Rtp,
                                     const double* u, const double* alpha, int64_t N, int S, int n_c, double* scratch,
                                     double* out, hipStream_t st) {
    const int ny = (S + 63) / 64;
    int64_t want = (N + 4 * 8 - 1) / (4 * 8);
    int nbx = (int)(want < 1 ? 1 : want);
    const int cap = 1024 / ny;  // scratch: 1024 partials
    if (nbx > cap) nbx = cap;
    if (D16 != nullptr && S >= 128 && SD % 2 == 0 && (reinterpret_cast<uintptr_t>(V) & 7) == 0) {
        const int ny2 = (S + 127) / 128;
        nbx = (int)(want < 1 ? 1 : want);
        if (nbx > 1024 / ny2) nbx = 1024 / ny2;
        if (S & 1)
            hipLaunchKernelGGL((k_cost_cols2<NKC, NU, true>), dim3(nbx, ny2), dim3(256), 0, st, V, D16, SD, Rtp, u, alpha, N, S,
                               n_c, scratch);
        else
            hipLaunchKernelGGL((k_cost_cols2<NKC, NU, false>), dim3(nbx, ny2), dim3(256), 0, st, V, D16, SD, Rtp, u, alpha, N, S,
                               n_c, scratch);
        hipLaunchKernelGGL(k_reduce_final<1>, dim3(1), dim3(256), 0, st, scratch, nbx * ny2, out, (const int*)nullptr);
        return hipGetLastError();
    }
    if (D16 != nullptr)
        hipLaunchKernelGGL((k_cost_cols<NKC, NU, true>), dim3(nbx, ny), dim3(256), 0, st, V, (const void*)D16, SD, Rtp, u,
                           alpha, N, S, n_c, scratch);
    else
        hipLaunchKernelGGL((k_cost_cols<NKC, NU, false>), dim3(nbx, ny), dim3(256), 0, st, V, (const void*)D, S, Rtp, u, alpha,
                           N, S, n_c, scratch);
    hipLaunchKernelGGL(k_reduce_final<1>, dim3(1), dim3(256), 0, st, scratch, nbx * ny, out, (const int*)nullptr);
    return hipGetLastError();
}

template <int NKC>
static hipError_t launch_cost_cols_nkc(int n_u, const double* V, const double* D, const unsigned short* D16, int SD,
                                       const double* Rtp, const double* u, const double* alpha, int64_t N, int S, int n_c,
                                       double* scratch, double* out, hipStream_t st) {
    switch (n_u) {
#define DMF_CASE(NU_) \
    case NU_: return launch_cost_cols_t<NKC, NU_>(V, D, D16, SD, Rtp, u, alpha, N, S, n_c, scratch, out, st);
        DMF_CASE(0) DMF_CASE(1) DMF_CASE(2) DMF_CASE(3) DMF_CASE(4)
#undef DMF_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_cost_cols(const double* V, const double* D, const unsigned short* D16, int SD, const double* Rtp,
                            const double* u, const double* alpha, int64_t N, int S, int n_c, int n_u, double* scratch,
                            double* out, hipStream_t st) {
    switch ((n_c + 3) / 4) {
#define DMF_NKC(X) \
    case X: return launch_cost_cols_nkc<X>(n_u, V, D, D16, SD, Rtp, u, alpha, N, S, n_c, scratch, out, st);
        DMF_NKC(0) DMF_NKC(1) DMF_NKC(2) DMF_NKC(3) DMF_NKC(4)
#undef DMF_NKC
        default: return hipErrorInvalidValue;
    }
}

// Wide row groups (5..16 unknowns): the two-samples-per-lane form only (u16 counts, S even and >= 128, 16-B aligned V);
// other shapes of that width stay on the generic k_cost.
bool cost_cols2_wide_supported(const double* V, const unsigned short* D16, int S, int SD, int n_c, int n_u) {
    // (below 128 samples part of the lanes idle; what competes is the any-shape k_cost, slower from ~32 samples on)
    return D16 != nullptr && n_c <= 16 && n_u >= 5 && n_u <= 16 && S >= 32 && SD % 2 == 0 &&
           (reinterpret_cast<uintptr_t>(V) & 7) == 0;
}

template <int NKC, int NU>
static hipError_t launch_cost_cols2_wide_t(const double* V, const unsigned short* D16, int SD, const double* Rtp,
                                           const double* u, const double* alpha, int64_t N, int S, int n_c, double* scratch,
                                           double* out, hipStream_t st) {
    const int ny2 = (S + 127) / 128;
    const int64_t want = (N + 4 * 8 - 1) / (4 * 8);
    int nbx = (int)(want < 1 ? 1 : want);
    if (nbx > 1024 / ny2) nbx = 1024 / ny2;  // scratch: 1024 partials
    if (S & 1)
        hipLaunchKernelGGL((k_cost_cols2<NKC, NU, true>), dim3(nbx, ny2), dim3(256), 0, st, V, D16, SD, Rtp, u, alpha, N, S, n_c,
                           scratch);
    else
        hipLaunchKernelGGL((k_cost_cols2<NKC, NU, false>), dim3(nbx, ny2), dim3(256), 0, st, V, D16, SD, Rtp, u, alpha, N, S, n_c,
                           scratch);
    hipLaunchKernelGGL(k_reduce_final<1>, dim3(1), dim3(256), 0, st, scratch, nbx * ny2, out, (const int*)nullptr);
    return hipGetLastError();
}

template <int NKC>
static hipError_t launch_cost_cols2_wide_nkc(int n_u, const double* V, const unsigned short* D16, int SD, const double* Rtp,
                                             const double* u, const double* alpha, int64_t N, int S, int n_c,
                                             double* scratch, double* out, hipStream_t st) {
    switch (n_u) {
#define DMF_CASE(NU_) \
    case NU_: return launch_cost_cols2_wide_t<NKC, NU_>(V, D16, SD, Rtp, u, alpha, N, S, n_c, scratch, out, st);
        DMF_CASE(5) DMF_CASE(6) DMF_CASE(7) DMF_CASE(8) DMF_CASE(9) DMF_CASE(10) DMF_CASE(11) DMF_CASE(12) DMF_CASE(13)
        DMF_CASE(14) DMF_CASE(15) DMF_CASE(16)
#undef DMF_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_cost_cols2_wide(const double* V, const unsigned short* D16, int SD, const double* Rtp, const double* u,
                                  const double* alpha, int64_t N, int S, int n_c, int n_u, double* scratch, double* out,
                                  hipStream_t st) {
    if (!cost_cols2_wide_supported(V, D16, S, SD, n_c, n_u)) return hipErrorInvalidValue;
    switch ((n_c + 3) / 4) {
#define DMF_NKC(X) \
    case X: return launch_cost_cols2_wide_nkc<X>(n_u, V, D16, SD, Rtp, u, alpha, N, S, n_c, scratch, out, st);
        DMF_NKC(0) DMF_NKC(1) DMF_NKC(2) DMF_NKC(3) DMF_NKC(4)
#undef DMF_NKC
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_cost(const double* V, const double* D, const double* Rt, const double* u,
                       const double* alpha, int64_t N, int S, int n_c, int n_u,
                       double* scratch, double* out, hipStream_t st) {
    const int K = n_c + n_u;
    const int tpr = S < 256 ? S : 256;
    const int rows_per_tile = 256 / tpr;
    int64_t tiles = (N + rows_per_tile - 1) / rows_per_tile;
    const int nb = (int)(tiles < 1024 ? (tiles < 1 ? 1 : tiles) : 1024);
    const size_t lds = (size_t)K * S * sizeof(double);
    const int in_lds = lds <= 48 * 1024;
    hipLaunchKernelGGL(k_cost, dim3(nb), dim3(256), in_lds ? lds : 0, st, V, D, Rt, u, alpha, N, S,
                       n_c, n_u, in_lds, scratch);
    hipLaunchKernelGGL(k_reduce_final<1>, dim3(1), dim3(256), 0, st, scratch, nb, out,
                       (const int*)nullptr);
    return hipGetLastError();
}

// ------------------------------------------------------------------ generic weighted Gram
// Extended row vector x_i(s) = (Rt_i0..Rt_i,nc-1, u_i0..u_i,nu-1, v_is).  Accumulator a with
// indices (k, l) holds sum_i d_is x_ik x_il for every sample s: (k,l < K) a Gram entry,
// (k < K, l = K) an entry of b_s = R^T (d_s * v_s), (K, K) the constant v_s^T D_s v_s.
static void gram_geometry(int64_t N, int S, int n_jobs, int* nsx, int* nz, int* ny,
                          int64_t* rows_per_chunk) {
    *nsx = (S + 63) / 64;
    *nz = (n_jobs + kGramChunk - 1) / kGramChunk;
    int64_t want = 4096 / ((int64_t)(*nsx) * (*nz));
    if (want < 1) want = 1;
    int64_t rpc = (N + want - 1) / want;
    if (rpc < 256) rpc = 256;
    *rows_per_chunk = rpc;
    *ny = (int)((N + rpc - 1) / rpc);
    if (*ny < 1) *ny = 1;
}

int64_t gram_slab_doubles(int64_t N, int S, int n_jobs) {
    int nsx, nz, ny;
    int64_t rpc;
    gram_geometry(N, S, n_jobs, &nsx, &nz, &ny, &rpc);
    return (int64_t)ny * n_jobs * S;
}

__device__ __forceinline__ double ext_row_value(const double* __restrict__ Rt,
                                                const double* __restrict__ u, int64_t i, int n_c,
                                                int n_u, int k, double v) {
    if (k < n_c) return Rt[i * n_c + k];
    if (k < n_c + n_u) return u[i * n_u + (k - n_c)];
    return v;
}

__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ V, const double* __restrict__ D,
                                              const double* __restrict__ Rt, const double* __restrict__ u,
                                              int64_t N, int S, int n_c, int n_u,
                                              const short* __restrict__ k_idx,
                                              const short* __restrict__ l_idx, int n_jobs,
                                              int64_t rows_per_chunk, double* __restrict__ slab,
                                              const int* __restrict__ done_flag) {
    __shared__ double red[3][kGramChunk][64];
    if (done_flag != nullptr && *done_flag) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 64 + lane;
    const bool active = s < S;
    const int a0 = blockIdx.z * kGramChunk;
    int kk[kGramChunk], ll[kGramChunk];
#pragma unroll
    for (int p = 0; p < kGramChunk; ++p) {
        const int a = a0 + p < n_jobs ? a0 + p : n_jobs - 1;  // clamp: padded slots repeat the last job
        kk[p] = k_idx[a];
        ll[p] = l_idx[a];
    }
    double acc[kGramChunk];
#pragma unroll
    for (int p = 0; p < kGramChunk; ++p) acc[p] = 0.0;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < N ? r0 + rows_per_chunk : N;
    for (int64_t i = r0 + wave; i < r1; i += 4) {
        const double d = active ? D[i * S + s] : 0.0;
        const double v = active ? V[i * S + s] : 0.0;
#pragma unroll
        for (int p = 0; p < kGramChunk; ++p) {
            const double xk = ext_row_value(Rt, u, i, n_c, n_u, kk[p], v);
            const double xl = ext_row_value(Rt, u, i, n_c, n_u, ll[p], v);
            acc[p] = fma(d * xk, xl, acc[p]);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int p = 0; p < kGramChunk; ++p) red[wave - 1][p][lane] = acc[p];
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
        for (int p = 0; p < kGramChunk; ++p) {
            if (a0 + p < n_jobs) {
                const double tot = ((acc[p] + red[0][p][lane]) + red[1][p][lane]) + red[2][p][lane];
                slab[((int64_t)blockIdx.y * n_jobs + (a0 + p)) * S + s] = tot;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_gram_reduce(const double* __restrict__ slab, int ny,
                                                     int n_jobs, int S,
                                                     const int* __restrict__ dst_row,
                                                     double* __restrict__ gb,
                                                     const int* __restrict__ done_flag) {
    // block = 32 sample columns x 8 slab lanes; lane yl sums slabs yl, yl + 8, ... (4 loads in flight),
    // then the 8 partial sums are added in fixed order: deterministic.
    __shared__ double part[8][32];
    if (done_flag != nullptr && *done_flag) return;
    const int c = threadIdx.x & 31, yl = threadIdx.x >> 5;
    const int s = blockIdx.x * 32 + c;
    const int a = blockIdx.y;
    const bool ok = s < S;
    const int sc = ok ? s : S - 1;
    const double* __restrict__ src = slab + (int64_t)a * S + sc;
    const int64_t ystride = (int64_t)n_jobs * S;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    int y = yl;
    for (; y + 24 < ny; y += 32) {
        t0 += src[(int64_t)y * ystride];
        t1 += src[(int64_t)(y + 8) * ystride];
        t2 += src[(int64_t)(y + 16) * ystride];
        t3 += src[(int64_t)(y + 24) * ystride];
    }
    for (; y < ny; y += 8) t0 += src[(int64_t)y * ystride];
    part[yl][c] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (yl == 0 && ok) {
        double tot = part[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot += part[k][c];
        gb[(int64_t)dst_row[a] * S + s] = tot;
    }
}

hipError_t launch_gram_reduce(const double* slab, int ny, int n_jobs, int S, const int* dst_row,
                              double* gb, const int* done_flag, hipStream_t st) {
    hipLaunchKernelGGL(k_gram_reduce, dim3((S + 31) / 32, n_jobs), dim3(256), 0, st, slab, ny, n_jobs, S,
                       dst_row, gb, done_flag);
    return hipGetLastError();
}

hipError_t launch_gram(const double* V, const double* D, const double* Rt, const double* u,
                       int64_t N, int S, int n_c, int n_u, GramJobTable jobs, double* slab,
                       int64_t slab_doubles, double* gb, const int* done_flag, hipStream_t st) {
    if (jobs.count <= 0) return hipSuccess;
    int nsx, nz, ny;
    int64_t rpc;
    gram_geometry(N, S, jobs.count, &nsx, &nz, &ny, &rpc);
    if ((int64_t)ny * jobs.count * S > slab_doubles) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_gram, dim3(nsx, ny, nz), dim3(256), 0, st, V, D, Rt, u, N, S, n_c, n_u,
                       jobs.k_idx, jobs.l_idx, jobs.count, rpc, slab, done_flag);
    hipLaunchKernelGGL(k_gram_reduce, dim3((S + 31) / 32, jobs.count), dim3(256), 0, st, slab, ny,
                       jobs.count, S, jobs.dst_row, gb, done_flag);
    return hipGetLastError();
}

}  // namespace dmf
