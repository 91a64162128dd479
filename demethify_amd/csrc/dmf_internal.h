// Internal declarations shared by the kernel translation units and the C-ABI layer.
// gfx950 only; no portability macros on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmf {

// Device-resident scalar state of one solve (demethify/deconvolution.py:192-204 and the
// scalars carried across outer iterations, :206-221).
struct SolverState {
    double a1;         // momentum scalar of the u phase (:192, :83-84)
    double a2;         // momentum scalar of the alpha phase (:193, :95-96)
    double l_w;        // ||alpha[-n_u:]||_F^2 * d (:198, :216)
    double l_w_prev;   // l_w_ (:199, :89)
    double l_h;        // ||R||_F^2 * d (:201, :212)
    double l_h_prev;   // l_h_ (:202, :101)
    double dsq;        // d = max(D)^2 (:197)
    double rt_norm2;   // ||R_trunc||_F^2 (constant part of ||R||_F^2)
    double u_norm2;    // ||u||_F^2 of the current u
    double cf;         // current cost (:204, :218)
    double cf_prev;    // cf_0 (:207)
    double tol;        // stop threshold of the running step() call (:220)
    long long iters;   // outer iterations completed
    double band;       // 1: |cf - cf_0| < tol stops; > 1: |cf - cf_0| < band x tol pauses for the host's confirmation
    int done;          // 1 once |cf - cf_0| < tol was met, 2 while paused: later launches are no-ops
    int arrive;        // workgroups of the alpha kernel that have finished this outer iteration (the last one closes it)
    // Momentum rows (dmf_solver_step).  The sequence a_t of deconvolution.py:83-84 / :95-96 does not depend on the data, so
    // the host runs it ahead for the iterations it enqueues and uploads one row per outer iteration:
    //   { a1 after the iteration, a2 after it, (a_{t-1} - 1) / a_t for the u phase [mom_n], the same for the alpha phase }
    // -- no kernel then spends its first microseconds on a chain of n_iter2 square roots and divisions (the row pass's
    // thread 0, every wave of the alpha kernel and the closing step each did).  Row mom_i is the current iteration's
    // (the closing step moves on); valid for launches with n_iter2 == mom_n only.
    const double* mom;
    int mom_stride, mom_rows, mom_i, mom_n;
};

// the row of momentum ratios of the current outer iteration, or nullptr (the kernel then runs the recurrence itself)
__device__ __forceinline__ const double* momentum_row(const SolverState* __restrict__ state, int n_iter2) {
    return (state->mom != nullptr && state->mom_n == n_iter2 && state->mom_i < state->mom_rows)
               ? state->mom + (int64_t)state->mom_i * state->mom_stride
               : nullptr;
}

// Packed upper triangle, column-major over (k <= l): independent of the matrix size.
__host__ __device__ inline int tri(int k, int l) { return l * (l + 1) / 2 + k; }

constexpr int kMaxK = 64;          // largest K = n_c + n_u the alpha kernels are built for
constexpr int kGramChunk = 16;     // accumulators per generic Gram job
constexpr int kRowsPerBlockU = 64; // rows handled by one block of the u-phase kernels

struct GramJobTable {              // device arrays, one entry per accumulator
    const short* k_idx;            // extended index (0..K; K means "the sample column v")
    const short* l_idx;
    const int* dst_row;            // destination row in the solver's packed Gram buffer
    int count;
};

// One percentile of numpy's "linear" method over n sorted values: lerp(sorted[k_prev], sorted[k_next], gamma)
struct PercentilePlan {
    long long k_prev, k_next;
    double gamma;
    int from_top;  // set by the launcher: which register tail of k_percentile_tails holds the two values
    int pad;
};

// ---- launch wrappers (dmf_kernels_*.hip) ---------------------------------------------------
// All wrappers enqueue on `st` and return the hipGetLastError() of their launches.

hipError_t launch_convert_counts(const long long* src, double* dst, int64_t n, hipStream_t st);
// max over a f64 array -> *out (device); scratch needs >= 1024 doubles
hipError_t launch_max_f64(const double* x, int64_t n, double* scratch, double* out, hipStream_t st);
// max |x - (double)(float)x| -> *out: 0 iff every element is exactly representable in f32
hipError_t launch_f32_residual_max(const double* x, int64_t n, double* scratch, double* out, hipStream_t st);
// max(x) if every element is a non-negative integer, +inf otherwise -> *out
hipError_t launch_int_count_max(const double* x, int64_t n, double* scratch, double* out, hipStream_t st);
// 0 if every element lies in [0, 1], 1 otherwise -> *out
hipError_t launch_unit_range_check(const double* x, int64_t n, double* scratch, double* out, hipStream_t st);
// sum of squares -> *out (device)
hipError_t launch_sumsq_f64(const double* x, int64_t n, double* scratch, double* out,
                            const int* done_flag, hipStream_t st);
hipError_t launch_pad_rows(const double* src, double* dst, int64_t n_rows, int width_src, int width_dst,
                           hipStream_t st);
hipError_t launch_index_range_check(const long long* idx, int64_t n_idx, int64_t n_src, unsigned int* flag, hipStream_t st);
hipError_t launch_gather_rows(const double* src, double* dst, const long long* idx, int64_t n_idx,
                              int64_t width, hipStream_t st);

// direct weighted cost: *out = sum d (v - [Rt|u] alpha)^2
hipError_t launch_cost(const double* V, const double* D, const double* Rt, const double* u,
                       const double* alpha, int64_t N, int S, int n_c, int n_u,
                       double* scratch, double* out, hipStream_t st);

// the same cost for n_c <= 16, n_u <= 4 with the lane's alpha column in registers: Rtp = padded R_trunc copy,
// D16 (u16 counts, row stride SD) is read instead of D when it is not null
bool cost_cols_supported(int S, int n_c, int n_u);
bool cost_cols2_wide_supported(const double* V, const unsigned short* D16, int S, int SD, int n_c, int n_u);
hipError_t launch_cost_cols2_wide(const double* V, const unsigned short* D16, int SD, const double* Rtp, const double* u,
                                  const double* alpha, int64_t N, int S, int n_c, int n_u, double* scratch, double* out,
                                  hipStream_t st);
int vdv_cols_grid(int64_t N);
hipError_t launch_vdv_cols(const double* V, const double* D, const unsigned short* D16, int SD, int64_t N, int S, double* slab,
                           double* out, hipStream_t st);
hipError_t launch_cost_cols(const double* V, const double* D, const unsigned short* D16, int SD, const double* Rtp,
                            const double* u, const double* alpha, int64_t N, int S, int n_c, int n_u, double* scratch,
                            double* out, hipStream_t st);

// generic weighted Gram accumulation over the extended row vector x = (Rt, u, v)
hipError_t launch_gram(const double* V, const double* D, const double* Rt, const double* u,
                       int64_t N, int S, int n_c, int n_u, GramJobTable jobs,
                       double* slab, int64_t slab_doubles, double* gb, const int* done_flag,
                       hipStream_t st);
int64_t gram_slab_doubles(int64_t N, int S, int n_jobs);
hipError_t launch_gram_reduce(const double* slab, int ny, int n_jobs, int S, const int* dst_row,
                              double* gb, const int* done_flag, hipStream_t st);
// shape-specialised one-pass Gram of the u-dependent entries (n_u <= 8, n_c <= 16); slab in job order.
// Rtp = R_trunc with rows zero-padded to a multiple of 4 doubles (dmf_problem::Rtp).
bool gram_u_supported(int n_c, int n_u);
int64_t gram_u_slab_doubles(int64_t N, int S, int n_c, int n_u);
hipError_t launch_gram_u(const double* V, const double* D, const double* Rtp, const double* u, int64_t N,
                         int S, int n_c, int n_u, double* slab, const int* done_flag, int* ny_out,
                         hipStream_t st);

// u phase, Gram form (n_u <= 8): all n_iter2 inner iterations in one launch
hipError_t launch_u_phase_gram(const double* V, const double* D, const double* Rt,
                               const double* alpha, double* u, double* u_prev,
                               const SolverState* state, int64_t N, int S, int n_c, int n_u,
                               int n_iter2, int mode, hipStream_t st);
bool u_phase_gram_supported(int S, int n_c, int n_u);
// u phase on the FP64 matrix cores (n_u <= 8, n_c <= 16, S <= 512); takes the padded Rtp as well
bool u_phase_mfma_supported(int S, int n_c, int n_u);
hipError_t launch_u_phase_mfma(const double* V, const double* D, const unsigned short* D16, int SD, const double* Rtp,
                               const double* alpha, double* u, double* u_prev, const SolverState* state, int64_t N, int S,
                               int n_c, int n_u, int n_iter2, int mode, hipStream_t st);
// fused row pass: u phase + u-dependent Gram slab + ||u||^2 / l_h in one read of V and D
// (S % 4 == 0, S <= 256, n_c <= 16, n_u <= 8, accumulators <= 80, counts exact in f32);
// grid_out = workgroups launched; the slab holds 2 rows per workgroup
bool rowpass_fused_supported(int S, int n_c, int n_u);
int rowpass_fused_grid(int64_t N, int S);
int64_t rowpass_fused_slab_doubles(int64_t N, int S, int n_c, int n_u);
hipError_t launch_rowpass_fused(const double* V, const double* D, const double* Rtp, const double* alpha,
                                double* u, double* u_prev, SolverState* state, int64_t N, int S, int n_c,
                                int n_u, int n_iter2, int mode, double* slab, double* u2_partials,
                                int* grid_out, hipStream_t st);
// sum of the fused kernel's per-workgroup ||u||^2 shares -> state->u_norm2 and l_h (deconvolution.py:212)
hipError_t launch_finish_u_norm(const double* u2_partials, int n, SolverState* state, hipStream_t st);
// u phase, schedule-faithful fallback: ONE inner iteration (index t) per launch
hipError_t launch_u_step_direct(const double* V, const double* D, const double* Rt,
                                const double* alpha, const double* u_cur, const double* u_prev,
                                double* u_next, const SolverState* state, int64_t N, int S,
                                int n_c, int n_u, int t, int mode, hipStream_t st);
bool u_step_direct_supported(int S, int n_c, int n_u);

// alpha phase on the packed Gram buffer gb[(K+1)(K+2)/2][S]
// thread_per_sample selects the one-thread-per-sample kernels (test levels 1 and 2) instead of the
// lane-parallel one (G lanes per sample); partials must hold 2 * (ceil(S / 64) + S) doubles
hipError_t launch_alpha_phase(const double* gb, double* alpha, double* alpha_prev,
                              SolverState* state, int S, int n_c, int n_u, int n_iter2,
                              double* partials, bool thread_per_sample, hipStream_t st);
hipError_t launch_set_lh(SolverState* state, hipStream_t st);
// purity-constrained alpha phase (Frank-Wolfe, deconvolution.py:280-302) on the same packed Gram buffer
hipError_t launch_alpha_frank_wolfe(const double* gb, double* alpha, const double* purity, SolverState* state,
                                    int S, int n_c, int n_u, int max_iter, double* partials, hipStream_t st);
hipError_t launch_project_simplex(const double* X, double* out, int K, int S, double z,
                                  hipStream_t st);
hipError_t launch_scatter_known_block(const double* gb_known, double* gb, int n_c, int K, int S,
                                      hipStream_t st);
hipError_t launch_init_state(SolverState* state, const double* consts, const double* alpha,
                             int S, int n_c, int n_u, hipStream_t st);

// the same u phase in two launches for many inner steps: c_i / M_i per row to `cm`, then inner iterations with
// every lane busy; cm holds u_phase_split_cm_doubles(N, n_u) doubles, beta n_iter2 doubles (<= 6144)
int64_t u_phase_split_cm_doubles(int64_t N, int n_u);
hipError_t launch_u_phase_split(const double* V, const double* D, const unsigned short* D16, int SD, const double* Rtp,
                                const double* alpha, double* u, double* u_prev, const SolverState* state, int64_t N, int S,
                                int n_c, int n_u, int n_iter2, int mode, double* cm, double* beta, hipStream_t st);

// split u phase whose producer runs M_i on the integer matrix cores (dmf_kernels_cm_i8.hip): n_u <= 16, 2 <= S <= 2048
// (panels of 256 samples), u16 counts with ND digit planes, alpha in [0, 1]
bool cm_i8_supported(const double* V, int S, int n_c, int n_u, int ND, int SD);
hipError_t launch_cm_i8(const double* V, const unsigned short* D16, int SD, int ND, const double* Rtp, const double* alpha,
                        const SolverState* state, int64_t N, int S, int n_c, int n_u, double* cm, hipStream_t st);
hipError_t launch_u_phase_split_i8(const double* V, const unsigned short* D16, int SD, int ND, const double* Rtp,
                                   const double* alpha, double* u, double* u_prev, const SolverState* state, int64_t N,
                                   int S, int n_c, int n_u, int n_iter2, int mode, double* cm, double* beta,
                                   hipStream_t st);

// ... and with the inner iterations fused with the b_u stream of the integer Gram route (k_inner_bu): slab holds
// u_inner_bu_grid(N, S) x n_u x S doubles, u2_partials one double per workgroup (their count comes back in grid_out)
bool u_inner_bu_supported(const double* V, int S, int SD, int n_u, int n_iter2);
int u_inner_bu_grid(int64_t N, int S);
hipError_t launch_u_phase_split_i8_bu(const double* V, const unsigned short* D16, int SD, int ND, const double* Rtp,
                                      const double* alpha, double* u, double* u_prev, const SolverState* state, int64_t N,
                                      int S, int n_c, int n_u, int n_iter2, int mode, double* cm, double* beta,
                                      double* slab, double* u2_partials, int* grid_out, hipStream_t st);

// u phase for 9 <= n_u <= 26 unknown types on the matrix cores (dmf_kernels_rowpass_big.hip); Rtp = padded R_trunc
bool u_phase_big_supported(int S, int n_c, int n_u, int n_iter2);
hipError_t launch_u_phase_big(const double* V, const double* D, const double* Rtp, const double* alpha, double* u,
                              double* u_prev, const SolverState* state, int64_t N, int S, int n_c, int n_u,
                              int n_iter2, int mode, hipStream_t st);

// any-shape Gram accumulation on the matrix cores (dmf_kernels_gram_mfma.hip): jobs [0, n_dense) have l < K,
// the rest are the "v" column; the slab ([ny][count][S]) is then summed by launch_gram_reduce
hipError_t launch_gram_mfma(const double* V, const double* D, const double* Rt, const double* u, int64_t N, int S,
                            int n_c, int n_u, GramJobTable jobs, int n_dense, double* slab, int64_t slab_doubles,
                            const int* done_flag, int* ny_out, hipStream_t st);
int64_t gram_mfma_slab_doubles(int64_t N, int S, int n_jobs);

// ---- second-generation row pass (dmf_kernels_rowpass2.hip) + integer-matrix-core Gram (dmf_kernels_gram_i8.hip)
// counts as u16 (D16[N16][SD], zero padded: N16 = N rounded up to 16, SD = S rounded up to 64) and as balanced 8-bit
// digit planes in the MFMA B layout (Dt8[ND][ceil(N/32)][SD/32][32][32]); ND = 1 (d <= 127) or 2 (d <= 32639)
hipError_t launch_gather_counts_int(const unsigned short* src16, const long long* idx, int64_t n_idx, int SD, int ND,
                                    unsigned short* D16, int64_t N16, signed char* Dt8, int64_t plane_stride,
                                    unsigned int* max_out, hipStream_t st);
hipError_t launch_build_counts_int(const double* D, int64_t N, int S, int ND, unsigned short* D16, int64_t N16, int SD,
                                   signed char* Dt8, int64_t plane_stride, hipStream_t st);
bool rowpass_v2_supported(int S, int n_c, int n_u, int n_iter2);
int rowpass_v2_grid(int64_t N, int S);
// u phase + b_u slab ([grid][n_u][S] doubles) + per-workgroup ||u||^2 shares in one read of V (f64) and D16
hipError_t launch_rowpass_v2(const double* V, const unsigned short* D16, int SD, const double* Rtp, const double* alpha,
                             double* u, double* u_prev, SolverState* state, int64_t N, int S, int n_c, int n_u,
                             int n_iter2, int mode, int nd, double* slab, double* u2_partials, int* grid_out, hipStream_t st);
bool gram_i8_supported(int n_c, int n_u, int ND, int64_t N, int SD);
int64_t gram_i8_slab_words(int64_t N, int SD, int n_c, int n_u);  // i64 words of the slab
int64_t gram_i8_acc_words(int S, int n_c, int n_u);               // i64 words of the reduction scratch (zero-initialised)
// the known block of the packed Gram through the same kernels (features = pairs of R_trunc columns; n_u = 0)
bool gram_i8_known_supported(int n_c, int ND, int64_t N, int SD);
int64_t gram_i8_slab_words_nf(int64_t N, int SD, int nf);
int64_t gram_i8_acc_words_nf(int S, int nf, int n_bu);
// exact cross / uu Gram entries: features p = (feat_a[p], feat_b[p]) over x = (Rt, u), i64 slab [ny][2][slots][SD];
// Rtp = the padded R_trunc copy (rows of 4 ceil(n_c / 4) doubles); Rtp, u, Dt8 16-byte aligned, u allocated to a
// multiple of 16 bytes
hipError_t launch_gram_i8(const signed char* Dt8, int64_t plane_stride, int SD, int ND, const double* Rtp, const double* u,
                          int64_t N, int n_c, int n_u, const short* feat_a, const short* feat_b, int NF, long long* slab,
                          int64_t slab_words, const int* done_flag, int* ny_out, hipStream_t st);
// b_u alone (for u phases that are kernels of their own): slab [n_slabs][n_u][S] doubles, n_u <= 20
int bu_cols_grid(int64_t N);
hipError_t launch_bu_cols(const double* V, const unsigned short* D16, int SD, const double* u, int64_t N, int S, int n_u,
                          double* slab, const int* done_flag, int* n_slabs_out, hipStream_t st, bool* with_vdv = nullptr);
// gb rows of the u-dependent jobs from the i64 slab (jobs < NF) and the row pass's b_u slabs (jobs NF .. NF + n_u);
// acc_words: gram_i8_acc_words() i64 words, all zero before the first call (the kernels leave them zero again);
// u2_partials != null: the finish kernel also sums the row pass's ||u||^2 shares into state (u_norm2, l_h)
hipError_t launch_gram_v2_reduce(const long long* slab_i8, int ny, int NF, int SD, const double* slab_bu, int n_bu_slabs,
                                 int n_u, int S, long long* acc_words, const int* dst_row, double* gb, const int* done_flag,
                                 const double* u2_partials, int n_u2, SolverState* state, hipStream_t st);

// two percentiles over axis 0 of x[n][m] -> out0[m], out1[m] (out1 may be null); dmf_kernels_percentile.hip
hipError_t launch_percentile_pair(const double* x, int64_t n, int64_t m, PercentilePlan p0, PercentilePlan p1,
                                  double* out0, double* out1, hipStream_t st);
int64_t percentile_max_replicates();

}  // namespace dmf
