/*
 * demethify_hip.h — C-ABI of libdemethify_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for DeMethify's solver hot path.  The reference has no FFI layer: its
 * seam is the set of Python callables that demethify/demethify.py:7, demethify/bootstrap.py:6
 * and demethify/ic.py:8 import from demethify/deconvolution.py.  Each entry point below names
 * the reference callable it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns a dmf_status (0 = ok) and never
 *     throws across the boundary; the caller allocates every output buffer.
 *   - matrices are dense, C-order (row-major) float64; counts are int64 or float64 (flag).
 *   - N = CpG rows, S = samples, n_c = known cell types, n_u = unknown, K = n_c + n_u.
 *     V = meth_frequency (N x S), D = d_x / counts (N x S), Rt = R_trunc (N x n_c),
 *     u (N x n_u), alpha (K x S; the LAST n_u rows are the unknown types).
 *   - pointer arguments are host pointers unless the call's `flags` carries
 *     DMF_PTR_DEVICE, in which case they are device pointers on the context's GPU
 *     (e.g. PyTorch-ROCm `tensor.data_ptr()`); device inputs are borrowed, never freed.
 *   - one context per GPU; a context is not thread-safe, independent contexts are.
 *   - inputs are never mutated (reference convention, deconvolution.py:194-195).
 */
#ifndef DEMETHIFY_HIP_H
#define DEMETHIFY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dmf_status {
    DMF_OK = 0,
    DMF_ERR_BAD_ARG = 1,      /* null pointer, non-positive size, n_u < 1 where required ...   */
    DMF_ERR_BAD_SHAPE = 2,    /* shapes inconsistent with the problem the handle was built for  */
    DMF_ERR_HIP = 3,          /* a HIP runtime call failed; see dmf_last_error()                */
    DMF_ERR_NONFINITE = 4,    /* NaN/Inf met in an input that the solver cannot propagate        */
    DMF_ERR_UNSUPPORTED = 5,  /* size beyond what the kernels are built for (K > 64, ...)        */
    DMF_ERR_NO_DEVICE = 6     /* no gfx950 device visible                                        */
} dmf_status;

enum {
    DMF_PTR_DEVICE = 1,       /* data pointers of this call are device pointers                  */
    DMF_COUNTS_F64 = 2,       /* `counts` is float64 (default: int64, as pandas yields)          */
    DMF_INIT_IN_UNIT_RANGE = 4 /* dmf_solver_create: the caller vouches that alpha0 lies inside [0, 1] (columns on the
                                * simplex: every initialiser of deconvolution.py:40-78 and :108-137 yields that), so the
                                * library skips its own check -- a device-to-host round trip for device arrays */
};

/* dmf_select_describe flags */
enum {
    DMF_SELECT_COUNTS_F32_EXACT = 1,   /* every count survives a round trip through f32                  */
    DMF_SELECT_PURITY = 2,             /* a purity vector is set (Frank-Wolfe alpha phase)               */
    DMF_SELECT_ALPHA_OUTSIDE_UNIT = 4, /* the starting alpha does not lie inside [0, 1]                  */
    DMF_SELECT_V_UNALIGNED = 8         /* meth_frequency starts 8 bytes off a 16-byte boundary           */
};

/* solver variants */
enum {
    DMF_MODE_PARTIAL = 0,     /* mdwbssmf_deconv: gradient of u taken at the extrapolated point  */
    DMF_MODE_UNSUPERVISED = 1 /* unsupervised_deconv: gradient taken at the previous iterate     */
};

/* kernel families whose device time a profiling context accumulates (dmf_context_kernel_time) */
enum {
    DMF_KERNEL_ROWPASS = 0,   /* u-phase row pass (the dominant, HBM-streaming kernel)           */
    DMF_KERNEL_GRAM = 1,      /* per-sample weighted Gram accumulation for the alpha phase        */
    DMF_KERNEL_ALPHA = 2,     /* alpha inner loop + cost + scalar bookkeeping                    */
    DMF_KERNEL_COST = 3,      /* streaming weighted cost                                         */
    DMF_KERNEL_FAMILIES = 4
};

typedef struct dmf_context dmf_context;
typedef struct dmf_problem dmf_problem;
typedef struct dmf_solver dmf_solver;

const char* dmf_status_string(int status);
/* Text of the last HIP error seen by this thread's most recent failing call ("" if none). */
const char* dmf_last_error(void);
/* Library/ABI version, bumped when a signature changes. */
int dmf_abi_version(void);

/* ---- context: one per GPU -------------------------------------------------------------- */
/* `stream` may be NULL (the context creates its own hipStream) or a hipStream_t to borrow. */
int dmf_context_create(int device, void* stream, dmf_context** out);
int dmf_context_destroy(dmf_context* ctx);
int dmf_context_synchronize(dmf_context* ctx);
/* Record HIP events around the launches of the kernel families above (costs a sync per read).
 * enabled: 0 = off, 1 = every family, else a mask with bit (1 + family) set, e.g. 2 = DMF_KERNEL_ROWPASS only. */
int dmf_context_set_profiling(dmf_context* ctx, int enabled);
int dmf_context_kernel_time(dmf_context* ctx, int family, double* total_ms, int64_t* launches);
int dmf_context_reset_kernel_time(dmf_context* ctx);
/* Kernel selection, for tests: 0 = fastest available (row pass on u16 counts + exact integer-matrix-core Gram,
 * else the first-generation fused FP64 row pass, else the unfused pair), 1 = any-shape Gram-form kernels without
 * MFMA, 2 = schedule-faithful one-launch-per-inner-step, 3 = the unfused pair (FP64-MFMA u-phase row pass + one-pass
 * Gram), 4 = the first-generation fused FP64 row pass (level 0's fall-back for counts beyond 32639 or reference
 * profiles outside [0, 1]).  Set it before creating problems: the integer count copies are built at level 0 only. */
int dmf_context_set_generic(dmf_context* ctx, int level);
/* How dmf_solver_step decides |cf - cf_0| < tol (deconvolution.py:218-220) for the solvers of this context:
 * 0 (default) = on the Gram-form cost of the loop, with the decisions near the threshold confirmed on the streaming
 * cost of deconvolution.py:15-17 where the Gram form's error bound (1e-15 N S max(counts)) reaches tol / 20;
 * 1 = every decision near the threshold (below 10 tol) is taken on streaming costs, whatever the bound;
 * 2 = Gram form only.  See dmf_solver_stop_info. */
int dmf_context_set_stop_confirmation(dmf_context* ctx, int mode);

/* ---- problem: V, D, Rt resident in HBM + the per-problem constants ----------------------
 * Replaces the (meth_frequency, d_x, R_trunc) argument triple every reference solver call
 * takes (deconvolution.py:40,81,93,107,190).  n_c may be 0 (Rt ignored: unsupervised).
 * Uploads (or borrows) the arrays, converts int64 counts to f64 once and precomputes
 * max(D)^2 (deconvolution.py:197), ||Rt||_F^2 and the known-type blocks of the per-sample
 * Gram matrices. */
int dmf_problem_create(dmf_context* ctx, int64_t N, int64_t S, int64_t n_c,
                       const double* V, const void* counts, const double* Rt,
                       int flags, dmf_problem** out);
/* Row-gathered copy for one bootstrap resample: rows idx[0..N) of V, D, Rt
 * (bootstrap.py:28, sklearn.utils.resample applied to the three arrays).  idx: host int64. */
int dmf_problem_gather(dmf_context* ctx, const dmf_problem* src, const int64_t* idx,
                       int64_t n_idx, dmf_problem** out);
/* The same with the row indices already in HBM (e.g. uploaded by dmf_stage_upload from the thread that drew them, beside
 * the previous replicate's solve): range-checked on the device, DMF_ERR_BAD_ARG when one lies outside [0, N). */
int dmf_problem_gather_device(dmf_context* ctx, const dmf_problem* src, const int64_t* idx_dev,
                              int64_t n_idx, dmf_problem** out);
int dmf_problem_destroy(dmf_problem* p);
int dmf_problem_shape(const dmf_problem* p, int64_t* N, int64_t* S, int64_t* n_c);

/* ---- single-function entry points (KAT parity of SURVEY.md section 8a rows 1-4) --------- */
/* cost_f_w(y, R, alpha, d_x), deconvolution.py:15-17, with R = [Rt | u]. */
int dmf_cost(dmf_context* ctx, const dmf_problem* p, const double* u, int64_t n_u,
             const double* alpha, int flags, double* out_cost);
/* projection_simplex_sort_2d(v, z), deconvolution.py:21-37; X and out are K x S. */
int dmf_project_simplex(dmf_context* ctx, const double* X, int64_t K, int64_t S, double z,
                        int flags, double* out);
/* update_u(u, alpha, n_iter2, a1, l_w_, l_w, u_, meth_frequency, R_trunc, n_u, d_x),
 * deconvolution.py:81-90 -> (u, u_, a1, l_w_).  mode selects the gradient point (row 6 of
 * SURVEY.md section 8a).  scalars_io = {a1, l_w_prev, l_w} in, {a1, l_w_prev, l_w} out. */
int dmf_update_u(dmf_context* ctx, const dmf_problem* p, const double* u, const double* u_prev,
                 const double* alpha, int64_t n_u, int64_t n_iter2, int mode, int flags,
                 double* scalars_io, double* out_u, double* out_u_prev);
/* update_alpha(n_iter2, alpha, a2, l_h_, l_h, alpha_, R, d_x, meth_frequency),
 * deconvolution.py:93-102 -> (alpha, alpha_, a2, l_h_), with R = [Rt | u].
 * scalars_io = {a2, l_h_prev, l_h}. */
int dmf_update_alpha(dmf_context* ctx, const dmf_problem* p, const double* u, int64_t n_u,
                     const double* alpha, const double* alpha_prev, int64_t n_iter2, int flags,
                     double* scalars_io, double* out_alpha, double* out_alpha_prev);

/* Bootstrap post-processing (bootstrap.py:51-54 proportions, :75-78 profile estimates):
 * np.percentile(x, q, axis=0) with numpy's default "linear" method, for x = [n replicates][m positions]
 * (C order), q = n_q percentiles in [0, 100]; out = [n_q][m].  Bit-identical to numpy for finite inputs
 * (NaNs are not ordered: inputs are proportions / methylation levels, never NaN).  n <= 19456. */
int dmf_percentile_axis0(dmf_context* ctx, const double* x, int64_t n, int64_t m, const double* q,
                         int64_t n_q, int flags, double* out);

/* ---- solver: the outer loop, resident on the device --------------------------------------
 * mdwbssmf_deconv (deconvolution.py:190-223) / unsupervised_deconv's loop (:139-184).
 * create = state init (:192-204); step = up to n_outer outer iterations, stopping early when
 * |cf - cf_0| < tol (:220); get = copy out the current (u, alpha). */
int dmf_solver_create(dmf_context* ctx, const dmf_problem* p, const double* u0,
                      const double* alpha0, int64_t n_u, int mode, int flags, dmf_solver** out);
/* Purity-constrained variant (mdwbssmf_deconv_p, deconvolution.py:306-337): after this call the alpha
 * phase of every step is Frank-Wolfe (frank_wolfe_nmf, :280-302) with the known block of sample s held at
 * mass purity[s] and the unknown block at 1 - purity[s]; n_iter2 of dmf_solver_step is then also the
 * number of Frank-Wolfe iterations.  purity: S doubles.  Partial-reference mode only. */
int dmf_solver_set_purity(dmf_solver* s, const double* purity, int flags);
int dmf_solver_step(dmf_solver* s, int64_t n_outer, int64_t n_iter2, double tol,
                    int64_t* iters_done_total, int* converged);
int dmf_solver_get(dmf_solver* s, int flags, double* out_u, double* out_alpha,
                   double* out_cost, int64_t* out_iters);
/* cost_f_w(meth_f, [Rt | u], alpha, counts) of the solver's CURRENT iterate by the streaming formula
 * (deconvolution.py:15-17), without moving u / alpha to the host: what the restart and model-selection loops
 * recompute after every solve (demethify.py:169,199; ic.py:206).  out_cost: host double. */
int dmf_solver_cost(dmf_solver* s, double* out_cost);
/* The same in two halves, for loops that run solve after solve (demethify.py:165-171,195-201; bootstrap.py:26; ic.py:192):
 * _begin enqueues the cost of the current iterate and returns at once, _end waits for it -- in between the caller sets up
 * (and may start stepping) its NEXT solver on the same context, so the GPU works on the cost while the host prepares.
 * One cost in flight per solver; the iterate must not be stepped between the two calls. */
int dmf_solver_cost_begin(dmf_solver* s);
int dmf_solver_cost_end(dmf_solver* s, double* out_cost);
int dmf_solver_destroy(dmf_solver* s);
/* Which kernels a step with n_iter2 inner iterations would launch for this solver, as text, e.g.
 * "rowpass=k_rowpass_fused<3,4> nw=4 grid=256 tail=5 gram=fused alpha=k_alpha_phase_row16".  For tests (every
 * parity case asserts the path it means to cover) and for bench.py's kernel label.  buf gets at most cap bytes
 * including the terminator. */
int dmf_solver_describe(const dmf_solver* s, int64_t n_iter2, char* buf, int64_t cap);
/* The same text for a shape that need not exist on a device: the kernel-selection table itself (csrc/dmf_select.hip),
 * a pure function of (N, S, n_c, n_u, count digit planes nd = 0 | 1 | 2, kernel level, n_iter2, DMF_SELECT_* flags).
 * No GPU is touched: tests/test_host.py enumerates a grid of shapes against tests/golden/kernel_selection.tsv.
 * DMF_ERR_UNSUPPORTED where no kernel takes the shape. */
int dmf_select_describe(int64_t N, int64_t S, int64_t n_c, int64_t n_u, int nd, int level, int64_t n_iter2, int flags,
                        char* buf, int64_t cap);
/* How the stop test |cf - cf_0| < tol (deconvolution.py:218-220) of this solver's dmf_solver_step calls was decided.
 * The loop's cost is the Gram form v^T D v - 2 a.b + a^T G a; where its error bound (it grows with N S max(counts)) is
 * not far below tol, an iteration whose Gram-form difference falls below 10 tol is decided on the STREAMING cost of
 * deconvolution.py:15-17 for this and the previous iterate (confirm_stops = 1; n_confirmed such decisions so far,
 * n_unconfirmed decisions inside that band that had only the Gram form -- the first iteration to enter it);
 * last_stream_cost = the latest streaming cost taken for a stop test (NaN: none).  Any pointer may be NULL. */
int dmf_solver_stop_info(const dmf_solver* s, int* confirm_stops, int64_t* n_confirmed, int64_t* n_unconfirmed,
                         double* last_stream_cost);
/* One-shot convenience: create + step(n_iter1) + get + destroy. */
int dmf_solve(dmf_context* ctx, const dmf_problem* p, const double* u0, const double* alpha0,
              int64_t n_u, int mode, int64_t n_iter1, int64_t n_iter2, double tol, int flags,
              double* out_u, double* out_alpha, double* out_cost, int64_t* out_iters);

/* ---- input tables (host side) ---------------------------------------------------------------
 * demethify/demethify.py:103-143 reads each sample file with pandas.read_csv and stacks its `percent_modified` and
 * `valid_coverage` columns.  dmf_table_scan finds the two columns by header name and counts the data rows;
 * dmf_table_read parses them (n_threads threads) into strided destinations: out_freq[row * stride_freq] =
 * value / divide_by (100 for bedmethyl percentages, 1 for csv fractions), out_cov[row * stride_cov] = coverage
 * (col_valid_coverage < 0: no such column, out_cov untouched).  Values are bit-identical to pandas' default parser; any
 * content that parser treats specially (quotes, NA spellings, blank lines, non-integer coverage) yields
 * DMF_ERR_UNSUPPORTED and the caller reads that file with pandas.  dmf_host_alloc returns page-locked host memory when a
 * GPU runtime is present (*pinned = 1), plain memory otherwise. */
int dmf_table_scan(const char* path, char sep, int64_t* n_rows, int* col_percent_modified, int* col_valid_coverage,
                   int* n_cols);
int dmf_table_read(const char* path, char sep, int col_percent_modified, int col_valid_coverage, int64_t n_rows,
                   double* out_freq, int64_t stride_freq, double divide_by, int64_t* out_cov, int64_t stride_cov,
                   int n_threads);
void* dmf_host_alloc(size_t bytes, int* pinned);
void dmf_host_free(void* p, int pinned);

/* ---- output tables.  The profile confidence intervals go out as the reference writes them (demethify/bootstrap.py:85-91:
 * a DataFrame of (lower, upper) tuples through DataFrame.to_csv): header_line, then per row the n_cols quoted cells
 * "(lo, hi)" -- or "(np.float64(lo), np.float64(hi))" with numpy_scalar_repr != 0, what numpy >= 2 makes of the same
 * tuple -- floats as repr() prints them.  lower / upper are row-major (n_rows x n_cols).  Byte-identical to pandas. */
int dmf_write_interval_csv(const char* path, const char* header_line, const double* lower, const double* upper,
                           int64_t n_rows, int n_cols, int numpy_scalar_repr, int n_threads);

/* ---- restart staging.  The reference draws a fresh (u0, alpha0) per restart and solves it, one after the other
 * (demethify/demethify.py:165-171 and 195-201).  dmf_stage_upload copies a host array (page-locked memory from
 * dmf_host_alloc makes it a direct DMA) to a new device buffer on a copy stream of the context's own and returns when
 * the copy is complete; it may be called from a worker thread while the context's stream iterates the restart before.
 * The buffer is then passed to dmf_solver_create with DMF_PTR_DEVICE and released with dmf_stage_free (ordered behind
 * the context's stream). */
int dmf_stage_upload(dmf_context* ctx, const void* host, size_t bytes, void** out_dev);
int dmf_stage_free(dmf_context* ctx, void* dev);

#ifdef __cplusplus
}
#endif
#endif /* DEMETHIFY_HIP_H */
