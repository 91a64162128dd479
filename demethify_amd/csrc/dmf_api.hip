// C-ABI of libdemethify_hip.so: handles, memory ownership, the outer-loop driver and the
// per-family HIP-event timers.  See include/demethify_hip.h for the contract.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/demethify_hip.h"
#include "dmf_internal.h"
#include "dmf_select.h"

using dmf::SolverState;

namespace {

thread_local char g_last_error[512] = "";

int hip_fail(hipError_t e, const char* what, int line) {
    snprintf(g_last_error, sizeof(g_last_error), "%s failed at dmf_api.hip:%d: %s", what, line,
             hipGetErrorString(e));
    return DMF_ERR_HIP;
}

#define HIP_TRY(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) return hip_fail(e_, #expr, __LINE__);     \
    } while (0)

#define DMF_TRY(expr)                 \
    do {                              \
        int s_ = (expr);              \
        if (s_ != DMF_OK) return s_;  \
    } while (0)

constexpr int kEventPool = 2048;

struct FamilyClock {
    std::vector<hipEvent_t> start, stop;
    int used = 0;
    double total_ms = 0.0;
    int64_t launches = 0;
};

}  // namespace

struct dmf_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    unsigned profiling = 0;  // bit f: record events around the launches of kernel family f
    int stop_confirmation = 0;  // dmf_context_set_stop_confirmation: 0 by error bound, 1 always, 2 never
    int generic_level = 0;  // 0 fused row pass, 1 any-shape Gram-form kernels, 2 schedule-faithful u steps,
                            // 3 separate MFMA row pass + one-pass Gram (the pieces the fused kernel is made of)
    double* scratch = nullptr;  // 4096 doubles of reduction scratch
    hipMemPool_t pool = nullptr;  // the context's own stream-ordered pool (the device's default pool is not touched)
    std::unordered_map<void*, size_t> live;                // large blocks handed out by pool_alloc (size by address)
    std::unordered_map<size_t, std::vector<void*>> kept;    // freed large blocks kept for the next allocation of that size
    size_t kept_bytes = 0;
    std::vector<hipEvent_t> events;       // ... and of their cost events
    std::vector<double*> pinned_moms;    // ... and of their momentum-row staging buffers
    std::vector<void*> pinned_states;   // page-locked SolverState mirrors of destroyed solvers, reused by the next ones
                                        // (hipHostMalloc / hipHostFree cost ~0.1 ms each: a restart loop makes one per restart)
    hipStream_t copy_stream = nullptr;  // dmf_stage_upload: uploads beside the kernels of `stream` (created on first use)
    std::mutex copy_mutex;
    FamilyClock clocks[DMF_KERNEL_FAMILIES];
};

struct dmf_problem {
    dmf_context* ctx = nullptr;
    int64_t N = 0, S = 0, n_c = 0;
    double *V = nullptr, *D = nullptr, *Rt = nullptr;
    bool own_V = false, own_D = false, own_Rt = false;
    double* Rtp = nullptr;       // R_trunc, rows zero-padded to a multiple of 4 doubles (== Rt if n_c % 4 == 0)
    bool own_Rtp = false;
    double* consts = nullptr;    // device {dsq, ||Rt||^2, dmax, max |D - f32(D)|, int-count max or inf, Rt outside [0,1]}
    double h_consts[6] = {0, 0, 0, 0, 0, 0};
    // integer copies of the counts for the second-generation kernels (dmf_kernels_rowpass2.hip, dmf_kernels_gram_i8.hip):
    // built when every count is an integer in [0, 32639], S <= 2048 and R_trunc lies in [0, 1]
    unsigned short* D16 = nullptr;  // [N16][SD], zero padded
    signed char* Dt8 = nullptr;     // [ND][ceil(N / 32)][SD / 32][32][32] balanced 8-bit digits, MFMA B layout
    int ND = 0;                     // count digits: 0 = no integer copies, 1 (d <= 127), 2 (d <= 32639)
    int SD = 0;
    int64_t N16 = 0, plane_stride = 0;
    bool d_f32_exact = false;    // every count survives a round trip through f32 (the fused tile stores D as f32)
    double* gb_known = nullptr;  // [(n_c+1)(n_c+2)/2][S]
};

// page-locked per-solver block: the SolverState mirror, then one double for dmf_solver_cost_begin's result
constexpr size_t kPinnedStateBytes = (sizeof(SolverState) + 15) / 16 * 16 + 16;

struct dmf_solver {
    dmf_context* ctx = nullptr;
    const dmf_problem* p = nullptr;
    int64_t n_u = 0;
    int mode = 0;
    dmf::ShapeKey key;      // what the kernel selection looks at (dmf_select.h) ...
    dmf::PathSpec spec;     // ... and what it fixed for this solver
    double* cm = nullptr;        // split u phase (many inner steps): per-row c_i / M_i, allocated on first use
    double* beta_tab = nullptr;  //   and the momentum coefficients of the inner steps
    int64_t beta_cap = 0;
    long long* slab_i8 = nullptr;   // i64 partial sums of the integer Gram (one slab per row range)
    int64_t slab_i8_words = 0;
    long long* acc_i8 = nullptr;    // reduction scratch of the integer Gram (kept zero between iterations)
    double* purity = nullptr;  // S per-sample known-block masses: set => Frank-Wolfe alpha phase
    double* u2_partials = nullptr;
    double *u = nullptr, *u_prev = nullptr, *u_next = nullptr;
    double *alpha = nullptr, *alpha_prev = nullptr;
    double* gb = nullptr;
    double* slab = nullptr;
    int64_t slab_doubles = 0;
    double* partials = nullptr;
    SolverState* state = nullptr;
    SolverState* h_state = nullptr;  // pinned
    short *job_k = nullptr, *job_l = nullptr;
    int* job_dst = nullptr;
    int n_jobs = 0;
    std::vector<short> h_job_k, h_job_l;  // (the uploads of the job table read these: kept for the solver's life, so that
    std::vector<int> h_job_dst;           //  dmf_solver_create need not wait for them)
    // deconvolution.py:204 -- the cost before the loop is only ever read by the first stop test (:220): it is computed when
    // a step() call with tol > 0 (or a get() before any iteration) needs it, one 0.5 ms pass over V and D at 1e6 x 256
    bool cf_pending = true;
    // Stop test (:218-220).  The loop's cost comes from the Gram form v^T D v - 2 a.b + a^T G a, whose cancellation error
    // grows with v^T D v (measured 1e-6 .. 1e-5 absolute at 1e6 x 256, depth 120 .. 2500) -- where its bound is not far below tol
    // (confirm_stops), an iteration whose Gram-form |cf - cf_0| falls below kConfirmBand x tol pauses the device
    // (state->done = 2), and the host decides on the streaming cost of deconvolution.py:15-17 for this and the previous
    // iterate (cf_stream, cf_stream_iter), exactly the reference's formula.
    // momentum rows (SolverState::mom): a page-locked staging buffer and its device copy, kMomRows rows of 2 + 2 kMomSteps
    double* mom_host = nullptr;
    double* mom_dev = nullptr;
    // dmf_solver_cost_begin / _end: the streaming cost taken WITHOUT waiting for it (the caller sets up its next solver
    // meanwhile); the event marks the result's arrival in the page-locked slot behind h_state
    hipEvent_t cost_event = nullptr;
    bool cost_pending = false;
    // Has anything been enqueued for this solver since the host last waited for the stream?  (dmf_solver_destroy then
    // waits; otherwise it must not: another solver's work may be running on the context's stream.)
    bool in_flight = true;
    bool confirm_stops = false;
    double cf_stream = 0.0;
    long long cf_stream_iter = -1;
    long long n_confirmed = 0, n_unconfirmed = 0;  // stop tests decided on streaming costs / on the Gram form inside the band
};

namespace {

int clock_drain(dmf_context* ctx, FamilyClock& c) {
    if (c.used == 0) return DMF_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < c.used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c.start[i], c.stop[i]));
        c.total_ms += ms;
    }
    c.launches += c.used;
    c.used = 0;
    return DMF_OK;
}

// Brackets one launch (or a fixed group of launches) of a kernel family with HIP events on
// the context's stream when profiling is enabled.
struct FamilyScope {
    dmf_context* ctx;
    FamilyClock* c = nullptr;
    int slot = -1;
    FamilyScope(dmf_context* ctx_, int family) : ctx(ctx_) {
        if (!((ctx->profiling >> family) & 1u)) return;
        c = &ctx->clocks[family];
        if (c->start.empty()) {
            c->start.resize(kEventPool);
            c->stop.resize(kEventPool);
            for (int i = 0; i < kEventPool; ++i) {
                // timing only: without the system-scope fence a default event carries (its cache write-back and invalidate
                // cost ~5 us of idle GPU per record between two kernels -- 22 us per outer iteration with two families timed)
                hipEventCreateWithFlags(&c->start[i], hipEventDisableSystemFence);
                hipEventCreateWithFlags(&c->stop[i], hipEventDisableSystemFence);
            }
        }
        if (c->used == kEventPool) clock_drain(ctx, *c);
        slot = c->used++;
        hipEventRecord(c->start[slot], ctx->stream);
    }
    ~FamilyScope() {
        if (c != nullptr) hipEventRecord(c->stop[slot], ctx->stream);
    }
};

// Device buffers come from a stream-ordered memory pool OF THE CONTEXT'S OWN on the context's stream.  The pool keeps
// what is freed (release threshold raised in dmf_context_create; the device's default pool, which other users of a
// borrowed device share, is left alone), so the multi-GB buffers of a problem or a solver that is destroyed and
// re-created with the same sizes -- every bootstrap replicate does that -- are handed back without a trip to the
// driver (hipMalloc / hipFree of 2 GB cost tens of milliseconds each).
static bool pool_enabled() {  // DEMETHIFY_NO_POOL=1: plain hipMalloc / hipFree (debugging aid)
    static const bool on = [] {
        const char* v = getenv("DEMETHIFY_NO_POOL");
        return !(v != nullptr && v[0] == '1');
    }();
    return on;
}
// Above the pool: freed blocks of 1 MB and more are kept by exact size and handed to the next allocation of that size
// (a bootstrap replicate frees and re-allocates the same seven multi-GB buffers; hipFreeAsync + hipMallocFromPoolAsync
// cost ~0.2 ms per large block even when the pool keeps the memory).  Everything that touches these blocks is enqueued
// on the context's one stream, so a block can be reused the moment it is "freed".  At most kKeepPerSize blocks per size
// and kKeepBytes in total are kept; the rest goes back to the pool.
constexpr size_t kKeepMinBytes = (size_t)1 << 20, kKeepBytes = (size_t)24 << 30;
constexpr int kKeepPerSize = 3;
static hipError_t pool_alloc(dmf_context* ctx, void** p, size_t bytes) {
    if (!pool_enabled() || ctx->pool == nullptr) return hipMalloc(p, bytes);
    if (bytes >= kKeepMinBytes) {
        auto it = ctx->kept.find(bytes);
        if (it != ctx->kept.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            ctx->kept_bytes -= bytes;
            ctx->live[*p] = bytes;
            return hipSuccess;
        }
    }
    hipError_t e = hipMallocFromPoolAsync(p, bytes, ctx->pool, ctx->stream);
    if (e != hipSuccess && ctx->kept_bytes > 0) {  // out of memory with blocks parked here: give them back, try again
        (void)hipGetLastError();
        for (auto& kv : ctx->kept)
            for (void* q : kv.second) (void)hipFreeAsync(q, ctx->stream);
        ctx->kept.clear();
        ctx->kept_bytes = 0;
        (void)hipStreamSynchronize(ctx->stream);
        e = hipMallocFromPoolAsync(p, bytes, ctx->pool, ctx->stream);
    }
    if (e == hipSuccess && bytes >= kKeepMinBytes) ctx->live[*p] = bytes;
    return e;
}
static void pool_free(dmf_context* ctx, void* p) {
    if (p == nullptr) return;
    if (!pool_enabled() || ctx->pool == nullptr) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(p);
        return;
    }
    auto it = ctx->live.find(p);
    if (it != ctx->live.end()) {
        const size_t bytes = it->second;
        ctx->live.erase(it);
        auto& slot = ctx->kept[bytes];
        if ((int)slot.size() < kKeepPerSize && ctx->kept_bytes + bytes <= kKeepBytes) {
            slot.push_back(p);
            ctx->kept_bytes += bytes;
            return;
        }
    }
    (void)hipFreeAsync(p, ctx->stream);
}

int import_array(dmf_context* ctx, const void* src, size_t bytes, int flags, void** dst, bool* owned) {
    if (bytes == 0) {
        *dst = nullptr;
        *owned = false;
        return DMF_OK;
    }
    if (flags & DMF_PTR_DEVICE) {
        *dst = const_cast<void*>(src);
        *owned = false;
        return DMF_OK;
    }
    void* d = nullptr;
    HIP_TRY(pool_alloc(ctx, &d, bytes));
    hipError_t e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        pool_free(ctx, d);
        return hip_fail(e, "hipMemcpyAsync(H2D)", __LINE__);
    }
    *dst = d;
    *owned = true;
    return DMF_OK;
}

int export_array(dmf_context* ctx, const void* dev_src, size_t bytes, int flags, void* dst) {
    if (bytes == 0 || dst == nullptr) return DMF_OK;
    const hipMemcpyKind kind = (flags & DMF_PTR_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    HIP_TRY(hipMemcpyAsync(dst, dev_src, bytes, kind, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DMF_OK;
}

// Builds the per-problem constants: max(D)^2, ||Rt||_F^2 and the known block of the packed Gram.
// `counts_done`: a row-resampled copy (dmf_problem_gather) whose integer count copies are already gathered from the
// source's and whose count constants -- max(D) in h_consts[2] and [4], the exactness flags copied -- are set by the
// caller: the three scans of D and the rebuild of the integer copies are skipped.
int problem_finalize(dmf_problem* p, bool counts_done = false) {
    dmf_context* ctx = p->ctx;
    const int64_t N = p->N, S = p->S, n_c = p->n_c;
    HIP_TRY(pool_alloc(ctx, (void**)&p->consts, 6 * sizeof(double)));
    if (!counts_done) HIP_TRY(dmf::launch_max_f64(p->D, N * S, ctx->scratch, p->consts + 2, ctx->stream));
    if (n_c > 0) {
        HIP_TRY(dmf::launch_sumsq_f64(p->Rt, N * n_c, ctx->scratch + 1024, p->consts + 1, nullptr, ctx->stream));
    } else {
        HIP_TRY(hipMemsetAsync(p->consts + 1, 0, sizeof(double), ctx->stream));
    }
    if (!counts_done) {
        HIP_TRY(dmf::launch_f32_residual_max(p->D, N * S, ctx->scratch + 2048, p->consts + 3, ctx->stream));
        HIP_TRY(dmf::launch_int_count_max(p->D, N * S, ctx->scratch + 3072, p->consts + 4, ctx->stream));
        if (n_c > 0) {
            HIP_TRY(dmf::launch_unit_range_check(p->Rt, N * n_c, ctx->scratch, p->consts + 5, ctx->stream));
        } else {
            HIP_TRY(hipMemsetAsync(p->consts + 5, 0, sizeof(double), ctx->stream));
        }
    }
    {
        double got[6];
        HIP_TRY(hipMemcpyAsync(got, p->consts, 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        p->h_consts[1] = got[1];
        if (!counts_done)
            for (int i = 2; i < 6; ++i) p->h_consts[i] = got[i];
    }
    p->h_consts[0] = p->h_consts[2] * p->h_consts[2];  // d = max(D)**2, deconvolution.py:197
    p->d_f32_exact = p->h_consts[3] == 0.0;
    HIP_TRY(hipMemcpyAsync(p->consts, p->h_consts, 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (!std::isfinite(p->h_consts[2]) || !std::isfinite(p->h_consts[1])) return DMF_ERR_NONFINITE;

    // integer copies of the counts (u16 row-major for the row pass, 8-bit digit planes for the integer-MFMA Gram)
    // (S <= 2048: the row pass itself stops at 512 samples; the panel producer, the integer Gram, b_u and cost kernels do not)
    if (!counts_done && ctx->generic_level == 0 && p->h_consts[4] <= 32639.0 && p->h_consts[5] == 0.0 && S >= 2 &&
        S <= 2048 && n_c <= 48) {
        p->ND = p->h_consts[4] <= 127.0 ? 1 : 2;
        p->SD = (int)((S + 63) / 64 * 64);
        p->N16 = (N + 15) / 16 * 16;
        p->plane_stride = ((N + 31) / 32) * (p->SD / 32) * 1024;
        HIP_TRY(pool_alloc(ctx, (void**)&p->D16, (size_t)p->N16 * p->SD * sizeof(unsigned short)));
        HIP_TRY(pool_alloc(ctx, (void**)&p->Dt8, (size_t)p->plane_stride * p->ND));
        HIP_TRY(dmf::launch_build_counts_int(p->D, N, (int)S, p->ND, p->D16, p->N16, p->SD, p->Dt8, p->plane_stride,
                                             ctx->stream));
    }

    // padded copy of R_trunc for the shape-specialised kernels (aligned, branch-free row loads)
    if (n_c > 0 && n_c <= 48) {  // (<= 16: every shape-specialised kernel; beyond: the wide-row-group producer, the integer Gram)
        const int nct = (int)((n_c + 3) / 4 * 4);
        if (nct == n_c) {
            p->Rtp = p->Rt;
        } else {
            HIP_TRY(pool_alloc(ctx, (void**)&p->Rtp, (size_t)N * nct * sizeof(double)));
            p->own_Rtp = true;
            HIP_TRY(dmf::launch_pad_rows(p->Rt, p->Rtp, N, (int)n_c, nct, ctx->stream));
        }
    }

    // known block: packed triangle over the extended indices (Rt_0..Rt_{n_c-1}, v)
    const int ext = (int)n_c + 1;
    const int n_jobs = ext * (ext + 1) / 2;
    std::vector<short> hk(n_jobs), hl(n_jobs);
    std::vector<int> hd(n_jobs);
    int a = 0;
    for (int l = 0; l < ext; ++l)
        for (int k = 0; k <= l; ++k, ++a) {
            hk[a] = (short)k;
            hl[a] = (short)l;
            hd[a] = dmf::tri(k, l);
        }
    short *dk = nullptr, *dl = nullptr;
    int* dd = nullptr;
    double* slab = nullptr;
    // all but the last job (v, v) are sums of row-feature products against D or D * V: the matrix-core Gram
    // kernel takes them (n_c <= 16 here: 136 + 16 jobs at most); v^T D v goes through the generic kernel alone
    const bool mfma = n_c >= 1 && ctx->generic_level != 1 && ctx->generic_level != 2;
    const int n_fast = mfma ? n_jobs - 1 : 0, n_dense = (int)(n_c * (n_c + 1) / 2);
    // With integer copies of the counts the known block takes the solver's own integer route: the dense pairs on the integer
    // matrix cores from the 8-bit planes (exact sums of fixed-point products), the right-hand sides sum_i Rt_ik d_is v_is
    // from the u16 stream kernel -- 0.26 + 2.6 GB instead of the 4.5 GB of V and the f64 counts that the FP64 matrix-core
    // kernel reads (1.5 ms at 1e6 x 256 x 12; every bootstrap replicate builds a problem).
    const bool int_known = mfma && ctx->generic_level == 0 && p->Dt8 != nullptr && p->D16 != nullptr && p->Rtp != nullptr &&
                           (reinterpret_cast<uintptr_t>(p->Rtp) & 15) == 0 &&
                           dmf::gram_i8_known_supported((int)n_c, p->ND, N, p->SD);
    int64_t slab_doubles = dmf::gram_slab_doubles(N, (int)S, mfma ? 1 : n_jobs);
    if (mfma && !int_known) {
        const int64_t need = dmf::gram_mfma_slab_doubles(N, (int)S, n_fast);
        if (need > slab_doubles) slab_doubles = need;
    }
    HIP_TRY(pool_alloc(ctx, (void**)&p->gb_known, (size_t)n_jobs * S * sizeof(double)));
    HIP_TRY(pool_alloc(ctx, (void**)&dk, n_jobs * sizeof(short)));
    HIP_TRY(pool_alloc(ctx, (void**)&dl, n_jobs * sizeof(short)));
    HIP_TRY(pool_alloc(ctx, (void**)&dd, n_jobs * sizeof(int)));
    HIP_TRY(pool_alloc(ctx, (void**)&slab, (size_t)slab_doubles * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(dk, hk.data(), n_jobs * sizeof(short), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dl, hl.data(), n_jobs * sizeof(short), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dd, hd.data(), n_jobs * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    hipError_t e = hipSuccess;
    bool vdv_done = true;  // (in: asked for; out: delivered)
    if (int_known) {
        long long *slab_i8 = nullptr, *acc = nullptr;
        double* slab_bu = nullptr;
        const int64_t slab_words = dmf::gram_i8_slab_words_nf(N, p->SD, n_dense);
        const int64_t acc_words = dmf::gram_i8_acc_words_nf((int)S, n_dense, (int)n_c + 1);
        e = pool_alloc(ctx, (void**)&slab_i8, (size_t)slab_words * sizeof(long long));
        if (e == hipSuccess) e = pool_alloc(ctx, (void**)&acc, (size_t)acc_words * sizeof(long long));
        if (e == hipSuccess) e = pool_alloc(ctx, (void**)&slab_bu, (size_t)dmf::bu_cols_grid(N) * (n_c + 1) * S * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(acc, 0, (size_t)acc_words * sizeof(long long), ctx->stream);
        int ny = 0, n_slabs = 0;
        if (e == hipSuccess)
            e = dmf::launch_gram_i8(p->Dt8, p->plane_stride, p->SD, p->ND, p->Rtp, nullptr, N, (int)n_c, 0, dk, dl, n_dense,
                                    slab_i8, slab_words, nullptr, &ny, ctx->stream);
        if (e == hipSuccess)  // (v^T D v rides along where the two-samples-per-lane form of the stream kernel runs)
            e = dmf::launch_bu_cols(p->V, p->D16, p->SD, p->Rt, N, (int)S, (int)n_c, slab_bu, nullptr, &n_slabs, ctx->stream,
                                    &vdv_done);
        // (dd lists the dense pairs first, then the n_c right-hand sides, then (v, v): the order of the reduce's jobs)
        if (e == hipSuccess)
            e = dmf::launch_gram_v2_reduce(slab_i8, ny, n_dense, p->SD, slab_bu, n_slabs, (int)n_c + (vdv_done ? 1 : 0), (int)S,
                                           acc, dd, p->gb_known, nullptr, nullptr, 0, nullptr, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        pool_free(ctx, slab_i8);
        pool_free(ctx, acc);
        pool_free(ctx, slab_bu);
    } else if (mfma) {
        dmf::GramJobTable fast{dk, dl, dd, n_fast};
        int ny = 0;
        e = dmf::launch_gram_mfma(p->V, p->D, p->Rt, nullptr, N, (int)S, (int)n_c, 0, fast, n_dense, slab,
                                  slab_doubles, nullptr, &ny, ctx->stream);
        if (e == hipSuccess)
            e = dmf::launch_gram_reduce(slab, ny, n_fast, (int)S, dd, p->gb_known, nullptr, ctx->stream);
    }
    if (e == hipSuccess && int_known && vdv_done) {
        // (nothing left)
    } else if (e == hipSuccess && n_jobs - n_fast == 1 && ctx->generic_level != 1 && ctx->generic_level != 2 &&
        (int64_t)dmf::vdv_cols_grid(N) * S <= slab_doubles) {
        // what is left is v^T D v alone: a stream kernel of its own (the generic kernel took 2.7 ms for it at 1e6 x 256)
        e = dmf::launch_vdv_cols(p->V, p->D, p->D16, p->SD, N, (int)S, slab, p->gb_known + (int64_t)hd[n_jobs - 1] * S,
                                 ctx->stream);
    } else if (e == hipSuccess) {
        dmf::GramJobTable rest{dk + n_fast, dl + n_fast, dd + n_fast, n_jobs - n_fast};
        e = dmf::launch_gram(p->V, p->D, p->Rt, nullptr, N, (int)S, (int)n_c, 0, rest, slab, slab_doubles,
                             p->gb_known, nullptr, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    pool_free(ctx, dk);
    pool_free(ctx, dl);
    pool_free(ctx, dd);
    pool_free(ctx, slab);
    if (e != hipSuccess) return hip_fail(e, "launch_gram(known block)", __LINE__);
    return DMF_OK;
}

// cost_f_w of (u, alpha) on the problem's data: the column-resident kernel when the shape allows, else the generic one
hipError_t enqueue_cost(dmf_context* ctx, const dmf_problem* p, const double* u, const double* alpha, int n_u,
                        double* scratch, double* out) {
    const bool rtp_ok = p->n_c == 0 || p->Rtp != nullptr;
    if ((ctx->generic_level == 0 || ctx->generic_level == 3 || ctx->generic_level == 4) && rtp_ok &&
        dmf::cost_cols_supported((int)p->S, (int)p->n_c, n_u))
        return dmf::launch_cost_cols(p->V, p->D, p->D16, p->SD, p->Rtp, u, alpha, p->N, (int)p->S, (int)p->n_c, n_u,
                                     scratch, out, ctx->stream);
    if (ctx->generic_level == 0 && rtp_ok &&
        dmf::cost_cols2_wide_supported(p->V, p->D16, (int)p->S, p->SD, (int)p->n_c, n_u))
        return dmf::launch_cost_cols2_wide(p->V, p->D16, p->SD, p->Rtp, u, alpha, p->N, (int)p->S, (int)p->n_c, n_u, scratch,
                                           out, ctx->stream);
    return dmf::launch_cost(p->V, p->D, p->Rt, u, alpha, p->N, (int)p->S, (int)p->n_c, n_u, scratch, out, ctx->stream);
}

int check_ctx(dmf_context* ctx) {
    if (ctx == nullptr) return DMF_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    return DMF_OK;
}

using dmf::kSplitInnerSteps;

// scratch of the split u phase: per-row c_i / M_i and the momentum coefficients of the inner steps (allocated on first use)
int ensure_split_scratch(dmf_solver* s, int n_iter2) {
    dmf_context* ctx = s->ctx;
    const dmf_problem* p = s->p;
    if (s->cm == nullptr)
        HIP_TRY(pool_alloc(ctx, (void**)&s->cm, (size_t)dmf::u_phase_split_cm_doubles(p->N, (int)s->n_u) * sizeof(double)));
    if (s->beta_cap < n_iter2 || s->beta_tab == nullptr) {
        pool_free(ctx, s->beta_tab);
        s->beta_tab = nullptr;
        HIP_TRY(pool_alloc(ctx, (void**)&s->beta_tab, (size_t)(n_iter2 > 0 ? n_iter2 : 1) * sizeof(double)));
        s->beta_cap = n_iter2 > 0 ? n_iter2 : 1;
    }
    return DMF_OK;
}

int enqueue_u_phase(dmf_solver* s, int n_iter2, dmf::RowKind row) {
    dmf_context* ctx = s->ctx;
    const dmf_problem* p = s->p;
    FamilyScope scope(ctx, DMF_KERNEL_ROWPASS);
    switch (row) {
        case dmf::RowKind::CmI8InnerRows:
            // wide row groups on u16 counts: per-row c_i / M_i with M_i on the integer matrix cores, then the inner
            // iterations chip-wide (dmf_kernels_cm_i8.hip)
            DMF_TRY(ensure_split_scratch(s, n_iter2));
            HIP_TRY(dmf::launch_u_phase_split_i8(p->V, p->D16, p->SD, p->ND, p->Rtp, s->alpha, s->u, s->u_prev, s->state, p->N,
                                                 (int)p->S, (int)p->n_c, (int)s->n_u, n_iter2, s->mode, s->cm, s->beta_tab,
                                                 ctx->stream));
            return DMF_OK;
        case dmf::RowKind::UPhaseBig:
            HIP_TRY(dmf::launch_u_phase_big(p->V, p->D, p->Rtp, s->alpha, s->u, s->u_prev, s->state, p->N, (int)p->S,
                                            (int)p->n_c, (int)s->n_u, n_iter2, s->mode, ctx->stream));
            return DMF_OK;
        case dmf::RowKind::UPhaseMfmaSplit:
            // many inner steps or wide row groups: one wave per workgroup running the inner steps is the bottleneck
            DMF_TRY(ensure_split_scratch(s, n_iter2));
            HIP_TRY(dmf::launch_u_phase_split(p->V, p->D, p->D16, p->SD, p->Rtp, s->alpha, s->u, s->u_prev, s->state, p->N,
                                              (int)p->S, (int)p->n_c, (int)s->n_u, n_iter2, s->mode, s->cm, s->beta_tab,
                                              ctx->stream));
            return DMF_OK;
        case dmf::RowKind::UPhaseMfma:
            HIP_TRY(dmf::launch_u_phase_mfma(p->V, p->D, p->D16, p->SD, p->Rtp, s->alpha, s->u, s->u_prev, s->state, p->N,
                                             (int)p->S, (int)p->n_c, (int)s->n_u, n_iter2, s->mode, ctx->stream));
            return DMF_OK;
        case dmf::RowKind::UPhaseGram:
            HIP_TRY(dmf::launch_u_phase_gram(p->V, p->D, p->Rt, s->alpha, s->u, s->u_prev, s->state, p->N,
                                             (int)p->S, (int)p->n_c, (int)s->n_u, n_iter2, s->mode, ctx->stream));
            return DMF_OK;
        case dmf::RowKind::UStepDirect:
            for (int t = 0; t < n_iter2; ++t) {
                HIP_TRY(dmf::launch_u_step_direct(p->V, p->D, p->Rt, s->alpha, s->u, s->u_prev, s->u_next,
                                                  s->state, p->N, (int)p->S, (int)p->n_c, (int)s->n_u, t,
                                                  s->mode, ctx->stream));
                double* old_prev = s->u_prev;
                s->u_prev = s->u;
                s->u = s->u_next;
                s->u_next = old_prev;
            }
            return DMF_OK;
        default: return DMF_ERR_BAD_ARG;  // (the one-launch row passes are enqueue_outer_iteration's)
    }
}

// the row kind of a u phase that runs as a kernel of its own (the single-function entry points: dmf_update_u)
dmf::RowKind standalone_row_kind(const dmf_solver* s, int n_iter2) {
    dmf::PathSpec spec = s->spec;
    spec.use_v2 = false;  // (plan_iteration then describes the split / fall-back u phase of this solver)
    spec.use_fused = false;
    spec.use_gram_i8 = false;
    return dmf::plan_iteration(s->key, spec, n_iter2, false).row;
}

// kind: GramKind::BuColsI8 only behind a u phase with at least one inner step (its clip puts u inside [0, 1], which the
// fixed-point features need); the dmf_update_alpha entry point hands over the caller's u and passes an FP64 kind
int enqueue_gram(dmf_solver* s, dmf::GramKind kind) {
    dmf_context* ctx = s->ctx;
    const dmf_problem* p = s->p;
    FamilyScope scope(ctx, DMF_KERNEL_GRAM);
    if (kind == dmf::GramKind::BuColsI8) {
        const int S = (int)p->S, n_c = (int)p->n_c, n_u = (int)s->n_u, nf = n_c * n_u + n_u * (n_u + 1) / 2;
        int n_slabs = 0, ny = 0;
        HIP_TRY(dmf::launch_bu_cols(p->V, p->D16, p->SD, s->u, p->N, S, n_u, s->slab, &s->state->done, &n_slabs, ctx->stream));
        HIP_TRY(dmf::launch_gram_i8(p->Dt8, p->plane_stride, p->SD, p->ND, p->Rtp, s->u, p->N, n_c, n_u, s->job_k, s->job_l, nf,
                                    s->slab_i8, s->slab_i8_words, &s->state->done, &ny, ctx->stream));
        HIP_TRY(dmf::launch_gram_v2_reduce(s->slab_i8, ny, nf, p->SD, s->slab, n_slabs, n_u, S, s->acc_i8, s->job_dst, s->gb,
                                           &s->state->done, nullptr, 0, s->state, ctx->stream));
        return DMF_OK;
    }
    if (kind == dmf::GramKind::GramU) {
        int ny = 0;
        HIP_TRY(dmf::launch_gram_u(p->V, p->D, p->Rtp, s->u, p->N, (int)p->S, (int)p->n_c, (int)s->n_u, s->slab,
                                   &s->state->done, &ny, ctx->stream));
        HIP_TRY(dmf::launch_gram_reduce(s->slab, ny, s->n_jobs, (int)p->S, s->job_dst, s->gb, &s->state->done,
                                        ctx->stream));
        return DMF_OK;
    }
    dmf::GramJobTable jobs{s->job_k, s->job_l, s->job_dst, s->n_jobs};
    if (kind == dmf::GramKind::GramMfma) {
        int ny = 0;
        HIP_TRY(dmf::launch_gram_mfma(p->V, p->D, p->Rt, s->u, p->N, (int)p->S, (int)p->n_c, (int)s->n_u, jobs,
                                      s->n_jobs - (int)s->n_u, s->slab, s->slab_doubles, &s->state->done, &ny,
                                      ctx->stream));
        HIP_TRY(dmf::launch_gram_reduce(s->slab, ny, s->n_jobs, (int)p->S, s->job_dst, s->gb, &s->state->done,
                                        ctx->stream));
        return DMF_OK;
    }
    HIP_TRY(dmf::launch_gram(p->V, p->D, p->Rt, s->u, p->N, (int)p->S, (int)p->n_c, (int)s->n_u, jobs,
                             s->slab, s->slab_doubles, s->gb, &s->state->done, ctx->stream));
    return DMF_OK;
}

dmf::GramKind fp64_gram_kind(const dmf_solver* s) {
    return s->spec.use_gram_spec ? dmf::GramKind::GramU : s->spec.use_gram_mfma ? dmf::GramKind::GramMfma : dmf::GramKind::Gram;
}

int enqueue_alpha_phase(dmf_solver* s, int n_iter2) {
    dmf_context* ctx = s->ctx;
    const dmf_problem* p = s->p;
    FamilyScope scope(ctx, DMF_KERNEL_ALPHA);
    if (s->purity != nullptr) {
        HIP_TRY(dmf::launch_alpha_frank_wolfe(s->gb, s->alpha, s->purity, s->state, (int)p->S, (int)p->n_c,
                                              (int)s->n_u, n_iter2, s->partials, ctx->stream));
        return DMF_OK;
    }
    const bool thread_per_sample = ctx->generic_level == 1 || ctx->generic_level == 2;
    HIP_TRY(dmf::launch_alpha_phase(s->gb, s->alpha, s->alpha_prev, s->state, (int)p->S, (int)p->n_c,
                                    (int)s->n_u, n_iter2, s->partials, thread_per_sample, ctx->stream));
    return DMF_OK;
}

int enqueue_outer_iteration(dmf_solver* s, int n_iter2) {
    dmf_context* ctx = s->ctx;
    s->in_flight = true;
    const dmf_problem* p = s->p;
    // which kernels: dmf_select.hip (one table for create / enqueue / describe)
    const dmf::IterationPlan plan = dmf::plan_iteration(s->key, s->spec, n_iter2, s->purity != nullptr);
    if (plan.row == dmf::RowKind::RowpassV2) {
        // Second generation: one read of V (f64) and of the u16 counts for the u phase and b_u, then the exact
        // integer-matrix-core GEMM for the u-dependent Gram entries on the 8-bit count planes.
        const int S = (int)p->S, n_c = (int)p->n_c, n_u = (int)s->n_u;
        const int nf = n_c * n_u + n_u * (n_u + 1) / 2;
        int grid = 0, ny = 0;
        {
            FamilyScope scope(ctx, DMF_KERNEL_ROWPASS);
            HIP_TRY(dmf::launch_rowpass_v2(p->V, p->D16, p->SD, p->Rtp, s->alpha, s->u, s->u_prev, s->state, p->N, S,
                                           n_c, n_u, n_iter2, s->mode, p->ND, s->slab, s->u2_partials, &grid, ctx->stream));
        }
        {
            FamilyScope scope(ctx, DMF_KERNEL_GRAM);
            HIP_TRY(dmf::launch_gram_i8(p->Dt8, p->plane_stride, p->SD, p->ND, p->Rtp, s->u, p->N, n_c, n_u, s->job_k,
                                        s->job_l, nf, s->slab_i8, s->slab_i8_words, &s->state->done, &ny, ctx->stream));
            HIP_TRY(dmf::launch_gram_v2_reduce(s->slab_i8, ny, nf, p->SD, s->slab, grid, n_u, S, s->acc_i8, s->job_dst,
                                               s->gb, &s->state->done, s->u2_partials, grid, s->state, ctx->stream));
        }
        DMF_TRY(enqueue_alpha_phase(s, n_iter2));
        return DMF_OK;
    }
    if (plan.row == dmf::RowKind::CmI8InnerBu) {
        // Wide row groups on u16 counts: c_i / M_i (M_i on the integer matrix cores), then the inner iterations fused with
        // the b_u stream and the ||u||^2 shares, then the integer Gram and its reduce -- four launches + the momentum table.
        const int S = (int)p->S, n_c = (int)p->n_c, n_u = (int)s->n_u;
        const int nf = n_c * n_u + n_u * (n_u + 1) / 2;
        int grid = 0, ny = 0;
        DMF_TRY(ensure_split_scratch(s, n_iter2));
        {
            FamilyScope scope(ctx, DMF_KERNEL_ROWPASS);
            HIP_TRY(dmf::launch_u_phase_split_i8_bu(p->V, p->D16, p->SD, p->ND, p->Rtp, s->alpha, s->u, s->u_prev, s->state,
                                                    p->N, S, n_c, n_u, n_iter2, s->mode, s->cm, s->beta_tab, s->slab,
                                                    s->u2_partials, &grid, ctx->stream));
        }
        {
            FamilyScope scope(ctx, DMF_KERNEL_GRAM);
            HIP_TRY(dmf::launch_gram_i8(p->Dt8, p->plane_stride, p->SD, p->ND, p->Rtp, s->u, p->N, n_c, n_u, s->job_k,
                                        s->job_l, nf, s->slab_i8, s->slab_i8_words, &s->state->done, &ny, ctx->stream));
            HIP_TRY(dmf::launch_gram_v2_reduce(s->slab_i8, ny, nf, p->SD, s->slab, grid, n_u, S, s->acc_i8, s->job_dst,
                                               s->gb, &s->state->done, s->u2_partials, grid, s->state, ctx->stream));
        }
        DMF_TRY(enqueue_alpha_phase(s, n_iter2));
        return DMF_OK;
    }
    if (plan.row == dmf::RowKind::RowpassFused) {
        // The fused kernel takes whole 16-row blocks; a ragged tail (< 16 rows) goes through the unfused
        // pair on offset pointers and contributes extra slab rows and one more ||u||^2 share.
        const int64_t n_full = p->N - (p->N & 15), n_tail = p->N - n_full;
        const int S = (int)p->S, n_c = (int)p->n_c, n_u = (int)s->n_u, nct = (n_c + 3) / 4 * 4;
        int grid = 0, ny_tail = 0;
        {
            FamilyScope scope(ctx, DMF_KERNEL_ROWPASS);
            HIP_TRY(dmf::launch_rowpass_fused(p->V, p->D, p->Rtp, s->alpha, s->u, s->u_prev, s->state, n_full, S, n_c,
                                              n_u, n_iter2, s->mode, s->slab, s->u2_partials, &grid, ctx->stream));
        }
        if (n_tail > 0) {
            const double* rt_tail = p->Rtp ? p->Rtp + n_full * nct : nullptr;
            double* u_tail = s->u + n_full * n_u;
            HIP_TRY(dmf::launch_u_phase_mfma(p->V + n_full * S, p->D + n_full * S, nullptr, 0, rt_tail, s->alpha, u_tail,
                                             s->u_prev + n_full * n_u, s->state, n_tail, S, n_c, n_u, n_iter2,
                                             s->mode, ctx->stream));
            HIP_TRY(dmf::launch_sumsq_f64(u_tail, n_tail * n_u, ctx->scratch, s->u2_partials + grid, &s->state->done,
                                          ctx->stream));
            HIP_TRY(dmf::launch_gram_u(p->V + n_full * S, p->D + n_full * S, rt_tail, u_tail, n_tail, S, n_c, n_u,
                                       s->slab + (int64_t)2 * grid * s->n_jobs * S, &s->state->done, &ny_tail,
                                       ctx->stream));
        }
        HIP_TRY(dmf::launch_finish_u_norm(s->u2_partials, grid + (n_tail > 0 ? 1 : 0), s->state, ctx->stream));
        {
            FamilyScope scope(ctx, DMF_KERNEL_GRAM);
            HIP_TRY(dmf::launch_gram_reduce(s->slab, 2 * grid + ny_tail, s->n_jobs, S, s->job_dst, s->gb,
                                            &s->state->done, ctx->stream));
        }
        DMF_TRY(enqueue_alpha_phase(s, n_iter2));
        return DMF_OK;
    }
    DMF_TRY(enqueue_u_phase(s, n_iter2, plan.row));
    HIP_TRY(dmf::launch_sumsq_f64(s->u, p->N * s->n_u, ctx->scratch, &s->state->u_norm2, &s->state->done,
                                  ctx->stream));
    HIP_TRY(dmf::launch_set_lh(s->state, ctx->stream));
    DMF_TRY(enqueue_gram(s, plan.gram));
    DMF_TRY(enqueue_alpha_phase(s, n_iter2));
    return DMF_OK;
}

int fetch_state(dmf_solver* s) {
    HIP_TRY(hipMemcpyAsync(s->h_state, s->state, sizeof(SolverState), hipMemcpyDeviceToHost, s->ctx->stream));
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    s->in_flight = false;
    return DMF_OK;
}

int push_state(dmf_solver* s) {
    HIP_TRY(hipMemcpyAsync(s->state, s->h_state, sizeof(SolverState), hipMemcpyHostToDevice, s->ctx->stream));
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    return DMF_OK;
}

double advance_momentum(double a, int64_t n) {
    for (int64_t t = 0; t < n; ++t) a = (1.0 + std::sqrt(1.0 + 4.0 * a * a)) / 2.0;
    return a;
}

}  // namespace

extern "C" {

const char* dmf_status_string(int status) {
    switch (status) {
        case DMF_OK: return "ok";
        case DMF_ERR_BAD_ARG: return "bad argument";
        case DMF_ERR_BAD_SHAPE: return "shape mismatch";
        case DMF_ERR_HIP: return "HIP runtime error";
        case DMF_ERR_NONFINITE: return "non-finite input";
        case DMF_ERR_UNSUPPORTED: return "unsupported size";
        case DMF_ERR_NO_DEVICE: return "no gfx950 device";
        default: return "unknown status";
    }
}

const char* dmf_last_error(void) { return g_last_error; }

int dmf_abi_version(void) { return 1; }

int dmf_context_create(int device, void* stream, dmf_context** out) {
    if (out == nullptr) return DMF_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return DMF_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return DMF_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_last_error, sizeof(g_last_error), "device %d is %s, this library is built for gfx950",
                 device, prop.gcnArchName);
        return DMF_ERR_NO_DEVICE;
    }
    dmf_context* ctx = new (std::nothrow) dmf_context();
    if (ctx == nullptr) return DMF_ERR_BAD_ARG;
    ctx->device = device;
    if (stream != nullptr) {
        ctx->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return hip_fail(e, "hipStreamCreate", __LINE__);
        }
        ctx->own_stream = true;
    }
    hipError_t e = hipMalloc((void**)&ctx->scratch, 4096 * sizeof(double));
    if (e != hipSuccess) {
        if (ctx->own_stream) hipStreamDestroy(ctx->stream);
        delete ctx;
        return hip_fail(e, "hipMalloc(scratch)", __LINE__);
    }
    // a pool of the context's own that keeps freed memory for the next problem / solver of the same size (see
    // pool_alloc); if the runtime cannot create one, plain hipMalloc / hipFree are used
    if (pool_enabled()) {
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = device;
        if (hipMemPoolCreate(&ctx->pool, &props) == hipSuccess && ctx->pool != nullptr) {
            uint64_t keep = UINT64_MAX;
            (void)hipMemPoolSetAttribute(ctx->pool, hipMemPoolAttrReleaseThreshold, &keep);
        } else {
            ctx->pool = nullptr;
            (void)hipGetLastError();
        }
    }
    *out = ctx;
    return DMF_OK;
}

int dmf_context_destroy(dmf_context* ctx) {
    if (ctx == nullptr) return DMF_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& c : ctx->clocks) {
        for (auto ev : c.start) hipEventDestroy(ev);
        for (auto ev : c.stop) hipEventDestroy(ev);
    }
    hipFree(ctx->scratch);
    for (auto& kv : ctx->kept)
        for (void* q : kv.second) (void)hipFreeAsync(q, ctx->stream);
    ctx->kept.clear();
    (void)hipStreamSynchronize(ctx->stream);
    for (void* h : ctx->pinned_states) (void)hipHostFree(h);
    for (double* h : ctx->pinned_moms) (void)hipHostFree(h);
    for (hipEvent_t ev : ctx->events) (void)hipEventDestroy(ev);
    if (ctx->copy_stream != nullptr) {
        hipStreamSynchronize(ctx->copy_stream);
        hipStreamDestroy(ctx->copy_stream);
    }
    if (ctx->pool != nullptr) (void)hipMemPoolDestroy(ctx->pool);  // hands the cached buffers back to the driver
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return DMF_OK;
}

/* Staging for a restart loop (demethify/demethify.py:165-171,195-201): the next restart's initialisation goes to the
 * device from a worker thread, on a copy stream of the context's own, while `stream` runs the current restart; the
 * solver is then created from the device copy (DMF_PTR_DEVICE).  Thread-safe; returns when the copy is complete. */
int dmf_stage_upload(dmf_context* ctx, const void* host, size_t bytes, void** out_dev) {
    if (ctx == nullptr || host == nullptr || out_dev == nullptr || bytes == 0) return DMF_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));  // (the current device is per thread)
    std::lock_guard<std::mutex> lock(ctx->copy_mutex);
    if (ctx->copy_stream == nullptr) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    void* d = nullptr;
    if (pool_enabled() && ctx->pool != nullptr) HIP_TRY(hipMallocFromPoolAsync(&d, bytes, ctx->pool, ctx->copy_stream));
    else HIP_TRY(hipMalloc(&d, bytes));
    hipError_t e = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(e, "dmf_stage_upload", __LINE__);
    }
    *out_dev = d;
    return DMF_OK;
}

int dmf_stage_free(dmf_context* ctx, void* dev) {
    if (ctx == nullptr) return DMF_ERR_BAD_ARG;
    if (dev == nullptr) return DMF_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    pool_free(ctx, dev);  // ordered behind the work of `stream` that read it
    return DMF_OK;
}

int dmf_context_synchronize(dmf_context* ctx) {
    DMF_TRY(check_ctx(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DMF_OK;
}

int dmf_context_set_profiling(dmf_context* ctx, int enabled) {
    DMF_TRY(check_ctx(ctx));
    if (!enabled)
        for (auto& c : ctx->clocks) DMF_TRY(clock_drain(ctx, c));
    // 0 = off, 1 = every family, otherwise a mask with bit (1 + family) set for the families to time
    // (2 = DMF_KERNEL_ROWPASS only, ...): each timed launch costs two event records on the stream
    ctx->profiling = enabled == 0 ? 0u : enabled == 1 ? ~0u : (unsigned)enabled >> 1;
    return DMF_OK;
}

int dmf_context_kernel_time(dmf_context* ctx, int family, double* total_ms, int64_t* launches) {
    DMF_TRY(check_ctx(ctx));
    if (family < 0 || family >= DMF_KERNEL_FAMILIES) return DMF_ERR_BAD_ARG;
    FamilyClock& c = ctx->clocks[family];
    DMF_TRY(clock_drain(ctx, c));
    if (total_ms) *total_ms = c.total_ms;
    if (launches) *launches = c.launches;
    return DMF_OK;
}

int dmf_context_reset_kernel_time(dmf_context* ctx) {
    DMF_TRY(check_ctx(ctx));
    for (auto& c : ctx->clocks) {
        DMF_TRY(clock_drain(ctx, c));
        c.total_ms = 0.0;
        c.launches = 0;
    }
    return DMF_OK;
}

int dmf_context_set_generic(dmf_context* ctx, int enabled) {
    if (ctx == nullptr) return DMF_ERR_BAD_ARG;
    if (enabled < 0 || enabled > 4) return DMF_ERR_BAD_ARG;
    ctx->generic_level = enabled;
    return DMF_OK;
}

int dmf_context_set_stop_confirmation(dmf_context* ctx, int mode) {
    if (ctx == nullptr || mode < 0 || mode > 2) return DMF_ERR_BAD_ARG;
    ctx->stop_confirmation = mode;
    return DMF_OK;
}

// ------------------------------------------------------------------------------- problem
int dmf_problem_create(dmf_context* ctx, int64_t N, int64_t S, int64_t n_c, const double* V,
                       const void* counts, const double* Rt, int flags, dmf_problem** out) {
    DMF_TRY(check_ctx(ctx));
    if (out == nullptr) return DMF_ERR_BAD_ARG;
    *out = nullptr;
    if (N <= 0 || S <= 0 || n_c < 0 || V == nullptr || counts == nullptr) return DMF_ERR_BAD_ARG;
    if (n_c > 0 && Rt == nullptr) return DMF_ERR_BAD_ARG;
    if (n_c > dmf::kMaxK || S > (1 << 24)) return DMF_ERR_UNSUPPORTED;
    dmf_problem* p = new (std::nothrow) dmf_problem();
    if (p == nullptr) return DMF_ERR_BAD_ARG;
    p->ctx = ctx;
    p->N = N;
    p->S = S;
    p->n_c = n_c;
    int st = import_array(ctx, V, (size_t)N * S * sizeof(double), flags, (void**)&p->V, &p->own_V);
    if (st == DMF_OK) {
        if (flags & DMF_COUNTS_F64) {
            st = import_array(ctx, counts, (size_t)N * S * sizeof(double), flags, (void**)&p->D, &p->own_D);
        } else {
            void* raw = nullptr;
            bool own_raw = false;
            st = import_array(ctx, counts, (size_t)N * S * sizeof(long long), flags, &raw, &own_raw);
            if (st == DMF_OK) {
                hipError_t e = pool_alloc(ctx, (void**)&p->D, (size_t)N * S * sizeof(double));
                if (e == hipSuccess) {
                    p->own_D = true;
                    e = dmf::launch_convert_counts((const long long*)raw, p->D, N * S, ctx->stream);
                }
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (own_raw) pool_free(ctx, raw);
                if (e != hipSuccess) st = hip_fail(e, "count conversion", __LINE__);
            }
        }
    }
    if (st == DMF_OK)
        st = import_array(ctx, Rt, (size_t)N * n_c * sizeof(double), flags, (void**)&p->Rt, &p->own_Rt);
    if (st == DMF_OK) st = problem_finalize(p);
    if (st != DMF_OK) {
        dmf_problem_destroy(p);
        return st;
    }
    *out = p;
    return DMF_OK;
}

// the row gather behind both entry points; idx_dev: the index array is the caller's device array (range-checked here, on the
// device), else a host array (checked by the caller of this function; uploaded here)
static int problem_gather(dmf_context* ctx, const dmf_problem* src, const int64_t* idx, int64_t n_idx, bool idx_dev,
                          dmf_problem** out) {
    dmf_problem* p = new (std::nothrow) dmf_problem();
    if (p == nullptr) return DMF_ERR_BAD_ARG;
    p->ctx = ctx;
    p->N = n_idx;
    p->S = src->S;
    p->n_c = src->n_c;
    long long* d_idx = nullptr;
    int st = DMF_OK;
    hipError_t e = hipSuccess;
    if (idx_dev) {
        unsigned int* d_bad = nullptr;
        unsigned int h_bad = 1;
        e = pool_alloc(ctx, (void**)&d_bad, sizeof(unsigned int));
        if (e == hipSuccess) e = dmf::launch_index_range_check(reinterpret_cast<const long long*>(idx), n_idx, src->N, d_bad, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        pool_free(ctx, d_bad);
        if (e != hipSuccess || h_bad != 0) {
            delete p;
            return e != hipSuccess ? hip_fail(e, "index range check", __LINE__) : DMF_ERR_BAD_ARG;
        }
        d_idx = reinterpret_cast<long long*>(const_cast<int64_t*>(idx));
    } else {
        e = pool_alloc(ctx, (void**)&d_idx, (size_t)n_idx * sizeof(long long));
        if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, (size_t)n_idx * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&p->V, (size_t)n_idx * p->S * sizeof(double));
    if (e == hipSuccess) p->own_V = true, e = pool_alloc(ctx, (void**)&p->D, (size_t)n_idx * p->S * sizeof(double));
    if (e == hipSuccess) p->own_D = true;
    if (e == hipSuccess && p->n_c > 0) {
        e = pool_alloc(ctx, (void**)&p->Rt, (size_t)n_idx * p->n_c * sizeof(double));
        if (e == hipSuccess) p->own_Rt = true;
    }
    if (e == hipSuccess) e = dmf::launch_gather_rows(src->V, p->V, d_idx, n_idx, p->S, ctx->stream);
    if (e == hipSuccess) e = dmf::launch_gather_rows(src->D, p->D, d_idx, n_idx, p->S, ctx->stream);
    if (e == hipSuccess && p->n_c > 0) e = dmf::launch_gather_rows(src->Rt, p->Rt, d_idx, n_idx, p->n_c, ctx->stream);
    // integer counts: the source's u16 copy is gathered too (0.5 GB instead of a rebuild from the 2 GB f64 copy) and the
    // resampled maximum comes out of the same pass; integrality / range of the counts and of R_trunc carry over from the
    // source, so none of the scans of problem_finalize has to run again
    bool counts_done = false;
    unsigned int* d_max = nullptr;
    if (e == hipSuccess && src->D16 != nullptr && src->ND > 0 && ctx->generic_level == 0) {
        p->ND = src->ND;
        p->SD = src->SD;
        p->N16 = (n_idx + 15) / 16 * 16;
        p->plane_stride = ((n_idx + 31) / 32) * (p->SD / 32) * 1024;
        e = pool_alloc(ctx, (void**)&p->D16, (size_t)p->N16 * p->SD * sizeof(unsigned short));
        if (e == hipSuccess) e = pool_alloc(ctx, (void**)&p->Dt8, (size_t)p->plane_stride * p->ND);
        if (e == hipSuccess) e = pool_alloc(ctx, (void**)&d_max, sizeof(unsigned int));
        if (e == hipSuccess)
            e = dmf::launch_gather_counts_int(src->D16, d_idx, n_idx, p->SD, p->ND, p->D16, p->N16, p->Dt8, p->plane_stride,
                                              d_max, ctx->stream);
        unsigned int h_max = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&h_max, d_max, sizeof(h_max), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) {
            p->h_consts[2] = p->h_consts[4] = (double)h_max;
            p->h_consts[3] = src->h_consts[3];
            p->h_consts[5] = src->h_consts[5];
            counts_done = true;
        }
        pool_free(ctx, d_max);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (!idx_dev) pool_free(ctx, d_idx);
    if (e != hipSuccess) st = hip_fail(e, "row gather", __LINE__);
    if (st == DMF_OK) st = problem_finalize(p, counts_done);
    if (st != DMF_OK) {
        dmf_problem_destroy(p);
        return st;
    }
    *out = p;
    return DMF_OK;
}

int dmf_problem_gather(dmf_context* ctx, const dmf_problem* src, const int64_t* idx, int64_t n_idx,
                       dmf_problem** out) {
    DMF_TRY(check_ctx(ctx));
    if (src == nullptr || idx == nullptr || out == nullptr || n_idx <= 0) return DMF_ERR_BAD_ARG;
    *out = nullptr;
    for (int64_t r = 0; r < n_idx; ++r)
        if (idx[r] < 0 || idx[r] >= src->N) return DMF_ERR_BAD_ARG;
    return problem_gather(ctx, src, idx, n_idx, false, out);
}

int dmf_problem_gather_device(dmf_context* ctx, const dmf_problem* src, const int64_t* idx_dev, int64_t n_idx,
                              dmf_problem** out) {
    DMF_TRY(check_ctx(ctx));
    if (src == nullptr || idx_dev == nullptr || out == nullptr || n_idx <= 0) return DMF_ERR_BAD_ARG;
    *out = nullptr;
    return problem_gather(ctx, src, idx_dev, n_idx, true, out);
}

int dmf_problem_destroy(dmf_problem* p) {
    if (p == nullptr) return DMF_OK;
    dmf_context* ctx = p->ctx;
    hipSetDevice(p->ctx->device);
    if (p->own_V) pool_free(ctx, p->V);
    if (p->own_D) pool_free(ctx, p->D);
    if (p->own_Rt) pool_free(ctx, p->Rt);
    if (p->own_Rtp) pool_free(ctx, p->Rtp);
    pool_free(ctx, p->consts);
    pool_free(ctx, p->gb_known);
    pool_free(ctx, p->D16);
    pool_free(ctx, p->Dt8);
    delete p;
    return DMF_OK;
}

int dmf_problem_shape(const dmf_problem* p, int64_t* N, int64_t* S, int64_t* n_c) {
    if (p == nullptr) return DMF_ERR_BAD_ARG;
    if (N) *N = p->N;
    if (S) *S = p->S;
    if (n_c) *n_c = p->n_c;
    return DMF_OK;
}

// ------------------------------------------------------------------------------- solver
int dmf_solver_create(dmf_context* ctx, const dmf_problem* p, const double* u0, const double* alpha0,
                      int64_t n_u, int mode, int flags, dmf_solver** out) {
    DMF_TRY(check_ctx(ctx));
    if (out == nullptr) return DMF_ERR_BAD_ARG;
    *out = nullptr;
    if (p == nullptr || u0 == nullptr || alpha0 == nullptr || n_u < 1) return DMF_ERR_BAD_ARG;
    if (mode != DMF_MODE_PARTIAL && mode != DMF_MODE_UNSUPERVISED) return DMF_ERR_BAD_ARG;
    const int64_t N = p->N, S = p->S, n_c = p->n_c, K = n_c + n_u;
    if (K > dmf::kMaxK) return DMF_ERR_UNSUPPORTED;
    dmf_solver* s = new (std::nothrow) dmf_solver();
    if (s == nullptr) return DMF_ERR_BAD_ARG;
    s->ctx = ctx;
    s->p = p;
    s->n_u = n_u;
    s->mode = mode;
    // ---- kernel selection: a pure function of this key (dmf_select.hip)
    dmf::ShapeKey& key = s->key;
    key.N = N;
    key.S = (int)S;
    key.n_c = (int)n_c;
    key.n_u = (int)n_u;
    key.nd = (p->ND > 0 && p->D16 != nullptr) ? p->ND : 0;
    key.SD = p->SD;
    key.level = ctx->generic_level;
    key.d_f32_exact = p->d_f32_exact;
    key.rtp_present = n_c == 0 || p->Rtp != nullptr;
    key.v_align = (unsigned)(reinterpret_cast<uintptr_t>(p->V) & 15);
    key.rtp_align = (unsigned)(reinterpret_cast<uintptr_t>(p->Rtp) & 15);
    key.alpha_unit = true;
    s->spec = dmf::select_path(key);
    if ((s->spec.use_v2 || s->spec.use_cm_i8) && !(flags & DMF_INIT_IN_UNIT_RANGE)) {
        // the integer row kernels write alpha_j alpha_l in fixed point on [0, 1]: true of every iterate (columns on the
        // simplex), checked for the caller's starting point -- in place for a host array, by a kernel for a device array
        // (callers that know where their alpha0 comes from say so with DMF_INIT_IN_UNIT_RANGE and skip the round trip)
        bool in_unit = true;
        if (flags & DMF_PTR_DEVICE) {
            double outside = 0.0;
            hipError_t ec = dmf::launch_unit_range_check(alpha0, K * S, ctx->scratch, ctx->scratch + 2048, ctx->stream);
            if (ec == hipSuccess) ec = hipMemcpyAsync(&outside, ctx->scratch + 2048, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
            if (ec == hipSuccess) ec = hipStreamSynchronize(ctx->stream);
            if (ec != hipSuccess) {
                delete s;
                return hip_fail(ec, "check alpha0", __LINE__);
            }
            in_unit = outside == 0.0;
        } else {
            for (int64_t i = 0; i < K * S; ++i)
                if (!(alpha0[i] >= 0.0 && alpha0[i] <= 1.0)) {
                    in_unit = false;
                    break;
                }
        }
        if (!in_unit) {
            key.alpha_unit = false;
            s->spec = dmf::select_path(key);
        }
    }
    if (!s->spec.supported) {
        delete s;
        return DMF_ERR_UNSUPPORTED;
    }
    // job table of the per-iteration part of the packed Gram: every (k, l) that involves u
    std::vector<short>&hk = s->h_job_k, &hl = s->h_job_l;
    std::vector<int>& hd = s->h_job_dst;
    for (int l = (int)n_c; l <= (int)K; ++l)
        for (int k = 0; k <= l; ++k) {
            if (l == (int)K && k < (int)n_c) continue;  // b of the known types is constant
            if (l == (int)K && k == (int)K) continue;   // v^T D v is constant
            hk.push_back((short)k);
            hl.push_back((short)l);
            hd.push_back(dmf::tri(k, l));
        }
    s->n_jobs = (int)hk.size();
    s->slab_doubles = dmf::gram_slab_doubles(N, (int)S, s->n_jobs);
    if (s->spec.use_gram_spec) {
        const int64_t spec = dmf::gram_u_slab_doubles(N, (int)S, (int)n_c, (int)n_u);
        if (spec > s->slab_doubles) s->slab_doubles = spec;
    }
    if (s->spec.use_gram_mfma) {
        const int64_t need = dmf::gram_mfma_slab_doubles(N, (int)S, s->n_jobs);
        if (need > s->slab_doubles) s->slab_doubles = need;
    }
    if (s->spec.use_fused) {
        const int64_t spec = dmf::rowpass_fused_slab_doubles(N - (N & 15), (int)S, (int)n_c, (int)n_u) +
                             dmf::gram_u_slab_doubles(16, (int)S, (int)n_c, (int)n_u);  // + ragged tail rows
        if (spec > s->slab_doubles) s->slab_doubles = spec;
    }
    if (s->spec.use_v2) {
        const int64_t bu = (int64_t)dmf::rowpass_v2_grid(N, (int)S) * n_u * S;
        if (bu > s->slab_doubles) s->slab_doubles = bu;
    }
    if (s->spec.use_gram_i8) {
        const int64_t bu = (int64_t)dmf::bu_cols_grid(N) * n_u * S;
        if (bu > s->slab_doubles) s->slab_doubles = bu;
        if (s->spec.use_cm_i8) {  // k_inner_bu writes one slab per workgroup
            const int64_t bu2 = (int64_t)dmf::u_inner_bu_grid(N, (int)S) * n_u * S;
            if (bu2 > s->slab_doubles) s->slab_doubles = bu2;
        }
    }
    const size_t un = (size_t)N * n_u * sizeof(double), an = (size_t)K * S * sizeof(double);
    const size_t un_alloc = (un + 15) & ~(size_t)15;  // the integer Gram kernel fetches u in 16-byte pieces
    const size_t gbn = (size_t)(K + 1) * (K + 2) / 2 * S * sizeof(double);
    const int nb_alpha = (int)((S + 63) / 64);
    hipError_t e = pool_alloc(ctx, (void**)&s->u, un_alloc);
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->u_prev, un_alloc);
    if (e == hipSuccess && s->spec.u_path == 2) e = pool_alloc(ctx, (void**)&s->u_next, un_alloc);
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->alpha, an);
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->alpha_prev, an);
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->gb, gbn);
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->slab, (size_t)s->slab_doubles * sizeof(double));
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->partials, (size_t)2 * (nb_alpha + S) * sizeof(double));
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->u2_partials, 4096 * sizeof(double));
    if (e == hipSuccess && (s->spec.use_v2 || s->spec.use_gram_i8)) {
        s->slab_i8_words = dmf::gram_i8_slab_words(N, p->SD, (int)n_c, (int)n_u);
        e = pool_alloc(ctx, (void**)&s->slab_i8, (size_t)s->slab_i8_words * sizeof(long long));
    }
    if (e == hipSuccess && (s->spec.use_v2 || s->spec.use_gram_i8)) {
        const size_t bytes = (size_t)dmf::gram_i8_acc_words((int)S, (int)n_c, (int)n_u) * sizeof(long long);
        e = pool_alloc(ctx, (void**)&s->acc_i8, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(s->acc_i8, 0, bytes, ctx->stream);
    }
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->state, sizeof(SolverState));
    if (e == hipSuccess) {
        if (!ctx->pinned_states.empty()) {
            s->h_state = (SolverState*)ctx->pinned_states.back();
            ctx->pinned_states.pop_back();
        } else {
            e = hipHostMalloc((void**)&s->h_state, kPinnedStateBytes);  // (state mirror + the cost slot of cost_begin / cost_end)
        }
    }
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->job_k, s->n_jobs * sizeof(short));
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->job_l, s->n_jobs * sizeof(short));
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->job_dst, s->n_jobs * sizeof(int));
    const hipMemcpyKind in_kind = (flags & DMF_PTR_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (e == hipSuccess) e = hipMemsetAsync(s->state, 0, sizeof(SolverState), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->gb, 0, gbn, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->u, u0, un, in_kind, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->u_prev, s->u, un, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->alpha, alpha0, an, in_kind, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->alpha_prev, s->alpha, an, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->job_k, hk.data(), s->n_jobs * sizeof(short), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->job_l, hl.data(), s->n_jobs * sizeof(short), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->job_dst, hd.data(), s->n_jobs * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = dmf::launch_scatter_known_block(p->gb_known, s->gb, (int)n_c, (int)K, (int)S, ctx->stream);
    if (e == hipSuccess) e = dmf::launch_sumsq_f64(s->u, N * n_u, ctx->scratch, &s->state->u_norm2, nullptr, ctx->stream);
    // (the cost before the loop, deconvolution.py:204: when a stop test needs it -- dmf_solver_step)
    if (e == hipSuccess) e = dmf::launch_init_state(s->state, p->consts, s->alpha, (int)S, (int)n_c, (int)n_u, ctx->stream);
    s->confirm_stops = false;  // (set per step() call from tol: dmf_solver_step)
    // Host arrays belong to the caller again when this returns: wait for the copies out of them.  (Device arrays -- the
    // restart loops' staged uploads -- are read in stream order; their release is stream-ordered too: dmf_stage_free.)
    if (e == hipSuccess && !(flags & DMF_PTR_DEVICE)) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        dmf_solver_destroy(s);
        return hip_fail(e, "solver set-up", __LINE__);
    }
    *out = s;
    return DMF_OK;
}

constexpr int kMomRows = 64, kMomSteps = 64;  // momentum rows per upload (= the largest batch), inner steps a row can hold

__global__ void k_set_momentum(SolverState* state, const double* mom, int stride, int rows, int n) {
    state->mom = mom;
    state->mom_stride = stride;
    state->mom_rows = rows;
    state->mom_i = 0;
    state->mom_n = n;
}

// Runs the momentum recurrence (deconvolution.py:83-84 / :95-96: a <- (1 + sqrt(1 + 4 a^2)) / 2, and the ratio
// (a_old - 1) / a that beta is the minimum of) ahead for `rows` outer iterations of n inner steps each, from the
// solver's current (a1, a2), and uploads the rows; the kernels of those iterations read them instead of running the
// recurrence themselves.  Plain IEEE double arithmetic, as numpy's in the reference (no contraction on the host).
static double momentum_advance(double& a) {  // a <- next a; returns (a_old - 1) / a_new
#pragma clang fp contract(off)
    const double a0 = a;
    const double sq = 4.0 * a0 * a0;
    a = (1.0 + std::sqrt(1.0 + sq)) / 2.0;
    return (a0 - 1.0) / a;
}

static int upload_momentum_rows(dmf_solver* s, int rows, int n) {
    dmf_context* ctx = s->ctx;
    const int stride = 2 + 2 * n;
    if (s->mom_host == nullptr) {
        if (!ctx->pinned_moms.empty()) {
            s->mom_host = ctx->pinned_moms.back();
            ctx->pinned_moms.pop_back();
        } else {
            HIP_TRY(hipHostMalloc((void**)&s->mom_host, (size_t)kMomRows * (2 + 2 * kMomSteps) * sizeof(double)));
        }
        HIP_TRY(pool_alloc(ctx, (void**)&s->mom_dev, (size_t)kMomRows * (2 + 2 * kMomSteps) * sizeof(double)));
    }
    double a1 = s->h_state->a1, a2 = s->h_state->a2;
    for (int r = 0; r < rows; ++r) {
        double* row = s->mom_host + (size_t)r * stride;
        for (int t = 0; t < n; ++t) {
            row[2 + t] = momentum_advance(a1);
            row[2 + n + t] = momentum_advance(a2);
        }
        row[0] = a1;
        row[1] = a2;
    }
    HIP_TRY(hipMemcpyAsync(s->mom_dev, s->mom_host, (size_t)rows * stride * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_set_momentum, dim3(1), dim3(1), 0, ctx->stream, s->state, (const double*)s->mom_dev, stride, rows, n);
    HIP_TRY(hipGetLastError());
    return DMF_OK;
}

// tol and the band factor of this step() call: the closing kernel pauses (done = 2) when |cf - cf_0| < band x tol
// with band > 1, and stops (done = 1) when band == 1
__global__ void k_set_tol(SolverState* state, double tol, double band) {
    state->tol = tol;
    state->band = band;
}

constexpr double kConfirmBand = 10.0;       // Gram-form differences below this multiple of tol are decided on streaming costs
// Bound of the Gram-form cost's absolute error per unit of N S max(D).  Measured at 1e6 x 256, 12 + 4 against the streaming
// cost of the same iterate (tests/test_gpu_stop_test.py): 1.4e-6 at depth 120, 9e-6 at depth 1000, 4e-6 at depth 2500,
// i.e. at most 3.2e-17 N S max(D); thirty times that is the bound.
constexpr double kGramCostRelErr = 1e-15;

// streaming cost of the solver's current iterate (deconvolution.py:15-17) to the host
static int stream_cost_now(dmf_solver* s, double* out) {
    dmf_context* ctx = s->ctx;
    {
        FamilyScope scope(ctx, DMF_KERNEL_COST);
        HIP_TRY(enqueue_cost(ctx, s->p, s->u, s->alpha, (int)s->n_u, ctx->scratch + 1024, ctx->scratch + 3072));
    }
    HIP_TRY(hipMemcpyAsync(out, ctx->scratch + 3072, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DMF_OK;
}

int dmf_solver_step(dmf_solver* s, int64_t n_outer, int64_t n_iter2, double tol,
                    int64_t* iters_done_total, int* converged) {
    if (s == nullptr || n_outer < 0 || n_iter2 < 0 || n_iter2 > (1 << 20)) return DMF_ERR_BAD_ARG;
    dmf_context* ctx = s->ctx;
    DMF_TRY(check_ctx(ctx));
    const dmf_problem* p = s->p;
    // Stops are confirmed with streaming costs where the Gram form's error bound is not far below the threshold.
    const double gram_err = kGramCostRelErr * (double)p->N * (double)p->S * p->h_consts[2];
    s->confirm_stops = tol > 0.0 && ctx->stop_confirmation != 2 && (ctx->stop_confirmation == 1 || gram_err >= tol / 20.0);
    hipLaunchKernelGGL(k_set_tol, dim3(1), dim3(1), 0, ctx->stream, s->state, tol, s->confirm_stops ? kConfirmBand : 1.0);
    HIP_TRY(hipGetLastError());
    if (s->cf_pending && tol > 0.0 && n_outer > 0) {
        // deconvolution.py:204: the cost before the loop, read by the first stop test only (a threshold of zero never fires)
        FamilyScope scope(ctx, DMF_KERNEL_COST);
        HIP_TRY(enqueue_cost(ctx, p, s->u, s->alpha, (int)s->n_u, ctx->scratch + 1024, &s->state->cf));
        s->cf_pending = false;
        s->cf_stream_iter = -2;  // (marks: state->cf of iteration 0 IS a streaming cost; resolved at the first fetch below)
    }
    DMF_TRY(fetch_state(s));
    if (s->cf_stream_iter == -2) {
        s->cf_stream = s->h_state->cf;
        s->cf_stream_iter = s->h_state->iters;
    }
    // The device freezes the iterate once the stop test fires (every kernel checks state->done),
    // so the host may run ahead by `check_every` enqueued iterations without overshooting.
    // The batches double (8, 16, 32, 64): a solve that runs for hundreds of iterations reads the state back a handful of
    // times, and what a late stop costs is a few dozen no-op launches.
    int64_t check_every = s->spec.u_path != 2 ? 8 : 1;
    // (a threshold of zero never fires -- |cf - cf_0| < 0 -- so a fixed-work run needs no look at the state in between)
    if (tol == 0.0 && s->spec.u_path != 2) check_every = kMomRows;
    const long long iters0 = s->h_state->iters;
    // momentum rows for the kernels that read them (the one-launch row pass and the DPP alpha kernel)
    const dmf::IterationPlan plan = dmf::plan_iteration(s->key, s->spec, (int)n_iter2, s->purity != nullptr);
    const bool use_mom = n_iter2 >= 1 && n_iter2 <= kMomSteps && plan.row == dmf::RowKind::RowpassV2 &&
                         plan.alpha == dmf::AlphaKind::PhaseRow16;
    if (!use_mom && s->h_state->mom_n >= 0) {  // (rows of an earlier call with another n_iter2 / another path: off)
        hipLaunchKernelGGL(k_set_momentum, dim3(1), dim3(1), 0, ctx->stream, s->state, (const double*)nullptr, 0, 0, -1);
        HIP_TRY(hipGetLastError());
    }
    while (s->h_state->iters - iters0 < n_outer && s->h_state->done != 1) {
        const int64_t left = n_outer - (s->h_state->iters - iters0);
        const int64_t batch = left < check_every ? left : check_every;
        if (use_mom) DMF_TRY(upload_momentum_rows(s, (int)batch, (int)n_iter2));  // (from h_state's a1 / a2: just fetched)
        for (int64_t b = 0; b < batch; ++b) DMF_TRY(enqueue_outer_iteration(s, (int)n_iter2));
        DMF_TRY(fetch_state(s));
        if (s->h_state->iters > iters0) s->cf_pending = false;  // (state->cf is the loop's cost from now on)
        if (s->h_state->done == 2) {
            // Paused inside the band at iteration h_state->iters (launches enqueued behind it were no-ops): decide
            // |cf - cf_0| < tol on the streaming costs of this and the previous iterate -- the reference's own formula.
            // The previous one is known when that iteration paused too (or was the starting point); the first
            // iteration inside the band has only the Gram form to go by.
            double cs = 0.0;
            DMF_TRY(stream_cost_now(s, &cs));
            bool stop;
            if (s->cf_stream_iter == s->h_state->iters - 1) {
                stop = std::fabs(cs - s->cf_stream) < tol;
                s->n_confirmed += 1;
            } else {
                stop = std::fabs(s->h_state->cf - s->h_state->cf_prev) < tol;
                s->n_unconfirmed += 1;
            }
            s->cf_stream = cs;
            s->cf_stream_iter = s->h_state->iters;
            s->h_state->done = stop ? 1 : 0;
            DMF_TRY(push_state(s));
            check_every = 1;  // (stay close: the next iterations are likely to pause again)
        } else if (s->spec.u_path != 2 && check_every < 64 && check_every > 1) {
            check_every *= 2;
        }
    }
    if (iters_done_total) *iters_done_total = s->h_state->iters;
    if (converged) *converged = s->h_state->done == 1;
    return DMF_OK;
}

int dmf_solver_set_purity(dmf_solver* s, const double* purity, int flags) {
    if (s == nullptr || purity == nullptr) return DMF_ERR_BAD_ARG;
    dmf_context* ctx = s->ctx;
    DMF_TRY(check_ctx(ctx));
    if (s->mode != DMF_MODE_PARTIAL) return DMF_ERR_BAD_ARG;
    const size_t bytes = (size_t)s->p->S * sizeof(double);
    if (s->purity == nullptr) HIP_TRY(pool_alloc(ctx, (void**)&s->purity, bytes));
    const hipMemcpyKind kind = (flags & DMF_PTR_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    HIP_TRY(hipMemcpyAsync(s->purity, purity, bytes, kind, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DMF_OK;
}

int dmf_solver_get(dmf_solver* s, int flags, double* out_u, double* out_alpha, double* out_cost,
                   int64_t* out_iters) {
    if (s == nullptr) return DMF_ERR_BAD_ARG;
    dmf_context* ctx = s->ctx;
    DMF_TRY(check_ctx(ctx));
    const dmf_problem* p = s->p;
    DMF_TRY(export_array(ctx, s->u, (size_t)p->N * s->n_u * sizeof(double), flags, out_u));
    DMF_TRY(export_array(ctx, s->alpha, (size_t)(p->n_c + s->n_u) * p->S * sizeof(double), flags, out_alpha));
    if (s->cf_pending && out_cost != nullptr) {  // no iteration has run: the cost of the starting point, now
        FamilyScope scope(ctx, DMF_KERNEL_COST);
        HIP_TRY(enqueue_cost(ctx, p, s->u, s->alpha, (int)s->n_u, ctx->scratch + 1024, &s->state->cf));
        s->cf_pending = false;
    }
    DMF_TRY(fetch_state(s));
    if (out_cost) *out_cost = s->h_state->cf;
    if (out_iters) *out_iters = s->h_state->iters;
    return DMF_OK;
}

int dmf_solver_cost(dmf_solver* s, double* out_cost) {
    if (s == nullptr || out_cost == nullptr) return DMF_ERR_BAD_ARG;
    dmf_context* ctx = s->ctx;
    DMF_TRY(check_ctx(ctx));
    const dmf_problem* p = s->p;
    {
        FamilyScope scope(ctx, DMF_KERNEL_COST);
        HIP_TRY(enqueue_cost(ctx, p, s->u, s->alpha, (int)s->n_u, ctx->scratch + 1024, ctx->scratch + 3072));
    }
    HIP_TRY(hipMemcpyAsync(out_cost, ctx->scratch + 3072, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DMF_OK;
}

int dmf_solver_cost_begin(dmf_solver* s) {
    if (s == nullptr) return DMF_ERR_BAD_ARG;
    dmf_context* ctx = s->ctx;
    DMF_TRY(check_ctx(ctx));
    if (s->cost_event == nullptr) {
        if (!ctx->events.empty()) {
            s->cost_event = ctx->events.back();
            ctx->events.pop_back();
        } else {
            HIP_TRY(hipEventCreateWithFlags(&s->cost_event, hipEventDisableTiming));
        }
    }
    double* slot = reinterpret_cast<double*>(reinterpret_cast<char*>(s->h_state) + kPinnedStateBytes - 16);
    {
        FamilyScope scope(ctx, DMF_KERNEL_COST);
        HIP_TRY(enqueue_cost(ctx, s->p, s->u, s->alpha, (int)s->n_u, ctx->scratch + 1024, ctx->scratch + 3072));
    }
    HIP_TRY(hipMemcpyAsync(slot, ctx->scratch + 3072, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipEventRecord(s->cost_event, ctx->stream));
    s->cost_pending = true;
    return DMF_OK;
}

int dmf_solver_cost_end(dmf_solver* s, double* out_cost) {
    if (s == nullptr || out_cost == nullptr || !s->cost_pending) return DMF_ERR_BAD_ARG;
    DMF_TRY(check_ctx(s->ctx));
    HIP_TRY(hipEventSynchronize(s->cost_event));
    s->cost_pending = false;
    *out_cost = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(s->h_state) + kPinnedStateBytes - 16);
    return DMF_OK;
}

int dmf_solver_destroy(dmf_solver* s) {
    if (s == nullptr) return DMF_OK;
    dmf_context* ctx = s->ctx;
    hipSetDevice(s->ctx->device);
    // The solver's buffers go back to the context's pool / free lists, whose reuse is ordered on the context's stream, and
    // its page-locked blocks to the next solver: what must not be in flight is a transfer INTO or OUT OF those blocks.
    if (s->cost_pending) (void)hipEventSynchronize(s->cost_event);
    if (s->in_flight) (void)hipStreamSynchronize(s->ctx->stream);
    if (s->cost_event) ctx->events.push_back(s->cost_event);
    pool_free(ctx, s->u);
    pool_free(ctx, s->u_prev);
    pool_free(ctx, s->u_next);
    pool_free(ctx, s->cm);
    pool_free(ctx, s->beta_tab);
    pool_free(ctx, s->alpha);
    pool_free(ctx, s->alpha_prev);
    pool_free(ctx, s->gb);
    pool_free(ctx, s->slab);
    pool_free(ctx, s->slab_i8);
    pool_free(ctx, s->acc_i8);
    pool_free(ctx, s->partials);
    pool_free(ctx, s->u2_partials);
    pool_free(ctx, s->purity);
    pool_free(ctx, s->state);
    pool_free(ctx, s->mom_dev);
    if (s->mom_host) ctx->pinned_moms.push_back(s->mom_host);
    if (s->h_state) ctx->pinned_states.push_back(s->h_state);
    pool_free(ctx, s->job_k);
    pool_free(ctx, s->job_l);
    pool_free(ctx, s->job_dst);
    delete s;
    return DMF_OK;
}

int dmf_solver_describe(const dmf_solver* s, int64_t n_iter2, char* buf, int64_t cap) {
    if (s == nullptr || buf == nullptr || cap < 1 || n_iter2 < 0) return DMF_ERR_BAD_ARG;
    const dmf::IterationPlan plan = dmf::plan_iteration(s->key, s->spec, (int)n_iter2, s->purity != nullptr);
    dmf::describe_plan(s->key, plan, buf, (size_t)cap);
    return DMF_OK;
}

int dmf_solver_stop_info(const dmf_solver* s, int* confirm_stops, int64_t* n_confirmed, int64_t* n_unconfirmed,
                         double* last_stream_cost) {
    if (s == nullptr) return DMF_ERR_BAD_ARG;
    if (confirm_stops) *confirm_stops = s->confirm_stops ? 1 : 0;
    if (n_confirmed) *n_confirmed = s->n_confirmed;
    if (n_unconfirmed) *n_unconfirmed = s->n_unconfirmed;
    if (last_stream_cost) *last_stream_cost = s->cf_stream_iter >= 0 ? s->cf_stream : std::nan("");
    return DMF_OK;
}

int dmf_select_describe(int64_t N, int64_t S, int64_t n_c, int64_t n_u, int nd, int level, int64_t n_iter2, int flags,
                        char* buf, int64_t cap) {
    if (buf == nullptr || cap < 1 || N < 1 || S < 1 || n_c < 0 || n_u < 1 || nd < 0 || nd > 2 || n_iter2 < 0 ||
        n_c + n_u > dmf::kMaxK)
        return DMF_ERR_BAD_ARG;
    dmf::ShapeKey key;
    key.N = N;
    key.S = (int)S;
    key.n_c = (int)n_c;
    key.n_u = (int)n_u;
    key.nd = (S <= 2048) ? nd : 0;  // (dmf_problem_create builds no integer copies beyond 2048 samples)
    key.SD = key.nd > 0 ? (int)((S + 63) / 64 * 64) : 0;
    key.level = level;
    key.d_f32_exact = (flags & DMF_SELECT_COUNTS_F32_EXACT) != 0;
    key.rtp_present = true;
    key.v_align = (flags & DMF_SELECT_V_UNALIGNED) ? 8 : 0;
    key.rtp_align = 0;
    key.alpha_unit = !(flags & DMF_SELECT_ALPHA_OUTSIDE_UNIT);
    const dmf::PathSpec spec = dmf::select_path(key);
    if (!spec.supported) return DMF_ERR_UNSUPPORTED;
    const dmf::IterationPlan plan = dmf::plan_iteration(key, spec, (int)n_iter2, (flags & DMF_SELECT_PURITY) != 0);
    dmf::describe_plan(key, plan, buf, (size_t)cap);
    return DMF_OK;
}

int dmf_solve(dmf_context* ctx, const dmf_problem* p, const double* u0, const double* alpha0, int64_t n_u,
              int mode, int64_t n_iter1, int64_t n_iter2, double tol, int flags, double* out_u,
              double* out_alpha, double* out_cost, int64_t* out_iters) {
    dmf_solver* s = nullptr;
    DMF_TRY(dmf_solver_create(ctx, p, u0, alpha0, n_u, mode, flags, &s));
    int st = dmf_solver_step(s, n_iter1, n_iter2, tol, nullptr, nullptr);
    if (st == DMF_OK) st = dmf_solver_get(s, flags, out_u, out_alpha, out_cost, out_iters);
    dmf_solver_destroy(s);
    return st;
}

// ------------------------------------------------------------------------------- single functions
int dmf_cost(dmf_context* ctx, const dmf_problem* p, const double* u, int64_t n_u, const double* alpha,
             int flags, double* out_cost) {
    DMF_TRY(check_ctx(ctx));
    if (p == nullptr || alpha == nullptr || out_cost == nullptr || n_u < 0) return DMF_ERR_BAD_ARG;
    if (n_u > 0 && u == nullptr) return DMF_ERR_BAD_ARG;
    const int64_t K = p->n_c + n_u;
    if (K < 1) return DMF_ERR_BAD_ARG;
    double *du = nullptr, *da = nullptr, *dout = nullptr;
    bool own_u = false, own_a = false;
    int st = import_array(ctx, u, (size_t)p->N * n_u * sizeof(double), flags, (void**)&du, &own_u);
    if (st == DMF_OK) st = import_array(ctx, alpha, (size_t)K * p->S * sizeof(double), flags, (void**)&da, &own_a);
    if (st == DMF_OK) {
        hipError_t e = pool_alloc(ctx, (void**)&dout, sizeof(double));
        if (e == hipSuccess) {
            FamilyScope scope(ctx, DMF_KERNEL_COST);
            e = enqueue_cost(ctx, p, du, da, (int)n_u, ctx->scratch + 1024, dout);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out_cost, dout, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = hip_fail(e, "cost", __LINE__);
    }
    if (own_u) pool_free(ctx, du);
    if (own_a) pool_free(ctx, da);
    pool_free(ctx, dout);
    return st;
}

int dmf_project_simplex(dmf_context* ctx, const double* X, int64_t K, int64_t S, double z, int flags,
                        double* out) {
    DMF_TRY(check_ctx(ctx));
    if (X == nullptr || out == nullptr || K < 1 || S < 1) return DMF_ERR_BAD_ARG;
    if (K > dmf::kMaxK) return DMF_ERR_UNSUPPORTED;
    const size_t bytes = (size_t)K * S * sizeof(double);
    double *dx = nullptr, *dout = nullptr;
    bool own_x = false;
    int st = import_array(ctx, X, bytes, flags, (void**)&dx, &own_x);
    if (st == DMF_OK) {
        hipError_t e = hipSuccess;
        if (flags & DMF_PTR_DEVICE) dout = out;
        else e = pool_alloc(ctx, (void**)&dout, bytes);
        if (e == hipSuccess) e = dmf::launch_project_simplex(dx, dout, (int)K, (int)S, z, ctx->stream);
        if (e == hipSuccess && !(flags & DMF_PTR_DEVICE))
            e = hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = hip_fail(e, "project_simplex", __LINE__);
        if (!(flags & DMF_PTR_DEVICE)) pool_free(ctx, dout);
    }
    if (own_x) pool_free(ctx, dx);
    return st;
}

// numpy's "linear" percentile (numpy/lib/_function_base_impl.py: _quantile, _get_indexes, _get_gamma):
// virtual index (n - 1) * (q / 100), its floor and the next index, gamma = the fractional part
static dmf::PercentilePlan percentile_plan(int64_t n, double q_percent) {
#pragma clang fp contract(off)  // numpy rounds the product before subtracting the floor
    dmf::PercentilePlan pl{};
    const double quantile = q_percent / 100.0;
    const double vi = (double)(n - 1) * quantile;
    if (vi >= (double)(n - 1)) {
        pl.k_prev = pl.k_next = n - 1;
        pl.gamma = 0.0;
    } else if (vi < 0.0) {
        pl.k_prev = pl.k_next = 0;
        pl.gamma = 0.0;
    } else {
        const double fl = std::floor(vi);
        pl.k_prev = (long long)fl;
        pl.k_next = pl.k_prev + 1;
        pl.gamma = vi - fl;
    }
    return pl;
}

int dmf_percentile_axis0(dmf_context* ctx, const double* x, int64_t n, int64_t m, const double* q, int64_t n_q,
                         int flags, double* out) {
    DMF_TRY(check_ctx(ctx));
    if (x == nullptr || q == nullptr || out == nullptr || n < 1 || m < 1 || n_q < 1) return DMF_ERR_BAD_ARG;
    for (int64_t i = 0; i < n_q; ++i)
        if (!(q[i] >= 0.0 && q[i] <= 100.0)) return DMF_ERR_BAD_ARG;  // numpy: "Percentiles must be in the range [0, 100]"
    if (n > dmf::percentile_max_replicates()) return DMF_ERR_UNSUPPORTED;
    const size_t in_bytes = (size_t)n * m * sizeof(double), out_bytes = (size_t)n_q * m * sizeof(double);
    double *dx = nullptr, *dout = nullptr;
    bool own_x = false;
    int st = import_array(ctx, x, in_bytes, flags, (void**)&dx, &own_x);
    if (st == DMF_OK) {
        hipError_t e = hipSuccess;
        if (flags & DMF_PTR_DEVICE) dout = out;
        else e = pool_alloc(ctx, (void**)&dout, out_bytes);
        for (int64_t i = 0; i < n_q && e == hipSuccess; i += 2) {
            const dmf::PercentilePlan p0 = percentile_plan(n, q[i]);
            const bool two = i + 1 < n_q;
            const dmf::PercentilePlan p1 = two ? percentile_plan(n, q[i + 1]) : p0;
            e = dmf::launch_percentile_pair(dx, n, m, p0, p1, dout + i * m, two ? dout + (i + 1) * m : nullptr,
                                            ctx->stream);
        }
        if (e == hipSuccess && !(flags & DMF_PTR_DEVICE))
            e = hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = hip_fail(e, "percentile_axis0", __LINE__);
        if (!(flags & DMF_PTR_DEVICE)) pool_free(ctx, dout);
    }
    if (own_x) pool_free(ctx, dx);
    return st;
}

int dmf_update_u(dmf_context* ctx, const dmf_problem* p, const double* u, const double* u_prev,
                 const double* alpha, int64_t n_u, int64_t n_iter2, int mode, int flags,
                 double* scalars_io, double* out_u, double* out_u_prev) {
    if (u_prev == nullptr || scalars_io == nullptr || out_u == nullptr || out_u_prev == nullptr || n_iter2 < 0)
        return DMF_ERR_BAD_ARG;
    dmf_solver* s = nullptr;
    DMF_TRY(dmf_solver_create(ctx, p, u, alpha, n_u, mode, flags, &s));
    const size_t un = (size_t)p->N * n_u * sizeof(double);
    const hipMemcpyKind in_kind = (flags & DMF_PTR_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    int st = DMF_OK;
    hipError_t e = hipMemcpyAsync(s->u_prev, u_prev, un, in_kind, ctx->stream);
    if (e != hipSuccess) st = hip_fail(e, "copy u_prev", __LINE__);
    if (st == DMF_OK) st = fetch_state(s);
    if (st == DMF_OK) {
        s->h_state->a1 = scalars_io[0];
        s->h_state->l_w_prev = scalars_io[1];
        s->h_state->l_w = scalars_io[2];
        st = push_state(s);
    }
    if (st == DMF_OK) st = enqueue_u_phase(s, (int)n_iter2, standalone_row_kind(s, (int)n_iter2));
    if (st == DMF_OK) st = export_array(ctx, s->u, un, flags, out_u);
    if (st == DMF_OK) st = export_array(ctx, s->u_prev, un, flags, out_u_prev);
    if (st == DMF_OK) {
        scalars_io[0] = advance_momentum(scalars_io[0], n_iter2);
        if (n_iter2 > 0) scalars_io[1] = scalars_io[2];
    }
    dmf_solver_destroy(s);
    return st;
}

int dmf_update_alpha(dmf_context* ctx, const dmf_problem* p, const double* u, int64_t n_u,
                     const double* alpha, const double* alpha_prev, int64_t n_iter2, int flags,
                     double* scalars_io, double* out_alpha, double* out_alpha_prev) {
    if (alpha_prev == nullptr || scalars_io == nullptr || out_alpha == nullptr || out_alpha_prev == nullptr ||
        n_iter2 < 0)
        return DMF_ERR_BAD_ARG;
    dmf_solver* s = nullptr;
    DMF_TRY(dmf_solver_create(ctx, p, u, alpha, n_u, DMF_MODE_PARTIAL, flags, &s));
    const size_t an = (size_t)(p->n_c + n_u) * p->S * sizeof(double);
    const hipMemcpyKind in_kind = (flags & DMF_PTR_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    int st = DMF_OK;
    hipError_t e = hipMemcpyAsync(s->alpha_prev, alpha_prev, an, in_kind, ctx->stream);
    if (e != hipSuccess) st = hip_fail(e, "copy alpha_prev", __LINE__);
    if (st == DMF_OK) st = fetch_state(s);
    if (st == DMF_OK) {
        s->h_state->a2 = scalars_io[0];
        s->h_state->l_h_prev = scalars_io[1];
        s->h_state->l_h = scalars_io[2];
        st = push_state(s);
    }
    if (st == DMF_OK) st = enqueue_gram(s, fp64_gram_kind(s));  // (the caller's u: FP64 kernels)
    if (st == DMF_OK) st = enqueue_alpha_phase(s, (int)n_iter2);
    if (st == DMF_OK) st = export_array(ctx, s->alpha, an, flags, out_alpha);
    if (st == DMF_OK) st = export_array(ctx, s->alpha_prev, an, flags, out_alpha_prev);
    if (st == DMF_OK) {
        scalars_io[0] = advance_momentum(scalars_io[0], n_iter2);
        if (n_iter2 > 0) scalars_io[1] = scalars_io[2];
    }
    dmf_solver_destroy(s);
    return st;
}

}  // extern "C"
