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
    int generic_level = 0;  // 0 fused row pass, 1 any-shape Gram-form kernels, 2 schedule-faithful u steps,
                            // 3 separate MFMA row pass + one-pass Gram (the pieces the fused kernel is made of)
    double* scratch = nullptr;  // 4096 doubles of reduction scratch
    hipMemPool_t pool = nullptr;  // the context's own stream-ordered pool (the device's default pool is not touched)
    std::unordered_map<void*, size_t> live;                // large blocks handed out by pool_alloc (size by address)
    std::unordered_map<size_t, std::vector<void*>> kept;    // freed large blocks kept for the next allocation of that size
    size_t kept_bytes = 0;
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

struct dmf_solver {
    dmf_context* ctx = nullptr;
    const dmf_problem* p = nullptr;
    int64_t n_u = 0;
    int mode = 0;
    int u_path = 0;         // 0 MFMA, 1 Gram-form VALU, 2 schedule-faithful direct steps
    bool use_gram_spec = false;
    bool use_gram_mfma = false;
    bool use_u_big = false;      // 9 <= n_u <= 26: matrix-core u phase with M_i in LDS
    bool use_cm_i8 = false;      // 5 <= n_u <= 16 with u16 counts: split u phase, M_i on the integer matrix cores
    double* cm = nullptr;        // split u phase (many inner steps): per-row c_i / M_i, allocated on first use
    double* beta_tab = nullptr;  //   and the momentum coefficients of the inner steps
    int64_t beta_cap = 0;  // shapes beyond the lane-per-sample kernel's registers: MFMA Gram
    bool use_fused = false;      // first-generation fused row pass (counts as f64 in HBM, FP64 Gram in the kernel)
    bool use_v2 = false;         // second generation: u16 counts in the row pass + integer-MFMA Gram
    bool use_gram_i8 = false;    // u phase as a kernel of its own (n_u > 4 ...), Gram on the integer matrix cores + k_bu_cols
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
                hipEventCreate(&c->start[i]);
                hipEventCreate(&c->stop[i]);
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
    int64_t slab_doubles = dmf::gram_slab_doubles(N, (int)S, mfma ? 1 : n_jobs);
    if (mfma) {
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
    if (mfma) {
        dmf::GramJobTable fast{dk, dl, dd, n_fast};
        int ny = 0;
        e = dmf::launch_gram_mfma(p->V, p->D, p->Rt, nullptr, N, (int)S, (int)n_c, 0, fast, n_dense, slab,
                                  slab_doubles, nullptr, &ny, ctx->stream);
        if (e == hipSuccess)
            e = dmf::launch_gram_reduce(slab, ny, n_fast, (int)S, dd, p->gb_known, nullptr, ctx->stream);
    }
    if (e == hipSuccess && n_jobs - n_fast == 1 && ctx->generic_level != 1 && ctx->generic_level != 2 &&
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

constexpr int kSplitInnerSteps = 50;  // beyond this the unfused / split u phase beats the fused kernel

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

int enqueue_u_phase(dmf_solver* s, int n_iter2) {
    dmf_context* ctx = s->ctx;
    const dmf_problem* p = s->p;
    FamilyScope scope(ctx, DMF_KERNEL_ROWPASS);
    if (s->use_cm_i8) {
        // wide row groups on u16 counts: per-row c_i / M_i with M_i on the integer matrix cores, then the inner iterations
        // chip-wide (dmf_kernels_cm_i8.hip)
        DMF_TRY(ensure_split_scratch(s, n_iter2));
        HIP_TRY(dmf::launch_u_phase_split_i8(p->V, p->D16, p->SD, p->ND, p->Rtp, s->alpha, s->u, s->u_prev, s->state, p->N,
                                             (int)p->S, (int)p->n_c, (int)s->n_u, n_iter2, s->mode, s->cm, s->beta_tab,
                                             ctx->stream));
        return DMF_OK;
    }
    if (s->use_u_big && dmf::u_phase_big_supported((int)p->S, (int)p->n_c, (int)s->n_u, n_iter2)) {
        HIP_TRY(dmf::launch_u_phase_big(p->V, p->D, p->Rtp, s->alpha, s->u, s->u_prev, s->state, p->N, (int)p->S,
                                        (int)p->n_c, (int)s->n_u, n_iter2, s->mode, ctx->stream));
        return DMF_OK;
    }
    // The split form also wins at few inner steps once the row groups are wide: with 7 or 8 unknowns the inner steps
    // inside k_u_phase_mfma broadcast through ds_bpermute on ONE wave per workgroup and the kernel spills (measured at
    // 5e5 x 128, 20 steps: 0+8 0.70 -> 0.57 ms, 12+6 0.62 -> 0.48; 0+5 equal, 0+6 0.37 -> 0.39): split from 7 unknowns
    // on, and from 5 when there are known types (their E product already fills the row kernel).
    // DMF_SPLIT_NU=n moves the threshold (experiments).
    static const int split_nu = [] {
        const char* v = getenv("DMF_SPLIT_NU");
        return v != nullptr && atoi(v) > 0 ? atoi(v) : 7;
    }();
    if (s->u_path == 0 && (n_iter2 > kSplitInnerSteps || (int)s->n_u >= split_nu || (p->n_c > 0 && s->n_u >= 5))) {
        // many inner steps: one wave per workgroup running them is the bottleneck (see enqueue_outer_iteration)
        DMF_TRY(ensure_split_scratch(s, n_iter2));
        HIP_TRY(dmf::launch_u_phase_split(p->V, p->D, p->D16, p->SD, p->Rtp, s->alpha, s->u, s->u_prev, s->state, p->N, (int)p->S,
                                          (int)p->n_c, (int)s->n_u, n_iter2, s->mode, s->cm, s->beta_tab, ctx->stream));
        return DMF_OK;
    }
    if (s->u_path == 0) {
        HIP_TRY(dmf::launch_u_phase_mfma(p->V, p->D, p->D16, p->SD, p->Rtp, s->alpha, s->u, s->u_prev, s->state, p->N,
                                         (int)p->S, (int)p->n_c, (int)s->n_u, n_iter2, s->mode, ctx->stream));
    } else if (s->u_path == 1) {
        HIP_TRY(dmf::launch_u_phase_gram(p->V, p->D, p->Rt, s->alpha, s->u, s->u_prev, s->state, p->N,
                                         (int)p->S, (int)p->n_c, (int)s->n_u, n_iter2, s->mode, ctx->stream));
    } else {
        for (int t = 0; t < n_iter2; ++t) {
            HIP_TRY(dmf::launch_u_step_direct(p->V, p->D, p->Rt, s->alpha, s->u, s->u_prev, s->u_next,
                                              s->state, p->N, (int)p->S, (int)p->n_c, (int)s->n_u, t,
                                              s->mode, ctx->stream));
            double* old_prev = s->u_prev;
            s->u_prev = s->u;
            s->u = s->u_next;
            s->u_next = old_prev;
        }
    }
    return DMF_OK;
}

int enqueue_gram(dmf_solver* s, bool after_u_phase = false) {
    dmf_context* ctx = s->ctx;
    const dmf_problem* p = s->p;
    FamilyScope scope(ctx, DMF_KERNEL_GRAM);
    if (s->use_gram_i8 && after_u_phase) {
        // (only behind a u phase: its clip puts u inside [0, 1], which the fixed-point features need; the
        // dmf_update_alpha entry point hands over the caller's u and stays on the FP64 kernels)
        const int S = (int)p->S, n_c = (int)p->n_c, n_u = (int)s->n_u, nf = n_c * n_u + n_u * (n_u + 1) / 2;
        int n_slabs = 0, ny = 0;
        HIP_TRY(dmf::launch_bu_cols(p->V, p->D16, p->SD, s->u, p->N, S, n_u, s->slab, &s->state->done, &n_slabs, ctx->stream));
        HIP_TRY(dmf::launch_gram_i8(p->Dt8, p->plane_stride, p->SD, p->ND, p->Rtp, s->u, p->N, n_c, n_u, s->job_k, s->job_l, nf,
                                    s->slab_i8, s->slab_i8_words, &s->state->done, &ny, ctx->stream));
        HIP_TRY(dmf::launch_gram_v2_reduce(s->slab_i8, ny, nf, p->SD, s->slab, n_slabs, n_u, S, s->acc_i8, s->job_dst, s->gb,
                                           &s->state->done, nullptr, 0, s->state, ctx->stream));
        return DMF_OK;
    }
    if (s->use_gram_spec) {
        int ny = 0;
        HIP_TRY(dmf::launch_gram_u(p->V, p->D, p->Rtp, s->u, p->N, (int)p->S, (int)p->n_c, (int)s->n_u, s->slab,
                                   &s->state->done, &ny, ctx->stream));
        HIP_TRY(dmf::launch_gram_reduce(s->slab, ny, s->n_jobs, (int)p->S, s->job_dst, s->gb, &s->state->done,
                                        ctx->stream));
        return DMF_OK;
    }
    dmf::GramJobTable jobs{s->job_k, s->job_l, s->job_dst, s->n_jobs};
    if (s->use_gram_mfma) {
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
    const dmf_problem* p = s->p;
    // With many inner steps the row-local iterations (one wave per workgroup in the fused kernel) dominate and
    // the unfused u-phase kernel, which keeps three workgroups per CU busy, wins: measured at the headline size
    // with n_iter2 = 500 (the CLI default under --purity) 18.5 ms fused against 7.6 + 1.2 ms; the estimated
    // break-even is around 50 inner steps.
    if (s->use_v2 && n_iter2 <= kSplitInnerSteps &&
        dmf::rowpass_v2_supported((int)p->S, (int)p->n_c, (int)s->n_u, n_iter2)) {
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
    if (s->use_cm_i8 && s->use_gram_i8 &&
        dmf::u_inner_bu_supported(p->V, (int)p->S, p->SD, (int)s->n_u, n_iter2)) {
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
    if (s->use_fused && n_iter2 <= kSplitInnerSteps) {
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
    DMF_TRY(enqueue_u_phase(s, n_iter2));
    HIP_TRY(dmf::launch_sumsq_f64(s->u, p->N * s->n_u, ctx->scratch, &s->state->u_norm2, &s->state->done,
                                  ctx->stream));
    HIP_TRY(dmf::launch_set_lh(s->state, ctx->stream));
    DMF_TRY(enqueue_gram(s, true));
    DMF_TRY(enqueue_alpha_phase(s, n_iter2));
    return DMF_OK;
}

int fetch_state(dmf_solver* s) {
    HIP_TRY(hipMemcpyAsync(s->h_state, s->state, sizeof(SolverState), hipMemcpyDeviceToHost, s->ctx->stream));
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
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

int dmf_problem_gather(dmf_context* ctx, const dmf_problem* src, const int64_t* idx, int64_t n_idx,
                       dmf_problem** out) {
    DMF_TRY(check_ctx(ctx));
    if (src == nullptr || idx == nullptr || out == nullptr || n_idx <= 0) return DMF_ERR_BAD_ARG;
    *out = nullptr;
    for (int64_t r = 0; r < n_idx; ++r)
        if (idx[r] < 0 || idx[r] >= src->N) return DMF_ERR_BAD_ARG;
    dmf_problem* p = new (std::nothrow) dmf_problem();
    if (p == nullptr) return DMF_ERR_BAD_ARG;
    p->ctx = ctx;
    p->N = n_idx;
    p->S = src->S;
    p->n_c = src->n_c;
    long long* d_idx = nullptr;
    int st = DMF_OK;
    hipError_t e = pool_alloc(ctx, (void**)&d_idx, (size_t)n_idx * sizeof(long long));
    if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, (size_t)n_idx * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
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
    pool_free(ctx, d_idx);
    if (e != hipSuccess) st = hip_fail(e, "row gather", __LINE__);
    if (st == DMF_OK) st = problem_finalize(p, counts_done);
    if (st != DMF_OK) {
        dmf_problem_destroy(p);
        return st;
    }
    *out = p;
    return DMF_OK;
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
    const bool fast = ctx->generic_level == 0 || ctx->generic_level == 3 || ctx->generic_level == 4;
    if (fast && dmf::u_phase_mfma_supported((int)S, (int)n_c, (int)n_u)) s->u_path = 0;
    else if (ctx->generic_level != 2 && dmf::u_phase_gram_supported((int)S, (int)n_c, (int)n_u)) s->u_path = 1;
    else s->u_path = 2;
    s->use_gram_spec = fast && dmf::gram_u_supported((int)n_c, (int)n_u);
    s->use_gram_mfma = fast && !s->use_gram_spec;
    s->use_u_big = fast && s->u_path != 0 && dmf::u_phase_big_supported((int)S, (int)n_c, (int)n_u, 64);
    s->use_v2 = ctx->generic_level == 0 && p->ND > 0 && p->D16 != nullptr && (n_c == 0 || p->Rtp != nullptr) &&
                (reinterpret_cast<uintptr_t>(p->V) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->Rtp) & 15) == 0 &&
                dmf::rowpass_v2_supported((int)S, (int)n_c, (int)n_u, 20) &&
                dmf::gram_i8_supported((int)n_c, (int)n_u, p->ND, N, p->SD);
    // Wide row groups (n_u 5..16) on u16 counts: the split u phase with the integer-matrix-core producer.  Measured at
    // 5e5 x 128 against what ran before: see DESIGN.md section 5.  DMF_CM_I8_MIN_NU moves the lower end (experiments).
    static const int cm_min_nu = [] { const char* v = getenv("DMF_CM_I8_MIN_NU"); return v != nullptr && atoi(v) > 0 ? atoi(v) : 5; }();
    // (narrow row groups reach it beyond the row pass's 512 samples -- the producer walks panels of 256 samples -- and
    // with more than 16 known types)
    s->use_cm_i8 = ctx->generic_level == 0 && !s->use_v2 && p->ND > 0 && p->D16 != nullptr && (n_u >= cm_min_nu || S > 512 || n_c > 16) && n_u <= 32 &&
                   (n_c == 0 || (p->Rtp != nullptr && (reinterpret_cast<uintptr_t>(p->Rtp) & 7) == 0)) &&
                   dmf::cm_i8_supported(p->V, (int)S, (int)n_c, (int)n_u, p->ND, p->SD);
    // shapes the second-generation row pass does not take (n_u 5..20, long inner loops): the u phase stays a kernel of its
    // own, the Gram pass becomes the integer GEMM + the b_u stream kernel (V f64 + u16 counts instead of V and D f64
    // and n_u (n_u + 3) / 2 + n_c n_u FP64 FMAs per element)
    // Measured at 5e5 x 128 (tools/gram_i8_vs_fp64.py): 6+6 (57 features) 0.25 against 0.35 ms for k_gram_u, but 0+5 / 0+8 /
    // 0+12 0.25 / 0.25 / 0.34 against 0.19 / 0.22 / 0.31 ms -- without known types the FP64 Gram is cheap and the four
    // launches of this route are not; so: only with known types and at least 40 features.
    // What the integer route competes with is k_gram_u, whose time grows with its accumulator count (padded known types
    // x unknowns + pairs + b_u) while the integer route is flat (k_bu_cols dominates it): at 5e5 x 128 the FP64 kernel
    // takes 0.21 ms with 51 accumulators (2+6, 4+6), 0.23 with 40 (1+5, 3+5), 0.36 with 60..76 (5+5, 1+8, 3+8), the
    // integer route 0.20..0.21 throughout with two samples per lane in k_bu_cols2 (tools/gram_i8_vs_fp64.py): from 48
    // accumulators on.  DMF_GRAM_I8_MIN moves the threshold (experiments).
    static const int i8_min_features = [] { const char* v = getenv("DMF_GRAM_I8_MIN"); return v != nullptr ? atoi(v) : 48; }();
    const int fp64_acc = (int)((n_c + 3) / 4 * 4 * n_u + n_u * (n_u + 1) / 2 + n_u);
    static const int i8_min_nc0 = [] { const char* v = getenv("DMF_GRAM_I8_MIN_NC0"); return v != nullptr ? atoi(v) : 33; }();
    // (without known types: from 8 unknowns -- 36 features -- on.  At 5e5 x 128 with k_bu_cols2 up to 16 unknowns:
    // 0+8 0.24 -> 0.20 ms, 0+12 0.36 -> 0.29, 0+16 0.67 (k_gram_mfma) -> 0.36; below 8 k_gram_u is cheaper.)
    // Behind k_cm_i8 the b_u stream rides along with the inner iterations (k_inner_bu) and the integer route is what is
    // left of the Gram pass: it then wins at every width (0+5 / 0+6 / 0+7 at 5e5 x 128: 0.58 / 0.55 / 0.59 -> 0.51 / 0.49 /
    // 0.53 ms per iteration against k_gram_u).
    const bool fused_bu = s->use_cm_i8 && dmf::u_inner_bu_supported(p->V, (int)S, p->SD, (int)n_u, 20);
    const bool known_ok = n_c > 0 ? (p->Rtp != nullptr && (reinterpret_cast<uintptr_t>(p->Rtp) & 15) == 0) : true;
    s->use_gram_i8 = ctx->generic_level == 0 && p->ND > 0 && p->D16 != nullptr && known_ok &&
                     (fused_bu || (n_c > 0 ? fp64_acc >= i8_min_features : n_u * (n_u + 1) / 2 >= i8_min_nc0)) && n_u <= 32 &&
                     dmf::gram_i8_supported((int)n_c, (int)n_u, p->ND, N, p->SD);
    if (s->use_v2 || s->use_cm_i8) {
        // the row pass writes alpha_j alpha_l in fixed point on [0, 1]: true of every iterate (columns on the simplex),
        // checked for the caller's starting point
        std::vector<double> ha((size_t)K * S);
        hipError_t ec = hipMemcpyAsync(ha.data(), alpha0, ha.size() * sizeof(double),
                                       (flags & DMF_PTR_DEVICE) ? hipMemcpyDeviceToHost : hipMemcpyHostToHost, ctx->stream);
        if (ec == hipSuccess) ec = hipStreamSynchronize(ctx->stream);
        if (ec != hipSuccess) {
            delete s;
            return hip_fail(ec, "copy alpha0", __LINE__);
        }
        for (double a : ha)
            if (!(a >= 0.0 && a <= 1.0)) {
                s->use_v2 = false;
                s->use_cm_i8 = false;
                break;
            }
    }
    s->use_fused = (ctx->generic_level == 0 || ctx->generic_level == 4) && p->d_f32_exact && N >= 16 &&
                   dmf::rowpass_fused_supported((int)S, (int)n_c, (int)n_u) &&
                   dmf::u_phase_mfma_supported((int)S, (int)n_c, (int)n_u) && dmf::gram_u_supported((int)n_c, (int)n_u);
    if (s->u_path == 2 && !dmf::u_step_direct_supported((int)S, (int)n_c, (int)n_u)) {
        delete s;
        return DMF_ERR_UNSUPPORTED;
    }
    // job table of the per-iteration part of the packed Gram: every (k, l) that involves u
    std::vector<short> hk, hl;
    std::vector<int> hd;
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
    if (s->use_gram_spec) {
        const int64_t spec = dmf::gram_u_slab_doubles(N, (int)S, (int)n_c, (int)n_u);
        if (spec > s->slab_doubles) s->slab_doubles = spec;
    }
    if (s->use_gram_mfma) {
        const int64_t need = dmf::gram_mfma_slab_doubles(N, (int)S, s->n_jobs);
        if (need > s->slab_doubles) s->slab_doubles = need;
    }
    if (s->use_fused) {
        const int64_t spec = dmf::rowpass_fused_slab_doubles(N - (N & 15), (int)S, (int)n_c, (int)n_u) +
                             dmf::gram_u_slab_doubles(16, (int)S, (int)n_c, (int)n_u);  // + ragged tail rows
        if (spec > s->slab_doubles) s->slab_doubles = spec;
    }
    if (s->use_v2) {
        const int64_t bu = (int64_t)dmf::rowpass_v2_grid(N, (int)S) * n_u * S;
        if (bu > s->slab_doubles) s->slab_doubles = bu;
    }
    if (s->use_gram_i8) {
        const int64_t bu = (int64_t)dmf::bu_cols_grid(N) * n_u * S;
        if (bu > s->slab_doubles) s->slab_doubles = bu;
        if (s->use_cm_i8) {  // k_inner_bu writes one slab per workgroup
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
    if (e == hipSuccess && s->u_path == 2) e = pool_alloc(ctx, (void**)&s->u_next, un_alloc);
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->alpha, an);
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->alpha_prev, an);
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->gb, gbn);
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->slab, (size_t)s->slab_doubles * sizeof(double));
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->partials, (size_t)2 * (nb_alpha + S) * sizeof(double));
    if (e == hipSuccess) e = pool_alloc(ctx, (void**)&s->u2_partials, 4096 * sizeof(double));
    if (e == hipSuccess && (s->use_v2 || s->use_gram_i8)) {
        s->slab_i8_words = dmf::gram_i8_slab_words(N, p->SD, (int)n_c, (int)n_u);
        e = pool_alloc(ctx, (void**)&s->slab_i8, (size_t)s->slab_i8_words * sizeof(long long));
    }
    if (e == hipSuccess && (s->use_v2 || s->use_gram_i8)) {
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
            e = hipHostMalloc((void**)&s->h_state, sizeof(SolverState));
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
    if (e == hipSuccess) {
        FamilyScope scope(ctx, DMF_KERNEL_COST);
        e = enqueue_cost(ctx, p, s->u, s->alpha, (int)n_u, ctx->scratch + 1024, &s->state->cf);
    }
    if (e == hipSuccess) e = dmf::launch_init_state(s->state, p->consts, s->alpha, (int)S, (int)n_c, (int)n_u, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // host job vectors go out of scope
    if (e != hipSuccess) {
        dmf_solver_destroy(s);
        return hip_fail(e, "solver set-up", __LINE__);
    }
    *out = s;
    return DMF_OK;
}

__global__ void k_set_tol(SolverState* state, double tol) { state->tol = tol; }

int dmf_solver_step(dmf_solver* s, int64_t n_outer, int64_t n_iter2, double tol,
                    int64_t* iters_done_total, int* converged) {
    if (s == nullptr || n_outer < 0 || n_iter2 < 0 || n_iter2 > (1 << 20)) return DMF_ERR_BAD_ARG;
    dmf_context* ctx = s->ctx;
    DMF_TRY(check_ctx(ctx));
    hipLaunchKernelGGL(k_set_tol, dim3(1), dim3(1), 0, ctx->stream, s->state, tol);
    HIP_TRY(hipGetLastError());
    DMF_TRY(fetch_state(s));
    // The device freezes the iterate once the stop test fires (every kernel checks state->done),
    // so the host may run ahead by `check_every` enqueued iterations without overshooting.
    // The batches double (8, 16, 32, 64): a solve that runs for hundreds of iterations reads the state back a handful of
    // times, and what a late stop costs is a few dozen no-op launches.
    int64_t check_every = s->u_path != 2 ? 8 : 1;
    int64_t enqueued = 0;
    while (enqueued < n_outer && !s->h_state->done) {
        int64_t batch = n_outer - enqueued < check_every ? n_outer - enqueued : check_every;
        for (int64_t b = 0; b < batch; ++b) DMF_TRY(enqueue_outer_iteration(s, (int)n_iter2));
        enqueued += batch;
        DMF_TRY(fetch_state(s));
        if (s->u_path != 2 && check_every < 64) check_every *= 2;
    }
    if (iters_done_total) *iters_done_total = s->h_state->iters;
    if (converged) *converged = s->h_state->done;
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

int dmf_solver_destroy(dmf_solver* s) {
    if (s == nullptr) return DMF_OK;
    dmf_context* ctx = s->ctx;
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
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
    if (s->h_state) ctx->pinned_states.push_back(s->h_state);
    pool_free(ctx, s->job_k);
    pool_free(ctx, s->job_l);
    pool_free(ctx, s->job_dst);
    delete s;
    return DMF_OK;
}

int dmf_solver_describe(const dmf_solver* s, int64_t n_iter2, char* buf, int64_t cap) {
    if (s == nullptr || buf == nullptr || cap < 1) return DMF_ERR_BAD_ARG;
    const dmf_problem* p = s->p;
    const int S = (int)p->S, n_c = (int)p->n_c, n_u = (int)s->n_u, K = n_c + n_u;
    char row[160], gram[64];
    if (s->use_v2 && n_iter2 <= kSplitInnerSteps && dmf::rowpass_v2_supported(S, n_c, n_u, (int)n_iter2)) {
        snprintf(row, sizeof(row), "k_rowpass_v2<%d,%d> nw=%d grid=%d tail=%d", (n_c + 3) / 4, n_u, (S + 63) / 64,
                 dmf::rowpass_v2_grid(p->N, S), (int)(p->N & 15));
        // (one count digit and more than 32 features: the eight-wave form of the integer Gram kernel takes the first 64)
        snprintf(gram, sizeof(gram), "k_gram_i8<nd=%d>/w8", p->ND);
    } else if (s->use_fused && n_iter2 <= kSplitInnerSteps) {
        const int64_t n_full = p->N - (p->N & 15);
        snprintf(row, sizeof(row), "k_rowpass_fused<%d,%d> nw=%d grid=%d tail=%d", (n_c + 3) / 4, n_u, (S + 63) / 64,
                 dmf::rowpass_fused_grid(n_full, S), (int)(p->N & 15));
        snprintf(gram, sizeof(gram), "fused");
    } else {
        if (s->use_cm_i8 && s->use_gram_i8 && dmf::u_inner_bu_supported(p->V, S, p->SD, n_u, (int)n_iter2))
            snprintf(row, sizeof(row), "k_cm_i8<nd=%d>+k_inner_bu", p->ND);
        else if (s->use_cm_i8) snprintf(row, sizeof(row), "k_cm_i8<nd=%d>+k_u_inner_rows", p->ND);
        else if (s->use_u_big && dmf::u_phase_big_supported(S, n_c, n_u, (int)n_iter2)) snprintf(row, sizeof(row), "k_u_phase_big");
        else if (s->u_path == 0 && (n_iter2 > kSplitInnerSteps || n_u >= 7 || (n_c > 0 && n_u >= 5)))
            snprintf(row, sizeof(row), "k_u_phase_mfma(split)+k_u_inner_rows");
        else if (s->u_path == 0) snprintf(row, sizeof(row), "k_u_phase_mfma");
        else if (s->u_path == 1) snprintf(row, sizeof(row), "k_u_phase_gram");
        else snprintf(row, sizeof(row), "k_u_step_direct");
        if (s->use_gram_i8 && s->use_cm_i8 && dmf::u_inner_bu_supported(p->V, S, p->SD, n_u, (int)n_iter2))
            snprintf(gram, sizeof(gram), "k_gram_i8<nd=%d>/w8", p->ND);  // (b_u comes from k_inner_bu)
        else if (s->use_gram_i8)
            snprintf(gram, sizeof(gram), "k_bu_cols+k_gram_i8<nd=%d>/w8", p->ND);
        else snprintf(gram, sizeof(gram), "%s", s->use_gram_spec ? "k_gram_u" : s->use_gram_mfma ? "k_gram_mfma" : "k_gram");
    }
    const bool tps = s->ctx->generic_level == 1 || s->ctx->generic_level == 2;
    const char* alpha = s->purity != nullptr ? (K <= 16 && n_c >= 1 ? "k_alpha_frank_wolfe_row16" : "k_alpha_frank_wolfe")
                        : (!tps && K <= 16)  ? "k_alpha_phase_row16"
                        : (!tps && K <= 64)  ? "k_alpha_phase_lanes"
                        : (tps && K <= 16)   ? "k_alpha_phase"
                                             : "k_alpha_phase_dyn";
    snprintf(buf, (size_t)cap, "rowpass=%s gram=%s alpha=%s", row, gram, alpha);
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
    if (st == DMF_OK) st = enqueue_u_phase(s, (int)n_iter2);
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
    if (st == DMF_OK) st = enqueue_gram(s);
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
