// Kernel selection (dmf_select.h): the one place where shape rules live.  Host code only.
#include "dmf_select.h"
#include "dmf_internal.h"
#include <cstdio>
#include <cstdlib>

namespace dmf {

namespace {

// A threshold: a named constant in the product.  Only a -DDMF_EXPERIMENT build (DMF_EXPERIMENT=1 python -m
// demethify_amd._build, what tools/wide_nu_sweep.py and tools/gram_i8_vs_fp64.py use for their before / after columns) lets
// the environment move it.
int knob(const char* name, int value) {
#ifdef DMF_EXPERIMENT
    if (const char* v = getenv(name)) return atoi(v);
#else
    (void)name;
#endif
    return value;
}

// (the support predicates of the stream kernels test the alignment of V through its address)
const double* aligned_like(unsigned align) { return reinterpret_cast<const double*>((uintptr_t)0x10000 + align); }

}  // namespace

PathSpec select_path(const ShapeKey& k) {
    PathSpec s;
    const int S = k.S, n_c = k.n_c, n_u = k.n_u;
    const bool fast = k.level == 0 || k.level == 3 || k.level == 4;
    const bool ints = k.nd > 0;  // the problem carries u16 / 8-bit-plane copies of its counts
    const double* v_like = aligned_like(k.v_align);

    // ---- fall-back u phase and FP64 Gram kernels (every level)
    if (fast && u_phase_mfma_supported(S, n_c, n_u)) s.u_path = 0;
    else if (k.level != 2 && u_phase_gram_supported(S, n_c, n_u)) s.u_path = 1;
    else s.u_path = 2;
    s.use_gram_spec = fast && gram_u_supported(n_c, n_u);
    s.use_gram_mfma = fast && !s.use_gram_spec;
    s.use_u_big = fast && s.u_path != 0 && u_phase_big_supported(S, n_c, n_u, 64);

    // ---- second generation: one-launch row pass on u16 counts + integer-matrix-core Gram (level 0 only)
    s.use_v2 = k.level == 0 && ints && (n_c == 0 || k.rtp_present) && k.v_align == 0 && k.rtp_align == 0 &&
               rowpass_v2_supported(S, n_c, n_u, 20) && gram_i8_supported(n_c, n_u, k.nd, k.N, k.SD);

    // ---- wide row groups (5..32 unknowns; narrow ones beyond the row pass's 512 samples or 16 known types): the split u
    // phase with the integer-matrix-core producer.  Measured at 5e5 x 128 against what ran before: DESIGN.md section 5.
    const int cm_min_nu = knob("DMF_CM_I8_MIN_NU", kCmI8MinNu);
    s.use_cm_i8 = k.level == 0 && !s.use_v2 && ints && (n_u >= cm_min_nu || S > 512 || n_c > 16) && n_u <= 32 &&
                  (n_c == 0 || (k.rtp_present && (k.rtp_align & 7) == 0)) && cm_i8_supported(v_like, S, n_c, n_u, k.nd, k.SD);

    // ---- integer Gram route behind a u phase that is a kernel of its own.  What it competes with is k_gram_u, whose time
    // grows with its accumulator count (padded known types x unknowns + pairs + b_u) while the integer route is flat
    // (k_bu_cols dominates it): at 5e5 x 128 k_gram_u takes 0.21 ms with 51 accumulators, 0.23 with 40, 0.36 with 60..76,
    // the integer route 0.20..0.21 throughout (tools/gram_i8_vs_fp64.py): from 48 accumulators on; without known types from
    // 8 unknowns (36 pairs) on.  Behind k_cm_i8 the b_u stream rides along with the inner iterations (k_inner_bu) and the
    // integer route is all that is left of the Gram pass: it then wins at every width.
    const int fp64_acc = (n_c + 3) / 4 * 4 * n_u + n_u * (n_u + 1) / 2 + n_u;
    const bool fused_bu = s.use_cm_i8 && u_inner_bu_supported(v_like, S, k.SD, n_u, 20);
    const bool known_ok = n_c > 0 ? (k.rtp_present && k.rtp_align == 0) : true;
    const bool wide_enough = n_c > 0 ? fp64_acc >= knob("DMF_GRAM_I8_MIN", kGramI8MinFp64Acc)
                                     : n_u * (n_u + 1) / 2 >= knob("DMF_GRAM_I8_MIN_NC0", kGramI8MinPairsNoKnown);
    s.use_gram_i8 = k.level == 0 && ints && known_ok && (fused_bu || wide_enough) && n_u <= 32 &&
                    gram_i8_supported(n_c, n_u, k.nd, k.N, k.SD);

    // the row kernels of the integer routes write alpha_j alpha_l in fixed point on [0, 1]: true of every iterate (columns
    // on the simplex), required of the caller's starting point
    if (!k.alpha_unit) {
        s.use_v2 = false;
        s.use_cm_i8 = false;
    }

    // ---- first generation: fused row pass on f64 counts (level 0 fall-back, level 4)
    s.use_fused = (k.level == 0 || k.level == 4) && k.d_f32_exact && k.N >= 16 && rowpass_fused_supported(S, n_c, n_u) &&
                  u_phase_mfma_supported(S, n_c, n_u) && gram_u_supported(n_c, n_u);

    s.supported = !(s.u_path == 2 && !u_step_direct_supported(S, n_c, n_u));
    return s;
}

IterationPlan plan_iteration(const ShapeKey& k, const PathSpec& s, int n_iter2, bool purity) {
    IterationPlan plan;
    const int S = k.S, n_c = k.n_c, n_u = k.n_u, K = n_c + n_u;
    const double* v_like = aligned_like(k.v_align);
    // The fixed-point features of the integer Gram need u inside [0, 1]: the u phase's clip puts it there -- after at
    // least one inner step.  With none, u is still the caller's array and the FP64 kernels run.
    const bool clipped = n_iter2 >= 1;
    if (s.use_v2 && clipped && n_iter2 <= kSplitInnerSteps && rowpass_v2_supported(S, n_c, n_u, n_iter2)) {
        plan.row = RowKind::RowpassV2;
        plan.gram = GramKind::I8;
    } else if (s.use_cm_i8 && s.use_gram_i8 && clipped && u_inner_bu_supported(v_like, S, k.SD, n_u, n_iter2)) {
        plan.row = RowKind::CmI8InnerBu;
        plan.gram = GramKind::I8;
    } else if (s.use_fused && n_iter2 <= kSplitInnerSteps) {
        // With many inner steps the row-local iterations (one wave per workgroup in the fused kernels) dominate and the
        // split u phase, which runs them chip-wide, wins: break-even around 50 inner steps.
        plan.row = RowKind::RowpassFused;
        plan.gram = GramKind::InRowPass;
    } else {
        // The split form of k_u_phase_mfma also wins at few inner steps once the row groups are wide (5e5 x 128, 20 steps:
        // 0+8 0.70 -> 0.57 ms, 12+6 0.62 -> 0.48; 0+5 equal): from 7 unknowns on, from 5 with known types.
        if (s.use_cm_i8) plan.row = RowKind::CmI8InnerRows;
        else if (s.use_u_big && u_phase_big_supported(S, n_c, n_u, n_iter2)) plan.row = RowKind::UPhaseBig;
        else if (s.u_path == 0 && (n_iter2 > kSplitInnerSteps || n_u >= knob("DMF_SPLIT_NU", kSplitMinNu) || (n_c > 0 && n_u >= 5)))
            plan.row = RowKind::UPhaseMfmaSplit;
        else if (s.u_path == 0) plan.row = RowKind::UPhaseMfma;
        else if (s.u_path == 1) plan.row = RowKind::UPhaseGram;
        else plan.row = RowKind::UStepDirect;
        if (s.use_gram_i8 && clipped) plan.gram = GramKind::BuColsI8;
        else if (s.use_gram_spec) plan.gram = GramKind::GramU;
        else if (s.use_gram_mfma) plan.gram = GramKind::GramMfma;
        else plan.gram = GramKind::Gram;
    }
    const bool tps = k.level == 1 || k.level == 2;  // the thread-per-sample alpha kernels
    if (purity) plan.alpha = (K <= 16 && n_c >= 1) ? AlphaKind::FrankWolfeRow16 : AlphaKind::FrankWolfe;
    else if (!tps && K <= 16) plan.alpha = AlphaKind::PhaseRow16;
    else if (!tps && K <= 64) plan.alpha = AlphaKind::PhaseLanes;
    else if (tps && K <= 16) plan.alpha = AlphaKind::Phase;
    else plan.alpha = AlphaKind::PhaseDyn;
    return plan;
}

int describe_plan(const ShapeKey& k, const IterationPlan& plan, char* buf, size_t cap) {
    char row[160], gram[64];
    const int S = k.S, n_c = k.n_c, n_u = k.n_u;
    switch (plan.row) {
        case RowKind::RowpassV2:
            snprintf(row, sizeof(row), "k_rowpass_v2<%d,%d> nw=%d grid=%d tail=%d", (n_c + 3) / 4, n_u, (S + 63) / 64,
                     rowpass_v2_grid(k.N, S), (int)(k.N & 15));
            break;
        case RowKind::RowpassFused:
            snprintf(row, sizeof(row), "k_rowpass_fused<%d,%d> nw=%d grid=%d tail=%d", (n_c + 3) / 4, n_u, (S + 63) / 64,
                     rowpass_fused_grid(k.N - (k.N & 15), S), (int)(k.N & 15));
            break;
        case RowKind::CmI8InnerBu: snprintf(row, sizeof(row), "k_cm_i8<nd=%d>+k_inner_bu", k.nd); break;
        case RowKind::CmI8InnerRows: snprintf(row, sizeof(row), "k_cm_i8<nd=%d>+k_u_inner_rows", k.nd); break;
        case RowKind::UPhaseBig: snprintf(row, sizeof(row), "k_u_phase_big"); break;
        case RowKind::UPhaseMfmaSplit: snprintf(row, sizeof(row), "k_u_phase_mfma(split)+k_u_inner_rows"); break;
        case RowKind::UPhaseMfma: snprintf(row, sizeof(row), "k_u_phase_mfma"); break;
        case RowKind::UPhaseGram: snprintf(row, sizeof(row), "k_u_phase_gram"); break;
        case RowKind::UStepDirect: snprintf(row, sizeof(row), "k_u_step_direct"); break;
    }
    switch (plan.gram) {
        case GramKind::InRowPass: snprintf(gram, sizeof(gram), "fused"); break;
        case GramKind::I8: snprintf(gram, sizeof(gram), "k_gram_i8<nd=%d>/w8", k.nd); break;
        case GramKind::BuColsI8: snprintf(gram, sizeof(gram), "k_bu_cols+k_gram_i8<nd=%d>/w8", k.nd); break;
        case GramKind::GramU: snprintf(gram, sizeof(gram), "k_gram_u"); break;
        case GramKind::GramMfma: snprintf(gram, sizeof(gram), "k_gram_mfma"); break;
        case GramKind::Gram: snprintf(gram, sizeof(gram), "k_gram"); break;
    }
    const char* alpha = "";
    switch (plan.alpha) {
        case AlphaKind::FrankWolfeRow16: alpha = "k_alpha_frank_wolfe_row16"; break;
        case AlphaKind::FrankWolfe: alpha = "k_alpha_frank_wolfe"; break;
        case AlphaKind::PhaseRow16: alpha = "k_alpha_phase_row16"; break;
        case AlphaKind::PhaseLanes: alpha = "k_alpha_phase_lanes"; break;
        case AlphaKind::Phase: alpha = "k_alpha_phase"; break;
        case AlphaKind::PhaseDyn: alpha = "k_alpha_phase_dyn"; break;
    }
    return snprintf(buf, cap, "rowpass=%s gram=%s alpha=%s", row, gram, alpha);
}

}  // namespace dmf
