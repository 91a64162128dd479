// Kernel selection as ONE table: which kernels a (shape, counts, level) gets for the u phase, the u-dependent Gram entries
// and the alpha phase of an outer iteration (deconvolution.py:206-221).  Pure host functions of the key -- no pointers, no
// context -- so that dmf_solver_create / enqueue_outer_iteration / dmf_solver_describe read the same answer and a CPU test
// can enumerate a grid of keys against a checked-in table (tests/golden/kernel_selection.tsv, dmf_select_describe).
#pragma once
#include <cstddef>
#include <cstdint>

namespace dmf {

// ---- thresholds (each measured; the measurement is cited where the constant is used in dmf_select.hip)
constexpr int kSplitInnerSteps = 50;        // beyond this many inner steps the split u phase beats the one-launch row pass
constexpr int kCmI8MinNu = 5;               // wide row groups: k_cm_i8 from this many unknowns on
constexpr int kSplitMinNu = 7;              // k_u_phase_mfma: split form from this many unknowns on (5 with known types)
constexpr int kGramI8MinFp64Acc = 48;       // integer Gram route instead of k_gram_u from this many FP64 accumulators on
constexpr int kGramI8MinPairsNoKnown = 33;  // ... and without known types from this many unknown pairs on (8 unknowns)

struct ShapeKey {
    int64_t N = 0;
    int S = 0, n_c = 0, n_u = 0;
    int nd = 0;                // count digit planes of the problem's integer copies (0: none -- counts not integral,
                               // beyond 32639, more than 2048 samples, R_trunc outside [0, 1])
    int SD = 0;                // padded sample count of those copies
    int level = 0;             // kernel selection level (dmf_context_set_generic): 0 fastest .. 4
    bool d_f32_exact = false;  // every count survives a round trip through f32
    bool rtp_present = true;   // the padded copy of R_trunc exists (n_c > 0)
    unsigned v_align = 0;      // address of V modulo 16
    unsigned rtp_align = 0;    // address of the padded R_trunc modulo 16
    bool alpha_unit = true;    // the starting alpha lies inside [0, 1] (the fixed-point features of the integer kernels)
};

// what dmf_solver_create fixes for the life of a solver
struct PathSpec {
    int u_path = 2;  // fall-back u phase: 0 k_u_phase_mfma, 1 k_u_phase_gram, 2 k_u_step_direct
    bool use_gram_spec = false, use_gram_mfma = false, use_u_big = false;
    bool use_v2 = false, use_cm_i8 = false, use_gram_i8 = false, use_fused = false;
    bool supported = true;  // false: no kernel takes this shape (DMF_ERR_UNSUPPORTED)
};

enum class RowKind {
    RowpassV2,        // k_rowpass_v2: u phase + b_u in one launch
    CmI8InnerBu,      // k_cm_i8 + k_inner_bu (b_u rides with the inner iterations)
    CmI8InnerRows,    // k_cm_i8 + k_u_inner_rows
    RowpassFused,     // first generation: k_rowpass_fused (Gram inside)
    UPhaseBig,        // k_u_phase_big
    UPhaseMfmaSplit,  // k_u_phase_mfma (split) + k_u_inner_rows
    UPhaseMfma,       // k_u_phase_mfma
    UPhaseGram,       // k_u_phase_gram
    UStepDirect       // k_u_step_direct, one launch per inner step
};
enum class GramKind { InRowPass, I8, BuColsI8, GramU, GramMfma, Gram };
enum class AlphaKind { FrankWolfeRow16, FrankWolfe, PhaseRow16, PhaseLanes, Phase, PhaseDyn };

// what ONE outer iteration with n_iter2 inner steps launches
struct IterationPlan {
    RowKind row = RowKind::UStepDirect;
    GramKind gram = GramKind::Gram;
    AlphaKind alpha = AlphaKind::PhaseDyn;
};

PathSpec select_path(const ShapeKey& k);
IterationPlan plan_iteration(const ShapeKey& k, const PathSpec& s, int n_iter2, bool purity);
// "rowpass=... gram=... alpha=..." (what dmf_solver_describe reports and the parity tests assert)
int describe_plan(const ShapeKey& k, const IterationPlan& plan, char* buf, size_t cap);

}  // namespace dmf
