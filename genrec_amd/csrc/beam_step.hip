// One beam-search iteration over compact candidates.  One workgroup per prompt
// group: top-n_keep selection over the K * (allowed tokens) accumulated scores,
// stopping criteria, running top-K, finished-beam merge, sequence / parent
// gather and the early-stop heuristic.  Candidate counts are tiny next to the
// vocabulary (at most K * max-fanout), so the whole step lives in LDS and
// wave64 reductions.
#include <math.h>
#include <string.h>

#include "common.h"
#include "resolve.h"

#ifdef BS_STAMP
// Diagnostic build only: per-workgroup phase timestamps (100 MHz counter).
__device__ unsigned long long g_stamps[1024 * 16];
#define STAMP(n) do { if (tid == 0 && b < 1024) g_stamps[b * 16 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int grx_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 1024 * 16);
}
#else
#define STAMP(n) do { } while (0)
#endif

#include "beam_body.h"

template <int MAXK, int MAXKEEP, int SURV_CAP, int TAIL_K, int HOLD = 0, int NT = BS_THREADS>
__global__ __launch_bounds__(NT) void beam_step_kernel(grx_beam_step_args a, BeamDerived d, PlanCtx pc,
                                                       grx_row_plan* __restrict__ plan_out) {
    beam_step_group<MAXK, MAXKEEP, SURV_CAP, TAIL_K, false, HOLD, NT>(a, d, pc, plan_out, (int)blockIdx.x);
}

// Argument checks of a beam step (nothing is enqueued, no state is touched).
int grx_beam_step_check(const char* who, const grx_beam_step_args* args) {
    GRX_REQUIRE(args != nullptr, GRX_E_INVALID, "%s: null args", who);
    const grx_beam_step_args& a = *args;
    GRX_REQUIRE(a.B >= 0 && a.K >= 1 && a.V >= 1 && a.cmax >= 1, GRX_E_INVALID, "%s: bad shape", who);
    GRX_REQUIRE(a.K <= BS_MAXK && a.n_keep <= BS_MAXKEEP, GRX_E_UNSUPPORTED,
                "%s: supports at most %d beams and %d kept candidates", who, BS_MAXK, BS_MAXKEEP);
    GRX_REQUIRE(a.n_keep >= a.K, GRX_E_INVALID, "%s: n_keep must be >= K", who);
    GRX_REQUIRE(a.V <= (1 << 24), GRX_E_UNSUPPORTED, "%s: vocabularies above 2^24 tokens are not supported", who);
    GRX_REQUIRE((int64_t)a.K * a.cmax <= 0x7FFFFFFF, GRX_E_UNSUPPORTED, "%s: K * cmax exceeds 31 bits", who);
    GRX_REQUIRE(a.prompt_len >= 0 && a.cur_len >= a.prompt_len && a.cur_len < a.max_length, GRX_E_INVALID,
                "%s: need prompt_len <= cur_len < max_length", who);
    GRX_REQUIRE(a.copy_from >= 0 && a.copy_from <= a.cur_len, GRX_E_INVALID, "%s: bad copy_from", who);
    GRX_REQUIRE(a.early_stopping >= 0 && a.early_stopping <= 2, GRX_E_INVALID, "%s: bad early_stopping", who);
    GRX_REQUIRE(a.cand_stride >= (int64_t)a.cmax + 1, GRX_E_INVALID, "%s: cand_stride smaller than cmax + 1", who);
    GRX_REQUIRE(a.cand_shared == 0 || a.cand_shared == 1, GRX_E_INVALID, "%s: cand_shared must be 0 or 1", who);
    if (a.B == 0) return GRX_OK;
    GRX_REQUIRE(a.cand && a.run_seq_in && a.run_seq_out && a.run_score_in &&
                    a.run_score_out && a.run_bidx_in && a.run_bidx_out && a.fin_seq_in && a.fin_seq_out &&
                    a.fin_score_in && a.fin_score_out && a.fin_bidx_in && a.fin_bidx_out && a.fin_done_in &&
                    a.fin_done_out && a.unsat,
                GRX_E_INVALID, "%s: null buffer", who);
    GRX_REQUIRE(a.run_seq_in != a.run_seq_out && a.fin_seq_in != a.fin_seq_out && a.run_score_in != a.run_score_out &&
                    a.fin_score_in != a.fin_score_out && a.run_bidx_in != a.run_bidx_out &&
                    a.fin_bidx_in != a.fin_bidx_out && a.fin_done_in != a.fin_done_out,
                GRX_E_INVALID, "%s: in and out buffers must differ", who);
    if (a.plan_next != nullptr) {
        const int rc = grx_plan_request_check(who, a.plan_next, a.B * a.K);
        if (rc != GRX_OK) return rc;
    }
    return GRX_OK;
}

bool grx_beam_step_is_small(const grx_beam_step_args& a) { return a.K <= BS_SMALL_K && a.n_keep <= BS_SMALL_KEEP; }

// What the kernel needs beyond the caller's arguments; opens the plan made ahead for the next
// processor call (a.plan_next), which touches the host side of its grx_state.
int grx_beam_step_derive(const char* who, const grx_beam_step_args& a, hipStream_t stream, BeamDerived* d_out, PlanCtx* pc,
                         grx_row_plan** plan_out) {
    BeamDerived d;
    const double lp = (double)a.length_penalty;
    d.fin_div = (float)pow((double)(a.cur_len + 1 - a.prompt_len), lp);
    const int hyp = (a.early_stopping == 2 && a.length_penalty > 0.0f) ? (a.max_length - a.prompt_len)
                                                                       : (a.cur_len + 1 - a.prompt_len);
    d.heur_div = (float)pow((double)hyp, lp);
    d.reg_keys = (a.K <= BS_PREFETCH && a.cmax <= BS_THREADS && a.n_keep <= GRX_WAVE) ? 1 : 0;
    const bool small = grx_beam_step_is_small(a);
    d.hold_keys = (!small && a.K <= BS_MID_K && a.n_keep <= BS_MID_KEEP && a.cmax <= BS_COLS) ? 1 : 0;
#ifdef BS_NO_HOLD  // diagnostic build: the two-pass route for every wide group
    d.hold_keys = 0;
#endif
    d.tails_in_lds = (a.K <= (small ? BS_SMALL_K : BS_TAIL_K) && a.cur_len + 1 - a.copy_from <= BS_TAIL_W &&
                      a.cur_len - a.prompt_len + 1 <= BS_TAIL_W) ? 1 : 0;
    *d_out = d;
    memset(pc, 0, sizeof(*pc));
    *plan_out = nullptr;
    if (a.plan_next != nullptr) {
        const int rc = grx_plan_ctx_for_next_call(who, a.plan_next, a.B * a.K, pc, plan_out, stream);
        if (rc != GRX_OK) return rc;
    }
    return GRX_OK;
}

void grx_beam_step_launch(const grx_beam_step_args& a, const BeamDerived& d, const PlanCtx& pc, grx_row_plan* plan_out,
                          hipStream_t stream) {
    if (grx_beam_step_is_small(a))
        hipLaunchKernelGGL((beam_step_kernel<BS_SMALL_K, BS_SMALL_KEEP, BS_SMALL_SURV, BS_SMALL_K>), dim3(a.B),
                           dim3(BS_THREADS), 0, stream, a, d, pc, plan_out);
    else if (d.hold_keys)
        hipLaunchKernelGGL((beam_step_kernel<BS_MID_K, BS_MID_KEEP, BS_SURV_CAP, BS_TAIL_K, BS_MID_HOLD, BS_MID_THREADS>),
                           dim3(a.B), dim3(BS_MID_THREADS), 0, stream, a, d, pc, plan_out);
    else
        hipLaunchKernelGGL((beam_step_kernel<BS_MAXK, BS_MAXKEEP, BS_SURV_CAP, BS_TAIL_K>), dim3(a.B), dim3(BS_THREADS),
                           0, stream, a, d, pc, plan_out);
}

extern "C" int grx_beam_step(const grx_beam_step_args* args, void* stream) {
    const char* who = "grx_beam_step";
    int rc = grx_beam_step_check(who, args);
    if (rc != GRX_OK) return rc;
    const grx_beam_step_args& a = *args;
    if (a.B == 0) return GRX_OK;
    BeamDerived d;
    PlanCtx pc;
    grx_row_plan* plan_out = nullptr;
    rc = grx_beam_step_derive(who, a, (hipStream_t)stream, &d, &pc, &plan_out);
    if (rc != GRX_OK) return rc;
    grx_beam_step_launch(a, d, pc, plan_out, (hipStream_t)stream);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

// ------------------------------------------------------------------ init ---
// Initial beam-search state (transformers generation/utils.py:3316-3351) written
// into both halves of every double buffer by one launch.
// With a plan request the first `plan_blocks` workgroups plan the rows of the first processor call
// (they only read the caller's input_ids, so they run beside the initialisation instead of in a
// kernel of their own): one thread per row in the count-based mode, one wave per row in the fixed-hint mode.
__global__ void beam_init_kernel(grx_beam_init_args a, PlanCtx pc, grx_row_plan* __restrict__ plan_out, int plan_rows,
                                 int64_t plan_stride, int plan_blocks) {
    if ((int)blockIdx.x < plan_blocks) {
        const int g = blockIdx.x * blockDim.x + threadIdx.x;
        if (pc.mode == GRX_MODE_COUNT) {
            if (g < plan_rows) {
                if (g == 0) plan_count_bookkeeping(pc);
                count_row_match(pc, a.input_ids + (int64_t)g * plan_stride, a.prompt_len, g);
            }
        } else {
            const int r = g / GRX_WAVE;
            if (r < plan_rows) {
                const grx_row_plan pl =
                    plan_fixed_row(pc, a.input_ids + (int64_t)r * plan_stride, a.prompt_len, threadIdx.x & (GRX_WAVE - 1));
                if ((threadIdx.x & (GRX_WAVE - 1)) == 0) plan_out[r] = pl;
            }
        }
        return;
    }
    const int64_t R = (int64_t)a.B * a.K;
    const int L = a.max_length, Lg = a.max_length - a.prompt_len;
    const int64_t total = R * L;
    const int64_t n_threads = (int64_t)(gridDim.x - plan_blocks) * blockDim.x;
    for (int64_t i = (int64_t)(blockIdx.x - plan_blocks) * blockDim.x + threadIdx.x; i < total; i += n_threads) {
        const int64_t r = i / L;
        const int p = (int)(i % L);
        const int64_t v = p < a.prompt_len ? a.input_ids[r * a.ids_stride + p] : a.fill_token;
        a.run_seq[0][i] = v;
        a.run_seq[1][i] = v;
        a.fin_seq[0][i] = v;
        a.fin_seq[1][i] = v;
        if (p < Lg) {
            const int64_t q = r * Lg + p;
            a.run_bidx[0][q] = -1;
            a.run_bidx[1][q] = -1;
            a.fin_bidx[0][q] = -1;
            a.fin_bidx[1][q] = -1;
        }
        if (p == 0) {
            const float rs = (r % a.K == 0) ? 0.0f : -1.0e9f;
            a.run_score[0][r] = rs;
            a.run_score[1][r] = rs;
            a.fin_score[0][r] = -1.0e9f;
            a.fin_score[1][r] = -1.0e9f;
            a.fin_done[0][r] = 0;
            a.fin_done[1][r] = 0;
            if (r % a.K == 0) a.unsat[r / a.K] = 1;
        }
    }
}

extern "C" int grx_beam_init(const grx_beam_init_args* args, void* stream) {
    const char* who = "grx_beam_init";
    GRX_REQUIRE(args != nullptr, GRX_E_INVALID, "%s: null args", who);
    const grx_beam_init_args& a = *args;
    GRX_REQUIRE(a.B >= 0 && a.K >= 1 && a.prompt_len >= 0 && a.max_length > a.prompt_len, GRX_E_INVALID,
                "%s: bad shape", who);
    if (a.B == 0) return GRX_OK;
    GRX_REQUIRE(a.ids_stride >= a.prompt_len && (a.prompt_len == 0 || a.input_ids), GRX_E_INVALID,
                "%s: bad input_ids", who);
    for (int h = 0; h < 2; ++h)
        GRX_REQUIRE(a.run_seq[h] && a.fin_seq[h] && a.run_score[h] && a.fin_score[h] && a.run_bidx[h] &&
                        a.fin_bidx[h] && a.fin_done[h],
                    GRX_E_INVALID, "%s: null buffer", who);
    GRX_REQUIRE(a.unsat != nullptr, GRX_E_INVALID, "%s: null unsat", who);
    const int64_t total = (int64_t)a.B * a.K * a.max_length;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    PlanCtx pc;
    memset(&pc, 0, sizeof(pc));
    grx_row_plan* plan_out = nullptr;
    int plan_rows = 0, plan_blocks = 0;
    int64_t plan_stride = 0;
    if (a.plan_next != nullptr) {  // the first processor call sees the prompts themselves
        const int per = a.plan_one_per_group ? a.K : 1;  // plan every K-th row only
        plan_rows = a.B * a.K / per;
        plan_stride = a.ids_stride * per;
        const int rc = grx_plan_ctx_for_next_call(who, a.plan_next, plan_rows, &pc, &plan_out, (hipStream_t)stream);
        if (rc != GRX_OK) return rc;
        const int64_t plan_threads = pc.mode == GRX_MODE_COUNT ? (int64_t)plan_rows : (int64_t)plan_rows * GRX_WAVE;
        plan_blocks = (int)((plan_threads + 255) / 256);
    }
    hipLaunchKernelGGL(beam_init_kernel, dim3((unsigned)(blocks + plan_blocks)), dim3(256), 0, (hipStream_t)stream, a, pc,
                       plan_out, plan_rows, plan_stride, plan_blocks);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}
