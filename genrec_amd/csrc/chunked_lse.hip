// The constrained log-softmax over logits that arrive in column chunks.
//
// The LM-head matmul that produces the [R, V] logits can be run over slices of the vocabulary;
// each slice [R, Vc] is consumed here right after it was produced -- while it still sits in the
// memory-side cache -- and may then be overwritten by the next one, so the full [R, V] tensor never
// exists.  Per slice one kernel keeps, for every row, a partial (max, sum-of-exp) pair and copies
// the raw logits of the row's allowed tokens that fall inside the slice into the row's candidate
// entries; a last small kernel merges the partial pairs in slot order and turns the raw logits into
// log-probabilities.  One wave streams one (row, part) span: no LDS, no barriers.
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <string.h>

#ifndef CH_NT
#define CH_NT 1           // 1: non-temporal loads for the slice (each byte is read once)
#endif
#define MLS_NT CH_NT
#include "lse_math.h"
#include "resolve.h"

#define CH_THREADS 256
#define CH_WAVES (CH_THREADS / GRX_WAVE)
#ifndef CH_UNROLL
#define CH_UNROLL 4
#endif

struct ChunkArgs {
    const int32_t* tok;
    const int32_t* off;
    grx_row_plan* plan;   // [R] in the caller's workspace: written by the launch that opens the call, read by the others
    grx_state_words* st;
    int walk;             // count-based call being opened: every wave walks its row, part 0 records the plan
    int bookkeep;
    int first;            // this launch opens the call: row errors are reported here
    int gather;           // some token of the trie (or EOS) lies inside this launch's columns
    PlanCtx pc;
    const int64_t* ids;
    int64_t ids_stride;
    int T;
    int R;
    const void* scores;   // [R, Vc]: column j is token v_begin + j
    int64_t scores_stride;
    int Vc;
    int v_begin;
    int eos;
    int P;                // waves per row in this launch
    float2* partials;     // [R, n_slots] (max, sum of exp(x - max)) pairs
    int n_slots;
    int slot0;
    grx_cand* cand;
    int64_t cand_stride;
    int cmax;
    // repetition penalty: raw logits of the tokens ids[r, pen_from:T) that lie inside this launch's columns
    float* pen_logits;    // [R, pen_stride], entry t belongs to ids[r, pen_from + t]; null: no penalty
    int64_t pen_stride;
    int pen_from;
};

// (max, sum of exp(x - max)) of everything the lanes of one wave have accumulated.
__device__ __forceinline__ float2 wave_combine(const MaxSum& a) {
    const float M = wave_max(a.m);
    const float C = (M == -INFINITY) ? 0.f : -M * kLog2e;
    const float S = wave_sum(a.s > 0.f ? a.s * fast_exp2(C - a.c) : 0.f);
    // S = sum 2^(x * log2e + C); with M * log2e + C = residual, sum exp(x - M) = S * 2^(-residual)
    return make_float2(M, (M == -INFINITY) ? S : S * fast_exp2(-fmaf(M, kLog2e, C)));
}

template <int DT, bool VEC>
__global__ __launch_bounds__(CH_THREADS) void chunk_lse_kernel(ChunkArgs a) {
    typedef Elem<DT> E;
    typedef typename E::T T;
    constexpr int EPV = VEC ? E::kPerVec : 1;
    const int lane = threadIdx.x & (GRX_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / GRX_WAVE);
    const int64_t item = (int64_t)blockIdx.x * CH_WAVES + wave;
    const int r = (int)(item / a.P);
    if (r >= a.R) return;  // whole waves leave; nothing below synchronises across waves
    const int part = (int)(item - (int64_t)r * a.P);
    const T* __restrict__ row = reinterpret_cast<const T*>(a.scores) + (int64_t)r * a.scores_stride;

    // this wave's share of the row's columns
    const int nvec = a.Vc / EPV;
    const int per = (nvec + a.P - 1) / a.P;
    const int v_lo = part * per < nvec ? part * per : nvec;
    const int v_hi = v_lo + per < nvec ? v_lo + per : nvec;
    const bool last_part = part == a.P - 1;
    const int col_lo = v_lo * EPV, col_hi = last_part ? a.Vc : v_hi * EPV;

    const RowStream vp(row, (int64_t)nvec * 16);
    MaxSum acc;
    acc.m = -INFINITY;
    acc.s = 0.f;
    acc.c = 0.f;
    u32x4 v_head[CH_UNROLL];
    const bool head = VEC && (v_lo + lane + (CH_UNROLL - 1) * GRX_WAVE < v_hi);
    if (head) {
#pragma unroll
        for (int u = 0; u < CH_UNROLL; ++u) v_head[u] = vp.load(v_lo + lane + u * GRX_WAVE);
    }

    // the row's trie node: walked here when this launch opens a count-based call (scalar work on
    // wave-uniform operands, beside the loads already in flight), else read from the plan
    int code = GRX_RES_EMPTY, c_off = 0, c_n = 0;
    if (a.walk) {
        code = count_row_node(a.pc, a.ids + (int64_t)r * a.ids_stride, a.T, r);
        c_off = code >= 0 ? a.off[code] : 0;
        c_n = code >= 0 ? a.off[code + 1] - c_off : (code == GRX_RES_EOS ? 1 : 0);
        if (part == 0 && lane == 0) {
            grx_row_plan pl;
            pl.code = code;
            pl.off = c_off;
            pl.n = c_n;
            pl.pad = 0;
            a.plan[r] = pl;
            if (a.bookkeep && item == 0) plan_count_bookkeeping(a.pc);
        }
    } else if (a.gather || a.first) {
        const grx_row_plan pl = a.plan[r];
        code = pl.code;
        c_off = pl.off;
        c_n = pl.n;
    }
    if (a.first && part == 0 && lane == 0) {
        if (code < 0 && code != GRX_RES_EOS)
            report_row_error(a.st, r, code == GRX_RES_NOMATCH ? GRX_ROWERR_NOMATCH : GRX_ROWERR_EMPTY);
        else if (c_n > a.cmax)
            report_row_error(a.st, r, GRX_ROWERR_OVERFLOW);
    }

    if (VEC) {
        auto consume = [&](const u32x4* v) {
            float f[CH_UNROLL * E::kPerVec];
#pragma unroll
            for (int u = 0; u < CH_UNROLL; ++u) E::unpack(v[u], f + u * E::kPerVec);
            accumulate<CH_UNROLL * E::kPerVec>(acc, f);
        };
        int i = v_lo + lane;
        if (head) {
            consume(v_head);
            i += CH_UNROLL * GRX_WAVE;
        }
        for (; i + (CH_UNROLL - 1) * GRX_WAVE < v_hi; i += CH_UNROLL * GRX_WAVE) {
            u32x4 v[CH_UNROLL];
#pragma unroll
            for (int u = 0; u < CH_UNROLL; ++u) v[u] = vp.load(i + u * GRX_WAVE);
            consume(v);
        }
        for (; i < v_hi; i += GRX_WAVE) {
            const u32x4 v = vp.load(i);
            float f[E::kPerVec];
            E::unpack(v, f);
            accumulate<E::kPerVec>(acc, f);
        }
    }
    // columns not covered by 16-byte vectors (all of them when !VEC)
    for (int e = (VEC ? (last_part ? nvec * EPV : a.Vc) : col_lo) + lane; e < col_hi; e += GRX_WAVE) {
        float f[1] = {E::to_float(row[e])};
        accumulate<1>(acc, f);
    }
    const float2 tot = wave_combine(acc);
    if (lane == 0) a.partials[(int64_t)r * a.n_slots + a.slot0 + part] = tot;

    // raw logits of the allowed tokens that live in this wave's columns
    if (a.gather && c_n <= a.cmax) {
        grx_cand* __restrict__ crow = a.cand + (int64_t)r * a.cand_stride;
        const int t_lo = a.v_begin + col_lo, t_hi = a.v_begin + col_hi;
        for (int i = lane; i < c_n; i += GRX_WAVE) {
            const int tokv = (code == GRX_RES_EOS) ? a.eos : a.tok[c_off + i];
            if (tokv >= t_lo && tokv < t_hi) {
                grx_cand v;
                v.x = tokv;
                v.y = __float_as_int(E::to_float(row[tokv - a.v_begin]));
                cand_store(crow + 1 + i, v);
            }
        }
    }
    // ... and of the tokens the repetition penalty will touch (the row's own ids)
    if (a.pen_logits != nullptr) {
        const int64_t* __restrict__ ids_row = a.ids + (int64_t)r * a.ids_stride;
        float* __restrict__ prow = a.pen_logits + (int64_t)r * a.pen_stride;
        const int64_t t_lo = (int64_t)a.v_begin + col_lo, t_hi = (int64_t)a.v_begin + col_hi;
        for (int t = a.pen_from + lane; t < a.T; t += GRX_WAVE) {
            const int64_t tk = ids_row[t];
            if (tk >= t_lo && tk < t_hi) prow[t - a.pen_from] = E::to_float(row[tk - a.v_begin]);
        }
    }
}

// One wave per row: merge the row's partial pairs in slot order, then rewrite the candidate row in
// whole 128-byte lines -- header {count, log-sum-exp}, {token, logit - log-sum-exp}, zero padding.
__global__ __launch_bounds__(CH_THREADS) void chunk_finish_kernel(const grx_row_plan* __restrict__ plan,
                                                                  const float2* __restrict__ partials, int n_slots,
                                                                  int R, grx_cand* __restrict__ cand,
                                                                  int64_t cand_stride, int cmax) {
    const int lane = threadIdx.x & (GRX_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / GRX_WAVE);
    const int r = blockIdx.x * CH_WAVES + wave;
    if (r >= R) return;
    const grx_row_plan pl = plan[r];
    int c_n = pl.n;
    if (c_n > cmax || c_n < 0) c_n = 0;  // reported as GRX_ROWERR_OVERFLOW when the call was opened
    grx_cand* __restrict__ crow = cand + (int64_t)r * cand_stride;
    const int n_line = ((c_n + 1 + 15) / 16) * 16;
    const float2* __restrict__ prow = partials + (int64_t)r * n_slots;
    float m = -INFINITY;
    for (int p = lane; p < n_slots; p += GRX_WAVE) m = fmaxf(m, prow[p].x);
    const float M = wave_max(m);
    const float Mref = (M == -INFINITY) ? 0.f : M;
    float s = 0.f;
    for (int p = lane; p < n_slots; p += GRX_WAVE) {
        const float2 v = prow[p];
        s += v.y * fast_exp2((v.x - Mref) * kLog2e);  // an empty span holds (-inf, 0): 0 * 2^-inf = 0
    }
    const float log_s = logf(wave_sum(s));
    for (int e = lane; e < n_line; e += GRX_WAVE) {
        grx_cand v = {0, 0};
        if (e == 0) {
            v.x = c_n;
            v.y = __float_as_int(M + log_s);
        } else if (e - 1 < c_n) {
            const grx_cand raw = crow[e];
            v.x = raw.x;
            v.y = __float_as_int((__int_as_float(raw.y) - M) - log_s);
        }
        cand_store(crow + e, v);
    }
}

// The same with transformers' repetition penalty applied to the log-softmax before the mask, as
// the whole-row kernel does it: y_v <- y_v * p (y_v < 0) or y_v / p for every distinct token v of
// ids[r, pen_from:T) inside the vocabulary, the row normalised again.  Only those tokens move, so the
// second log-sum-exp follows from the first: sum_v exp(y'_v) = (S - sum_U exp(x_v - M)) / S + sum_U exp(y'_v).
// One workgroup per row; the distinct tokens are an open-addressing set in LDS, which the candidate
// pass then consults.
struct FinishPenArgs {
    const grx_row_plan* plan;
    const float2* partials;
    int n_slots;
    int R;
    grx_cand* cand;
    int64_t cand_stride;
    int cmax;
    const int64_t* ids;
    int64_t ids_stride;
    int T;
    int pen_from;
    int v_total;
    float penalty;
    const float* pen_logits;
    int64_t pen_stride;
};

__global__ __launch_bounds__(CH_THREADS) void chunk_finish_pen_kernel(FinishPenArgs a) {
    __shared__ int s_pen[PEN_SLOTS];
    __shared__ float s_red[2][CH_WAVES];
    const int tid = threadIdx.x;
    const int lane = tid & (GRX_WAVE - 1);
    const int r = blockIdx.x;
    const grx_row_plan pl = a.plan[r];
    int c_n = pl.n;
    if (c_n > a.cmax || c_n < 0) c_n = 0;
    grx_cand* __restrict__ crow = a.cand + (int64_t)r * a.cand_stride;
    const int n_line = ((c_n + 1 + 15) / 16) * 16;
    // every wave merges the row's pairs for itself (same order in each: identical results, no exchange)
    const float2* __restrict__ prow = a.partials + (int64_t)r * a.n_slots;
    float m = -INFINITY;
    for (int p = lane; p < a.n_slots; p += GRX_WAVE) m = fmaxf(m, prow[p].x);
    const float M = wave_max(m);
    const float Mref = (M == -INFINITY) ? 0.f : M;
    float s = 0.f;
    for (int p = lane; p < a.n_slots; p += GRX_WAVE) {
        const float2 v = prow[p];
        s += v.y * fast_exp2((v.x - Mref) * kLog2e);
    }
    const float S = wave_sum(s);
    const float log_s = logf(S);
    auto penalised = [&](float y) { return y < 0.f ? y * a.penalty : y / a.penalty; };

    for (int i = tid; i < PEN_SLOTS; i += CH_THREADS) s_pen[i] = -1;
    __syncthreads();
    const int64_t* __restrict__ ids_row = a.ids + (int64_t)r * a.ids_stride;
    const float* __restrict__ pen_row = a.pen_logits + (int64_t)r * a.pen_stride;
    float sub = 0.f, add = 0.f;
    for (int t = a.pen_from + tid; t < a.T; t += CH_THREADS) {
        const int64_t tk = ids_row[t];
        if (tk < 0 || tk >= (int64_t)a.v_total) continue;
        const int tokv = (int)tk;
        uint32_t h = ((uint32_t)tokv * 0x9E3779B1u) >> (32 - PEN_SLOT_BITS);
        bool first = false;
        for (;;) {
            const int old = atomicCAS(&s_pen[h], -1, tokv);
            if (old == -1) { first = true; break; }
            if (old == tokv) break;
            h = (h + 1) & (PEN_SLOTS - 1);
        }
        if (first) {
            const float xm = pen_row[t - a.pen_from] - M;
            sub += fast_exp2(xm * kLog2e);
            add += fast_exp2(penalised(xm - log_s) * kLog2e);
        }
    }
    sub = wave_sum(sub);
    add = wave_sum(add);
    if (lane == 0) {
        s_red[0][tid / GRX_WAVE] = sub;
        s_red[1][tid / GRX_WAVE] = add;
    }
    __syncthreads();  // also publishes the finished token set
    sub = 0.f;
    add = 0.f;
#pragma unroll
    for (int w = 0; w < CH_WAVES; ++w) {
        sub += s_red[0][w];
        add += s_red[1][w];
    }
    const float lse2 = logf(fmaxf(S - sub, 0.f) / S + add);
    auto in_pen_set = [&](int tokv) {
        uint32_t h = ((uint32_t)tokv * 0x9E3779B1u) >> (32 - PEN_SLOT_BITS);
        for (;;) {
            const int cur = s_pen[h];
            if (cur == tokv) return true;
            if (cur < 0) return false;
            h = (h + 1) & (PEN_SLOTS - 1);
        }
    };
    for (int e = tid; e < n_line; e += CH_THREADS) {
        grx_cand v = {0, 0};
        if (e == 0) {
            v.x = c_n;
            v.y = __float_as_int(M + log_s + lse2);
        } else if (e - 1 < c_n) {
            const grx_cand raw = crow[e];
            float y = (__int_as_float(raw.y) - M) - log_s;
            if (in_pen_set(raw.x)) y = penalised(y);
            v.x = raw.x;
            v.y = __float_as_int(y - lse2);
        }
        cand_store(crow + e, v);
    }
}

// ---- host side --------------------------------------------------------------------
static size_t ch_dtype_size(int dt) { return dt == GRX_F32 ? 4 : 2; }

extern "C" int32_t grx_chunk_parts(int32_t R, int32_t Vc, int32_t dtype) {
    if (R <= 0 || Vc <= 0) return 1;
    const int nvec = Vc / (16 / (int)ch_dtype_size(dtype));
    int P = 1;
    // enough waves to fill the chip (256 CUs x 4 SIMDs x 8 waves) while a lane keeps >= 4 vectors
    while (P < 32 && (int64_t)R * P < 8192 && nvec / (P * 2) >= 4 * GRX_WAVE) P *= 2;
    return P;
}

template <int DT>
static void launch_chunk(bool vec, const ChunkArgs& a, hipStream_t stream) {
    const int64_t items = (int64_t)a.R * a.P;
    const dim3 grid((unsigned)((items + CH_WAVES - 1) / CH_WAVES));
    if (vec) hipLaunchKernelGGL((chunk_lse_kernel<DT, true>), grid, dim3(CH_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((chunk_lse_kernel<DT, false>), grid, dim3(CH_THREADS), 0, stream, a);
}

// Does any token on an edge of the trie (sorted, distinct) fall inside [lo, hi)?
static bool trie_token_in_range(const grx_trie* trie, int lo, int hi) {
    const int32_t* t = trie->tok_sorted;
    int a = 0, b = trie->n_tok_sorted;
    while (a < b) {  // first token >= lo
        const int mid = (a + b) / 2;
        if (t[mid] < lo) a = mid + 1;
        else b = mid;
    }
    return a < trie->n_tok_sorted && t[a] < hi;
}

extern "C" int grx_constrained_partial(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                       const grx_constraint* con, const grx_chunk* chunk, float* partials,
                                       grx_cand* cand, int64_t cand_stride, int32_t cmax, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
    const char* who = "grx_constrained_partial";
    hipStream_t stream = (hipStream_t)stream_;
    GRX_REQUIRE(rows != nullptr && chunk != nullptr, GRX_E_INVALID, "%s: null rows/chunk", who);
    int rc = grx_check_step_args(who, trie, state, rows, con, workspace, workspace_bytes, chunk->v_total);
    if (rc != GRX_OK) return rc;
    GRX_REQUIRE(chunk->first == 0 || chunk->first == 1, GRX_E_INVALID, "%s: first must be 0 or 1", who);
    GRX_REQUIRE(chunk->v_begin >= 0 && (int64_t)chunk->v_begin + rows->V <= chunk->v_total, GRX_E_INVALID,
                "%s: columns [%d, %lld) outside the vocabulary of %d", who, chunk->v_begin,
                (long long)chunk->v_begin + rows->V, chunk->v_total);
    GRX_REQUIRE(con->eos_token_id < chunk->v_total, GRX_E_INVALID, "%s: eos_token_id %d outside V=%d", who,
                con->eos_token_id, chunk->v_total);
    GRX_REQUIRE(chunk->parts >= 1 && chunk->parts <= 1024, GRX_E_INVALID, "%s: parts must be in [1, 1024]", who);
    GRX_REQUIRE(chunk->n_slots >= 1 && chunk->n_slots <= GRX_MAX_PARTIAL_SLOTS && chunk->slot >= 0 &&
                    chunk->slot + chunk->parts <= chunk->n_slots,
                GRX_E_INVALID, "%s: slots [%d, %d) outside the %d slots per row (at most %d)", who, chunk->slot,
                chunk->slot + chunk->parts, chunk->n_slots, GRX_MAX_PARTIAL_SLOTS);
    const bool penalise = con->repetition_penalty != 0.f && con->repetition_penalty != 1.f;
    GRX_REQUIRE(con->assume_normalized == 0, GRX_E_UNSUPPORTED,
                "%s: assume_normalized needs whole rows (use grx_constrained_candidates)", who);
    GRX_REQUIRE(con->keep_top == 0, GRX_E_UNSUPPORTED, "%s: keep_top needs whole rows (use grx_constrained_candidates)",
                who);
    if (penalise) {
        GRX_REQUIRE(con->repetition_penalty > 0.f, GRX_E_INVALID, "%s: repetition_penalty must be positive", who);
        GRX_REQUIRE(con->penalty_skip >= 0, GRX_E_INVALID, "%s: penalty_skip must be >= 0", who);
        GRX_REQUIRE(rows->T - con->penalty_skip <= GRX_MAX_PENALTY_TOKENS, GRX_E_UNSUPPORTED,
                    "%s: repetition penalty over more than %d tokens per row", who, GRX_MAX_PENALTY_TOKENS);
        GRX_REQUIRE(rows->R == 0 || rows->T <= con->penalty_skip ||
                        (chunk->pen_logits != nullptr && ((uintptr_t)chunk->pen_logits & 3u) == 0 &&
                         chunk->pen_stride >= rows->T - con->penalty_skip),
                    GRX_E_INVALID, "%s: the repetition penalty needs grx_chunk.pen_logits [R, >= %d] fp32", who,
                    rows->T - con->penalty_skip);
    }
    GRX_REQUIRE(chunk->first == 1 || rows->planned == 0, GRX_E_INVALID,
                "%s: planned = 1 is only meaningful on the call that opens the step", who);
    GRX_REQUIRE(cmax >= 1, GRX_E_INVALID, "%s: cmax must be >= 1", who);
    GRX_REQUIRE(cand_stride >= grx_candidate_stride(cmax) && cand_stride % 16 == 0, GRX_E_INVALID,
                "%s: cand_stride must be a multiple of 16 and >= %lld", who, (long long)grx_candidate_stride(cmax));
    GRX_REQUIRE(rows->R == 0 || (cand != nullptr && ((uintptr_t)cand & 127u) == 0), GRX_E_INVALID,
                "%s: cand must be a 128-byte aligned device buffer", who);
    GRX_REQUIRE(rows->R == 0 || (partials != nullptr && ((uintptr_t)partials & 7u) == 0), GRX_E_INVALID,
                "%s: partials must be an 8-byte aligned device buffer", who);
    GRX_REQUIRE((int64_t)rows->R * chunk->parts <= 0x7FFFFFFF, GRX_E_UNSUPPORTED, "%s: too many (row, part) spans", who);

    ChunkArgs a;
    memset(&a, 0, sizeof(a));
    grx_row_plan* plan = rows->R > 0
                             ? reinterpret_cast<grx_row_plan*>(reinterpret_cast<char*>(workspace) + GRX_WS_PLAN_OFFSET)
                             : nullptr;
    if (chunk->first) {
        // opens the processor call: state slot, counter, and whatever must be known about all rows
        rc = grx_plan_rows(trie, state, rows, con, plan, stream, &a.pc, &a.bookkeep);
        if (rc != GRX_OK) return rc;
        a.walk = con->mode == GRX_MODE_COUNT ? 1 : 0;
    }
    if (rows->R == 0) return GRX_OK;
    a.first = chunk->first;
    a.tok = trie->child_tok;
    a.off = trie->child_off;
    a.plan = plan;
    a.st = state ? state->dev : nullptr;
    a.ids = rows->input_ids;
    a.ids_stride = rows->ids_stride;
    a.T = rows->T;
    a.R = rows->R;
    a.scores = rows->scores;
    a.scores_stride = rows->scores_stride;
    a.Vc = rows->V;
    a.v_begin = chunk->v_begin;
    a.eos = con->eos_token_id;
    a.P = chunk->parts;
    a.partials = reinterpret_cast<float2*>(partials);
    a.n_slots = chunk->n_slots;
    a.slot0 = chunk->slot;
    a.cand = cand;
    a.cand_stride = cand_stride;
    a.cmax = cmax;
    if (penalise && rows->T > con->penalty_skip) {
        a.pen_logits = chunk->pen_logits;
        a.pen_stride = chunk->pen_stride;
        a.pen_from = con->penalty_skip;
    }
    const int lo = chunk->v_begin, hi = chunk->v_begin + rows->V;
    a.gather = (trie_token_in_range(trie, lo, hi) || (con->eos_token_id >= lo && con->eos_token_id < hi)) ? 1 : 0;
    const size_t es = ch_dtype_size(rows->dtype);
    const bool vec = (((uintptr_t)rows->scores) & 15u) == 0 && ((rows->scores_stride * (int64_t)es) & 15) == 0;
    if (rows->dtype == GRX_F32) launch_chunk<GRX_F32>(vec, a, stream);
    else if (rows->dtype == GRX_BF16) launch_chunk<GRX_BF16>(vec, a, stream);
    else launch_chunk<GRX_F16>(vec, a, stream);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

extern "C" int grx_constrained_finish(int32_t R, int32_t n_slots, const float* partials, grx_cand* cand,
                                      int64_t cand_stride, int32_t cmax, const void* workspace,
                                      size_t workspace_bytes, void* stream) {
    const char* who = "grx_constrained_finish";
    GRX_REQUIRE(R >= 0, GRX_E_INVALID, "%s: negative row count", who);
    if (R == 0) return GRX_OK;
    GRX_REQUIRE(n_slots >= 1 && n_slots <= GRX_MAX_PARTIAL_SLOTS, GRX_E_INVALID, "%s: n_slots must be in [1, %d]", who,
                GRX_MAX_PARTIAL_SLOTS);
    GRX_REQUIRE(partials != nullptr && ((uintptr_t)partials & 7u) == 0, GRX_E_INVALID,
                "%s: partials must be an 8-byte aligned device buffer", who);
    GRX_REQUIRE(cmax >= 1 && cand_stride >= grx_candidate_stride(cmax) && cand_stride % 16 == 0, GRX_E_INVALID,
                "%s: bad cmax / cand_stride", who);
    GRX_REQUIRE(cand != nullptr && ((uintptr_t)cand & 127u) == 0, GRX_E_INVALID,
                "%s: cand must be a 128-byte aligned device buffer", who);
    GRX_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15u) == 0 && workspace_bytes >= grx_workspace_bytes(R),
                GRX_E_WORKSPACE, "%s: needs the workspace of the partial calls (%zu bytes)", who, grx_workspace_bytes(R));
    const grx_row_plan* plan =
        reinterpret_cast<const grx_row_plan*>(reinterpret_cast<const char*>(workspace) + GRX_WS_PLAN_OFFSET);
    hipLaunchKernelGGL(chunk_finish_kernel, dim3((unsigned)((R + CH_WAVES - 1) / CH_WAVES)), dim3(CH_THREADS), 0,
                       (hipStream_t)stream, plan, reinterpret_cast<const float2*>(partials), n_slots, R, cand,
                       cand_stride, cmax);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

extern "C" int grx_constrained_finish_penalised(const grx_rows* rows, const grx_constraint* con, int32_t v_total,
                                                int32_t n_slots, const float* partials, const float* pen_logits,
                                                int64_t pen_stride, grx_cand* cand, int64_t cand_stride,
                                                int32_t cmax, const void* workspace, size_t workspace_bytes,
                                                void* stream) {
    const char* who = "grx_constrained_finish_penalised";
    GRX_REQUIRE(rows != nullptr && con != nullptr, GRX_E_INVALID, "%s: null rows/constraint", who);
    const bool penalise = con->repetition_penalty != 0.f && con->repetition_penalty != 1.f;
    if (!penalise || rows->T <= con->penalty_skip || rows->R <= 0)  // nothing to penalise
        return grx_constrained_finish(rows->R, n_slots, partials, cand, cand_stride, cmax, workspace, workspace_bytes,
                                      stream);
    const int32_t R = rows->R;
    GRX_REQUIRE(con->repetition_penalty > 0.f && con->penalty_skip >= 0, GRX_E_INVALID,
                "%s: repetition_penalty must be positive, penalty_skip >= 0", who);
    GRX_REQUIRE(rows->T - con->penalty_skip <= GRX_MAX_PENALTY_TOKENS, GRX_E_UNSUPPORTED,
                "%s: repetition penalty over more than %d tokens per row", who, GRX_MAX_PENALTY_TOKENS);
    GRX_REQUIRE(rows->input_ids != nullptr && rows->ids_stride >= rows->T, GRX_E_INVALID, "%s: bad input_ids", who);
    GRX_REQUIRE(v_total >= 1, GRX_E_INVALID, "%s: v_total must be >= 1", who);
    GRX_REQUIRE(pen_logits != nullptr && ((uintptr_t)pen_logits & 3u) == 0 && pen_stride >= rows->T - con->penalty_skip,
                GRX_E_INVALID, "%s: needs the pen_logits [R, >= %d] of the partial calls", who,
                rows->T - con->penalty_skip);
    GRX_REQUIRE(n_slots >= 1 && n_slots <= GRX_MAX_PARTIAL_SLOTS, GRX_E_INVALID, "%s: n_slots must be in [1, %d]", who,
                GRX_MAX_PARTIAL_SLOTS);
    GRX_REQUIRE(partials != nullptr && ((uintptr_t)partials & 7u) == 0, GRX_E_INVALID,
                "%s: partials must be an 8-byte aligned device buffer", who);
    GRX_REQUIRE(cmax >= 1 && cand_stride >= grx_candidate_stride(cmax) && cand_stride % 16 == 0, GRX_E_INVALID,
                "%s: bad cmax / cand_stride", who);
    GRX_REQUIRE(cand != nullptr && ((uintptr_t)cand & 127u) == 0, GRX_E_INVALID,
                "%s: cand must be a 128-byte aligned device buffer", who);
    GRX_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15u) == 0 && workspace_bytes >= grx_workspace_bytes(R),
                GRX_E_WORKSPACE, "%s: needs the workspace of the partial calls (%zu bytes)", who, grx_workspace_bytes(R));
    FinishPenArgs a;
    memset(&a, 0, sizeof(a));
    a.plan = reinterpret_cast<const grx_row_plan*>(reinterpret_cast<const char*>(workspace) + GRX_WS_PLAN_OFFSET);
    a.partials = reinterpret_cast<const float2*>(partials);
    a.n_slots = n_slots;
    a.R = R;
    a.cand = cand;
    a.cand_stride = cand_stride;
    a.cmax = cmax;
    a.ids = rows->input_ids;
    a.ids_stride = rows->ids_stride;
    a.T = rows->T;
    a.pen_from = con->penalty_skip;
    a.v_total = v_total;
    a.penalty = con->repetition_penalty;
    a.pen_logits = pen_logits;
    a.pen_stride = pen_stride;
    hipLaunchKernelGGL(chunk_finish_pen_kernel, dim3((unsigned)R), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}
