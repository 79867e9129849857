// Trie upload, device-resident processor state, and the preparation of a
// processor call (integer work, one pass over the token rows).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "resolve.h"

// ---------------------------------------------------------------- errors ---
static thread_local char g_err[512] = "";

void grx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* grx_last_error(void) { return g_err; }
extern "C" int grx_abi_version(void) { return GRX_ABI_VERSION; }

// ------------------------------------------------------------------ trie ---
extern "C" int grx_trie_create(const int32_t* child_off, const int32_t* child_tok, int32_t n_nodes,
                               int32_t n_edges, grx_trie** out) {
    GRX_REQUIRE(out != nullptr && child_off != nullptr, GRX_E_INVALID, "grx_trie_create: null pointer");
    GRX_REQUIRE(n_nodes >= 1 && n_edges >= 0 && n_nodes == n_edges + 1, GRX_E_INVALID,
                "grx_trie_create: need n_nodes == n_edges + 1 >= 1 (got %d nodes, %d edges)", n_nodes, n_edges);
    GRX_REQUIRE(n_edges == 0 || child_tok != nullptr, GRX_E_INVALID, "grx_trie_create: null child_tok");
    GRX_REQUIRE(child_off[0] == 0 && child_off[n_nodes] == n_edges, GRX_E_INVALID,
                "grx_trie_create: child_off must start at 0 and end at n_edges");
    int32_t max_fanout = 0, max_token = -1;
    for (int32_t n = 0; n < n_nodes; ++n) {
        const int32_t a = child_off[n], b = child_off[n + 1];
        GRX_REQUIRE(a <= b && b <= n_edges, GRX_E_INVALID, "grx_trie_create: child_off not monotone at node %d", n);
        for (int32_t e = a; e < b; ++e) {
            GRX_REQUIRE(child_tok[e] >= 0, GRX_E_INVALID, "grx_trie_create: negative token on edge %d", e);
            GRX_REQUIRE(e == a || child_tok[e - 1] < child_tok[e], GRX_E_INVALID,
                        "grx_trie_create: tokens of node %d are not strictly increasing", n);
            // children are numbered after their parent, so a walk always terminates
            GRX_REQUIRE(e + 1 > n, GRX_E_INVALID, "grx_trie_create: edge %d points backwards", e);
            if (child_tok[e] > max_token) max_token = child_tok[e];
        }
        if (b - a > max_fanout) max_fanout = b - a;
    }
    grx_trie* t = new grx_trie();
    memset(t, 0, sizeof(*t));
    t->n_nodes = n_nodes;
    t->n_edges = n_edges;
    t->max_fanout = max_fanout;
    t->max_token = max_token;
    // unary chain below the root
    t->chain_n = 0;
    t->chain_node = 0;
    while (t->chain_n < GRX_CHAIN_MAX && child_off[t->chain_node + 1] - child_off[t->chain_node] == 1) {
        const int32_t edge = child_off[t->chain_node];
        t->chain_tok[t->chain_n++] = child_tok[edge];
        t->chain_node = edge + 1;
    }
    // edge table: power-of-two size >= 2 * n_edges, linear probing
    uint32_t slots = 16;
    while (slots < 2u * (uint32_t)n_edges) slots <<= 1;
    t->hash_mask = slots - 1;
    std::vector<grx_edge_slot> table(slots, grx_edge_slot{-1, -1, -1, 0});
    for (int32_t n = 0; n < n_nodes; ++n) {
        for (int32_t e = child_off[n]; e < child_off[n + 1]; ++e) {
            uint32_t h = grx_edge_hash((uint32_t)n, (uint32_t)child_tok[e]) & t->hash_mask;
            while (table[h].parent >= 0) h = (h + 1) & t->hash_mask;
            table[h] = grx_edge_slot{n, child_tok[e], e + 1, 0};
        }
    }
    {
        std::vector<int32_t> distinct(child_tok, child_tok + n_edges);
        std::sort(distinct.begin(), distinct.end());
        distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        t->n_tok_sorted = (int32_t)distinct.size();
        t->tok_sorted = (int32_t*)malloc(sizeof(int32_t) * (distinct.size() + 1));
        if (t->tok_sorted == nullptr) {
            grx_set_error("grx_trie_create: out of host memory");
            delete t;
            return GRX_E_INVALID;
        }
        if (!distinct.empty()) memcpy(t->tok_sorted, distinct.data(), sizeof(int32_t) * distinct.size());
    }
    hipError_t e = hipGetDevice(&t->device);
    if (e == hipSuccess) e = hipMalloc((void**)&t->edge_hash, sizeof(grx_edge_slot) * (size_t)slots);
    if (e == hipSuccess)
        e = hipMemcpy(t->edge_hash, table.data(), sizeof(grx_edge_slot) * (size_t)slots, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&t->child_off, sizeof(int32_t) * (size_t)(n_nodes + 1));
    if (e == hipSuccess) e = hipMalloc((void**)&t->child_tok, sizeof(int32_t) * (size_t)(n_edges > 0 ? n_edges : 1));
    if (e == hipSuccess)
        e = hipMemcpy(t->child_off, child_off, sizeof(int32_t) * (size_t)(n_nodes + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_edges > 0)
        e = hipMemcpy(t->child_tok, child_tok, sizeof(int32_t) * (size_t)n_edges, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        grx_set_error("grx_trie_create: %s", hipGetErrorString(e));
        if (t->child_off) (void)hipFree(t->child_off);
        if (t->child_tok) (void)hipFree(t->child_tok);
        if (t->edge_hash) (void)hipFree(t->edge_hash);
        free(t->tok_sorted);
        delete t;
        return GRX_E_HIP;
    }
    *out = t;
    return GRX_OK;
}

extern "C" int grx_trie_destroy(grx_trie* t) {
    if (!t) return GRX_OK;
    (void)hipFree(t->child_off);
    (void)hipFree(t->child_tok);
    (void)hipFree(t->edge_hash);
    free(t->tok_sorted);
    delete t;
    return GRX_OK;
}

extern "C" int32_t grx_trie_max_fanout(const grx_trie* t) { return t ? t->max_fanout : 0; }
extern "C" int32_t grx_trie_max_token(const grx_trie* t) { return t ? t->max_token : -1; }

// ----------------------------------------------------------------- state ---
__global__ void state_clear_error_kernel(grx_state_words* st) { st->err = GRX_NO_ERR; }
// Forget the row matches a dropped plan recorded for the call that owns `slot`.
__global__ void state_rearm_kernel(grx_state_words* st, int slot) { st->first_match[slot] = GRX_NO_MATCH; }

extern "C" int grx_state_create(grx_state** out) {
    GRX_REQUIRE(out != nullptr, GRX_E_INVALID, "grx_state_create: null pointer");
    grx_state* s = new grx_state();
    s->dev = nullptr;
    s->calls = 0;
    s->count_pending = false;
    s->pending_count = 0;
    s->ahead_open = false;
    hipError_t e = hipGetDevice(&s->device);
    if (e == hipSuccess) e = hipMalloc((void**)&s->dev, sizeof(grx_state_words));
    if (e == hipSuccess) {
        grx_state_words init;
        for (int i = 0; i < 3; ++i) {
            init.count[i] = -1;  // count after "call -1" is 0
            init.first_match[i] = GRX_NO_MATCH;
        }
        init.err = GRX_NO_ERR;
        init.pad = 0;
        e = hipMemcpy(s->dev, &init, sizeof(init), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        grx_set_error("grx_state_create: %s", hipGetErrorString(e));
        if (s->dev) (void)hipFree(s->dev);
        delete s;
        return GRX_E_HIP;
    }
    *out = s;
    return GRX_OK;
}

extern "C" int grx_state_destroy(grx_state* s) {
    if (!s) return GRX_OK;
    (void)hipFree(s->dev);
    delete s;
    return GRX_OK;
}

static inline int prev_slot_of(const grx_state* s) { return (int)((s->calls + 2u) % 3u); }

extern "C" int grx_state_set_count(grx_state* s, int32_t count, void* stream) {
    GRX_REQUIRE(s != nullptr, GRX_E_INVALID, "grx_state_set_count: null state");
    (void)stream;  // nothing is enqueued: the next call's planning kernel is handed the value
    s->count_pending = true;
    s->pending_count = count;
    return GRX_OK;
}

extern "C" int grx_state_clear_error(grx_state* s, void* stream) {
    GRX_REQUIRE(s != nullptr, GRX_E_INVALID, "grx_state_clear_error: null state");
    hipLaunchKernelGGL(state_clear_error_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, s->dev);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

extern "C" int grx_state_read(grx_state* s, int32_t out[4], void* stream) {
    GRX_REQUIRE(s != nullptr && out != nullptr, GRX_E_INVALID, "grx_state_read: null pointer");
    grx_state_words w;
    GRX_CHECK_HIP(hipMemcpyAsync(&w, s->dev, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GRX_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    out[0] = s->count_pending ? s->pending_count : state_count_after(w, prev_slot_of(s));
    if (w.err == GRX_NO_ERR) {
        out[1] = GRX_ROWERR_NONE;
        out[2] = -1;
    } else {
        out[1] = (int32_t)(w.err & 3u);
        out[2] = (int32_t)(w.err >> 2);
    }
    out[3] = (int32_t)s->calls;
    return GRX_OK;
}

// ------------------------------------------------------ call preparation ---
// What has to be known about all rows before the streaming kernel starts: in the
// fixed-hint mode the trie node that governs each row (found by a wave-wide scan of
// the whole row), in the count-based mode only whether -- and from which row on --
// the shared counter is reset.
// Preparation of a count-based call whose rows nobody prepared ahead: the prefix_ids match
// reduction and the once-per-call bookkeeping.  The rows themselves are walked by the
// streaming kernel.
__global__ void count_match_rows_kernel(PlanCtx c, const int64_t* __restrict__ ids, int64_t ids_stride, int R, int T) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (r == 0) plan_count_bookkeeping(c);
    count_row_match(c, ids + (int64_t)r * ids_stride, T, r);
}

__global__ void plan_fixed_rows_kernel(PlanCtx c, const int64_t* __restrict__ ids, int64_t ids_stride, int R, int T,
                                       grx_row_plan* __restrict__ plan) {
    const int lane = threadIdx.x & (GRX_WAVE - 1);
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) / GRX_WAVE;
    if (r >= R) return;
    const grx_row_plan pl = plan_fixed_row(c, ids + (int64_t)r * ids_stride, T, lane);
    if (lane == 0) plan[r] = pl;
}

// Bookkeeping of a count-based call that has no rows to process.
__global__ void state_bookkeeping_kernel(grx_state_words* st, int slot, int count_set, int count_value) {
    const int prev = (slot + 2) % 3, next = (slot + 1) % 3;
    st->count[slot] = count_set ? count_value : state_count_after(*st, prev);
    st->first_match[next] = GRX_NO_MATCH;
}

extern "C" size_t grx_workspace_bytes(int32_t R) {
    const size_t rows = R > 0 ? (size_t)R : 0;
    return GRX_WS_PLAN_OFFSET + rows * sizeof(grx_row_plan);
}

// Number of workgroups that share one row.  Launches this small are latency-bound: what counts is
// that all workgroups are resident at once (a second round of workgroups waits for the first to end:
// 64 rows x 32 parts = 2048 workgroups took 37.6 us, 64 x 8 = 512 took 26.4 us) and that the merge of a
// row's partial results -- a chain of device-scope round trips that grows with the parts -- does not
// outweigh the shorter streams.  Measured under a replayed graph, whole steps, any number of parts
// (tools/sweep_parts.sh): up to about 900 workgroups per launch (256 rows: 3 parts 31.4 us, 4 parts 32.1;
// 400 rows: 2 parts 47.0, 3 parts 49.9; 512 rows: whole rows 41.5, 2 parts 44.0), at most four parts per
// row (64 rows: 4 parts 23.6 us, 5 parts 23.9, 6 parts 24.2, 8 parts 24.7; 128 rows: 3 / 4 / 5 / 6 parts
// 26.9 / 25.9 / 26.4 / 26.6) and six below 33 rows (16 rows: 4 / 5 / 6 / 8 parts 21.3 / 21.1 / 21.1 / 21.4;
// 32 rows: 22.1 / 21.5 / 21.6 / 21.9).  GRX_SPLIT_ROWS / GRX_SPLIT_TARGET / GRX_SPLIT_PARTS in the
// environment override the limits (tuning; read once).
int grx_split_factor(int R, int row_vecs) {
    static const int max_rows = [] { const char* e = getenv("GRX_SPLIT_ROWS"); return e && *e ? atoi(e) : 512; }();
    static const int target = [] { const char* e = getenv("GRX_SPLIT_TARGET"); return e && *e ? atoi(e) : 0; }();
    static const int env_parts = [] {
        const char* e = getenv("GRX_SPLIT_PARTS");
        const int v = e && *e ? atoi(e) : 0;
        return v < 0 ? 0 : (v > 32 ? 32 : v);
    }();
    if (R <= 0 || R > GRX_SPLIT_MAX_ROWS || R > max_rows) return 1;
    const int limit = target > 0 ? target : 900;
    int S = env_parts > 0 ? env_parts : (target > 0 ? 8 : (R <= 32 ? 6 : 4));
    if (S > GRX_SPLIT_MAX_PARTS / R) S = GRX_SPLIT_MAX_PARTS / R;
    if (S > limit / R) S = limit / R;
    if (S > row_vecs / 512) S = row_vecs / 512;  // every part streams at least two batches of loads
    return S < 1 ? 1 : S;
}

TrieView grx_trie_view(const grx_trie* trie) {
    TrieView tv;
    tv.off = trie->child_off;
    tv.tok = trie->child_tok;
    tv.hash = trie->edge_hash;
    tv.hash_mask = trie->hash_mask;
    tv.chain_n = trie->chain_n;
    tv.chain_node = trie->chain_node;
    for (int i = 0; i < GRX_CHAIN_MAX; ++i) tv.chain_tok[i] = i < trie->chain_n ? trie->chain_tok[i] : -1;
    return tv;
}

PrefixArg grx_prefix_arg(const grx_constraint* con) {
    PrefixArg pfx;
    memset(&pfx, 0, sizeof(pfx));
    pfx.n = con->n_prefix < 0 ? -1 : con->n_prefix;
    for (int i = 0; i < pfx.n; ++i) pfx.ids[i] = con->prefix_ids[i];
    return pfx;
}

static PlanCtx make_plan_ctx(const grx_trie* trie, grx_state* state, const grx_constraint* con, int slot) {
    PlanCtx c;
    memset(&c, 0, sizeof(c));
    c.st = state ? state->dev : nullptr;
    c.mode = con->mode;
    c.slot = slot;
    c.count_set = (state && state->count_pending) ? 1 : 0;
    c.count_value = state ? state->pending_count : 0;
    c.prefix_index = con->prefix_index;
    c.eos = con->eos_token_id;
    c.tv = grx_trie_view(trie);
    c.pfx = grx_prefix_arg(con);
    return c;
}

int grx_launch_plan_kernel(const PlanCtx& ctx, const int64_t* ids, int64_t ids_stride, int R, int T,
                           grx_row_plan* plan, hipStream_t stream) {
    const int threads = 256;
    if (R <= 0) return GRX_OK;
    if (ctx.mode == GRX_MODE_COUNT) {
        hipLaunchKernelGGL(count_match_rows_kernel, dim3((R + threads - 1) / threads), dim3(threads), 0, stream, ctx, ids,
                           ids_stride, R, T);
    } else {
        const int64_t total = (int64_t)R * GRX_WAVE;
        hipLaunchKernelGGL(plan_fixed_rows_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0,
                           stream, ctx, ids, ids_stride, R, T, plan);
    }
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

// Open a processor call on `stream`.  Count-based mode: pick the state slot the call owns, advance
// the host-side call counter and hand back the context with which the streaming kernel walks its
// rows; unless the rows arrive prepared (grx_rows.planned), the call is prepared here -- by the
// match kernel when prefix_ids is given, else by the streaming kernel itself (*bookkeep_out).
// Fixed-hint mode: enqueue the row planning kernel unless the rows arrive planned.
int grx_plan_rows(const grx_trie* trie, grx_state* state, const grx_rows* rows, const grx_constraint* con,
                  grx_row_plan* plan, hipStream_t stream, PlanCtx* ctx_out, int* bookkeep_out) {
    int slot = 0;
    *bookkeep_out = 0;
    if (con->mode == GRX_MODE_COUNT) {
        slot = (int)(state->calls % 3u);
        state->calls += 1;
    }
    // a counter set since the last call reaches this call (prepared ahead or now) and ends here
    const PlanCtx ctx = make_plan_ctx(trie, state, con, slot);
    *ctx_out = ctx;
    if (state) state->count_pending = false;
    if (con->mode == GRX_MODE_COUNT) {
        if (state->ahead_open && !rows->planned) {  // a preparation made ahead for this slot is not the one that runs
            hipLaunchKernelGGL(state_rearm_kernel, dim3(1), dim3(1), 0, stream, state->dev, slot);
            GRX_CHECK_HIP(hipGetLastError());
        }
        state->ahead_open = false;
        if (rows->R == 0) {
            // an empty call cannot arrive prepared: whoever prepared had no rows to attach the bookkeeping to
            hipLaunchKernelGGL(state_bookkeeping_kernel, dim3(1), dim3(1), 0, stream, state->dev, slot, ctx.count_set,
                               ctx.count_value);
            GRX_CHECK_HIP(hipGetLastError());
            return GRX_OK;
        }
        if (rows->planned) return GRX_OK;
        if (con->n_prefix < 0) {  // no match reduction to wait for: the streaming kernel does the bookkeeping
            *bookkeep_out = 1;
            return GRX_OK;
        }
    }
    if (rows->planned || rows->R == 0) return GRX_OK;
    return grx_launch_plan_kernel(ctx, rows->input_ids, rows->ids_stride, rows->R, rows->T, plan, stream);
}

int grx_plan_request_check(const char* who, const grx_plan_request* req, int32_t n_rows) {
    GRX_REQUIRE(req->trie && req->con, GRX_E_INVALID, "%s: plan request without trie / constraint", who);
    GRX_REQUIRE(req->con->mode == GRX_MODE_COUNT || req->con->mode == GRX_MODE_FIXED_HINT, GRX_E_INVALID,
                "%s: plan request with bad mode %d", who, req->con->mode);
    GRX_REQUIRE(req->con->mode != GRX_MODE_COUNT || req->state != nullptr, GRX_E_INVALID,
                "%s: planning a count-based call needs its grx_state", who);
    GRX_REQUIRE(req->con->n_prefix <= GRX_MAX_PREFIX, GRX_E_UNSUPPORTED, "%s: prefix_ids longer than %d tokens", who,
                GRX_MAX_PREFIX);
    GRX_REQUIRE(req->workspace != nullptr && ((uintptr_t)req->workspace & 15u) == 0 &&
                    req->workspace_bytes >= grx_workspace_bytes(n_rows),
                GRX_E_WORKSPACE, "%s: plan request needs a 16-byte aligned workspace of %zu bytes", who,
                grx_workspace_bytes(n_rows));
    return GRX_OK;
}

int grx_plan_ctx_for_next_call(const char* who, const grx_plan_request* req, int32_t n_rows, PlanCtx* ctx,
                               grx_row_plan** plan, hipStream_t stream) {
    const int rc = grx_plan_request_check(who, req, n_rows);
    if (rc != GRX_OK) return rc;
    const int slot = req->con->mode == GRX_MODE_COUNT ? (int)(req->state->calls % 3u) : 0;
    if (req->con->mode == GRX_MODE_COUNT) {
        if (req->state->ahead_open) {  // the plan made ahead before this one was never used
            hipLaunchKernelGGL(state_rearm_kernel, dim3(1), dim3(1), 0, stream, req->state->dev, slot);
            GRX_CHECK_HIP(hipGetLastError());
        }
        req->state->ahead_open = true;
    }
    *ctx = make_plan_ctx(req->trie, req->state, req->con, slot);
    *plan = reinterpret_cast<grx_row_plan*>(reinterpret_cast<char*>(req->workspace) + GRX_WS_PLAN_OFFSET);
    return GRX_OK;
}

int grx_check_step_args(const char* who, const grx_trie* trie, const grx_state* state, const grx_rows* rows,
                        const grx_constraint* con, const void* workspace, size_t workspace_bytes, int32_t vocab) {
    GRX_REQUIRE(trie && rows && con, GRX_E_INVALID, "%s: null trie/rows/constraint", who);
    GRX_REQUIRE(rows->R >= 0 && rows->T >= 0 && rows->V >= 1, GRX_E_INVALID, "%s: bad shape R=%d T=%d V=%d", who,
                rows->R, rows->T, rows->V);
    GRX_REQUIRE(con->mode == GRX_MODE_COUNT || con->mode == GRX_MODE_FIXED_HINT, GRX_E_INVALID, "%s: bad mode %d", who,
                con->mode);
    GRX_REQUIRE(con->mode != GRX_MODE_COUNT || state != nullptr, GRX_E_INVALID,
                "%s: the count-based mode needs a grx_state", who);
    GRX_REQUIRE(con->n_prefix <= GRX_MAX_PREFIX, GRX_E_UNSUPPORTED, "%s: prefix_ids longer than %d tokens", who,
                GRX_MAX_PREFIX);
    GRX_REQUIRE(rows->dtype == GRX_F32 || rows->dtype == GRX_BF16 || rows->dtype == GRX_F16, GRX_E_INVALID,
                "%s: bad dtype %d", who, rows->dtype);
    if (rows->R > 0) {
        GRX_REQUIRE(rows->scores != nullptr && (rows->T == 0 || rows->input_ids != nullptr), GRX_E_INVALID,
                    "%s: null input_ids/scores", who);
        GRX_REQUIRE(rows->ids_stride >= rows->T && rows->scores_stride >= rows->V, GRX_E_INVALID,
                    "%s: row stride smaller than the row", who);
        GRX_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15u) == 0, GRX_E_WORKSPACE,
                    "%s: a 16-byte aligned workspace is required", who);
        GRX_REQUIRE(workspace_bytes >= grx_workspace_bytes(rows->R), GRX_E_WORKSPACE,
                    "%s: workspace of %zu bytes, need %zu", who, workspace_bytes, grx_workspace_bytes(rows->R));
    }
    const int32_t width = vocab >= 0 ? vocab : rows->V;  // rows may be a slice of the vocabulary
    GRX_REQUIRE(trie->max_token < width, GRX_E_INVALID,
                "%s: the trie holds token id %d but rows have only V=%d columns", who, trie->max_token, width);
    return GRX_OK;
}
