// Trie upload, device-resident processor state, and the per-row resolution of
// the allowed-token node (integer work, one pass over the token rows).
#include <stdarg.h>
#include <string.h>

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
    hipError_t e = hipGetDevice(&t->device);
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
    delete t;
    return GRX_OK;
}

extern "C" int32_t grx_trie_max_fanout(const grx_trie* t) { return t ? t->max_fanout : 0; }
extern "C" int32_t grx_trie_max_token(const grx_trie* t) { return t ? t->max_token : -1; }

// ----------------------------------------------------------------- state ---
// set_count: make the NEXT call see `count` (its previous slot then reads count - 1, no match).
__global__ void state_set_kernel(grx_state_words* st, int set_count, int prev_slot, int32_t count, int clear_err) {
    if (set_count) {
        st->count[prev_slot] = count - 1;
        st->first_match[prev_slot] = GRX_NO_MATCH;
    }
    if (clear_err) st->err = GRX_NO_ERR;
}

extern "C" int grx_state_create(grx_state** out) {
    GRX_REQUIRE(out != nullptr, GRX_E_INVALID, "grx_state_create: null pointer");
    grx_state* s = new grx_state();
    s->dev = nullptr;
    s->calls = 0;
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
    hipLaunchKernelGGL(state_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, s->dev, 1, prev_slot_of(s), count,
                       0);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

extern "C" int grx_state_clear_error(grx_state* s, void* stream) {
    GRX_REQUIRE(s != nullptr, GRX_E_INVALID, "grx_state_clear_error: null state");
    hipLaunchKernelGGL(state_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, s->dev, 0, 0, 0, 1);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

extern "C" int grx_state_read(grx_state* s, int32_t out[4], void* stream) {
    GRX_REQUIRE(s != nullptr && out != nullptr, GRX_E_INVALID, "grx_state_read: null pointer");
    grx_state_words w;
    GRX_CHECK_HIP(hipMemcpyAsync(&w, s->dev, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GRX_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    out[0] = state_count_after(w, prev_slot_of(s));
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

// ------------------------------------------------------------ match scan ---
// Lowest row whose last prefix_index tokens equal prefix_ids
// (logit_processor.py:39-40): from that row on the shared counter is 0.
__global__ void match_scan_kernel(grx_state_words* st, int slot, const int64_t* __restrict__ ids, int64_t ids_stride,
                                  int R, int T, PrefixArg pfx, int prefix_index) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int64_t* row = ids + (int64_t)r * ids_stride;
    const int s0 = py_tail_start(prefix_index, T);
    bool match = (T - s0) == pfx.n;
    for (int i = 0; match && i < pfx.n; ++i) match = row[s0 + i] == pfx.ids[i];
    if (match) atomicMin(&st->first_match[slot], r);
}

// Bookkeeping of a count-based call that has no rows to process.
__global__ void state_bookkeeping_kernel(grx_state_words* st, int slot) {
    const int prev = (slot + 2) % 3, next = (slot + 1) % 3;
    st->count[slot] = state_count_after(*st, prev);
    st->first_match[next] = GRX_NO_MATCH;
}

extern "C" size_t grx_workspace_bytes(int32_t R) {
    (void)R;
    return 256;  // the current kernels keep all per-row state in registers and LDS
}

PrefixArg grx_prefix_arg(const grx_constraint* con) {
    PrefixArg pfx;
    memset(&pfx, 0, sizeof(pfx));
    pfx.n = con->n_prefix < 0 ? -1 : con->n_prefix;
    for (int i = 0; i < pfx.n; ++i) pfx.ids[i] = con->prefix_ids[i];
    return pfx;
}

// Open a count-based call: returns its slot and enqueues the prefix match scan.
int grx_begin_count_call(grx_state* state, const grx_rows* rows, const grx_constraint* con, hipStream_t stream,
                         int* slot_out) {
    const int slot = (int)(state->calls % 3u);
    *slot_out = slot;
    if (rows->R == 0) {
        hipLaunchKernelGGL(state_bookkeeping_kernel, dim3(1), dim3(1), 0, stream, state->dev, slot);
        GRX_CHECK_HIP(hipGetLastError());
    } else if (con->n_prefix >= 0) {
        const int threads = 256;
        hipLaunchKernelGGL(match_scan_kernel, dim3((rows->R + threads - 1) / threads), dim3(threads), 0, stream,
                           state->dev, slot, rows->input_ids, rows->ids_stride, rows->R, rows->T, grx_prefix_arg(con),
                           con->prefix_index);
        GRX_CHECK_HIP(hipGetLastError());
    }
    state->calls += 1;
    return GRX_OK;
}

int grx_check_step_args(const char* who, const grx_trie* trie, const grx_state* state, const grx_rows* rows,
                        const grx_constraint* con, const void* workspace, size_t workspace_bytes) {
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
        (void)workspace;  // reserved: the current kernels need no device scratch
        (void)workspace_bytes;
    }
    GRX_REQUIRE(trie->max_token < rows->V, GRX_E_INVALID,
                "%s: the trie holds token id %d but rows have only V=%d columns", who, trie->max_token, rows->V);
    return GRX_OK;
}
