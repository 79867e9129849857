// Internal helpers shared by the kernels of libgenrec_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "genrec_hip.h"

#define GRX_WAVE 64

// ---------------------------------------------------------------- errors ---
void grx_set_error(const char* fmt, ...);

#define GRX_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            grx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GRX_E_HIP;                                                                 \
        }                                                                                     \
    } while (0)

#define GRX_REQUIRE(cond, code, ...)    \
    do {                                \
        if (!(cond)) {                  \
            grx_set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

// ------------------------------------------------------- opaque handles ---
#define GRX_CHAIN_MAX 8

// One slot of the open-addressing edge table; parent == -1 marks an empty slot.
struct grx_edge_slot {
    int32_t parent;
    int32_t tok;
    int32_t child;
    int32_t pad;
};

__host__ __device__ __forceinline__ uint32_t grx_edge_hash(uint32_t parent, uint32_t tok) {
    uint32_t h = parent * 0x9E3779B1u ^ (tok * 0x85EBCA6Bu);
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}

struct grx_trie {
    int32_t* child_off;  // device [n_nodes + 1]
    int32_t* child_tok;  // device [n_edges]
    int32_t n_nodes;
    int32_t n_edges;
    int32_t max_fanout;
    int32_t max_token;
    int device;
    // Edge hash (parent, token) -> child for one-load lookups, and the unary chain
    // below the root (the assistant-prefix tokens every item starts with).
    struct grx_edge_slot* edge_hash;  // device [hash_mask + 1]
    uint32_t hash_mask;
    int32_t chain_n;
    int32_t chain_tok[GRX_CHAIN_MAX];
    int32_t chain_node;
    // Host copy of the distinct edge tokens, ascending: lets a call over a slice of the vocabulary
    // know whether any allowed token can lie inside it.
    int32_t* tok_sorted;
    int32_t n_tok_sorted;
};

// Device words behind grx_state.  The counter and the "a row ended in
// prefix_ids" marker are kept in a ring of three slots indexed by the call
// number: a call reads the previous slot, owns the current one, and re-arms the
// next one, so no kernel ever resets a word another kernel of the same call reads.
struct grx_state_words {
    int32_t count[3];        // window count in force at the start of the call that owns the slot
    int32_t first_match[3];  // lowest row whose tail equals prefix_ids in that call, or GRX_NO_MATCH
    uint32_t err;            // 0xFFFFFFFF = none, else (row << 2) | GRX_ROWERR_*
    int32_t pad;
};

struct grx_state {
    grx_state_words* dev;
    int device;
    uint32_t calls;          // count-based calls issued so far (host side, stream order)
    // grx_state_set_count only notes the value here; the kernel that plans the next call receives
    // it as an argument (no launch for a counter reset), and that call clears it.
    bool count_pending;
    int32_t pending_count;
    // A beam kernel has planned the next count-based call (and min-reduced row matches into that
    // call's first_match word) but the call has not come yet.  If the plan is dropped -- the next
    // call plans other rows itself, or another plan is made ahead first -- the word is re-armed.
    bool ahead_open;
};

// keep_top (grx_constraint): longest kept list, and the widest node whose children can be ranked in LDS
#define BS_KEEP_CAP 512
#define GRX_KEEP_TOP_MAX_FANOUT 1024

#define GRX_NO_MATCH 0x7FFFFFFF
#define GRX_NO_ERR 0xFFFFFFFFu

// Row resolution results.
#define GRX_RES_EMPTY (-1)    // trie.get(...) == []            -> GRX_ROWERR_EMPTY
#define GRX_RES_EOS (-2)      // row ends in EOS                -> allowed = {eos}
#define GRX_RES_NOMATCH (-3)  // no valid prefix occurrence     -> GRX_ROWERR_NOMATCH

// ------------------------------------------------------ device helpers ---
struct TrieView {
    const int32_t* __restrict__ off;
    const int32_t* __restrict__ tok;
    const grx_edge_slot* __restrict__ hash;
    uint32_t hash_mask;
    int32_t chain_n;
    int32_t chain_node;
    int32_t chain_tok[GRX_CHAIN_MAX];
};

// Child of `node` along token `t`, or -1.  One 16-byte load per probe of the
// edge table (load factor <= 1/2), on wave-uniform addresses.
__device__ __forceinline__ int trie_child(const TrieView& tv, int node, int64_t t) {
    if (t < 0 || t > 0x7FFFFFFFLL) return -1;
    const int32_t key = (int32_t)t;
    uint32_t h = grx_edge_hash((uint32_t)node, (uint32_t)key) & tv.hash_mask;
    for (;;) {
        const grx_edge_slot s = tv.hash[h];
        if (s.parent == node && s.tok == key) return s.child;
        if (s.parent < 0) return -1;
        h = (h + 1) & tv.hash_mask;
    }
}

// Node reached from the root along ids[0..n), or GRX_RES_EMPTY when the path
// leaves the trie or ends on a node without children (Trie.get returns []).
// Rows normally start with the unary chain below the root, which is compared
// against the kernel arguments instead of being walked.
__device__ __forceinline__ int trie_walk(const TrieView& tv, const int64_t* __restrict__ ids, int n) {
    int node = 0, i = 0;
    if (tv.chain_n > 0 && n >= tv.chain_n) {
        bool same = true;
#pragma unroll
        for (int j = 0; j < GRX_CHAIN_MAX; ++j)
            if (j < tv.chain_n) same = same && ids[j] == (int64_t)tv.chain_tok[j];
        if (same) {
            node = tv.chain_node;
            i = tv.chain_n;
        }
    }
    for (; i < n; ++i) {
        node = trie_child(tv, node, ids[i]);
        if (node < 0) return GRX_RES_EMPTY;
    }
    return (tv.off[node + 1] > tv.off[node]) ? node : GRX_RES_EMPTY;
}

// Start index of the Python slice sent[-k:] on a row of T tokens.
__device__ __forceinline__ int py_tail_start(int k, int T) {
    if (k > 0) return (T - k > 0) ? T - k : 0;
    const int s = -k;
    return s < T ? s : T;
}

// Wave-wide reductions on the data-parallel-primitive path of the vector ALU: four row shifts fold
// each row of 16 lanes into its last lane, two row broadcasts fold the four rows into lane 63, whose
// value is read back as a scalar.  Six dependent ALU operations instead of six LDS permutes (a
// shuffle costs an LDS round trip each, and these sit at the very end of a workgroup's life, when its
// slot issues no loads).  Every lane must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float ident, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

__device__ __forceinline__ float wave_max(float v) {
    const float ninf = -__builtin_inff();
    v = fmaxf(v, dpp_take<0x111, 0xF>(ninf, v));  // row_shr:1
    v = fmaxf(v, dpp_take<0x112, 0xF>(ninf, v));  // row_shr:2
    v = fmaxf(v, dpp_take<0x114, 0xF>(ninf, v));  // row_shr:4
    v = fmaxf(v, dpp_take<0x118, 0xF>(ninf, v));  // row_shr:8
    v = fmaxf(v, dpp_take<0x142, 0xA>(ninf, v));  // row_bcast:15 into rows 1 and 3
    v = fmaxf(v, dpp_take<0x143, 0xC>(ninf, v));  // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_take<0x111, 0xF>(0.f, v);
    v += dpp_take<0x112, 0xF>(0.f, v);
    v += dpp_take<0x114, 0xF>(0.f, v);
    v += dpp_take<0x118, 0xF>(0.f, v);
    v += dpp_take<0x142, 0xA>(0.f, v);
    v += dpp_take<0x143, 0xC>(0.f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFull), o, GRX_WAVE);
        const unsigned hi = __shfl_xor((unsigned)(v >> 32), o, GRX_WAVE);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w > v ? w : v;
    }
    return v;
}

// Monotone map float -> uint32 (larger float => larger integer; -0 < +0).
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

__device__ __forceinline__ void report_row_error(grx_state_words* st, int row, int code) {
    if (st) atomicMin(&st->err, ((uint32_t)row << 2) | (uint32_t)code);
}

// Count the reference's shared counter holds after the call that owns `slot`:
// reset to 0 if some row ended in prefix_ids, then incremented once.
__device__ __host__ __forceinline__ int state_count_after(const grx_state_words& w, int slot) {
    return (w.first_match[slot] != GRX_NO_MATCH ? 0 : w.count[slot]) + 1;
}

struct PrefixArg {
    int32_t n;  // -1 = None
    int64_t ids[GRX_MAX_PREFIX];
};

// Fixed-hint resolution of one row by one full wave (fixed_hint_logit_processor.py:18-61):
// 64 start positions are tested for the prefix at a time and the occurrences are tried
// left to right until one continues inside the trie.  Returns a node id or GRX_RES_*.
__device__ __forceinline__ int resolve_fixed_row(const TrieView& tv, const int64_t* __restrict__ row, int T,
                                                 const PrefixArg& pfx, int eos, int lane) {
    if (eos >= 0 && T > 0 && row[T - 1] == (int64_t)eos) return GRX_RES_EOS;
    if (pfx.n <= 0) return trie_walk(tv, row, T);  // start = 0 when prefix_ids is empty or None
    const int n = pfx.n;
    for (int base = 0; base + n <= T; base += GRX_WAVE) {
        const int s = base + lane;
        bool m = s + n <= T;
        for (int i = 0; m && i < n; ++i) m = row[s + i] == pfx.ids[i];
        unsigned long long mask = __ballot(m);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int start = base + b;
            const int node = trie_walk(tv, row + start, T - start);
            if (node >= 0) return node;
        }
    }
    return GRX_RES_NOMATCH;
}

// ------------------------------------------------------------ row plans ---
// Plan of one row for the streaming kernel (fixed-hint mode): the governing trie node (or a
// negative GRX_RES_* code) and its child range.
struct grx_row_plan {
    int32_t code;
    int32_t off;
    int32_t n;
    int32_t pad;
};

// What a kernel needs to plan rows for a processor call (kernel argument, by value).
struct PlanCtx {
    grx_state_words* st;  // null in fixed-hint mode without a state
    int mode;             // GRX_MODE_*
    int slot;             // state slot the planned call owns (count mode)
    int count_set;        // 1: the caller set the counter since the last call; it starts at count_value
    int count_value;
    int count_ready;      // 1: count[slot] already holds the count this call starts with (whoever prepared the
                          //    call recorded it), so nothing of the previous call's slot is read
    int prefix_index;
    int eos;
    TrieView tv;
    PrefixArg pfx;
};

// Count-based mode (logit_processor.py:38-44).  The shared counter a call starts with is the
// counter after the previous call, unless the caller set it in between.
__device__ __forceinline__ int plan_start_count(const PlanCtx& c) {
    if (c.count_ready) return c.st->count[c.slot];
    return c.count_set ? c.count_value : state_count_after(*c.st, (c.slot + 2) % 3);
}

// Preparation of a count-based call, one thread per row: rows whose last prefix_index tokens
// equal prefix_ids reset the shared counter for themselves and every later row
// (logit_processor.py:39-40); the lowest such row is min-reduced into first_match[slot].
// Nothing to do when prefix_ids is None.
__device__ __forceinline__ void count_row_match(const PlanCtx& c, const int64_t* __restrict__ row, int T, int r) {
    if (c.pfx.n < 0) return;
    const int s0 = py_tail_start(c.prefix_index, T);
    bool match = (T - s0) == c.pfx.n;
    for (int i = 0; match && i < c.pfx.n; ++i) match = row[s0 + i] == c.pfx.ids[i];
    if (match) atomicMin(&c.st->first_match[c.slot], r);
}

// The node that governs row r of a prepared count-based call (first_match[slot] is final): the
// window is the last prefix_index + count tokens, count = 0 from the first matching row on
// (logit_processor.py:39-41).  Every operand is uniform over the workgroup that calls it.
__device__ __forceinline__ int count_row_node(const PlanCtx& c, const int64_t* __restrict__ row, int T, int r) {
    int count = plan_start_count(c);
    if (c.pfx.n >= 0 && r >= c.st->first_match[c.slot]) count = 0;
    const int start = py_tail_start(c.prefix_index + count, T);
    return trie_walk(c.tv, row + start, T - start);
}

// Once per planned count-based call: record the counter the call runs with and re-arm the
// match word of the call after it.  No kernel of the planned call reads these two words.
__device__ __forceinline__ void plan_count_bookkeeping(const PlanCtx& c) {
    const int count = plan_start_count(c);
    c.st->count[c.slot] = count;
    c.st->first_match[(c.slot + 1) % 3] = GRX_NO_MATCH;
}

// Fixed-hint mode (fixed_hint_logit_processor.py:43-61), one wave per row; every lane
// returns the same plan.
__device__ __forceinline__ grx_row_plan plan_fixed_row(const PlanCtx& c, const int64_t* __restrict__ row, int T,
                                                       int lane) {
    const int code = resolve_fixed_row(c.tv, row, T, c.pfx, c.eos, lane);
    grx_row_plan pl;
    pl.code = code;
    pl.off = code >= 0 ? c.tv.off[code] : 0;
    pl.n = code >= 0 ? c.tv.off[code + 1] - pl.off : (code == GRX_RES_EOS ? 1 : 0);
    pl.pad = 0;
    return pl;
}
