// Masked log-softmax over logit rows: the HBM-bound pass of the constrained
// decoding step.  One 256-thread workgroup streams one row exactly once with
// 16-byte loads, keeping an online (max, sum-of-exp) pair per thread; the
// allowed tokens of the row (children of its trie node) are gathered while the
// stream is in flight and finished once the row's log-sum-exp is known.
#include <hip/hip_bf16.h>
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>

#include "common.h"
#include "resolve.h"

#include "lse_math.h"
#ifdef FUSE_STAMP
// Diagnostic build only: phase timestamps (100 MHz counter) of the beam steps that run inside the streaming kernel.
__device__ unsigned long long g_fuse_stamps[1024 * 16];
#define STAMP(n) do { if (tid == 0 && b < 1024) g_fuse_stamps[b * 16 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int grx_debug_read_fuse_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fuse_stamps), sizeof(unsigned long long) * 1024 * 16);
}
#endif
#include "beam_body.h"

enum { OUT_COMPACT = 0, OUT_FULL_F32 = 1, OUT_FULL_SAME = 2 };

#ifndef PEN_STASH
#define PEN_STASH 4  // 16-byte LDS slots per thread for vectors that hold penalised tokens (repetition penalty)
#endif
#ifndef PEN_POOL
#define PEN_POOL 48  // ... and spare slots shared by the workgroup, for the few threads that meet more
#endif
#ifndef MLS_HEAD_BATCHES
#define MLS_HEAD_BATCHES 4  // batches of MLS_UNROLL loads per thread that may be issued before a row's trie node is looked up
#endif

struct MlsArgs {
    const int32_t* tok;    // child tokens of the trie (CSR)
    const int32_t* off;    // child offsets of the trie (CSR)
    const grx_row_plan* plan;  // fixed-hint mode: [R] from the planning kernel of this call
    grx_state_words* st;   // may be null (fixed-hint without error reporting)
    int walk;              // count mode: every workgroup works out its row's node itself (pc, ids, T) ...
    int bookkeep;          // ... and workgroup 0 records the call's counter / re-arms the next match word
    PlanCtx pc;
    int R;
    const void* scores;
    int64_t scores_stride;
    int V;
    int eos;
    void* out;             // full modes
    int64_t out_stride;
    grx_cand* cand;        // compact mode: [R, cand_stride] rows of {token, logp} behind a {count, lse} header
    int64_t cand_stride;
    int cmax;
    // Split rows (few rows): a row is streamed by S workgroups; the last one to
    // finish merges the S partial (max, sum) pairs and emits the row.
    // Repetition penalty (0: off) over the tokens of ids[r, pen_from:T).
    float penalty;
    int pen_from;
    int pen_shared;        // tokens [0, pen_shared) of a row equal those of the first row of its group of pen_group rows
    int pen_group;
    int pen_bm_bytes;      // bytes of the LDS token bitmap (one bit per token of V, rounded up to 16)
    int pen_vm_bytes;      // bytes of the LDS vector masks (one bit per 16-byte vector of a row, in words per thread)
    const int64_t* ids;
    int64_t ids_stride;
    int T;
    int S;
    int skip_stream;       // rows are log-softmax values already (log-sum-exp 0): gather only
    int keep_top;          // compact mode, > 0: rows with more candidates keep only their keep_top best
    const float* row_bias; // ... ranked by logp + row_bias[r] (null: 0), the sum the beam step ranks by
    int* split_count;      // [R] arrival counters, zero between calls
    float2* split_part;    // [R * S][2] partial (max, sum) and, with the penalty, (sub, add)
    // Dense output kept by the caller from call to call: last[r] is the node whose tokens row r of
    // `out` holds finite values for (written by every such call).  prefilled: `out` is -inf everywhere
    // else already, so instead of a fill pass the kernel resets just those positions.
    grx_row_plan* last;
    int prefilled;
    int head_batches;      // opening batches of loads per thread (<= MLS_HEAD_BATCHES)
};


// Output element of a dense mode: fp32, or the input element type.
template <int DT, int OUT>
__device__ __forceinline__ auto to_out(float v) {
    if constexpr (OUT == OUT_FULL_F32) return v;
    else return Elem<DT>::from_float(v);
}

#ifdef MLS_STAMP
// Diagnostic build only: phase timestamps (100 MHz counter) of the first 4096 workgroups.
__device__ unsigned long long g_mls_stamps[4096 * 8];
#define MSTAMP(n) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_mls_stamps[blockIdx.x * 8 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int grx_debug_read_mls_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mls_stamps), sizeof(unsigned long long) * 4096 * 8);
}
#else
#define MSTAMP(n) do { } while (0)
#endif

// ---- the keep_top best of a row ---------------------------------------------------
// One wave picks the `keep` largest of the n keys in s_key (n <= 64 * Q; ties: lower index first) and
// stores {token, logp} of each at entry 1 + its slot of the candidate row, slots in no particular
// order.  Lane l holds keys l, l + 64, ...: the keep-th largest key is built bit by bit from ballot
// counts (32 rounds of Q compares, nothing leaves the registers), keys above it take the first slots
// in candidate order, keys equal to it the remaining ones.
template <int Q, bool COH>
__device__ __forceinline__ void select_top(const uint32_t* s_key, const float* s_lp, const int32_t* __restrict__ toks,
                                           int n, int keep, int lane, grx_cand* __restrict__ crow) {
    uint32_t key[Q];
    bool in[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        in[q] = lane + q * GRX_WAVE < n;
        key[q] = in[q] ? s_key[lane + q * GRX_WAVE] : 0u;
    }
    uint32_t T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) cnt += __popcll(__ballot(in[q] && key[q] >= trial));
        if (cnt >= keep) T = trial;
    }
    int n_gt = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) n_gt += __popcll(__ballot(in[q] && key[q] > T));
    const unsigned long long below = (1ull << lane) - 1ull;
    int gt_seen = 0, eq_seen = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = lane + q * GRX_WAVE;
        const bool gt = in[q] && key[q] > T, eq = in[q] && key[q] == T;
        const unsigned long long m_gt = __ballot(gt), m_eq = __ballot(eq);
        int pos = -1;
        if (gt) pos = gt_seen + __popcll(m_gt & below);
        if (eq) {
            const int er = eq_seen + __popcll(m_eq & below);
            if (n_gt + er < keep) pos = n_gt + er;
        }
        if (pos >= 0) {
            grx_cand v;
            v.x = toks[i];
            v.y = __float_as_int(s_lp[i]);
            cand_store<COH>(crow + 1 + pos, v);
        }
        gt_seen += __popcll(m_gt);
        eq_seen += __popcll(m_eq);
    }
}

// ---- the kernel --------------------------------------------------------------
// DT: element type of the logits.  OUT: OUT_*.  VEC: rows (and, for the full
// modes, output rows) are 16-byte aligned so the stream uses 16-byte accesses;
// otherwise one element per access.
// FUSE (compact rows only): the workgroup that completes a prompt group -- the last of its K rows to be
// emitted -- runs the group's beam step (`fb`) on the spot, while other groups still stream.
struct FusedBeam {
    grx_beam_step_args a;
    BeamDerived d;
    PlanCtx pc;              // the processor call AFTER this one, planned by the beam step
    grx_row_plan* plan_out;
};

#ifndef MLS_ABL_COHSTORE
#define MLS_ABL_COHSTORE 0  // diagnostic: 1 = the unfused kernel writes its candidate rows with device-scope stores too
#endif
#ifndef MLS_ABL_TAILWAIT
#define MLS_ABL_TAILWAIT 0  // diagnostic: 1 = the unfused kernel ends every row like the fused one (store wait, barrier, atomic), 2 = without the store wait
#endif

template <int DT, int OUT, bool VEC, bool SPLIT, bool PEN, bool FUSE>
__device__ __forceinline__ void mls_body(const MlsArgs& a, const FusedBeam* fb) {
    constexpr bool COHST = FUSE || MLS_ABL_COHSTORE;
    typedef Elem<DT> E;
    typedef typename E::T T;
    constexpr int EPV = VEC ? E::kPerVec : 1;  // elements per access
    __shared__ float s_m[MLS_WAVES], s_s[MLS_WAVES];
#ifdef MLS_LDS_PAD
    // occupancy experiment: extra LDS per workgroup lowers the number of workgroups a CU holds
    __shared__ int s_pad[MLS_LDS_PAD / 4];
    if (a.R < 0) s_pad[threadIdx.x] = a.V;
#endif

    const int r = SPLIT ? (int)blockIdx.x / a.S : (int)blockIdx.x;
    const int part = SPLIT ? (int)blockIdx.x - r * a.S : 0;
    const int tid = threadIdx.x;
    const T* __restrict__ row = reinterpret_cast<const T*>(a.scores) + (int64_t)r * a.scores_stride;
    MSTAMP(0);
#ifdef FUSE_STAMP
    if (FUSE && blockIdx.x == 0 && tid == 0) g_fuse_stamps[15] = __builtin_amdgcn_s_memrealtime();  // the launch's first workgroup
#endif
#ifdef MLS_LDS_PAD
    if (a.R < 0 && s_pad[(tid * 7) % (MLS_LDS_PAD / 4)] == 3) return;
#endif

    // The opening batches of the stream are issued before anything else: the node
    // look-up below runs on scalar loads, which do not queue behind them.
    const int nvec = a.V / EPV;
    int v_lo = 0, v_hi = nvec;  // this workgroup's share of the row's vectors
    if (SPLIT) {
        const int per = (nvec + a.S - 1) / a.S;
        v_lo = part * per < nvec ? part * per : nvec;
        v_hi = v_lo + per < nvec ? v_lo + per : nvec;
    }
    const RowStream vp(row, (int64_t)nvec * 16);
    // Repetition penalty: the row's own tokens are asked for first, so that they arrive ahead of the
    // opening batches below (loads return in order) and the token bitmap is ready when those land.
    constexpr int PEN_PER_THREAD = GRX_MAX_PENALTY_TOKENS / MLS_THREADS;
    static_assert(PEN_PER_THREAD * MLS_THREADS == GRX_MAX_PENALTY_TOKENS, "penalty tokens per thread");
    int64_t p_tok[PEN_PER_THREAD];
    auto pen_load_ids = [&]() {
        const int64_t* __restrict__ ids_row = a.ids + (int64_t)r * a.ids_stride;
        // the part of the row that the whole group shares is read from the group's first row: the other
        // rows of the group that this XCD handles find it in the L2
        const int64_t* __restrict__ ids_first = a.ids + (int64_t)(r - r % a.pen_group) * a.ids_stride;
#pragma unroll
        for (int k = 0; k < PEN_PER_THREAD; ++k) {
            // uniform guard, clamped index: no lane waits on a branch for its token (a token loaded twice
            // through the clamp is marked twice, which changes nothing)
            const int t0 = a.pen_from + k * MLS_THREADS;
            p_tok[k] = -1;
#ifndef PEN_ABL_NOIDS
            if (t0 < a.T) {
                const int t = t0 + tid < a.T ? t0 + tid : a.T - 1;
                p_tok[k] = (t < a.pen_shared ? ids_first : ids_row)[t];
            }
#endif
        }
    };
    if constexpr (PEN && VEC) pen_load_ids();
    // The opening batches: 64 KB of the row per workgroup are in flight while the node look-up below waits
    // for its dependent scalar loads (about 10 us under load).  With one batch the memory system ran dry
    // behind every look-up -- all of them at once when a launch starts: 370.2 -> 366.5 us per launch.
    constexpr int HEADN = MLS_UNROLL * MLS_HEAD_BATCHES;
    u32x4 v_head[HEADN];
    const int n_full = (v_hi - v_lo) / (MLS_UNROLL * MLS_THREADS);  // batches in which every thread takes part
    const int hbn = (VEC && !a.skip_stream) ? (n_full < a.head_batches ? n_full : a.head_batches) : 0;
#pragma unroll
    for (int hb = 0; hb < MLS_HEAD_BATCHES; ++hb) {
        if (hb < hbn) {
#pragma unroll
            for (int u = 0; u < MLS_UNROLL; ++u)
                v_head[hb * MLS_UNROLL + u] = vp.load(v_lo + tid + (hb * MLS_UNROLL + u) * MLS_THREADS);
        }
    }

    // Which trie node governs this row.  Count-based mode: walked here -- every operand is uniform
    // over the workgroup, so the walk is scalar loads and scalar arithmetic that run beside the
    // stream loads already in flight, and the kernel's duration is set by the stream either way.
    // Fixed-hint mode: planned by the kernel that ran just before (one 16-byte scalar load).
    int code, c_off, c_n;
    if (a.walk) {
        code = count_row_node(a.pc, a.ids + (int64_t)r * a.ids_stride, a.T, r);
        c_off = code >= 0 ? a.off[code] : 0;
        c_n = code >= 0 ? a.off[code + 1] - c_off : 0;
        if (a.bookkeep && blockIdx.x == 0 && tid == 0) plan_count_bookkeeping(a.pc);
    } else {
        const grx_row_plan pl = a.plan[r];
        code = pl.code;
        c_off = pl.off;
        c_n = pl.n;
    }
    MSTAMP(1);
    if (code < 0 && code != GRX_RES_EOS && tid == 0 && part == 0) {
        report_row_error(a.st, r, code == GRX_RES_NOMATCH ? GRX_ROWERR_NOMATCH : GRX_ROWERR_EMPTY);
    }
    // compact rows keep either every candidate or, past keep_top of them, the keep_top best
    // (only rows that shed at least GRX_KEEP_TOP_SLACK entries: the selection holds the workgroup's slot
    // for about a microsecond, which a few saved entries do not pay for)
    const bool trunc = OUT == OUT_COMPACT && a.keep_top > 0 && c_n > a.keep_top + GRX_KEEP_TOP_SLACK;
    if (OUT == OUT_COMPACT && (trunc ? a.keep_top : c_n) > a.cmax) {
        if (tid == 0 && part == 0) report_row_error(a.st, r, GRX_ROWERR_OVERFLOW);
        c_n = 0;
    }

    // Gather the first allowed logits now so they overlap the stream.  In compact mode
    // a thread owns MLS_EARLY adjacent entries e = tid * MLS_EARLY + k, i.e. candidates
    // e - 1 (entry 0 is the row header), which it later writes with one 16-byte store.
    // A split row is emitted by whichever workgroup finishes last, which gathers then (gathering in
    // every part up front instead changed nothing: 25.96 vs 26.04 us per step at 64 rows).
    constexpr int SHIFT = (OUT == OUT_COMPACT) ? 1 : 0;
    int e_tok[MLS_EARLY];
    T e_raw[MLS_EARLY];  // as loaded; converted where it is used, so that nothing here waits for it
    auto gather = [&]() {
        // Straight-line code, two round trips at most: every lane loads a token (index clamped into the
        // node's child range), then the logit of that token; lanes without a candidate drop the result.
        const bool have = code >= 0 && c_n > 0;
        int tk[MLS_EARLY];
#pragma unroll
        for (int k = 0; k < MLS_EARLY; ++k) {
            const int i = (OUT == OUT_COMPACT) ? tid * MLS_EARLY + k - SHIFT : tid + k * MLS_THREADS;
            tk[k] = 0;
            if (have) tk[k] = a.tok[c_off + (i < 0 ? 0 : (i < c_n ? i : c_n - 1))];
        }
#pragma unroll
        for (int k = 0; k < MLS_EARLY; ++k) {
            const int i = (OUT == OUT_COMPACT) ? tid * MLS_EARLY + k - SHIFT : tid + k * MLS_THREADS;
            if (code == GRX_RES_EOS) tk[k] = a.eos;
            e_tok[k] = (i >= 0 && i < c_n) ? tk[k] : -1;
#ifdef MLS_ABL_NOGATHER
            e_raw[k] = 0;
#else
            e_raw[k] = row[e_tok[k] < 0 ? 0 : e_tok[k]];
#endif
        }
    };
    if (!SPLIT) gather();

    // Repetition penalty.  The distinct tokens of ids[r, pen_from:T) are marked in an LDS bitmap over
    // the vocabulary before the stream starts.  The second normaliser needs the raw logit of each of
    // them, and every one passes through the registers of exactly one thread while the row streams.
    // A thread knows which of its own vectors hold a marked token from a bit mask it keeps in a
    // register (one LDS word per 32 of its vectors): the common case, nothing marked in a batch, is a
    // shift and a compare, and a marked vector is put aside whole -- one 16-byte LDS store into the
    // thread's own PEN_STASH slots, nothing else inside the loop.  What was put aside is taken apart
    // after the stream, when the row's maximum and sum are known; a thread with more marked vectors
    // than slots (a handful per row at a few hundred tokens) reads the surplus from memory again.
    // No other logit is read twice.  Keys and log-probs of the keep_top selection reuse the same LDS
    // once the bitmap is final.
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ uint32_t s_sel[PEN ? 4 : GRX_KEEP_TOP_MAX_FANOUT * 2];
    __shared__ float s_pen_red[2][MLS_WAVES];
    constexpr bool pen_on = PEN;  // a.penalty > 0: its own instantiation, so the others carry none of this
    constexpr bool pen_stream = PEN && VEC;  // marked values are caught in the stream (a split row: in each part's share)
    uint32_t* const s_bm = reinterpret_cast<uint32_t*>(s_dyn);
    const unsigned char* const s_bm8 = s_dyn;
    // bit j % 32 of word [j / 32][tid] stands for vector tid + j * MLS_THREADS of the row
    uint32_t* const s_vmask = reinterpret_cast<uint32_t*>(s_dyn + (PEN ? a.pen_bm_bytes : 0));
    u32x4* const s_stash = reinterpret_cast<u32x4*>(s_dyn + (PEN ? a.pen_bm_bytes + a.pen_vm_bytes : 0));
    // the workgroup's spare slots: a thread whose own slots are full takes up to two of these (numbered by
    // an LDS counter) before it falls back on reading a vector from memory a second time
    u32x4* const s_pool = s_stash + PEN_STASH * MLS_THREADS;
    __shared__ int s_pool_n;
    int pool_slot0 = -1, pool_slot1 = -1;
    const float pen_l2 = a.penalty * kLog2e;
    // sub = sum exp(x_u - m) and add = sum exp(p (x_u - m - log s)) over the distinct penalised tokens u
    // (log-probs are never positive, so the penalty always multiplies)
    float pen_sub = 0.f, pen_add = 0.f;
    auto pen_value = [&](float x, float m, float log_s) {
        if (x == -INFINITY) return;
        pen_sub += fast_exp2((x - m) * kLog2e);
        pen_add += fast_exp2((x - m - log_s) * pen_l2);
    };
    // Marks the row's tokens.  FETCH (rows that this workgroup did not stream whole itself: split rows,
    // rows without 16-byte alignment; called once m and log s are known): the thread that marks a
    // token first reads its logit from memory.
    auto pen_mark = [&](bool fetch, float m, float log_s) {
        u32x4* const bm4 = reinterpret_cast<u32x4*>(s_dyn);
        const u32x4 z = {0u, 0u, 0u, 0u};
        const int n16 = (a.pen_bm_bytes + (fetch ? 0 : a.pen_vm_bytes)) / 16;
        for (int i = tid; i < n16; i += MLS_THREADS) bm4[i] = z;
        if (tid == 0) s_pool_n = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PEN_PER_THREAD; ++k) {
            if (p_tok[k] < 0 || p_tok[k] >= (int64_t)a.V) continue;
            const int tokv = (int)p_tok[k];
            const uint32_t bit = 1u << (tokv & 31);
            if (fetch) {
                const uint32_t old = atomicOr(&s_bm[tokv >> 5], bit);
                if (!(old & bit)) pen_value(E::to_float(row[tokv]), m, log_s);
            } else {
                atomicOr(&s_bm[tokv >> 5], bit);
                const int iv = tokv / EPV - v_lo;  // position among the vectors this workgroup streams
                if (iv >= 0 && iv < v_hi - v_lo) {
                    const int j = iv / MLS_THREADS;
                    atomicOr(&s_vmask[(j >> 5) * MLS_THREADS + (iv - j * MLS_THREADS)], 1u << (j & 31));
                }
            }
        }
        __syncthreads();
    };
    if (pen_stream) pen_mark(false, 0.f, 0.f);
    // marked elements of vector iv: one bit per element
    auto pen_bits = [&](int iv) -> uint32_t {
        if constexpr (E::kPerVec == 8) return s_bm8[iv];
        else return (s_bm8[iv >> 1] >> ((iv & 1) * 4)) & 0xFu;
    };
    int pen_n_hit = 0;  // marked vectors this thread has met in the stream so far
    auto pen_put_aside = [&](const u32x4& vec) {
        if (pen_n_hit < PEN_STASH) {
            s_stash[pen_n_hit * MLS_THREADS + tid] = vec;
        } else if (pen_n_hit < PEN_STASH + 2) {
            const int slot = atomicAdd(&s_pool_n, 1);
            if (slot < PEN_POOL) {
                s_pool[slot] = vec;
                if (pen_n_hit == PEN_STASH) pool_slot0 = slot;
                else pool_slot1 = slot;
            }
        }
        ++pen_n_hit;
    };

    MSTAMP(2);
    // Stream the row once.
    MaxSum acc;
    acc.m = -INFINITY;
    acc.s = 0.f;
    acc.c = 0.f;

    if (VEC && !a.skip_stream) {
        // The dense output of a vectorised launch was filled with -inf by
        // fill_rows_kernel beforehand (write-only and read-only traffic run
        // faster back to back than interleaved), so the stream only reads.
        uint32_t vm = 0;  // the word of s_vmask that covers the current batch
        auto consume = [&](const u32x4* v, int i, int jb) {  // batch jb of this thread: its vectors 4 jb .. 4 jb + 3
            float f[MLS_UNROLL * E::kPerVec];
#pragma unroll
            for (int u = 0; u < MLS_UNROLL; ++u) E::unpack(v[u], f + u * E::kPerVec);
            accumulate<MLS_UNROLL * E::kPerVec>(acc, f);
#ifndef PEN_ABL_NOTEST
            if constexpr (pen_stream) {
                static_assert(MLS_UNROLL == 4, "eight batches of four vectors per mask word");
                if ((jb & 7) == 0) vm = s_vmask[(jb >> 3) * MLS_THREADS + tid];
                const uint32_t hit = (vm >> ((jb & 7) * 4)) & 0xFu;
                if (hit) {
#pragma unroll
                    for (int u = 0; u < MLS_UNROLL; ++u) {
                        if (hit & (1u << u)) pen_put_aside(v[u]);
                    }
                }
            }
#endif
        };

        int i = v_lo + tid, jb = 0;
#pragma unroll
        for (int hb = 0; hb < MLS_HEAD_BATCHES; ++hb) {
            if (hb < hbn) {
                consume(v_head + hb * MLS_UNROLL, i, hb);
                i += MLS_UNROLL * MLS_THREADS;
                jb = hb + 1;
            }
        }
        for (; i + (MLS_UNROLL - 1) * MLS_THREADS < v_hi; i += MLS_UNROLL * MLS_THREADS, ++jb) {
            u32x4 v[MLS_UNROLL];
#pragma unroll
            for (int u = 0; u < MLS_UNROLL; ++u) v[u] = vp.load(i + u * MLS_THREADS);
            consume(v, i, jb);
        }
        // What is left of the share: at most MLS_UNROLL - 1 vectors per thread.  Asked for together (one round
        // trip instead of one per vector, which is what a latency-bound launch of split rows feels); places
        // beyond the share read as -inf, which adds nothing.
        if (i < v_hi) {
            u32x4 v[MLS_UNROLL - 1];
#pragma unroll
            for (int u = 0; u < MLS_UNROLL - 1; ++u) {
                const u32x4 ninf = {E::kNegInfPacked, E::kNegInfPacked, E::kNegInfPacked, E::kNegInfPacked};
                v[u] = ninf;
                if (i + u * MLS_THREADS < v_hi) v[u] = vp.load(i + u * MLS_THREADS);
            }
            float f[(MLS_UNROLL - 1) * E::kPerVec];
#pragma unroll
            for (int u = 0; u < MLS_UNROLL - 1; ++u) E::unpack(v[u], f + u * E::kPerVec);
            accumulate<(MLS_UNROLL - 1) * E::kPerVec>(acc, f);
            if constexpr (pen_stream) {
#pragma unroll
                for (int u = 0; u < MLS_UNROLL - 1; ++u) {
                    const int j = (i - v_lo) / MLS_THREADS + u;
                    if (i + u * MLS_THREADS < v_hi) {
                        vm = s_vmask[(j >> 5) * MLS_THREADS + tid];
                        if ((vm >> (j & 31)) & 1u) pen_put_aside(v[u]);
                    }
                }
            }
        }
    }
    // Elements not covered by 16-byte vectors (all of them when !VEC); the last part of a split row.
    // Rows that are not streamed (skip_stream) are only visited here for the -inf fill of an
    // element-wise dense output.
    if ((!SPLIT || part == a.S - 1) && !(a.skip_stream && (VEC || OUT == OUT_COMPACT || a.prefilled))) {
        const float ninf_f = -INFINITY;
        for (int e = (VEC ? nvec * EPV : 0) + tid; e < a.V; e += MLS_THREADS) {
            if (!a.skip_stream) {
                float f[1] = {E::to_float(row[e])};
                accumulate<1>(acc, f);
            }
            if (VEC || a.prefilled) continue;  // filled by fill_rows_kernel / still filled from the call before
            if (OUT == OUT_FULL_F32) {
                reinterpret_cast<float*>(a.out)[(int64_t)r * a.out_stride + e] = ninf_f;
            } else if (OUT == OUT_FULL_SAME) {
                reinterpret_cast<T*>(a.out)[(int64_t)r * a.out_stride + e] = E::from_float(ninf_f);
            }
        }
    }

    typedef typename std::conditional<OUT == OUT_FULL_F32, float, T>::type OT;
    // An output row kept from the call before: its finite positions -- the tokens of the node recorded
    // in last[r] -- go back to -inf, which restores the all -inf row a fill pass would have left.
    auto reset_last = [&]() {
        if constexpr (OUT != OUT_COMPACT) {
            const grx_row_plan was = a.last[r];
            OT* __restrict__ orow = reinterpret_cast<OT*>(a.out) + (int64_t)r * a.out_stride;
            for (int i = tid; i < was.n; i += MLS_THREADS) {
                const int tokv = was.code == GRX_RES_EOS ? a.eos : a.tok[was.off + i];
                orow[tokv] = to_out<DT, OUT>(-INFINITY);
            }
        }
    };
    if (OUT != OUT_COMPACT && a.prefilled && !SPLIT) reset_last();
    if (OUT != OUT_COMPACT && (!VEC || (a.prefilled && !SPLIT))) {
        // The -inf stores of every thread (fill or reset) must have reached memory before any
        // thread -- of any wave -- overwrites an allowed position: each thread waits for its own
        // stores, then the workgroup meets.  (The barrier inside block_combine cannot stand in
        // for this one: rows that are not streamed never get there.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    MSTAMP(3);
    MaxSum tot;
    PenSums pen_tot = {0.f, 0.f};
    if constexpr (pen_stream) {
        // What this thread put aside is taken apart now, ahead of the workgroup's combine: every marked
        // element x_u passed through this thread, so it is no larger than the thread's own maximum and
        // 2^(x_u log2e + c) with the thread's own offset c cannot overflow.  The two sums the second
        // normaliser needs -- sum 2^(x_u log2e + c) and sum 2^(p (x_u log2e + c)) -- then travel through
        // the same reductions, and the same single barrier, as the thread's sum of the whole row.
        auto own_value = [&](float x) {
            if (x == -INFINITY) return;
            const float e = fmaf(x, kLog2e, acc.c);
            pen_tot.sub += fast_exp2(e);
            pen_tot.add += fast_exp2(a.penalty * e);
        };
        int k = 0;  // in stream order = the order of the bits of the thread's mask words
#ifdef PEN_ABL_NODECODE
        for (int w = 0; w < 0; ++w) {
#else
        for (int w = 0; w * MLS_THREADS * 4 < a.pen_vm_bytes; ++w) {
#endif
            uint32_t word = s_vmask[w * MLS_THREADS + tid];
            while (word) {
                const int iv = v_lo + tid + (w * 32 + __ffs((int)word) - 1) * MLS_THREADS;
                word &= word - 1;
                u32x4 vec;
                if (k < PEN_STASH) vec = s_stash[k * MLS_THREADS + tid];
                else if (k == PEN_STASH && pool_slot0 >= 0) vec = s_pool[pool_slot0];
                else if (k == PEN_STASH + 1 && pool_slot1 >= 0) vec = s_pool[pool_slot1];
                else vec = vp.load(iv);  // no slot was left: the one case in which a logit is read twice
                ++k;
                uint32_t b = pen_bits(iv);
                while (b) {
                    own_value(E::element(vec, __ffs((int)b) - 1));
                    b &= b - 1;
                }
            }
        }
        // marked elements behind the row's last whole vector (streamed by this same thread above)
        if (!SPLIT || part == a.S - 1)
            for (int e = nvec * EPV + tid; e < a.V; e += MLS_THREADS)
                if ((s_bm[e >> 5] >> (e & 31)) & 1u) own_value(E::to_float(row[e]));
        tot = block_combine_pen(acc, a.penalty, pen_tot, s_m, s_s, s_pen_red);  // contains a workgroup barrier
    } else if (a.skip_stream) {  // the caller's rows are normalised: max-shifted sum 1 at maximum 0, log-sum-exp 0
        tot.m = 0.f;
        tot.s = 1.f;
        tot.c = 0.f;
    } else {
        tot = block_combine(acc, s_m, s_s);  // contains a workgroup barrier
    }
    MSTAMP(4);
    if (SPLIT) {
        // Publish this part; the workgroup that observes the S-th arrival owns the row.
        // Partials and counters only ever move through device-scope atomics (which
        // bypass the per-XCD L2), so no cache write-back or invalidate is needed.
        __shared__ int s_last;
        // two words per part: (max, sum) and, with the repetition penalty, the part's two penalty sums relative
        // to its own maximum -- sum exp(x_u - m) and sum exp(p (x_u - m)) over the marked tokens it streamed
        unsigned long long* parts = reinterpret_cast<unsigned long long*>(a.split_part) + (int64_t)r * a.S * 2;
        if (tid == 0) {
            const unsigned long long packed =
                ((unsigned long long)__float_as_uint(tot.s) << 32) | (unsigned long long)__float_as_uint(tot.m);
            __hip_atomic_store(parts + part * 2, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (pen_stream) {
                const float add_m = pen_tot.add * fast_exp2(-a.penalty * fmaf(tot.m, kLog2e, tot.c));
                const unsigned long long pen_packed =
                    ((unsigned long long)__float_as_uint(add_m) << 32) | (unsigned long long)__float_as_uint(pen_tot.sub);
                __hip_atomic_store(parts + part * 2 + 1, pen_packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial has landed before the arrival is counted
            const int seen = __hip_atomic_fetch_add(&a.split_count[r], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = seen == a.S - 1;
        }
        __syncthreads();
        if (!s_last) return;
        float M = -INFINITY, sum = 0.f;
        {
            // lane p of every wave fetches partial p (S <= 32); merged in index order, so the
            // result does not depend on which part came last
            const int lane = tid & (GRX_WAVE - 1);
            unsigned long long mine = 0, mine_pen = 0;
            if (lane < a.S) {
                mine = __hip_atomic_load(parts + lane * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if constexpr (pen_stream)
                    mine_pen = __hip_atomic_load(parts + lane * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int p = 0; p < a.S; ++p) {
                const unsigned long long v = __shfl(mine, p, GRX_WAVE);
                M = fmaxf(M, __uint_as_float((uint32_t)(v & 0xFFFFFFFFull)));
            }
            const float Mref = (M == -INFINITY) ? 0.f : M;
            float sub = 0.f, add = 0.f;
            for (int p = 0; p < a.S; ++p) {
                const unsigned long long v = __shfl(mine, p, GRX_WAVE);
                const float g = (__uint_as_float((uint32_t)(v & 0xFFFFFFFFull)) - Mref) * kLog2e;  // <= 0, or -inf for an empty part
                sum += __uint_as_float((uint32_t)(v >> 32)) * fast_exp2(g);
                if constexpr (pen_stream) {
                    const unsigned long long w = __shfl(mine_pen, p, GRX_WAVE);
                    const float sub_p = __uint_as_float((uint32_t)(w & 0xFFFFFFFFull)), add_p = __uint_as_float((uint32_t)(w >> 32));
                    if (sub_p > 0.f) {
                        sub += sub_p * fast_exp2(g);
                        add += add_p * fast_exp2(a.penalty * g);
                    }
                }
            }
            if constexpr (pen_stream) {  // from here on relative to the row's maximum exactly
                pen_tot.sub = sub;
                pen_tot.add = add;
            }
        }
        tot.m = M;
        tot.s = sum;
        tot.c = (M == -INFINITY) ? 0.f : -M * kLog2e;
        if (tid == 0)  // ready for the next call
            __hip_atomic_store(&a.split_count[r], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (OUT != OUT_COMPACT && a.prefilled) {  // the emitting workgroup tidies the kept output row
            reset_last();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        gather();
    }
    const float log_s = logf(tot.s);
    MSTAMP(5);

    // Optional repetition penalty, applied where transformers applies it: on
    // y = log_softmax(row), y_v <- y_v * p (y_v < 0) or y_v / p for every token v that
    // occurs in the row's ids; the result is normalised again.  Only the penalised
    // tokens move, so the second log-sum-exp follows from the first:
    //   sum_v exp(y'_v) = (S - sum_U exp(x_v - m)) / S + sum_U exp(y'_v),  U = distinct ids.
    // Both terms are O(1) or smaller and the total is >= ~0.5, so the subtraction
    // costs no accuracy that matters at fp32.
    float lse2 = 0.f;
    auto penalised = [&](float y) { return y < 0.f ? y * a.penalty : y / a.penalty; };
    auto in_pen_set = [&](int tokv) { return ((s_bm[tokv >> 5] >> (tokv & 31)) & 1u) != 0; };
    if (pen_on) {
        float sub, add;
        if constexpr (pen_stream) {
            // pen_tot.add is relative to the offset the workgroup summed at: 2^(p (x_u log2e + c)) with
            // m log2e + c = residual, so exp(p (x_u - m - log s)) takes a factor 2^(-p (residual + log2 s))
            // (a split row's merged sums are relative to the row's maximum already: no residual)
            const float residual = SPLIT ? 0.f : fmaf(tot.m, kLog2e, tot.c);
            sub = pen_tot.sub;
            add = pen_tot.add * fast_exp2(-a.penalty * (residual + log_s * kLog2e));
        } else {
            pen_load_ids();  // rows without 16-byte alignment: the marked tokens' logits are fetched now
            pen_mark(true, tot.m, log_s);
            sub = wave_sum(pen_sub);
            add = wave_sum(pen_add);
            if ((tid & (GRX_WAVE - 1)) == 0) {
                s_pen_red[0][tid / GRX_WAVE] = sub;
                s_pen_red[1][tid / GRX_WAVE] = add;
            }
            __syncthreads();
            sub = 0.f;
            add = 0.f;
#pragma unroll
            for (int w = 0; w < MLS_WAVES; ++w) {
                sub += s_pen_red[0][w];
                add += s_pen_red[1][w];
            }
        }
        lse2 = logf(fmaxf(tot.s - sub, 0.f) / tot.s + add);
    }
    // log-prob of allowed token `tokv` whose logit minus the row maximum is `xm`
    auto final_logp = [&](int tokv, float xm) {
        float y = xm - log_s;
        if (pen_on) {
            if (in_pen_set(tokv)) y = penalised(y);
            y -= lse2;
        }
        return y;
    };

    // Emit the allowed tokens.
    if (OUT == OUT_COMPACT) {
        // Whole 128-byte lines only: header + candidates + zero padding up to the next 16 entries.
        grx_cand* __restrict__ crow = a.cand + (int64_t)r * a.cand_stride;
        const int n_line = ((c_n + 1 + 15) / 16) * 16;
        static_assert(MLS_EARLY == 2, "the compact epilogue pairs two 8-byte entries per 16-byte store");
        if (trunc && c_n > 0) {
            // More candidates than the beam step can ever keep from one row: hand over only the keep_top
            // best by (logp + row bias) descending, token ascending -- the order the beam step ranks by.
            // Every thread finishes the log-probs of its candidates (candidate i lives with thread
            // (i + 1) / 2, or from i = 511 on with thread (i - 511) mod 256) and leaves them, with their
            // sort keys, in the LDS block of the penalty set, which nobody reads once the log-probs are
            // final.  Wave 0 alone then selects: the keep_top-th largest key is built bit by bit from
            // ballot counts (keys in registers, 32 rounds, no barriers), keys above it are kept, keys equal
            // to it in candidate order until keep_top entries are placed.  The kept entries come out in no
            // particular order: the beam step does not need one.
            constexpr int KPL = GRX_KEEP_TOP_MAX_FANOUT / GRX_WAVE;  // keys per lane of wave 0, at most
            // (penalty instantiation: the bitmap's LDS, at least GRX_KEEP_TOP_MAX_FANOUT * 8 bytes by launch)
            uint32_t* const s_key = PEN ? reinterpret_cast<uint32_t*>(s_dyn) : s_sel;
            float* const s_lp = reinterpret_cast<float*>(s_key + GRX_KEEP_TOP_MAX_FANOUT);
            const int keep = a.keep_top;
            const float bias = a.row_bias ? a.row_bias[r] : 0.f;
            float lp0 = 0.f, lp1 = 0.f;
            const int i0 = tid * 2 - 1, i1 = tid * 2;
            if (i0 >= 0 && i0 < c_n) lp0 = final_logp(e_tok[0], E::to_float(e_raw[0]) - tot.m);
            if (i1 < c_n) lp1 = final_logp(e_tok[1], E::to_float(e_raw[1]) - tot.m);
            constexpr int FIRST = 2 * MLS_THREADS - 1;  // candidates gathered up front
            constexpr int XTRA = (GRX_KEEP_TOP_MAX_FANOUT - FIRST + MLS_THREADS - 1) / MLS_THREADS;
            float lpx[XTRA];
            if (c_n > FIRST) {  // wider nodes: the rest is gathered now
#pragma unroll
                for (int k = 0; k < XTRA; ++k) {
                    const int i = FIRST + k * MLS_THREADS + tid;
                    lpx[k] = 0.f;
                    if (i < c_n) {
                        const int tokv = a.tok[c_off + i];
                        lpx[k] = final_logp(tokv, E::to_float(row[tokv]) - tot.m);
                    }
                }
            }
            __syncthreads();  // the penalty bitmap has been consulted for the last time
            if (i0 >= 0 && i0 < c_n) {
                s_lp[i0] = lp0;
                s_key[i0] = float_to_ordered(lp0 + bias);
            }
            if (i1 < c_n) {
                s_lp[i1] = lp1;
                s_key[i1] = float_to_ordered(lp1 + bias);
            }
            if (c_n > FIRST) {
#pragma unroll
                for (int k = 0; k < XTRA; ++k) {
                    const int i = FIRST + k * MLS_THREADS + tid;
                    if (i < c_n) {
                        s_lp[i] = lpx[k];
                        s_key[i] = float_to_ordered(lpx[k] + bias);
                    }
                }
            }
            __syncthreads();
            if (!FUSE && !MLS_ABL_TAILWAIT && tid >= GRX_WAVE) return;
            if (tid < GRX_WAVE) {
                const int lane = tid;
                const int q_n = (c_n + GRX_WAVE - 1) / GRX_WAVE;
                if (q_n <= 2) select_top<2, COHST>(s_key, s_lp, a.tok + c_off, c_n, keep, lane, crow);
                else if (q_n <= 4) select_top<4, COHST>(s_key, s_lp, a.tok + c_off, c_n, keep, lane, crow);
                else if (q_n <= 8) select_top<8, COHST>(s_key, s_lp, a.tok + c_off, c_n, keep, lane, crow);
                else select_top<KPL, COHST>(s_key, s_lp, a.tok + c_off, c_n, keep, lane, crow);
                const int n_kept_line = ((keep + 1 + 15) / 16) * 16;
                if (lane == 0) {
                    grx_cand h;
                    h.x = keep;
                    h.y = __float_as_int(tot.m + log_s + lse2);
                    cand_store<COHST>(crow, h);
                }
                const grx_cand zero = {0, 0};
                for (int e = keep + 1 + lane; e < n_kept_line; e += GRX_WAVE) cand_store<COHST>(crow + e, zero);
            }
            if (!FUSE && !MLS_ABL_TAILWAIT) return;
        } else {
        auto entry = [&](int e, int tokv, float xm) {  // compact entry e of this row
            grx_cand v = {0, 0};
            if (e == 0) {
                v.x = c_n;
                v.y = __float_as_int(tot.m + log_s + lse2);
            } else if (e - 1 < c_n) {
                v.x = tokv;
                v.y = __float_as_int(final_logp(tokv, xm));
            }
            return v;
        };
        {
            const int e = tid * 2;
#ifdef MLS_ABL_NOSTORE
            if (e == 0) {
#else
            if (e < n_line) {  // n_line is even, so the pair is inside the row
#endif
                cand_store2<COHST>(crow + e, entry(e, e_tok[0], E::to_float(e_raw[0]) - tot.m), entry(e + 1, e_tok[1], E::to_float(e_raw[1]) - tot.m));
            }
        }
#ifdef MLS_ABL_NOSTORE
        for (int e = (tid + MLS_THREADS) * 2; e < 0; e += MLS_THREADS * 2) {
#else
        for (int e = (tid + MLS_THREADS) * 2; e < n_line; e += MLS_THREADS * 2) {
#endif
            // an EOS-only row has c_n == 1 and never gets here
            const int t0 = e - 1 < c_n ? a.tok[c_off + e - 1] : 0, t1 = e < c_n ? a.tok[c_off + e] : 0;
            const float x0 = e - 1 < c_n ? E::to_float(row[t0]) - tot.m : 0.f;
            const float x1 = e < c_n ? E::to_float(row[t1]) - tot.m : 0.f;
            cand_store2<COHST>(crow + e, entry(e, t0, x0), entry(e + 1, t1, x1));
        }
        }  // rows emitted whole
#if MLS_ABL_TAILWAIT
        if constexpr (!FUSE) {
            __shared__ int s_dummy;
            if (MLS_ABL_TAILWAIT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                int* const arrivals = &const_cast<grx_row_plan*>(a.plan)[(int64_t)(r / 16) * 16].pad;
                s_dummy = __hip_atomic_fetch_add(arrivals, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (s_dummy == 12345) return;
        }
#endif
        if constexpr (FUSE) {
            // This row is done.  Its entries were written with device-scope stores; once every wave has
            // seen its own stores complete, the workgroup counts itself into its prompt group -- the
            // counter is the spare word of the plan entry of the group's first row, zero between calls --
            // and the workgroup that completes the group runs its beam step right here, reading the K
            // candidate rows with device-scope loads, while the rows of other groups still stream.
            __shared__ int s_completes_group;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int K = fb->a.K;
            const int g = r / K;
            if (tid == 0) {
                int* const arrivals = &const_cast<grx_row_plan*>(a.plan)[(int64_t)g * K].pad;
                const int seen = __hip_atomic_fetch_add(arrivals, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen == K - 1) __hip_atomic_store(arrivals, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_completes_group = seen == K - 1;
            }
            __syncthreads();
            if (!s_completes_group) return;
#ifdef FUSE_ABL_NOTAIL
            if (fb->a.V != -12345) return;  // diagnostic: the step is left to the beam kernel behind this one
#endif
            beam_step_group<BS_SMALL_K, BS_SMALL_KEEP, BS_SMALL_SURV, BS_SMALL_K, true>(fb->a, fb->d, fb->pc, fb->plan_out, g);
        }
    } else {
        OT* __restrict__ orow = reinterpret_cast<OT*>(a.out) + (int64_t)r * a.out_stride;
        auto put = [&](int tok, float lp) {
            orow[tok] = to_out<DT, OUT>(lp);
        };
#pragma unroll
        for (int k = 0; k < MLS_EARLY; ++k) {
            const int i = tid + k * MLS_THREADS;
            if (i < c_n) put(e_tok[k], final_logp(e_tok[k], E::to_float(e_raw[k]) - tot.m));
        }
        for (int i = tid + MLS_EARLY * MLS_THREADS; i < c_n; i += MLS_THREADS) {
            const int tok = a.tok[c_off + i];  // an EOS-only row has c_n == 1 and never gets here
            put(tok, final_logp(tok, E::to_float(row[tok]) - tot.m));
        }
        if (a.last != nullptr && tid == 0) {  // what this row of `out` now holds, for the next call on it
            grx_row_plan now;
            now.code = code;
            now.off = c_off;
            now.n = (code >= 0 || code == GRX_RES_EOS) ? c_n : 0;
            now.pad = 0;
            a.last[r] = now;
        }
    }
#ifdef MLS_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MSTAMP(6);
#endif
}

template <int DT, int OUT, bool VEC, bool SPLIT, bool PEN>
__global__ __launch_bounds__(MLS_THREADS) void masked_lsm_kernel(MlsArgs a) {
    mls_body<DT, OUT, VEC, SPLIT, PEN, false>(a, nullptr);
}

// The step in one launch: compact candidates, and the beam step of every prompt group as soon as the
// group's rows are complete.  16-byte aligned rows without repetition penalty only.
template <int DT, bool SPLIT>
__global__ __launch_bounds__(MLS_THREADS) void masked_lsm_beam_kernel(MlsArgs a, FusedBeam fb) {
    mls_body<DT, OUT_COMPACT, true, SPLIT, false, true>(a, &fb);
}

// ---- dense output fill ---------------------------------------------------------
// Writes -inf over V elements of every output row: 16-byte stores plus a scalar
// tail.  One short-lived workgroup per FILL_CHUNK vectors of a row.
#define FILL_THREADS 256
#ifndef FILL_PER_THREAD
#define FILL_PER_THREAD 1   // measured: one 16-byte store per thread over a huge grid writes fastest (6.9 TB/s)
#endif
#define FILL_CHUNK (FILL_THREADS * FILL_PER_THREAD)

__global__ __launch_bounds__(FILL_THREADS) void fill_rows_kernel(void* out, int64_t row_bytes, int row_vecs,
                                                                 int64_t V, int elsize, uint32_t packed_ninf,
                                                                 int chunks) {
    const int r = blockIdx.x / chunks;
    const int chunk = blockIdx.x - r * chunks;
    char* __restrict__ rowp = reinterpret_cast<char*>(out) + (int64_t)r * row_bytes;
    u32x4* __restrict__ op = reinterpret_cast<u32x4*>(rowp);
    const u32x4 fill = {packed_ninf, packed_ninf, packed_ninf, packed_ninf};
    const int64_t base = (int64_t)chunk * FILL_CHUNK + threadIdx.x;
#pragma unroll
    for (int u = 0; u < FILL_PER_THREAD; ++u) {
        const int64_t j = base + u * FILL_THREADS;
        if (j < row_vecs) fill_store(op + j, fill);
    }
    if (chunk == chunks - 1) {  // elements behind the last whole vector
        const int64_t done = (int64_t)row_vecs * (16 / elsize);
        for (int64_t e = done + threadIdx.x; e < V; e += FILL_THREADS) {
            if (elsize == 4) reinterpret_cast<uint32_t*>(rowp)[e] = packed_ninf;
            else reinterpret_cast<uint16_t*>(rowp)[e] = (uint16_t)(packed_ninf & 0xFFFFu);
        }
    }
}

static uint32_t packed_neg_inf(int dtype) {
    return dtype == GRX_F32 ? 0xFF800000u : (dtype == GRX_BF16 ? 0xFF80FF80u : 0xFC00FC00u);
}

// ---- optional kernel timing ---------------------------------------------------
#define TIMING_SLOTS 4096
static int g_timing_every = 0;  // 0: off; k: every k-th call carries events
static int g_timing_calls = 0;  // calls seen since the enable
static int g_timing_n = 0;
static hipEvent_t g_ev_start[TIMING_SLOTS], g_ev_stop[TIMING_SLOTS];
static int g_ev_created = 0;

extern "C" int grx_timing_enable(int every) {
    g_timing_every = every > 0 ? every : 0;
    g_timing_calls = 0;
    g_timing_n = 0;
    return GRX_OK;
}

extern "C" int grx_timing_read(double* total_ms, int32_t* launches) {
    GRX_REQUIRE(total_ms && launches, GRX_E_INVALID, "grx_timing_read: null pointer");
    double sum = 0.0;
    for (int i = 0; i < g_timing_n; ++i) {
        GRX_CHECK_HIP(hipEventSynchronize(g_ev_stop[i]));
        float ms = 0.f;
        GRX_CHECK_HIP(hipEventElapsedTime(&ms, g_ev_start[i], g_ev_stop[i]));
        sum += (double)ms;
    }
    *total_ms = sum;
    *launches = g_timing_n;
    return GRX_OK;
}

static int timing_slot() {
    if (g_timing_every == 0) return -1;
    // the every-th, 2 x every-th, ... call: not the first one, whose launch the host has not run ahead of yet
    if ((g_timing_calls++ % g_timing_every) != g_timing_every - 1 || g_timing_n >= TIMING_SLOTS) return -1;
    while (g_ev_created <= g_timing_n) {
        if (hipEventCreate(&g_ev_start[g_ev_created]) != hipSuccess) return -1;
        if (hipEventCreate(&g_ev_stop[g_ev_created]) != hipSuccess) return -1;
        ++g_ev_created;
    }
    return g_timing_n;
}

// ---- read probe ---------------------------------------------------------------
// The streaming kernel's access pattern with everything else taken away: one workgroup per row,
// MLS_UNROLL 16-byte non-temporal loads in flight per thread, the loaded words folded into one
// checksum word per wave.  What this reaches is the read throughput the part gives this pattern.
__global__ __launch_bounds__(MLS_THREADS) void read_probe_kernel(const char* base, int64_t stride_bytes, int nvec,
                                                                 uint32_t* __restrict__ out) {
    const int tid = threadIdx.x;
    const RowStream vp(base + (int64_t)blockIdx.x * stride_bytes, (int64_t)nvec * 16);
    uint32_t acc = 0;
    int i = tid;
    for (; i + (MLS_UNROLL - 1) * MLS_THREADS < nvec; i += MLS_UNROLL * MLS_THREADS) {
        u32x4 v[MLS_UNROLL];
#pragma unroll
        for (int u = 0; u < MLS_UNROLL; ++u) v[u] = vp.load(i + u * MLS_THREADS);
#pragma unroll
        for (int u = 0; u < MLS_UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < nvec; i += MLS_THREADS) {
        const u32x4 v = vp.load(i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc ^= __shfl_xor(acc, o, GRX_WAVE);
    if ((tid & (GRX_WAVE - 1)) == 0) out[(int64_t)blockIdx.x * MLS_WAVES + tid / GRX_WAVE] = acc;
}

extern "C" int grx_read_probe(const void* rows, int64_t stride_bytes, int32_t R, int64_t row_bytes, uint32_t* out,
                              void* stream) {
    const char* who = "grx_read_probe";
    GRX_REQUIRE(R >= 0 && row_bytes >= 0 && stride_bytes >= row_bytes, GRX_E_INVALID, "%s: bad shape", who);
    if (R == 0) return GRX_OK;
    GRX_REQUIRE(rows != nullptr && out != nullptr, GRX_E_INVALID, "%s: null pointer", who);
    GRX_REQUIRE(((uintptr_t)rows & 15u) == 0 && (stride_bytes & 15) == 0, GRX_E_INVALID,
                "%s: rows must be 16-byte aligned", who);
    GRX_REQUIRE(row_bytes / 16 <= 0x7FFFFFFF, GRX_E_UNSUPPORTED, "%s: row too long", who);
    const int slot = timing_slot();
    hipEvent_t ev0 = slot >= 0 ? g_ev_start[slot] : nullptr, ev1 = slot >= 0 ? g_ev_stop[slot] : nullptr;
    if (ev0)
        hipExtLaunchKernelGGL(read_probe_kernel, dim3((unsigned)R), dim3(MLS_THREADS), 0, (hipStream_t)stream, ev0, ev1, 0,
                              (const char*)rows, stride_bytes, (int)(row_bytes / 16), out);
    else
        hipLaunchKernelGGL(read_probe_kernel, dim3((unsigned)R), dim3(MLS_THREADS), 0, (hipStream_t)stream,
                           (const char*)rows, stride_bytes, (int)(row_bytes / 16), out);
    GRX_CHECK_HIP(hipGetLastError());
    if (slot >= 0) g_timing_n = slot + 1;
    return GRX_OK;
}

// ---- launch -------------------------------------------------------------------
// Dynamic LDS of the penalty instantiation: token bitmap, per-thread vector masks, PEN_STASH 16-byte
// slots per thread; never less than the keep_top selection needs when it takes the block over.
#define PEN_MAX_V GRX_MAX_PENALTY_VOCAB  // the bitmap and vector masks of a wider vocabulary would not fit a workgroup's LDS
static unsigned pen_lds_bytes(int bm_bytes, int vm_bytes) {
    const unsigned need = (unsigned)bm_bytes + (unsigned)vm_bytes + (PEN_STASH * MLS_THREADS + PEN_POOL) * 16u;
    return need < GRX_KEEP_TOP_MAX_FANOUT * 8u ? GRX_KEEP_TOP_MAX_FANOUT * 8u : need;
}

// With events the dispatch itself carries them (start/stop timestamps of this
// kernel only, no extra packets on the stream).
template <typename KernelT>
static void launch_one(KernelT kernel, const MlsArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const dim3 grid((unsigned)a.R * (unsigned)(a.S > 1 ? a.S : 1));
    const unsigned lds = a.penalty > 0.f ? pen_lds_bytes(a.pen_bm_bytes, a.pen_vm_bytes) : 0u;
    if (ev0 || ev1)
        hipExtLaunchKernelGGL(kernel, grid, dim3(MLS_THREADS), lds, stream, ev0, ev1, 0, a);
    else
        hipLaunchKernelGGL(kernel, grid, dim3(MLS_THREADS), lds, stream, a);
}

template <int DT, int OUT, bool VEC, bool SPLIT>
static void launch_pen(const MlsArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (a.penalty > 0.f) launch_one(masked_lsm_kernel<DT, OUT, VEC, SPLIT, true>, a, stream, ev0, ev1);
    else launch_one(masked_lsm_kernel<DT, OUT, VEC, SPLIT, false>, a, stream, ev0, ev1);
}

template <int DT, int OUT, bool VEC>
static void launch_split(const MlsArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if constexpr (VEC) {  // only 16-byte streams are split
        if (a.S > 1) {
            launch_pen<DT, OUT, VEC, true>(a, stream, ev0, ev1);
            return;
        }
    }
    launch_pen<DT, OUT, VEC, false>(a, stream, ev0, ev1);
}

template <int DT, int OUT>
static void launch_vec(bool vec, const MlsArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (vec) launch_split<DT, OUT, true>(a, stream, ev0, ev1);
    else launch_split<DT, OUT, false>(a, stream, ev0, ev1);
}

template <int OUT>
static void launch_dtype(int dtype, bool vec, const MlsArgs& a, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (dtype == GRX_F32) launch_vec<GRX_F32, OUT>(vec, a, stream, ev0, ev1);
    else if (dtype == GRX_BF16) launch_vec<GRX_BF16, OUT>(vec, a, stream, ev0, ev1);
    else launch_vec<GRX_F16, OUT>(vec, a, stream, ev0, ev1);
}

static size_t dtype_size(int dt) { return dt == GRX_F32 ? 4 : 2; }

template <int DT>
static void launch_fused_split(const MlsArgs& a, const FusedBeam& fb, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    const dim3 grid((unsigned)a.R * (unsigned)(a.S > 1 ? a.S : 1));
    if (a.S > 1) {
        if (ev0 || ev1)
            hipExtLaunchKernelGGL((masked_lsm_beam_kernel<DT, true>), grid, dim3(MLS_THREADS), 0, stream, ev0, ev1, 0, a, fb);
        else
            hipLaunchKernelGGL((masked_lsm_beam_kernel<DT, true>), grid, dim3(MLS_THREADS), 0, stream, a, fb);
    } else {
        if (ev0 || ev1)
            hipExtLaunchKernelGGL((masked_lsm_beam_kernel<DT, false>), grid, dim3(MLS_THREADS), 0, stream, ev0, ev1, 0, a, fb);
        else
            hipLaunchKernelGGL((masked_lsm_beam_kernel<DT, false>), grid, dim3(MLS_THREADS), 0, stream, a, fb);
    }
}

static void launch_fused(int dtype, const MlsArgs& a, const FusedBeam& fb, hipStream_t stream, hipEvent_t ev0,
                         hipEvent_t ev1) {
    if (dtype == GRX_F32) launch_fused_split<GRX_F32>(a, fb, stream, ev0, ev1);
    else if (dtype == GRX_BF16) launch_fused_split<GRX_BF16>(a, fb, stream, ev0, ev1);
    else launch_fused_split<GRX_F16>(a, fb, stream, ev0, ev1);
}

static bool rows_16b_aligned(const void* base, int64_t stride, size_t elsize) {
    return (((uintptr_t)base) & 15u) == 0 && ((stride * (int64_t)elsize) & 15) == 0;
}

// Largest launch whose beam steps run inside the streaming kernel.  A group's step holds a workgroup slot
// for about 20 us, and the steps of the groups that finish last run after the stream has ended: with
// thousands of rows that costs more than the separate beam kernel it replaces (8192 rows: 418 against
// 385 us per step; 1024 rows: even), with few rows -- launch-bound steps -- it wins (64 rows: 45 against
// 52 us).  GRX_FUSE_MAX_ROWS in the environment overrides the limit (tuning; read once).
static int fuse_max_rows() {
    static const int limit = [] {
        const char* e = getenv("GRX_FUSE_MAX_ROWS");
        return e != nullptr && *e ? atoi(e) : 512;
    }();
    return limit;
}

// `beam` (compact form only): the beam step that consumes the candidates, to be carried out by this same
// call -- inside the streaming kernel where that is possible, by the beam kernel right behind it otherwise.
static int run_step(const char* who, const grx_trie* trie, grx_state* state, const grx_rows* rows,
                    const grx_constraint* con, MlsArgs& a, int out_kind, bool out_aligned, void* workspace,
                    size_t workspace_bytes, hipStream_t stream, const grx_beam_step_args* beam = nullptr) {
    int rc = grx_check_step_args(who, trie, state, rows, con, workspace, workspace_bytes);
    if (rc != GRX_OK) return rc;
    if (beam != nullptr) {
        rc = grx_beam_step_check(who, beam);
        if (rc != GRX_OK) return rc;
        GRX_REQUIRE(beam->cand == a.cand && beam->cand_stride == a.cand_stride && beam->cmax == a.cmax, GRX_E_INVALID,
                    "%s: the beam step must read the candidate buffer this call writes", who);
        GRX_REQUIRE((int64_t)beam->B * (beam->cand_shared ? 1 : beam->K) == rows->R, GRX_E_INVALID,
                    "%s: %d rows do not make %d groups of %d beams", who, rows->R, beam->B, beam->K);
    }
    GRX_REQUIRE(con->eos_token_id < rows->V, GRX_E_INVALID, "%s: eos_token_id %d outside V=%d", who,
                con->eos_token_id, rows->V);
    const bool penalise = con->repetition_penalty != 0.f && con->repetition_penalty != 1.f;
    GRX_REQUIRE(con->assume_normalized == 0 || con->assume_normalized == 1, GRX_E_INVALID,
                "%s: assume_normalized must be 0 or 1", who);
    GRX_REQUIRE(!(penalise && con->assume_normalized), GRX_E_UNSUPPORTED,
                "%s: the fused repetition penalty needs the rows streamed (assume_normalized must be 0)", who);
    if (penalise) {
        GRX_REQUIRE(con->repetition_penalty > 0.f, GRX_E_INVALID, "%s: repetition_penalty must be positive", who);
        GRX_REQUIRE(con->penalty_skip >= 0, GRX_E_INVALID, "%s: penalty_skip must be >= 0", who);
        GRX_REQUIRE(con->shared_prefix_len >= 0 && con->shared_prefix_len <= rows->T, GRX_E_INVALID,
                    "%s: shared_prefix_len must lie in [0, T]", who);
        GRX_REQUIRE(con->shared_prefix_len == 0 || (con->shared_group >= 1 && rows->R % con->shared_group == 0), GRX_E_INVALID,
                    "%s: with shared_prefix_len, R must be a multiple of shared_group >= 1", who);
        GRX_REQUIRE(rows->T - con->penalty_skip <= GRX_MAX_PENALTY_TOKENS, GRX_E_UNSUPPORTED,
                    "%s: repetition penalty over more than %d tokens per row", who, GRX_MAX_PENALTY_TOKENS);
        GRX_REQUIRE(rows->V <= PEN_MAX_V, GRX_E_UNSUPPORTED,
                    "%s: the fused repetition penalty serves vocabularies of up to %d tokens (this one: %d)", who,
                    PEN_MAX_V, rows->V);
    }
    GRX_REQUIRE(con->keep_top >= 0 && con->keep_top <= BS_KEEP_CAP, GRX_E_INVALID, "%s: keep_top must be in [0, %d]", who,
                BS_KEEP_CAP);
    GRX_REQUIRE(con->keep_top == 0 || out_kind != OUT_COMPACT || trie->max_fanout <= GRX_KEEP_TOP_MAX_FANOUT,
                GRX_E_UNSUPPORTED, "%s: keep_top serves tries of fan-out <= %d (this one: %d)", who,
                GRX_KEEP_TOP_MAX_FANOUT, trie->max_fanout);
    // Nothing below this line fails on an argument: the call now counts and its kernels are enqueued.
    // phase 0: one small kernel works out the trie node of every row
    grx_row_plan* plan = rows->R > 0
                             ? reinterpret_cast<grx_row_plan*>(reinterpret_cast<char*>(workspace) + GRX_WS_PLAN_OFFSET)
                             : nullptr;
    rc = grx_plan_rows(trie, state, rows, con, plan, stream, &a.pc, &a.bookkeep);
    if (rc != GRX_OK) return rc;
    if (rows->R == 0) return GRX_OK;  // no rows, no groups: the beam step has nothing to do either
    a.walk = con->mode == GRX_MODE_COUNT ? 1 : 0;
    a.ids = rows->input_ids;
    a.ids_stride = rows->ids_stride;
    a.T = rows->T;

    a.tok = trie->child_tok;
    a.off = trie->child_off;
    a.plan = plan;
    a.st = state ? state->dev : nullptr;
    a.penalty = 0.f;
    if (penalise) {
        a.penalty = con->repetition_penalty;
        a.pen_from = con->penalty_skip;
        static const bool ignore_shared = [] { const char* e = getenv("GRX_IGNORE_SHARED_PREFIX"); return e && *e == '1'; }();  // tuning
        a.pen_shared = ignore_shared ? 0 : con->shared_prefix_len;
        a.pen_group = a.pen_shared > 0 ? con->shared_group : 1;
        a.pen_bm_bytes = ((rows->V + 127) / 128) * 16;
        {   // a thread owns every MLS_THREADS-th vector of the row: one mask word per 32 of them
            const int nvec = rows->V / (16 / (int)dtype_size(rows->dtype));
            const int per_thread = (nvec + MLS_THREADS - 1) / MLS_THREADS;
            a.pen_vm_bytes = ((per_thread + 31) / 32) * MLS_THREADS * 4;
        }
    }
    a.R = rows->R;
    a.scores = rows->scores;
    a.scores_stride = rows->scores_stride;
    a.V = rows->V;
    a.eos = con->eos_token_id;
    const bool vec = rows_16b_aligned(rows->scores, rows->scores_stride, dtype_size(rows->dtype)) && out_aligned;
    a.S = 1;
    a.keep_top = out_kind == OUT_COMPACT ? con->keep_top : 0;
    a.row_bias = a.keep_top > 0 ? con->row_bias : nullptr;
    a.skip_stream = con->assume_normalized;
    if (vec && !a.skip_stream) {
        const int S = grx_split_factor(rows->R, (int)(rows->V / (16 / dtype_size(rows->dtype))));
        if (S > 1) {
            a.S = S;
            a.split_count = reinterpret_cast<int*>(workspace);
            a.split_part = reinterpret_cast<float2*>(reinterpret_cast<char*>(workspace) + GRX_SPLIT_COUNT_BYTES);
        }
    }
    // launches of several rounds of workgroups open every row with the deep prefetch; a launch that fits the GPU
    // in one round (few rows, split over several workgroups each) holds a few batches per thread in all: three
    // of them up front save a round trip (64 rows 25.2 -> 24.8 us per step, 400 rows with the penalty 48.9 -> 47.5),
    // all four when the launch has at most one workgroup per CU (64 rows 23.7 -> 23.4 us; 256 and 400 rows lose
    // 0.6 and 1.0 us with four)
    static const int small_head = [] {
        const char* e = getenv("GRX_HEAD_BATCHES_SMALL");
        const int v = e ? atoi(e) : 0;
        return v < 1 ? 0 : (v > MLS_HEAD_BATCHES ? MLS_HEAD_BATCHES : v);
    }();
    const int64_t n_wg = (int64_t)a.R * a.S;
    a.head_batches = n_wg >= 2048 ? MLS_HEAD_BATCHES : (small_head > 0 ? small_head : (n_wg <= 256 ? MLS_HEAD_BATCHES : 3));
    const int slot = timing_slot();
    hipEvent_t ev0 = slot >= 0 ? g_ev_start[slot] : nullptr, ev1 = slot >= 0 ? g_ev_stop[slot] : nullptr;
    if (out_kind != OUT_COMPACT && vec && !a.prefilled) {
        // Dense output, phase 1: -inf everywhere (write-only); the stream kernel then only reads
        // and scatters the allowed positions.  A timed call spans both kernels.
        const int odt = out_kind == OUT_FULL_F32 ? GRX_F32 : rows->dtype;
        const int osz = (int)dtype_size(odt);
        // contiguous rows are filled as one flat span (a single "row" of R * V elements)
        const bool flat = a.out_stride == rows->V && (int64_t)rows->R * rows->V * osz / 16 <= 0x7FFFFFFF;
        const int fill_rows = flat ? 1 : rows->R;
        const int64_t fill_elems = flat ? (int64_t)rows->R * rows->V : rows->V;
        const int row_vecs = (int)((fill_elems * osz) / 16);
        const int chunks = row_vecs > 0 ? (row_vecs + FILL_CHUNK - 1) / FILL_CHUNK : 1;
        GRX_REQUIRE((int64_t)chunks * fill_rows <= 0x7FFFFFFF, GRX_E_UNSUPPORTED, "%s: output too large", who);
        const dim3 grid((unsigned)((int64_t)chunks * fill_rows));
        if (ev0)
            hipExtLaunchKernelGGL(fill_rows_kernel, grid, dim3(FILL_THREADS), 0, stream, ev0, nullptr, 0, a.out,
                                  a.out_stride * osz, row_vecs, fill_elems, osz, packed_neg_inf(odt), chunks);
        else
            hipLaunchKernelGGL(fill_rows_kernel, grid, dim3(FILL_THREADS), 0, stream, a.out, a.out_stride * osz,
                               row_vecs, fill_elems, osz, packed_neg_inf(odt), chunks);
        GRX_CHECK_HIP(hipGetLastError());
        ev0 = nullptr;

    }
    if (beam != nullptr) {
        // One launch when the streaming kernel can carry the beam step: the 16-beam instantiation, one
        // candidate row per beam, 16-byte aligned rows that are streamed, no penalty, and a call whose
        // counter bookkeeping somebody else has done (a fused kernel must not read the state words of the
        // previous call, which the plan it makes for the next call re-arms).
        const bool fuse = vec && !a.skip_stream && a.penalty == 0.f && a.bookkeep == 0 && !beam->cand_shared &&
                          grx_beam_step_is_small(*beam) && rows->R <= fuse_max_rows();
        FusedBeam fb;
        fb.a = *beam;
        rc = grx_beam_step_derive(who, *beam, stream, &fb.d, &fb.pc, &fb.plan_out);
        if (rc != GRX_OK) return rc;
        if (fuse) {
            if (con->mode == GRX_MODE_COUNT) a.pc.count_ready = 1;
            launch_fused(rows->dtype, a, fb, stream, ev0, ev1);
            GRX_CHECK_HIP(hipGetLastError());
#ifdef FUSE_ABL_NOTAIL
            grx_beam_step_launch(*beam, fb.d, fb.pc, fb.plan_out, stream);
#endif
        } else {
            launch_dtype<OUT_COMPACT>(rows->dtype, vec, a, stream, ev0, ev1);
            GRX_CHECK_HIP(hipGetLastError());
            grx_beam_step_launch(*beam, fb.d, fb.pc, fb.plan_out, stream);
            GRX_CHECK_HIP(hipGetLastError());
        }
        if (slot >= 0) g_timing_n = slot + 1;
        return GRX_OK;
    }
    if (out_kind == OUT_COMPACT) launch_dtype<OUT_COMPACT>(rows->dtype, vec, a, stream, ev0, ev1);
    else if (out_kind == OUT_FULL_F32) launch_dtype<OUT_FULL_F32>(rows->dtype, vec, a, stream, ev0, ev1);
    else launch_dtype<OUT_FULL_SAME>(rows->dtype, vec, a, stream, ev0, ev1);
    GRX_CHECK_HIP(hipGetLastError());
    if (slot >= 0) g_timing_n = slot + 1;
    return GRX_OK;
}

extern "C" int grx_constrained_logsoftmax(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                          const grx_constraint* con, void* out, int64_t out_stride,
                                          int32_t out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    const char* who = "grx_constrained_logsoftmax";
    GRX_REQUIRE(rows != nullptr, GRX_E_INVALID, "%s: null rows", who);
    GRX_REQUIRE(out_dtype == GRX_F32 || out_dtype == rows->dtype, GRX_E_UNSUPPORTED,
                "%s: output must be fp32 or the input element type", who);
    GRX_REQUIRE(rows->R == 0 || (out != nullptr && out_stride >= rows->V), GRX_E_INVALID, "%s: bad output", who);
    GRX_REQUIRE(out != rows->scores || rows->R == 0, GRX_E_INVALID, "%s: out must not alias scores", who);
    MlsArgs a;
    memset(&a, 0, sizeof(a));
    a.out = out;
    a.out_stride = out_stride;
    const int kind = (out_dtype == GRX_F32) ? OUT_FULL_F32 : OUT_FULL_SAME;
    const bool out_aligned = rows_16b_aligned(out, out_stride, dtype_size(out_dtype));
    return run_step(who, trie, state, rows, con, a, kind, out_aligned, workspace, workspace_bytes,
                    (hipStream_t)stream);
}

extern "C" size_t grx_reuse_bytes(int32_t R) { return R > 0 ? (size_t)R * sizeof(grx_row_plan) : 0; }

extern "C" int grx_constrained_logsoftmax_reuse(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                                const grx_constraint* con, void* out, int64_t out_stride,
                                                int32_t out_dtype, void* last_rows, int32_t fresh, void* workspace,
                                                size_t workspace_bytes, void* stream) {
    const char* who = "grx_constrained_logsoftmax_reuse";
    GRX_REQUIRE(rows != nullptr, GRX_E_INVALID, "%s: null rows", who);
    GRX_REQUIRE(out_dtype == GRX_F32 || out_dtype == rows->dtype, GRX_E_UNSUPPORTED,
                "%s: output must be fp32 or the input element type", who);
    GRX_REQUIRE(rows->R == 0 || (out != nullptr && out_stride >= rows->V), GRX_E_INVALID, "%s: bad output", who);
    GRX_REQUIRE(out != rows->scores || rows->R == 0, GRX_E_INVALID, "%s: out must not alias scores", who);
    GRX_REQUIRE(fresh == 0 || fresh == 1, GRX_E_INVALID, "%s: fresh must be 0 or 1", who);
    GRX_REQUIRE(rows->R == 0 || (last_rows != nullptr && ((uintptr_t)last_rows & 15u) == 0), GRX_E_INVALID,
                "%s: last_rows must be a 16-byte aligned device buffer of grx_reuse_bytes(R) bytes", who);
    MlsArgs a;
    memset(&a, 0, sizeof(a));
    a.out = out;
    a.out_stride = out_stride;
    a.last = reinterpret_cast<grx_row_plan*>(last_rows);
    a.prefilled = fresh ? 0 : 1;
    const int kind = (out_dtype == GRX_F32) ? OUT_FULL_F32 : OUT_FULL_SAME;
    const bool out_aligned = rows_16b_aligned(out, out_stride, dtype_size(out_dtype));
    return run_step(who, trie, state, rows, con, a, kind, out_aligned, workspace, workspace_bytes,
                    (hipStream_t)stream);
}

extern "C" int64_t grx_candidate_stride(int32_t cmax) {
    if (cmax < 0) return 0;
    return (((int64_t)cmax + 1 + 15) / 16) * 16;
}

static int candidates_call(const char* who, const grx_trie* trie, grx_state* state, const grx_rows* rows,
                           const grx_constraint* con, grx_cand* cand, int64_t cand_stride, int32_t cmax,
                           void* workspace, size_t workspace_bytes, void* stream,
                           const grx_beam_step_args* beam = nullptr) {
    GRX_REQUIRE(rows != nullptr, GRX_E_INVALID, "%s: null rows", who);
    GRX_REQUIRE(cmax >= 1, GRX_E_INVALID, "%s: cmax must be >= 1", who);
    GRX_REQUIRE(cand_stride >= grx_candidate_stride(cmax) && cand_stride % 16 == 0, GRX_E_INVALID,
                "%s: cand_stride must be a multiple of 16 and >= %lld", who, (long long)grx_candidate_stride(cmax));
    GRX_REQUIRE(rows->R == 0 || (cand != nullptr && ((uintptr_t)cand & 127u) == 0), GRX_E_INVALID,
                "%s: cand must be a 128-byte aligned device buffer", who);
    MlsArgs a;
    memset(&a, 0, sizeof(a));
    a.cand = cand;
    a.cand_stride = cand_stride;
    a.cmax = cmax;
    return run_step(who, trie, state, rows, con, a, OUT_COMPACT, true, workspace, workspace_bytes,
                    (hipStream_t)stream, beam);
}

extern "C" int grx_constrained_beam_step(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                         const grx_constraint* con, grx_cand* cand, int64_t cand_stride, int32_t cmax,
                                         void* workspace, size_t workspace_bytes, const grx_beam_step_args* beam,
                                         void* stream) {
    const char* who = "grx_constrained_beam_step";
    GRX_REQUIRE(beam != nullptr, GRX_E_INVALID, "%s: null beam step arguments", who);
    return candidates_call(who, trie, state, rows, con, cand, cand_stride, cmax, workspace, workspace_bytes, stream, beam);
}

extern "C" int grx_constrained_candidates(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                          const grx_constraint* con, grx_cand* cand, int64_t cand_stride,
                                          int32_t cmax, void* workspace, size_t workspace_bytes, void* stream) {
    return candidates_call("grx_constrained_candidates", trie, state, rows, con, cand, cand_stride, cmax, workspace,
                           workspace_bytes, stream);
}

// ---- the entry point that allocates nothing -------------------------------------
// grx_constrained_candidates with the workspace (arrival counters and partial sums of split rows,
// fixed-hint row plans) taken from a device array that is part of the library image, for callers
// that cannot allocate.  Each GPU holds its own copy of the array at its own address; that address
// is looked up once per device and is what every kernel and memset below receives -- in host code
// the array's name is only a handle for the lookup, never a pointer to pass on.
#define GRX_NOALLOC_SCRATCH_BYTES (GRX_WS_PLAN_OFFSET + (size_t)GRX_NOALLOC_MAX_ROWS * sizeof(grx_row_plan))
#define GRX_NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[GRX_NOALLOC_SCRATCH_BYTES];
static void* g_scratch_ptr[GRX_NOALLOC_MAX_DEVICES];  // device address of `scratch` on each GPU, once looked up
static std::mutex g_scratch_mutex;

static int scratch_on_current_device(const char* who, void** out) {
    int dev = -1;
    GRX_CHECK_HIP(hipGetDevice(&dev));
    GRX_REQUIRE(dev >= 0 && dev < GRX_NOALLOC_MAX_DEVICES, GRX_E_UNSUPPORTED, "%s: device index %d not supported", who,
                dev);
    std::lock_guard<std::mutex> guard(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        GRX_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        GRX_REQUIRE(p != nullptr && ((uintptr_t)p & 15u) == 0, GRX_E_HIP, "%s: no device address for the scratch array",
                    who);
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return GRX_OK;
}

extern "C" int32_t grx_noalloc_max_rows(void) { return GRX_NOALLOC_MAX_ROWS; }

extern "C" int grx_constrained_candidates_noalloc(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                                  const grx_constraint* con, grx_cand* cand, int64_t cand_stride,
                                                  int32_t cmax, void* stream) {
    const char* who = "grx_constrained_candidates_noalloc";
    GRX_REQUIRE(rows != nullptr && trie != nullptr, GRX_E_INVALID, "%s: null trie/rows", who);
    GRX_REQUIRE(rows->R <= GRX_NOALLOC_MAX_ROWS, GRX_E_UNSUPPORTED,
                "%s: %d rows, the built-in scratch holds at most %d (use grx_constrained_candidates with a workspace)",
                who, rows->R, GRX_NOALLOC_MAX_ROWS);
    GRX_REQUIRE(rows->planned == 0, GRX_E_UNSUPPORTED,
                "%s: rows planned ahead live in the planner's workspace; pass planned = 0", who);
    int dev = -1;
    GRX_CHECK_HIP(hipGetDevice(&dev));
    GRX_REQUIRE(dev == trie->device, GRX_E_INVALID, "%s: the trie lives on device %d, current device is %d", who,
                trie->device, dev);
    void* ws = nullptr;
    int rc = scratch_on_current_device(who, &ws);
    if (rc != GRX_OK) return rc;
    // The array keeps what the last call left in it.  Row plans are rewritten by every call that
    // reads them; the arrival counters of split rows must start at zero, and although every
    // completed call hands them back zeroed, a call that was abandoned half way (device reset of
    // the stream, a capture that was dropped) would not have -- so they are cleared whenever they
    // are about to be used.
    if (rows->R > 0 && rows->R <= GRX_SPLIT_MAX_ROWS)
        GRX_CHECK_HIP(hipMemsetAsync(ws, 0, GRX_SPLIT_COUNT_BYTES, (hipStream_t)stream));
    return candidates_call(who, trie, state, rows, con, cand, cand_stride, cmax, ws, GRX_NOALLOC_SCRATCH_BYTES, stream);
}
