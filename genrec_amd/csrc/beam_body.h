// One beam-search iteration of one prompt group over compact candidates, as a device function: the body of
// beam_step_kernel, and the tail of the streaming kernel when that runs a group's step itself.
#pragma once

#include <math.h>

#include "common.h"

#define BS_THREADS 256
#define BS_WAVES (BS_THREADS / GRX_WAVE)
#define BS_MAXK 256
#define BS_MAXKEEP 512
#define BS_SURV_CAP 1024       // candidates ranked pairwise after the threshold cut
#define BS_SMALL_K 16          // the instantiation for common searches: at most this many beams,
#define BS_SMALL_KEEP 64       // ... kept candidates
#define BS_SMALL_SURV 512      // ... and survivors
#define BS_PREFETCH 16         // beams whose candidate rows are fetched in the opening batch of loads
#define BS_TAIL_K 64           // beams / positions per row of the sequence tails staged in LDS
#define BS_TAIL_W 16
#define BS_COLS 256            // candidate columns a row is dealt over (one per thread of a 256-thread part)
#define BS_MID_THREADS 1024    // the instantiation for groups of up to 64 beams with rows of at most BS_COLS candidates:
#define BS_MID_K 64            // ... four 256-thread parts, each holding one entry of every fourth beam in registers
#define BS_MID_KEEP 128
#define BS_MID_HOLD 16

struct BeamDerived {
    float fin_div;    // (cur_len + 1 - prompt_len) ** length_penalty
    float heur_div;   // best_hypothetical_length ** length_penalty
    int reg_keys;     // every candidate slot is covered by the opening batch of loads: keys stay in registers
    int tails_in_lds; // the re-ordered parts of the sequence / parent rows fit the LDS tail buffers
    int hold_keys;    // HOLD instantiation: part q = tid / BS_COLS of the workgroup keeps entry (tid - 37 k) mod BS_COLS of the
                      // beams k = q, q + parts, ... in registers, fetched in the opening batch of loads
};

// 64-bit sort key: ordered score in the high word; the low word is the
// complement of (beam << 24 | token), which orders like beam * V + token.
__device__ __forceinline__ unsigned long long make_key(float score, uint32_t beam, uint32_t tok) {
    return ((unsigned long long)float_to_ordered(score) << 32) |
           (unsigned long long)(0xFFFFFFFFu - ((beam << 24) | tok));
}

__device__ __forceinline__ bool ranks_before(float va, int ia, float vb, int ib) {
    return va > vb || (va == vb && ia < ib);
}

#ifndef STAMP
#define STAMP(n) do { } while (0)
#endif

// A candidate entry.  COH: the rows were written earlier in this same kernel by other workgroups (with
// device-scope stores); they are read with device-scope loads, which do not stop at this XCD's L2.
template <bool COH>
__device__ __forceinline__ grx_cand cand_load(const grx_cand* p) {
    if constexpr (COH) {
        const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        grx_cand e;
        e.x = (int32_t)(uint32_t)(w & 0xFFFFFFFFull);
        e.y = (int32_t)(uint32_t)(w >> 32);
        return e;
    } else {
        return *p;
    }
}

// Layout: the candidates of a group are the K x cmax block behind its first row;
// slot c = beam * cmax + position holds a key, or 0 when position >= count[beam].
// The LDS arrays are sized by the instantiation: MAXK beams, MAXKEEP kept candidates, SURV_CAP
// survivors of the threshold cut, tail buffers for TAIL_K beams.  The common search (at most 16
// beams, 64 kept candidates) gets by with 14 KB where the largest supported one takes 54 KB, and the
// number of groups a CU works on at once follows from that.
// NT: threads of the workgroup (a multiple of BS_COLS).  HOLD > 0: entries per thread the instantiation can keep in
// registers (d.hold_keys; K <= HOLD * NT / BS_COLS).
// b: the group.  Every thread of the workgroup must call it.
template <int MAXK, int MAXKEEP, int SURV_CAP, int TAIL_K, bool COH, int HOLD = 0, int NT = BS_THREADS>
__device__ __forceinline__ void beam_step_group(const grx_beam_step_args& a, const BeamDerived& d, const PlanCtx& pc,
                                                grx_row_plan* __restrict__ plan_out, const int b) {
    constexpr int NW = NT / GRX_WAVE;
    constexpr int PARTS = NT / BS_COLS;
    static_assert(NT % BS_COLS == 0 && (NT & (NT - 1)) == 0, "workgroup = whole 256-thread parts");
    __shared__ int s_cnt[MAXK];
    __shared__ int s_base[MAXK + 1];
    __shared__ float s_runin[MAXK];
    __shared__ unsigned char s_fdone_in[MAXK];
    __shared__ unsigned long long s_tmax[NT];
    __shared__ unsigned long long s_pmax[HOLD > 0 ? BS_COLS : 1];  // HOLD: column maxima over the parts
    __shared__ int s_prank[HOLD > 0 ? BS_COLS : 1];                // ... and their ranks
    __shared__ unsigned long long s_surv[SURV_CAP];
    __shared__ int s_nsurv;
    __shared__ unsigned long long s_thresh;
    __shared__ unsigned long long s_red[2][NW];
    __shared__ float s_score[MAXKEEP];
    __shared__ int s_parent[MAXKEEP];
    __shared__ int s_tok[MAXKEEP];
    __shared__ unsigned char s_hit[MAXKEEP];
    __shared__ float s_rs[MAXKEEP];
    __shared__ float s_merged[MAXK + MAXKEEP];
    __shared__ int s_run_idx[MAXK];
    __shared__ int s_fin_idx[MAXK];
    __shared__ float s_newfin[MAXK];
    __shared__ unsigned char s_newdone[MAXK];
    __shared__ int64_t s_rseq[TAIL_K * BS_TAIL_W];
    __shared__ int64_t s_fseq[TAIL_K * BS_TAIL_W];
    __shared__ int32_t s_rbidx[TAIL_K * BS_TAIL_W];
    __shared__ int32_t s_fbidx[TAIL_K * BS_TAIL_W];

    const int tid = threadIdx.x;
    const int lane = tid & (GRX_WAVE - 1);
    const int wave = tid / GRX_WAVE;
    const int K = a.K, n_keep = a.n_keep, cmax = a.cmax;
    const int64_t row0 = (int64_t)b * K;
    const int L = a.max_length;
    const int Lg = a.max_length - a.prompt_len;
    const int gpos = a.cur_len - a.prompt_len;
    const int w_run = a.cur_len - a.copy_from;        // positions of a running row that move with a re-order
    const int w_fin = a.cur_len + 1 - a.copy_from;    // ... of a finished row
    const int slots = K * cmax;
    STAMP(0);

    // ---- phase A: every global read the step needs is issued before the first wait.
    // Order of issue: the per-beam scalars and the sequence tails first, the candidate entries last, and every LDS
    // store after all of them.  (The compiler moves the first instructions of the key build up to the candidate
    // loads and waits for them there: with the candidates asked for first, the other loads went out a round trip
    // later, and load-and-store pairs cost one round trip each.)
    const bool has_k = tid < K;  // K <= MAXK <= NT
    static_assert(MAXK <= NT, "one thread per beam for the per-beam scalars");
    int c_cnt = 0;
    float c_run = 0.f, c_fin = 0.f;
    unsigned char c_done = 0;
    const int64_t bstride = a.cand_shared ? 0 : a.cand_stride;
    // candidate row of beam k: g_cand + k * bstride (every beam reads the group's one row when they are shared)
    const grx_cand* __restrict__ g_cand = a.cand + (a.cand_shared ? (int64_t)b : row0) * a.cand_stride;
    if (has_k) {
        c_cnt = cand_load<COH>(g_cand + (int64_t)tid * bstride).x;
        c_run = a.run_score_in[row0 + tid];
        c_done = a.fin_done_in[row0 + tid];
        c_fin = a.fin_score_in[row0 + tid];
    }
    const bool unsat_old = a.unsat[b] != 0;
    // K <= BS_TAIL_K rows of at most BS_TAIL_W moving positions: thread = (row, position)
    const int tp = tid & (BS_TAIL_W - 1), tk0 = tid / BS_TAIL_W;
    const bool t_on = d.tails_in_lds && tk0 < K;
    int64_t t_fs = 0, t_rs = 0;
    int32_t t_fb = 0, t_rb = 0;
    if (t_on) {
        if (tp < w_fin) t_fs = a.fin_seq_in[(row0 + tk0) * L + a.copy_from + tp];
        if (tp < w_run) t_rs = a.run_seq_in[(row0 + tk0) * L + a.copy_from + tp];
        if (tp < gpos + 1) t_fb = a.fin_bidx_in[(row0 + tk0) * Lg + tp];
        if (tp < gpos) t_rb = a.run_bidx_in[(row0 + tk0) * Lg + tp];
    }
    // Thread t prefetches one entry of each of the first BS_PREFETCH beams: entry (t - 37 u) mod 256 of
    // beam u (coalesced, rotated rows).  The rotation spreads short rows over all threads, so that the
    // per-lane maxima below stay informative when every row holds only a few candidates: unrotated, rows
    // of c < n_keep entries would leave all but c lanes empty and nothing could be cut.
    grx_cand pre[BS_PREFETCH];
    auto pre_col = [&](int u) { return (tid - 37 * u) & (NT - 1); };
    if (HOLD == 0 && d.reg_keys) {
#pragma unroll
        for (int u = 0; u < BS_PREFETCH; ++u) {
            // entries past a beam's count hold stale values; they are masked below
            pre[u] = cand_load<COH>(g_cand + ((u < K && pre_col(u) < cmax) ? (int64_t)u * bstride + 1 + pre_col(u) : 0));
        }
    }
    // Wide groups, same idea with more threads: part q holds entry (tid - 37 k) mod BS_COLS of the beams k = q + PARTS v.
    // Everything is asked for at once, before the counts are known (entries past a row's count are masked
    // when the keys are built): one round trip to the candidates instead of two passes of several rounds.
    constexpr int HN = HOLD > 0 ? HOLD : 1;
    unsigned long long hk[HN];
    const int hq = tid / BS_COLS;
    auto hold_col = [&](int k) { return (tid - 37 * k) & (BS_COLS - 1); };
    if constexpr (HOLD > 0) {
        if (d.hold_keys) {
#pragma unroll
            for (int v = 0; v < HOLD; ++v) {
                const int k = hq + PARTS * v;
                const bool there = k < K && hold_col(k) < cmax;
                const grx_cand e = cand_load<COH>(g_cand + (there ? (int64_t)k * bstride + 1 + hold_col(k) : 0));
                hk[v] = ((unsigned long long)(uint32_t)e.y << 32) | (uint32_t)e.x;
            }
            if (tid < BS_COLS) {
                s_pmax[tid] = 0;
                s_prank[tid] = 0;
            }
        }
    }
    if (has_k) {
        s_cnt[tid] = c_cnt;
        s_runin[tid] = c_run;
        s_fdone_in[tid] = c_done;
        s_merged[tid] = c_fin;
    }
    if (t_on) {
        if (tp < w_fin) s_fseq[tk0 * BS_TAIL_W + tp] = t_fs;
        if (tp < w_run) s_rseq[tk0 * BS_TAIL_W + tp] = t_rs;
        if (tp < gpos + 1) s_fbidx[tk0 * BS_TAIL_W + tp] = t_fb;
        if (tp < gpos) s_rbidx[tk0 * BS_TAIL_W + tp] = t_rb;
    }
    if (d.tails_in_lds) {  // further rows (more beams than NT / BS_TAIL_W)
        for (int k = tk0 + NT / BS_TAIL_W; k < K; k += NT / BS_TAIL_W) {
            int64_t fs = 0, rs = 0;
            int32_t fb = 0, rb = 0;
            if (tp < w_fin) fs = a.fin_seq_in[(row0 + k) * L + a.copy_from + tp];
            if (tp < w_run) rs = a.run_seq_in[(row0 + k) * L + a.copy_from + tp];
            if (tp < gpos + 1) fb = a.fin_bidx_in[(row0 + k) * Lg + tp];
            if (tp < gpos) rb = a.run_bidx_in[(row0 + k) * Lg + tp];
            if (tp < w_fin) s_fseq[k * BS_TAIL_W + tp] = fs;
            if (tp < w_run) s_rseq[k * BS_TAIL_W + tp] = rs;
            if (tp < gpos + 1) s_fbidx[k * BS_TAIL_W + tp] = fb;
            if (tp < gpos) s_rbidx[k * BS_TAIL_W + tp] = rb;
        }
    }
    if (tid == 0) {
        s_nsurv = 0;
        s_thresh = 0;
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);
#if defined(BS_ABLATE) && BS_ABLATE == 1
    if (tid == 0) a.unsat[b] = (unsigned char)(s_cnt[0] + pre[0].x + pre[1].y + (int)s_fseq[0] + s_rbidx[0]);
    return;
#endif

    auto key_at = [&](int k, int i, float lp, int tk) -> unsigned long long {
        return i < s_cnt[k] ? make_key(lp + s_runin[k], (uint32_t)k, (uint32_t)tk) : 0ull;
    };
    auto key_of = [&](int c) -> unsigned long long {  // flat slot c = beam * cmax + position
        const int k = c / cmax, i = c - k * cmax;
        const grx_cand e = cand_load<COH>(g_cand + (int64_t)k * bstride + 1 + i);
        return key_at(k, i, __int_as_float(e.y), e.x);
    };
    auto emit = [&](int j, unsigned long long key) {
        if (key == 0) {  // fewer candidates than n_keep: a -inf filler slot
            s_score[j] = -INFINITY;
            s_parent[j] = 0;
            s_tok[j] = 0;
        } else {
            const uint32_t packed = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
            s_score[j] = ordered_to_float((uint32_t)(key >> 32));
            s_parent[j] = (int)(packed >> 24);
            s_tok[j] = (int)(packed & 0xFFFFFFu);
        }
    };

    // ---- top n_keep by (score desc, beam * V + token asc).  Keys are unique, so
    // the rank of a key is the number of larger keys.  Fast path: a threshold
    // (the n_keep-th largest of the per-thread maxima, which at least n_keep keys
    // reach) leaves a few dozen survivors; they are ranked against each other.
    bool selected = false;
    if (HOLD == 0 && d.reg_keys) {
        // K <= BS_PREFETCH beams of at most NT candidates, n_keep <= 64: thread t holds one entry
        // of every beam.  64 partition maxima (lane-wise over the four waves) give the threshold.
        unsigned long long rk[BS_PREFETCH];
        unsigned long long tmax = 0;
        // counts and running scores of all beams first, unconditionally (entries beyond K are never used): read
        // beam by beam inside the loop, every key waited for two LDS round trips of its own
        static_assert(BS_PREFETCH <= MAXK, "one count and one running score per prefetched beam");
        int cnt_r[BS_PREFETCH];
        float rin_r[BS_PREFETCH];
#pragma unroll
        for (int u = 0; u < BS_PREFETCH; ++u) {
            cnt_r[u] = s_cnt[u];
            rin_r[u] = s_runin[u];
        }
#pragma unroll
        for (int u = 0; u < BS_PREFETCH; ++u) {
            const bool there = u < K && pre_col(u) < cmax && pre_col(u) < cnt_r[u];
            rk[u] = there ? make_key(__int_as_float(pre[u].y) + rin_r[u], (uint32_t)u, (uint32_t)pre[u].x) : 0ull;
            tmax = rk[u] > tmax ? rk[u] : tmax;
        }
        s_tmax[tid] = tmax;
        STAMP(3);
        __syncthreads();
        if (wave == 0) {
            unsigned long long pm = s_tmax[lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) pm = s_tmax[lane + w * GRX_WAVE] > pm ? s_tmax[lane + w * GRX_WAVE] : pm;
            int rank = 0;  // keys are unique, so the partition maxima are distinct (or 0)
            const int pm_lo = (int)(unsigned)(pm & 0xFFFFFFFFull), pm_hi = (int)(unsigned)(pm >> 32);
#pragma unroll
            for (int j = 0; j < GRX_WAVE; ++j) {  // lane j's value as a scalar: no LDS traffic
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane(pm_lo, j);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane(pm_hi, j);
                rank += (((unsigned long long)hi << 32) | lo) > pm ? 1 : 0;
            }
            if (pm != 0 && rank == n_keep - 1) s_thresh = pm;  // stays 0 when fewer than n_keep partitions hold a key
        }
        STAMP(5);
        __syncthreads();
        const unsigned long long thresh = s_thresh;
#pragma unroll
        for (int u = 0; u < BS_PREFETCH; ++u) {
            if (rk[u] != 0 && rk[u] >= thresh) {
                const int at = atomicAdd(&s_nsurv, 1);
                if (at < SURV_CAP) s_surv[at] = rk[u];
            }
        }
        STAMP(6);
        __syncthreads();
        const int M = s_nsurv;
        if (M <= SURV_CAP) {
            for (int i = tid; i < M; i += NT) {
                const unsigned long long kv = s_surv[i];
                int rank = 0;
#pragma unroll 8
                for (int j = 0; j < M; ++j) rank += s_surv[j] > kv ? 1 : 0;
                if (rank < n_keep) emit(rank, kv);
            }
            for (int j = M + tid; j < n_keep; j += NT) emit(j, 0);
            selected = true;
        }
    } else if (HOLD > 0 && d.hold_keys) {
        if constexpr (HOLD > 0) {
            unsigned long long tmax = 0;
            static_assert(HOLD * PARTS <= MAXK, "one count and one running score per held beam");
            int cnt_r[HN];
            float rin_r[HN];
#pragma unroll
            for (int v = 0; v < HOLD; ++v) {  // unconditional reads (entries beyond K are never used), all in flight at once
                cnt_r[v] = s_cnt[hq + PARTS * v];
                rin_r[v] = s_runin[hq + PARTS * v];
            }
#pragma unroll
            for (int v = 0; v < HOLD; ++v) {
                const int k = hq + PARTS * v;
                const bool there = k < K && hold_col(k) < cmax && hold_col(k) < cnt_r[v];
                const unsigned long long raw = hk[v];
                const unsigned long long kv =
                    there ? make_key(__int_as_float((int)(uint32_t)(raw >> 32)) + rin_r[v], (uint32_t)k, (uint32_t)raw) : 0ull;
                hk[v] = kv;
                tmax = kv > tmax ? kv : tmax;
            }
            // BS_COLS column maxima (over the parts) give the threshold; the ranking of a column's maximum is
            // shared by the threads that hold the column, 1 / PARTS of the comparisons each
            const int col = tid & (BS_COLS - 1);
            if (tmax != 0) atomicMax(&s_pmax[col], tmax);
            STAMP(4);
            __syncthreads();
            const unsigned long long pm = s_pmax[col];
            {
                constexpr int SPAN = BS_COLS / PARTS;
                int cnt = 0;
#pragma unroll 16
                for (int j = hq * SPAN; j < (hq + 1) * SPAN; ++j) cnt += s_pmax[j] > pm ? 1 : 0;
                if (pm != 0 && cnt != 0) atomicAdd(&s_prank[col], cnt);
            }
            __syncthreads();
            if (hq == 0 && pm != 0 && s_prank[col] == n_keep - 1) s_thresh = pm;  // stays 0 with fewer than n_keep columns
            STAMP(5);
            __syncthreads();
            const unsigned long long thresh = s_thresh;
#pragma unroll
            for (int v = 0; v < HOLD; ++v) {
                const unsigned long long kv = hk[v];
                if (kv != 0 && kv >= thresh) {
                    const int at = atomicAdd(&s_nsurv, 1);
                    if (at < SURV_CAP) s_surv[at] = kv;
                }
            }
            STAMP(6);
            __syncthreads();
            const int M = s_nsurv;
            if (M <= SURV_CAP) {
                // PARTS neighbouring lanes share a survivor: each counts every PARTS-th comparison
                for (int i = tid / PARTS; i < M; i += NT / PARTS) {
                    const unsigned long long kv = s_surv[i];
                    int rank = 0;
#pragma unroll 8
                    for (int j = tid % PARTS; j < M; j += PARTS) rank += s_surv[j] > kv ? 1 : 0;
#pragma unroll
                    for (int o = 1; o < PARTS; o <<= 1) rank += __shfl_xor(rank, o, GRX_WAVE);
                    if (tid % PARTS == 0 && rank < n_keep) emit(rank, kv);
                }
                for (int j = M + tid; j < n_keep; j += NT) emit(j, 0);
                selected = true;
            }
        }
    } else if (HOLD == 0) {
        // Wider groups: two passes over the existing candidates straight from global memory (they were
        // written a moment ago and sit in L2) -- per-thread maxima and threshold first, survivors second.
        // Two ways to deal the candidates to the threads, whichever needs fewer rounds of loads:
        //  * packed: candidate c of the group (entry c - s_base[k] of beam k, the largest k with
        //    s_base[k] <= c) goes to thread c mod 256 -- few candidates, few rounds;
        //  * by column: thread t takes one entry per beam and 256-entry chunk, entry (t - 37 k) mod 256
        //    of beam k (the rotation spreads short rows over all threads, which keeps the per-thread
        //    maxima informative) -- the loads of different beams need no index search and are issued
        //    back to back, which wins when the rows are long.
        for (int k = tid; k <= K; k += NT) {
            int acc = 0;
            for (int j = 0; j < k; ++j) acc += s_cnt[j] < cmax ? s_cnt[j] : cmax;
            s_base[k] = acc;
        }
        __syncthreads();
        const int total = s_base[K];
        const int n_chunk = (cmax + NT - 1) / NT;
        const bool by_column = (total + NT - 1) / NT > 2 * ((K + 15) / 16) * n_chunk;
        auto packed_key = [&](int c) -> unsigned long long {
            int lo = 0, hi = K;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_base[mid] <= c) lo = mid; else hi = mid;
            }
            const grx_cand e = cand_load<COH>(g_cand + (int64_t)lo * bstride + 1 + (c - s_base[lo]));
            return make_key(__int_as_float(e.y) + s_runin[lo], (uint32_t)lo, (uint32_t)e.x);
        };
        // keys of the entries that thread `tid` holds for beams k0 .. k0 + 15 in chunk j: the 16 loads
        // are unconditional (the row header stands in where there is no entry) and issued together
        constexpr int CB = 16;
        auto column_keys = [&](int k0, int j, unsigned long long* kv) {
            grx_cand e[CB];
            bool ok[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int k = k0 + u < K ? k0 + u : K - 1;
                const int i = j * NT + ((tid - 37 * k) & (NT - 1));
                ok[u] = k0 + u < K && i < (s_cnt[k] < cmax ? s_cnt[k] : cmax);
                e[u] = cand_load<COH>(g_cand + (int64_t)k * bstride + (ok[u] ? 1 + i : 0));
            }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int k = k0 + u < K ? k0 + u : K - 1;
                kv[u] = ok[u] ? make_key(__int_as_float(e[u].y) + s_runin[k], (uint32_t)k, (uint32_t)e[u].x) : 0ull;
            }
        };
        unsigned long long tmax = 0;
        if (by_column) {
            for (int j = 0; j < n_chunk; ++j) {
                for (int k0 = 0; k0 < K; k0 += CB) {
                    unsigned long long kv[CB];
                    column_keys(k0, j, kv);
#pragma unroll
                    for (int u = 0; u < CB; ++u) tmax = kv[u] > tmax ? kv[u] : tmax;
                }
            }
        } else {
#pragma unroll 4
            for (int c = tid; c < total; c += NT) {
                const unsigned long long kv = packed_key(c);
                tmax = kv > tmax ? kv : tmax;
            }
        }
        s_tmax[tid] = tmax;
        STAMP(4);
        __syncthreads();
        if (tmax != 0) {
            int rank = 0;
#pragma unroll 16
            for (int j = 0; j < NT; ++j) rank += s_tmax[j] > tmax ? 1 : 0;
            if (rank == n_keep - 1) s_thresh = tmax;  // stays 0 when fewer than n_keep threads hold a key
        }
        STAMP(5);
        __syncthreads();
        const unsigned long long thresh = s_thresh;
        auto keep = [&](unsigned long long kv) {
            if (kv != 0 && kv >= thresh) {
                const int at = atomicAdd(&s_nsurv, 1);
                if (at < SURV_CAP) s_surv[at] = kv;
            }
        };
        if (by_column) {
            for (int j = 0; j < n_chunk; ++j) {
                for (int k0 = 0; k0 < K; k0 += CB) {
                    unsigned long long kv[CB];
                    column_keys(k0, j, kv);
#pragma unroll
                    for (int u = 0; u < CB; ++u) keep(kv[u]);
                }
            }
        } else {
#pragma unroll 4
            for (int c = tid; c < total; c += NT) keep(packed_key(c));
        }
        STAMP(6);
        __syncthreads();
        const int M = s_nsurv;
        if (M <= SURV_CAP) {
            for (int i = tid; i < M; i += NT) {
                const unsigned long long kv = s_surv[i];
                int rank = 0;
#pragma unroll 8
                for (int j = 0; j < M; ++j) rank += s_surv[j] > kv ? 1 : 0;
                if (rank < n_keep) emit(rank, kv);
            }
            for (int j = M + tid; j < n_keep; j += NT) emit(j, 0);
            selected = true;
        }
    }
    if (!selected) {
        // general path: n_keep rounds of "largest key below the previous winner"
        unsigned long long prev = ~0ull;
        for (int j = 0; j < n_keep; ++j) {
            unsigned long long best = 0;
            if (prev != 0) {
                for (int c = tid; c < slots; c += NT) {
                    const unsigned long long kv = key_of(c);
                    if (kv < prev && kv > best) best = kv;
                }
                best = wave_max_u64(best);
                if (lane == 0) s_red[j & 1][wave] = best;
                __syncthreads();
                best = s_red[j & 1][0];
#pragma unroll
                for (int w = 1; w < NW; ++w) best = s_red[j & 1][w] > best ? s_red[j & 1][w] : best;
            }
            if (tid == 0) emit(j, best);
            prev = best;
        }
    }
    STAMP(7);
    __syncthreads();
    STAMP(8);
#if defined(BS_ABLATE) && BS_ABLATE == 2
    if (tid == 0) a.unsat[b] = (unsigned char)(s_tok[0] + s_parent[1]);
    return;
#endif

    // ---- stopping criteria and the two masked score vectors
    const bool at_max_len = (a.cur_len + 1) >= a.max_length;
    int n_done = 0;
    if (K <= GRX_WAVE) {
        n_done = (int)__popcll(__ballot(lane < K && s_fdone_in[lane < K ? lane : 0] != 0));
    } else {
#pragma unroll 8
        for (int k = 0; k < K; ++k) n_done += s_fdone_in[k] != 0 ? 1 : 0;
    }
    const bool all_done = n_done == K;
    const bool beams_full = all_done && (a.early_stopping == 1);
    for (int j = tid; j < n_keep; j += NT) {
        const bool hit = at_max_len || (a.eos_token_id >= 0 && s_tok[j] == a.eos_token_id);
        s_hit[j] = hit ? 1 : 0;
        const float sc = s_score[j];
        s_rs[j] = hit ? sc + (-1.0e9f) : sc;
        float fs = sc / d.fin_div;
        if (beams_full) fs += -1.0e9f;
        if (!unsat_old) fs += -1.0e9f;
        if (!(hit && j < K)) fs += -1.0e9f;
        s_merged[K + j] = fs;
    }
    STAMP(9);
    __syncthreads();

    // ---- running top-K (of n_keep) and finished top-K (of K + n_keep) by rank counting
    const int n_merged = K + n_keep;
    if (HOLD == 0 && n_keep + n_merged <= NT) {
        // Short lists (with many beams the 64-bit keys of the next branch cost less per comparison).  One
        // element of either list per group of rp neighbouring lanes (rp a power of two, as many as the
        // workgroup has to spare): each lane counts every rp-th comparison, the group adds up.
        const int total = n_keep + n_merged;
        int sh = 0;
        while (sh < 4 && (total << (sh + 1)) <= NT) ++sh;
        const int rp = 1 << sh;
        const int idx = tid >> sh, part = tid & (rp - 1);
        const bool running = idx < n_keep;
        const float* __restrict__ vals = running ? s_rs : s_merged;
        const int n = running ? n_keep : n_merged;
        const int j = running ? idx : idx - n_keep;
        int rank = 0;
        if (idx < total) {
            const float v = vals[j];
#pragma unroll 4
            for (int i = part; i < n; i += rp) rank += ranks_before(vals[i], i, v, j) ? 1 : 0;
        }
        for (int o = 1; o < rp; o <<= 1) rank += __shfl_xor(rank, o, GRX_WAVE);
        if (idx < total && part == 0 && rank < K) (running ? s_run_idx : s_fin_idx)[rank] = j;
    } else if (n_keep + n_merged <= SURV_CAP) {
        // Larger lists: (score descending, index ascending) becomes one unique 64-bit key per element --
        // kept in the survivor buffer, which is free by now -- and the rank of an element is the number
        // of larger keys: one 64-bit compare per pair instead of three compares and their logic.
        unsigned long long* const rkey = s_surv;
        unsigned long long* const mkey = s_surv + n_keep;
        for (int j = tid; j < n_keep + n_merged; j += NT) {
            const bool running = j < n_keep;
            const int i = running ? j : j - n_keep;
            s_surv[j] = ((unsigned long long)float_to_ordered(running ? s_rs[i] : s_merged[i]) << 32) |
                        (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
        }
        __syncthreads();
        // both lists at once; PARTS neighbouring lanes share an element and count every PARTS-th comparison
        for (int idx = tid / PARTS; idx < n_keep + n_merged; idx += NT / PARTS) {
            const bool running = idx < n_keep;
            const unsigned long long* __restrict__ keys = running ? rkey : mkey;
            const int n = running ? n_keep : n_merged;
            const int j = running ? idx : idx - n_keep;
            const unsigned long long mine = keys[j];
            int rank = 0;
#pragma unroll 8
            for (int i = tid % PARTS; i < n; i += PARTS) rank += keys[i] > mine ? 1 : 0;
#pragma unroll
            for (int o = 1; o < PARTS; o <<= 1) rank += __shfl_xor(rank, o, GRX_WAVE);
            if (tid % PARTS == 0 && rank < K) (running ? s_run_idx : s_fin_idx)[rank] = j;
        }
    } else {
        for (int j = tid; j < n_keep; j += NT) {
            const float v = s_rs[j];
            int rank = 0;
#pragma unroll 8
            for (int i = 0; i < n_keep; ++i) rank += ranks_before(s_rs[i], i, v, j) ? 1 : 0;
            if (rank < K) s_run_idx[rank] = j;
        }
        for (int j = tid; j < n_merged; j += NT) {
            const float v = s_merged[j];
            int rank = 0;
#pragma unroll 8
            for (int i = 0; i < n_merged; ++i) rank += ranks_before(s_merged[i], i, v, j) ? 1 : 0;
            if (rank < K) s_fin_idx[rank] = j;
        }
    }
    STAMP(10);
    __syncthreads();
    STAMP(11);
#if defined(BS_ABLATE) && BS_ABLATE == 3
    if (tid == 0) a.unsat[b] = (unsigned char)(s_run_idx[0] + s_fin_idx[1]);
    return;
#endif

    // ---- write the per-step results
    for (int j = tid; j < n_keep; j += NT) {
        const int64_t o = (int64_t)b * n_keep + j;
        if (a.topk_score) a.topk_score[o] = s_score[j];
        if (a.topk_parent) a.topk_parent[o] = s_parent[j];
        if (a.topk_token) a.topk_token[o] = s_tok[j];
        if (a.topk_hit) a.topk_hit[o] = s_hit[j];
    }
    for (int k = tid; k < K; k += NT) {
        const int j = s_run_idx[k];
        a.run_score_out[row0 + k] = s_rs[j];
        a.run_seq_out[(row0 + k) * L + a.cur_len] = (int64_t)s_tok[j];
        a.run_bidx_out[(row0 + k) * Lg + gpos] = (int)row0 + s_parent[j];
        if (a.next_parent) a.next_parent[row0 + k] = (int)row0 + s_parent[j];

        const int m = s_fin_idx[k];
        const float fscore = s_merged[m];
        unsigned char fdone;
        if (m < K) {
            fdone = s_fdone_in[m];
        } else {
            const int jj = m - K;
            fdone = (s_hit[jj] && jj < K) ? 1 : 0;
            a.fin_seq_out[(row0 + k) * L + a.cur_len] = (int64_t)s_tok[jj];
            a.fin_bidx_out[(row0 + k) * Lg + gpos] = (int)row0 + s_parent[jj];
        }
        a.fin_score_out[row0 + k] = fscore;
        a.fin_done_out[row0 + k] = fdone;
        s_newfin[k] = fscore;
        s_newdone[k] = fdone;
    }

    // sequence bodies [copy_from, cur_len) (+ position cur_len for carried-over finished rows)
    // and parent trails [0, gpos) (+ position gpos for carried-over finished rows)
    if (d.tails_in_lds) {
        const int p = tid & (BS_TAIL_W - 1);
        for (int k = tid / BS_TAIL_W; k < K; k += NT / BS_TAIL_W) {
            const int rsrc = s_parent[s_run_idx[k]];
            const int m = s_fin_idx[k];
            if (p < w_run) a.run_seq_out[(row0 + k) * L + a.copy_from + p] = s_rseq[rsrc * BS_TAIL_W + p];
            if (p < gpos) a.run_bidx_out[(row0 + k) * Lg + p] = s_rbidx[rsrc * BS_TAIL_W + p];
            if (m < K) {
                if (p < w_fin) a.fin_seq_out[(row0 + k) * L + a.copy_from + p] = s_fseq[m * BS_TAIL_W + p];
                if (p < gpos + 1) a.fin_bidx_out[(row0 + k) * Lg + p] = s_fbidx[m * BS_TAIL_W + p];
            } else {
                const int fsrc = s_parent[m - K];
                if (p < w_run) a.fin_seq_out[(row0 + k) * L + a.copy_from + p] = s_rseq[fsrc * BS_TAIL_W + p];
                if (p < gpos) a.fin_bidx_out[(row0 + k) * Lg + p] = s_rbidx[fsrc * BS_TAIL_W + p];
            }
        }
    } else {
        for (int idx = tid; idx < K * w_run; idx += NT) {
            const int k = idx / w_run, p = idx % w_run;
            const int src = s_parent[s_run_idx[k]];
            a.run_seq_out[(row0 + k) * L + a.copy_from + p] = a.run_seq_in[(row0 + src) * L + a.copy_from + p];
        }
        for (int idx = tid; idx < K * w_fin; idx += NT) {
            const int k = idx / w_fin, p = idx % w_fin;
            const int m = s_fin_idx[k];
            const int64_t at = (row0 + k) * L + a.copy_from + p;
            if (m < K) {
                a.fin_seq_out[at] = a.fin_seq_in[(row0 + m) * L + a.copy_from + p];
            } else if (p < w_run) {
                a.fin_seq_out[at] = a.run_seq_in[(row0 + s_parent[m - K]) * L + a.copy_from + p];
            }
        }
        for (int idx = tid; idx < K * gpos; idx += NT) {
            const int k = idx / gpos, q = idx % gpos;
            a.run_bidx_out[(row0 + k) * Lg + q] = a.run_bidx_in[(row0 + s_parent[s_run_idx[k]]) * Lg + q];
        }
        for (int idx = tid; idx < K * (gpos + 1); idx += NT) {
            const int k = idx / (gpos + 1), q = idx % (gpos + 1);
            const int m = s_fin_idx[k];
            if (m < K) {
                a.fin_bidx_out[(row0 + k) * Lg + q] = a.fin_bidx_in[(row0 + m) * Lg + q];
            } else if (q < gpos) {
                a.fin_bidx_out[(row0 + k) * Lg + q] = a.run_bidx_in[(row0 + s_parent[m - K]) * Lg + q];
            }
        }
    }
    STAMP(12);
    __syncthreads();
    STAMP(13);

    // ---- early-stop heuristic for the next iteration
    if (K <= GRX_WAVE) {
        if (wave == 0) {
            const float best_possible = s_rs[s_run_idx[0]] / d.heur_div;
            const float mine = lane < K ? s_newfin[lane] : INFINITY;
            float worst_all = mine;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) worst_all = fminf(worst_all, __shfl_xor(worst_all, o, GRX_WAVE));
            const float worst = (lane < K && s_newdone[lane]) ? worst_all : -1.0e9f;
            const bool any = __ballot(lane < K && best_possible > worst) != 0ull;
            if (lane == 0) a.unsat[b] = (unsat_old && any) ? 1 : 0;
        }
    } else if (tid == 0) {
        const float best_possible = s_rs[s_run_idx[0]] / d.heur_div;
        float worst_all = s_newfin[0];
#pragma unroll 8
        for (int k = 1; k < K; ++k) worst_all = fminf(worst_all, s_newfin[k]);
        bool any = false;
#pragma unroll 8
        for (int k = 0; k < K; ++k) {
            const float worst = s_newdone[k] ? worst_all : -1.0e9f;
            any = any || (best_possible > worst);
        }
        a.unsat[b] = (unsat_old && any) ? 1 : 0;
    }
    STAMP(14);

    // ---- optionally plan the next processor call for the rows this group just produced
    if (plan_out != nullptr) {
        // the sequence rows were written by other threads of this workgroup (barrier above STAMP(13))
        const int T_next = a.cur_len + 1;
        if (pc.mode == GRX_MODE_COUNT) {
            // the streaming kernel walks the rows itself; what it needs from here is the once-per-call
            // bookkeeping and the lowest row that ends in prefix_ids
            if (b == 0 && tid == 0) plan_count_bookkeeping(pc);
            for (int k = tid; k < K; k += NT)
                count_row_match(pc, a.run_seq_out + (row0 + k) * L, T_next, (int)(row0 + k));
        } else {
            // Fixed-hint mode.  A new row is its parent's row plus one token, so its first valid prefix
            // occurrence is the parent's whenever the parent's node continues with that token onto a
            // node that has children: occurrences before the parent's stay invalid with one more token.
            // Rows that end in EOS take the EOS rule; everything else (parent already finished, leaf
            // reached without an EOS id) is rescanned the general way, one wave per row.
            __shared__ int s_pcode[MAXK];
            __shared__ unsigned char s_rescan[MAXK];
            // This call's plans sit where the next call's are about to be written: a workgroup reads and
            // then overwrites its own K entries.  A call on one row per group has its B plans spread over
            // entries that OTHER workgroups overwrite, so there the parents' nodes are not used and every
            // row takes the general scan (first step of a search only).
            for (int k = tid; k < K; k += NT)
                s_pcode[k] = a.cand_shared ? GRX_RES_EMPTY : plan_out[row0 + k].code;
            __syncthreads();
            for (int k = tid; k < K; k += NT) {
                const int j = s_run_idx[k];
                const int tokv = s_tok[j], pcode = s_pcode[s_parent[j]];
                grx_row_plan pl;
                pl.code = GRX_RES_EMPTY;
                bool known = false;
                if (pc.eos >= 0 && tokv == pc.eos) {
                    pl.code = GRX_RES_EOS;
                    pl.off = 0;
                    pl.n = 1;
                    known = true;
                } else if (pcode >= 0) {
                    const int child = trie_child(pc.tv, pcode, (int64_t)tokv);
                    if (child >= 0) {
                        const int lo = pc.tv.off[child], hi = pc.tv.off[child + 1];
                        if (hi > lo) {
                            pl.code = child;
                            pl.off = lo;
                            pl.n = hi - lo;
                            known = true;
                        }
                    }
                }
                pl.pad = 0;
                s_rescan[k] = known ? 0 : 1;
                if (known) plan_out[row0 + k] = pl;
            }
            __syncthreads();
            for (int k = wave; k < K; k += NW) {
                if (!s_rescan[k]) continue;
                const grx_row_plan pl = plan_fixed_row(pc, a.run_seq_out + (row0 + k) * L, T_next, lane);
                if (lane == 0) plan_out[row0 + k] = pl;
            }
        }
    }
}
