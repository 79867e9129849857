// LM head fused with the constrained step: logits = hidden [R, H] x weight^T [H, V] are formed tile by
// tile on the matrix cores and never written out.  What leaves a tile is, per row, the (max, sum-of-exp)
// pair over the tile's columns -- the same partial pair the sliced step keeps -- and, for the few column
// tiles at the top of the vocabulary where the semantic-ID tokens live, the fp32 logits themselves.
// A finishing kernel merges a row's pairs into its log-sum-exp, looks the row's trie node up and emits
// the usual candidate row.  bf16 operands, fp32 accumulation (v_mfma_f32_32x32x16_bf16).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "resolve.h"

#include "lse_math.h"

// ---- the tile -----------------------------------------------------------------------------------------
// One 512-thread workgroup forms a 256 (vocabulary) x 256 (rows) tile of logits^T: the vocabulary is the
// M side of the MFMA, the rows are the N side, so a lane holds ONE row's logits for 16 vocabulary entries
// per accumulator tile and the row-wise max / sum-of-exp is plain register arithmetic.  Eight waves as
// 2 (vocabulary) x 4 (rows): 128 x 64 per wave = 4 x 2 accumulator tiles of 32 x 32.
// K runs in steps of 64: the two operand tiles (256 x 64 bf16 = 32 KB each) are copied global -> LDS by
// the DMA path (16 bytes per lane) into one of two buffers while the other one is being multiplied.
// Measured against this structure at 8192 x 152,640 x 2048 (4.49 ms): a ring of four 32-deep steps with
// counted waits and raw barriers (4.98 ms; 4.86 with fragment double-buffering in registers), a persistent
// tile loop that prefetches the next tile's first step under the reduction (4.68 ms, 256 registers), and
// four waves of 128 x 128 with one wave per SIMD and the accumulators in the AGPR half of the file, the
// library kernel's geometry (5.48 ms: nothing hides a lone wave's LDS latency in compiler-scheduled code).
// On top of the fragment pipelining below (4.47 -> 4.35 ms): weight tiles in a ring of three LDS slots, two steps
// ahead (its copies alone 2.54 instead of 3.53 ms; the whole kernel 4.36 ms), tiles staged through registers
// (global -> VGPR -> ds_write, two steps ahead: 4.42 ms), and the four-wave geometry once more, now with
// pipelined fragments and the ring (4.70 ms).  The products alone take 3.57 ms, the copies alone 2.5 - 3.5 ms,
// both at the same 1.8 GHz the library kernel runs at (3.71 ms): together they take 4.35 - 4.4 ms in every one of
// these structures, so neither the copies' latency nor their path into LDS is what the library does better.
// Counters of this kernel against the library's (rocprofv3 --pmc, same product): equal matrix-pipe busy
// cycles, no LDS bank conflicts (library: 318 M), but 35 % of its wave cycles parked at the barrier /
// its DMA wait against the library's 9 %.
#define LH_THREADS 512
#define LH_BM 256   // vocabulary entries per tile
#define LH_BN 256   // rows per tile
#define LH_BK 64
#define LH_TILE_BYTES (256 * LH_BK * 2)          // one operand tile
#define LH_BUF_BYTES (2 * LH_TILE_BYTES)         // weight tile + hidden tile
#define LH_LDS_BYTES (2 * LH_BUF_BYTES)          // two buffers: 128 KB

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_void_t;

struct LmHeadArgs {
    const uint16_t* hidden;  // [R, H] bf16
    int64_t hidden_stride;
    const uint16_t* weight;  // [V, H] bf16
    int64_t weight_stride;
    int R, H, V;
    int n_row_tiles;         // ceil(R / 256)
    int n_voc_tiles;         // ceil(V / 256) = partial slots per row
    float2* partials;        // [n_voc_tiles][R_pad] (max, sum of exp(x - max)) per (vocabulary tile, row)
    int64_t R_pad;           // n_row_tiles * 256
    int tail_begin;          // first column whose logits are kept (a multiple of 256; >= V: none)
    float* tail;             // [R, tail_stride] fp32 logits of columns tail_begin ..
    int64_t tail_stride;
};

// LDS image of an operand tile: 256 rows of 128 bytes (64 bf16), the 16-byte chunk c of row r stored at
// chunk position c ^ ((r >> 1) & 7).  Two rows share a 256-byte bank row, so the 16 rows that a group of
// 16 lanes reads (same chunk, consecutive rows) land on 16 different 16-byte slots: no bank conflicts.
// The DMA writes lane l of an instruction at base + 16 l, i.e. row l / 8, position l % 8 of an 8-row
// slab: the permutation is applied to the SOURCE address (which chunk a lane fetches), and again when
// the fragments are read.
__device__ __forceinline__ int lh_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__global__ __launch_bounds__(LH_THREADS) void lm_head_lse_kernel(LmHeadArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2;  // vocabulary half of the tile
    const int wn = wave & 3;   // row quarter of the tile
    // Tile order: consecutive workgroups share the weight tile (the large operand, streamed once) and walk
    // the row tiles; workgroups are dealt to the 8 XCDs round robin, so every XCD keeps meeting the same
    // n_row_tiles / 8 hidden-state tiles, which stay in its L2.
    const int tv = blockIdx.x / a.n_row_tiles;
    const int tb = blockIdx.x - tv * a.n_row_tiles;
    const int v0 = tv * LH_BM, r0 = tb * LH_BN;

    // staging: every wave copies 4 slabs of 8 rows x 128 bytes of each operand per K step
    const int s_row = lane >> 3, s_pos = lane & 7;
    const uint16_t* w_src[4];
    const uint16_t* h_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + s_row;  // row of the tile
        const int chunk = lh_swz(row, s_pos);          // the chunk this lane fetches (lands at position s_pos)
        int vr = v0 + row;
        vr = vr < a.V ? vr : a.V - 1;                  // rows past the end read the last one; masked below
        int hr = r0 + row;
        hr = hr < a.R ? hr : a.R - 1;
        w_src[i] = a.weight + (int64_t)vr * a.weight_stride + chunk * 8;
        h_src[i] = a.hidden + (int64_t)hr * a.hidden_stride + chunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* base = lds + buf * LH_BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned char* dst_w = base + (wave * 4 + i) * 1024;               // wave-uniform: the DMA adds 16 * lane
            unsigned char* dst_h = base + LH_TILE_BYTES + (wave * 4 + i) * 1024;
            __builtin_amdgcn_global_load_lds((global_void_t*)(w_src[i] + kt * LH_BK), (lds_void_t*)dst_w, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((global_void_t*)(h_src[i] + kt * LH_BK), (lds_void_t*)dst_h, 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    // fragment addresses: lane l holds row (l & 31), k = 8 (l >> 5) + j of a 32 x 16 operand block
    const int f_row = lane & 31, f_half = lane >> 5;
    const int nk = a.H / LH_BK;
    stage(0, 0);
    __syncthreads();  // waits for the DMA (vmcnt) as well
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
#ifndef LH_ABL_NODMA
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
#endif
        const unsigned char* wt = lds + cur * LH_BUF_BYTES;
        const unsigned char* ht = wt + LH_TILE_BYTES;
#ifdef LH_ABL_NOMFMA
        if (a.H < 0)
#endif
        {
            // Four K sub-steps of 16.  Two register sets of fragments: the reads of sub-step s + 1 are issued
            // between the products of sub-step s (one read per product, by the scheduling hints below), so
            // the matrix pipe waits for LDS once per K step, right behind the barrier, instead of once per
            // four products.
            auto read_set = [&](int sub, bf16x8* fa, bf16x8* fb) {
                const int chunk = 2 * sub + f_half;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int row = wm * 128 + m * 32 + f_row;
                    fa[m] = *reinterpret_cast<const bf16x8*>(wt + row * 128 + lh_swz(row, chunk) * 16);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int row = wn * 64 + n * 32 + f_row;
                    fb[n] = *reinterpret_cast<const bf16x8*>(ht + row * 128 + lh_swz(row, chunk) * 16);
                }
            };
            auto multiply = [&](const bf16x8* fa, const bf16x8* fb) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
            };
            auto interleave = [&]() {  // 8 products, 6 fragment reads: product, read, product, read, ...
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            };
            bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
            read_set(0, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            read_set(1, fa1, fb1);
            multiply(fa0, fb0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            read_set(2, fa0, fb0);
            multiply(fa1, fb1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            read_set(3, fa1, fb1);
            multiply(fa0, fb0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            multiply(fa1, fb1);
        }
        __syncthreads();  // the next tile has landed, and everybody is done with this one
    }

    // ---- epilogue: element e of acc[m][n] in lane l is
    //   vocabulary entry v0 + 128 wm + 32 m + (e & 3) + 8 (e >> 2) + 4 (l >> 5),  row r0 + 64 wn + 32 n + (l & 31)
    float2* const red = reinterpret_cast<float2*>(lds);  // [2 vocabulary halves][256 rows]
    const bool keep = v0 >= a.tail_begin;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row_in_tile = wn * 64 + n * 32 + f_row;
        const int r = r0 + row_in_tile;
        float mx = -INFINITY;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int v = v0 + wm * 128 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * f_half;
                if (v >= a.V) acc[m][n][e] = -INFINITY;
                mx = fmaxf(mx, acc[m][n][e]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // the other half of the lanes holds the same row
        const float c = (mx == -INFINITY) ? 0.f : -mx * kLog2e;
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += fast_exp2(fmaf(acc[m][n][e], kLog2e, c));
        sum += __shfl_xor(sum, 32, 64);
        if (f_half == 0) red[wm * 256 + row_in_tile] = make_float2(mx, sum);
        if (keep && r < a.R) {
            // fp32 logits of the kept columns: four consecutive vocabulary entries per store
            float* trow = a.tail + (int64_t)r * a.tail_stride + (v0 - a.tail_begin) + wm * 128 + 4 * f_half;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 q = make_float4(acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2],
                                                 acc[m][n][4 * g + 3]);
                    *reinterpret_cast<float4*>(trow + m * 32 + 8 * g) = q;
                }
        }
    }
    __syncthreads();
    if (tid < 256) {
        const float2 p0 = red[tid], p1 = red[256 + tid];
        const float m = fmaxf(p0.x, p1.x);
        float s = 0.f;
        if (m != -INFINITY) s = p0.y * fast_exp2((p0.x - m) * kLog2e) + p1.y * fast_exp2((p1.x - m) * kLog2e);
        a.partials[(int64_t)tv * a.R_pad + r0 + tid] = make_float2(m, s);
    }
}

// ---- finishing kernel ----------------------------------------------------------------------------------
// 16 rows per 256-thread workgroup.  Thread (row i, part j) merges the pairs of the vocabulary tiles
// j, j + 16, ... of its row (for one tile, 16 rows are 128 contiguous bytes); the 16 parts of a row meet
// through shuffles.  Then every wave takes four of the rows in turn: trie node (walked here in the
// count-based mode, planned ahead in the fixed-hint mode), the logits of the node's tokens -- from the
// kept columns, or, for a token below them, as a dot product of the row's hidden state with that
// token's weight row -- and the candidate row in the format of the streaming kernel.
struct LmFinishArgs {
    LmHeadArgs g;
    const int32_t* tok;
    const int32_t* off;
    const grx_row_plan* plan;
    grx_state_words* st;
    int walk, bookkeep;
    PlanCtx pc;
    const int64_t* ids;
    int64_t ids_stride;
    int T;
    int eos;
    grx_cand* cand;
    int64_t cand_stride;
    int cmax;
};

__global__ __launch_bounds__(256) void lm_head_finish_kernel(LmFinishArgs a) {
    __shared__ float s_lse_m[16], s_lse_s[16];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int rbase = blockIdx.x * 16;
    {
        const int i = tid & 15, j = tid >> 4;
        float m = -INFINITY, s = 0.f;
        for (int t = j; t < a.g.n_voc_tiles; t += 16) {
            const float2 p = a.g.partials[(int64_t)t * a.g.R_pad + rbase + i];  // rows beyond R hold the clamped row's pairs
            const float mn = fmaxf(m, p.x);
            if (mn != -INFINITY) s = s * fast_exp2((m - mn) * kLog2e) + p.y * fast_exp2((p.x - mn) * kLog2e);
            m = mn;
        }
        // parts j, j ^ 1, j ^ 2, j ^ 4 ... live 16 threads apart: four of them inside a wave, the rest across waves
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            const float om = __shfl_xor(m, o, 64), os = __shfl_xor(s, o, 64);
            const float mn = fmaxf(m, om);
            s = (mn == -INFINITY) ? 0.f : s * fast_exp2((m - mn) * kLog2e) + os * fast_exp2((om - mn) * kLog2e);
            m = mn;
        }
        __shared__ float s_pm[4][16], s_ps[4][16];
        if (lane < 16) {
            s_pm[wave][lane] = m;
            s_ps[wave][lane] = s;
        }
        __syncthreads();
        if (tid < 16) {
            float M = s_pm[0][tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) M = fmaxf(M, s_pm[w][tid]);
            float S = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (s_pm[w][tid] != -INFINITY) S += s_ps[w][tid] * fast_exp2((s_pm[w][tid] - M) * kLog2e);
            s_lse_m[tid] = M;
            s_lse_s[tid] = S;
        }
        __syncthreads();
    }
    if (a.bookkeep && blockIdx.x == 0 && tid == 0) plan_count_bookkeeping(a.pc);
    for (int q = 0; q < 4; ++q) {
        const int i = wave * 4 + q;
        const int r = rbase + i;
        if (r >= a.g.R) break;
        int code, c_off, c_n;
        if (a.walk) {
            code = count_row_node(a.pc, a.ids + (int64_t)r * a.ids_stride, a.T, r);
            c_off = code >= 0 ? a.off[code] : 0;
            c_n = code >= 0 ? a.off[code + 1] - c_off : 0;
        } else {
            const grx_row_plan pl = a.plan[r];
            code = pl.code;
            c_off = pl.off;
            c_n = pl.n;
        }
        if (code == GRX_RES_EOS) c_n = 1;
        if (code < 0 && code != GRX_RES_EOS && lane == 0)
            report_row_error(a.st, r, code == GRX_RES_NOMATCH ? GRX_ROWERR_NOMATCH : GRX_ROWERR_EMPTY);
        if (c_n > a.cmax) {
            if (lane == 0) report_row_error(a.st, r, GRX_ROWERR_OVERFLOW);
            c_n = 0;
        }
        const float lse = s_lse_m[i] + logf(s_lse_s[i]);
        grx_cand* __restrict__ crow = a.cand + (int64_t)r * a.cand_stride;
        const int n_line = ((c_n + 1 + 15) / 16) * 16;
        const float* trow = a.g.tail + (int64_t)r * a.g.tail_stride;
        for (int e = lane; e < n_line; e += 64) {
            grx_cand v = {0, 0};
            if (e == 0) {
                v.x = c_n;
                v.y = __float_as_int(lse);
            } else if (e - 1 < c_n) {
                const int tokv = code == GRX_RES_EOS ? a.eos : a.tok[c_off + e - 1];
                float x;
                if (tokv >= a.g.tail_begin) {
                    x = trow[tokv - a.g.tail_begin];
                } else {
                    // a token below the kept columns (rare: an EOS id or prefix token far from the SID tokens)
                    const uint16_t* hrow = a.g.hidden + (int64_t)r * a.g.hidden_stride;
                    const uint16_t* wrow = a.g.weight + (int64_t)tokv * a.g.weight_stride;
                    x = 0.f;
                    for (int k = 0; k < a.g.H; ++k)
                        x = fmaf(__uint_as_float((uint32_t)hrow[k] << 16), __uint_as_float((uint32_t)wrow[k] << 16), x);
                }
                v.x = tokv;
                v.y = __float_as_int(x - lse);
            }
            crow[e] = v;
        }
    }
}

// ---- host ----------------------------------------------------------------------------------------------
#define GRX_LM_TAIL_MAX 4096  // kept columns reach at most this far below the top of the vocabulary

static int lm_tail_begin(const grx_trie* trie, int V) {
    // the lowest edge token within GRX_LM_TAIL_MAX of the top of the vocabulary, rounded down to a tile
    int begin = V;
    for (int i = trie->n_tok_sorted - 1; i >= 0; --i) {
        const int t = trie->tok_sorted[i];
        if (t >= V) continue;
        if (t < V - GRX_LM_TAIL_MAX) break;
        begin = t;
    }
    if (begin >= V) return ((V + LH_BM - 1) / LH_BM) * LH_BM;  // nothing kept
    return (begin / LH_BM) * LH_BM;
}

struct LmLayout {
    int n_row_tiles, n_voc_tiles, tail_begin;
    int64_t R_pad, tail_stride;
    size_t partial_bytes, tail_bytes;
};

static LmLayout lm_layout(const grx_trie* trie, int R, int V) {
    LmLayout L;
    L.n_row_tiles = (R + LH_BN - 1) / LH_BN;
    L.n_voc_tiles = (V + LH_BM - 1) / LH_BM;
    L.R_pad = (int64_t)L.n_row_tiles * LH_BN;
    L.tail_begin = lm_tail_begin(trie, V);
    const int64_t v_pad = (int64_t)L.n_voc_tiles * LH_BM;
    L.tail_stride = v_pad > L.tail_begin ? v_pad - L.tail_begin : 0;
    L.partial_bytes = (size_t)L.n_voc_tiles * (size_t)L.R_pad * sizeof(float2);
    L.tail_bytes = (size_t)R * (size_t)L.tail_stride * sizeof(float);
    return L;
}

extern "C" size_t grx_lm_head_scratch_bytes(const grx_trie* trie, int32_t R, int32_t V) {
    if (trie == nullptr || R <= 0 || V <= 0) return 0;
    const LmLayout L = lm_layout(trie, R, V);
    return L.partial_bytes + L.tail_bytes + 256;
}

extern "C" int grx_lm_head_candidates(const grx_trie* trie, grx_state* state, const grx_rows* rows,
                                      const grx_constraint* con, const grx_lm_head* head, void* scratch,
                                      size_t scratch_bytes, grx_cand* cand, int64_t cand_stride, int32_t cmax,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    const char* who = "grx_lm_head_candidates";
    hipStream_t stream = (hipStream_t)stream_;
    GRX_REQUIRE(rows != nullptr && head != nullptr, GRX_E_INVALID, "%s: null rows / head", who);
    grx_rows probe = *rows;
    probe.scores = head->hidden;  // the shared argument checks want a non-null scores pointer
    probe.scores_stride = rows->V;
    int rc = grx_check_step_args(who, trie, state, &probe, con, workspace, workspace_bytes);
    if (rc != GRX_OK) return rc;
    GRX_REQUIRE(rows->dtype == GRX_BF16, GRX_E_UNSUPPORTED, "%s: bf16 hidden states and weights only", who);
    GRX_REQUIRE(head->H >= 64 && head->H % 64 == 0, GRX_E_UNSUPPORTED, "%s: H must be a multiple of 64", who);
    GRX_REQUIRE(head->hidden_stride >= head->H && head->weight_stride >= head->H && head->hidden_stride % 8 == 0 &&
                    head->weight_stride % 8 == 0,
                GRX_E_INVALID, "%s: operand rows must be 16-byte aligned and at least H long", who);
    GRX_REQUIRE(cmax >= 1 && cand_stride >= grx_candidate_stride(cmax) && cand_stride % 16 == 0, GRX_E_INVALID,
                "%s: bad candidate buffer shape", who);
    const bool penalise = con->repetition_penalty != 0.f && con->repetition_penalty != 1.f;
    GRX_REQUIRE(!penalise && !con->assume_normalized && con->keep_top == 0, GRX_E_UNSUPPORTED,
                "%s: no repetition penalty, assume_normalized or keep_top in this form", who);
    GRX_REQUIRE(con->eos_token_id < rows->V, GRX_E_INVALID, "%s: eos_token_id %d outside V=%d", who, con->eos_token_id,
                rows->V);
    if (rows->R > 0) {
        GRX_REQUIRE(head->hidden != nullptr && head->weight != nullptr && cand != nullptr && scratch != nullptr,
                    GRX_E_INVALID, "%s: null buffer", who);
        GRX_REQUIRE((((uintptr_t)head->hidden | (uintptr_t)head->weight | (uintptr_t)scratch) & 15u) == 0 &&
                        ((uintptr_t)cand & 127u) == 0,
                    GRX_E_INVALID, "%s: misaligned buffer", who);
        GRX_REQUIRE(scratch_bytes >= grx_lm_head_scratch_bytes(trie, rows->R, rows->V), GRX_E_WORKSPACE,
                    "%s: scratch of %zu bytes, need %zu", who, scratch_bytes,
                    grx_lm_head_scratch_bytes(trie, rows->R, rows->V));
    }
    LmFinishArgs f;
    memset(&f, 0, sizeof(f));
    grx_row_plan* plan = rows->R > 0
                             ? reinterpret_cast<grx_row_plan*>(reinterpret_cast<char*>(workspace) + GRX_WS_PLAN_OFFSET)
                             : nullptr;
    rc = grx_plan_rows(trie, state, rows, con, plan, stream, &f.pc, &f.bookkeep);
    if (rc != GRX_OK) return rc;
    if (rows->R == 0) return GRX_OK;
    const LmLayout L = lm_layout(trie, rows->R, rows->V);
    LmHeadArgs& g = f.g;
    g.hidden = reinterpret_cast<const uint16_t*>(head->hidden);
    g.hidden_stride = head->hidden_stride;
    g.weight = reinterpret_cast<const uint16_t*>(head->weight);
    g.weight_stride = head->weight_stride;
    g.R = rows->R;
    g.H = head->H;
    g.V = rows->V;
    g.n_row_tiles = L.n_row_tiles;
    g.n_voc_tiles = L.n_voc_tiles;
    g.partials = reinterpret_cast<float2*>(scratch);
    g.R_pad = L.R_pad;
    g.tail_begin = L.tail_begin;
    g.tail = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + ((L.partial_bytes + 255) / 256) * 256);
    g.tail_stride = L.tail_stride;
    static bool lds_opted = false;
    if (!lds_opted) {
        GRX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lm_head_lse_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LH_LDS_BYTES));
        lds_opted = true;
    }
    GRX_REQUIRE((int64_t)L.n_row_tiles * L.n_voc_tiles <= 0x7FFFFFFF, GRX_E_UNSUPPORTED, "%s: too many tiles", who);
    const int n_tiles = L.n_row_tiles * L.n_voc_tiles;
    hipLaunchKernelGGL(lm_head_lse_kernel, dim3((unsigned)n_tiles), dim3(LH_THREADS), LH_LDS_BYTES, stream, g);
    GRX_CHECK_HIP(hipGetLastError());
    f.tok = trie->child_tok;
    f.off = trie->child_off;
    f.plan = plan;
    f.st = state ? state->dev : nullptr;
    f.walk = con->mode == GRX_MODE_COUNT ? 1 : 0;
    f.ids = rows->input_ids;
    f.ids_stride = rows->ids_stride;
    f.T = rows->T;
    f.eos = con->eos_token_id;
    f.cand = cand;
    f.cand_stride = cand_stride;
    f.cmax = cmax;
    hipLaunchKernelGGL(lm_head_finish_kernel, dim3((unsigned)((rows->R + 15) / 16)), dim3(256), 0, stream, f);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}
