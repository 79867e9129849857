// Reward, ranking-metric and advantage kernels: integer comparisons on SID code
// rows and small per-group fp32 statistics.  Bytes are negligible next to the
// logit stream; these kernels exist so that rollout scoring stays on the device
// and in stream order behind the beam step.
#include <math.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.h"

#define SC_THREADS 256

// ------------------------------------------------------------- rewards ---
// One thread per row; a workgroup holds whole prompt groups (K consecutive rows)
// so that "does the group contain an exact row" is an LDS lookup.
struct RewardArgs {
    const void* pred; const int32_t* pred_len; int Lp; int64_t pred_stride;
    const void* gt; const int32_t* gt_len; int Lg; int64_t gt_stride;
    const int32_t* table; int table_size;
    int R, K, rows_per_block, prefix_normalize;
    const float* ndcg_w;
    float* rule_out; float* prefix_out; float* ndcg_out; int out_stride;
    // Optional second half (grx_rollout_score_tokens): the weighted reward of every row and its
    // group-relative advantage, from the values just computed -- what grx_group_advantage would
    // read back from the reward matrix.  n_cols == 0: off.
    int n_cols; int col_kind[3];     // kind of reward column c: 0 rule, 1 prefix, 2 ndcg
    const float* adv_w;              // device [n_cols] reward weights
    int scale; float eps;            // 0: none, 1: group std
    float* rewards; float* adv; float* gmean; float* gstd;
};

// WIDE: rows are int64 token ids mapped through the code table; otherwise int32 codes.
template <bool WIDE>
__device__ __forceinline__ int sid_code(const RewardArgs& a, const void* row, int i) {
    if (!WIDE) return static_cast<const int32_t*>(row)[i];
    const int64_t v = static_cast<const int64_t*>(row)[i];
    return (v >= 0 && v < a.table_size) ? a.table[v] : -1;
}

template <bool WIDE>
__global__ __launch_bounds__(SC_THREADS) void sid_rewards_kernel(RewardArgs a) {
    __shared__ unsigned char s_exact[SC_THREADS];
    const int t = threadIdx.x;
    const int r = blockIdx.x * a.rows_per_block + t;
    const bool active = t < a.rows_per_block && r < a.R;
    bool exact = false;
    int matched = 0, ng = 0;
    if (active) {
        const size_t esz = WIDE ? sizeof(int64_t) : sizeof(int32_t);
        const void* p = static_cast<const char*>(a.pred) + (int64_t)r * a.pred_stride * esz;
        const void* q = static_cast<const char*>(a.gt) + (int64_t)r * a.gt_stride * esz;
        const int lp = a.pred_len ? a.pred_len[r] : a.Lp;
        const int lg = a.gt_len ? a.gt_len[r] : a.Lg;
        int i = 0, j = 0, np = 0;
        bool open = true;  // still inside the common prefix
        while (true) {
            int cp = -1, cg = -1;
            // token rows: entries without a code (EOS, padding) are skipped
            while (i < lp && (cp = sid_code<WIDE>(a, p, i)) < 0 && WIDE) ++i;
            while (j < lg && (cg = sid_code<WIDE>(a, q, j)) < 0 && WIDE) ++j;
            const bool hp = i < lp, hg = j < lg;
            if (!hp && !hg) break;
            np += hp ? 1 : 0;
            ng += hg ? 1 : 0;
            if (hp && hg && open && cp == cg) ++matched; else open = false;
            i += hp ? 1 : 0;
            j += hg ? 1 : 0;
        }
        exact = (np == ng) && (matched == np);
    }
    s_exact[t] = exact ? 1 : 0;
    __syncthreads();
    const int g0 = (t / a.K) * a.K;                  // first row of the group inside the block
    float v_rule = 0.0f, v_prefix = 0.0f, v_ndcg = 0.0f;
    if (active) {
        const int64_t o = (int64_t)r * a.out_stride;
        v_rule = exact ? 1.0f : 0.0f;
        if (a.rule_out) a.rule_out[o] = v_rule;
        if (ng > 0) v_prefix = a.prefix_normalize ? (float)((double)matched / (double)ng) : (float)matched;
        if (a.prefix_out) a.prefix_out[o] = v_prefix;
        if (a.ndcg_out) {
            const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block + g0;
            if (r0 + a.K <= a.R) {                  // rows of an incomplete trailing group get no value
                bool any = false;
                for (int k = 0; k < a.K; ++k) any = any || s_exact[g0 + k];
                v_ndcg = (any && !exact) ? a.ndcg_w[t - g0] : 0.0f;
                a.ndcg_out[o] = v_ndcg;
            }
        }
    }
    if (a.n_cols == 0) return;  // uniform
    // weighted reward (nansum over the columns in their order) and the group statistics, as
    // group_advantage_kernel forms them from the stored matrix
    __shared__ float s_rew[SC_THREADS];
    float rew = 0.0f;
    for (int c = 0; c < a.n_cols; ++c) {
        const float val = a.col_kind[c] == 0 ? v_rule : (a.col_kind[c] == 1 ? v_prefix : v_ndcg);
        const float v = val * a.adv_w[c];
        rew += (v != v) ? 0.0f : v;
    }
    s_rew[t] = active ? rew : 0.0f;
    __syncthreads();
    if (!active) return;
    double sum = 0.0;
    for (int k = 0; k < a.K; ++k) sum += (double)s_rew[g0 + k];
    const double mean_d = sum / (double)a.K;
    double q = 0.0;
    for (int k = 0; k < a.K; ++k) {
        const double dlt = (double)s_rew[g0 + k] - mean_d;
        q += dlt * dlt;
    }
    const float mean = (float)mean_d;
    const float sd = (float)sqrt(q / (double)(a.K - 1));
    float adv = rew - mean;
    if (a.scale != 0) adv = adv / (sd + a.eps);
    if (a.rewards) a.rewards[r] = rew;
    if (a.adv) a.adv[r] = adv;
    if (a.gmean) a.gmean[r] = mean;
    if (a.gstd) a.gstd[r] = sd;
}

static int launch_rewards(const char* who, RewardArgs& a, bool wide, void* stream) {
    GRX_REQUIRE(a.R >= 0 && a.K >= 1 && a.Lp >= 0 && a.Lg >= 0 && a.out_stride >= 1, GRX_E_INVALID, "%s: bad shape",
                who);
    GRX_REQUIRE(a.K <= SC_THREADS, GRX_E_UNSUPPORTED, "%s: at most %d rows per group", who, SC_THREADS);
    if (a.R == 0) return GRX_OK;
    GRX_REQUIRE((a.Lp == 0 || a.pred) && (a.Lg == 0 || a.gt), GRX_E_INVALID, "%s: null input", who);
    GRX_REQUIRE(a.pred_stride >= a.Lp && a.gt_stride >= a.Lg, GRX_E_INVALID, "%s: row stride smaller than the row",
                who);
    GRX_REQUIRE(a.ndcg_out == nullptr || a.ndcg_w != nullptr, GRX_E_INVALID, "%s: ndcg_out needs ndcg_weights", who);
    a.rows_per_block = (SC_THREADS / a.K) * a.K;
    const int blocks = (a.R + a.rows_per_block - 1) / a.rows_per_block;
    if (wide)
        hipLaunchKernelGGL(sid_rewards_kernel<true>, dim3(blocks), dim3(SC_THREADS), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(sid_rewards_kernel<false>, dim3(blocks), dim3(SC_THREADS), 0, (hipStream_t)stream, a);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

extern "C" int grx_sid_rewards(const int32_t* pred, const int32_t* pred_len, int32_t Lp, const int32_t* gt,
                               const int32_t* gt_len, int32_t Lg, int32_t R, int32_t K, int32_t prefix_normalize,
                               const float* ndcg_weights, float* rule_out, float* prefix_out, float* ndcg_out,
                               int32_t out_stride, void* stream) {
    RewardArgs a;
    memset(&a, 0, sizeof(a));
    a.pred = pred; a.pred_len = pred_len; a.Lp = Lp; a.pred_stride = Lp;
    a.gt = gt; a.gt_len = gt_len; a.Lg = Lg; a.gt_stride = Lg;
    a.table = nullptr; a.table_size = 0;
    a.R = R; a.K = K; a.prefix_normalize = prefix_normalize; a.ndcg_w = ndcg_weights;
    a.rule_out = rule_out; a.prefix_out = prefix_out; a.ndcg_out = ndcg_out; a.out_stride = out_stride;
    return launch_rewards("grx_sid_rewards", a, false, stream);
}

extern "C" int grx_sid_rewards_tokens(const int64_t* pred_tok, int64_t pred_stride, int32_t Lp, const int64_t* gt_tok,
                                      int64_t gt_stride, int32_t Lg, const int32_t* code_table, int32_t table_size,
                                      int32_t R, int32_t K, int32_t prefix_normalize, const float* ndcg_weights,
                                      float* rule_out, float* prefix_out, float* ndcg_out, int32_t out_stride,
                                      void* stream) {
    GRX_REQUIRE(code_table != nullptr && table_size > 0, GRX_E_INVALID, "grx_sid_rewards_tokens: empty code table");
    RewardArgs a;
    memset(&a, 0, sizeof(a));
    a.pred = pred_tok; a.pred_len = nullptr; a.Lp = Lp; a.pred_stride = pred_stride;
    a.gt = gt_tok; a.gt_len = nullptr; a.Lg = Lg; a.gt_stride = gt_stride;
    a.table = code_table; a.table_size = table_size;
    a.R = R; a.K = K; a.prefix_normalize = prefix_normalize; a.ndcg_w = ndcg_weights;
    a.rule_out = rule_out; a.prefix_out = prefix_out; a.ndcg_out = ndcg_out; a.out_stride = out_stride;
    return launch_rewards("grx_sid_rewards_tokens", a, true, stream);
}

extern "C" int grx_rollout_score_tokens(const int64_t* pred_tok, int64_t pred_stride, int32_t Lp, const int64_t* gt_tok,
                                        int64_t gt_stride, int32_t Lg, const int32_t* code_table, int32_t table_size,
                                        int32_t R, int32_t K, int32_t prefix_normalize, const float* ndcg_weights,
                                        float* rewards_per_func, int32_t n_funcs, const int32_t* func_kinds,
                                        const float* reward_weights, int32_t scale, float eps, float* rewards,
                                        float* advantages, float* group_mean, float* group_std, void* stream) {
    const char* who = "grx_rollout_score_tokens";
    GRX_REQUIRE(code_table != nullptr && table_size > 0, GRX_E_INVALID, "%s: empty code table", who);
    GRX_REQUIRE(n_funcs >= 1 && n_funcs <= 3 && func_kinds != nullptr, GRX_E_INVALID, "%s: 1 to 3 reward columns", who);
    GRX_REQUIRE(scale == 0 || scale == 1, GRX_E_UNSUPPORTED,
                "%s: scale must be 0 (none) or 1 (group); the batch std needs every rank's rows (grx_group_advantage)",
                who);
    GRX_REQUIRE(K >= 1 && R >= 0 && R % K == 0, GRX_E_INVALID, "%s: %d rows do not split into groups of %d", who, R, K);
    GRX_REQUIRE(R == 0 || (rewards_per_func != nullptr && reward_weights != nullptr), GRX_E_INVALID, "%s: null buffer",
                who);
    RewardArgs a;
    memset(&a, 0, sizeof(a));
    a.pred = pred_tok; a.Lp = Lp; a.pred_stride = pred_stride;
    a.gt = gt_tok; a.Lg = Lg; a.gt_stride = gt_stride;
    a.table = code_table; a.table_size = table_size;
    a.R = R; a.K = K; a.prefix_normalize = prefix_normalize; a.ndcg_w = ndcg_weights;
    a.out_stride = n_funcs;
    int seen = 0;
    for (int c = 0; c < n_funcs; ++c) {
        const int kind = func_kinds[c];
        GRX_REQUIRE(kind >= 0 && kind <= 2 && !(seen & (1 << kind)), GRX_E_INVALID,
                    "%s: reward kinds must be distinct values of 0 (rule), 1 (prefix), 2 (ndcg)", who);
        seen |= 1 << kind;
        a.col_kind[c] = kind;
        (kind == 0 ? a.rule_out : (kind == 1 ? a.prefix_out : a.ndcg_out)) = rewards_per_func + c;
    }
    a.n_cols = n_funcs;
    a.adv_w = reward_weights;
    a.scale = scale; a.eps = eps;
    a.rewards = rewards; a.adv = advantages; a.gmean = group_mean; a.gstd = group_std;
    return launch_rewards(who, a, true, stream);
}

// ------------------------------------------------------------ hit ranks ---
__global__ void hit_ranks_kernel(const int32_t* __restrict__ beams, const int32_t* __restrict__ target, int n,
                                 int n_beam, int L, int32_t* __restrict__ first_hit, int32_t* __restrict__ rank_hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t* t = target + (int64_t)i * L;
    int rank = n_beam;
    for (int bidx = 0; bidx < n_beam; ++bidx) {
        const int32_t* p = beams + ((int64_t)i * n_beam + bidx) * L;
        bool eq = true;
        for (int l = 0; eq && l < L; ++l) eq = p[l] == t[l];
        if (eq) {
            rank = bidx;
            break;
        }
    }
    if (first_hit) first_hit[i] = rank;
    if (rank_hist) atomicAdd(&rank_hist[rank], 1);
}

extern "C" int grx_hit_ranks(const int32_t* beams, const int32_t* target, int32_t n, int32_t n_beam, int32_t L,
                             int32_t* first_hit, int32_t* rank_hist, void* stream) {
    const char* who = "grx_hit_ranks";
    GRX_REQUIRE(n >= 0 && n_beam >= 0 && L >= 0, GRX_E_INVALID, "%s: bad shape", who);
    if (n == 0) return GRX_OK;
    GRX_REQUIRE((L == 0 || n_beam == 0 || beams) && (L == 0 || target), GRX_E_INVALID, "%s: null input", who);
    hipLaunchKernelGGL(hit_ranks_kernel, dim3((n + SC_THREADS - 1) / SC_THREADS), dim3(SC_THREADS), 0,
                       (hipStream_t)stream, beams, target, n, n_beam, L, first_hit, rank_hist);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

// ------------------------------------------------------------ advantage ---
__device__ __forceinline__ float weighted_nansum(const float* __restrict__ rpf, const float* __restrict__ w, int r,
                                                 int n_funcs) {
    float acc = 0.0f;
    for (int f = 0; f < n_funcs; ++f) {
        const float v = rpf[(int64_t)r * n_funcs + f] * w[f];
        acc += (v != v) ? 0.0f : v;
    }
    return acc;
}

// Unbiased std of all R weighted rewards (scale "batch"), computed ONCE by a single workgroup in two
// passes over the L2-resident rewards (fp64 sums) and left in stats[0] for the kernel below.
#define BS_STAT_THREADS 1024
__global__ __launch_bounds__(BS_STAT_THREADS) void batch_std_kernel(const float* __restrict__ rpf,
                                                                    const float* __restrict__ w, int R, int n_funcs,
                                                                    float* __restrict__ stats) {
    __shared__ double s_tmp[BS_STAT_THREADS / GRX_WAVE];
    const int lane = threadIdx.x & (GRX_WAVE - 1), wave = threadIdx.x / GRX_WAVE;
    auto block_sum = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, GRX_WAVE);
        __syncthreads();
        if (lane == 0) s_tmp[wave] = v;
        __syncthreads();
        double t = 0.0;
        for (int i = 0; i < BS_STAT_THREADS / GRX_WAVE; ++i) t += s_tmp[i];
        return t;
    };
    double s = 0.0;
    for (int r = threadIdx.x; r < R; r += BS_STAT_THREADS) s += (double)weighted_nansum(rpf, w, r, n_funcs);
    const double mean = block_sum(s) / (double)R;
    double q = 0.0;
    for (int r = threadIdx.x; r < R; r += BS_STAT_THREADS) {
        const double dlt = (double)weighted_nansum(rpf, w, r, n_funcs) - mean;
        q += dlt * dlt;
    }
    const double m2 = block_sum(q);
    if (threadIdx.x == 0) stats[0] = (float)sqrt(m2 / (double)(R - 1));  // R == 1: 0 / 0 = nan, as torch.std
}

// One thread per row; a workgroup holds whole groups.  scale: 0 none, 1 group, 2 batch (std from
// batch_stats[0]).
__global__ __launch_bounds__(SC_THREADS) void group_advantage_kernel(const float* __restrict__ rpf,
                                                                      const float* __restrict__ w, int R, int n_funcs,
                                                                      int K, int rows_per_block, int scale, float eps,
                                                                      const float* __restrict__ batch_stats,
                                                                      float* __restrict__ rewards,
                                                                      float* __restrict__ adv,
                                                                      float* __restrict__ gmean,
                                                                      float* __restrict__ gstd) {
    __shared__ float s_rew[SC_THREADS];
    const int t = threadIdx.x;
    const int r = blockIdx.x * rows_per_block + t;
    const bool active = t < rows_per_block && r < R;
    const float rew = active ? weighted_nansum(rpf, w, r, n_funcs) : 0.0f;
    s_rew[t] = rew;
    __syncthreads();
    if (!active) return;
    const int g0 = (t / K) * K;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)s_rew[g0 + k];
    const double mean_d = s / (double)K;
    double q = 0.0;
    for (int k = 0; k < K; ++k) {
        const double dlt = (double)s_rew[g0 + k] - mean_d;
        q += dlt * dlt;
    }
    const float mean = (float)mean_d;
    const float gsd = (float)sqrt(q / (double)(K - 1));
    const float sd = (scale == 2) ? batch_stats[0] : gsd;
    float a = rew - mean;
    if (scale != 0) a = a / (sd + eps);
    if (rewards) rewards[r] = rew;
    if (adv) adv[r] = a;
    if (gmean) gmean[r] = mean;
    if (gstd) gstd[r] = sd;
}

// Scratch words for scale "batch": 64 slots of 16 bytes per GPU, handed out round robin, so that calls
// in flight on different streams of one GPU do not share a word (the 65th call reuses the first slot).
#define BATCH_STAT_SLOTS 64
#define BATCH_STAT_DEVICES 64
static __device__ __align__(16) float g_batch_stats[BATCH_STAT_SLOTS * 4];
static float* g_batch_stats_ptr[BATCH_STAT_DEVICES];
static std::atomic<unsigned> g_batch_stats_next{0};
static std::mutex g_batch_stats_mutex;

static int batch_stats_slot(const char* who, float** out) {
    int dev = -1;
    GRX_CHECK_HIP(hipGetDevice(&dev));
    GRX_REQUIRE(dev >= 0 && dev < BATCH_STAT_DEVICES, GRX_E_UNSUPPORTED, "%s: device index %d not supported", who, dev);
    {
        std::lock_guard<std::mutex> guard(g_batch_stats_mutex);
        if (g_batch_stats_ptr[dev] == nullptr) {
            void* p = nullptr;  // the array's address on THIS device (its name is only a handle in host code)
            GRX_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_batch_stats)));
            GRX_REQUIRE(p != nullptr, GRX_E_HIP, "%s: no device address for the batch statistics", who);
            g_batch_stats_ptr[dev] = reinterpret_cast<float*>(p);
        }
    }
    *out = g_batch_stats_ptr[dev] + 4 * (g_batch_stats_next.fetch_add(1) % BATCH_STAT_SLOTS);
    return GRX_OK;
}

extern "C" int grx_group_advantage(const float* rewards_per_func, const float* weights, int32_t R, int32_t n_funcs,
                                   int32_t K, int32_t scale, float eps, float* rewards, float* advantages,
                                   float* group_mean, float* group_std, void* stream_) {
    const char* who = "grx_group_advantage";
    hipStream_t stream = (hipStream_t)stream_;
    GRX_REQUIRE(R >= 0 && n_funcs >= 1 && K >= 1, GRX_E_INVALID, "%s: bad shape", who);
    GRX_REQUIRE(K <= SC_THREADS, GRX_E_UNSUPPORTED, "%s: at most %d rows per group", who, SC_THREADS);
    GRX_REQUIRE(R % K == 0, GRX_E_INVALID, "%s: %d rows do not split into groups of %d", who, R, K);
    GRX_REQUIRE(scale >= 0 && scale <= 2, GRX_E_INVALID, "%s: bad scale %d", who, scale);
    if (R == 0) return GRX_OK;
    GRX_REQUIRE(rewards_per_func && weights, GRX_E_INVALID, "%s: null input", who);
    float* stats = nullptr;
    if (scale == 2) {
        // the batch std travels from the reducing kernel to the normalising one through a word of a device
        // array that is part of the library image (nothing is allocated: the call can be captured in a graph)
        const int rc = batch_stats_slot(who, &stats);
        if (rc != GRX_OK) return rc;
        hipLaunchKernelGGL(batch_std_kernel, dim3(1), dim3(BS_STAT_THREADS), 0, stream, rewards_per_func, weights, R,
                           n_funcs, stats);
        GRX_CHECK_HIP(hipGetLastError());
    }
    const int rows_per_block = (SC_THREADS / K) * K;
    hipLaunchKernelGGL(group_advantage_kernel, dim3((R + rows_per_block - 1) / rows_per_block), dim3(SC_THREADS), 0,
                       stream, rewards_per_func, weights, R, n_funcs, K, rows_per_block, scale, eps, stats, rewards,
                       advantages, group_mean, group_std);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

// ------------------------------------------------- token-wise normalise ---
// One thread per (group, token position); consecutive threads take consecutive
// positions so every load is coalesced along the token axis.
__global__ void group_normalize_tokenwise_kernel(const float* __restrict__ values, const float* __restrict__ mask,
                                                 int G, int Tc, int K, float eps, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)G * Tc) return;
    const int g = (int)(idx / Tc), t = (int)(idx % Tc);
    const int64_t base = (int64_t)g * K * Tc + t;
    float cnt = 0.0f, sum = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float m = mask[base + (int64_t)k * Tc];
        cnt += m;
        sum += values[base + (int64_t)k * Tc] * m;
    }
    const float denom = fmaxf(cnt, 1.0f);
    const float mean = sum / denom;
    float sq = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float c = (values[base + (int64_t)k * Tc] - mean) * mask[base + (int64_t)k * Tc];
        sq += c * c;
    }
    const float sd = sqrtf(sq / denom + eps);
    for (int k = 0; k < K; ++k) {
        const int64_t at = base + (int64_t)k * Tc;
        out[at] = ((values[at] - mean) / sd) * mask[at];
    }
}

extern "C" int grx_group_normalize_tokenwise(const float* values, const float* mask, int32_t R, int32_t Tc, int32_t K,
                                             float eps, float* out, void* stream) {
    const char* who = "grx_group_normalize_tokenwise";
    GRX_REQUIRE(R >= 0 && Tc >= 0 && K >= 1, GRX_E_INVALID, "%s: bad shape", who);
    GRX_REQUIRE(R % K == 0, GRX_E_INVALID, "%s: %d rows do not split into groups of %d", who, R, K);
    if (R == 0 || Tc == 0) return GRX_OK;
    GRX_REQUIRE(values && mask && out, GRX_E_INVALID, "%s: null pointer", who);
    const int G = R / K;
    const int64_t total = (int64_t)G * Tc;
    hipLaunchKernelGGL(group_normalize_tokenwise_kernel, dim3((unsigned)((total + SC_THREADS - 1) / SC_THREADS)),
                       dim3(SC_THREADS), 0, (hipStream_t)stream, values, mask, G, Tc, K, eps, out);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
}

// ------------------------------------------- token-level prefix advantages ---
// The scoring of the token-level trainer (token_prefix_grpo_trainer.py:107-149, 196-278) for the rows
// of one rank.  Thread = row, a workgroup holds whole groups of K rows.  Every thread matches its
// completion against its ground truth IN TOKEN SPACE -- valid positions are those with mask > 0, a
// terminal EOS takes no part -- which gives the row's token rewards (1/len(gt) or 1 on the matched
// prefix), its sequence reward (their sum) and its error-token mask (valid positions behind the
// matched prefix).  Then, per group:
//  * total_token_normalize: raw token rewards  w_p * prefix + w_n * ndcg_tok  go to `scratch`
//    (ndcg_tok: nothing / rank penalty of the row on its error tokens if any row of the group matched
//    at all / the row's NDCG reward on every valid token) and grx_token_prefix_advantages normalises
//    them per (group, position) with the mask-aware kernel above;
//  * otherwise: sequence rewards are normalised per group (mean, unbiased std, fp64 sums) and spread
//    over the matched tokens in proportion to their token rewards, plus the normalised NDCG reward
//    on every valid token.
struct TokenPrefixArgs {
    grx_token_prefix_args u;
    int rows_per_block;
};

// A row's ground-truth length as the kernel uses it: never beyond the row's own gt_stride entries (a
// length that disagrees with the buffer must not turn into reads behind the row), never negative.
__device__ __forceinline__ int bounded_gt_len(const grx_token_prefix_args& u, int r) {
    int n = u.gt_len[r];
    if (u.gt_stride > 0 && (int64_t)n > u.gt_stride) n = (int)u.gt_stride;
    return n > 0 ? n : 0;
}

__global__ __launch_bounds__(SC_THREADS) void token_prefix_kernel(TokenPrefixArgs a) {
    __shared__ float s_seq[SC_THREADS];
    __shared__ float s_ndcg[SC_THREADS];
    const grx_token_prefix_args& u = a.u;
    const int t = threadIdx.x;
    const int r = blockIdx.x * a.rows_per_block + t;
    const bool active = t < a.rows_per_block && r < u.R;
    const int Tc = u.Tc, K = u.K;
    float seq = 0.f;
    int matched = 0, n_eff = 0;
    double per_tok = 1.0;
    if (active) {
        const int64_t* __restrict__ ids = u.completion_ids + (int64_t)r * u.ids_stride;
        const float* __restrict__ mask = u.completion_mask + (int64_t)r * Tc;
        const int64_t* __restrict__ gt = u.gt_ids + (int64_t)r * u.gt_stride;
        const int gt_len = bounded_gt_len(u, r);
        int nv = 0;
        int64_t last = -1;
        for (int p = 0; p < Tc; ++p)
            if (mask[p] > 0.f) {
                ++nv;
                last = ids[p];
            }
        n_eff = (nv > 0 && u.eos_token_id >= 0 && last == (int64_t)u.eos_token_id) ? nv - 1 : nv;
        per_tok = (u.normalize && gt_len > 0) ? 1.0 / (double)gt_len : 1.0;
        const float per_tok_f = (float)per_tok;
        bool matching = true;
        int j = 0;
        for (int p = 0; p < Tc; ++p) {
            float rew = 0.f, err = 0.f;
            if (mask[p] > 0.f) {
                if (j < n_eff) {
                    if (matching && j < gt_len && ids[p] == gt[j]) {
                        rew = per_tok_f;
                        ++matched;
                    } else {
                        matching = false;
                        err = 1.f;
                    }
                }
                ++j;
            }
            if (u.token_rewards) u.token_rewards[(int64_t)r * Tc + p] = rew;
            if (u.error_mask) u.error_mask[(int64_t)r * Tc + p] = err;
        }
        seq = matched > 0 ? (float)((double)matched * per_tok) : 0.f;
        if (u.seq_rewards) u.seq_rewards[r] = seq;
    }
    s_seq[t] = seq;
    s_ndcg[t] = (active && u.has_ndcg && u.ndcg) ? u.ndcg[r] : 0.f;
    __syncthreads();
    if (!active) return;
    const int g0 = (t / K) * K;
    const float* __restrict__ mask = u.completion_mask + (int64_t)r * Tc;
    const int64_t* __restrict__ ids = u.completion_ids + (int64_t)r * u.ids_stride;
    const int64_t* __restrict__ gt = u.gt_ids + (int64_t)r * u.gt_stride;
    const int gt_len = bounded_gt_len(u, r);
    const float per_tok_f = (float)per_tok;
    // the row's token rewards / error mask again, position by position (cheaper than keeping them)
    auto for_each_position = [&](auto&& body) {
        bool matching = true;
        int j = 0;
        for (int p = 0; p < Tc; ++p) {
            float rew = 0.f, err = 0.f;
            if (mask[p] > 0.f) {
                if (j < n_eff) {
                    if (matching && j < gt_len && ids[p] == gt[j]) rew = per_tok_f;
                    else {
                        matching = false;
                        err = 1.f;
                    }
                }
                ++j;
            }
            body(p, rew, err, mask[p]);
        }
    };
    if (u.total_token_normalize) {
        float has_prefix = 0.f;
        for (int k = 0; k < K; ++k) has_prefix = s_seq[g0 + k] > 0.f ? 1.f : has_prefix;
        const float rank = (u.has_ndcg && u.error_token_penalty) ? u.rank_weights[(r + u.rank_offset) % K] : 0.f;
        const float nd = s_ndcg[t];
        float* __restrict__ out = u.scratch + (int64_t)r * Tc;
        for_each_position([&](int p, float rew, float err, float m) {
            float ndcg_tok = 0.f;
            if (u.has_ndcg) ndcg_tok = u.error_token_penalty ? __fmul_rn(__fmul_rn(rank, err), has_prefix) : __fmul_rn(nd, m);
            out[p] = __fadd_rn(__fmul_rn(u.prefix_weight, rew), __fmul_rn(u.ndcg_weight, ndcg_tok));
        });
        return;
    }
    // sequence-then-distribute
    auto group_normalised = [&](const float* s_val) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += (double)s_val[g0 + k];
        const double mean = s / (double)K;
        double q = 0.0;
        for (int k = 0; k < K; ++k) {
            const double d = (double)s_val[g0 + k] - mean;
            q += d * d;
        }
        const float sd = (float)sqrt(q / (double)(K - 1));
        return (s_val[t] - (float)mean) / (sd + u.eps);
    };
    const float adv_prefix = group_normalised(s_seq);
    const float adv_ndcg = u.has_ndcg ? group_normalised(s_ndcg) : 0.f;
    float mass = 0.f;
    for_each_position([&](int, float rew, float, float) { mass += rew; });
    const float scale = __fmul_rn(u.prefix_weight, adv_prefix);
    const float spread = __fmul_rn(u.ndcg_weight, adv_ndcg);
    float* __restrict__ out = u.token_advantages + (int64_t)r * Tc;
    for_each_position([&](int p, float rew, float, float m) {
        const float dist = mass > 0.f ? rew / (mass + 1e-8f) : 0.f;
        float v = __fmul_rn(scale, dist);
        if (u.has_ndcg) v = __fadd_rn(v, __fmul_rn(spread, m));
        out[p] = v;
    });
}

extern "C" int grx_token_prefix_advantages(const grx_token_prefix_args* args, void* stream) {
    const char* who = "grx_token_prefix_advantages";
    GRX_REQUIRE(args != nullptr, GRX_E_INVALID, "%s: null args", who);
    const grx_token_prefix_args& u = *args;
    GRX_REQUIRE(u.R >= 0 && u.Tc >= 0 && u.K >= 1, GRX_E_INVALID, "%s: bad shape", who);
    GRX_REQUIRE(u.K <= SC_THREADS, GRX_E_UNSUPPORTED, "%s: at most %d rows per group", who, SC_THREADS);
    GRX_REQUIRE(u.R % u.K == 0, GRX_E_INVALID, "%s: %d rows do not split into groups of %d", who, u.R, u.K);
    if (u.R == 0) return GRX_OK;
    if (u.Tc == 0) {  // no completion tokens: the reference's sequence rewards are zeros, the [R, 0] outputs are empty
        if (u.seq_rewards) GRX_CHECK_HIP(hipMemsetAsync(u.seq_rewards, 0, sizeof(float) * (size_t)u.R, (hipStream_t)stream));
        return GRX_OK;
    }
    GRX_REQUIRE(u.completion_ids && u.completion_mask && u.gt_ids && u.gt_len && u.token_advantages, GRX_E_INVALID,
                "%s: null pointer", who);
    GRX_REQUIRE(u.ids_stride >= u.Tc && u.gt_stride >= 0, GRX_E_INVALID, "%s: bad strides", who);
    GRX_REQUIRE(!u.total_token_normalize || u.scratch, GRX_E_INVALID,
                "%s: total_token_normalize needs scratch [R, Tc]", who);
    GRX_REQUIRE(!(u.has_ndcg && u.error_token_penalty && u.total_token_normalize) || u.rank_weights, GRX_E_INVALID,
                "%s: the error-token penalty needs rank_weights [K]", who);
    GRX_REQUIRE(!u.has_ndcg || (u.total_token_normalize && u.error_token_penalty) || u.ndcg, GRX_E_INVALID,
                "%s: has_ndcg needs the rows' NDCG rewards", who);
    GRX_REQUIRE(u.rank_offset >= 0, GRX_E_INVALID, "%s: rank_offset must be >= 0", who);
    TokenPrefixArgs a;
    a.u = u;
    a.rows_per_block = (SC_THREADS / u.K) * u.K;
    hipLaunchKernelGGL(token_prefix_kernel, dim3((u.R + a.rows_per_block - 1) / a.rows_per_block), dim3(SC_THREADS), 0,
                       (hipStream_t)stream, a);
    GRX_CHECK_HIP(hipGetLastError());
    if (u.total_token_normalize)
        return grx_group_normalize_tokenwise(u.scratch, u.completion_mask, u.R, u.Tc, u.K, u.eps, u.token_advantages,
                                             stream);
    return GRX_OK;
}
