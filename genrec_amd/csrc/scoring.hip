// Reward, ranking-metric and advantage kernels: integer comparisons on SID code
// rows and small per-group fp32 statistics.  Bytes are negligible next to the
// logit stream; these kernels exist so that rollout scoring stays on the device
// and in stream order behind the beam step.
#include <math.h>

#include "common.h"

#define SC_THREADS 256

// ------------------------------------------------------------- rewards ---
// One thread per prompt group of K consecutive rows.
__global__ void sid_rewards_kernel(const int32_t* __restrict__ pred, const int32_t* __restrict__ pred_len, int Lp,
                                   const int32_t* __restrict__ gt, const int32_t* __restrict__ gt_len, int Lg, int R,
                                   int K, int prefix_normalize, const float* __restrict__ ndcg_w,
                                   float* __restrict__ rule_out, float* __restrict__ prefix_out,
                                   float* __restrict__ ndcg_out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = g * K;
    if (r0 >= R) return;
    const int r1 = (r0 + K < R) ? r0 + K : R;
    bool any_exact = false;
    for (int r = r0; r < r1; ++r) {
        const int np = pred_len[r], ng = gt_len[r];
        const int32_t* p = pred + (int64_t)r * Lp;
        const int32_t* q = gt + (int64_t)r * Lg;
        const int n = np < ng ? np : ng;
        int matched = 0;
        while (matched < n && p[matched] == q[matched]) ++matched;
        const bool exact = (np == ng) && (matched == n);
        any_exact = any_exact || exact;
        if (rule_out) rule_out[r] = exact ? 1.0f : 0.0f;
        if (prefix_out) {
            float v = 0.0f;
            if (ng > 0) v = prefix_normalize ? (float)((double)matched / (double)ng) : (float)matched;
            prefix_out[r] = v;
        }
        // provisional NDCG value; zeroed below if the group has no exact row
        if (ndcg_out && r1 - r0 == K) ndcg_out[r] = exact ? 0.0f : ndcg_w[r - r0];
    }
    if (ndcg_out && r1 - r0 == K && !any_exact) {
        for (int r = r0; r < r1; ++r) ndcg_out[r] = 0.0f;
    }
}

extern "C" int grx_sid_rewards(const int32_t* pred, const int32_t* pred_len, int32_t Lp, const int32_t* gt,
                               const int32_t* gt_len, int32_t Lg, int32_t R, int32_t K, int32_t prefix_normalize,
                               const float* ndcg_weights, float* rule_out, float* prefix_out, float* ndcg_out,
                               void* stream) {
    const char* who = "grx_sid_rewards";
    GRX_REQUIRE(R >= 0 && K >= 1 && Lp >= 0 && Lg >= 0, GRX_E_INVALID, "%s: bad shape", who);
    if (R == 0) return GRX_OK;
    GRX_REQUIRE(pred_len && gt_len && (Lp == 0 || pred) && (Lg == 0 || gt), GRX_E_INVALID, "%s: null input", who);
    GRX_REQUIRE(ndcg_out == nullptr || ndcg_weights != nullptr, GRX_E_INVALID, "%s: ndcg_out needs ndcg_weights", who);
    const int groups = (R + K - 1) / K;
    hipLaunchKernelGGL(sid_rewards_kernel, dim3((groups + SC_THREADS - 1) / SC_THREADS), dim3(SC_THREADS), 0,
                       (hipStream_t)stream, pred, pred_len, Lp, gt, gt_len, Lg, R, K, prefix_normalize, ndcg_weights,
                       rule_out, prefix_out, ndcg_out);
    GRX_CHECK_HIP(hipGetLastError());
    return GRX_OK;
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
__device__ __forceinline__ double block_sum_f64(double v, double* s_tmp) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, GRX_WAVE);
    const int lane = threadIdx.x & (GRX_WAVE - 1), wave = threadIdx.x / GRX_WAVE;
    __syncthreads();  // s_tmp may still be read from a previous call
    if (lane == 0) s_tmp[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < SC_THREADS / GRX_WAVE; ++w) t += s_tmp[w];
    return t;
}

__device__ __forceinline__ float weighted_nansum(const float* __restrict__ rpf, const float* __restrict__ w, int r,
                                                 int n_funcs) {
    float acc = 0.0f;
    for (int f = 0; f < n_funcs; ++f) {
        const float v = rpf[(int64_t)r * n_funcs + f] * w[f];
        acc += (v != v) ? 0.0f : v;
    }
    return acc;
}

// One thread per group.  scale: 0 none, 1 group, 2 batch.
__global__ __launch_bounds__(SC_THREADS) void group_advantage_kernel(const float* __restrict__ rpf,
                                                                      const float* __restrict__ w, int R, int n_funcs,
                                                                      int K, int scale, float eps,
                                                                      float* __restrict__ rewards,
                                                                      float* __restrict__ adv,
                                                                      float* __restrict__ gmean,
                                                                      float* __restrict__ gstd) {
    __shared__ double s_tmp[SC_THREADS / GRX_WAVE];
    float batch_std = 0.0f;
    if (scale == 2) {
        // every workgroup recomputes the unbiased std of all R rewards (R is small)
        double s = 0.0;
        for (int r = threadIdx.x; r < R; r += SC_THREADS) s += (double)weighted_nansum(rpf, w, r, n_funcs);
        const double mean = block_sum_f64(s, s_tmp) / (double)R;
        double q = 0.0;
        for (int r = threadIdx.x; r < R; r += SC_THREADS) {
            const double dlt = (double)weighted_nansum(rpf, w, r, n_funcs) - mean;
            q += dlt * dlt;
        }
        batch_std = (float)sqrt(block_sum_f64(q, s_tmp) / (double)(R - 1));
    }
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = g * K;
    if (r0 >= R) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)weighted_nansum(rpf, w, r0 + k, n_funcs);
    const float mean = (float)(s / (double)K);
    double q = 0.0;
    for (int k = 0; k < K; ++k) {
        const double dlt = (double)weighted_nansum(rpf, w, r0 + k, n_funcs) - s / (double)K;
        q += dlt * dlt;
    }
    const float gsd = (float)sqrt(q / (double)(K - 1));
    const float sd = (scale == 2) ? batch_std : gsd;
    for (int k = 0; k < K; ++k) {
        const int r = r0 + k;
        const float rew = weighted_nansum(rpf, w, r, n_funcs);
        float a = rew - mean;
        if (scale != 0) a = a / (sd + eps);
        if (rewards) rewards[r] = rew;
        if (adv) adv[r] = a;
        if (gmean) gmean[r] = mean;
        if (gstd) gstd[r] = sd;
    }
}

extern "C" int grx_group_advantage(const float* rewards_per_func, const float* weights, int32_t R, int32_t n_funcs,
                                   int32_t K, int32_t scale, float eps, float* rewards, float* advantages,
                                   float* group_mean, float* group_std, void* stream) {
    const char* who = "grx_group_advantage";
    GRX_REQUIRE(R >= 0 && n_funcs >= 1 && K >= 1, GRX_E_INVALID, "%s: bad shape", who);
    GRX_REQUIRE(R % K == 0, GRX_E_INVALID, "%s: %d rows do not split into groups of %d", who, R, K);
    GRX_REQUIRE(scale >= 0 && scale <= 2, GRX_E_INVALID, "%s: bad scale %d", who, scale);
    if (R == 0) return GRX_OK;
    GRX_REQUIRE(rewards_per_func && weights, GRX_E_INVALID, "%s: null input", who);
    const int groups = R / K;
    hipLaunchKernelGGL(group_advantage_kernel, dim3((groups + SC_THREADS - 1) / SC_THREADS), dim3(SC_THREADS), 0,
                       (hipStream_t)stream, rewards_per_func, weights, R, n_funcs, K, scale, eps, rewards, advantages,
                       group_mean, group_std);
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
