// Device helpers of the log-sum-exp kernels (gfx950): 16-byte row streaming, element traits of the
// logit types, and the online (max, sum-of-exp) accumulator with its wave / workgroup combines.
#pragma once

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <math.h>

#include "common.h"

#ifndef MLS_THREADS
#define MLS_THREADS 256
#endif
#define MLS_WAVES (MLS_THREADS / GRX_WAVE)
#ifndef MLS_UNROLL
#define MLS_UNROLL 4      // 16-byte loads in flight per thread
#endif
#ifndef MLS_NT
#define MLS_NT 1          // 1: non-temporal loads for the logit stream (each byte is read once)
#endif
#define MLS_EARLY (512 / MLS_THREADS > 0 ? 512 / MLS_THREADS : 1)  // allowed tokens gathered per thread up front
#define PEN_SLOT_BITS 12
#define PEN_SLOTS (1 << PEN_SLOT_BITS)  // open-addressing set of the penalised tokens of a row (<= 1536 tokens)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

#ifndef MLS_BUF
#define MLS_BUF 0         // 1: raw buffer loads with cache-policy bits MLS_AUX
#endif
#ifndef MLS_AUX
#define MLS_AUX 0
#endif

struct RowStream {
    const u32x4* p;
#if MLS_BUF
    __amdgpu_buffer_rsrc_t rsrc;
#endif
    __device__ __forceinline__ RowStream(const void* row, int64_t bytes) : p(reinterpret_cast<const u32x4*>(row)) {
#if MLS_BUF
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(row), 0, (int)bytes, 0x00020000);
#endif
    }
    __device__ __forceinline__ u32x4 load(int i) const {
#if MLS_BUF
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, i * 16, 0, MLS_AUX);
#elif MLS_NT
        return __builtin_nontemporal_load(p + i);
#else
        return p[i];
#endif
    }
};

#ifndef MLS_NTST
#define MLS_NTST 1        // 1: non-temporal stores for the candidate rows (measured: 397 -> 386 us at 162 candidates per row, 378 -> 373 at 43)
#endif
#ifndef MLS_NTFILL
#define MLS_NTFILL 0      // 1: non-temporal stores for the -inf fill of the dense output
#endif
__device__ __forceinline__ void fill_store(u32x4* p, const u32x4& v) {
#if MLS_NTFILL
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
typedef int i32x2 __attribute__((ext_vector_type(2)));
// Candidate stores.  COH = false: non-temporal, for rows the NEXT kernel reads.  COH = true: device-scope
// (written through this XCD's L2), for rows that another workgroup of this same kernel reads once the
// writer has waited for them (s_waitcnt vmcnt(0)) and counted itself in.
// Two adjacent entries (16 bytes; p is 16-byte aligned).
template <bool COH = false>
__device__ __forceinline__ void cand_store2(grx_cand* p, const grx_cand& v0, const grx_cand& v1) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 w = {v0.x, v0.y, v1.x, v1.y};
    if constexpr (COH) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(w) : "memory");
    } else {
#if MLS_NTST
        __builtin_nontemporal_store(w, reinterpret_cast<i32x4*>(p));
#else
        *reinterpret_cast<i32x4*>(p) = w;
#endif
    }
}
template <bool COH = false>
__device__ __forceinline__ void cand_store(grx_cand* p, const grx_cand& v) {
    const i32x2 w = {v.x, v.y};
    if constexpr (COH) {
        asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(w) : "memory");
    } else {
#if MLS_NTST
        __builtin_nontemporal_store(w, reinterpret_cast<i32x2*>(p));
#else
        *reinterpret_cast<i32x2*>(p) = w;
#endif
    }
}

// ---- element traits --------------------------------------------------------
template <int DT> struct Elem;
template <> struct Elem<GRX_F32> {
    typedef float T;
    static constexpr int kPerVec = 4;
    static __device__ __forceinline__ float to_float(float v) { return v; }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ float from_float(float v) { return v; }
    // element k of a loaded vector, k not known at compile time
    static __device__ __forceinline__ float element(const u32x4& v, int k) {
        return __uint_as_float(k < 2 ? (k == 0 ? v.x : v.y) : (k == 2 ? v.z : v.w));
    }
    static constexpr uint32_t kNegInfPacked = 0xFF800000u;
};
template <> struct Elem<GRX_BF16> {
    typedef uint16_t T;
    static constexpr int kPerVec = 8;
    static __device__ __forceinline__ float to_float(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xFFFF0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xFFFF0000u);
    }
    static __device__ __forceinline__ uint16_t from_float(float v) {
        __hip_bfloat16 b = __float2bfloat16(v);
        return *reinterpret_cast<uint16_t*>(&b);
    }
    static __device__ __forceinline__ float element(const u32x4& v, int k) {
        const uint32_t w = k < 4 ? (k < 2 ? v.x : v.y) : (k < 6 ? v.z : v.w);
        return __uint_as_float((k & 1) ? (w & 0xFFFF0000u) : (w << 16));
    }
    static constexpr uint32_t kNegInfPacked = 0xFF80FF80u;
};
template <> struct Elem<GRX_F16> {
    typedef uint16_t T;
    static constexpr int kPerVec = 8;
    static __device__ __forceinline__ float to_float(uint16_t v) {
        __half h = *reinterpret_cast<__half*>(&v);
        return __half2float(h);
    }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = to_float((uint16_t)(w[i] & 0xFFFFu));
            f[2 * i + 1] = to_float((uint16_t)(w[i] >> 16));
        }
    }
    static __device__ __forceinline__ uint16_t from_float(float v) {
        __half h = __float2half(v);
        return *reinterpret_cast<uint16_t*>(&h);
    }
    static __device__ __forceinline__ float element(const u32x4& v, int k) {
        const uint32_t w = k < 4 ? (k < 2 ? v.x : v.y) : (k < 6 ? v.z : v.w);
        return to_float((uint16_t)((k & 1) ? (w >> 16) : (w & 0xFFFFu)));
    }
    static constexpr uint32_t kNegInfPacked = 0xFC00FC00u;
};

// ---- online softmax accumulator ---------------------------------------------
// A thread's share of a row: m = largest element seen, s = sum of 2^(x * log2e + c) over the
// elements seen, c = -(m * log2e) as rounded when m last changed (0 while nothing finite was seen).
// exp(x - m) is evaluated as exp2(fma(x, log2e, c)): one fused multiply-add per element, two
// elements per instruction (packed fp32 math).  Keeping c itself -- instead of recomputing the
// offset from m -- makes the rescale factor of an unchanged maximum exactly 1; recomputing it
// drifts by the rounding residual of m * log2e once per batch of elements.
struct MaxSum {
    float m, s, c;
};

template <int N>
__device__ __forceinline__ void accumulate(MaxSum& a, const float* v) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float cm = v[0];
#pragma unroll
    for (int j = 1; j < N; ++j) cm = fmaxf(cm, v[j]);
    const float mn = fmaxf(a.m, cm);
    const float c = (mn == -INFINITY) ? 0.f : -mn * kLog2e;
    const f32x2 L2 = {kLog2e, kLog2e}, C2 = {c, c};
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int j = 0; j + 1 < N; j += 2) {
        const f32x2 x = {v[j], v[j + 1]};
        const f32x2 t = __builtin_elementwise_fma(x, L2, C2);
        acc0 += fast_exp2(t.x);
        acc1 += fast_exp2(t.y);
    }
    if (N & 1) acc0 += fast_exp2(fmaf(v[N - 1], kLog2e, c));
    // c <= a.c whenever a.s > 0 (the maximum only grows), so the factor is in (0, 1]; a sum that is
    // still empty has no offset yet and must not meet 2^(c - 0), which overflows for c > 128
    a.s = (a.s > 0.f ? a.s * fast_exp2(c - a.c) : 0.f) + (acc0 + acc1);
    a.m = mn;
    a.c = c;
}

// Workgroup total as (m, s) with s = sum of exp(x - m): the threads' sums are brought to the
// offset of the workgroup maximum, and the rounding residual of that offset is taken out at the end.
__device__ __forceinline__ MaxSum block_combine(MaxSum a, float* s_m, float* s_s) {
    const int lane = threadIdx.x & (GRX_WAVE - 1);
    const int wave = threadIdx.x / GRX_WAVE;
    const float M = wave_max(a.m);
    const float C = (M == -INFINITY) ? 0.f : -M * kLog2e;
    const float S = wave_sum(a.s > 0.f ? a.s * fast_exp2(C - a.c) : 0.f);
    if (lane == 0) {
        s_m[wave] = M;
        s_s[wave] = S;
    }
    __syncthreads();
    float bm = s_m[0];
#pragma unroll
    for (int w = 1; w < MLS_WAVES; ++w) bm = fmaxf(bm, s_m[w]);
    const float bc = (bm == -INFINITY) ? 0.f : -bm * kLog2e;
    float bs = 0.f;
#pragma unroll
    for (int w = 0; w < MLS_WAVES; ++w) {
        const float wc = (s_m[w] == -INFINITY) ? 0.f : -s_m[w] * kLog2e;  // the offset that wave summed at
        bs += s_s[w] > 0.f ? s_s[w] * fast_exp2(bc - wc) : 0.f;
    }
    MaxSum r;
    r.m = bm;
    // bs = sum 2^(x * log2e + bc); with m * log2e + bc = residual, sum exp(x - m) = bs * 2^(-residual)
    r.s = (bm == -INFINITY) ? bs : bs * fast_exp2(-fmaf(bm, kLog2e, bc));
    r.c = bc;
    return r;
}

// block_combine for a row with a repetition penalty.  Besides its share (m, s, c) of the row a thread
// brings sub = sum 2^(x_u log2e + c) and add = sum 2^(p (x_u log2e + c)) over the penalised elements x_u
// it streamed itself (x_u <= m, so neither overflows).  They are brought to the wave's and then to the
// workgroup's offset like s -- add with the factor raised to the power p -- and come back as
// sub = sum exp(x_u - m) and add = sum 2^(p (x_u log2e + r.c)); the caller finishes add once log s is known.
struct PenSums {
    float sub, add;
};

__device__ __forceinline__ MaxSum block_combine_pen(MaxSum a, float p, PenSums& ps, float* s_m, float* s_s,
                                                    float (*s_pen)[MLS_WAVES]) {
    const int lane = threadIdx.x & (GRX_WAVE - 1);
    const int wave = threadIdx.x / GRX_WAVE;
    const float M = wave_max(a.m);
    const float C = (M == -INFINITY) ? 0.f : -M * kLog2e;
    const bool live = a.s > 0.f;
    const float d = live ? C - a.c : 0.f;  // <= 0
    const float S = wave_sum(live ? a.s * fast_exp2(d) : 0.f);
    const float SUB = wave_sum(live ? ps.sub * fast_exp2(d) : 0.f);
    const float ADD = wave_sum(live ? ps.add * fast_exp2(p * d) : 0.f);
    if (lane == 0) {
        s_m[wave] = M;
        s_s[wave] = S;
        s_pen[0][wave] = SUB;
        s_pen[1][wave] = ADD;
    }
    __syncthreads();
    float bm = s_m[0];
#pragma unroll
    for (int w = 1; w < MLS_WAVES; ++w) bm = fmaxf(bm, s_m[w]);
    const float bc = (bm == -INFINITY) ? 0.f : -bm * kLog2e;
    float bs = 0.f, bsub = 0.f, badd = 0.f;
#pragma unroll
    for (int w = 0; w < MLS_WAVES; ++w) {
        const float wc = (s_m[w] == -INFINITY) ? 0.f : -s_m[w] * kLog2e;
        if (s_s[w] > 0.f) {
            const float g = bc - wc;
            bs += s_s[w] * fast_exp2(g);
            bsub += s_pen[0][w] * fast_exp2(g);
            badd += s_pen[1][w] * fast_exp2(p * g);
        }
    }
    MaxSum r;
    r.m = bm;
    const float unres = (bm == -INFINITY) ? 1.f : fast_exp2(-fmaf(bm, kLog2e, bc));
    r.s = bs * unres;
    r.c = bc;
    ps.sub = bsub * unres;
    ps.add = badd;
    return r;
}
