// Read-throughput ceiling of the MI355X for the logit buffer under different traversal orders.
// hipcc --offload-arch=gfx950 -O3 tools/probes/read_probe.hip -o genrec_amd/read_probe_bin && ./genrec_amd/read_probe_bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define T 256
#define U 4
__device__ __forceinline__ uint32_t fold(const u32x4& v) { return v.x ^ v.y ^ v.z ^ v.w; }
__device__ __forceinline__ void finish(uint32_t acc, uint32_t* out) {
    for (int o = 32; o > 0; o >>= 1) acc ^= __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (T / 64) + threadIdx.x / 64] = acc;
}
// (1) one workgroup per row (the streaming kernel's order)
__global__ __launch_bounds__(T) void by_row(const u32x4* p, int64_t row_vecs, uint32_t* out) {
    const u32x4* r = p + (int64_t)blockIdx.x * row_vecs;
    uint32_t acc = 0;
    int64_t i = threadIdx.x;
    for (; i + (U - 1) * T < row_vecs; i += U * T) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(r + i + u * T);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= fold(v[u]);
    }
    for (; i < row_vecs; i += T) acc ^= fold(__builtin_nontemporal_load(r + i));
    finish(acc, out);
}
// (1b) one WAVE per row: a workgroup of 4 waves holds 4 rows, no cross-wave work at all
template <int UU>
__global__ __launch_bounds__(T) void by_row_wave(const u32x4* p, int64_t row_vecs, uint32_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* r = p + ((int64_t)blockIdx.x * (T / 64) + wave) * row_vecs;
    uint32_t acc = 0;
    int64_t i = lane;
    for (; i + (UU - 1) * 64 < row_vecs; i += UU * 64) {
        u32x4 v[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) v[u] = __builtin_nontemporal_load(r + i + u * 64);
#pragma unroll
        for (int u = 0; u < UU; ++u) acc ^= fold(v[u]);
    }
    for (; i < row_vecs; i += 64) acc ^= fold(__builtin_nontemporal_load(r + i));
    finish(acc, out);
}
// (1c) one workgroup per row with the trip count known at compile time (fully unrolled)
template <int ROW_VECS>
__global__ __launch_bounds__(T) void by_row_static(const u32x4* p, uint32_t* out) {
    const u32x4* r = p + (int64_t)blockIdx.x * ROW_VECS;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < (ROW_VECS + U * T - 1) / (U * T); ++j) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = (j * U + u) * T + threadIdx.x;
            v[u] = i < ROW_VECS ? __builtin_nontemporal_load(r + i) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= fold(v[u]);
    }
    finish(acc, out);
}
// (2) flat sweep: workgroup b reads chunk b (CH vectors) of the whole buffer
template <int CH>
__global__ __launch_bounds__(T) void by_chunk(const u32x4* p, int64_t total_vecs, uint32_t* out) {
    const int64_t base = (int64_t)blockIdx.x * CH;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < CH / (U * T); ++j) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (j * U + u) * T + threadIdx.x;
            v[u] = i < total_vecs ? __builtin_nontemporal_load(p + i) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= fold(v[u]);
    }
    finish(acc, out);
}
// (3) persistent grid-stride sweep
__global__ __launch_bounds__(T) void by_stride(const u32x4* p, int64_t total_vecs, uint32_t* out) {
    uint32_t acc = 0;
    const int64_t step = (int64_t)gridDim.x * T * U;
    for (int64_t i = (int64_t)blockIdx.x * T * U + threadIdx.x; i < total_vecs; i += step) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i + u * T < total_vecs) ? __builtin_nontemporal_load(p + i + u * T) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= fold(v[u]);
    }
    finish(acc, out);
}
template <typename F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int n = 20;
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-46s %8.1f us  %7.0f GB/s\n", name, ms / n * 1e3, bytes / (ms / n) / 1e6);
}
int main() {
    const int64_t R = 8192, row_bytes = 152640 * 2, row_vecs = row_bytes / 16, total = R * row_vecs;
    u32x4* p;
    uint32_t* out;
    hipMalloc(&p, total * 16);
    hipMemset(p, 1, total * 16);
    hipMalloc(&out, 4 * 1024 * 1024 * 4);
    const double bytes = (double)total * 16;
    timeit("one workgroup per row (305 KB each)", bytes, [&] { by_row<<<R, T>>>(p, row_vecs, out); });
    timeit("one workgroup per row, static trip count", bytes, [&] { by_row_static<19080><<<R, T>>>(p, out); });
    timeit("one workgroup per 256 KB 'row' (dynamic loop)", bytes, [&] { by_row<<<total / 16384, T>>>(p, 16384, out); });
    timeit("one wave per row, 4 loads in flight", bytes, [&] { by_row_wave<4><<<R / 4, T>>>(p, row_vecs, out); });
    timeit("one wave per row, 8 loads in flight", bytes, [&] { by_row_wave<8><<<R / 4, T>>>(p, row_vecs, out); });
    timeit("flat sweep, 16 KB chunk per workgroup", bytes, [&] { by_chunk<1024><<<(total + 1023) / 1024, T>>>(p, total, out); });
    timeit("flat sweep, 64 KB chunk per workgroup", bytes, [&] { by_chunk<4096><<<(total + 4095) / 4096, T>>>(p, total, out); });
    timeit("flat sweep, 256 KB chunk per workgroup", bytes, [&] { by_chunk<16384><<<(total + 16383) / 16384, T>>>(p, total, out); });
    timeit("persistent grid-stride, 2048 workgroups", bytes, [&] { by_stride<<<2048, T>>>(p, total, out); });
    timeit("persistent grid-stride, 1024 workgroups", bytes, [&] { by_stride<<<1024, T>>>(p, total, out); });
    timeit("persistent grid-stride, 4096 workgroups", bytes, [&] { by_stride<<<4096, T>>>(p, total, out); });
    return 0;
}
