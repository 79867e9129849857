// Write-only bandwidth probe: ways of filling a 5 GB buffer with a 32-bit pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int FPT, bool NT>
__global__ __launch_bounds__(256) void fill_strided(u32x4* p, int64_t n, uint32_t v) {
    const u32x4 f = {v, v, v, v};
    const int64_t base = (int64_t)blockIdx.x * 256 * FPT + threadIdx.x;
#pragma unroll
    for (int u = 0; u < FPT; ++u) {
        const int64_t j = base + u * 256;
        if (j < n) { if (NT) __builtin_nontemporal_store(f, p + j); else p[j] = f; }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void fill_persistent(u32x4* p, int64_t n, uint32_t v) {
    const u32x4 f = {v, v, v, v};
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(f, p + j); else p[j] = f;
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void fill_persistent_chunk(u32x4* p, int64_t n, uint32_t v) {
    // each workgroup owns a contiguous slab
    const u32x4 f = {v, v, v, v};
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) {
        if (NT) __builtin_nontemporal_store(f, p + j); else p[j] = f;
    }
}

int main() {
    const int64_t bytes = 8192ll * 152640 * 4;
    const int64_t n = bytes / 16;
    void* buf;
    CK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto fn) {
        for (int i = 0; i < 2; ++i) fn();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) fn();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f us  %6.0f GB/s\n", name, ms * 100, bytes / (ms / 10) / 1e6);
    };
    u32x4* p = (u32x4*)buf;
    const uint32_t v = 0xFF800000u;
    run("memsetD32", [&] { hipMemsetD32Async((hipDeviceptr_t)buf, (int)v, bytes / 4, 0); });
    run("strided fpt1", [&] { hipLaunchKernelGGL((fill_strided<1, false>), dim3((n + 255) / 256), dim3(256), 0, 0, p, n, v); });
    run("strided fpt4", [&] { hipLaunchKernelGGL((fill_strided<4, false>), dim3((n + 1023) / 1024), dim3(256), 0, 0, p, n, v); });
    run("strided fpt8", [&] { hipLaunchKernelGGL((fill_strided<8, false>), dim3((n + 2047) / 2048), dim3(256), 0, 0, p, n, v); });
    run("strided fpt8 nt", [&] { hipLaunchKernelGGL((fill_strided<8, true>), dim3((n + 2047) / 2048), dim3(256), 0, 0, p, n, v); });
    run("strided fpt1 nt", [&] { hipLaunchKernelGGL((fill_strided<1, true>), dim3((n + 255) / 256), dim3(256), 0, 0, p, n, v); });
    for (int g : {1024, 2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "persistent g=%d", g);
        run(nm, [&] { hipLaunchKernelGGL((fill_persistent<false>), dim3(g), dim3(256), 0, 0, p, n, v); });
        snprintf(nm, 64, "persistent nt g=%d", g);
        run(nm, [&] { hipLaunchKernelGGL((fill_persistent<true>), dim3(g), dim3(256), 0, 0, p, n, v); });
        snprintf(nm, 64, "slab g=%d", g);
        run(nm, [&] { hipLaunchKernelGGL((fill_persistent_chunk<false>), dim3(g), dim3(256), 0, 0, p, n, v); });
        snprintf(nm, 64, "slab nt g=%d", g);
        run(nm, [&] { hipLaunchKernelGGL((fill_persistent_chunk<true>), dim3(g), dim3(256), 0, 0, p, n, v); });
    }
    return 0;
}
