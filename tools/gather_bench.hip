// gather_bench.hip — microbenchmark of the MI355X random-access ceilings that bound the k-mer
// index probe: independent random 8/16-byte gathers and random uint32 atomic adds as a function of
// table size (L2 4 MiB/XCD, Infinity Cache 256 MiB, HBM).  Not part of the product; its numbers
// go into DESIGN.md ("sector-granular bound").
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench tools/gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// same as k_gather<uint4> but with non-temporal loads
template <int U>
__global__ void __launch_bounds__(256) k_gather_nt(const uint4 *__restrict__ tab, uint64_t n_elems, int iters,
                                                   uint64_t *sink)
{
    uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t r = mix(tid * 1315423911ull + (uint64_t)it * U + u);
            uint64_t idx = __umul64hi(r, n_elems);
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(&tab[idx]));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u][0];
    }
    if (acc == 0x1234567887654321ull) *sink = acc;
}

template <typename T, int U>
__global__ void __launch_bounds__(256) k_gather(const T *__restrict__ tab, uint64_t n_elems, int iters,
                                                uint64_t *sink)
{
    uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t r = mix(tid * 1315423911ull + (uint64_t)it * U + u);
            uint64_t idx = __umul64hi(r, n_elems);
            v[u] = tab[idx];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x;
    }
    if (acc == 0x1234567887654321ull) *sink = acc;
}

template <int U>
__global__ void __launch_bounds__(256) k_atomic(uint32_t *tab, uint64_t n_elems, int iters, int active_pct)
{
    uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t r = mix(tid * 1315423911ull + (uint64_t)it * U + u);
            uint64_t idx = __umul64hi(r, n_elems);
            if ((int)(r % 100) < active_pct) atomicAdd(&tab[idx], 1u);
        }
    }
}

template <typename T, int U>
double run_gather(const void *tab, size_t bytes, int grid, int iters)
{
    uint64_t *sink; CK(hipMalloc(&sink, 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    uint64_t n = bytes / sizeof(T);
    hipLaunchKernelGGL((k_gather<T, U>), dim3(grid), dim3(256), 0, 0, (const T *)tab, n, 2, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_gather<T, U>), dim3(grid), dim3(256), 0, 0, (const T *)tab, n, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipFree(sink));
    return (double)grid * 256 * iters * U / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv)
{
    size_t max_bytes = (size_t)8 << 30;
    void *tab; CK(hipMalloc(&tab, max_bytes));
    CK(hipMemset(tab, 1, max_bytes));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 32;
    printf("# random gathers, G loads/s (grid %d x 256 threads)\n", grid);
    printf("%10s %12s %12s %12s %12s\n", "table_MiB", "8B_U4", "8B_U8", "16B_U4", "16B_U8");
    size_t sizes[] = {(size_t)2 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)160 << 20, (size_t)320 << 20,
                      (size_t)640 << 20, (size_t)1600 << 20, (size_t)3200 << 20, (size_t)8 << 30};
    for (size_t s : sizes) {
        double a = run_gather<uint2, 4>(tab, s, grid, 16);
        double b = run_gather<uint2, 8>(tab, s, grid, 8);
        double c = run_gather<uint4, 4>(tab, s, grid, 16);
        double d = run_gather<uint4, 8>(tab, s, grid, 8);
        printf("%10zu %12.1f %12.1f %12.1f %12.1f\n", s >> 20, a, b, c, d);
        fflush(stdout);
    }
    printf("# 16-byte gathers with non-temporal loads, U=4\n");
    for (size_t s : {(size_t)320 << 20, (size_t)3200 << 20}) {
        uint64_t *sink; CK(hipMalloc(&sink, 8));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL((k_gather_nt<4>), dim3(grid), dim3(256), 0, 0, (const uint4 *)tab, s / 16, 2, sink);
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_gather_nt<4>), dim3(grid), dim3(256), 0, 0, (const uint4 *)tab, s / 16, 16, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%10zu %12.1f\n", s >> 20, (double)grid * 256 * 16 * 4 / (ms * 1e-3) / 1e9);
    }
    printf("# random uint32 atomicAdd (no return), G atomics/s; active = %% of lanes issuing\n");
    printf("%10s %12s %12s %12s\n", "table_MiB", "act100", "act20", "act5");
    size_t asizes[] = {(size_t)4 << 10, (size_t)1 << 20, (size_t)40 << 20, (size_t)400 << 20, (size_t)4 << 30};
    for (size_t s : asizes) {
        double r[3]; int pcts[3] = {100, 20, 5};
        for (int i = 0; i < 3; ++i) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            hipLaunchKernelGGL((k_atomic<4>), dim3(grid), dim3(256), 0, 0, (uint32_t *)tab, s / 4, 1, pcts[i]);
            CK(hipEventRecord(a));
            int iters = 8;
            hipLaunchKernelGGL((k_atomic<4>), dim3(grid), dim3(256), 0, 0, (uint32_t *)tab, s / 4, iters, pcts[i]);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            r[i] = (double)grid * 256 * iters * 4 * pcts[i] / 100.0 / (ms * 1e-3) / 1e9;
        }
        printf("%10.3f %12.2f %12.2f %12.2f\n", s / 1048576.0, r[0], r[1], r[2]);
        fflush(stdout);
    }
    CK(hipFree(tab));
    return 0;
}
