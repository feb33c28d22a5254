// chunk_read_bench.hip — how fast can MI355X read short contiguous runs scattered over HBM?
// Passes 2 and 3 of the radix path read runs of ~32-43 k-mers (256-344 bytes, 8-byte aligned, not line aligned)
// that lie ~64 KB apart.  This microbenchmark reads random runs of C bytes from an 8 GiB buffer with LPR-lane
// copiers (one 8-byte load per lane, U independent runs in flight per copier) and reports USEFUL TB/s, for
// 8-byte-granular and for 128-byte-aligned run starts.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o chunk_read_bench tools/chunk_read_bench.hip && ./chunk_read_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// run of `words` 8-byte words; copier of LPR lanes; words > LPR: the copier walks the run in LPR-word steps
template <int LPR, int U, bool ALIGNED>
__global__ void __launch_bounds__(512) k_chunks(const uint64_t *__restrict__ buf, uint64_t n_words, int words, int iters,
                                                uint64_t *sink)
{
    const int lg = threadIdx.x % LPR;
    const uint64_t copier = ((uint64_t)blockIdx.x * 512 + threadIdx.x) / LPR;
    uint64_t acc = 0;
    const int steps = (words + LPR - 1) / LPR;
    for (int it = 0; it < iters; ++it) {
        uint64_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t r = mix(copier * 1315423911ull + (uint64_t)(it / steps) * U + u);
            uint64_t start = __umul64hi(r, n_words - words - 16);
            if (ALIGNED)
                start &= ~15ull;
            const int w = (it % steps) * LPR + lg;
            v[u] = w < words ? buf[start + w] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc += v[u];
    }
    if (acc == 0x1234567887654321ull)
        *sink = acc;
}

// 16-lane copiers, runs cut at 128-byte line boundaries: every load instruction of a copier touches ONE line
template <int LPR, int U>
__global__ void __launch_bounds__(512) k_chunks_linecut(const uint64_t *__restrict__ buf, uint64_t n_words, int words,
                                                        int runs, uint64_t *sink)
{
    const int lg = threadIdx.x % LPR;
    const uint64_t copier = ((uint64_t)blockIdx.x * 512 + threadIdx.x) / LPR;
    uint64_t acc = 0;
    const int steps = (words + LPR - 2) / LPR + 1; // most pieces a run can be cut into
    for (int r = 0; r < runs; r += U) {
        for (int piece = 0; piece < steps; ++piece) {
            uint64_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t rr = mix(copier * 1315423911ull + (uint64_t)r + u);
                const uint64_t start = __umul64hi(rr, n_words - words - 16);
                const uint64_t w = (start / LPR + piece) * LPR + lg;
                v[u] = (w >= start && w < start + words) ? buf[w] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc += v[u];
        }
    }
    if (acc == 0x1234567887654321ull)
        *sink = acc;
}

template <int LPR, int U>
double run_linecut(const uint64_t *buf, size_t bytes, int words, int runs_per_copier)
{
    uint64_t *sink; CK(hipMalloc(&sink, 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 8;
    hipLaunchKernelGGL((k_chunks_linecut<LPR, U>), dim3(grid), dim3(512), 0, 0, buf, bytes / 8, words, U, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_chunks_linecut<LPR, U>), dim3(grid), dim3(512), 0, 0, buf, bytes / 8, words, runs_per_copier, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipFree(sink));
    return (double)grid * 512 / LPR * runs_per_copier * words * 8.0 / (ms * 1e-3) / 1e12;
}

template <int LPR, int U, bool ALIGNED>
double run(const uint64_t *buf, size_t bytes, int words, int runs_per_copier)
{
    uint64_t *sink; CK(hipMalloc(&sink, 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 8;
    const int steps = (words + LPR - 1) / LPR;
    const int iters = runs_per_copier / U * steps;
    hipLaunchKernelGGL((k_chunks<LPR, U, ALIGNED>), dim3(grid), dim3(512), 0, 0, buf, bytes / 8, words, steps, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_chunks<LPR, U, ALIGNED>), dim3(grid), dim3(512), 0, 0, buf, bytes / 8, words, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipFree(sink));
    const double copiers = (double)grid * 512 / LPR;
    return copiers * (iters / steps) * U * words * 8.0 / (ms * 1e-3) / 1e12;
}

int main()
{
    const size_t bytes = (size_t)8 << 30;
    uint64_t *buf; CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    printf("# useful TB/s reading random runs from 8 GiB (grid 2048 x 512 threads, 16 runs in flight per copier)\n");
    printf("%8s %8s %10s %10s\n", "bytes", "lanes", "any_8B", "aligned");
    const int sizes[] = {64, 128, 192, 256, 344, 512, 688, 1024, 2048, 4096};
    for (int c : sizes) {
        const int words = c / 8;
        double x, y;
        if (words <= 8) { x = run<8, 16, false>(buf, bytes, words, 256); y = run<8, 16, true>(buf, bytes, words, 256); }
        else if (words <= 16) { x = run<16, 16, false>(buf, bytes, words, 256); y = run<16, 16, true>(buf, bytes, words, 256); }
        else if (words <= 32) { x = run<32, 16, false>(buf, bytes, words, 256); y = run<32, 16, true>(buf, bytes, words, 256); }
        else { x = run<64, 16, false>(buf, bytes, words, 128); y = run<64, 16, true>(buf, bytes, words, 128); }
        printf("%8d %8d %10.2f %10.2f\n", c, words <= 8 ? 8 : words <= 16 ? 16 : words <= 32 ? 32 : 64, x, y);
        fflush(stdout);
    }
    printf("# same, 16-lane copiers whatever the run length (pass 3's shape)\n");
    for (int c : {128, 256, 344, 512, 1024}) {
        const double x = run<16, 16, false>(buf, bytes, c / 8, 256);
        printf("%8d %8d %10.2f\n", c, 16, x);
        fflush(stdout);
    }
    printf("# runs cut at copier-width boundaries (16 lanes: one 128-byte line per load instruction; 32 lanes: two), 8 runs in flight\n");
    printf("%8s %10s %10s\n", "bytes", "16_lanes", "32_lanes");
    for (int c : {128, 256, 344, 512, 1024}) {
        const double x = run_linecut<16, 8>(buf, bytes, c / 8, 256), y = run_linecut<32, 8>(buf, bytes, c / 8, 256);
        printf("%8d %10.2f %10.2f\n", c, x, y);
        fflush(stdout);
    }
    return 0;
}
