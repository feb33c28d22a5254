// stream_bench.hip — what does one MI355X sustain on plain streams?  The ceilings the radix passes are read against:
// read-only, write-only, copy (1 read : 1 write) and pass 1's mix (1 read : 6.5 written), 16 bytes per lane per access,
// grid-stride over 4 GiB, plain and non-temporal stores.  Reports TB/s of bytes moved (read + written).
// Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o stream_bench tools/stream_bench.hip && ./stream_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

typedef uint32_t v4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void __launch_bounds__(256) k_read(const v4 *__restrict__ a, size_t n, uint32_t *sink)
{
    v4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = __builtin_nontemporal_load(a + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
        *sink = acc.x;
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) k_write(v4 *__restrict__ a, size_t n)
{
    const v4 val = {threadIdx.x, blockIdx.x, 3u, 4u};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT)
                __builtin_nontemporal_store(val, a + i + u * stride);
            else
                a[i + u * stride] = val;
        }
    }
}

// write-only in the shapes the passes use: a workgroup writes TILE contiguous bytes (B bytes per lane and store), tiles
// grid-strided (pass 1: 64 KB tiles, 8 bytes per lane); CHUNKED: every workgroup owns one contiguous region instead
template <typename T, bool CHUNKED>
__global__ void __launch_bounds__(256) k_write_tiles(T *__restrict__ a, size_t n, size_t tile, T val)
{
    const size_t n_tiles = n / tile;
    if (CHUNKED) {
        const size_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
        const size_t t0 = blockIdx.x * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
        for (size_t i = t0 * tile + threadIdx.x; i < t1 * tile; i += 256)
            a[i] = val;
    } else {
        for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x)
            for (size_t i = threadIdx.x; i < tile; i += 256)
                a[t * tile + i] = val;
    }
}

// R reads of 16 B per W writes of 16 B per lane and round
template <int R, int W, bool NT>
__global__ void __launch_bounds__(256) k_mix(const v4 *__restrict__ a, size_t n_a, v4 *__restrict__ b, size_t n_b)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t ia = (size_t)blockIdx.x * 256 + threadIdx.x, ib = ia;
    for (; ia + (R - 1) * stride < n_a && ib + (W - 1) * stride < n_b; ia += R * stride, ib += W * stride) {
        v4 v[R];
#pragma unroll
        for (int u = 0; u < R; ++u)
            v[u] = __builtin_nontemporal_load(a + ia + u * stride);
        v4 s = v[0];
#pragma unroll
        for (int u = 1; u < R; ++u)
            s ^= v[u];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if (NT)
                __builtin_nontemporal_store(s, b + ib + u * stride);
            else
                b[ib + u * stride] = s;
            s.x += 1u;
        }
    }
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r)
        launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const size_t bytes = 4ull << 30, n = bytes / 16;
    v4 *a, *b;
    uint32_t *sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs; 4 GiB per array, 16 B per lane per access; TB/s of bytes moved\n", p.name, cus);
    for (int wg_per_cu : {4, 8, 16}) {
        const int grid = cus * wg_per_cu;
        double t;
        t = time_ms([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, a, n, sink); }, 10);
        printf("grid %5d  read-only                      %6.2f TB/s\n", grid, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((k_write<4, false>), dim3(grid), dim3(256), 0, 0, b, n); }, 10);
        printf("grid %5d  write-only                     %6.2f TB/s\n", grid, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((k_write<4, true>), dim3(grid), dim3(256), 0, 0, b, n); }, 10);
        printf("grid %5d  write-only, non-temporal       %6.2f TB/s\n", grid, bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((k_mix<2, 2, false>), dim3(grid), dim3(256), 0, 0, a, n, b, n); }, 10);
        printf("grid %5d  copy 1:1                       %6.2f TB/s\n", grid, 2.0 * bytes / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((k_mix<2, 2, true>), dim3(grid), dim3(256), 0, 0, a, n, b, n); }, 10);
        printf("grid %5d  copy 1:1, non-temporal stores  %6.2f TB/s\n", grid, 2.0 * bytes / t / 1e9);
        // pass 1: 1.25 B read per 8 B written  ~ 1 : 6.4  (here 1 : 6, bounded by the written array)
        t = time_ms([&] { hipLaunchKernelGGL((k_mix<1, 6, false>), dim3(grid), dim3(256), 0, 0, a, n, b, n); }, 10);
        printf("grid %5d  1 read : 6 written (pass 1)    %6.2f TB/s\n", grid, (bytes + bytes / 6.0) / t / 1e9);
        // pass 2 with the filter: 8 B read per 4.1 B written ~ 2 : 1
        t = time_ms([&] { hipLaunchKernelGGL((k_mix<4, 2, false>), dim3(grid), dim3(256), 0, 0, a, n, b, n); }, 10);
        printf("grid %5d  2 read : 1 written (pass 2)    %6.2f TB/s\n", grid, (bytes + bytes / 2.0) / t / 1e9);
    }
    {
        const v4 z4 = {0, 0, 0, 0}, x4 = {0x12345678u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u};
        for (int grid : {cus * 2, cus * 4, cus * 8, cus * 16}) {
            double t;
            t = time_ms([&] { hipLaunchKernelGGL((k_write_tiles<v4, false>), dim3(grid), dim3(256), 0, 0, b, n, (size_t)4096, x4); }, 10);
            printf("grid %5d  write 64 KB tiles, 16 B/lane     %6.2f TB/s\n", grid, bytes / t / 1e9);
            t = time_ms([&] { hipLaunchKernelGGL((k_write_tiles<v4, false>), dim3(grid), dim3(256), 0, 0, b, n, (size_t)4096, z4); }, 10);
            printf("grid %5d  ... of zeros                     %6.2f TB/s\n", grid, bytes / t / 1e9);
            t = time_ms([&] { hipLaunchKernelGGL((k_write_tiles<uint64_t, false>), dim3(grid), dim3(256), 0, 0, (uint64_t *)b, n * 2, (size_t)8192, (uint64_t)0x123456789abcdefull); }, 10);
            printf("grid %5d  write 64 KB tiles, 8 B/lane      %6.2f TB/s\n", grid, bytes / t / 1e9);
            t = time_ms([&] { hipLaunchKernelGGL((k_write_tiles<v4, true>), dim3(grid), dim3(256), 0, 0, b, n, (size_t)4096, x4); }, 10);
            printf("grid %5d  write, one region per workgroup  %6.2f TB/s\n", grid, bytes / t / 1e9);
            t = time_ms([&] { hipLaunchKernelGGL((k_write_tiles<v4, false>), dim3(grid), dim3(256), 0, 0, b, n, (size_t)256, x4); }, 10);
            printf("grid %5d  write 4 KB tiles, 16 B/lane      %6.2f TB/s\n", grid, bytes / t / 1e9);
        }
    }
    CK(hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice));
    double t = time_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
    printf("hipMemcpy device to device             %6.2f TB/s\n", 2.0 * bytes / t / 1e9);
    t = time_ms([&] { CK(hipMemsetAsync(b, 0, bytes, 0)); }, 10);
    printf("hipMemset of zeros                     %6.2f TB/s\n", bytes / t / 1e9);
    t = time_ms([&] { CK(hipMemsetD32Async((hipDeviceptr_t)b, 0x12345678, bytes / 4, 0)); }, 10);
    printf("hipMemsetD32 of 0x12345678             %6.2f TB/s\n", bytes / t / 1e9);
    return 0;
}
