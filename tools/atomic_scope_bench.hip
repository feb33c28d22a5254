// atomic_scope_bench.hip — are L2-scope (workgroup-scope) global atomics faster than agent-scope ones
// when every address is only touched from ONE XCD, and are their results complete after the kernel?
// Each workgroup reads HW_REG_XCC_ID and adds only into the slice owned by that XCD.
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

template <int SCOPE>
__global__ void __launch_bounds__(256) k_add(uint32_t *tab, uint32_t slice_elems, int iters, uint32_t *per_xcd_adds)
{
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    uint32_t *mine = tab + (size_t)xcc * slice_elems;
    uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        uint64_t r = mix(tid * 1315423911ull + it);
        uint32_t idx = (uint32_t)__umul64hi(r, (uint64_t)slice_elems);
        if (SCOPE == 0) atomicAdd(&mine[idx], 1u);
        else __hip_atomic_fetch_add(&mine[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (threadIdx.x == 0) atomicAdd(&per_xcd_adds[xcc], 256u * iters);
}

int main()
{
    const int grid = 256 * 32, iters = 64;
    for (uint32_t slice_kb : {256u, 1024u, 4096u}) {
        uint32_t slice_elems = slice_kb * 1024 / 4;
        size_t total = (size_t)slice_elems * 8;
        uint32_t *tab, *adds;
        CK(hipMalloc(&tab, total * 4)); CK(hipMalloc(&adds, 32));
        for (int scope = 0; scope < 2; ++scope) {
            CK(hipMemset(tab, 0, total * 4)); CK(hipMemset(adds, 0, 32));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a));
            if (scope == 0) hipLaunchKernelGGL((k_add<0>), dim3(grid), dim3(256), 0, 0, tab, slice_elems, iters, adds);
            else hipLaunchKernelGGL((k_add<1>), dim3(grid), dim3(256), 0, 0, tab, slice_elems, iters, adds);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            std::vector<uint32_t> h(total); uint32_t ha[8];
            CK(hipMemcpy(h.data(), tab, total * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ha, adds, 32, hipMemcpyDeviceToHost));
            bool ok = true; uint64_t grand = 0;
            for (int x = 0; x < 8; ++x) {
                uint64_t s = 0;
                for (uint32_t i = 0; i < slice_elems; ++i) s += h[(size_t)x * slice_elems + i];
                if (s != ha[x]) ok = false;
                grand += s;
            }
            printf("slice %5u KiB/XCD scope=%s : %7.2f G atomics/s, sums %s (xcd adds:", slice_kb,
                   scope ? "workgroup(L2)" : "agent", (double)grid * 256 * iters / (ms * 1e-3) / 1e9, ok ? "EXACT" : "WRONG");
            for (int x = 0; x < 8; ++x) printf(" %u", ha[x] / (256u * iters));
            printf(")\n");
        }
        CK(hipFree(tab)); CK(hipFree(adds));
    }
    return 0;
}
