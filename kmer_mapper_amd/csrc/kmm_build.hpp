// kmm_build.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip inside its anonymous namespace.
// Index construction on the GPU: flat (k-mer, node) pairs -> the five arrays of the Kmer Index format
// (what graph_kmer_index's KmerIndex.from_flat_kmers(flat_kmers, modulo) produces; reference call site
// tests/test_mapping.py:36-38).  A counting sort by hash, written out by hand:
//   k_bi_hist     n_kmers[h] = number of entries with kmer % modulo == h          (global atomics)
//   scan          hashes_to_index = exclusive prefix of n_kmers                   (multi-level block scan)
//   k_bi_scatter  entries -> a slot of their bucket, in arrival order             (atomic cursors)
//   k_bi_place    every entry ranks itself inside its bucket by ORIGINAL position (so the result equals a
//                 stable sort by hash, bit for bit reproducible) and counts the entries of its bucket that
//                 hold the same k-mer -> frequencies (clipped to uint16 like the format)
// Buckets are short (load factor ~0.5), so the per-bucket quadratic steps of k_bi_place are a handful of loads per
// entry; buckets with more than BI_BIG entries (a k-mer with thousands of hits — the reason the lookup has a
// frequency cutoff, mapper.pyx:64-66) are listed by k_bi_list_big and ordered by k_bi_big instead, one workgroup
// per bucket, with two bitonic sorts (original positions; k-mers for the frequencies): O(c log^2 c), not O(c^2).
#pragma once

__global__ void k_bi_hist(const uint64_t *__restrict__ kmers, int64_t n, uint64_t modulo, uint64_t magic,
                          uint32_t *__restrict__ nk)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&nk[fastmod(kmers[i], modulo, magic)], 1u);
}

// Exclusive scan of 1024-element blocks: out = local exclusive prefix, block_sum[b] = block total.
__global__ void __launch_bounds__(1024) k_scan_blocks(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                      uint32_t *__restrict__ block_sum, uint64_t n)
{
    __shared__ uint32_t s_a[1024];
    const int t = threadIdx.x;
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + t;
    const uint32_t c = i < n ? in[i] : 0u;
    s_a[t] = c;
    __syncthreads();
    block_scan_1024(s_a);
    if (i < n)
        out[i] = s_a[t] - c;
    if (t == 1023)
        block_sum[blockIdx.x] = s_a[t];
}

__global__ void k_scan_add(uint32_t *__restrict__ out, const uint32_t *__restrict__ block_pre, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] += block_pre[i >> 10];
}

__global__ void k_bi_scatter(const uint64_t *__restrict__ kmers, int64_t n, uint64_t modulo, uint64_t magic,
                             const uint32_t *__restrict__ h2i, uint32_t *__restrict__ cursor,
                             uint32_t *__restrict__ slot_src)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = fastmod(kmers[i], modulo, magic);
        slot_src[h2i[h] + atomicAdd(&cursor[h], 1u)] = (uint32_t)i;
    }
}

constexpr int BI_BIG = 64;

__global__ void k_bi_list_big(const uint32_t *__restrict__ nk, uint64_t modulo, uint32_t *__restrict__ list,
                              uint32_t *__restrict__ count)
{
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < modulo; h += (uint64_t)gridDim.x * blockDim.x)
        if (nk[h] > (uint32_t)BI_BIG)
            list[atomicAdd(count, 1u)] = (uint32_t)h;
}

// Ascending bitonic sort of a[0..n) in global memory by one workgroup, any n: the normalised network (every
// comparator puts the minimum at the lower index; first step of each merge mirrored), comparators whose upper
// element does not exist are skipped (= padding with +inf).  Loads bypass the CU's L1 (agent scope): another
// wave of this workgroup may have just rewritten the element.
template <typename T>
__device__ __forceinline__ void bi_cswap(T *a, uint32_t i, uint32_t l)
{
    const T x = __hip_atomic_load(&a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const T y = __hip_atomic_load(&a[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (x > y) {
        a[i] = y;
        a[l] = x;
    }
}

template <typename T>
__device__ __forceinline__ void bi_sort_global(T *a, uint32_t n)
{
    for (uint64_t k = 2; (k >> 1) < n; k <<= 1) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint64_t l = (uint64_t)i ^ (k - 1);
            if (l > i && l < n)
                bi_cswap(a, i, (uint32_t)l);
        }
        __syncthreads();
        for (uint64_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
                const uint64_t l = (uint64_t)i ^ j;
                if (l > i && l < n)
                    bi_cswap(a, i, (uint32_t)l);
            }
            __syncthreads();
        }
    }
}

// One workgroup per listed bucket: entries ordered by original position (like the stable sort upstream), the
// frequency of an entry = length of its k-mer's run among the bucket's sorted k-mers.
__global__ void __launch_bounds__(1024) k_bi_big(const uint64_t *__restrict__ kmers, const int32_t *__restrict__ nodes,
                                                 const uint32_t *__restrict__ h2i, const uint32_t *__restrict__ nk,
                                                 const uint32_t *__restrict__ list, const uint32_t *__restrict__ count,
                                                 uint32_t *slot_src, uint64_t *ksort, uint64_t *__restrict__ kmers_out,
                                                 int32_t *__restrict__ nodes_out, uint16_t *__restrict__ freqs_out)
{
    const uint32_t n_big = *count;
    for (uint32_t b = blockIdx.x; b < n_big; b += gridDim.x) {
        const uint32_t h = list[b], b0 = h2i[h], c = nk[h];
        uint32_t *src = slot_src + b0;
        uint64_t *ks = ksort + b0;
        __syncthreads();
        bi_sort_global(src, c);
        for (uint32_t i = threadIdx.x; i < c; i += blockDim.x)
            ks[i] = kmers[__hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)];
        __syncthreads();
        bi_sort_global(ks, c);
        for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
            const uint32_t s = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t km = kmers[s];
            uint32_t lo = 0, hi = c; // first position with ks >= km
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (__hip_atomic_load(&ks[mid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < km)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const uint32_t first = lo;
            hi = c;                  // first position with ks > km
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (__hip_atomic_load(&ks[mid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= km)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const uint32_t same = lo - first;
            kmers_out[b0 + i] = km;
            nodes_out[b0 + i] = nodes[s];
            freqs_out[b0 + i] = (uint16_t)(same > 65535u ? 65535u : same);
        }
    }
}

__global__ void k_bi_place(const uint64_t *__restrict__ kmers, const int32_t *__restrict__ nodes, int64_t n,
                           uint64_t modulo, uint64_t magic, const uint32_t *__restrict__ h2i,
                           const uint32_t *__restrict__ nk, const uint32_t *__restrict__ slot_src,
                           uint64_t *__restrict__ kmers_out, int32_t *__restrict__ nodes_out,
                           uint16_t *__restrict__ freqs_out)
{
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t src = slot_src[p];
        const uint64_t km = kmers[src];
        const uint64_t h = fastmod(km, modulo, magic);
        const uint32_t b0 = h2i[h], c = nk[h];
        if (c > (uint32_t)BI_BIG)
            continue; // k_bi_big
        uint32_t rank = 0, same = 0;
        for (uint32_t j = 0; j < c; ++j) {
            const uint32_t other = slot_src[b0 + j];
            rank += other < src ? 1u : 0u;
            same += kmers[other] == km ? 1u : 0u;
        }
        kmers_out[b0 + rank] = km;
        nodes_out[b0 + rank] = nodes[src];
        freqs_out[b0 + rank] = (uint16_t)(same > 65535u ? 65535u : same);
    }
}

// The format stores 0 in hashes_to_index for empty buckets (only used buckets are filled upstream).
__global__ void k_bi_zero_empty(uint32_t *__restrict__ h2i, const uint32_t *__restrict__ nk, uint64_t modulo)
{
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < modulo; h += (uint64_t)gridDim.x * blockDim.x)
        if (nk[h] == 0)
            h2i[h] = 0;
}
