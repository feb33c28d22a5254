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
// Buckets are short (load factor ~0.5), so the per-bucket quadratic steps are a handful of loads per entry.
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
