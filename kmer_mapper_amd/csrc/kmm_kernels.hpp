// kmm_kernels.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip inside its anonymous namespace.
// Direct-path kernels (k_map_reads, k_map_kmers), operator facades, index repack.
#pragma once

// ------------------------------------------------------------------------------------------------
// K1 (direct path): fused reads -> counts, every probe goes to HBM.  Used for small batches and for
// indexes whose hash space cannot be cut into L2-sized partitions.
// ------------------------------------------------------------------------------------------------

template <int S, int MODE, int PROBE>
__device__ __forceinline__ void map_one_tile(const ReadsView &rv, const IndexView &iv, const TileConst &tc,
                                             int64_t tile, int k, int max_freq, int also_rc, TileSmem<S> &sm,
                                             NodeAgg &agg, LaneStats &st)
{
    uint64_t q[S];
    const uint32_t valid = tile_kmers<S, MODE>(rv, tc, tile, k, sm, q);
    if (__builtin_amdgcn_ballot_w64(valid != 0)) {
        probe_batch<S, PROBE>(iv, agg, st, q, valid, max_freq);
        if (also_rc) {
#pragma unroll
            for (int j = 0; j < S; ++j)
                q[j] = revcomp(q[j], k);
            probe_batch<S, PROBE>(iv, agg, st, q, valid, max_freq);
        }
    }
}

// `queue` == null: static grid-stride schedule.  Otherwise the workgroups are persistent (one per CU
// slot) and pull chunks of `chunk` consecutive tiles from a device-side counter, so that the last
// round of a large launch does not leave CU slots idle while a few workgroups finish their fixed share.
template <int S, int MODE, int PROBE>
__global__ void __launch_bounds__(256) k_map_reads(ReadsView rv, IndexView iv, int k, int max_freq,
                                                   int also_rc, int64_t tile_begin, int64_t tile_end,
                                                   unsigned long long *queue, int chunk)
{
    __shared__ TileSmem<S> sm;
    __shared__ NodeAgg agg;
    __shared__ unsigned long long s_next;
    LaneStats st;
    sm.lut[threadIdx.x] = rv.lut[threadIdx.x];
    agg_init(agg); // ordered before the first agg_add by the barriers inside tile_kmers
    const TileConst tc = tile_const(rv, k);
    if (!queue) {
        for (int64_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x)
            map_one_tile<S, MODE, PROBE>(rv, iv, tc, tile, k, max_freq, also_rc, sm, agg, st);
    } else {
        for (;;) {
            if (threadIdx.x == 0)
                s_next = atomicAdd(queue, (unsigned long long)chunk);
            __syncthreads();
            const int64_t first = tile_begin + (int64_t)s_next;
            __syncthreads(); // s_next may be rewritten only after everyone has read it
            if (first >= tile_end)
                break;
            const int64_t last = first + chunk < tile_end ? first + chunk : tile_end;
            for (int64_t tile = first; tile < last; ++tile)
                map_one_tile<S, MODE, PROBE>(rv, iv, tc, tile, k, max_freq, also_rc, sm, agg, st);
        }
    }
    stats_reduce(agg, st);
    __syncthreads();
    agg_flush(iv, agg);
}

// ------------------------------------------------------------------------------------------------
// K2: operator-level lookup, uint64 k-mers already in HBM (drop-in for map_kmers_to_graph_index).
// ------------------------------------------------------------------------------------------------
template <int U, int PROBE>
__device__ __forceinline__ void map_kmer_span(const uint64_t *__restrict__ kmers, int64_t n, int64_t base,
                                              const IndexView &iv, int max_freq, int also_rc, int k,
                                              NodeAgg &agg, LaneStats &st)
{
    uint64_t q[U];
    uint32_t valid = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int64_t i = base + (int64_t)u * 256 + threadIdx.x;
        q[u] = 0;
        if (i < n) {
            q[u] = __builtin_nontemporal_load(&kmers[i]);
            valid |= 1u << u;
        }
    }
    probe_batch<U, PROBE>(iv, agg, st, q, valid, max_freq);
    if (also_rc) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            q[u] = revcomp(q[u], k);
        probe_batch<U, PROBE>(iv, agg, st, q, valid, max_freq);
    }
}

// `queue` as in k_map_reads: null = static grid-stride over spans of 256*U k-mers, otherwise persistent
// workgroups pulling `chunk` consecutive spans per grab.
template <int U, int PROBE>
__global__ void __launch_bounds__(256) k_map_kmers(const uint64_t *__restrict__ kmers, int64_t n,
                                                   IndexView iv, int max_freq, int also_rc, int k,
                                                   unsigned long long *queue, int chunk)
{
    __shared__ NodeAgg agg;
    __shared__ unsigned long long s_next;
    LaneStats st;
    agg_init(agg);
    __syncthreads();
    const int64_t span = (int64_t)256 * U;
    if (!queue) {
        for (int64_t base = (int64_t)blockIdx.x * span; base < n; base += (int64_t)gridDim.x * span)
            map_kmer_span<U, PROBE>(kmers, n, base, iv, max_freq, also_rc, k, agg, st);
    } else {
        for (;;) {
            if (threadIdx.x == 0)
                s_next = atomicAdd(queue, (unsigned long long)chunk);
            __syncthreads();
            const int64_t first = (int64_t)s_next * span;
            __syncthreads();
            if (first >= n)
                break;
            for (int c = 0; c < chunk && first + c * span < n; ++c)
                map_kmer_span<U, PROBE>(kmers, n, first + c * span, iv, max_freq, also_rc, k, agg, st);
        }
    }
    stats_reduce(agg, st);
    __syncthreads();
    agg_flush(iv, agg);
}

// General path helper: the read-start bitset of the chunk, bit p set iff some read starts at base position p
// (offsets[r] for r = 0..n_reads; offsets[n_reads] = total marks the end).  The words are cleared by the caller.
__global__ void k_mark_starts(const int64_t *__restrict__ offs, int64_t n_reads, int64_t total, uint32_t *__restrict__ bits)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = offs[r];
        if (o >= 0 && o <= total) // (out-of-range offsets are reported by k_check_offsets / the host checks)
            atomicOr(&bits[o >> 5], 1u << (o & 31));
    }
}

// read_offsets must be non-decreasing; checked here (off the host's critical path) and reported at the
// next synchronising call.  A violation cannot make the map kernels touch memory out of bounds.
__global__ void k_check_offsets(const int64_t *__restrict__ offs, int64_t n_reads, unsigned long long *first_bad)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (int64_t)gridDim.x * blockDim.x)
        if (offs[r + 1] < offs[r])
            atomicMin(&first_bad[2], (unsigned long long)r);
}

__global__ void k_iota_offsets(int64_t *out, int64_t n_reads, int64_t read_len)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_reads)
        out[i] = i * read_len;
}

// ------------------------------------------------------------------------------------------------
// Operator façade kernels (not on the fused path).
// ------------------------------------------------------------------------------------------------
// get_kmer_hashes_from_chunk_sequence (util.py:71-75) as an operator: the k-mers of all reads,
// flattened in (read, offset) order, i.e. in increasing base position.  The output slot of a window is
// therefore the number of valid windows before it: pass 1 counts the valid windows per tile, a
// two-level scan turns the counts into tile bases, pass 2 regenerates the k-mers (cheaper than storing
// them) and writes each lane's run of up to S k-mers contiguously.
template <int S, int MODE>
__global__ void __launch_bounds__(256) k_extract_count(ReadsView rv, int k, int64_t tile_begin,
                                                       int64_t tile_end, uint32_t *__restrict__ tile_cnt)
{
    __shared__ TileSmem<S> sm;
    __shared__ uint32_t s_wave[4];
    sm.lut[threadIdx.x] = rv.lut[threadIdx.x];
    const TileConst tc = tile_const(rv, k);
    for (int64_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
        uint64_t q[S];
        uint32_t c = (uint32_t)__popc(tile_kmers<S, MODE>(rv, tc, tile, k, sm, q));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
            c += __shfl_xor(c, d);
        if ((threadIdx.x & 63) == 0)
            s_wave[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0)
            tile_cnt[tile - tile_begin] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    }
}

// Exclusive scan of the (<= 1024) super-tile totals, in place; total -> *total_out.
__global__ void __launch_bounds__(1024) k_super_scan(uint32_t *super_tot, int n_super, uint32_t *total_out)
{
    __shared__ uint32_t s_a[1024];
    const int t = threadIdx.x;
    const uint32_t c = t < n_super ? super_tot[t] : 0u;
    s_a[t] = c;
    __syncthreads();
    block_scan_1024(s_a);
    if (t < n_super)
        super_tot[t] = s_a[t] - c;
    if (t == 1023)
        *total_out = s_a[t];
}

template <int S, int MODE>
__global__ void __launch_bounds__(256) k_extract_write(ReadsView rv, int k, int64_t tile_begin,
                                                       int64_t tile_end, const uint32_t *__restrict__ tile_pre,
                                                       const uint32_t *__restrict__ super_pre,
                                                       uint64_t *__restrict__ out)
{
    __shared__ TileSmem<S> sm;
    __shared__ uint32_t s_wave[4];
    sm.lut[threadIdx.x] = rv.lut[threadIdx.x];
    const TileConst tc = tile_const(rv, k);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
        uint64_t q[S];
        const uint32_t valid = tile_kmers<S, MODE>(rv, tc, tile, k, sm, q);
        const uint32_t c = (uint32_t)__popc(valid);
        const uint32_t inc = wave_scan_incl(c);
        if (lane == 63)
            s_wave[wave] = inc;
        __syncthreads();
        uint32_t base = inc - c;
        for (int w = 0; w < wave; ++w)
            base += s_wave[w];
        const int64_t lt = tile - tile_begin;
        uint64_t slot = (uint64_t)super_pre[lt >> 10] + tile_pre[lt] + base;
#pragma unroll
        for (int j = 0; j < S; ++j)
            if ((valid >> j) & 1u)
                out[slot++] = q[j];
        __syncthreads(); // s_wave is rewritten by the next tile
    }
}

// in_graph_index (mapper.pyx:112-127): first match wins, no frequency filter.
__global__ void k_in_index(const uint64_t *__restrict__ kmers, int64_t n, IndexView iv,
                           uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t q = kmers[i];
        const uint64_t hq = fastmod(q, iv.modulo, iv.magic);
        if (iv.wide) {
            const uint4 a = iv.buckets[2 * hq];
            const uint32_t kd = a.w & 3u;
            uint8_t hw = 0;
            if (kd == 1u || kd == 2u)
                hw = (((uint64_t)a.x | ((uint64_t)a.y << 32)) == q) ? 1 : 0;
            if (kd == 2u && !hw) {
                const uint4 b2 = iv.buckets[2 * hq + 1];
                hw = (((uint64_t)b2.x | ((uint64_t)b2.y << 32)) == q) ? 1 : 0;
            } else if (kd == 3u) {
                for (uint32_t j = 0; j < a.y; ++j) {
                    uint4 e = iv.entries[(uint64_t)a.x + j];
                    if (((uint64_t)e.x | ((uint64_t)e.y << 32)) == q) {
                        hw = 1;
                        break;
                    }
                }
            }
            out[i] = hw;
            continue;
        }
        const uint4 b = iv.buckets[hq];
        const uint32_t kind = b.w & 3u;
        uint8_t hit = 0;
        if (kind == 1u) {
            hit = (((uint64_t)b.x | ((uint64_t)b.y << 32)) == q) ? 1 : 0;
        } else if (kind == 2u) {
            for (uint32_t j = 0; j < b.y; ++j) {
                uint4 e = iv.entries[(uint64_t)b.x + j];
                if (((uint64_t)e.x | ((uint64_t)e.y << 32)) == q) {
                    hit = 1;
                    break;
                }
            }
        }
        out[i] = hit;
    }
}

// ------------------------------------------------------------------------------------------------
// Index repack (on the GPU, at load).  Also the validation the reference does not do.
// err bit 0: bucket outside [0, n_entries); bit 1: node outside [0, max_node_id].
// The direct view is packed either from the caller's arrays (IdxRaw: at kmm_index_create, small indexes) or later,
// on the first batch that takes the direct path, from the radix view's bucket-ordered copy (IdxRx: large indexes
// keep ONE view resident until the other is needed — HBM budget).
// ------------------------------------------------------------------------------------------------
struct IdxRaw {
    const int32_t *h2i, *nk;
    const uint64_t *kmers;
    const int32_t *nodes;
    const uint16_t *freqs;
    __device__ __forceinline__ int64_t count(uint64_t h) const { return nk[h]; }
    __device__ __forceinline__ int64_t start(uint64_t h) const { return h2i[h]; }
    __device__ __forceinline__ int64_t node(int64_t l) const { return nodes[l]; }
};

struct IdxRx { // entries in bucket order: validated and clamped when the radix view was built
    const uint32_t *pstart;
    const uint64_t *kmers;
    const uint32_t *nodes;
    const uint16_t *freqs;
    __device__ __forceinline__ int64_t count(uint64_t h) const { return (int64_t)(pstart[h + 1] - pstart[h]); }
    __device__ __forceinline__ int64_t start(uint64_t h) const { return pstart[h]; }
    __device__ __forceinline__ int64_t node(int64_t l) const { return nodes[l]; }
};

// validation only (the direct view of a large index is packed later, the errors are reported at creation)
__global__ void k_validate_index(const int32_t *__restrict__ h2i, const int32_t *__restrict__ nk,
                                 const int32_t *__restrict__ nodes, uint64_t modulo, int64_t n_entries,
                                 int64_t max_node_id, uint32_t *err)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t e = 0;
    for (uint64_t h = gid; h < modulo; h += stride) {
        const int32_t c = nk[h], s = h2i[h];
        if (c > 0 && (s < 0 || (int64_t)s + c > n_entries))
            e |= 1u;
    }
    for (uint64_t l = gid; l < (uint64_t)n_entries; l += stride) {
        const int32_t nd = nodes[l];
        if (nd < 0 || (int64_t)nd > max_node_id)
            e |= 2u;
    }
    if (e)
        atomicOr(err, e);
}

template <typename Src>
__global__ void k_pack_buckets(Src src, uint64_t modulo, int64_t n_entries, int64_t max_node_id,
                               uint4 *__restrict__ buckets, uint32_t *err)
{
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < modulo;
         h += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t c = src.count(h), s = src.start(h);
        uint4 b = make_uint4(0u, 0u, 0u, 0u);
        if (c > 0) { // c <= 0: `for j in range(n_local_hits)` runs zero times (mapper.pyx:58)
            if (s < 0 || s + c > n_entries) {
                atomicOr(err, 1u);
            } else if (c == 1) {
                const uint64_t km = src.kmers[s];
                int64_t nd = src.node(s);
                if (nd < 0 || nd > max_node_id)
                    nd = 0; // reported by k_pack_entries / k_validate_index
                b = make_uint4((uint32_t)km, (uint32_t)(km >> 32), (uint32_t)nd,
                               ((uint32_t)src.freqs[s] << 16) | 1u);
            } else if (c <= 3) {
                const uint32_t f0 = kmer_fp16(src.kmers[s]), f1 = kmer_fp16(src.kmers[s + 1]);
                const uint32_t f2 = c > 2 ? kmer_fp16(src.kmers[s + 2]) : 0u;
                b = make_uint4((uint32_t)s, (uint32_t)c, f0 | (f1 << 16), (f2 << 16) | 4u | 2u);
            } else {
                b = make_uint4((uint32_t)s, (uint32_t)c, 0u, 2u);
            }
        }
        buckets[h] = b;
    }
}

// Wide layout: 32-byte bucket = halves A (2h) and B (2h+1), up to two entries inline.
template <typename Src>
__global__ void k_pack_buckets_wide(Src src, uint64_t modulo, int64_t n_entries, int64_t max_node_id,
                                    uint4 *__restrict__ buckets, uint32_t *err)
{
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < modulo;
         h += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t c = src.count(h), s = src.start(h);
        uint4 a = make_uint4(0u, 0u, 0u, 0u), b = make_uint4(0u, 0u, 0u, 0u);
        if (c > 0) {
            if (s < 0 || s + c > n_entries) {
                atomicOr(err, 1u);
            } else if (c <= 2) {
                for (int j = 0; j < (int)c; ++j) {
                    const uint64_t km = src.kmers[s + j];
                    int64_t nd = src.node(s + j);
                    if (nd < 0 || nd > max_node_id)
                        nd = 0; // reported by k_pack_entries / k_validate_index
                    const uint4 e = make_uint4((uint32_t)km, (uint32_t)(km >> 32), (uint32_t)nd,
                                               ((uint32_t)src.freqs[s + j] << 16) | (uint32_t)c);
                    if (j == 0)
                        a = e;
                    else
                        b = e;
                }
            } else {
                a = make_uint4((uint32_t)s, (uint32_t)c, 0u, 3u);
            }
        }
        buckets[2 * h] = a;
        buckets[2 * h + 1] = b;
    }
}

template <typename Src>
__global__ void k_pack_entries(Src src, int64_t n, int64_t max_node_id, uint4 *__restrict__ entries, uint32_t *err)
{
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n;
         l += (int64_t)gridDim.x * blockDim.x) {
        uint64_t km = src.kmers[l];
        int64_t nd = src.node(l);
        if (nd < 0 || nd > max_node_id) {
            atomicOr(err, 2u);
            nd = 0;
        }
        entries[l] = make_uint4((uint32_t)km, (uint32_t)(km >> 32), (uint32_t)nd, (uint32_t)src.freqs[l]);
    }
}

// Word-blocked Bloom filter over the index k-mers: two bits inside one 32-bit word per key.
__global__ void k_build_bloom(const uint64_t *__restrict__ kmers, int64_t n, uint32_t n_words,
                              uint32_t *__restrict__ occ)
{
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n; l += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t hs = kmers[l] * 0x9E3779B97F4A7C15ull;
        const uint32_t wi = (uint32_t)(((hs >> 32) * (uint64_t)n_words) >> 32);
        atomicOr(&occ[wi], (1u << ((hs >> 7) & 31u)) | (1u << ((hs >> 12) & 31u)));
    }
}

// Occupancy bitmap with 2^shift bits per bucket: every index entry sets bit (h << shift) | (fingerprint of
// its k-mer & (2^shift - 1)).  The bitmap must be zeroed first.
__global__ void k_build_occ(const uint64_t *__restrict__ kmers, int64_t n, uint64_t modulo, uint64_t magic,
                            int shift, uint32_t *__restrict__ occ)
{
    const uint32_t sub = (1u << shift) - 1u;
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n; l += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t km = kmers[l];
        const uint64_t bit = (fastmod(km, modulo, magic) << shift) | (kmer_fp16(km) & sub);
        atomicOr(&occ[bit >> 5], 1u << (bit & 31u));
    }
}
