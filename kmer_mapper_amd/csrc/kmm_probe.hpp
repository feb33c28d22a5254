// kmm_probe.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip inside its anonymous namespace.
// HBM layout of the index, exact modulo, reverse complement, LDS node-count aggregation, bucket probe.
#pragma once

// ------------------------------------------------------------------------------------------------
// device-side data layout (ours; the .npz surface is unchanged)
//   bucket h : uint4, 16 B — ONE gather resolves an empty or single-entry bucket:
//        w & 3 == 0  empty
//        w & 3 == 1  single entry stored inline: {x,y} = k-mer, z = node, w >> 16 = frequency
//        w & 3 == 2  two or more entries: x = start, y = count into `entries`; if w & 4, the bucket has at
//                    most three entries and z[15:0], z[31:16], w[31:16] hold their 16-bit fingerprints
//   wide variant (indexes too large for the L2 bitmap): bucket h = two uint4 halves A = buckets[2h],
//        B = buckets[2h+1] inside one 64-byte fabric request; A.w & 3 == 1 one entry in A, == 2 two entries
//        in A and B (B is then an L2 hit), == 3 three or more: A.x = start, A.y = count into `entries`
//   entry  l : uint4 {kmer_lo, kmer_hi, node, freq}, 16 B, in the index's own order (grouped by hash)
// The MI355X random-access ceiling is ~55 G L2-missing requests/s whatever their width (8 or 16 B,
// profiles/r01/gather_bench_mi355x.txt), so the layout minimises REQUESTS per k-mer, not bytes.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct IndexView {
    const uint4 *buckets;
    const uint4 *entries;
    const uint32_t *occ; // optional L2-resident occupancy bitmap (bit h = bucket h non-empty), or null
    int occ_shift;       // log2(bits per bucket) of the bitmap: bit index = (h << occ_shift) | low fingerprint bits
    uint32_t bloom_words; // != 0: `occ` is a word-blocked Bloom filter of that many 32-bit words (two bits per key
                          // inside one word chosen by a hash of the k-mer) instead of the per-bucket bitmap
    int wide;            // 1: 32-byte buckets with two inline entries (never together with occ)
    uint32_t *counts;
    unsigned long long *stats; // [0] k-mer lookups performed, [1] count increments (hits)
    uint64_t modulo;
    uint64_t magic; // floor(2^64 / modulo) (all ones for modulo == 1)
};

struct ReadsView {
    const uint8_t *bases;
    int64_t total;             // number of base bytes
    const int64_t *offsets;    // n_reads + 1 (general path)
    int64_t n_reads;
    const uint32_t *start_bits; // general path: bit p is set iff a read starts at base position p (k_mark_starts)
    int64_t n_start_words;
    uint64_t read_len;         // uniform path
    uint64_t read_len_magic;   // floor(2^64 / read_len)
    // packed uniform tiles (radix pass 1, kmm_tile.hpp tile_packed_*): a tile = pk_rpt whole reads, pk_lpr lanes per read
    // with pk_S consecutive windows each — every window a lane computes is a real k-mer (0 = packed tiles not used)
    uint32_t pk_rpt, pk_lpr, pk_S, pk_W, pk_inv;
    uint32_t codes2;           // != 0: `bases` is a stream of 2-bit codes, 16 per 32-bit word (position p = bits [2 (p & 15), +2)
                               // of word p >> 4): what k_rec_scatter compacts raw records into; no lookup table, no invalid bytes
    const uint8_t *lut;        // 256 bytes in HBM
    unsigned long long *first_bad; // [0] min position of a non-nucleotide byte, [1] of a malformed
                                   //     record line (both init ~0)
    // records mode (raw FASTQ / two-line FASTA bytes): newlines before every tile
    const uint32_t *tile_nl;   // per tile: newlines before the tile inside its super-tile (1024 tiles)
    const uint32_t *super_nl;  // per super-tile: newlines before it
    uint32_t period_mask;      // lines per record - 1 (3 for FASTQ, 1 for two-line FASTA)
    uint32_t header_char;      // '@' or '>'
};

enum { MODE_GENERAL = 0, MODE_UNIFORM = 1, MODE_RECORDS = 2, MODE_PACKED = 4 }; // (3 = MODE_KMERS, kmm_radix.hpp)

// Exact x % m for any m >= 1 with one 64x64->hi multiply: q = hi64(x * floor(2^64/m)) is either
// floor(x/m) or one less (x * (2^64/m - magic) / 2^64 < 1), so a single conditional subtract
// restores the remainder.  The reference computes kmers[i] % modulo with a hardware divide
// (mapper.pyx:54); results are identical for every x.
__device__ __forceinline__ uint64_t fastmod(uint64_t x, uint64_t m, uint64_t magic)
{
    uint64_t q = __umul64hi(x, magic);
    uint64_t r = x - q * m;
    return r >= m ? r - m : r;
}

__device__ __forceinline__ uint64_t fastdiv(uint64_t x, uint64_t m, uint64_t magic, uint64_t *rem)
{
    uint64_t q = __umul64hi(x, magic);
    uint64_t r = x - q * m;
    if (r >= m) {
        r -= m;
        q += 1;
    }
    *rem = r;
    return q;
}

// Same for m < 2^31 (every index in the reference's int32 format): quotient and 32-bit remainder.  The true
// remainder before the correction step is < 2m < 2^32, so it is exact modulo 2^32 and one 32-bit multiply
// replaces the 64-bit product.
__device__ __forceinline__ uint64_t fastdiv_m31(uint64_t x, uint32_t m, uint64_t magic, uint32_t *rem)
{
    uint64_t q = __umul64hi(x, magic);
    uint32_t r = (uint32_t)x - (uint32_t)q * m;
    if (r >= m) {
        r -= m;
        q += 1;
    }
    *rem = r;
    return q;
}

// (Dividing through the FP64 pipe instead — t = fma(x, 1/m, 2^52), quotient from t's mantissa, 32-bit remainder
// correction; exact for m >= 2^14 — was measured and dropped: pass 1 5.68 ms against 3.51 ms with the integer
// multiplies above: v_cvt_f64_u32 / v_fma_f64 are no cheaper than the quarter-rate 32-bit multiplies here.)

// Inclusive prefix sum over the 64 lanes of a wavefront with six DPP additions (row shifts by 1, 2, 4, 8 inside the
// rows of 16 lanes, then lane 15's and lane 31's broadcasts into the following rows) instead of six ds_bpermute
// round trips through the LDS crossbar: the scans of the radix passes sit on the critical path between two
// workgroup barriers.  Call with every lane of the wavefront active.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}

// Reverse complement under A,C,G,T = 0,1,2,3, first base in the lowest bits: complement every
// 2-bit group (NOT), reverse the groups, realign (the `-r` operation).
__device__ __forceinline__ uint64_t revcomp(uint64_t x, int k)
{
    x = ~x;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * k);
}

// 16-bit fingerprint of a k-mer, independent of its bucket (all k-mers of a bucket agree mod `modulo`).
// Multi-entry buckets of the 16-byte layout keep the fingerprints of their (up to three) entries in the
// bucket record, so a k-mer that matches none of them — nearly every k-mer that is not in the index —
// never touches `entries`.
__device__ __forceinline__ uint32_t kmer_fp16(uint64_t q)
{
    return (uint32_t)((q * 0x9E3779B97F4A7C15ull) >> 48);
}

// ------------------------------------------------------------------------------------------------
// Node-count accumulation (mapper.pyx:68: node_counts[nodes[l]] += 1).
// Real graph indexes map many k-mers to few nodes, so hits are first aggregated in a small
// direct-mapped table in LDS that lives as long as the workgroup: a hit claims the slot of its
// node (ds_cmpst) and bumps the slot's counter (ds_add); a hit whose slot belongs to another node
// falls through to one global atomicAdd.  The table is flushed with one global atomicAdd per used
// slot when the workgroup retires.  uint32 wrap-around is preserved (sums of sums mod 2^32).
// ------------------------------------------------------------------------------------------------
constexpr int KMM_STAT_SHARDS = 256;
constexpr int KMM_STAT_STRIDE = 32; // unsigned long longs = 256 bytes between shards
// slots: [0] k-mer lookups, [1] hits, [2] k-mers gathered by radix pass 2, [3] probed by radix pass 3, [4..15] phase
// timers of diagnostic builds, [16] k-mers emitted by radix pass 1, [17] dropped by pass 2's empty-bucket filter
constexpr int KMM_STAT_RX_P1 = 16, KMM_STAT_RX_DROPPED = 17;
constexpr int AGG_LOG_SLOTS = 11;
constexpr int AGG_SLOTS = 1 << AGG_LOG_SLOTS;
constexpr uint32_t AGG_EMPTY = 0xFFFFFFFFu; // node ids are < 2^31

struct NodeAgg {
    uint32_t key[AGG_SLOTS];
    uint32_t val[AGG_SLOTS];
    uint32_t st[2]; // workgroup totals of LaneStats, see stats_reduce
};

__device__ __forceinline__ void agg_init(NodeAgg &agg)
{
    for (int i = threadIdx.x; i < AGG_SLOTS; i += blockDim.x) {
        agg.key[i] = AGG_EMPTY;
        agg.val[i] = 0;
    }
    if (threadIdx.x < 2)
        agg.st[threadIdx.x] = 0;
}

__device__ __forceinline__ void agg_add_n(const IndexView &iv, NodeAgg &agg, uint32_t node, uint32_t inc)
{
    const uint32_t slot = (node * 2654435761u) >> (32 - AGG_LOG_SLOTS);
    const uint32_t prev = atomicCAS(&agg.key[slot], AGG_EMPTY, node);
    if (prev == AGG_EMPTY || prev == node)
        atomicAdd(&agg.val[slot], inc);
    else
        atomicAdd(&iv.counts[node], inc);
}

__device__ __forceinline__ void agg_add(const IndexView &iv, NodeAgg &agg, uint32_t node, uint32_t &hits)
{
    ++hits;
    agg_add_n(iv, agg, node, 1u);
}

// Call after a __syncthreads() that follows the workgroup's last agg_add.
__device__ __forceinline__ void agg_flush_counts(const IndexView &iv, NodeAgg &agg)
{
    for (int i = threadIdx.x; i < AGG_SLOTS; i += blockDim.x) {
        const uint32_t v = agg.val[i];
        if (v)
            atomicAdd(&iv.counts[agg.key[i]], v);
    }
}

__device__ __forceinline__ void agg_flush(const IndexView &iv, NodeAgg &agg)
{
    agg_flush_counts(iv, agg);
    if (threadIdx.x == 0) {
        unsigned long long *shard = iv.stats + (size_t)(blockIdx.x % KMM_STAT_SHARDS) * KMM_STAT_STRIDE;
        atomicAdd(&shard[0], (unsigned long long)agg.st[0]);
        atomicAdd(&shard[1], (unsigned long long)agg.st[1]);
    }
}

// Per-lane work counters (kmm_get_stats), kept in registers for the lifetime of the workgroup.
// stats_reduce (before the workgroup's final barrier) folds them into LDS; agg_flush's thread 0 then
// issues ONE pair of global atomics per workgroup, on a shard chosen by the workgroup id: thousands of
// workgroups retiring together on a single counter would serialise on that address.
struct LaneStats {
    uint32_t lookups = 0, hits = 0;
};

__device__ __forceinline__ void stats_reduce(NodeAgg &agg, const LaneStats &st)
{
    uint32_t a = st.lookups, b = st.hits;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        a += __shfl_xor(a, d);
        b += __shfl_xor(b, d);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&agg.st[0], a);
        atomicAdd(&agg.st[1], b);
    }
}

// mapper.pyx:60-68 for one entry.
__device__ __forceinline__ void count_if_match(const IndexView &iv, NodeAgg &agg, uint4 e, uint64_t q,
                                               int max_freq, uint32_t &hits)
{
    uint64_t ek = (uint64_t)e.x | ((uint64_t)e.y << 32);
    if (ek == q && (int)e.w <= max_freq)
        agg_add(iv, agg, e.z, hits);
}

// Pre-filter stage shared by the probe flavours: clears the `valid` bit of every k-mer the L2/Infinity-Cache
// resident filter proves absent from the index (Bloom filter keyed by the k-mer, or per-bucket bitmap keyed
// by bucket and fingerprint bits).  One 4-byte access per k-mer, all U in flight together.
template <int U>
__device__ __forceinline__ uint32_t filter_stage(const IndexView &iv, const uint64_t (&q)[U],
                                                 const uint64_t (&h)[U], uint32_t valid)
{
    if (iv.bloom_words) {
        // Word-blocked Bloom filter: the hash of the k-mer picks one 32-bit word and two bits inside it.
        uint32_t w[U], need[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t hs = q[u] * 0x9E3779B97F4A7C15ull;
            const uint32_t wi = (uint32_t)(((hs >> 32) * (uint64_t)iv.bloom_words) >> 32);
            need[u] = (1u << ((hs >> 7) & 31u)) | (1u << ((hs >> 12) & 31u));
            w[u] = ((valid >> u) & 1u) ? iv.occ[wi] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if ((w[u] & need[u]) != need[u])
                valid &= ~(1u << u);
    } else {
        // Per-bucket bitmap with 2^occ_shift bits per bucket: an entry sets the bit selected by the low
        // bits of its k-mer's fingerprint, so a k-mer that is not in the index passes a single-entry bucket
        // only 1 / 2^occ_shift of the time.
        uint64_t bit[U];
        uint32_t w[U];
        const uint32_t sub = (1u << iv.occ_shift) - 1u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bit[u] = (h[u] << iv.occ_shift) | (kmer_fp16(q[u]) & sub);
            w[u] = ((valid >> u) & 1u) ? iv.occ[bit[u] >> 5] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (!((w[u] >> (bit[u] & 31u)) & 1u))
                valid &= ~(1u << u);
    }
    return valid;
}

// The probe of mapper.pyx:53-69 for U k-mers per lane.  All U bucket gathers are in flight before
// any is consumed; empty and single-entry buckets (the common cases) finish there.  Buckets with
// two or more entries (hash collisions, k-mers present under several nodes) then load their first
// two entries together and walk the rest.
template <int U, bool FILTER>
__device__ __forceinline__ void probe_batch_impl(const IndexView &iv, NodeAgg &agg, LaneStats &ls,
                                                 const uint64_t (&q)[U], uint32_t valid, int max_freq)
{
    uint32_t &hits = ls.hits;
    ls.lookups += (uint32_t)__popc(valid);
    uint64_t h[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        h[u] = fastmod(q[u], iv.modulo, iv.magic);
    if (FILTER)
        valid = filter_stage<U>(iv, q, h, valid);
    uint4 b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        b[u] = make_uint4(0u, 0u, 0u, 0u);
        if ((valid >> u) & 1u) {
            if (FILTER) { // streamed once: keep the bitmap, not these lines, in L2
                u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(&iv.buckets[h[u]]));
                b[u] = make_uint4(x[0], x[1], x[2], x[3]);
            } else {
                b[u] = iv.buckets[h[u]];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t kind = b[u].w & 3u;
        if (kind == 1u) {
            uint64_t ek = (uint64_t)b[u].x | ((uint64_t)b[u].y << 32);
            if (ek == q[u] && (int)(b[u].w >> 16) <= max_freq)
                agg_add(iv, agg, b[u].z, hits);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if ((b[u].w & 3u) == 2u) {
            const uint32_t st = b[u].x, cn = b[u].y; // cn >= 2 by construction
            if (b[u].w & 4u) { // fingerprints present: skip the walk unless one of them matches
                const uint32_t f = kmer_fp16(q[u]);
                const bool may = f == (b[u].z & 0xFFFFu) || f == (b[u].z >> 16) ||
                                 (cn > 2u && f == (b[u].w >> 16));
                if (!may)
                    continue;
            }
            const uint4 e0 = iv.entries[st];
            const uint4 e1 = iv.entries[(uint64_t)st + 1];
            count_if_match(iv, agg, e0, q[u], max_freq, hits);
            count_if_match(iv, agg, e1, q[u], max_freq, hits);
            for (uint32_t j = 2; j < cn; ++j)
                count_if_match(iv, agg, iv.entries[(uint64_t)st + j], q[u], max_freq, hits);
        }
    }
}

// Same probe on the wide (32-byte) bucket layout: one L2-missing gather (A) resolves empty and single
// buckets, two-entry buckets add an L2 hit (B), only >= 3 entries (2.9 % of probes at load factor 0.5) walk
// `entries`.
template <int U, bool FILTER>
__device__ __forceinline__ void probe_batch_wide(const IndexView &iv, NodeAgg &agg, LaneStats &ls,
                                                 const uint64_t (&q)[U], uint32_t valid, int max_freq)
{
    uint32_t &hits = ls.hits;
    ls.lookups += (uint32_t)__popc(valid);
    uint64_t h[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        h[u] = fastmod(q[u], iv.modulo, iv.magic);
    if (FILTER)
        valid = filter_stage<U>(iv, q, h, valid);
    uint4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        a[u] = make_uint4(0u, 0u, 0u, 0u);
        if ((valid >> u) & 1u)
            a[u] = iv.buckets[2 * h[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        b[u] = make_uint4(0u, 0u, 0u, 0u);
        if ((a[u].w & 3u) == 2u)
            b[u] = iv.buckets[2 * h[u] + 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t kind = a[u].w & 3u;
        if (kind == 1u || kind == 2u) {
            const uint64_t ek = (uint64_t)a[u].x | ((uint64_t)a[u].y << 32);
            if (ek == q[u] && (int)(a[u].w >> 16) <= max_freq)
                agg_add(iv, agg, a[u].z, hits);
        }
        if (kind == 2u) {
            const uint64_t ek = (uint64_t)b[u].x | ((uint64_t)b[u].y << 32);
            if (ek == q[u] && (int)(b[u].w >> 16) <= max_freq)
                agg_add(iv, agg, b[u].z, hits);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if ((a[u].w & 3u) == 3u) {
            const uint32_t st = a[u].x, cn = a[u].y; // cn >= 3 by construction
            const uint4 e0 = iv.entries[st];
            const uint4 e1 = iv.entries[(uint64_t)st + 1];
            const uint4 e2 = iv.entries[(uint64_t)st + 2];
            count_if_match(iv, agg, e0, q[u], max_freq, hits);
            count_if_match(iv, agg, e1, q[u], max_freq, hits);
            count_if_match(iv, agg, e2, q[u], max_freq, hits);
            for (uint32_t j = 3; j < cn; ++j)
                count_if_match(iv, agg, iv.entries[(uint64_t)st + j], q[u], max_freq, hits);
        }
    }
}

// Probe flavours are separate kernel instantiations (not run-time branches) so that each keeps its own
// register budget: the bitmap flavour runs at 8 waves/SIMD, the wide one needs ~84 VGPRs.
enum { PROBE_NARROW = 0, PROBE_BITMAP = 1, PROBE_WIDE = 2, PROBE_WIDE_FILTER = 3 };

template <int U, int PROBE>
__device__ __forceinline__ void probe_batch(const IndexView &iv, NodeAgg &agg, LaneStats &st,
                                            const uint64_t (&q)[U], uint32_t valid, int max_freq)
{
    if (PROBE == PROBE_BITMAP)
        probe_batch_impl<U, true>(iv, agg, st, q, valid, max_freq);
    else if (PROBE == PROBE_WIDE)
        probe_batch_wide<U, false>(iv, agg, st, q, valid, max_freq);
    else if (PROBE == PROBE_WIDE_FILTER)
        probe_batch_wide<U, true>(iv, agg, st, q, valid, max_freq);
    else
        probe_batch_impl<U, false>(iv, agg, st, q, valid, max_freq);
}
