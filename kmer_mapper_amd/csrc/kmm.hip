// kmm.hip — MI355X (gfx950 / CDNA4) kernels and C-ABI host code for kmer_mapper's hot path:
// reads -> 2-bit codes -> rolling k-mer pack -> modulo hash -> bucket gather -> compare/filter ->
// per-node atomic counts.  See include/kmm.h for the boundary and DESIGN.md for the layout.
//
// Reference semantics restated (never copied): kmer_mapper/mapper.pyx:53-69 (lookup),
// kmer_mapper/util.py:71-75 (extraction), kmer_mapper/command_line_interface.py:41 (N->A).
//
// Integer / gather work: no MFMA.  The bound is random HBM accesses, so the kernels are built for
// memory-level parallelism (U independent bucket gathers in flight per lane, then U independent
// first-entry gathers) at high occupancy; reads are staged through LDS as packed 2-bit codes so
// each byte is fetched from HBM exactly once with 16-byte coalesced loads.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kmm.h"

namespace {

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(KMM_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                        hipGetErrorString(e_));                                                    \
    } while (0)

#define KMMCHK(expr)                                                                               \
    do {                                                                                           \
        int r_ = (expr);                                                                           \
        if (r_ != KMM_OK)                                                                          \
            return r_;                                                                             \
    } while (0)

// ------------------------------------------------------------------------------------------------
// device-side data layout (ours; the .npz surface is unchanged)
//   bucket h : uint4, 16 B — ONE gather resolves an empty or single-entry bucket:
//        w & 3 == 0  empty
//        w & 3 == 1  single entry stored inline: {x,y} = k-mer, z = node, w >> 16 = frequency
//        w & 3 == 2  two or more entries: x = start, y = count into `entries`
//   entry  l : uint4 {kmer_lo, kmer_hi, node, freq}, 16 B, in the index's own order (grouped by hash)
// The MI355X random-access ceiling is ~55 G L2-missing requests/s whatever their width (8 or 16 B,
// profiles/r01/gather_bench_mi355x.txt), so the layout minimises REQUESTS per k-mer, not bytes.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct IndexView {
    const uint4 *buckets;
    const uint4 *entries;
    const uint32_t *occ; // optional L2-resident occupancy bitmap (bit h = bucket h non-empty), or null
    uint32_t *counts;
    uint64_t modulo;
    uint64_t magic; // floor(2^64 / modulo) (all ones for modulo == 1)
};

struct ReadsView {
    const uint8_t *bases;
    int64_t total;             // number of base bytes
    const int64_t *offsets;    // n_reads + 1 (general path)
    int64_t n_reads;
    const int64_t *tile_first; // per tile: first r with offsets[r] > tile start (general path)
    uint64_t read_len;         // uniform path
    uint64_t read_len_magic;   // floor(2^64 / read_len)
    const uint8_t *lut;        // 256 bytes in HBM
    unsigned long long *first_bad; // [0] min position of a non-nucleotide byte, [1] of a malformed
                                   //     record line (both init ~0)
    // records mode (raw FASTQ / two-line FASTA bytes): newlines before every tile
    const uint32_t *tile_nl;   // per tile: newlines before the tile inside its super-tile (1024 tiles)
    const uint32_t *super_nl;  // per super-tile: newlines before it
    uint32_t period_mask;      // lines per record - 1 (3 for FASTQ, 1 for two-line FASTA)
    uint32_t header_char;      // '@' or '>'
};

enum { MODE_GENERAL = 0, MODE_UNIFORM = 1, MODE_RECORDS = 2 };

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

// ------------------------------------------------------------------------------------------------
// Node-count accumulation (mapper.pyx:68: node_counts[nodes[l]] += 1).
// Real graph indexes map many k-mers to few nodes, so hits are first aggregated in a small
// direct-mapped table in LDS that lives as long as the workgroup: a hit claims the slot of its
// node (ds_cmpst) and bumps the slot's counter (ds_add); a hit whose slot belongs to another node
// falls through to one global atomicAdd.  The table is flushed with one global atomicAdd per used
// slot when the workgroup retires.  uint32 wrap-around is preserved (sums of sums mod 2^32).
// ------------------------------------------------------------------------------------------------
constexpr int AGG_LOG_SLOTS = 11;
constexpr int AGG_SLOTS = 1 << AGG_LOG_SLOTS;
constexpr uint32_t AGG_EMPTY = 0xFFFFFFFFu; // node ids are < 2^31

struct NodeAgg {
    uint32_t key[AGG_SLOTS];
    uint32_t val[AGG_SLOTS];
};

__device__ __forceinline__ void agg_init(NodeAgg &agg)
{
    for (int i = threadIdx.x; i < AGG_SLOTS; i += blockDim.x) {
        agg.key[i] = AGG_EMPTY;
        agg.val[i] = 0;
    }
}

__device__ __forceinline__ void agg_add(const IndexView &iv, NodeAgg &agg, uint32_t node)
{
    const uint32_t slot = (node * 2654435761u) >> (32 - AGG_LOG_SLOTS);
    const uint32_t prev = atomicCAS(&agg.key[slot], AGG_EMPTY, node);
    if (prev == AGG_EMPTY || prev == node)
        atomicAdd(&agg.val[slot], 1u);
    else
        atomicAdd(&iv.counts[node], 1u);
}

// Call after a __syncthreads() that follows the workgroup's last agg_add.
__device__ __forceinline__ void agg_flush(const IndexView &iv, NodeAgg &agg)
{
    for (int i = threadIdx.x; i < AGG_SLOTS; i += blockDim.x) {
        const uint32_t v = agg.val[i];
        if (v)
            atomicAdd(&iv.counts[agg.key[i]], v);
    }
}

// mapper.pyx:60-68 for one entry.
__device__ __forceinline__ void count_if_match(const IndexView &iv, NodeAgg &agg, uint4 e, uint64_t q,
                                               int max_freq)
{
    uint64_t ek = (uint64_t)e.x | ((uint64_t)e.y << 32);
    if (ek == q && (int)e.w <= max_freq)
        agg_add(iv, agg, e.z);
}

// The probe of mapper.pyx:53-69 for U k-mers per lane.  All U bucket gathers are in flight before
// any is consumed; empty and single-entry buckets (the common cases) finish there.  Buckets with
// two or more entries (hash collisions, k-mers present under several nodes) then load their first
// two entries together and walk the rest.
template <int U, bool FILTER>
__device__ __forceinline__ void probe_batch_impl(const IndexView &iv, NodeAgg &agg,
                                                 const uint64_t (&q)[U], uint32_t valid, int max_freq)
{
    uint64_t h[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        h[u] = fastmod(q[u], iv.modulo, iv.magic);
    if (FILTER) {
        // Small indexes: one bit per bucket fits the XCD's L2 (4 MiB), and an L2 hit is ~4.6x cheaper
        // than the HBM request it saves for every k-mer whose bucket is empty.
        uint32_t w[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            w[u] = ((valid >> u) & 1u) ? iv.occ[h[u] >> 5] : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (!((w[u] >> (h[u] & 31u)) & 1u))
                valid &= ~(1u << u);
    }
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
                agg_add(iv, agg, b[u].z);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if ((b[u].w & 3u) == 2u) {
            const uint32_t st = b[u].x, cn = b[u].y; // cn >= 2 by construction
            const uint4 e0 = iv.entries[st];
            const uint4 e1 = iv.entries[(uint64_t)st + 1];
            count_if_match(iv, agg, e0, q[u], max_freq);
            count_if_match(iv, agg, e1, q[u], max_freq);
            for (uint32_t j = 2; j < cn; ++j)
                count_if_match(iv, agg, iv.entries[(uint64_t)st + j], q[u], max_freq);
        }
    }
}

template <int U>
__device__ __forceinline__ void probe_batch(const IndexView &iv, NodeAgg &agg, const uint64_t (&q)[U],
                                            uint32_t valid, int max_freq)
{
    if (iv.occ) // wave-uniform
        probe_batch_impl<U, true>(iv, agg, q, valid, max_freq);
    else
        probe_batch_impl<U, false>(iv, agg, q, valid, max_freq);
}

// ------------------------------------------------------------------------------------------------
// K2: operator-level lookup, uint64 k-mers already in HBM (drop-in for map_kmers_to_graph_index).
// ------------------------------------------------------------------------------------------------
template <int U>
__global__ void __launch_bounds__(256) k_map_kmers(const uint64_t *__restrict__ kmers, int64_t n,
                                                   IndexView iv, int max_freq, int also_rc, int k)
{
    __shared__ NodeAgg agg;
    agg_init(agg);
    __syncthreads();
    const int64_t span = (int64_t)256 * U;
    for (int64_t base = (int64_t)blockIdx.x * span; base < n; base += (int64_t)gridDim.x * span) {
        uint64_t q[U];
        uint32_t valid = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = base + (int64_t)u * 256 + threadIdx.x;
            q[u] = 0;
            if (i < n) {
                q[u] = kmers[i];
                valid |= 1u << u;
            }
        }
        probe_batch<U>(iv, agg, q, valid, max_freq);
        if (also_rc) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                q[u] = revcomp(q[u], k);
            probe_batch<U>(iv, agg, q, valid, max_freq);
        }
    }
    __syncthreads();
    agg_flush(iv, agg);
}

// ------------------------------------------------------------------------------------------------
// Tile front end shared by every kernel that starts from read bytes.  A workgroup (4 wavefronts)
// owns tiles of T = 256*S consecutive base positions of the chunk's flat byte stream:
//   1. 16-byte coalesced loads of the T + 48 bytes the tile's windows can touch; each byte goes
//      through the 256-entry LDS lookup table and 16 codes are packed into one 32-bit LDS word;
//   2. read starts that fall inside the tile are marked in an LDS bitset (general path) so that
//      no window spans two reads (bionumpy's ragged windowing, util.py:72);
//   3. each lane takes S consecutive positions: three LDS words give it S+31 bases in a 128-bit
//      register window, and successive k-mers are 2-bit funnel shifts of that window
//      (first base in the lowest bits).
// Returns the lane's S k-mers and the bitmask of those that are real windows.
// ------------------------------------------------------------------------------------------------
template <int S>
struct TileSmem {
    static constexpr int T = 256 * S;
    static constexpr int NV = T / 16 + 3; // 16-base words staged per tile (T + 48 positions)
    static constexpr int NB = T / 32 + 3; // 32-position words of the read-start bitset
    uint8_t lut[256];
    uint32_t codes[NV + 1];
    uint32_t bits[NB + 1];
};

struct TileConst {
    uint64_t kmask; // low 2k bits
    uint64_t bmask; // read starts in (p, p+k-1] kill the window at p
    bool aligned;   // bases pointer is 16-byte aligned
};

__device__ __forceinline__ TileConst tile_const(const ReadsView &rv, int k)
{
    TileConst c;
    c.kmask = (1ull << (2 * k)) - 1ull; // k <= 31
    c.bmask = (1ull << (k - 1)) - 1ull;
    c.aligned = (((uintptr_t)rv.bases) & 15u) == 0;
    return c;
}

// SWAR: 0x80 in every byte of x that equals the byte value c (exact, no cross-byte carries).
__device__ __forceinline__ uint32_t bytes_equal(uint32_t x, uint32_t c)
{
    const uint32_t y = x ^ (c * 0x01010101u);
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

// 0x80-per-byte flags of four bytes -> 4-bit mask (bit i = byte i).
__device__ __forceinline__ uint32_t flags_to_bits(uint32_t f)
{
    return ((f >> 7) & 1u) | ((f >> 14) & 2u) | ((f >> 21) & 4u) | ((f >> 28) & 8u);
}

template <int S, int MODE>
__device__ __forceinline__ uint32_t tile_kmers(const ReadsView &rv, const TileConst &tc, int64_t tile,
                                               int k, TileSmem<S> &sm, uint64_t (&q)[S])
{
    constexpr bool UNIFORM = MODE == MODE_UNIFORM;
    constexpr bool RECORDS = MODE == MODE_RECORDS;
    constexpr int T = TileSmem<S>::T;
    constexpr int NV = TileSmem<S>::NV;
    constexpr int NB = TileSmem<S>::NB;
    static_assert(NV <= 256, "one staged 16-byte vector per thread");
    const int tid = threadIdx.x;
    const int64_t total = rv.total;
    const int64_t t0 = tile * T;
    if (MODE == MODE_GENERAL)
        for (int i = tid; i < NB + 1; i += 256)
            sm.bits[i] = 0;
    __syncthreads(); // LUT visible; bitset cleared; previous tile's LDS readers are done

    if (RECORDS) {
        // ---- records mode, stage 1: raw file bytes.  A byte is a base iff it lies on the sequence
        // line of its record (line index mod period == 1) and is not a line terminator; every other
        // byte is a "break" that no window may contain, so k-mers never leave their read.
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        uint32_t nl = 0, cr = 0; // 16-bit masks: byte i is '\n' / '\r'
        const int v = tid;
        const int64_t p = t0 + (int64_t)v * 16;
        if (v < NV) {
            if (tc.aligned && p + 16 <= total) {
                u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rv.bases + p));
                w[0] = x[0]; w[1] = x[1]; w[2] = x[2]; w[3] = x[3];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int64_t pp = p + i * 4 + j;
                        uint32_t c = (pp < total) ? rv.bases[pp] : 0u;
                        acc |= c << (8 * j);
                    }
                    w[i] = acc;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                nl |= flags_to_bits(bytes_equal(w[i], 10u)) << (4 * i);
                cr |= flags_to_bits(bytes_equal(w[i], 13u)) << (4 * i);
            }
            sm.codes[v] = (uint32_t)__popc(nl); // borrowed as the per-vector newline count
        }
        __syncthreads();
        uint32_t brk = 0, code = 0;
        int bad = -1, malformed = -1;
        if (v < NV) {
            uint32_t line0 = rv.super_nl[tile >> 10] + rv.tile_nl[tile]; // newlines before the tile
            for (int i = 0; i < v; ++i)
                line0 += sm.codes[i];
            // first byte of a line: preceded by '\n' (or the very first byte of the chunk)
            const uint32_t prev_nl = (p == 0) ? 1u : (p - 1 < total ? (rv.bases[p - 1] == 10u) : 0u);
            const uint32_t first = ((nl << 1) | prev_nl) & 0xFFFFu;
#pragma unroll
            for (int i = 15; i >= 0; --i) {
                const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                const uint32_t line = line0 + (uint32_t)__popc(nl & ((1u << i) - 1u));
                const uint32_t phase = line & rv.period_mask;
                const bool term = ((nl | cr) >> i) & 1u;
                const bool is_seq = phase == 1u && !term && p + i < total;
                const uint32_t l = sm.lut[c];
                if (is_seq && l == 0xFFu)
                    bad = i;
                if (((first >> i) & 1u) && p + i < total &&
                    ((phase == 0u && c != rv.header_char) || (phase == 2u && c != '+')))
                    malformed = i;
                brk |= (is_seq ? 0u : 1u) << i;
                code |= (l & 3u) << (2 * i);
            }
        }
        __syncthreads(); // every thread has read the borrowed per-vector counts
        if (v < NV) {
            sm.codes[v] = code;
            reinterpret_cast<uint16_t *>(sm.bits)[v] = (uint16_t)brk;
            if (bad >= 0)
                atomicMin(&rv.first_bad[0], (unsigned long long)(p + bad));
            if (malformed >= 0)
                atomicMin(&rv.first_bad[1], (unsigned long long)(p + malformed));
        }
        __syncthreads();
    } else {

    // ---- stage 1: bytes -> 2-bit codes in LDS ----------------------------------------------
    for (int v = tid; v < NV; v += 256) {
        const int64_t p = t0 + (int64_t)v * 16;
        uint32_t w[4];
        if (tc.aligned && p + 16 <= total) {
            // streamed once: non-temporal so the read bytes do not displace index lines in L2
            u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rv.bases + p));
            w[0] = x[0]; w[1] = x[1]; w[2] = x[2]; w[3] = x[3];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t acc = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int64_t pp = p + i * 4 + j;
                    uint32_t c = (pp < total) ? rv.bases[pp] : 0u;
                    acc |= c << (8 * j);
                }
                w[i] = acc;
            }
        }
        uint32_t code = 0;
        int bad = -1;
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            uint32_t l = sm.lut[c];
            if (l == 0xFFu && p + i < total)
                bad = i;
            code |= (l & 3u) << (2 * i);
        }
        sm.codes[v] = code;
        if (bad >= 0)
            atomicMin(rv.first_bad, (unsigned long long)(p + bad));
    }
    // ---- stage 2: read starts inside (t0, t0 + T + k - 2] ----------------------------------
    if (MODE == MODE_GENERAL) {
        for (int64_t r = rv.tile_first[tile] + tid; r <= rv.n_reads; r += 256) {
            int64_t o = rv.offsets[r] - t0;
            if (o > (int64_t)T + k - 2)
                break;
            if (o >= 1)
                atomicOr(&sm.bits[o >> 5], 1u << (o & 31));
        }
    }
    __syncthreads();
    } // !RECORDS

    // ---- stage 3: S consecutive windows per lane -------------------------------------------
    const int q0 = tid * S;
    const int64_t p0 = t0 + q0;
    uint64_t lo, hi;
    {
        const int w = q0 >> 4;
        const uint32_t c0 = sm.codes[w], c1 = sm.codes[w + 1], c2 = sm.codes[w + 2];
        const int sh = (q0 & 15) * 2;
        lo = ((uint64_t)c1 << 32) | c0;
        hi = c2;
        if (sh) {
            lo = (lo >> sh) | (hi << (64 - sh));
            hi >>= sh;
        }
    }
    uint32_t valid = 0;
    if (UNIFORM) {
        uint64_t o;
        (void)fastdiv((uint64_t)p0, rv.read_len, rv.read_len_magic, &o);
#pragma unroll
        for (int j = 0; j < S; ++j) {
            uint64_t oj = o + j;
            if (oj >= rv.read_len)
                oj -= rv.read_len;
            if (oj + k <= rv.read_len && p0 + j < total)
                valid |= 1u << j;
        }
    } else {
        const int sw = q0 >> 5, off = q0 & 31;
        uint64_t B = ((uint64_t)sm.bits[sw + 1] << 32) | sm.bits[sw];
        if (off)
            B = (B >> off) | ((uint64_t)sm.bits[sw + 2] << (64 - off));
        if (RECORDS) { // no break byte inside [p, p+k-1]
            const uint64_t wmask = (1ull << k) - 1ull;
#pragma unroll
            for (int j = 0; j < S; ++j)
                if (((B >> j) & wmask) == 0 && p0 + j + k <= total)
                    valid |= 1u << j;
        } else {       // no read start inside (p, p+k-1]
#pragma unroll
            for (int j = 0; j < S; ++j)
                if (((B >> (j + 1)) & tc.bmask) == 0 && p0 + j + k <= total)
                    valid |= 1u << j;
        }
    }
#pragma unroll
    for (int j = 0; j < S; ++j)
        q[j] = (j == 0 ? lo : ((lo >> (2 * j)) | (hi << (64 - 2 * j)))) & tc.kmask;
    return valid;
}

// ------------------------------------------------------------------------------------------------
// K1 (direct path): fused reads -> counts, every probe goes to HBM.  Used for small batches and for
// indexes whose hash space cannot be cut into L2-sized partitions.
// ------------------------------------------------------------------------------------------------
template <int S, int MODE>
__global__ void __launch_bounds__(256) k_map_reads(ReadsView rv, IndexView iv, int k, int max_freq,
                                                   int also_rc, int64_t tile_begin, int64_t tile_end)
{
    __shared__ TileSmem<S> sm;
    __shared__ NodeAgg agg;
    sm.lut[threadIdx.x] = rv.lut[threadIdx.x];
    agg_init(agg); // ordered before the first agg_add by the barriers inside tile_kmers
    const TileConst tc = tile_const(rv, k);
    for (int64_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
        uint64_t q[S];
        const uint32_t valid = tile_kmers<S, MODE>(rv, tc, tile, k, sm, q);
        if (__builtin_amdgcn_ballot_w64(valid != 0)) {
            probe_batch<S>(iv, agg, q, valid, max_freq);
            if (also_rc) {
#pragma unroll
                for (int j = 0; j < S; ++j)
                    q[j] = revcomp(q[j], k);
                probe_batch<S>(iv, agg, q, valid, max_freq);
            }
        }
    }
    __syncthreads();
    agg_flush(iv, agg);
}

// ------------------------------------------------------------------------------------------------
// Partitioned path.  Random probes that miss L2 are capped at ~55 G requests/s on MI355X while
// L2-resident probes run ~4.6x faster, so large batches are first grouped by hash range
// (partition = h >> shift, each partition's bucket-table slice ~1 MiB, i.e. L2-resident) and then
// probed partition by partition, each XCD working on its own partitions so that the slice stays in
// that XCD's 4 MiB L2.  Per sub-batch, all streaming except the L2-local gathers:
//   k_part_hist     reads -> k-mers per (partition, workgroup): every workgroup owns a fixed set
//                   of tiles (grid-stride), so its histogram row is private — no global atomics
//   k_part_scan1/2  exclusive scan over (partition, workgroup) -> a private, exactly sized output
//                   range per workgroup inside every partition
//   k_part_scatter  same tiles again: k-mers are counting-sorted by partition inside LDS so that
//                   each partition's run leaves the workgroup as contiguous 8-byte stores at the
//                   workgroup's private cursor (kept in LDS)
//   k_part_probe    workgroup b takes chunks of the k-mers of XCD (b mod 8) — partitions are laid
//                   out XCD-major, p mod 8 = XCD — in lock step with the other workgroups of that
//                   XCD.  blockIdx mod 8 is where the dispatcher has been observed to place a
//                   workgroup; it is used for L2 affinity only — every chunk is processed exactly
//                   once whatever the placement.
// ------------------------------------------------------------------------------------------------
constexpr int KMM_MAX_PARTS = 1024;
constexpr int KMM_N_XCD = 8;
constexpr int KMM_CHUNK = 2048;    // k-mers per probe work item (256 lanes x 8)
constexpr int KMM_PART_GRID = 2048; // workgroups of the hist / scatter kernels (fixed: rows of wg_hist)

struct PartView {
    int shift; // partition = hash >> shift
    int P;     // number of partitions, <= KMM_MAX_PARTS
    int PX;    // partition slots per XCD = ceil(P / 8); slot(p) = (p % 8) * PX + p / 8
    uint32_t *wg_hist;  // [8*PX][KMM_PART_GRID] k-mers per (slot, workgroup); after scan1: exclusive
                        //                       prefix over the workgroups of the slot
    uint32_t *slot_tot; // [8*PX]     k-mers per slot
    uint32_t *slot_off; // [8*PX + 1] exclusive prefix of slot_tot (XCD x owns [x*PX, (x+1)*PX))
    uint64_t *kmers;    // grouped k-mers of the sub-batch
};

__device__ __forceinline__ uint32_t slot_of(const IndexView &iv, const PartView &pv, uint64_t q)
{
    const uint32_t p = (uint32_t)(fastmod(q, iv.modulo, iv.magic) >> pv.shift);
    return (p & (KMM_N_XCD - 1)) * pv.PX + (p >> 3);
}

template <int S, int MODE>
__global__ void __launch_bounds__(256) k_part_hist(ReadsView rv, IndexView iv, int k, int also_rc,
                                                   PartView pv, int64_t tile_begin, int64_t tile_end)
{
    __shared__ TileSmem<S> sm;
    __shared__ uint32_t s_hist[KMM_MAX_PARTS + KMM_N_XCD];
    const int n_slots = KMM_N_XCD * pv.PX;
    sm.lut[threadIdx.x] = rv.lut[threadIdx.x];
    for (int i = threadIdx.x; i < n_slots; i += 256)
        s_hist[i] = 0;
    const TileConst tc = tile_const(rv, k);
    for (int64_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
        uint64_t q[S];
        const uint32_t valid = tile_kmers<S, MODE>(rv, tc, tile, k, sm, q);
#pragma unroll
        for (int j = 0; j < S; ++j)
            if ((valid >> j) & 1u) {
                atomicAdd(&s_hist[slot_of(iv, pv, q[j])], 1u);
                if (also_rc)
                    atomicAdd(&s_hist[slot_of(iv, pv, revcomp(q[j], k))], 1u);
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_slots; i += 256)
        pv.wg_hist[(size_t)i * KMM_PART_GRID + blockIdx.x] = s_hist[i];
}

// One workgroup per slot: exclusive scan of the slot's KMM_PART_GRID per-workgroup counts, in place.
__global__ void __launch_bounds__(256) k_part_scan1(PartView pv)
{
    __shared__ uint32_t s_wave[4];
    constexpr int PER = KMM_PART_GRID / 256;
    uint32_t *row = pv.wg_hist + (size_t)blockIdx.x * KMM_PART_GRID;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = row[tid * PER + i];
        sum += v[i];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d);
        if (lane >= d)
            inc += o;
    }
    if (lane == 63)
        s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w)
        base += s_wave[w];
    uint32_t run = base + inc - sum;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        row[tid * PER + i] = run;
        run += v[i];
    }
    if (tid == 255)
        pv.slot_tot[blockIdx.x] = run;
}

// One workgroup of 1024 threads: exclusive scan of the (<= 1032) slot totals.
__global__ void __launch_bounds__(1024) k_part_scan2(PartView pv)
{
    __shared__ uint32_t s_a[2048];
    const int n_slots = KMM_N_XCD * pv.PX;
    const int t = threadIdx.x;
    const uint32_t c0 = t < n_slots ? pv.slot_tot[t] : 0u;
    const uint32_t c1 = t + 1024 < n_slots ? pv.slot_tot[t + 1024] : 0u;
    s_a[t] = c0;
    s_a[t + 1024] = c1;
    __syncthreads();
    for (int d = 1; d < 2048; d <<= 1) { // Hillis-Steele inclusive scan over 2048 slots
        uint32_t v0 = t >= d ? s_a[t - d] : 0u;
        uint32_t v1 = s_a[t + 1024 - d];
        __syncthreads();
        s_a[t] += v0;
        s_a[t + 1024] += v1;
        __syncthreads();
    }
    if (t < n_slots)
        pv.slot_off[t] = s_a[t] - c0;
    if (t + 1024 < n_slots)
        pv.slot_off[t + 1024] = s_a[t + 1024] - c1;
    if (t == 0)
        pv.slot_off[n_slots] = s_a[2047];
}

// Exclusive scan of s_cnt[0..n) into s_loc[0..n) (n <= 1280) by one 256-thread workgroup; returns
// the total.
__device__ __forceinline__ uint32_t block_excl_scan(const uint32_t *s_cnt, uint32_t *s_loc, int n,
                                                    uint32_t *s_wave)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = 5; // 256 x 5 = 1280 >= KMM_MAX_PARTS + 8
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = tid * PER + i;
        v[i] = p < n ? s_cnt[p] : 0u;
        sum += v[i];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d);
        if (lane >= d)
            inc += o;
    }
    __syncthreads(); // s_wave may still be read by the previous call
    if (lane == 63)
        s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w)
        base += s_wave[w];
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    uint32_t run = base + inc - sum;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = tid * PER + i;
        if (p < n)
            s_loc[p] = run;
        run += v[i];
    }
    __syncthreads();
    return total;
}

template <int S, int MODE>
__global__ void __launch_bounds__(256) k_part_scatter(ReadsView rv, IndexView iv, int k, int also_rc,
                                                      PartView pv, int64_t tile_begin,
                                                      int64_t tile_end)
{
    constexpr int T = 256 * S;
    constexpr int NS = KMM_MAX_PARTS + KMM_N_XCD;
    __shared__ TileSmem<S> sm;
    __shared__ uint32_t s_cur[NS]; // this workgroup's next free slot per partition (private range)
    __shared__ uint32_t s_cnt[NS]; // k-mers of this tile per partition
    __shared__ uint32_t s_loc[NS]; // where the partition's run starts in s_km
    __shared__ uint32_t s_wave[4];
    __shared__ uint64_t s_km[T];
    __shared__ uint16_t s_pd[T];
    const int tid = threadIdx.x;
    const int n_slots = KMM_N_XCD * pv.PX;
    sm.lut[tid] = rv.lut[tid];
    for (int i = tid; i < n_slots; i += 256)
        s_cur[i] = pv.slot_off[i] + pv.wg_hist[(size_t)i * KMM_PART_GRID + blockIdx.x];
    const TileConst tc = tile_const(rv, k);
    for (int64_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
        uint64_t q[S];
        const uint32_t valid = tile_kmers<S, MODE>(rv, tc, tile, k, sm, q);
        for (int round = 0; round < (also_rc ? 2 : 1); ++round) {
            if (round == 1) {
#pragma unroll
                for (int j = 0; j < S; ++j)
                    q[j] = revcomp(q[j], k);
            }
            for (int i = tid; i < n_slots; i += 256)
                s_cnt[i] = 0;
            __syncthreads(); // also: the previous round's readers of s_km / s_pd / s_loc are done
            uint16_t pid[S], rk[S];
#pragma unroll
            for (int j = 0; j < S; ++j) {
                pid[j] = 0;
                rk[j] = 0;
                if ((valid >> j) & 1u) {
                    pid[j] = (uint16_t)slot_of(iv, pv, q[j]);
                    rk[j] = (uint16_t)atomicAdd(&s_cnt[pid[j]], 1u);
                }
            }
            __syncthreads();
            const uint32_t n_tile = block_excl_scan(s_cnt, s_loc, n_slots, s_wave);
#pragma unroll
            for (int j = 0; j < S; ++j)
                if ((valid >> j) & 1u) {
                    const uint32_t pos = s_loc[pid[j]] + rk[j];
                    s_km[pos] = q[j];
                    s_pd[pos] = pid[j];
                }
            __syncthreads();
            for (uint32_t i = tid; i < n_tile; i += 256) {
                const uint32_t sl = s_pd[i];
                pv.kmers[s_cur[sl] + (i - s_loc[sl])] = s_km[i];
            }
            __syncthreads();
            for (int i = tid; i < n_slots; i += 256)
                s_cur[i] += s_cnt[i];
        }
    }
}

template <int U>
__global__ void __launch_bounds__(256) k_part_probe(IndexView iv, PartView pv, int max_freq)
{
    static_assert(256 * U == KMM_CHUNK, "chunk = one k-mer per lane per unroll slot");
    __shared__ NodeAgg agg;
    agg_init(agg);
    __syncthreads();
    const int tid = threadIdx.x;
    const int x = blockIdx.x & (KMM_N_XCD - 1); // expected XCD of this workgroup (speed only)
    const uint32_t j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const uint32_t begin_x = pv.slot_off[x * pv.PX], end_x = pv.slot_off[(x + 1) * pv.PX];
    const uint32_t n_chunks = (end_x - begin_x + KMM_CHUNK - 1) / KMM_CHUNK;
    for (uint32_t c = j; c < n_chunks; c += nj) {
        const uint32_t begin = begin_x + c * KMM_CHUNK;
        uint64_t q[U];
        uint32_t valid = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = begin + u * 256 + tid;
            q[u] = 0;
            if (i < end_x) {
                q[u] = __builtin_nontemporal_load(&pv.kmers[i]);
                valid |= 1u << u;
            }
        }
        probe_batch<U>(iv, agg, q, valid, max_freq);
    }
    __syncthreads();
    agg_flush(iv, agg);
}

// ------------------------------------------------------------------------------------------------
// Records mode pre-pass: newline census of a raw FASTQ / two-line FASTA chunk (tile = 1024 bytes,
// super-tile = 1024 tiles), so that every tile knows the line number of its first byte, and the
// position where the last complete record ends.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rec_count(const uint8_t *__restrict__ raw, int64_t n,
                                                   int64_t n_tiles, uint32_t *__restrict__ tile_cnt)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); // one wavefront per tile
    if (tile >= n_tiles)
        return;
    const int64_t p = tile * 1024 + lane * 16;
    uint32_t c = 0;
    if ((((uintptr_t)raw) & 15u) == 0 && p + 16 <= n) {
        u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(raw + p));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            c += (uint32_t)__popc(bytes_equal(x[i], 10u));
    } else {
        for (int i = 0; i < 16; ++i)
            if (p + i < n && raw[p + i] == 10u)
                ++c;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        c += __shfl_xor(c, d);
    if (lane == 0)
        tile_cnt[tile] = c;
}

// Inclusive Hillis-Steele scan of s_a[0..1024) by a 1024-thread workgroup.
__device__ __forceinline__ void block_scan_1024(uint32_t *s_a)
{
    const int t = threadIdx.x;
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = t >= d ? s_a[t - d] : 0u;
        __syncthreads();
        s_a[t] += v;
        __syncthreads();
    }
}

// One workgroup per super-tile: counts -> exclusive prefix inside the super-tile (in place) + total.
__global__ void __launch_bounds__(1024) k_rec_scan1(uint32_t *tile_cnt, uint32_t *super_tot)
{
    __shared__ uint32_t s_a[1024];
    const int t = threadIdx.x;
    const size_t idx = (size_t)blockIdx.x * 1024 + t;
    const uint32_t c = tile_cnt[idx];
    s_a[t] = c;
    __syncthreads();
    block_scan_1024(s_a);
    tile_cnt[idx] = s_a[t] - c;
    if (t == 1023)
        super_tot[blockIdx.x] = s_a[t];
}

// One workgroup: exclusive prefix over the super-tiles, then the byte position just after the last
// newline that completes a record (records have `period` lines).  out = {consumed, n_records, n_lines}.
__global__ void __launch_bounds__(1024) k_rec_scan2(const uint8_t *__restrict__ raw, int64_t n,
                                                    int n_super, const uint32_t *__restrict__ tile_pre,
                                                    uint32_t *super_tot, uint32_t period, int64_t *out)
{
    __shared__ uint32_t s_a[1024];
    __shared__ uint32_t s_super, s_rem, s_super_cnt;
    __shared__ int64_t s_tile;
    const int t = threadIdx.x;
    const uint32_t c = t < n_super ? super_tot[t] : 0u;
    s_a[t] = c;
    __syncthreads();
    block_scan_1024(s_a);
    const uint32_t excl = s_a[t] - c;
    const uint32_t total = s_a[1023];
    __syncthreads();
    if (t < n_super)
        super_tot[t] = excl;
    const uint32_t target = total - total % period;
    if (target == 0) { // same for every thread
        if (t == 0) {
            out[0] = 0;
            out[1] = 0;
            out[2] = total;
        }
        return;
    }
    if (t < n_super && excl < target && target <= excl + c) {
        s_super = (uint32_t)t;
        s_rem = target - excl;
        s_super_cnt = c;
    }
    __syncthreads();
    const uint32_t sup = s_super, rem = s_rem;
    const uint32_t pre = tile_pre[(size_t)sup * 1024 + t];
    const uint32_t nxt = t < 1023 ? tile_pre[(size_t)sup * 1024 + t + 1] : s_super_cnt;
    if (pre < rem && rem <= nxt)
        s_tile = (int64_t)sup * 1024 + t;
    __syncthreads();
    const int64_t tile = s_tile;
    const uint32_t r = rem - tile_pre[tile];
    const int64_t pos = tile * 1024 + t;
    const uint32_t is_nl = (pos < n && raw[pos] == 10u) ? 1u : 0u;
    s_a[t] = is_nl;
    __syncthreads();
    block_scan_1024(s_a);
    if (is_nl && s_a[t] == r)
        out[0] = pos + 1;
    if (t == 0) {
        out[1] = target / period;
        out[2] = total;
    }
}

// General path helper: for every tile, the first read index r in [0, n_reads+1] whose start lies
// strictly after the tile's first position (upper bound over the n_reads+1 offsets).
__global__ void k_tile_first(const int64_t *__restrict__ offs, int64_t n_reads, int64_t n_tiles,
                             int T, int64_t *__restrict__ out)
{
    int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= n_tiles)
        return;
    const int64_t t0 = tile * T;
    int64_t lo = 0, hi = n_reads + 1;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (offs[mid] <= t0)
            lo = mid + 1;
        else
            hi = mid;
    }
    out[tile] = lo;
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
// get_kmer_hashes_from_chunk_sequence (util.py:71-75): one lane per base position.
__global__ void k_extract_kmers(const uint8_t *__restrict__ bases, const int64_t *__restrict__ offs,
                                const int64_t *__restrict__ kmer_offs, int64_t n_reads, int k,
                                const uint8_t *__restrict__ lut, uint64_t *__restrict__ out,
                                unsigned long long *first_bad)
{
    const int64_t total = offs[n_reads];
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
         p += (int64_t)gridDim.x * blockDim.x) {
        if (lut[bases[p]] == 0xFFu)
            atomicMin(first_bad, (unsigned long long)p);
        // read containing p: last r with offs[r] <= p
        int64_t lo = 0, hi = n_reads + 1;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (offs[mid] <= p)
                lo = mid + 1;
            else
                hi = mid;
        }
        const int64_t r = lo - 1;
        if (p + k > offs[r + 1])
            continue;
        uint64_t w = 0;
        for (int j = 0; j < k; ++j)
            w |= (uint64_t)(lut[bases[p + j]] & 3u) << (2 * j);
        out[kmer_offs[r] + (p - offs[r])] = w;
    }
}

// in_graph_index (mapper.pyx:112-127): first match wins, no frequency filter.
__global__ void k_in_index(const uint64_t *__restrict__ kmers, int64_t n, IndexView iv,
                           uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t q = kmers[i];
        const uint4 b = iv.buckets[fastmod(q, iv.modulo, iv.magic)];
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
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_buckets(const int32_t *__restrict__ h2i, const int32_t *__restrict__ nk,
                               const uint64_t *__restrict__ kmers, const int32_t *__restrict__ nodes,
                               const uint16_t *__restrict__ freqs, uint64_t modulo,
                               int64_t n_entries, int64_t max_node_id, uint4 *__restrict__ buckets,
                               uint32_t *err)
{
    for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < modulo;
         h += (uint64_t)gridDim.x * blockDim.x) {
        const int32_t c = nk[h], s = h2i[h];
        uint4 b = make_uint4(0u, 0u, 0u, 0u);
        if (c > 0) { // c <= 0: `for j in range(n_local_hits)` runs zero times (mapper.pyx:58)
            if (s < 0 || (int64_t)s + c > n_entries) {
                atomicOr(err, 1u);
            } else if (c == 1) {
                const uint64_t km = kmers[s];
                int32_t nd = nodes[s];
                if (nd < 0 || (int64_t)nd > max_node_id)
                    nd = 0; // reported by k_pack_entries
                b = make_uint4((uint32_t)km, (uint32_t)(km >> 32), (uint32_t)nd,
                               ((uint32_t)freqs[s] << 16) | 1u);
            } else {
                b = make_uint4((uint32_t)s, (uint32_t)c, 0u, 2u);
            }
        }
        buckets[h] = b;
    }
}

__global__ void k_pack_entries(const uint64_t *__restrict__ kmers, const int32_t *__restrict__ nodes,
                               const uint16_t *__restrict__ freqs, int64_t n, int64_t max_node_id,
                               uint4 *__restrict__ entries, uint32_t *err)
{
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n;
         l += (int64_t)gridDim.x * blockDim.x) {
        uint64_t km = kmers[l];
        int32_t nd = nodes[l];
        if (nd < 0 || (int64_t)nd > max_node_id) {
            atomicOr(err, 2u);
            nd = 0;
        }
        entries[l] = make_uint4((uint32_t)km, (uint32_t)(km >> 32), (uint32_t)nd, (uint32_t)freqs[l]);
    }
}

// One bit per bucket: set iff the bucket holds at least one entry.
__global__ void k_build_occ(const uint4 *__restrict__ buckets, uint64_t modulo, uint32_t *__restrict__ occ)
{
    const uint64_t n_words = (modulo + 31) / 32;
    for (uint64_t wd = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wd < n_words;
         wd += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t bits = 0;
        for (int i = 0; i < 32; ++i) {
            const uint64_t h = wd * 32 + i;
            if (h < modulo && (buckets[h].w & 3u))
                bits |= 1u << i;
        }
        occ[wd] = bits;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

int ensure(DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p)
        return KMM_OK;
    if (b.p) {
        HIPCHK(hipFree(b.p)); // blocks until the device is idle: safe w.r.t. in-flight kernels
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(e == hipErrorOutOfMemory ? KMM_ERR_NOMEM : KMM_ERR_HIP,
                    "hipMalloc(%zu bytes) -> %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return KMM_OK;
}

void release(DevBuf &b)
{
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // unregistered host memory: clear the sticky error
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

uint64_t magic_for(uint64_t m)
{
    if (m <= 1)
        return ~0ull;
    return (uint64_t)(((unsigned __int128)1 << 64) / m);
}

void default_lut(uint8_t lut[256])
{
    memset(lut, 0xFF, 256);
    lut['A'] = lut['a'] = 0;
    lut['C'] = lut['c'] = 1;
    lut['G'] = lut['g'] = 2;
    lut['T'] = lut['t'] = 3;
    lut['N'] = lut['n'] = 0; // command_line_interface.py:41
}

struct Stage {
    DevBuf bases, offsets, tile_first, kmers, lut;
    hipEvent_t done = nullptr; // the last kernel that read this stage has finished
    bool used = false;
};

constexpr unsigned long long NO_BAD = ~0ull;

} // namespace

struct TimedEvent {
    hipEvent_t start, stop;
    int kernel_id;
};

struct kmm_index {
    int device = 0;
    hipStream_t stream = nullptr;      // kernels
    hipStream_t copy_stream = nullptr; // host -> HBM staging, overlaps the previous kernel
    hipEvent_t copied = nullptr;
    uint4 *buckets = nullptr;
    uint4 *entries = nullptr;
    uint32_t *occ = nullptr;           // occupancy bitmap, only for indexes small enough (see occ_max_bytes)
    bool use_occ = true;
    uint32_t *counts = nullptr;
    uint32_t *own_counts_buf = nullptr;
    uint8_t *lut_default = nullptr;
    unsigned long long *first_bad = nullptr;
    uint64_t modulo = 0, magic = 0;
    int64_t n_entries = 0, max_node_id = 0;
    Stage stage[2];
    int cur = 0;
    int n_cu = 256;
    // path selection / partitioned path state
    int path = 0;        // 0 auto, 1 direct, 2 partitioned
    int grid_per_cu = 64; // workgroups per CU of the grid-stride fused kernel
    int part_shift = 16; // 2^16 buckets x 16 B = 1 MiB bucket-table slice per partition
    DevBuf part_meta;    // hist, part_off, cursor, xcd_cum, xcd_queue
    DevBuf part_kmers;
    // timing
    bool timing = false;
    std::vector<TimedEvent> ev_used;
    std::vector<TimedEvent> ev_free;
    double ms_total[KMM_N_KERNELS] = {0};
    int64_t launches[KMM_N_KERNELS] = {0};
};

namespace {

IndexView view_of(const kmm_index *ix)
{
    IndexView v;
    v.buckets = ix->buckets;
    v.entries = ix->entries;
    v.occ = ix->use_occ ? ix->occ : nullptr;
    v.counts = ix->counts;
    v.modulo = ix->modulo;
    v.magic = ix->magic;
    return v;
}

struct ScopedTimer {
    kmm_index *ix;
    TimedEvent ev{};
    bool active = false;
    int begin(kmm_index *ix_, int kernel_id)
    {
        ix = ix_;
        if (!ix->timing)
            return KMM_OK;
        if (!ix->ev_free.empty()) {
            ev = ix->ev_free.back();
            ix->ev_free.pop_back();
        } else {
            HIPCHK(hipEventCreate(&ev.start));
            HIPCHK(hipEventCreate(&ev.stop));
        }
        ev.kernel_id = kernel_id;
        HIPCHK(hipEventRecord(ev.start, ix->stream));
        active = true;
        return KMM_OK;
    }
    int end()
    {
        if (!active)
            return KMM_OK;
        HIPCHK(hipEventRecord(ev.stop, ix->stream));
        ix->ev_used.push_back(ev);
        active = false;
        return KMM_OK;
    }
};

// Drain the streams and surface deferred device-side errors (invalid bases).
int drain(kmm_index *ix)
{
    HIPCHK(hipStreamSynchronize(ix->copy_stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    unsigned long long bad[2] = {NO_BAD, NO_BAD};
    HIPCHK(hipMemcpy(bad, ix->first_bad, sizeof bad, hipMemcpyDeviceToHost));
    if (bad[0] != NO_BAD || bad[1] != NO_BAD) {
        unsigned long long reset[2] = {NO_BAD, NO_BAD};
        HIPCHK(hipMemcpy(ix->first_bad, reset, sizeof reset, hipMemcpyHostToDevice));
        if (bad[1] != NO_BAD)
            return fail(KMM_ERR_MALFORMED,
                        "record structure violated at byte offset %llu of a mapped chunk (a record line "
                        "does not start with '@' / '+' / '>'): multi-line FASTA/FASTQ is not supported by "
                        "the GPU reader", bad[1]);
        return fail(KMM_ERR_INVALID_BASE,
                    "read byte at offset %llu of a mapped chunk is not a nucleotide under the "
                    "lookup table (the reference's DNA encoder raises here)", bad[0]);
    }
    return KMM_OK;
}

int grid_for(const kmm_index *ix, int64_t work_items, int per_cu)
{
    int64_t cap = (int64_t)ix->n_cu * per_cu;
    int64_t g = work_items < cap ? work_items : cap;
    return (int)(g < 1 ? 1 : g);
}

// Stage a host array into the given device buffer on the copy stream; device arrays pass through.
template <typename TT>
int stage_in(kmm_index *ix, DevBuf &buf, const TT *src, size_t count, const TT **dev, bool *staged)
{
    if (count == 0) {
        KMMCHK(ensure(buf, 256));
        *dev = static_cast<const TT *>(buf.p);
        return KMM_OK;
    }
    if (is_device_ptr(src)) {
        *dev = src;
        return KMM_OK;
    }
    KMMCHK(ensure(buf, count * sizeof(TT)));
    HIPCHK(hipMemcpyAsync(buf.p, src, count * sizeof(TT), hipMemcpyHostToDevice, ix->copy_stream));
    *dev = static_cast<const TT *>(buf.p);
    *staged = true;
    return KMM_OK;
}

Stage &next_stage(kmm_index *ix)
{
    Stage &s = ix->stage[ix->cur];
    ix->cur ^= 1;
    return s;
}

// Every map call: (1) the stage's buffers may be overwritten once the kernels of the call that
// last used them are done; (2) kernels may start once the copies are in; (3) borrowed host buffers
// are free again once the copies are done.
int stage_acquire(kmm_index *ix, Stage &s)
{
    if (s.used)
        HIPCHK(hipStreamWaitEvent(ix->copy_stream, s.done, 0));
    return KMM_OK;
}

int stage_copies_done(kmm_index *ix)
{
    HIPCHK(hipEventRecord(ix->copied, ix->copy_stream));
    HIPCHK(hipStreamWaitEvent(ix->stream, ix->copied, 0));
    return KMM_OK;
}

int stage_release(kmm_index *ix, Stage &s, bool staged)
{
    HIPCHK(hipEventRecord(s.done, ix->stream));
    s.used = true;
    if (staged)
        HIPCHK(hipEventSynchronize(ix->copied));
    return KMM_OK;
}

int resolve_lut(kmm_index *ix, Stage &s, const uint8_t *lut, const uint8_t **dev, bool *staged)
{
    if (!lut) {
        *dev = ix->lut_default;
        return KMM_OK;
    }
    return stage_in<uint8_t>(ix, s.lut, lut, 256, dev, staged);
}

constexpr size_t KMM_OCC_MAX_BYTES = (size_t)3 << 20; // 25 M buckets
constexpr int TILE_S = 4;
constexpr int TILE_T = 256 * TILE_S;

int part_count(const kmm_index *ix)
{
    const uint64_t per = 1ull << ix->part_shift;
    return (int)((ix->modulo + per - 1) / per);
}

bool use_partitioned(const kmm_index *ix, int64_t total_positions)
{
    (void)total_positions;
    // r01 measurements (profiles/r01/partitioned_path_ablation.md): the direct kernel is faster on
    // every configuration tried, so "auto" (0) means direct; the partitioned path is opt-in.
    if (ix->path != 2)
        return false;
    const uint64_t per = 1ull << ix->part_shift;
    return (ix->modulo + per - 1) / per <= (uint64_t)KMM_MAX_PARTS;
}

int part_view(kmm_index *ix, size_t kmer_capacity, PartView *pv)
{
    const int P = part_count(ix);
    const int PX = (P + KMM_N_XCD - 1) / KMM_N_XCD;
    const size_t n_slots = (size_t)KMM_N_XCD * PX;
    const size_t words = n_slots * KMM_PART_GRID + n_slots + (n_slots + 1);
    KMMCHK(ensure(ix->part_meta, words * 4));
    KMMCHK(ensure(ix->part_kmers, kmer_capacity * 8));
    uint32_t *w = (uint32_t *)ix->part_meta.p;
    pv->shift = ix->part_shift;
    pv->P = P;
    pv->PX = PX;
    pv->wg_hist = w; w += n_slots * KMM_PART_GRID;
    pv->slot_tot = w; w += n_slots;
    pv->slot_off = w;
    pv->kmers = (uint64_t *)ix->part_kmers.p;
    return KMM_OK;
}

template <int MODE>
int launch_map_reads(kmm_index *ix, const ReadsView &rv, int k, int max_freq, int also_rc)
{
    const IndexView iv = view_of(ix);
    const int64_t n_tiles = (rv.total + TILE_T - 1) / TILE_T;
    if (!use_partitioned(ix, rv.total)) {
        ScopedTimer tm;
        KMMCHK(tm.begin(ix, KMM_KERNEL_MAP_READS));
        hipLaunchKernelGGL((k_map_reads<TILE_S, MODE>), dim3(grid_for(ix, n_tiles, ix->grid_per_cu)), dim3(256),
                           0, ix->stream, rv, iv, k, max_freq, also_rc, (int64_t)0, n_tiles);
        HIPCHK(hipGetLastError());
        return tm.end();
    }
    // partitioned path, in sub-batches whose grouped k-mers fit 32-bit slots
    const int64_t sub_tiles = ((int64_t)1 << 29) / TILE_T; // 2^29 positions -> <= 2^30 k-mers with -r
    const int64_t max_tiles = n_tiles < sub_tiles ? n_tiles : sub_tiles;
    PartView pv;
    KMMCHK(part_view(ix, (size_t)max_tiles * TILE_T * (also_rc ? 2 : 1), &pv));
    for (int64_t t0 = 0; t0 < n_tiles; t0 += sub_tiles) {
        const int64_t t1 = t0 + sub_tiles < n_tiles ? t0 + sub_tiles : n_tiles;
        // every workgroup writes its (possibly all-zero) histogram row: the grid is always full
        const int grid = KMM_PART_GRID;
        const int n_slots = KMM_N_XCD * pv.PX;
        ScopedTimer tm;
        KMMCHK(tm.begin(ix, KMM_KERNEL_PART_HIST));
        hipLaunchKernelGGL((k_part_hist<TILE_S, MODE>), dim3(grid), dim3(256), 0, ix->stream, rv, iv,
                           k, also_rc, pv, t0, t1);
        HIPCHK(hipGetLastError());
        KMMCHK(tm.end());
        hipLaunchKernelGGL(k_part_scan1, dim3(n_slots), dim3(256), 0, ix->stream, pv);
        hipLaunchKernelGGL(k_part_scan2, dim3(1), dim3(1024), 0, ix->stream, pv);
        HIPCHK(hipGetLastError());
        KMMCHK(tm.begin(ix, KMM_KERNEL_PART_SCATTER));
        hipLaunchKernelGGL((k_part_scatter<TILE_S, MODE>), dim3(grid), dim3(256), 0, ix->stream, rv,
                           iv, k, also_rc, pv, t0, t1);
        HIPCHK(hipGetLastError());
        KMMCHK(tm.end());
        KMMCHK(tm.begin(ix, KMM_KERNEL_PART_PROBE));
        IndexView iv_nofilter = iv; // bucket gathers are L2 hits here: the bitmap would only add requests
        iv_nofilter.occ = nullptr;
        hipLaunchKernelGGL((k_part_probe<KMM_CHUNK / 256>), dim3(ix->n_cu * 8), dim3(256), 0,
                           ix->stream, iv_nofilter, pv, max_freq);
        HIPCHK(hipGetLastError());
        KMMCHK(tm.end());
    }
    return KMM_OK;
}

int check_k(int k)
{
    if (k < 1 || k > KMM_MAX_K)
        return fail(KMM_ERR_INVALID_ARG, "k=%d outside [1, %d]", k, KMM_MAX_K);
    return KMM_OK;
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char *kmm_version(void) { return "kmm 0.2.0 (gfx950)"; }

const char *kmm_last_error(void) { return g_err.c_str(); }

int kmm_device_count(int *n_devices)
{
    if (!n_devices)
        return fail(KMM_ERR_INVALID_ARG, "n_devices is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *n_devices = n;
    return KMM_OK;
}

void kmm_index_destroy(kmm_index_t *ix)
{
    if (!ix)
        return;
    (void)hipSetDevice(ix->device);
    if (ix->stream)
        (void)hipStreamSynchronize(ix->stream);
    if (ix->copy_stream)
        (void)hipStreamSynchronize(ix->copy_stream);
    for (Stage &s : ix->stage) {
        release(s.bases);
        release(s.offsets);
        release(s.tile_first);
        release(s.kmers);
        release(s.lut);
        if (s.done)
            (void)hipEventDestroy(s.done);
    }
    release(ix->part_meta);
    release(ix->part_kmers);
    for (auto &ev : ix->ev_used) {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    for (auto &ev : ix->ev_free) {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    if (ix->copied)
        (void)hipEventDestroy(ix->copied);
    if (ix->buckets)
        (void)hipFree(ix->buckets);
    if (ix->entries)
        (void)hipFree(ix->entries);
    if (ix->occ)
        (void)hipFree(ix->occ);
    if (ix->own_counts_buf)
        (void)hipFree(ix->own_counts_buf);
    if (ix->lut_default)
        (void)hipFree(ix->lut_default);
    if (ix->first_bad)
        (void)hipFree(ix->first_bad);
    if (ix->copy_stream)
        (void)hipStreamDestroy(ix->copy_stream);
    if (ix->stream)
        (void)hipStreamDestroy(ix->stream);
    delete ix;
}

static int index_create_impl(kmm_index *ix, const int32_t *h2i, const int32_t *nk,
                             const uint64_t *kmers, const int32_t *nodes, const uint16_t *freqs)
{
    const uint64_t M = ix->modulo;
    const int64_t N = ix->n_entries;
    HIPCHK(hipSetDevice(ix->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, ix->device));
    ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ix->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ix->copied, hipEventDisableTiming));
    for (Stage &s : ix->stage)
        HIPCHK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));

    HIPCHK(hipMalloc(&ix->buckets, sizeof(uint4) * (size_t)M));
    HIPCHK(hipMalloc(&ix->entries, sizeof(uint4) * (size_t)(N > 0 ? N : 1)));
    HIPCHK(hipMalloc(&ix->own_counts_buf, sizeof(uint32_t) * (size_t)(ix->max_node_id + 1)));
    ix->counts = ix->own_counts_buf;
    HIPCHK(hipMemsetAsync(ix->counts, 0, sizeof(uint32_t) * (size_t)(ix->max_node_id + 1), ix->stream));
    HIPCHK(hipMalloc(&ix->lut_default, 256));
    HIPCHK(hipMalloc(&ix->first_bad, 2 * sizeof(unsigned long long)));
    uint8_t lut[256];
    default_lut(lut);
    HIPCHK(hipMemcpy(ix->lut_default, lut, 256, hipMemcpyHostToDevice));
    unsigned long long nb[2] = {NO_BAD, NO_BAD};
    HIPCHK(hipMemcpy(ix->first_bad, nb, sizeof nb, hipMemcpyHostToDevice));

    // raw arrays -> HBM (temporary), repack + validate on the GPU
    DevBuf d_h2i, d_nk, d_km, d_nd, d_fr, d_err;
    bool staged = false;
    const int32_t *p_h2i = nullptr, *p_nk = nullptr, *p_nd = nullptr;
    const uint64_t *p_km = nullptr;
    const uint16_t *p_fr = nullptr;
    int rc = KMM_OK;
    do {
        if ((rc = stage_in<int32_t>(ix, d_h2i, h2i, (size_t)M, &p_h2i, &staged))) break;
        if ((rc = stage_in<int32_t>(ix, d_nk, nk, (size_t)M, &p_nk, &staged))) break;
        if ((rc = stage_in<uint64_t>(ix, d_km, kmers, (size_t)N, &p_km, &staged))) break;
        if ((rc = stage_in<int32_t>(ix, d_nd, nodes, (size_t)N, &p_nd, &staged))) break;
        if ((rc = stage_in<uint16_t>(ix, d_fr, freqs, (size_t)N, &p_fr, &staged))) break;
        if ((rc = ensure(d_err, 4))) break;
    } while (0);
    if (rc == KMM_OK) {
        hipError_t e = hipMemsetAsync(d_err.p, 0, 4, ix->copy_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->copy_stream);
        if (e != hipSuccess)
            rc = fail(KMM_ERR_HIP, "index upload: %s", hipGetErrorString(e));
    }
    uint32_t err = 0;
    if (rc == KMM_OK) {
        hipLaunchKernelGGL(k_pack_buckets, dim3(grid_for(ix, (int64_t)((M + 255) / 256), 16)),
                           dim3(256), 0, ix->stream, p_h2i, p_nk, p_km, p_nd, p_fr, M, N,
                           ix->max_node_id, ix->buckets, (uint32_t *)d_err.p);
        if (N > 0)
            hipLaunchKernelGGL(k_pack_entries, dim3(grid_for(ix, (N + 255) / 256, 16)), dim3(256),
                               0, ix->stream, p_km, p_nd, p_fr, N, ix->max_node_id, ix->entries,
                               (uint32_t *)d_err.p);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        if (e == hipSuccess) e = hipMemcpy(&err, d_err.p, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            rc = fail(KMM_ERR_HIP, "index repack: %s", hipGetErrorString(e));
    }
    release(d_h2i); release(d_nk); release(d_km); release(d_nd); release(d_fr); release(d_err);
    if (rc != KMM_OK)
        return rc;
    if (err & 1u)
        return fail(KMM_ERR_INDEX, "index inconsistent: a non-empty bucket "
                    "(hashes_to_index[h], n_kmers[h]) reaches outside [0, n_entries=%lld)",
                    (long long)N);
    if (err & 2u)
        return fail(KMM_ERR_INDEX, "index inconsistent: a node id lies outside [0, max_node_id=%lld]",
                    (long long)ix->max_node_id);
    // occupancy bitmap for indexes whose bitmap stays resident in one XCD's 4 MiB L2
    const size_t occ_bytes = (size_t)((M + 31) / 32) * 4;
    if (occ_bytes <= KMM_OCC_MAX_BYTES) {
        HIPCHK(hipMalloc(&ix->occ, occ_bytes));
        hipLaunchKernelGGL(k_build_occ, dim3(grid_for(ix, (int64_t)((M / 32 + 256) / 256), 16)), dim3(256),
                           0, ix->stream, ix->buckets, M, ix->occ);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ix->stream));
    }
    // default partition granularity: 1 MiB bucket-table slices, coarser if that needs > 1024 parts
    ix->part_shift = 16;
    while (((M + (1ull << ix->part_shift) - 1) >> ix->part_shift) > (uint64_t)KMM_MAX_PARTS &&
           ix->part_shift < 18)
        ix->part_shift++;
    return KMM_OK;
}

int kmm_index_create(const int32_t *hashes_to_index, const int32_t *n_kmers, uint64_t modulo,
                     const uint64_t *kmers, const int32_t *nodes, const uint16_t *frequencies,
                     int64_t n_entries, int64_t max_node_id, int device, kmm_index_t **out)
{
    if (!out)
        return fail(KMM_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!hashes_to_index || !n_kmers)
        return fail(KMM_ERR_INVALID_ARG, "hashes_to_index / n_kmers is NULL");
    if (modulo < 1)
        return fail(KMM_ERR_INVALID_ARG, "modulo must be >= 1");
    if (n_entries < 0 || max_node_id < 0)
        return fail(KMM_ERR_INVALID_ARG, "n_entries=%lld / max_node_id=%lld negative",
                    (long long)n_entries, (long long)max_node_id);
    if (n_entries > 0 && (!kmers || !nodes || !frequencies))
        return fail(KMM_ERR_INVALID_ARG, "kmers / nodes / frequencies is NULL");
    if (n_entries > 0x7FFFFFFFll)
        return fail(KMM_ERR_INVALID_ARG, "n_entries exceeds the int32 bucket offsets of the index format");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        return fail(KMM_ERR_HIP, "no HIP device available (%s): libkmm has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    }
    if (device < 0 || device >= ndev)
        return fail(KMM_ERR_INVALID_ARG, "device %d outside [0, %d)", device, ndev);
    kmm_index *ix = new kmm_index();
    ix->device = device;
    ix->modulo = modulo;
    ix->magic = magic_for(modulo);
    ix->n_entries = n_entries;
    ix->max_node_id = max_node_id;
    int rc = index_create_impl(ix, hashes_to_index, n_kmers, kmers, nodes, frequencies);
    if (rc != KMM_OK) {
        std::string keep = g_err;
        kmm_index_destroy(ix);
        g_err = keep;
        return rc;
    }
    *out = ix;
    return KMM_OK;
}

int kmm_reset_counts(kmm_index_t *ix)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipMemsetAsync(ix->counts, 0, sizeof(uint32_t) * (size_t)(ix->max_node_id + 1), ix->stream));
    return KMM_OK;
}

int kmm_bind_counts(kmm_index_t *ix, uint32_t *device_counts)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipStreamSynchronize(ix->stream));
    if (!device_counts) {
        ix->counts = ix->own_counts_buf;
        return KMM_OK;
    }
    if (!is_device_ptr(device_counts))
        return fail(KMM_ERR_INVALID_ARG, "kmm_bind_counts needs a device pointer");
    ix->counts = device_counts;
    return KMM_OK;
}

int kmm_counts_device_ptr(kmm_index_t *ix, uint32_t **out)
{
    if (!ix || !out)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    *out = ix->counts;
    return KMM_OK;
}

int kmm_synchronize(kmm_index_t *ix)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    HIPCHK(hipSetDevice(ix->device));
    return drain(ix);
}

int kmm_get_node_counts(kmm_index_t *ix, uint32_t *out)
{
    if (!ix || !out)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    HIPCHK(hipSetDevice(ix->device));
    KMMCHK(drain(ix));
    HIPCHK(hipMemcpy(out, ix->counts, sizeof(uint32_t) * (size_t)(ix->max_node_id + 1),
                     is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return KMM_OK;
}

int kmm_map_kmers(kmm_index_t *ix, const uint64_t *kmers, int64_t n, int max_freq, int also_revcomp,
                  int k)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    if (n < 0 || (n > 0 && !kmers))
        return fail(KMM_ERR_INVALID_ARG, "kmers NULL or n negative");
    if (also_revcomp)
        KMMCHK(check_k(k));
    if (n == 0)
        return KMM_OK;
    HIPCHK(hipSetDevice(ix->device));
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    bool staged = false;
    const uint64_t *d_kmers = nullptr;
    KMMCHK(stage_in<uint64_t>(ix, s.kmers, kmers, (size_t)n, &d_kmers, &staged));
    KMMCHK(stage_copies_done(ix));
    constexpr int U = 8;
    ScopedTimer tm;
    KMMCHK(tm.begin(ix, KMM_KERNEL_MAP_KMERS));
    hipLaunchKernelGGL((k_map_kmers<U>), dim3(grid_for(ix, (n + 256 * U - 1) / (256 * U), 64)),
                       dim3(256), 0, ix->stream, d_kmers, n, view_of(ix), max_freq,
                       also_revcomp ? 1 : 0, k);
    HIPCHK(hipGetLastError());
    KMMCHK(tm.end());
    return stage_release(ix, s, staged);
}

static int map_reads_common(kmm_index_t *ix, const uint8_t *bases, const int64_t *read_offsets,
                            int64_t n_reads, int64_t read_len, int k, int max_freq,
                            int also_revcomp, const uint8_t *lut)
{
    const bool uniform = (read_offsets == nullptr);
    HIPCHK(hipSetDevice(ix->device));
    int64_t total = 0;
    const bool offs_on_device = !uniform && is_device_ptr(read_offsets);
    if (uniform) {
        total = n_reads * read_len;
    } else {
        int64_t ends[2] = {0, 0};
        if (offs_on_device) {
            HIPCHK(hipMemcpy(&ends[0], read_offsets, 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(&ends[1], read_offsets + n_reads, 8, hipMemcpyDeviceToHost));
        } else {
            ends[0] = read_offsets[0];
            ends[1] = read_offsets[n_reads];
        }
        if (ends[0] != 0)
            return fail(KMM_ERR_INVALID_ARG, "read_offsets[0] must be 0 (got %lld)", (long long)ends[0]);
        if (ends[1] < 0)
            return fail(KMM_ERR_INVALID_ARG, "read_offsets[n_reads] negative");
        if (!offs_on_device)
            for (int64_t r = 0; r < n_reads; ++r)
                if (read_offsets[r + 1] < read_offsets[r])
                    return fail(KMM_ERR_INVALID_ARG, "read_offsets not non-decreasing at read %lld",
                                (long long)r);
        total = ends[1];
    }
    if (total == 0)
        return KMM_OK;
    if (!bases)
        return fail(KMM_ERR_INVALID_ARG, "bases is NULL");

    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    bool staged = false;
    ReadsView rv;
    memset(&rv, 0, sizeof rv);
    KMMCHK(stage_in<uint8_t>(ix, s.bases, bases, (size_t)total, &rv.bases, &staged));
    KMMCHK(resolve_lut(ix, s, lut, &rv.lut, &staged));
    rv.total = total;
    rv.n_reads = n_reads;
    rv.first_bad = ix->first_bad;
    const int64_t n_tiles = (total + TILE_T - 1) / TILE_T;
    // the uniform kernel's wrap-around handles one read boundary per lane: needs read_len >= S
    const bool uniform_kernel = uniform && read_len >= 16;
    if (uniform_kernel) {
        rv.read_len = (uint64_t)read_len;
        rv.read_len_magic = magic_for((uint64_t)read_len);
        KMMCHK(stage_copies_done(ix));
        KMMCHK(launch_map_reads<MODE_UNIFORM>(ix, rv, k, max_freq, also_revcomp ? 1 : 0));
    } else {
        if (uniform) {
            KMMCHK(ensure(s.offsets, (size_t)(n_reads + 1) * 8));
            hipLaunchKernelGGL(k_iota_offsets, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256),
                               0, ix->copy_stream, (int64_t *)s.offsets.p, n_reads, read_len);
            HIPCHK(hipGetLastError());
            rv.offsets = (const int64_t *)s.offsets.p;
        } else {
            KMMCHK(stage_in<int64_t>(ix, s.offsets, read_offsets, (size_t)(n_reads + 1), &rv.offsets,
                                     &staged));
        }
        KMMCHK(ensure(s.tile_first, (size_t)n_tiles * 8));
        rv.tile_first = (const int64_t *)s.tile_first.p;
        KMMCHK(stage_copies_done(ix));
        hipLaunchKernelGGL(k_tile_first, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0,
                           ix->stream, rv.offsets, n_reads, n_tiles, TILE_T, (int64_t *)s.tile_first.p);
        HIPCHK(hipGetLastError());
        KMMCHK(launch_map_reads<MODE_GENERAL>(ix, rv, k, max_freq, also_revcomp ? 1 : 0));
    }
    return stage_release(ix, s, staged);
}

int kmm_map_reads(kmm_index_t *ix, const uint8_t *bases, const int64_t *read_offsets, int64_t n_reads,
                  int k, int max_freq, int also_revcomp, const uint8_t *lut)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    KMMCHK(check_k(k));
    if (n_reads < 0)
        return fail(KMM_ERR_INVALID_ARG, "n_reads negative");
    if (n_reads == 0)
        return KMM_OK;
    if (!read_offsets)
        return fail(KMM_ERR_INVALID_ARG, "read_offsets is NULL");
    return map_reads_common(ix, bases, read_offsets, n_reads, 0, k, max_freq, also_revcomp, lut);
}

int kmm_map_reads_uniform(kmm_index_t *ix, const uint8_t *bases, int64_t n_reads, int64_t read_len,
                          int k, int max_freq, int also_revcomp, const uint8_t *lut)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    KMMCHK(check_k(k));
    if (n_reads < 0 || read_len < 0)
        return fail(KMM_ERR_INVALID_ARG, "n_reads / read_len negative");
    if (n_reads == 0 || read_len == 0)
        return KMM_OK;
    return map_reads_common(ix, bases, nullptr, n_reads, read_len, k, max_freq, also_revcomp, lut);
}

int kmm_map_records(kmm_index_t *ix, const uint8_t *raw, int64_t n_bytes, int format, int k,
                    int max_freq, int also_revcomp, const uint8_t *lut, int64_t *consumed,
                    int64_t *n_records)
{
    static_assert(TILE_T == 1024, "records mode counts newlines per 1024-byte tile");
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    KMMCHK(check_k(k));
    if (format != KMM_FORMAT_FASTQ && format != KMM_FORMAT_FASTA2)
        return fail(KMM_ERR_INVALID_ARG, "format must be KMM_FORMAT_FASTQ (4) or KMM_FORMAT_FASTA2 (2)");
    if (n_bytes < 0 || n_bytes > ((int64_t)1 << 30))
        return fail(KMM_ERR_INVALID_ARG, "n_bytes outside [0, 2^30]: cut the file into smaller chunks");
    if (consumed)
        *consumed = 0;
    if (n_records)
        *n_records = 0;
    if (n_bytes == 0)
        return KMM_OK;
    if (!raw)
        return fail(KMM_ERR_INVALID_ARG, "raw is NULL");
    HIPCHK(hipSetDevice(ix->device));
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    bool staged = false;
    ReadsView rv;
    memset(&rv, 0, sizeof rv);
    KMMCHK(stage_in<uint8_t>(ix, s.bases, raw, (size_t)n_bytes, &rv.bases, &staged));
    KMMCHK(resolve_lut(ix, s, lut, &rv.lut, &staged));
    const int64_t n_tiles = (n_bytes + TILE_T - 1) / TILE_T;
    const int n_super = (int)((n_tiles + 1023) / 1024);
    KMMCHK(ensure(s.tile_first, (size_t)n_super * 1024 * 4));
    KMMCHK(ensure(s.offsets, (size_t)n_super * 4 + 64));
    uint32_t *tile_cnt = (uint32_t *)s.tile_first.p;
    uint32_t *super_tot = (uint32_t *)s.offsets.p;
    int64_t *d_out = (int64_t *)((uint8_t *)s.offsets.p + (((size_t)n_super * 4 + 15) & ~(size_t)15));
    HIPCHK(hipMemsetAsync(tile_cnt, 0, (size_t)n_super * 1024 * 4, ix->copy_stream));
    hipLaunchKernelGGL(k_rec_count, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, ix->copy_stream,
                       rv.bases, n_bytes, n_tiles, tile_cnt);
    hipLaunchKernelGGL(k_rec_scan1, dim3(n_super), dim3(1024), 0, ix->copy_stream, tile_cnt, super_tot);
    hipLaunchKernelGGL(k_rec_scan2, dim3(1), dim3(1024), 0, ix->copy_stream, rv.bases, n_bytes, n_super,
                       tile_cnt, super_tot, (uint32_t)format, d_out);
    HIPCHK(hipGetLastError());
    int64_t out[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, ix->copy_stream));
    HIPCHK(hipStreamSynchronize(ix->copy_stream)); // the borrowed host buffer is free from here on
    if (consumed)
        *consumed = out[0];
    if (n_records)
        *n_records = out[1];
    if (out[0] > 0) {
        rv.total = out[0];
        rv.first_bad = ix->first_bad;
        rv.tile_nl = tile_cnt;
        rv.super_nl = super_tot;
        rv.period_mask = (uint32_t)format - 1u;
        rv.header_char = format == KMM_FORMAT_FASTQ ? (uint32_t)'@' : (uint32_t)'>';
        KMMCHK(stage_copies_done(ix));
        KMMCHK(launch_map_reads<MODE_RECORDS>(ix, rv, k, max_freq, also_revcomp ? 1 : 0));
    }
    return stage_release(ix, s, false);
}

int kmm_in_index(kmm_index_t *ix, const uint64_t *kmers, int64_t n, uint8_t *out)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    if (n < 0 || (n > 0 && (!kmers || !out)))
        return fail(KMM_ERR_INVALID_ARG, "NULL argument or n negative");
    if (n == 0)
        return KMM_OK;
    HIPCHK(hipSetDevice(ix->device));
    Stage &s = next_stage(ix);
    KMMCHK(stage_acquire(ix, s));
    bool staged = false;
    const uint64_t *d_kmers = nullptr;
    KMMCHK(stage_in<uint64_t>(ix, s.kmers, kmers, (size_t)n, &d_kmers, &staged));
    KMMCHK(stage_copies_done(ix));
    const bool out_dev = is_device_ptr(out);
    DevBuf tmp;
    uint8_t *d_out = out;
    if (!out_dev) {
        KMMCHK(ensure(tmp, (size_t)n));
        d_out = (uint8_t *)tmp.p;
    }
    hipLaunchKernelGGL(k_in_index, dim3(grid_for(ix, (n + 255) / 256, 32)), dim3(256), 0, ix->stream,
                       d_kmers, n, view_of(ix), d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !out_dev)
        e = hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, ix->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ix->stream);
    release(tmp);
    if (e != hipSuccess)
        return fail(KMM_ERR_HIP, "kmm_in_index: %s", hipGetErrorString(e));
    return stage_release(ix, s, staged);
}

int kmm_extract_kmers(int device, const uint8_t *bases, const int64_t *read_offsets, int64_t n_reads,
                      int k, const uint8_t *lut, uint64_t *out, int64_t n_out)
{
    KMMCHK(check_k(k));
    if (n_reads < 0 || n_out < 0)
        return fail(KMM_ERR_INVALID_ARG, "negative size");
    if (n_reads == 0)
        return n_out == 0 ? KMM_OK : fail(KMM_ERR_INVALID_ARG, "n_out != 0 for zero reads");
    if (!read_offsets)
        return fail(KMM_ERR_INVALID_ARG, "read_offsets is NULL");
    int ndev = 0;
    hipError_t e0 = hipGetDeviceCount(&ndev);
    if (e0 != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        return fail(KMM_ERR_HIP, "no HIP device available: libkmm has no CPU fallback");
    }
    HIPCHK(hipSetDevice(device));
    // host copy of the offsets for the prefix sums of per-read k-mer counts
    std::vector<int64_t> offs((size_t)n_reads + 1);
    if (is_device_ptr(read_offsets))
        HIPCHK(hipMemcpy(offs.data(), read_offsets, offs.size() * 8, hipMemcpyDeviceToHost));
    else
        memcpy(offs.data(), read_offsets, offs.size() * 8);
    if (offs[0] != 0)
        return fail(KMM_ERR_INVALID_ARG, "read_offsets[0] must be 0");
    std::vector<int64_t> koffs((size_t)n_reads + 1);
    koffs[0] = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        int64_t len = offs[r + 1] - offs[r];
        if (len < 0)
            return fail(KMM_ERR_INVALID_ARG, "read_offsets not non-decreasing at read %lld", (long long)r);
        koffs[r + 1] = koffs[r] + (len >= k ? len - k + 1 : 0);
    }
    if (koffs[n_reads] != n_out)
        return fail(KMM_ERR_INVALID_ARG, "n_out=%lld but the reads hold %lld k-mers", (long long)n_out,
                    (long long)koffs[n_reads]);
    const int64_t total = offs[n_reads];
    if (total == 0 || n_out == 0)
        return KMM_OK;
    if (!bases || !out)
        return fail(KMM_ERR_INVALID_ARG, "bases / out is NULL");

    DevBuf d_bases, d_offs, d_koffs, d_lut, d_out, d_bad;
    int rc = KMM_OK;
    hipError_t e = hipSuccess;
    const uint8_t *p_bases = bases;
    uint64_t *p_out = out;
    const bool bases_dev = is_device_ptr(bases), out_dev = is_device_ptr(out);
    uint8_t lutbuf[256];
    if (lut) {
        if (is_device_ptr(lut))
            e = hipMemcpy(lutbuf, lut, 256, hipMemcpyDeviceToHost);
        else
            memcpy(lutbuf, lut, 256);
    } else {
        default_lut(lutbuf);
    }
    unsigned long long bad = NO_BAD;
    do {
        if (e != hipSuccess) break;
        if (!bases_dev) {
            if ((rc = ensure(d_bases, (size_t)total))) break;
            if ((e = hipMemcpy(d_bases.p, bases, (size_t)total, hipMemcpyHostToDevice))) break;
            p_bases = (const uint8_t *)d_bases.p;
        }
        if (!out_dev) {
            if ((rc = ensure(d_out, (size_t)n_out * 8))) break;
            p_out = (uint64_t *)d_out.p;
        }
        if ((rc = ensure(d_offs, offs.size() * 8))) break;
        if ((rc = ensure(d_koffs, koffs.size() * 8))) break;
        if ((rc = ensure(d_lut, 256))) break;
        if ((rc = ensure(d_bad, 8))) break;
        if ((e = hipMemcpy(d_offs.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice))) break;
        if ((e = hipMemcpy(d_koffs.p, koffs.data(), koffs.size() * 8, hipMemcpyHostToDevice))) break;
        if ((e = hipMemcpy(d_lut.p, lutbuf, 256, hipMemcpyHostToDevice))) break;
        if ((e = hipMemcpy(d_bad.p, &bad, 8, hipMemcpyHostToDevice))) break;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_extract_kmers, dim3((unsigned)blocks), dim3(256), 0, 0, p_bases,
                           (const int64_t *)d_offs.p, (const int64_t *)d_koffs.p, n_reads, k,
                           (const uint8_t *)d_lut.p, p_out, (unsigned long long *)d_bad.p);
        if ((e = hipGetLastError())) break;
        if ((e = hipDeviceSynchronize())) break;
        if ((e = hipMemcpy(&bad, d_bad.p, 8, hipMemcpyDeviceToHost))) break;
        if (!out_dev)
            if ((e = hipMemcpy(out, d_out.p, (size_t)n_out * 8, hipMemcpyDeviceToHost))) break;
    } while (0);
    release(d_bases); release(d_offs); release(d_koffs); release(d_lut); release(d_out); release(d_bad);
    if (rc != KMM_OK)
        return rc;
    if (e != hipSuccess)
        return fail(KMM_ERR_HIP, "kmm_extract_kmers: %s", hipGetErrorString(e));
    if (bad != NO_BAD)
        return fail(KMM_ERR_INVALID_BASE, "read byte at offset %llu is not a nucleotide under the "
                    "lookup table (the reference's DNA encoder raises here)", bad);
    return KMM_OK;
}

int kmm_set_timing(kmm_index_t *ix, int enabled)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    ix->timing = enabled != 0;
    return KMM_OK;
}

int kmm_get_timing(kmm_index_t *ix, int kernel_id, double *kernel_ms, int64_t *n_launches)
{
    if (!ix)
        return fail(KMM_ERR_INVALID_ARG, "idx is NULL");
    if (kernel_id < 0 || kernel_id >= KMM_N_KERNELS)
        return fail(KMM_ERR_INVALID_ARG, "kernel_id %d outside [0, %d)", kernel_id, KMM_N_KERNELS);
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipStreamSynchronize(ix->stream));
    for (auto &ev : ix->ev_used) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev.start, ev.stop));
        ix->ms_total[ev.kernel_id] += ms;
        ix->launches[ev.kernel_id] += 1;
        ix->ev_free.push_back(ev);
    }
    ix->ev_used.clear();
    if (kernel_ms)
        *kernel_ms = ix->ms_total[kernel_id];
    if (n_launches)
        *n_launches = ix->launches[kernel_id];
    ix->ms_total[kernel_id] = 0.0;
    ix->launches[kernel_id] = 0;
    return KMM_OK;
}

int kmm_set_param(kmm_index_t *ix, const char *name, int64_t value)
{
    if (!ix || !name)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipStreamSynchronize(ix->stream)); // scratch layouts depend on the knobs
    if (!strcmp(name, "path")) {
        if (value < 0 || value > 2)
            return fail(KMM_ERR_INVALID_ARG, "path must be 0 (auto), 1 (direct) or 2 (partitioned)");
        ix->path = (int)value;
    } else if (!strcmp(name, "part_shift")) {
        if (value < 4 || value > 30)
            return fail(KMM_ERR_INVALID_ARG, "part_shift outside [4, 30]");
        ix->part_shift = (int)value;
        release(ix->part_meta); // re-laid out (and re-zeroed) on next use
    } else if (!strcmp(name, "grid_per_cu")) {
        if (value < 1 || value > 1024)
            return fail(KMM_ERR_INVALID_ARG, "grid_per_cu outside [1, 1024]");
        ix->grid_per_cu = (int)value;
    } else if (!strcmp(name, "occupancy_filter")) {
        ix->use_occ = value != 0;

    } else {
        return fail(KMM_ERR_INVALID_ARG, "unknown parameter '%s'", name);
    }
    return KMM_OK;
}

int kmm_get_param(kmm_index_t *ix, const char *name, int64_t *value)
{
    if (!ix || !name || !value)
        return fail(KMM_ERR_INVALID_ARG, "NULL argument");
    if (!strcmp(name, "path"))
        *value = ix->path;
    else if (!strcmp(name, "part_shift"))
        *value = ix->part_shift;
    else if (!strcmp(name, "grid_per_cu"))
        *value = ix->grid_per_cu;
    else if (!strcmp(name, "occupancy_filter"))
        *value = (ix->use_occ && ix->occ) ? 1 : 0;
    else if (!strcmp(name, "n_partitions"))
        *value = part_count(ix);
    else if (!strcmp(name, "partitioned_available"))
        *value = part_count(ix) <= KMM_MAX_PARTS ? 1 : 0;
    else
        return fail(KMM_ERR_INVALID_ARG, "unknown parameter '%s'", name);
    return KMM_OK;
}

} // extern "C"
