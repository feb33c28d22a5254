// kmm_records.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip inside its anonymous namespace.
// Records mode pre-pass: newline census of raw FASTQ / two-line FASTA chunks.
#pragma once

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
// tile_bytes: 1024 (direct path's census) or 4096 (the radix path's compaction).
__global__ void __launch_bounds__(1024) k_rec_scan2(const uint8_t *__restrict__ raw, int64_t n,
                                                    int n_super, const uint32_t *__restrict__ tile_pre,
                                                    uint32_t *super_tot, uint32_t period, int64_t *out, int tile_bytes)
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
    uint32_t r = rem - tile_pre[tile]; // the r-th newline of the tile ends the last complete record
    for (int part = 0; part < tile_bytes / 1024; ++part) {
        const int64_t pos = tile * tile_bytes + part * 1024 + t;
        const uint32_t is_nl = (pos < n && raw[pos] == 10u) ? 1u : 0u;
        __syncthreads();
        s_a[t] = is_nl;
        __syncthreads();
        block_scan_1024(s_a);
        if (is_nl && s_a[t] == r)
            out[0] = pos + 1;
        const uint32_t in_part = s_a[1023];
        if (r <= in_part)
            break; // (uniform)
        r -= in_part;
    }
    if (t == 0) {
        out[1] = target / period;
        out[2] = total;
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-line FASTA (sequences wrapped over several lines — what `bnp.open` also reads in the reference,
// command_line_interface.py:102,109): a device-side pre-pass UNWRAPS the chunk into two-line FASTA, which the records
// mode above then maps.  A byte is dropped iff it is a line terminator ('\n', or '\r' right before one) that ends a
// SEQUENCE line and is followed by another sequence line; everything else is copied, compacted.  Whether a
// terminator ends a sequence line depends on the first byte of its line, i.e. on where the previous newline lies — a
// "position of the last newline" prefix (max) over the chunk: per 1024-byte tile, then over tiles.
//   k_ml_tile_last    last newline of every tile (-1: none)
//   k_ml_scan         prefix maximum over the tiles (single workgroup, 1024 tiles per round, carried)
//   k_ml_flags        keep flags -> kept bytes per tile (tile_cnt) + the start of the last header line of the chunk
//   (k_rec_scan1 + k_super_scan: prefix sums of the counts)
//   k_ml_scatter      the kept bytes before `limit`, compacted
// Only whole records are unwrapped: `limit` = start of the last header line (the record it opens may continue in the
// next chunk) unless the caller says the chunk is the last of the file.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ml_tile_last(const uint8_t *__restrict__ raw, int64_t n, int64_t n_tiles,
                                                      int32_t *__restrict__ tile_last)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); // one wavefront per tile
    if (tile >= n_tiles)
        return;
    const int64_t p = tile * 1024 + lane * 16;
    int32_t last = -1;
    for (int i = 0; i < 16; ++i)
        if (p + i < n && raw[p + i] == 10u)
            last = (int32_t)(lane * 16 + i);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int32_t o = __shfl_xor(last, d);
        last = o > last ? o : last;
    }
    if (lane == 0)
        tile_last[tile] = last < 0 ? -1 : (int32_t)(tile * 1024 + last); // chunks are < 2^31 bytes (pieces of 2^30)
}

// tile_prev[t] = position of the last newline before tile t (-1: none).  One workgroup walks the tiles 1024 at a time.
__global__ void __launch_bounds__(1024) k_ml_scan(const int32_t *__restrict__ tile_last, int64_t n_tiles,
                                                  int32_t *__restrict__ tile_prev)
{
    __shared__ int32_t s_a[1024];
    int32_t carry = -1;
    for (int64_t base = 0; base < n_tiles; base += 1024) {
        const int64_t t = base + threadIdx.x;
        const int32_t v = t < n_tiles ? tile_last[t] : -1;
        s_a[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) { // inclusive prefix maximum (Hillis-Steele)
            const int32_t o = (int)threadIdx.x >= d ? s_a[threadIdx.x - d] : -1;
            __syncthreads();
            if (o > s_a[threadIdx.x])
                s_a[threadIdx.x] = o;
            __syncthreads();
        }
        const int32_t excl = threadIdx.x ? s_a[threadIdx.x - 1] : -1;
        if (t < n_tiles)
            tile_prev[t] = excl > carry ? excl : carry;
        const int32_t tot = s_a[1023];
        __syncthreads();
        carry = tot > carry ? tot : carry;
    }
}

// keep flag of byte i (see above); prev_nl = position of the last newline before i (-1: none)
__device__ __forceinline__ bool ml_keep(const uint8_t *__restrict__ raw, int64_t n, int64_t i, uint32_t c, int32_t prev_nl)
{
    bool term = c == 10u;
    if (c == 13u && i + 1 < n && raw[i + 1] == 10u)
        term = true; // '\r' of a "\r\n"
    if (!term)
        return true;
    const bool header = raw[prev_nl + 1] == (uint8_t)'>';
    int64_t nx = i + 1;
    if (c == 13u)
        ++nx; // the byte after the "\r\n"
    const bool next_is_seq = nx < n && raw[nx] != (uint8_t)'>';
    return header || !next_is_seq;
}

// One wavefront per tile, 16 bytes per lane: keep flags (16-bit mask per lane, recomputed by the scatter), kept bytes
// per tile, and the start of the chunk's last header line (atomicMax).
__global__ void __launch_bounds__(256) k_ml_flags(const uint8_t *__restrict__ raw, int64_t n, int64_t n_tiles,
                                                  const int32_t *__restrict__ tile_prev, uint32_t *__restrict__ tile_cnt,
                                                  int *__restrict__ last_header)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles)
        return;
    const int64_t p = tile * 1024 + lane * 16;
    // position of the last newline before this lane's 16 bytes: from the lanes before it, else from the tiles before
    int32_t mine = -1;
    for (int i = 0; i < 16; ++i)
        if (p + i < n && raw[p + i] == 10u)
            mine = (int32_t)(p + i);
    int32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(inc, d);
        if (lane >= d && o > inc)
            inc = o;
    }
    int32_t prev = __shfl_up(inc, 1);
    if (lane == 0)
        prev = -1;
    const int32_t tp = tile_prev[tile];
    prev = prev > tp ? prev : tp;
    uint32_t kept = 0;
    int hdr = -1;
    for (int i = 0; i < 16; ++i) {
        const int64_t q = p + i;
        if (q >= n)
            break;
        const uint32_t c = raw[q];
        if (ml_keep(raw, n, q, c, prev))
            ++kept;
        if (c == (uint32_t)'>' && (int64_t)prev + 1 == q)
            hdr = (int)q; // a header line starts here
        if (c == 10u)
            prev = (int32_t)q;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        kept += __shfl_xor(kept, d);
        const int o = __shfl_xor(hdr, d);
        hdr = o > hdr ? o : hdr;
    }
    if (lane == 0) {
        tile_cnt[tile] = kept;
        if (hdr >= 0)
            atomicMax(last_header, hdr);
    }
}

// Kept bytes of [0, limit) to their compacted places: out[tile_pre + super_pre + rank inside the tile].
__global__ void __launch_bounds__(256) k_ml_scatter(const uint8_t *__restrict__ raw, int64_t n, int64_t limit, int64_t n_tiles,
                                                    const int32_t *__restrict__ tile_prev, const uint32_t *__restrict__ tile_pre,
                                                    const uint32_t *__restrict__ super_pre, uint8_t *__restrict__ out,
                                                    unsigned long long *out_len)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles || tile * 1024 > limit) // (the tile that STARTS at `limit` still reports the length)
        return;
    const int64_t p = tile * 1024 + lane * 16;
    int32_t mine = -1;
    for (int i = 0; i < 16; ++i)
        if (p + i < n && raw[p + i] == 10u)
            mine = (int32_t)(p + i);
    int32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(inc, d);
        if (lane >= d && o > inc)
            inc = o;
    }
    int32_t prev = __shfl_up(inc, 1);
    if (lane == 0)
        prev = -1;
    const int32_t tp = tile_prev[tile];
    prev = prev > tp ? prev : tp;
    uint32_t mask = 0;
    int32_t pv = prev;
    for (int i = 0; i < 16; ++i) {
        const int64_t q = p + i;
        if (q >= n)
            break;
        const uint32_t c = raw[q];
        if (ml_keep(raw, n, q, c, pv))
            mask |= 1u << i;
        if (c == 10u)
            pv = (int32_t)q;
    }
    uint32_t cnt = (uint32_t)__popc(mask), incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d)
            incl += o;
    }
    size_t dst = (size_t)super_pre[tile >> 10] + tile_pre[tile] + (incl - cnt);
    for (int i = 0; i < 16; ++i) {
        const int64_t q = p + i;
        if (q >= limit) {
            if (q == limit)
                *out_len = (unsigned long long)dst; // kept bytes before `limit` = length of the unwrapped chunk
            break;
        }
        if (q >= n)
            break;
        if ((mask >> i) & 1u)
            out[dst++] = raw[q];
    }
}

// ------------------------------------------------------------------------------------------------
// Records -> flat reads, for the radix path (r04).  Pass 1 of the radix path took raw FASTQ through its records
// front end (line numbering per byte, four <4>-window tiles per block) at 0.57 TB/s — ten times slower than on flat
// reads (profiles/r03/final_entry_points.txt), and that is the path `kmer_mapper map` runs
// (command_line_interface.py:102-111).  Now the raw chunk is COMPACTED on the device first, at line granularity: only
// the bytes of sequence lines survive, as a stream of 2-BIT CODES (16 per 32-bit word, first base in the lowest bits:
// the very words pass 1's tiles keep in LDS, so its byte -> code stage disappears; the lookup table — and with it the
// reference encoder's error for a non-nucleotide, util.py:72 — is applied here, where raw byte offsets are still
// known), every read start is marked in the bitset the ragged-read front end takes, and pass 1 then runs on flat reads
// (on packed tiles when all reads of the chunk have one length, which k_rec_uniform checks).
//   k_rec_count2   per 4 KiB tile: newlines + non-terminator bytes per (line inside the tile) mod 4
//   (k_rec_scan1, k_rec_scan2: newline prefix + where the last complete record ends, as before)
//   k_rec_seq_scan per tile: its sequence bytes (the count that belongs to the phase of its first line) -> prefix
//   (k_super_scan)
//   k_rec_scatter  sequence bytes before `limit` -> 2-bit codes, compacted; read starts -> bitset; record structure checked
//   k_rec_uniform  do all reads have one length?
// Both per-byte kernels are persistent: one wavefront per 4 KiB tile, 64 contiguous bytes per lane (four 16-byte loads),
// the next tile's bytes requested before the current tile is processed.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rec_load16(const uint8_t *__restrict__ raw, uint32_t n, uint32_t p, uint32_t (&w)[4])
{
    // (positions inside a piece of at most 2^30 bytes: 32-bit)
    if ((((uintptr_t)raw) & 15u) == 0 && p + 16u <= n) {
        const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(raw + p));
        w[0] = x[0]; w[1] = x[1]; w[2] = x[2]; w[3] = x[3];
    } else if (p + 16u <= n) {
        // a piece that starts where the previous one's last record ended: any byte offset.  One 16-byte load at an
        // unaligned address (gfx950 runs with unaligned access enabled; the compiler emits global_load_dwordx4 for it) —
        // with byte-wise loads the middle piece of a 3 GiB call took 2.6 ms instead of 1.3
        u32x4 x;
        __builtin_memcpy(&x, raw + p, 16);
        w[0] = x[0]; w[1] = x[1]; w[2] = x[2]; w[3] = x[3];
    } else if (p < n) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = tile_load_bytes4(raw, (int64_t)n, (int64_t)p + 4 * i);
    } else {
        w[0] = w[1] = w[2] = w[3] = 0u;
    }
}

// 16-bit masks of the lane's bytes: '\n', '\r'
__device__ __forceinline__ void rec_masks(const uint32_t (&w)[4], uint32_t &nl, uint32_t &cr)
{
    nl = 0;
    cr = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        nl |= flags_to_bits(bytes_equal(w[i], 10u)) << (4 * i);
        cr |= flags_to_bits(bytes_equal(w[i], 13u)) << (4 * i);
    }
}

// sum over the wavefront's lanes, in every lane (DPP additions: no LDS crossbar)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(v), 63);
}

// bit i of the result = XOR of the bits 0 .. i of x
__device__ __forceinline__ uint64_t prefix_xor64(uint64_t x)
{
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    x ^= x << 32;
    return x;
}

// For the lane's 64 bytes with newline mask nl: (number of newlines before byte i + base) mod 4, as two 64-bit masks
// (s0 = bit 0, s1 = bit 1 of that sum for every byte) — loop-free: a byte's count is odd iff the prefix XOR of the
// newline bits before it is 1, and bit 1 of the count flips at every SECOND newline (the newlines that have an odd
// number of newlines before them).  A newline itself still belongs to the line it ends.
__device__ __forceinline__ void rec_line_phase(uint64_t nl, uint32_t base, uint64_t &s0, uint64_t &s1)
{
    const uint64_t r0 = prefix_xor64(nl << 1);            // parity of the newlines before byte i
    const uint64_t r1 = prefix_xor64((nl & r0) << 1);     // ... of the 2nd, 4th, ... newlines before byte i
    const uint64_t b0 = (base & 1u) ? ~0ull : 0ull, b1 = (base & 2u) ? ~0ull : 0ull;
    s0 = r0 ^ b0;
    s1 = r1 ^ b1 ^ (r0 & b0);
}

constexpr uint32_t REC_TB = 4096; // bytes per tile of the compaction kernels: one wavefront, 64 contiguous bytes per lane

// the lane's 64 bytes (four 16-byte loads in flight) and their '\n' / '\r' masks
struct RecLane {
    uint32_t w[4][4];
};
__device__ __forceinline__ void rec_load64(const uint8_t *__restrict__ raw, uint32_t n, uint32_t p, RecLane &r)
{
#pragma unroll
    for (int j = 0; j < 4; ++j)
        rec_load16(raw, n, p + 16u * (uint32_t)j, r.w[j]);
}
__device__ __forceinline__ void rec_masks64(const RecLane &r, uint64_t &nl, uint64_t &cr)
{
    nl = 0;
    cr = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t a, b;
        rec_masks(r.w[j], a, b);
        nl |= (uint64_t)a << (16 * j);
        cr |= (uint64_t)b << (16 * j);
    }
}
__device__ __forceinline__ uint64_t rec_inside64(int32_t left) // the first `left` of 64 bytes
{
    return left >= 64 ? ~0ull : (left > 0 ? (1ull << left) - 1ull : 0ull);
}

// tile_cnt[t] = newlines of tile t; tile_seq[t] = four 16-bit counts: bytes that are no line terminator, by the number
// of newlines before them inside the tile, mod 4 (bytes past the end of the chunk count nowhere).
__global__ void __launch_bounds__(256) k_rec_count2(const uint8_t *__restrict__ raw, int64_t n, int64_t n_tiles,
                                                    uint32_t *__restrict__ tile_cnt, unsigned long long *__restrict__ tile_seq)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t stride = gridDim.x * 4u, nt = (uint32_t)n_tiles, n32 = (uint32_t)n; // (a piece holds at most 2^30 bytes)
    uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6); // one wavefront per tile
    if (tile >= nt)
        return;
    RecLane cur, nxt;
    rec_load64(raw, n32, tile * REC_TB + lane * 64u, cur);
    for (; tile < nt; tile += stride) {
        const uint32_t p = tile * REC_TB + lane * 64u;
        if (tile + stride < nt)
            rec_load64(raw, n32, (tile + stride) * REC_TB + lane * 64u, nxt); // the next tile is in flight while this one is counted
        uint64_t nl, cr;
        rec_masks64(cur, nl, cr);
        const uint64_t inside = rec_inside64((int32_t)n32 - (int32_t)p);
        nl &= inside;
        const uint32_t c = (uint32_t)__popcll(nl);
        const uint32_t inc = wave_scan_incl(c);
        // the lane's bytes by (newlines before them inside the tile) mod 4; two 16-bit counters per 32-bit word
        uint64_t s0, s1;
        rec_line_phase(nl, inc - c, s0, s1);
        const uint64_t body = inside & ~(nl | cr);
        const uint32_t a01 = (uint32_t)__popcll(body & ~s1 & ~s0) | ((uint32_t)__popcll(body & ~s1 & s0) << 16);
        const uint32_t a23 = (uint32_t)__popcll(body & s1 & ~s0) | ((uint32_t)__popcll(body & s1 & s0) << 16);
        const uint32_t s01 = wave_sum(a01), s23 = wave_sum(a23); // (no field exceeds 4096)
        if (lane == 63) {
            tile_cnt[tile] = inc;
            tile_seq[tile] = ((unsigned long long)s23 << 32) | s01;
        }
        cur = nxt;
    }
}

// One workgroup per super-tile, after the newline prefix is complete: the sequence bytes of every tile — the bytes of
// the lines with (line number mod period) == 1 — as an exclusive prefix inside the super-tile + the super-tile's total.
__global__ void __launch_bounds__(1024) k_rec_seq_scan(const unsigned long long *__restrict__ tile_seq,
                                                       const uint32_t *__restrict__ tile_nl, const uint32_t *__restrict__ super_nl,
                                                       int64_t n_tiles, uint32_t period_mask, uint32_t *__restrict__ tile_pre,
                                                       uint32_t *__restrict__ super_tot)
{
    __shared__ uint32_t s_a[1024];
    const int t = threadIdx.x;
    const int64_t tile = (int64_t)blockIdx.x * 1024 + t;
    uint32_t c = 0;
    if (tile < n_tiles) {
        const uint32_t line0 = super_nl[blockIdx.x] + tile_nl[tile];
        // a byte with r newlines before it inside the tile lies on line line0 + r: sequence iff (line0 + r) & mask == 1
        unsigned long long f = tile_seq[tile];
        for (uint32_t r = 0; r < 4u; ++r, f >>= 16)
            if (((line0 + r) & period_mask) == 1u)
                c += (uint32_t)(f & 0xFFFFu);
    }
    s_a[t] = c;
    __syncthreads();
    block_scan_1024(s_a);
    if (tile < n_tiles)
        tile_pre[tile] = s_a[t] - c;
    if (t == 1023)
        super_tot[blockIdx.x] = s_a[t];
}

// One wavefront per 4 KiB tile (64 contiguous bytes per lane), over a contiguous range of tiles.  info = {consumed,
// n_records, n_lines} of k_rec_scan2 (read on the device: no host round trip between the census and the scatter);
// out_info[0] = flat bases behind the piece (flat_base + the sequence bytes before `consumed`).  codes: the 2-bit stream
// (zeroed by the caller: the words two tiles share are OR-ed in).  first_bad as in the tile front end: [0] a sequence byte
// without a code, [1] a record line that does not start with the header character / '+' (raw byte offsets).
// (r04 history: 16 bytes per lane and 1 KiB tiles cost ~700 instructions per tile, most of them per-tile overhead —
// scans, table lookups, loop control — and ran at 1.35 ms per GiB; 64 bytes per lane share that overhead four ways.)
__global__ void __launch_bounds__(256) k_rec_scatter(const uint8_t *__restrict__ raw, int64_t n, int64_t n_tiles,
                                                     const uint32_t *__restrict__ tile_nl, const uint32_t *__restrict__ super_nl,
                                                     const uint32_t *__restrict__ tile_pre, const uint32_t *__restrict__ super_pre,
                                                     const int64_t *__restrict__ info, const uint8_t *__restrict__ lut,
                                                     uint32_t period_mask, uint32_t header_char, uint32_t *__restrict__ codes,
                                                     uint64_t flat_base, uint32_t *__restrict__ start_bits,
                                                     unsigned long long *__restrict__ first_bad,
                                                     unsigned long long *__restrict__ out_info)
{
    __shared__ uint32_t s_lut[256];
    __shared__ uint32_t s_row[4][264]; // a tile's codes: at most 4096 + 15 -> 257 words
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (positions inside a piece fit 32 bits — at most 2^30 bytes: no 64-bit arithmetic per tile or lane)
    const uint32_t limit = (uint32_t)info[0], n32 = (uint32_t)n;
    const uint32_t n_live = (limit + REC_TB - 1u) / REC_TB < (uint32_t)n_tiles ? (limit + REC_TB - 1u) / REC_TB : (uint32_t)n_tiles;
    // A wavefront takes a CONTIGUOUS range of tiles: what a tile needs besides its bytes — the line number and the
    // flat position of its first byte, and whether the byte before it ends a line — comes from one coalesced load
    // per 64 tiles (lane i holds tile t0 + i's) and from the previous tile.
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t per_wave = (n_live + n_waves - 1u) / n_waves;
    const uint32_t t_begin = (blockIdx.x * 4u + (uint32_t)wv) * per_wave;
    const uint32_t t_end = t_begin + per_wave < n_live ? t_begin + per_wave : n_live;
    if (t_begin >= t_end)
        return;
    uint32_t *row = s_row[wv];
    RecLane cur, nxt;
    rec_load64(raw, n32, t_begin * REC_TB + (uint32_t)lane * 64u, cur);
    uint32_t carry_nl = t_begin == 0 ? 1u : (raw[t_begin * REC_TB - 1u] == 10u ? 1u : 0u); // does the byte before the tile end a line?
    uint32_t carry_cr = t_begin == 0 ? 0u : (raw[t_begin * REC_TB - 1u] == 13u ? 1u : 0u); // ... is it a '\r'?
    uint32_t m_line = 0, m_dst = 0;
    for (uint32_t tile = t_begin; tile < t_end; ++tile) {
        const uint32_t p = tile * REC_TB + (uint32_t)lane * 64u;
        const int slot = (int)((tile - t_begin) & 63u);
        if (slot == 0) { // the next 64 tiles' table entries
            const uint32_t t = tile + (uint32_t)lane;
            if (t < t_end) {
                m_line = super_nl[t >> 10] + tile_nl[t];
                m_dst = super_pre[t >> 10] + tile_pre[t];
            }
        }
        if (tile + 1u < t_end)
            rec_load64(raw, n32, (tile + 1u) * REC_TB + (uint32_t)lane * 64u, nxt);
        const uint32_t line0 = (uint32_t)__builtin_amdgcn_readlane((int)m_line, slot);
        const uint64_t dst0 = flat_base + (uint32_t)__builtin_amdgcn_readlane((int)m_dst, slot); // flat position of the tile's first sequence byte
#pragma unroll
        for (int j = 0; j < 4; ++j)
            row[lane + 64 * j] = 0u;
        if (lane < 8)
            row[256 + lane] = 0u;
        uint64_t nl, cr;
        rec_masks64(cur, nl, cr);
        const uint64_t inside = rec_inside64((int32_t)limit - (int32_t)p); // the lane's bytes before the limit
        const uint32_t last_nl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(nl >> 63), 63) & 1u; // (before the limit is applied)
        const uint32_t last_cr = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cr >> 63), 63) & 1u;
        nl &= inside;
        const uint32_t c_nl = (uint32_t)__popcll(nl);
        const uint32_t line_lane = line0 + (wave_scan_incl(c_nl) - c_nl); // line of the lane's first byte
        // first byte of a line: preceded by '\n' (or the first byte of the chunk)
        uint32_t prev_nl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(nl >> 63), 0x138, 0xF, 0xF, false) & 1u; // wave_shr:1
        if (lane == 0)
            prev_nl = carry_nl;
        carry_nl = last_nl;
        const uint32_t carry_cr_now = carry_cr;
        carry_cr = last_cr;
        const uint64_t first = ((nl << 1) | prev_nl) & inside;
        // the line phase of every byte (line number mod period), loop-free; sequence lines have phase 1
        uint64_t s0, s1;
        rec_line_phase(nl, line_lane, s0, s1);
        const uint64_t hi_ok = period_mask == 3u ? ~s1 : ~0ull; // (two-line FASTA: only bit 0 of the line number counts)
        const uint64_t seq_line = s0 & hi_ok & inside;           // (the line's terminator included)
        {
            // record structure: a header line starts with the header character, the third line of a FASTQ record with '+'
            uint64_t chk = first & ~s0;                           // line starts with phase 0 or 2
            uint32_t bad_struct = 0xFFFFFFFFu;
            while (chk) {
                const uint32_t i = (uint32_t)__builtin_ctzll(chk);
                chk &= chk - 1ull;
                const uint32_t ch = raw[p + i];                   // (rare: a cache hit on the lane's own bytes)
                const bool third = period_mask == 3u && ((s1 >> i) & 1ull);
                if (third ? ch != (uint32_t)'+' : ch != header_char)
                    bad_struct = bad_struct < i ? bad_struct : i;
            }
            if (bad_struct != 0xFFFFFFFFu)
                atomicMin(&first_bad[1], (unsigned long long)(p + bad_struct));
        }
        const uint64_t seq = seq_line & ~(nl | cr);
        const uint32_t cnt = (uint32_t)__popcll(seq);
        const uint32_t incl = wave_scan_incl(cnt);
        const uint32_t pre = incl - cnt;
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        // read starts: the first byte of every sequence line; and whatever follows a '\r' inside a sequence line (no
        // window may span it: the records front end treats every '\r' as a break)
        {
            uint64_t marks = (first | ((cr & seq_line) << 1)) & seq_line;
            while (marks) {
                const uint32_t i = (uint32_t)__builtin_ctzll(marks);
                marks &= marks - 1ull;
                const uint64_t f = dst0 + pre + (uint32_t)__popcll(seq & ((1ull << i) - 1ull));
                atomicOr(&start_bits[f >> 5], 1u << (f & 31u));
            }
            // (a '\r' in the previous lane's last byte — for lane 0: in the last byte of the tile before, round 5: until then a
            // '\r' that fell on the last byte of a 4 KiB tile did not break the read — : this lane's first byte, if it is on
            // the sequence line (a '\r' ends no line: it lies on the same one), gets the mark)
            uint32_t cr_before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(cr >> 63), 0x138, 0xF, 0xF, false) & 1u;
            if (lane == 0)
                cr_before = carry_cr_now;
            if (cr_before && (seq_line & 1ull)) {
                const uint64_t f = dst0 + pre;
                atomicOr(&start_bits[f >> 5], 1u << (f & 31u));
            }
        }
        // 16 bytes at a time: all of them through the table (the codes of bytes that are no sequence bytes are dropped),
        // the runs of sequence bytes moved down to bit 0, the word OR-ed into the wavefront's row at code position
        // (dst0 & 15) + rank: word j of the row is word (dst0 >> 4) + j of the output
        const uint32_t mis = (uint32_t)dst0 & 15u;
        uint32_t q = mis + pre, bad = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t seq16 = (uint32_t)(seq >> (16 * j)) & 0xFFFFu;
            if (!seq16)
                continue;
            uint32_t call = 0, bad_m = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i) {
                const uint32_t l = s_lut[(cur.w[j][i >> 2] >> (8u * (i & 3u))) & 0xFFu];
                bad_m |= (l == 0xFFu ? 1u : 0u) << i;
                call |= (l & 3u) << (2u * i);
            }
            bad_m &= seq16;
            if (bad_m && bad == 0xFFFFFFFFu)
                bad = 16u * (uint32_t)j + (uint32_t)__builtin_ctz(bad_m);
            uint32_t cw = 0, k2 = 0, m = seq16;
            while (m) {
                const uint32_t a = (uint32_t)__builtin_ctz(m);               // the run [a, a + len)
                const uint32_t inv = ~(m >> a);
                const uint32_t len = (uint32_t)__builtin_ctz(inv);            // (m has 16 bits: inv is never 0)
                const uint32_t run = len >= 16u ? 0xFFFFFFFFu : (1u << (2u * len)) - 1u;
                cw |= ((call >> (2u * a)) & run) << k2;
                k2 += 2u * len;
                m &= ~(((1u << len) - 1u) << a);
            }
            const uint32_t sh = (q & 15u) * 2u;
            atomicOr(&row[q >> 4], cw << sh);
            if (sh && (k2 + sh > 32u))
                atomicOr(&row[(q >> 4) + 1u], cw >> (32u - sh));
            q += k2 >> 1;
        }
        if (bad != 0xFFFFFFFFu)
            atomicMin(&first_bad[0], (unsigned long long)(p + bad));
        __builtin_amdgcn_wave_barrier();
        // words only this tile writes by plain stores; the first and last word may be shared with the neighbouring tiles
        const uint32_t n_w = (mis + tot + 15u) >> 4;       // words of the row that hold codes
        uint32_t *out = codes + (dst0 >> 4);
        for (uint32_t j = (uint32_t)lane; j < n_w; j += 64u) {
            const uint32_t v = row[j];
            if (j == 0u || j + 1u == n_w) {
                if (v)
                    atomicOr(&out[j], v);
            } else {
                out[j] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the flat length = the flat position behind the lane that holds the last byte before the limit
        const int32_t left = (int32_t)limit - (int32_t)p;
        if (left >= 1 && left <= 64)
            out_info[0] = dst0 + pre + cnt;
        cur = nxt;
    }
}

// Do all reads of the compacted piece have one length?  The piece's flat bases are [flat_base, out_info[0]), info[1] =
// its records; counts the read starts inside that range into out_info[1] and those whose distance from flat_base is no
// multiple of L = bases / records into out_info[2]: with as many distinct starts as records, all of them multiples of L,
// every multiple of L is a start.
__global__ void __launch_bounds__(256) k_rec_uniform(const uint32_t *__restrict__ start_bits, uint64_t flat_base,
                                                     const int64_t *__restrict__ info, unsigned long long *__restrict__ out_info)
{
    const uint64_t end = out_info[0], recs = (uint64_t)info[1];
    if (info[0] <= 0 || !recs || end <= flat_base || (end - flat_base) % recs)
        return; // (out_info[1] stays 0: not uniform)
    const uint32_t L = (uint32_t)((end - flat_base) / recs);
    const uint64_t w0 = flat_base >> 5, w1 = (end + 31) >> 5;
    uint32_t n_set = 0, n_off = 0;
    for (uint64_t wd = w0 + (uint64_t)blockIdx.x * 256 + threadIdx.x; wd < w1; wd += (uint64_t)gridDim.x * 256) {
        uint32_t m = start_bits[wd];
        if (wd * 32 < flat_base)
            m &= ~((1u << (flat_base - wd * 32)) - 1u);
        if (wd * 32 + 32 > end)
            m &= (1u << (end - wd * 32)) - 1u;
        n_set += (uint32_t)__popc(m);
        while (m) {
            const uint64_t pos = wd * 32 + (uint32_t)__builtin_ctz(m) - flat_base;
            m &= m - 1u;
            n_off += pos % L ? 1u : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        n_set += __shfl_xor(n_set, d);
        n_off += __shfl_xor(n_off, d);
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_set)
            atomicAdd(&out_info[1], (unsigned long long)n_set);
        if (n_off)
            atomicAdd(&out_info[2], (unsigned long long)n_off);
    }
}
