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
// the bytes of sequence lines survive, as 2-bit codes one per byte (the lookup table — and with it the reference
// encoder's error for a non-nucleotide, util.py:72 — is applied here, where raw byte offsets are still known), every
// read start is marked in the bitset the ragged-read front end takes, and pass 1 then runs on flat reads (on packed
// tiles when all reads of the chunk have one length, which k_rec_uniform checks).
//   k_rec_count2   per 1024-byte tile: newlines + non-terminator bytes per (line inside the tile) mod 4
//   (k_rec_scan1, k_rec_scan2: newline prefix + where the last complete record ends, as before)
//   k_rec_seq_scan per tile: its sequence bytes (the count that belongs to the phase of its first line) -> prefix
//   (k_super_scan)
//   k_rec_scatter  sequence bytes before `limit` -> codes, compacted; read starts -> bitset; record structure checked
//   k_rec_uniform  do all reads have one length?
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rec_load16(const uint8_t *__restrict__ raw, int64_t n, int64_t p, uint32_t (&w)[4])
{
    if ((((uintptr_t)raw) & 15u) == 0 && p + 16 <= n) {
        const u32x4 x = *reinterpret_cast<const u32x4 *>(raw + p);
        w[0] = x[0]; w[1] = x[1]; w[2] = x[2]; w[3] = x[3];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = tile_load_bytes4(raw, n, p + 4 * i);
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

// tile_cnt[t] = newlines of tile t; tile_seq[t] = four 16-bit counts: bytes that are no line terminator, by the number
// of newlines before them inside the tile, mod 4 (bytes past the end of the chunk count nowhere).
__global__ void __launch_bounds__(256) k_rec_count2(const uint8_t *__restrict__ raw, int64_t n, int64_t n_tiles,
                                                    uint32_t *__restrict__ tile_cnt, unsigned long long *__restrict__ tile_seq)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); // one wavefront per tile
    if (tile >= n_tiles)
        return;
    const int64_t p = tile * 1024 + lane * 16;
    uint32_t w[4], nl, cr;
    rec_load16(raw, n, p, w);
    rec_masks(w, nl, cr);
    const int64_t left = n - p;
    const uint32_t inside = left >= 16 ? 0xFFFFu : (left > 0 ? (1u << left) - 1u : 0u);
    nl &= inside;
    const uint32_t c = (uint32_t)__popc(nl);
    const uint32_t before = wave_scan_incl(c) - c; // newlines of the tile before this lane's bytes
    // the lane's bytes fall into popc(nl) + 1 segments of one line each
    unsigned long long acc = 0;
    uint32_t body = inside & ~(nl | cr), rest = nl, from = 0, rel = before;
    for (;;) {
        const uint32_t to = rest ? (uint32_t)__builtin_ctz(rest) : 16u; // the segment's bytes: [from, to)
        const uint32_t seg = (to >= 16u ? 0xFFFFu : (1u << to) - 1u) & ~((1u << from) - 1u);
        acc += (unsigned long long)__popc(body & seg) << (16u * (rel & 3u));
        if (!rest)
            break;
        rest &= rest - 1u;
        from = to + 1u;
        ++rel;
    }
    uint32_t cs = c;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        cs += __shfl_xor(cs, d);
        acc += __shfl_xor(acc, d); // (no field exceeds 1024)
    }
    if (lane == 0) {
        tile_cnt[tile] = cs;
        tile_seq[tile] = acc;
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

// One wavefront per tile.  info = {consumed, n_records, n_lines} of k_rec_scan2 (read on the device: no host round trip
// between the census and the scatter); out_info[0] = number of flat bases (the sequence bytes before `consumed`).
// first_bad as in the tile front end: [0] a sequence byte without a code, [1] a record line that does not start with
// the header character / '+' (raw byte offsets).
__global__ void __launch_bounds__(256) k_rec_scatter(const uint8_t *__restrict__ raw, int64_t n, int64_t n_tiles,
                                                     const uint32_t *__restrict__ tile_nl, const uint32_t *__restrict__ super_nl,
                                                     const uint32_t *__restrict__ tile_pre, const uint32_t *__restrict__ super_pre,
                                                     const int64_t *__restrict__ info, const uint8_t *__restrict__ lut,
                                                     uint32_t period_mask, uint32_t header_char, uint8_t *__restrict__ flat,
                                                     uint64_t flat_base, uint32_t *__restrict__ start_bits,
                                                     unsigned long long *__restrict__ first_bad,
                                                     unsigned long long *__restrict__ out_info)
{
    __shared__ uint32_t s_lut[256];
    __shared__ uint32_t s_out[4][1024 / 4 + 2];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    const int64_t limit = info[0];
    if (tile >= n_tiles || tile * 1024 >= limit) // (limit > 0: the caller does not launch for a chunk without a record)
        return;
    const int64_t p = tile * 1024 + lane * 16;
    uint32_t w[4], nl, cr;
    rec_load16(raw, n, p, w);
    rec_masks(w, nl, cr);
    const uint64_t w01 = ((uint64_t)w[1] << 32) | w[0], w23 = ((uint64_t)w[3] << 32) | w[2];
    auto byte_at = [&](uint32_t i) { return (uint32_t)((i < 8u ? w01 >> (8u * i) : w23 >> (8u * (i - 8u))) & 0xFFu); };
    const int64_t left = limit - p; // bytes of this lane before the limit
    const uint32_t inside = left >= 16 ? 0xFFFFu : (left > 0 ? (1u << left) - 1u : 0u);
    nl &= inside;
    const uint32_t c_nl = (uint32_t)__popc(nl);
    const uint32_t line_lane = super_nl[tile >> 10] + tile_nl[tile] + (wave_scan_incl(c_nl) - c_nl); // line of the lane's first byte
    // first byte of a line: preceded by '\n' (or the first byte of the chunk)
    uint32_t prev_nl = (uint32_t)__shfl_up((int)(nl >> 15), 1) & 1u;
    if (lane == 0)
        prev_nl = p == 0 ? 1u : (raw[p - 1] == 10u ? 1u : 0u);
    const uint32_t first = ((nl << 1) | prev_nl) & inside;
    // sequence bytes, line by line (segments as in k_rec_count2)
    uint32_t seq = 0, seq_line = 0, bad_struct = 0xFFFFFFFFu;
    {
        uint32_t rest = nl, from = 0, line = line_lane;
        for (;;) {
            const uint32_t to = rest ? (uint32_t)__builtin_ctz(rest) : 16u;
            const uint32_t seg = (to >= 16u ? 0xFFFFu : (1u << to) - 1u) & ~((1u << from) - 1u);
            const uint32_t phase = line & period_mask;
            if (phase == 1u)
                seq_line |= seg | (to < 16u ? 1u << to : 0u); // (the line's terminator included)
            if ((first >> from) & 1u) { // the segment starts its line: the record structure is checked on that byte
                const uint32_t ch = byte_at(from);
                if ((phase == 0u && ch != header_char) || (phase == 2u && ch != (uint32_t)'+'))
                    bad_struct = bad_struct < from ? bad_struct : from;
            }
            if (!rest)
                break;
            rest &= rest - 1u;
            from = to + 1u;
            ++line;
        }
        seq_line &= inside;
        seq = seq_line & ~(nl | cr);
    }
    if (bad_struct != 0xFFFFFFFFu)
        atomicMin(&first_bad[1], (unsigned long long)(p + bad_struct));
    const uint32_t cnt = (uint32_t)__popc(seq);
    const uint32_t pre = wave_scan_incl(cnt) - cnt;
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)(pre + cnt), 63);
    const uint64_t dst0 = flat_base + super_pre[tile >> 10] + tile_pre[tile]; // flat position of the tile's first sequence byte
    // read starts: the first byte of every sequence line; and whatever follows a '\r' inside a sequence line (no window
    // may span it: the records front end treats every '\r' as a break)
    {
        uint32_t marks = (first | ((cr & seq_line) << 1)) & seq_line & 0xFFFFu;
        while (marks) {
            const uint32_t i = (uint32_t)__builtin_ctz(marks);
            marks &= marks - 1u;
            const uint64_t f = dst0 + pre + (uint32_t)__popc(seq & ((1u << i) - 1u));
            atomicOr(&start_bits[f >> 5], 1u << (f & 31u));
        }
        // (a '\r' in the lane's last byte: the next lane's first byte — if it is a sequence byte it gets the mark there)
        const uint32_t cr_before = (uint32_t)__shfl_up((int)((cr & seq_line) >> 15), 1) & 1u;
        if (lane != 0 && cr_before && (seq_line & 1u)) {
            const uint64_t f = dst0 + pre;
            atomicOr(&start_bits[f >> 5], 1u << (f & 31u));
        }
    }
    // codes of the sequence bytes -> the wavefront's LDS row, at byte offset (dst0 & 3) + rank: LDS word j then is word
    // (dst0 >> 2) + j of the output
    uint8_t *row = reinterpret_cast<uint8_t *>(s_out[wv]);
    const uint32_t mis = (uint32_t)dst0 & 3u;
    uint32_t bad = 0xFFFFFFFFu;
    {
        uint32_t m = seq, o = mis + pre;
        while (m) {
            const uint32_t i = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            const uint32_t l = s_lut[byte_at(i)];
            if (l == 0xFFu)
                bad = bad < i ? bad : i;
            row[o++] = (uint8_t)(l & 3u);
        }
    }
    if (bad != 0xFFFFFFFFu)
        atomicMin(&first_bad[0], (unsigned long long)(p + bad));
    __builtin_amdgcn_wave_barrier();
    __threadfence_block(); // the row is written and read by this wavefront only
    // whole words by 4-byte stores; the bytes of the first and last word that belong to this tile by byte stores (the
    // neighbouring tiles write the rest of those words)
    const uint32_t end = mis + tot;               // bytes [mis, end) of the row are this tile's
    const uint32_t w_lo = mis ? 1u : 0u;          // whole words: [w_lo, w_hi)
    const uint32_t w_hi = (end >> 2) > w_lo ? end >> 2 : w_lo;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(flat) + (dst0 >> 2);
    for (uint32_t j = w_lo + (uint32_t)lane; j < w_hi; j += 64u)
        out32[j] = s_out[wv][j];
    if ((uint32_t)lane < 8u) {
        // head bytes [mis, min(end, 4 w_lo)) by lanes 0..3, tail bytes [4 w_hi, end) by lanes 4..7
        const uint32_t b = (uint32_t)lane < 4u ? (uint32_t)lane : (w_hi << 2) + ((uint32_t)lane - 4u);
        const bool mine = (uint32_t)lane < 4u ? (b >= mis && b < end && b < (w_lo << 2)) : (b < end);
        if (mine)
            flat[(dst0 - mis) + b] = row[b];
    }
    // the flat length = the flat position behind the lane that holds the last byte before the limit
    if (left >= 1 && left <= 16)
        out_info[0] = dst0 + pre + cnt;
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
