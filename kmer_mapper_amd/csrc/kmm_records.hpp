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
