// kmm_tile.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip inside its anonymous namespace.
// Tile front end: read bytes (flat, uniform-length or raw FASTA/FASTQ records) -> packed k-mers per lane.
#pragma once

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
    uint32_t lut[256]; // one dword per entry: lanes looking up different letters hit different banks
    uint32_t codes[NV + 1];
    uint32_t bits[NB + 1];
};

struct TileConst {
    uint64_t kmask; // low 2k bits
    uint64_t bmask; // read starts in (p, p+k-1] kill the window at p
    bool aligned;   // bases pointer is 16-byte aligned
};

// The lane's register window behind its first position: bases 0 .. 63 in `lo`, 64 .. in `hi` (first base in the lowest
// bits).  Callers that slide over the window themselves (radix pass 1's incremental division) take it instead of q[].
struct TileWin {
    uint64_t lo, hi;
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

// What one thread stages of a tile, loaded before tile_kmers consumes it so that a caller can keep the loads of
// several tiles in flight (the radix path's pass 1).  Flat reads (general / uniform): thread t < T/16 owns the 16
// bytes at t0 + 16t, and threads t < 12 also one dword of the 48-byte halo behind the tile, so the byte -> code
// stage is spread over all lanes.  Records mode: thread t < NV owns the 16 bytes at t0 + 16t (halo vectors included).
// Ragged reads: thread t <= NB also owns word t of the tile's slice of the read-start bitset.
struct TileRaw {
    uint32_t w[4];
    uint32_t halo;
    uint32_t sbits;
};

// Byte-wise loads of a chunk's last vector / of chunks whose base address is not 16-byte aligned: a real call, kept out
// of the callers' register budget (inlined into pass 1 its 16 addresses and bounds pushed the thread index into
// scratch, and every reload in the block loop waited for vmcnt(0): the prefetched tile and the copy-out stores).
__device__ __attribute__((noinline)) uint32_t tile_load_bytes4(const uint8_t *bases, int64_t total, int64_t p)
{
    uint32_t acc = 0;
    for (int j = 0; j < 4; ++j) {
        const int64_t pp = p + j;
        const uint32_t c = (pp < total) ? bases[pp] : 0u;
        acc |= c << (8 * j);
    }
    return acc;
}

__device__ __forceinline__ uint32_t tile_load_bytes4(const ReadsView &rv, int64_t p)
{
    return tile_load_bytes4(rv.bases, rv.total, p);
}

// C2: `bases` is a stream of 2-bit codes (ReadsView::codes2; a template parameter so that the byte path's kernels do
// not carry the second loader: with a run-time branch pass 1's packed-tile kernel spilled three registers)
template <int S, int MODE, bool C2 = false>
__device__ __forceinline__ void tile_load_vec(const ReadsView &rv, const TileConst &tc, int64_t tile, const int tid,
                                              TileRaw &raw)
{
    constexpr int T = TileSmem<S>::T;
    constexpr int NV = TileSmem<S>::NV;
    constexpr int NMAIN = MODE == MODE_RECORDS ? NV : T / 16;
    static_assert(NMAIN <= 256, "one staged 16-byte vector per thread");
    raw.w[0] = raw.w[1] = raw.w[2] = raw.w[3] = 0u;
    raw.halo = 0u;
    raw.sbits = 0u;
    const int64_t total = rv.total;
    if (MODE == MODE_GENERAL && tid <= TileSmem<S>::NB) { // T is a multiple of 1024: the slice starts at a word
        const int64_t wi = tile * (T / 32) + tid;
        if (wi < rv.n_start_words)
            raw.sbits = rv.start_bits[wi];
    }
    if constexpr (C2) { // 2-bit codes: one staged word (16 positions) per thread, three halo words
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(rv.bases);
        const int64_t n_words = (total + 15) >> 4;
        if (tid < T / 16) {
            const int64_t wi = tile * (T / 16) + tid;
            raw.w[0] = wi < n_words ? __builtin_nontemporal_load(cw + wi) : 0u;
        }
        if (tid < 3) {
            const int64_t wi = tile * (T / 16) + T / 16 + tid;
            raw.halo = wi < n_words ? __builtin_nontemporal_load(cw + wi) : 0u;
        }
        return;
    }
    if (tid < NMAIN) {
        const int64_t p = tile * T + (int64_t)tid * 16;
        if (tc.aligned && p + 16 <= total) {
            // streamed once: non-temporal so the read bytes do not displace index lines in L2
            u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rv.bases + p));
            raw.w[0] = x[0]; raw.w[1] = x[1]; raw.w[2] = x[2]; raw.w[3] = x[3];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                raw.w[i] = tile_load_bytes4(rv, p + i * 4);
        }
    }
    if (MODE != MODE_RECORDS && tid < 12) {
        const int64_t p = tile * T + T + (int64_t)tid * 4;
        if (tc.aligned && p + 4 <= total)
            raw.halo = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(rv.bases + p));
        else
            raw.halo = tile_load_bytes4(rv, p);
    }
}

// `tid` is the thread's index inside the 256-thread group that owns the tile: threadIdx.x for 256-thread
// workgroups; wider workgroups (the radix path) run one tile per 256-thread quarter, every quarter with its
// own TileSmem, and all of them pass through the same barriers.
// TOPBAR = false: the caller guarantees that a workgroup barrier already lies between the previous tile's last LDS
// read (and the code table's staging) and this call.
template <int S, int MODE, bool TOPBAR = true, bool C2 = false>
__device__ __forceinline__ uint32_t tile_kmers(const ReadsView &rv, const TileConst &tc, int64_t tile,
                                               int k, TileSmem<S> &sm, uint64_t (&q)[S], const int tid,
                                               const TileRaw &raw, TileWin *win = nullptr)
{
    const uint32_t (&w)[4] = raw.w;
    constexpr bool UNIFORM = MODE == MODE_UNIFORM;
    constexpr bool RECORDS = MODE == MODE_RECORDS;
    constexpr int T = TileSmem<S>::T;
    constexpr int NV = TileSmem<S>::NV;
    constexpr int NB = TileSmem<S>::NB;
    static_assert(!RECORDS || NV <= 256, "records mode: one staged 16-byte vector per thread");
    static_assert(T % 1024 == 0, "tile_first / tile_nl are kept per 1024 positions");
    const int64_t total = rv.total;
    const int64_t t0 = tile * T;
    if (TOPBAR)
        __syncthreads(); // LUT visible; every wave has finished reading the previous tile's LDS words
    if (MODE == MODE_GENERAL && tid <= NB) // read starts inside [t0, t0 + 32 (NB + 1)): the bitset's own words
        sm.bits[tid] = raw.sbits;

    if (RECORDS) {
        // ---- records mode, stage 1: raw file bytes.  A byte is a base iff it lies on the sequence
        // line of its record (line index mod period == 1) and is not a line terminator; every other
        // byte is a "break" that no window may contain, so k-mers never leave their read.
        uint32_t nl = 0, cr = 0; // 16-bit masks: byte i is '\n' / '\r'
        const int v = tid;
        const int64_t p = t0 + (int64_t)v * 16;
        if (v < NV) {
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
            // newlines before the tile (tiles past the end of the chunk — the radix path rounds the tile
            // count up to whole blocks — hold no bytes and must not index the census arrays)
            uint32_t line0 = t0 < total ? rv.super_nl[(tile * (T / 1024)) >> 10] + rv.tile_nl[tile * (T / 1024)] : 0u;
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
    if constexpr (C2) { // (the input already is what this stage produces)
        if (tid < T / 16)
            sm.codes[tid] = w[0];
        if (tid < 3)
            sm.codes[T / 16 + tid] = raw.halo;
    } else {
    if (tid < T / 16) {
        const int v = tid;
        const int64_t p = t0 + (int64_t)v * 16;
        // (bad: bit i = byte i has no code.  Tried instead: a flag above the code bits in the LDS table and one OR per
        // byte, the position worked out only when the flag shows up — pass 1 3.39 vs 3.35 ms, not kept.)
        uint32_t code = 0, bad = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            const uint32_t l = sm.lut[c];
            bad |= (l == 0xFFu ? 1u : 0u) << i;
            code |= (l & 3u) << (2 * i);
        }
        sm.codes[v] = code;
        if (bad) {
            const int64_t left = total - p; // bytes of this vector inside the chunk (those past its end are staged as 0)
            if (left < 16)
                bad &= left > 0 ? (1u << left) - 1u : 0u;
            if (bad)
                atomicMin(rv.first_bad, (unsigned long long)(p + __builtin_ctz(bad)));
        }
    }
    if (tid < 12) { // the 48 bytes behind the tile, four per thread -> one byte of codes each
        const int64_t p = t0 + T + (int64_t)tid * 4;
        uint32_t code = 0;
        int bad = -1;
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            uint32_t c = (raw.halo >> (8 * i)) & 0xFFu;
            uint32_t l = sm.lut[c];
            if (l == 0xFFu && p + i < total)
                bad = i;
            code |= (l & 3u) << (2 * i);
        }
        reinterpret_cast<uint8_t *>(&sm.codes[T / 16])[tid] = (uint8_t)code;
        if (bad >= 0)
            atomicMin(rv.first_bad, (unsigned long long)(p + bad));
    }
    }
    __syncthreads();
    } // !RECORDS

    // ---- stage 3: S consecutive windows per lane -------------------------------------------
    const int q0 = tid * S;
    const int64_t p0 = t0 + q0;
    uint64_t lo, hi;
    {
        const int wi = q0 >> 4;
#ifdef KMM_WINDOW_SHUFFLE
        // A/B variant (profiles/r02/window_shift_ab.md): the lane reads only the packed word its own positions lie
        // in and pulls the 30-base halo from the lanes that own the next two words (ds_bpermute / DPP); the lanes
        // whose neighbours sit in the next wavefront read LDS like the default build.
        constexpr int LPW = 16 / S >= 1 ? 16 / S : 1;      // lanes per packed word
        const uint32_t c0 = sm.codes[wi];
        uint32_t c1 = __shfl_down(c0, LPW), c2 = __shfl_down(c0, 2 * LPW);
        const int lane = tid & 63;
        if (lane + LPW > 63)
            c1 = sm.codes[wi + 1];
        if (lane + 2 * LPW > 63)
            c2 = sm.codes[wi + 2];
#else
        const uint32_t c0 = sm.codes[wi], c1 = sm.codes[wi + 1], c2 = sm.codes[wi + 2];
#endif
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
        // o = offset of the lane's first position inside its read.  The 64-bit division is done ONCE per tile, on a
        // value every lane shares (the tile's start; readfirstlane lets it run on the scalar unit); the lane adds
        // its own position in the tile and reduces with 32-bit arithmetic.  (One 64-bit division per lane and 64-bit
        // mask arithmetic cost pass 1 ~11 VALU instructions per k-mer: the ragged-read kernel was FASTER, 2.50 vs
        // 2.81 ms.)
        uint64_t o;
        if (rv.read_len < (1ull << 30)) {
            const uint32_t L32 = (uint32_t)rv.read_len;
            const uint64_t t0u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)t0 >> 32)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)t0);
            uint64_t bo;
            (void)fastdiv(t0u, rv.read_len, rv.read_len_magic, &bo);
            uint32_t sft = (uint32_t)bo + (uint32_t)q0; // < L + T
            if (L32 >= (uint32_t)T) {
                sft = sft >= L32 ? sft - L32 : sft;
            } else { // < 2 T <= 2^14: a float reciprocal is exact to +-1
                const uint32_t qe = (uint32_t)((float)sft * __builtin_amdgcn_rcpf((float)L32));
                int32_t r = (int32_t)(sft - __umul24(qe, L32));
                r = r < 0 ? r + (int32_t)L32 : r;
                r = r >= (int32_t)L32 ? r - (int32_t)L32 : r;
                sft = (uint32_t)r;
            }
            o = sft;
        } else {
            (void)fastdiv((uint64_t)p0, rv.read_len, rv.read_len_magic, &o);
        }
        if (rv.read_len >= (uint64_t)(k + S)) {
            // offsets o, o+1, ... wrap at most once inside the lane's S windows and the windows after the wrap are
            // whole again: the invalid ones are exactly those at offsets (L-k, L-1], a single run of bits
            // [a, b] of window indices (32-bit arithmetic for every read length below 2^30)
            auto run_mask = [&](auto L, auto oo) {
                const auto ra = L - k + 1 - oo, rb = L - 1 - oo;
                const int64_t lim = total - p0; // windows that start inside the chunk
                uint32_t m = lim >= S ? ((S >= 32) ? 0xFFFFFFFFu : ((1u << S) - 1u)) : (lim > 0 ? ((1u << lim) - 1u) : 0u);
                if (ra < S) {
                    const uint32_t lo_bit = ra > 0 ? (uint32_t)ra : 0u;
                    const uint32_t hi_bit = rb < S - 1 ? (uint32_t)rb : (uint32_t)(S - 1);
                    m &= ~(((2u << hi_bit) - 1u) & ~((1u << lo_bit) - 1u));
                }
                return m;
            };
            valid = rv.read_len < (1ull << 30) ? run_mask((int32_t)rv.read_len, (int32_t)o)
                                               : run_mask((int64_t)rv.read_len, (int64_t)o);
        } else {
#pragma unroll
            for (int j = 0; j < S; ++j) {
                uint64_t oj = o + j;
                if (oj >= rv.read_len)
                    oj -= rv.read_len;
                if (oj + k <= rv.read_len && p0 + j < total)
                    valid |= 1u << j;
            }
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
        } else {       // no read start inside (p, p+k-1] (util.py:72: no k-mer spans reads).  A start at relative
            // position b kills the windows j in [b - k + 1, b - 1]; a lane sees few starts (reads are longer than
            // a few bases), so it walks the set bits instead of testing every window.
            uint64_t X = B >> 1; // bit i: a start at relative position i + 1
            if (S + k - 2 < 64)
                X &= (1ull << (S + k - 2)) - 1ull; // starts beyond the last window's reach
            uint32_t dead = 0;
            while (X) {
                const int b = __builtin_ctzll(X) + 1;
                X &= X - 1ull;
                const int lo_j = b - k + 1 > 0 ? b - k + 1 : 0, hi_j = b - 1 < S - 1 ? b - 1 : S - 1;
                if (lo_j <= hi_j)
                    dead |= ((2u << hi_j) - 1u) & ~((1u << lo_j) - 1u);
            }
            const int64_t lim = total - p0 - (k - 1); // windows that end inside the chunk
            const uint32_t inside = lim >= S ? ((S >= 32) ? 0xFFFFFFFFu : ((1u << S) - 1u)) : (lim > 0 ? ((1u << lim) - 1u) : 0u);
            valid = inside & ~dead;
        }
    }
    if (win) {
        win->lo = lo;
        win->hi = hi;
    }
#pragma unroll
    for (int j = 0; j < S; ++j)
        q[j] = (j == 0 ? lo : ((lo >> (2 * j)) | (hi << (64 - 2 * j)))) & tc.kmask;
    return valid;
}

template <int S, int MODE>
__device__ __forceinline__ uint32_t tile_kmers(const ReadsView &rv, const TileConst &tc, int64_t tile,
                                               int k, TileSmem<S> &sm, uint64_t (&q)[S], const int tid)
{
    TileRaw raw;
    tile_load_vec<S, MODE>(rv, tc, tile, tid, raw);
    return tile_kmers<S, MODE>(rv, tc, tile, k, sm, q, tid, raw);
}

template <int S, int MODE>
__device__ __forceinline__ uint32_t tile_kmers(const ReadsView &rv, const TileConst &tc, int64_t tile,
                                               int k, TileSmem<S> &sm, uint64_t (&q)[S])
{
    return tile_kmers<S, MODE>(rv, tc, tile, k, sm, q, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Packed tiles for reads of one length L (radix pass 1).  A position-based tile computes a window at EVERY base
// position and discards those that cross a read boundary: k - 1 of every L (20 % at L = 150, k = 31) go through the
// division by the modulo and the ranking for nothing.  A packed tile is made of WHOLE reads instead: a 256-thread group
// owns pk_rpt consecutive reads, pk_lpr lanes per read, and lane (r, s) takes the windows s * pk_S .. s * pk_S + pk_S - 1
// of read r (pk_S = ceil(W / pk_lpr) <= 16, W = L - k + 1; the last lane of a read may hold fewer) — every window
// a lane computes is a real k-mer, blocks of pass 1 hold ~17 % more k-mers, and there are as many fewer of them.
// The bytes of the tile's reads (pk_rpt * L <= 8176) are staged like the flat tile's: 16-byte loads from the
// 16-byte-aligned address below the first read, two vectors per thread at most.
// Host side: map_reads_common sets the geometry (or pk_rpt = 0: position-based tiles) — reads shorter than 2 k or so
// and reads longer than 4 KB keep the flat path.
// ------------------------------------------------------------------------------------------------
struct TilePackedSmem {
    static constexpr int NV = 512;     // 16-base words staged per tile
    uint32_t lut[256];
    uint32_t codes[NV + 4];
};

struct TilePackedRaw {
    uint32_t w[2][4];
};

template <bool C2 = false>
__device__ __forceinline__ void tile_packed_load(const ReadsView &rv, int64_t tile, const int tid, TilePackedRaw &raw)
{
    const int64_t first = tile * (int64_t)rv.pk_rpt * (int64_t)rv.read_len; // byte of the tile's first read
    const int64_t base = first & ~(int64_t)15;
    const uint32_t nbytes = (uint32_t)(first - base) + rv.pk_rpt * (uint32_t)rv.read_len;
    const bool aligned = (((uintptr_t)rv.bases) & 15u) == 0;
    if constexpr (C2) { // 2-bit codes: the tile's words (16 positions each) from the aligned position below its first read
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(rv.bases);
        const int64_t n_words = (rv.total + 15) >> 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t wo = (uint32_t)(tid + h * 256);
            const int64_t wi = (base >> 4) + wo;
            raw.w[h][0] = (wo * 16u < nbytes && wi < n_words) ? __builtin_nontemporal_load(cw + wi) : 0u;
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t off = (uint32_t)(tid + h * 256) * 16u;
        const int64_t p = base + off;
        raw.w[h][0] = raw.w[h][1] = raw.w[h][2] = raw.w[h][3] = 0u;
        if (off < nbytes) {
            if (aligned && p + 16 <= rv.total) {
                u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rv.bases + p));
                raw.w[h][0] = x[0]; raw.w[h][1] = x[1]; raw.w[h][2] = x[2]; raw.w[h][3] = x[3];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    raw.w[h][i] = tile_load_bytes4(rv, p + i * 4);
            }
        }
    }
}

// Stage the tile's bytes as 2-bit codes in LDS (bytes before the first read / past the chunk are staged as code 0,
// unflagged).  A workgroup barrier must follow before tile_packed_fetch; one must lie between the previous tile's
// fetch and this call.
template <bool C2 = false>
__device__ __forceinline__ void tile_packed_stage(const ReadsView &rv, int64_t tile, TilePackedSmem &sm, const int tid,
                                                  const TilePackedRaw &raw)
{
    const uint32_t L = (uint32_t)rv.read_len;
    const int64_t first = tile * (int64_t)rv.pk_rpt * (int64_t)L;
    const int64_t base = first & ~(int64_t)15;
    const uint32_t delta = (uint32_t)(first - base);
    const uint32_t nbytes = delta + rv.pk_rpt * L;
    if constexpr (C2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t wo = (uint32_t)(tid + h * 256);
            if (wo * 16u < nbytes)
                sm.codes[wo] = raw.w[h][0];
        }
        if (tid < 4)
            sm.codes[((nbytes + 15u) >> 4) + (uint32_t)tid] = 0u;
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t off = (uint32_t)(tid + h * 256) * 16u;
        if (off < nbytes) {
            const int64_t p = base + off;
            uint32_t code = 0, bad = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t c = (raw.w[h][i >> 2] >> (8 * (i & 3))) & 0xFFu;
                const uint32_t l = sm.lut[c];
                bad |= (l == 0xFFu ? 1u : 0u) << i;
                code |= (l & 3u) << (2 * i);
            }
            sm.codes[off >> 4] = code;
            if (bad) {
                // only bytes of the tile's own reads inside the chunk count (the tile's neighbours report their own)
                const int64_t lo = first > p ? first - p : 0, hi = rv.total - p < (int64_t)(nbytes - off) ? rv.total - p : (int64_t)(nbytes - off);
                uint32_t m = hi >= 16 ? 0xFFFFu : (hi > 0 ? (1u << hi) - 1u : 0u);
                m &= lo >= 16 ? 0u : ~((1u << lo) - 1u);
                bad &= m;
                if (bad)
                    atomicMin(rv.first_bad, (unsigned long long)(p + __builtin_ctz(bad)));
            }
        }
    }
    if (tid < 4) // (a lane's 16-window fetch may touch up to three words past the last staged one)
        sm.codes[((nbytes + 15u) >> 4) + (uint32_t)tid] = 0u;
}

// Lane (r, s): the windows s * S .. of read r from the staged codes; returns the mask of the real ones.
__device__ __forceinline__ uint32_t tile_packed_fetch(const ReadsView &rv, const TileConst &tc, int64_t tile,
                                                      const TilePackedSmem &sm, uint64_t (&q)[16], const int tid,
                                                      TileWin *win = nullptr)
{
    const uint32_t L = (uint32_t)rv.read_len;
    const int64_t first = tile * (int64_t)rv.pk_rpt * (int64_t)L;
    const uint32_t delta = (uint32_t)(first - (first & ~(int64_t)15));
    const uint32_t r = ((uint32_t)tid * rv.pk_inv) >> 16;          // tid / pk_lpr (exact for tid < 256)
    const uint32_t s = (uint32_t)tid - r * rv.pk_lpr;
    const uint32_t o = s * rv.pk_S;                                // first window of the lane inside its read
    const int64_t read = tile * (int64_t)rv.pk_rpt + r;
    uint32_t nwin = 0;
    if (r < rv.pk_rpt && read < rv.n_reads && o < rv.pk_W)
        nwin = rv.pk_W - o < rv.pk_S ? rv.pk_W - o : rv.pk_S;
    const uint32_t q0 = delta + r * L + o;                         // position inside the staged bytes
    const uint32_t wi = (r < rv.pk_rpt ? q0 : 0u) >> 4;
    // (unlike the flat tile's lanes, which start at word boundaries, a lane may start anywhere inside a word: 16 windows
    // of up to 31 bases from offset <= 15 need 61 bases = four words)
    const uint32_t c0 = sm.codes[wi], c1 = sm.codes[wi + 1], c2 = sm.codes[wi + 2], c3 = sm.codes[wi + 3];
    const int sh = (q0 & 15) * 2;
    uint64_t lo = ((uint64_t)c1 << 32) | c0, hi = ((uint64_t)c3 << 32) | c2;
    if (sh) {
        lo = (lo >> sh) | (hi << (64 - sh));
        hi >>= sh;
    }
    if (win) {
        win->lo = lo;
        win->hi = hi;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
        q[j] = (j == 0 ? lo : ((lo >> (2 * j)) | (hi << (64 - 2 * j)))) & tc.kmask;
    return nwin >= 16 ? 0xFFFFu : (1u << nwin) - 1u;
}

// Returns the lane's windows (q[0 .. pk_S)) and the mask of the real ones.  TOPBAR as in tile_kmers.
template <bool TOPBAR, bool C2 = false>
__device__ __forceinline__ uint32_t tile_packed_kmers(const ReadsView &rv, const TileConst &tc, int64_t tile, int k,
                                                      TilePackedSmem &sm, uint64_t (&q)[16], const int tid,
                                                      const TilePackedRaw &raw, TileWin *win = nullptr)
{
    (void)k;
    if (TOPBAR)
        __syncthreads(); // every wave has finished reading the previous tile's LDS words
    tile_packed_stage<C2>(rv, tile, sm, tid, raw);
    __syncthreads();
    return tile_packed_fetch(rv, tc, tile, sm, q, tid, win);
}
