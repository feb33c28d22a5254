// kmm_gpu_inflate.hpp — part of libkmm: BGZF members inflated ON THE GPU (included by kmm.hip; compiled by itself with g++
// in tests/test_host.py, where the very same decoder runs on the CPU against zlib).
//
// Why.  BASELINE configs[4] is "gzipped .fq input overlapped with compute" (reference Readme.md:11: ".fa, .fq, .fa.gz, or
// fq.gz"; the igzip reader the reference reached for, kmer_mapper/util.py:78-101).  The host inflates BGZF at ~0.7 GB/s per
// core; the 16 cores of a rank deliver 10.8 GB/s of FASTQ while the GPU maps 40 times that.  A BGZF file (bgzip, htslib)
// is a chain of INDEPENDENT gzip members of at most 64 KiB of data each, every one with its compressed size in the header
// and CRC32 + ISIZE in the trailer: thousands of independent deflate streams per batch — work for the GPU.  The
// compressed bytes cross PCIe (a quarter of the raw ones), the members are inflated in HBM, and the raw FASTQ goes
// straight into the device-side record parser (kmm_records.hpp).
//
// How.  One THREAD per member — not a wavefront: a deflate stream is a serial chain (the position of every symbol depends
// on the one before), lanes cannot share one; but a batch holds tens of thousands of members (3 GB of FASTQ = 50 000),
// so every lane of every wavefront gets a stream of its own.  A lane's time is memory round trips, not arithmetic, and the
// design follows from that (inflate_stream below): the primary Huffman tables in LDS, lane-interleaved; block headers, symbol
// decoding and match copies as separate phases that the 64 lanes of a wavefront pass through together; the input through a
// 16-byte register FIFO; matches copied up to eight at a time with one request per source and at most two per destination.
// Stored, fixed and dynamic blocks (RFC 1951); every access is bounds-checked against the member's ISIZE / compressed size,
// so a damaged member ends in an error code, never in a stray access; the CRC32 of the output is checked on the device too
// (k_crc_bgzf behind the inflater: slicing-by-8, the tables in LDS, four lanes per member whose registers are combined in GF(2)).
// The decoder is restated from RFC 1951 / RFC 1952 and the BGZF section of the SAM specification; no code taken.
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define KMM_HD __host__ __device__
#else
#define KMM_HD
#endif

namespace kmm_gz {

// Decoding tables: 16-bit entries.  The PRIMARY tables — indexed by the next LIT_PB / DIST_PB bits of the stream — live in
// LDS on the GPU (256 + 32 entries + 32 words of construction state = 640 bytes per lane, 40 KB per wavefront: FOUR
// wavefronts per CU), codes longer than that go through a link to a SECONDARY table in the lane's scratch area in HBM.
//   direct entry   bits 0..3 code length (subtables: length beyond the primary bits; 0 = no code), bits 4..12 symbol
//   link           bit 15, bits 4..14 offset of the subtable in the secondary table, bits 0..3 its index bits
// On the GPU entry x of lane i is word x * 64 + i of the wavefront's LDS area (PS = 64): whatever the 64 lanes' indexes are,
// they fall into 32 different banks (two lanes share a 32-bit word).  With one contiguous table per lane every lane's table
// started in the same bank, and lanes looking up the same frequent symbol — FASTQ has few — queued up behind each other.
// Table width against residency, measured (profiles/r05/gz_phase_table_widths.txt; 49 939 members = 3.26 GB of FASTQ):
// 9 / 6 bits (76 KB, two wavefronts per CU) 65.6 ms, 8 / 6 bits (three) 67.6 ms, 8 / 5 bits (four) 45.9 ms — a lane waits on
// memory most of its time, and what hides that is another wavefront on the CU, not a rarer subtable lookup.
#ifndef KMM_GZ_LIT_PB
#define KMM_GZ_LIT_PB 8
#endif
#ifndef KMM_GZ_DIST_PB
#define KMM_GZ_DIST_PB 5
#endif
constexpr int LIT_PB = KMM_GZ_LIT_PB, DIST_PB = KMM_GZ_DIST_PB;
constexpr int PRIM_LIT = 1 << LIT_PB, PRIM_DIST = 1 << DIST_PB, PRIM_TMP = 32;
constexpr int PRIM_WORDS = PRIM_LIT + PRIM_DIST + PRIM_TMP;                                          // uint16 per lane (LDS)
#if defined(KMM_GZ_EXPERIMENT_SEC_IN_LDS) || defined(KMM_GZ_EXPERIMENT_PAD_LDS) // (tools/gz_phase.py: what do the subtables' trips cost?)
constexpr int SEC_LIT = 512, SEC_DIST = 384, SEC_WORDS = SEC_LIT + SEC_DIST;
#else
constexpr int SEC_LIT = 1024, SEC_DIST = 1024, SEC_WORDS = SEC_LIT + SEC_DIST;                     // uint16 per lane (HBM)
#endif
constexpr uint32_t LINK = 0x8000u;
#if defined(__HIP_DEVICE_COMPILE__)
constexpr int PS = 64; // stride between a lane's consecutive primary entries
#else
constexpr int PS = 1;
#endif
constexpr int LENS_WORDS = 20; // 320 code lengths, 4 bits each

enum Err {
    OK = 0,
    E_HEADER = 1,      // not a BGZF member header / sizes inconsistent
    E_BTYPE = 2,       // reserved block type
    E_STORED = 3,      // stored block: LEN / NLEN mismatch or beyond the input
    E_CODELEN = 4,     // dynamic block: code-length code / repeat without a previous length / too many lengths
    E_TABLE = 5,       // over-subscribed or incomplete code, no end-of-block code, table larger than its scratch
    E_SYMBOL = 6,      // a bit pattern without a code / length or distance symbol out of range
    E_DISTANCE = 7,    // distance reaches before the start of the member's output
    E_OUTPUT = 8,      // more output than ISIZE says
    E_INPUT = 9,       // compressed data ended inside a block
    E_ISIZE = 10,      // fewer bytes than ISIZE says
    E_CRC = 11,        // CRC32 of the inflated bytes differs from the trailer
};

KMM_HD inline uint32_t rev_bits(uint32_t v, int n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return n > 0 ? __brev(v) >> (32 - n) : 0u;
#else
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
#endif
}

KMM_HD inline uint32_t rd16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
KMM_HD inline uint32_t rd32(const uint8_t *p) { return rd16(p) | (rd16(p + 2) << 16); }

// code length i of a block header (4 bits each, 16 per word)
KMM_HD inline uint32_t lens_get(const uint64_t *w, int i) { return (uint32_t)(w[i >> 4] >> (4 * (i & 15))) & 15u; }

// Decoding tables of a canonical Huffman code (RFC 1951 3.2.2) from its code lengths lens[at .. at + n) (4 bits each):
// prim[0 .. 2^pb) (stride PS), subtables in sec[0 .. sec_cap).  tmp: 32 words (stride PS) — symbols per length and the next
// code of every length: the two arrays the sweeps index by a code length live in LDS, not in the lane's private memory,
// where every step of such a chain is a round trip to HBM / L2.  Two sweeps over the symbols: the primary entries of the
// prefixes that need a subtable first hold the longest code length under them, then the link.  Returns OK or E_TABLE.
// allow_incomplete: a distance code may consist of one code.
KMM_HD inline int build_table(const uint64_t *lens, int at, int n, int pb, uint16_t *prim, uint16_t *tmp, uint16_t *sec, int sec_cap,
                              bool allow_incomplete)
{
    uint16_t *count = tmp, *nx = tmp + 16 * PS;
    for (int i = 0; i < 16; ++i)
        count[i * PS] = 0;
    {
        uint64_t w = 0;
        for (int i = 0; i < n; ++i) {
            const int j = at + i;
            if (i == 0 || (j & 15) == 0)
                w = lens[j >> 4] >> (4 * (j & 15));
            count[(int)(w & 15u) * PS]++;
            w >>= 4;
        }
    }
    if (count[0] == n)
        return E_TABLE;
    int left = 1; // Kraft: code space still free
    bool any_long = false;
    {
        uint32_t code = 0, prev = 0;
        for (int l = 1; l <= 15; ++l) {
            const uint32_t c = count[l * PS];
            left = (left << 1) - (int)c;
            if (left < 0)
                return E_TABLE;
            code = (code + prev) << 1;
            nx[l * PS] = (uint16_t)code;
            prev = c;
            any_long = any_long || (l > pb && c);
        }
    }
    if (left > 0 && !allow_incomplete)
        return E_TABLE;
    const uint32_t psize = 1u << pb, pmask = psize - 1u;
    for (uint32_t i = 0; i < psize; ++i)
        prim[i * PS] = 0;
    if (any_long) {
        uint64_t w = 0;
        for (int s = 0; s < n; ++s) { // sweep 1: longest code under every primary index
            const int j = at + s;
            if (s == 0 || (j & 15) == 0)
                w = lens[j >> 4] >> (4 * (j & 15));
            const int l = (int)(w & 15u);
            w >>= 4;
            if (!l)
                continue;
            const uint32_t r = rev_bits(nx[l * PS]++, l);
            if (l > pb && (uint32_t)l > prim[(r & pmask) * PS])
                prim[(r & pmask) * PS] = (uint16_t)l;
        }
        uint32_t total = 0;
        for (uint32_t i = 0; i < psize; ++i)
            if (prim[i * PS]) {
                const uint32_t sb = (uint32_t)prim[i * PS] - (uint32_t)pb;
                if (!sec || total + (1u << sb) > (uint32_t)sec_cap || total >= 2048u)
                    return E_TABLE;
                prim[i * PS] = (uint16_t)(LINK | (total << 4) | sb);
                total += 1u << sb;
            }
        for (uint32_t i = 0; i < total; ++i)
            sec[i] = 0;
        for (int l = 1; l <= 15; ++l) // back to every length's first code
            nx[l * PS] = (uint16_t)(nx[l * PS] - count[l * PS]);
    }
    uint64_t w = 0;
    for (int s = 0; s < n; ++s) { // sweep 2: the entries
        const int j = at + s;
        if (s == 0 || (j & 15) == 0)
            w = lens[j >> 4] >> (4 * (j & 15));
        const int l = (int)(w & 15u);
        w >>= 4;
        if (!l)
            continue;
        const uint32_t r = rev_bits(nx[l * PS]++, l);
        if (l <= pb) {
            const uint16_t e = (uint16_t)(((uint32_t)s << 4) | (uint32_t)l);
            for (uint32_t i = r; i < psize; i += 1u << l)
                prim[i * PS] = e;
        } else {
            const uint32_t link = prim[(r & pmask) * PS], off = (link >> 4) & 0x7FFu, sb = link & 15u;
            const uint16_t e = (uint16_t)(((uint32_t)s << 4) | (uint32_t)(l - pb));
            for (uint32_t i = r >> pb; i < (1u << sb); i += 1u << (l - pb))
                sec[off + i] = e;
        }
    }
    return OK;
}

// Bit reader over in[0, n): a 64-bit buffer topped up 32 bits at a time from a 16-byte block held in registers; the NEXT
// block is requested when the current one is opened — a block lasts ~14 symbols, so the load's latency lies a whole block
// behind its use.  Bytes behind the end read as zero (the consumed count is checked at the end of every block).
struct Bits {
    const uint8_t *in;
    uint32_t n, pos;     // next byte not yet requested
    uint64_t buf;
    int cnt;             // valid bits in buf
    uint64_t flo, fhi;   // the rest of the open block, lowest word next
    int fw;              // 32-bit words left in it
    uint64_t nlo, nhi;   // the block behind it (bytes [pos - 16, pos)) as block_load returned it: block_fix when it is taken
};

// The 16-byte blocks of the input.  block_load: ONE request whatever pos is and never an access outside in[0, n) — at the end
// of the input the 16 bytes that END there are loaded (a byte loop there was sixteen dependent round trips, and the 64 lanes of
// a wavefront reach the ends of their members in different rounds).  block_fix turns what block_load returned into bytes
// [pos, pos + 16) with zeros behind the end.  (n >= 16: inflate_stream pads a shorter input.)  Two functions because the fix USES the loaded registers: it belongs where the
// block is taken (bits_advance), not where it is requested — the wavefront waits for a load at the first use of its registers.
KMM_HD inline void block_load(const uint8_t *in, uint32_t n, uint32_t pos, uint64_t &lo, uint64_t &hi) // n >= 16
{
    const uint32_t p = pos + 16u <= n ? pos : n - 16u;
    memcpy(&lo, in + p, 8); // (unaligned loads)
    memcpy(&hi, in + p + 8, 8);
}

KMM_HD inline void block_fix(uint32_t n, uint32_t pos, uint64_t &lo, uint64_t &hi)
{
    if (pos + 16u > n) {
        const uint32_t k = pos - (n - 16u); // bytes of the load that lie in front of pos (>= 1)
        const uint32_t sh = 8u * (k & 7u);
        if (k >= 16u) {
            lo = 0;
            hi = 0;
        } else if (k >= 8u) {
            lo = hi >> sh;
            hi = 0;
        } else {
            lo = (lo >> sh) | (hi << (64u - sh)); // (0 < sh < 64 here)
            hi >>= sh;
        }
    }
}

KMM_HD inline void load16(const uint8_t *in, uint32_t n, uint32_t pos, uint64_t &lo, uint64_t &hi)
{
    block_load(in, n, pos, lo, hi);
    block_fix(n, pos, lo, hi);
}

KMM_HD inline void bits_start(Bits &b, uint32_t at) // (re)start reading at byte `at`
{
    uint64_t lo, hi;
    load16(b.in, b.n, at, lo, hi);
    b.buf = lo;
    b.cnt = 64;
    b.flo = hi;
    b.fhi = 0;
    b.fw = 2;
    block_load(b.in, b.n, at + 16u, b.nlo, b.nhi); // (nlo / nhi: as loaded — fixed when the block is taken)
    b.pos = at + 32u;
}

KMM_HD inline void bits_refill(Bits &b)
{
    if (b.cnt <= 32) {
        if (b.fw == 0) {
            b.flo = b.nlo;
            b.fhi = b.nhi;
            block_fix(b.n, b.pos - 16u, b.flo, b.fhi);
            b.fw = 4;
            block_load(b.in, b.n, b.pos, b.nlo, b.nhi);
            b.pos += 16u;
        }
        const uint32_t x = (uint32_t)b.flo;
        b.flo = (b.flo >> 32) | (b.fhi << 32);
        b.fhi >>= 32;
        --b.fw;
        b.buf |= (uint64_t)x << b.cnt;
        b.cnt += 32;
    }
}

KMM_HD inline uint32_t bits_take(Bits &b, int k) // k <= 16 bits, after a refill
{
    const uint32_t v = (uint32_t)b.buf & ((1u << k) - 1u);
    b.buf >>= k;
    b.cnt -= k;
    return v;
}

// The symbol loop's refill: 32 bits more from the open block — never a memory access.  The loop runs only while buffer and
// open block together hold a whole symbol (48 bits), so there is a word to take whenever fewer than 33 bits are buffered
// and more are needed.
KMM_HD inline void bits_refill_local(Bits &b)
{
    if (b.cnt <= 32 && b.fw > 0) {
        const uint32_t x = (uint32_t)b.flo;
        b.flo = (b.flo >> 32) | (b.fhi << 32);
        b.fhi >>= 32;
        --b.fw;
        b.buf |= (uint64_t)x << b.cnt;
        b.cnt += 32;
    }
}

// The open block no longer holds a symbol (fewer than 48 bits with the buffer: at most one word is left, it moves into
// the buffer): the block requested one block-time ago takes its place (bits_advance: registers only — where the end of the
// input is fixed up) and the one behind it is requested (bits_prefetch).
KMM_HD inline void bits_advance(Bits &b)
{
    if (b.fw > 0) {
        b.buf |= (uint64_t)(uint32_t)b.flo << b.cnt;
        b.cnt += 32;
    }
    b.flo = b.nlo;
    b.fhi = b.nhi;
    block_fix(b.n, b.pos - 16u, b.flo, b.fhi);
    b.fw = 4;
    b.pos += 16u;
}

KMM_HD inline void bits_prefetch(Bits &b) // nlo / nhi = the block at pos - 16, as loaded
{
    block_load(b.in, b.n, b.pos - 16u, b.nlo, b.nhi);
}

KMM_HD inline void bits_next_block(Bits &b)
{
    bits_advance(b);
    bits_prefetch(b);
}

// bytes of the input the decoder has really used (the buffer, the open block and the block behind it hold bytes it has not)
KMM_HD inline uint32_t bits_consumed_bytes(const Bits &b)
{
    return b.pos - 16u - 4u * (uint32_t)b.fw - ((uint32_t)b.cnt >> 3);
}

// the first n (1 .. 16) of the 16 bytes in (lo, hi) -> q, exactly: at most two stores, the second one overlapping the first
// (a byte loop is up to seven store instructions per match, and on the GPU what a wavefront of 64 scattered lanes pays for
// is the NUMBER of requests, not their size)
KMM_HD inline void store_upto16(uint8_t *q, uint64_t lo, uint64_t hi, uint32_t n)
{
    if (n >= 8u) {
        memcpy(q, &lo, 8);
        if (n > 8u) {
            const uint32_t s = 8u * (n - 8u); // 8 .. 64
            const uint64_t tail = s == 64u ? hi : (lo >> s) | (hi << (64u - s));
            memcpy(q + n - 8u, &tail, 8);
        }
    } else if (n >= 4u) {
        const uint32_t a = (uint32_t)lo, t = (uint32_t)(lo >> (8u * (n - 4u)));
        memcpy(q, &a, 4);
        memcpy(q + n - 4u, &t, 4);
    } else if (n >= 2u) {
        const uint16_t a = (uint16_t)lo, t = (uint16_t)(lo >> (8u * (n - 2u)));
        memcpy(q, &a, 2);
        memcpy(q + n - 2u, &t, 2);
    } else if (n == 1u) {
        q[0] = (uint8_t)lo;
    }
}

// 16 bytes at p (any alignment) in one request
KMM_HD inline void load16u(const uint8_t *p, uint64_t &lo, uint64_t &hi)
{
    struct Pair { uint64_t a, b; } v;
    memcpy(&v, p, 16);
    lo = v.a;
    hi = v.b;
}

// One deflate stream in[0, n_in) -> out[0, n_out) exactly (the caller's buffer has 16 readable bytes of slack behind
// out[n_out)).  prim: PRIM_WORDS uint16 (LDS on the GPU), sec: SEC_WORDS uint16 of scratch, list: LIST_CAP uint64 of scratch.
// Returns OK or an Err.
//
// On the GPU every lane of a wavefront decodes a member of its own; the lanes are in different places of their streams but
// execute ONE instruction stream, and a lane's time is the number of dependent memory round trips it pays for (~2 us each
// to HBM / L2 at the two to three wavefronts per CU a batch of members gives), not its arithmetic.  Decoding and copying are
// therefore SEPARATED:
//   phase A  decode up to DECODE_RUN symbols: the tables are in LDS and the input arrives in 16-byte blocks requested a
//            block ahead, so a symbol costs no trip to memory at all; a literal is stored right away (a store waits for
//            nothing), a match only RESERVES its bytes of the output and is written down — (position, length, distance) —
//            in the lane's list;
//   phase B  the list is carried out in order, up to four matches per step whose sources all lie in front of the first one's
//            destination (they cannot depend on each other): their loads leave together, ONE round trip serves four
//            matches; a match that reaches into the bytes of the one before it (or is longer than 16 bytes, or repeats a
//            pattern shorter than 16) takes a step of its own.
// A block header (code lengths, table construction: ~0.1 ms of serial work per lane, the first one of all 64 lanes at once)
// and the rare code longer than the primary tables' index (a subtable in HBM) are the other costs.
// History (profiles/r05/bgzf_e2e_*.txt, bgzf_v*_kernel_stats.csv, gz_phase_*.txt; 3.26 GB of FASTQ in 50 000 members): v1, a
// plain symbol loop, the match copy inside it, all tables in HBM: ~75 ms per 23 000 members; v2, the same as a
// one-access-per-turn state machine: 74 ms; v3, primary tables in LDS, up to six literals per turn: 48 ms; v4 / v5, decode and
// copy phases: 34 ms per wavefront whatever the batch, 84 ms for the whole file; v6 (phase timers, tools/gz_phase.py, said:
// decoding 14 ms, copies 17 ms, CRC 2 ms, headers 0.4 ms per lane): headers as a phase of their own, lane-interleaved tables,
// the length / distance bases as arithmetic, at most two stores per match, one request per source and per two list entries,
// 8 / 5-bit tables for four wavefronts per CU: 46 ms for the whole file; v7: what bounds the phases then — not their stores,
// not the input's latency, not the table widths: chains of dependent instructions at one wavefront per SIMD; v8 therefore
// shortens the chains: ONE exit from the symbol loop (errors deferred to the round's end: 330 -> 230 instructions per
// symbol, 15.0 -> 11.9 ms per lane) and ONE kind of work in the copy phase (every match as pieces of at most 16 bytes from
// the head of the list: 11.8 -> 9.0 ms): 37 ms for the whole file = 87 GB/s of FASTQ out.
constexpr int DECODE_RUN = 1024;    // symbols per phase A
constexpr int LIST_CAP = 512;       // matches per phase A (8 bytes each)
#ifndef KMM_GZ_GROUP
#define KMM_GZ_GROUP 6
#endif
constexpr int GROUP = KMM_GZ_GROUP; // matches per step of phase B (even)
constexpr int LIST_ALLOC = LIST_CAP + GROUP; // (a step's entry loads may reach GROUP entries behind the last one)
enum State { S_HDR = 0, S_SYM, S_DONE };
constexpr int SCRATCH_BYTES = LIST_ALLOC * 8 + SEC_WORDS * 2; // per lane, in HBM: the match list, then the subtables

// Phase timers (tools/gz_phase.hip builds with -DKMM_GZ_TIMERS): 100 MHz ticks a lane spends in [0] block headers,
// [1] symbol decoding, [2] the match copies; [3] subtable lookups (literal / length code: low half, distance code: high half);
// [4] block headers seen (low half), symbols (high half), [5] rounds of the outer loop, [6] steps of the copy
// phase, [7] matches.
#if defined(KMM_GZ_TIMERS)
#if defined(__HIP_DEVICE_COMPILE__)
#define KMM_GZ_NOW() wall_clock64()
#else
#define KMM_GZ_NOW() 0ull
#endif
#define KMM_GZ_T(slot)                                                                                                \
    do {                                                                                                              \
        if (tm) {                                                                                                     \
            const unsigned long long now_ = KMM_GZ_NOW();                                                             \
            tm[slot] += now_ - t_last;                                                                                \
            t_last = now_;                                                                                            \
        }                                                                                                             \
    } while (0)
#define KMM_GZ_COUNT(slot, v) do { if (tm) tm[slot] += (v); } while (0)
#define KMM_GZ_T0 unsigned long long t_last = tm ? KMM_GZ_NOW() : 0ull
#else
#define KMM_GZ_T(slot) do { } while (0)
#define KMM_GZ_COUNT(slot, v) do { } while (0)
#define KMM_GZ_T0 do { } while (0)
#endif

// length symbol 257 + s -> (base, extra bits); distance symbol d -> (base, extra bits): RFC 1951 3.2.5's two tables as
// arithmetic — a table indexed by a lane's own symbol is a gather from memory, a round trip per match
KMM_HD inline uint32_t len_extra_bits(uint32_t s) { return (s < 8u || s == 28u) ? 0u : (s - 4u) >> 2; }
KMM_HD inline uint32_t len_base_of(uint32_t s) { return s < 8u ? 3u + s : s == 28u ? 258u : 3u + ((4u + (s & 3u)) << ((s - 4u) >> 2)); }
KMM_HD inline uint32_t dist_extra_bits(uint32_t d) { return d < 4u ? 0u : (d >> 1) - 1u; }
KMM_HD inline uint32_t dist_base_of(uint32_t d) { return d < 4u ? 1u + d : 1u + ((2u + (d & 1u)) << ((d >> 1) - 1u)); }

KMM_HD inline void lens_put(uint64_t *w, int i, uint32_t v)
{
    if ((i & 15) == 0)
        w[i >> 4] = (uint64_t)v;
    else
        w[i >> 4] |= (uint64_t)v << (4 * (i & 15));
}

// One block header at the bit reader's position (RFC 1951 3.2.3-3.2.7).  A stored block is copied right here.  Returns OK or
// an Err; *stored = the block was a stored one (no tables made), *final = BFINAL.
KMM_HD inline int block_header(Bits &b, uint32_t n_in, uint8_t *out, uint32_t n_out, uint32_t &o, uint16_t *prim, uint16_t *sec,
                               uint32_t *final, bool *stored)
{
    uint16_t *lit = prim, *dst = prim + PRIM_LIT * PS, *tmp = prim + (PRIM_LIT + PRIM_DIST) * PS;
    uint16_t *lit2 = sec, *dst2 = sec ? sec + SEC_LIT : nullptr;
    bits_refill(b);
    *final = bits_take(b, 1);
    const uint32_t type = bits_take(b, 2);
    *stored = type == 0u;
    if (type == 3u)
        return E_BTYPE;
    if (type == 0u) { // stored: to the next byte boundary, LEN, NLEN, LEN bytes
        bits_take(b, b.cnt & 7);
        bits_refill(b);
        const uint32_t slen = bits_take(b, 16);
        bits_refill(b);
        const uint32_t nlen = bits_take(b, 16);
        if ((slen ^ nlen) != 0xFFFFu)
            return E_STORED;
        const uint32_t at = bits_consumed_bytes(b); // (on a byte boundary: the bytes follow in the input as they are)
        if (at + slen > n_in)
            return E_STORED;
        if (o + slen > n_out)
            return E_OUTPUT;
        for (uint32_t j = 0; j < slen; ++j)
            out[o + j] = b.in[at + j];
        o += slen;
        bits_start(b, at + slen);
        return bits_consumed_bytes(b) > n_in ? E_INPUT : OK;
    }
    uint64_t lens[LENS_WORDS];
    int hlit = 288, hdist = 30;
    if (type == 1u) { // fixed code (RFC 1951 3.2.6): 144 x 8, 112 x 9, 24 x 7, 8 x 8 bits; 30 distance codes of 5
        for (int i = 0; i < 9; ++i) lens[i] = 0x8888888888888888ull;
        for (int i = 9; i < 16; ++i) lens[i] = 0x9999999999999999ull;
        lens[16] = 0x7777777777777777ull;
        lens[17] = 0x8888888877777777ull;
        lens[18] = 0x5555555555555555ull;
        lens[19] = 0x0055555555555555ull;
    } else {
        hlit = (int)bits_take(b, 5) + 257;
        hdist = (int)bits_take(b, 5) + 1;
        const int hclen = (int)bits_take(b, 4) + 4;
        if (hlit > 286 || hdist > 30)
            return E_CODELEN;
        // the code-length code's lengths arrive in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (5 bits each below)
        const uint64_t order_lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 |
                                  10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
        const uint64_t order_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
        uint64_t cl[2] = {0, 0}; // 19 lengths of 3 bits, stored 4 bits each
        for (int i = 0; i < hclen; ++i) {
            bits_refill(b);
            const uint32_t sym = (uint32_t)((i < 12 ? order_lo >> (5 * i) : order_hi >> (5 * (i - 12))) & 31u);
            const uint64_t v = (uint64_t)bits_take(b, 3) << (4 * (sym & 15u));
            if (sym < 16u)
                cl[0] |= v;
            else
                cl[1] |= v;
        }
        // the code-length code (codes of at most 7 bits): a 7-bit table in the literal table's place (rebuilt below)
        if (build_table(cl, 0, 19, 7, lit, tmp, nullptr, 0, false) != OK)
            return E_CODELEN;
        int i = 0;
        uint32_t prev = 0;
        while (i < hlit + hdist) {
            bits_refill(b);
            const uint32_t ce = lit[((uint32_t)b.buf & 127u) * PS];
            const int cl_len = (int)(ce & 15u);
            if (!cl_len)
                return E_CODELEN;
            bits_take(b, cl_len);
            const uint32_t sym = ce >> 4;
            if (sym < 16u) {
                lens_put(lens, i++, sym);
                prev = sym;
            } else {
                int rep;
                uint32_t v = 0;
                if (sym == 16u) {
                    if (i == 0)
                        return E_CODELEN;
                    v = prev;
                    rep = 3 + (int)bits_take(b, 2);
                } else if (sym == 17u) {
                    rep = 3 + (int)bits_take(b, 3);
                } else {
                    rep = 11 + (int)bits_take(b, 7);
                }
                if (i + rep > hlit + hdist)
                    return E_CODELEN;
                for (int j = 0; j < rep; ++j)
                    lens_put(lens, i++, v);
                prev = v;
            }
        }
        if (lens_get(lens, 256) == 0)
            return E_TABLE; // no end-of-block code
    }
    if (build_table(lens, 0, hlit, LIT_PB, lit, tmp, lit2, SEC_LIT, false) != OK ||
        build_table(lens, hlit, hdist, DIST_PB, dst, tmp, dst2, SEC_DIST, true) != OK)
        return E_TABLE;
    return OK;
}

KMM_HD inline int inflate_stream(const uint8_t *in, uint32_t n_in, uint8_t *out, uint32_t n_out, uint16_t *prim, uint16_t *sec,
                                 uint64_t *list, unsigned long long *tm = nullptr)
{
    (void)tm;
    const uint16_t *lit = prim, *dst = prim + PRIM_LIT * PS;
    const uint16_t *lit2 = sec, *dst2 = sec ? sec + SEC_LIT : nullptr;
    Bits b;
    b.in = in;
    b.n = n_in;
    if (n_in < 16u) { // (BGZF's end-of-file member: two bytes) the block loads want 16 bytes: a zero-padded copy at the far end
        uint8_t *tiny = reinterpret_cast<uint8_t *>(list + LIST_CAP - 2); // of the match list, which so little input cannot fill
        for (uint32_t j = 0; j < 16u; ++j)
            tiny[j] = j < n_in ? in[j] : (uint8_t)0;
        b.in = tiny;
        b.n = 16u;
    }
    bits_start(b, 0);
    uint32_t o = 0, final = 0;
    int state = S_HDR;
    KMM_GZ_T0;
    while (state != S_DONE) {
        KMM_GZ_COUNT(5, 1);
        // ---- block header: a phase of its own — the lanes of a wavefront that stand at a header work through it TOGETHER
        // here (inside the symbol loop each lane's header, thousands of instructions, was executed while the other 63 waited)
        if (state == S_HDR) {
            bool stored;
            const int rc = block_header(b, n_in, out, n_out, o, prim, sec, &final, &stored);
            if (rc != OK)
                return rc;
            state = stored ? (final ? S_DONE : S_HDR) : S_SYM;
            KMM_GZ_COUNT(4, 1);
        }
        KMM_GZ_T(0);
        // ---- phase A: decode until the block ends; literals are stored, matches written down
        // The input: the symbol loop proper (inner) runs on the bits at hand — buffer + open 16-byte block, both registers —
        // and never touches memory for input; when they no longer hold a whole symbol the outer loop opens the next block.
        // (Before, the refill sat inside the symbol and the compiler waited for every block load on the spot — its registers
        // are copied into loop-carried ones at the branch's end.  Moving it out changed nothing measurable, 15.0 against
        // 15.2 ms per lane, nor did leaving out this phase's stores altogether, 13.0: the symbol phase is bound by its chain of
        // dependent instructions at one wavefront per SIMD — ~1 900 clocks per symbol — not by memory;
        // profiles/r05/gz_phase_v7_*.txt.)
        // What a symbol costs, measured with one wavefront per CU (tools/gz_phase.py, profiles/r05/gz_phase_v9_*.txt): 1.04 us,
        // of which 0.35 us are the trips to the subtables in HBM — one literal / length code in fifty and one distance code in
        // ten is longer than the primary index, but SOME lane of 64 has one in nearly every turn (with the subtables in LDS,
        // which four wavefronts per CU leave no room for: 0.69 us) — and the rest the chain of ~200 dependent instructions.
        // Tried and not kept: rounds of "up to three literals from LDS, then the one symbol that needs more, with one load for
        // whichever subtable a lane needs" — 30 % slower: FASTQ's symbols are matches by three quarters, and a round's
        // instructions outweigh the trip it saves; and one block request per lane and turn with no branch around it, so that
        // no load is waited for where it is issued — 5 % slower (34.8 against 33.1 ms): sixty-four more requests per turn.
        uint32_t n_list = 0;
        int t = 0, bad = 0;
        unsigned long long n_sub = 0; // (timers build: subtable lookups, literal / length low half, distance high half)
        for (;;) {
        // ONE way out of the loop — its condition.  A symbol that cannot be (no code, a distance before the start, more
        // output than ISIZE) sets `bad` and the lane decodes on, harmlessly (every access below is in bounds whatever the
        // bits say), until the round ends and the error is returned: with a `return` at every check the compiler wove
        // thirteen exits into the loop, and the 64 lanes paid for the exec-mask and register bookkeeping of all of them on
        // every symbol (~330 instructions per symbol, most of them that).
        while (t < DECODE_RUN && state == S_SYM && n_list < (uint32_t)LIST_CAP && b.cnt + 32 * b.fw >= 48) {
            ++t;
            bits_refill_local(b);
            uint32_t e = lit[((uint32_t)b.buf & (uint32_t)(PRIM_LIT - 1)) * PS];
            if (e & LINK) { // a code longer than the primary index: its subtable (HBM)
                e = lit2[((e >> 4) & 0x7FFu) + (((uint32_t)b.buf >> LIT_PB) & ((1u << (e & 15u)) - 1u))];
                bits_take(b, LIT_PB);
                n_sub += 1ull;
            }
            const uint32_t l = e & 15u, sym = e >> 4;
            bad = bad ? bad : (l ? 0 : (int)E_SYMBOL);
            bits_take(b, (int)l);
            if (sym < 256u) {
                if (o < n_out) {
#ifndef KMM_GZ_EXPERIMENT_NO_DECODE_STORES // (tools/gz_phase.py: what do the decode phase's stores cost? output is garbage)
                    out[o] = (uint8_t)sym;
#endif
                    ++o;
                } else {
                    bad = bad ? bad : (int)E_OUTPUT;
                }
            } else if (sym == 256u) {
                bad = bad ? bad : (bits_consumed_bytes(b) > n_in ? (int)E_INPUT : 0);
                state = final ? S_DONE : S_HDR;
            } else {
                const uint32_t ls = sym <= 285u ? sym - 257u : 0u;
                const uint32_t len = len_base_of(ls) + bits_take(b, (int)len_extra_bits(ls));
                bits_refill_local(b);
                uint32_t d = dst[((uint32_t)b.buf & (uint32_t)(PRIM_DIST - 1)) * PS];
                if (d & LINK) {
                    d = dst2[((d >> 4) & 0x7FFu) + (((uint32_t)b.buf >> DIST_PB) & ((1u << (d & 15u)) - 1u))];
                    bits_take(b, DIST_PB);
                    n_sub += 1ull << 32;
                }
                const uint32_t dl = d & 15u, dcode = d >> 4, dsym = dcode <= 29u ? dcode : 0u;
                bits_take(b, (int)dl);
                const uint32_t dist = dist_base_of(dsym) + bits_take(b, (int)dist_extra_bits(dsym)); // (a refill leaves >= 33 bits: 15 + 13 fit)
                const int why = sym > 285u || !dl || dcode > 29u ? (int)E_SYMBOL : dist > o ? (int)E_DISTANCE : o + len > n_out ? (int)E_OUTPUT : 0;
                if (!why) {
#ifndef KMM_GZ_EXPERIMENT_NO_DECODE_STORES
                    list[n_list] = (uint64_t)o | ((uint64_t)len << 32) | ((uint64_t)dist << 41);
#endif
                    ++n_list;
                    o += len;
                } else {
                    bad = bad ? bad : why;
                }
            }
        }
        if (!(t < DECODE_RUN && state == S_SYM && n_list < (uint32_t)LIST_CAP))
            break;
        bits_next_block(b);
        }
        if (bad)
            return bad;
        KMM_GZ_COUNT(4, (unsigned long long)t << 32);
        KMM_GZ_COUNT(3, n_sub);
        (void)n_sub;
        KMM_GZ_T(1);
        // ---- phase B: the matches, in order; up to GROUP per step when none of them can depend on another.  The list lies in
        // HBM: the entries a step looks at were requested during the step before (two per request; entries beyond the end
        // read as an entry that ends a group), so a step costs ONE round trip — its sources'.
        // ONE kind of work: a piece of at most 16 bytes whose source ends in front of its destination.  A match that is longer,
        // or repeats a pattern shorter than itself (distance < length: runs), gives up such a piece from its head and stays at
        // the front of the list with the rest — a pattern's distance doubles with every piece (d, 2d, 4d .. bytes, then 16
        // at a time: all the same bytes).  Until v8 those matches went through a loop of their own; with 64 lanes some lane
        // was in it at nearly every step and the other 63 waited.
        uint32_t i = 0;
        uint64_t ent[GROUP];
        for (int x = 0; x < GROUP; x += 2) {
            load16u(reinterpret_cast<const uint8_t *>(list + x), ent[x], ent[x + 1]);
            ent[x] = (uint32_t)x < n_list ? ent[x] : ~0ull;
            ent[x + 1] = (uint32_t)x + 1u < n_list ? ent[x + 1] : ~0ull;
        }
        while (i < n_list) {
            const uint32_t o0 = (uint32_t)ent[0], l0 = (uint32_t)(ent[0] >> 32) & 0x1FFu, d0 = (uint32_t)(ent[0] >> 41);
            const bool pattern = d0 < l0 && d0 < 16u;
            const uint32_t pl = pattern ? d0 : (l0 < 16u ? l0 : 16u); // the piece the head entry gives up now (pl <= d0)
            const bool whole = pl == l0;
            uint32_t g = 1;
            if (whole) {
                for (; g < (uint32_t)GROUP; ++g) { // (an entry behind the end has length 511: it ends the group)
                    const uint32_t og = (uint32_t)ent[g], lg = (uint32_t)(ent[g] >> 32) & 0x1FFu, dg = (uint32_t)(ent[g] >> 41);
                    if (lg > 16u || dg < lg || og - dg + lg > o0)
                        break;
                }
            }
            const uint32_t adv = whole ? g : 0u;
            // the next step's entries, and this step's sources: all loads leave before the first is used
            uint64_t nxt[GROUP];
            for (int x = 0; x < GROUP; x += 2) {
                const uint32_t at = i + adv + (uint32_t)x;
                if (at < n_list) {
                    load16u(reinterpret_cast<const uint8_t *>(list + at), nxt[x], nxt[x + 1]);
                    nxt[x + 1] = at + 1u < n_list ? nxt[x + 1] : ~0ull;
                } else {
                    nxt[x] = nxt[x + 1] = ~0ull;
                }
            }
            if (!whole) // the rest of the head entry stays in front
                nxt[0] = (uint64_t)(o0 + pl) | ((uint64_t)(l0 - pl) << 32) | ((uint64_t)(pattern ? 2u * d0 : d0) << 41);
            uint64_t lo[GROUP], hi[GROUP];
            for (uint32_t x = 0; x < (uint32_t)GROUP; ++x) {
                lo[x] = hi[x] = 0;
                if (x < g)
                    load16u(out + (uint32_t)ent[x] - (uint32_t)(ent[x] >> 41), lo[x], hi[x]);
            }
            store_upto16(out + o0, lo[0], hi[0], pl);
            for (uint32_t x = 1; x < (uint32_t)GROUP; ++x)
                if (x < g)
                    store_upto16(out + (uint32_t)ent[x], lo[x], hi[x], (uint32_t)(ent[x] >> 32) & 0x1FFu);
            for (int x = 0; x < GROUP; ++x)
                ent[x] = nxt[x];
            i += adv;
            KMM_GZ_COUNT(6, 1);
            KMM_GZ_COUNT(7, adv);
        }
        KMM_GZ_T(2);
    }
    if (bits_consumed_bytes(b) > n_in)
        return E_INPUT;
    return o == n_out ? OK : E_ISIZE;
}

// ---- BGZF member framing (RFC 1952 + the BC extra subfield of the SAM specification, section 4.1) ----
// total size of the member at p (n bytes available), or 0 if it is not one
KMM_HD inline uint32_t bgzf_member_size(const uint8_t *p, uint64_t n)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4))
        return 0;
    const uint32_t xlen = rd16(p + 10);
    if (12ull + xlen + 8ull > n)
        return 0;
    uint32_t q = 12;
    while (q + 4u <= 12u + xlen) { // the extra subfields: SI1 SI2 SLEN data
        const uint32_t slen = rd16(p + q + 2);
        if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2u && q + 6u <= 12u + xlen) {
            const uint32_t total = rd16(p + q + 4) + 1u;
            return total >= 12u + xlen + 8u ? total : 0u;
        }
        q += 4u + slen;
    }
    return 0;
}

// CRC32 (IEEE 802.3, reflected, as gzip uses it): tables for slicing-by-8; T[k][b] = CRC of byte b followed by k zero bytes
KMM_HD inline uint32_t crc_table_entry(int k, uint32_t b)
{
    uint32_t c = b;
    for (int j = 0; j < 8; ++j)
        c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    for (int z = 0; z < k; ++z) {
        uint32_t x = c & 0xFFu;
        uint32_t t = x;
        for (int j = 0; j < 8; ++j)
            t = (t >> 1) ^ (0xEDB88320u & (0u - (t & 1u)));
        c = (c >> 8) ^ t;
    }
    return c;
}

// The CRC of a member from the CRCs of its PARTS.  The CRC register is linear in the message: after part A the register holds
// r_A; running on over part B gives r_A * x^(8 |B|) + r_B(from a zero register) in GF(2)[x] modulo the CRC's polynomial.  So the
// parts are summed independently (the first from the register's initial 0xFFFFFFFF, the others from 0), each register is moved
// forward by the bytes BEHIND its part with one multiplication, and the exclusive-or of the results is the register after the
// whole message.  Polynomials in the CRC's reflected bit order: bit 31 is x^0.
KMM_HD inline uint32_t gf2_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        p ^= b & (0u - ((a >> (31 - i)) & 1u));
        b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1u)));
    }
    return p;
}

constexpr int CRC_SHIFT_WORDS = 32, CRC_TABLE_WORDS = 8 * 256 + CRC_SHIFT_WORDS;

// X[k] = x^(8 * 2^k): the tables' tail (crcT[8 * 256 + k])
KMM_HD inline uint32_t crc_shift_table_entry(int k)
{
    uint32_t v = 0x00800000u; // x^8
    for (int i = 0; i < k; ++i)
        v = gf2_mul(v, v);
    return v;
}

// register r moved forward by n zero bytes
KMM_HD inline uint32_t crc_shift(const uint32_t *X, uint32_t r, uint32_t n)
{
    for (int k = 0; n; ++k, n >>= 1)
        if (n & 1u)
            r = gf2_mul(X[k], r);
    return r;
}

constexpr uint32_t CRC_PARTS = 4;

// part j of n bytes: [a, b); the cuts at multiples of 16 bytes
KMM_HD inline void crc_part_range(uint32_t n, uint32_t j, uint32_t &a, uint32_t &b)
{
    const uint32_t q = n / CRC_PARTS & ~15u;
    a = j * q;
    b = j + 1u < CRC_PARTS ? a + q : n;
}

// the register after p[0, n) from the start value c (slicing by 8; T as in crc32_sliced)
KMM_HD inline uint32_t crc_register(const uint32_t *T, const uint8_t *p, uint32_t n, uint32_t c)
{
    uint32_t i = 0;
    auto step8 = [&](uint64_t w) {
        const uint32_t lo = (uint32_t)w ^ c, hi = (uint32_t)(w >> 32);
        c = T[7 * 256 + (lo & 0xFFu)] ^ T[6 * 256 + ((lo >> 8) & 0xFFu)] ^ T[5 * 256 + ((lo >> 16) & 0xFFu)] ^ T[4 * 256 + (lo >> 24)] ^
            T[3 * 256 + (hi & 0xFFu)] ^ T[2 * 256 + ((hi >> 8) & 0xFFu)] ^ T[1 * 256 + ((hi >> 16) & 0xFFu)] ^ T[0 * 256 + (hi >> 24)];
    };
    for (; i + 16u <= n; i += 16u) { // (one request per 16 bytes of the output)
        uint64_t w0, w1;
        load16u(p + i, w0, w1);
        step8(w0);
        step8(w1);
    }
    for (; i < n; ++i)
        c = (c >> 8) ^ T[(c ^ p[i]) & 0xFFu];
    return c;
}

// crc over p[0, n) with the 8 x 256 tables T (flat: T[k * 256 + b])
KMM_HD inline uint32_t crc32_sliced(const uint32_t *T, const uint8_t *p, uint32_t n)
{
    return ~crc_register(T, p, n, 0xFFFFFFFFu);
}

// part j's share of the whole message's register (the exclusive-or over the parts, complemented, is the CRC)
KMM_HD inline uint32_t crc_part_share(const uint32_t *T, const uint32_t *X, const uint8_t *p, uint32_t n, uint32_t j)
{
    uint32_t a, b;
    crc_part_range(n, j, a, b);
    return crc_shift(X, crc_register(T, p + a, b - a, j ? 0u : 0xFFFFFFFFu), n - b);
}

// One BGZF member at m (its total size msize from the header) -> out[0, n_out), n_out = the trailer's ISIZE as the caller
// planned it.  crcT: the sliced CRC tables.
KMM_HD inline int inflate_bgzf_member(const uint8_t *m, uint32_t msize, uint8_t *out, uint32_t n_out, uint16_t *prim, uint16_t *sec,
                                      uint64_t *list, const uint32_t *crcT, unsigned long long *tm = nullptr)
{
    if (bgzf_member_size(m, msize) != msize)
        return E_HEADER;
    const uint32_t xlen = rd16(m + 10);
    const uint8_t *payload = m + 12 + xlen;
    const uint32_t plen = msize - 12u - xlen - 8u;
    if (rd32(m + msize - 4) != n_out)
        return E_HEADER;
    const int rc = inflate_stream(payload, plen, out, n_out, prim, sec, list, tm);
    if (rc != OK)
        return rc;
    if (!crcT) // (the GPU checks the CRC in a kernel of its own, k_crc_bgzf)
        return OK;
    KMM_GZ_T0;
    const bool same = crc32_sliced(crcT, out, n_out) == rd32(m + msize - 8);
    KMM_GZ_T(3);
    return same ? OK : E_CRC;
}

#if defined(__HIPCC__)
// One thread per member (see the head of the file).  m_off[i] / o_off[i]: where member i starts in comp / its bytes in out
// (n_members + 1 entries each); tabs: SCRATCH_BYTES of scratch per thread of the grid (subtables, match list); crcT: the 8 x 256 CRC tables
// (made once per handle by the host); err: [0] members in error, [1] the first of them (atomic minimum), [2] its error code.
__global__ void __launch_bounds__(64) k_inflate_bgzf(const uint8_t *__restrict__ comp, const unsigned long long *__restrict__ m_off,
                                                     const unsigned long long *__restrict__ o_off, uint8_t *__restrict__ out,
                                                     uint32_t n_members, uint8_t *__restrict__ tabs, const uint32_t *__restrict__ crcT,
                                                     unsigned int *__restrict__ err, unsigned long long *__restrict__ timers,
                                                     uint8_t *__restrict__ status)
{
    __shared__ uint16_t s_prim[64 * PRIM_WORDS]; // 40 KB: four wavefronts per CU
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x, stride = gridDim.x * 64u;
    uint16_t *prim = s_prim + threadIdx.x;       // entry x of this lane: prim[x * PS]
    unsigned long long *tm = timers ? timers + (size_t)slot * 8 : nullptr;
    uint64_t *list = reinterpret_cast<uint64_t *>(tabs + (size_t)slot * SCRATCH_BYTES);
    uint16_t *sec = reinterpret_cast<uint16_t *>(list + LIST_CAP);
#if defined(KMM_GZ_EXPERIMENT_SEC_IN_LDS) || defined(KMM_GZ_EXPERIMENT_PAD_LDS)
    __shared__ uint16_t s_sec[64 * SEC_WORDS]; // 64 KB more: one wavefront per CU in both experiments
#if defined(KMM_GZ_EXPERIMENT_SEC_IN_LDS)
    sec = s_sec + threadIdx.x * SEC_WORDS;
#else
    s_sec[threadIdx.x] = 0;
    if (n_members == 0xFFFFFFFFu)
        err[3] = s_sec[threadIdx.x ^ 1];
#endif
#endif
    for (uint32_t m = slot; m < n_members; m += stride) {
        const unsigned long long a = m_off[m], b = m_off[m + 1], oa = o_off[m], ob = o_off[m + 1];
        const int rc = inflate_bgzf_member(comp + a, (uint32_t)(b - a), out + oa, (uint32_t)(ob - oa), prim, sec, list, crcT, tm);
        if (rc != OK) {
            if (status)
                status[m] = (uint8_t)rc;
            atomicAdd(&err[0], 1u);
            if (atomicMin(&err[1], m) > m)
                err[2] = (unsigned int)rc; // (the code of the lowest member seen so far; a later, lower member overwrites it)
        }
    }
}

// The CRC32 of every member's output against its trailer, a kernel of its own behind k_inflate_bgzf (crcT = nullptr there):
// inside the inflater the check ran at that kernel's four wavefronts per CU — 4 096 dependent rounds of table look-ups per lane,
// 2-3 ms of its 24 — here nothing holds the occupancy down and the slicing tables (8 KB) sit in LDS.  CRC_PARTS neighbouring
// threads per member: each sums its part of the bytes and moves its register forward by the bytes behind it (crc_part_share),
// the exclusive-or over the four lanes is the member's register — a quarter of the dependent rounds per lane.
// status[m] != 0: the inflater has refused the member already.
__global__ void __launch_bounds__(256) k_crc_bgzf(const uint8_t *__restrict__ comp, const unsigned long long *__restrict__ m_off,
                                                  const unsigned long long *__restrict__ o_off, const uint8_t *__restrict__ out,
                                                  uint32_t n_members, const uint32_t *__restrict__ crcT, unsigned int *__restrict__ err,
                                                  const uint8_t *__restrict__ status)
{
    static_assert(CRC_PARTS == 4, "the lanes of a member are combined by two butterfly steps");
    __shared__ uint32_t T[CRC_TABLE_WORDS];
    for (uint32_t i = threadIdx.x; i < (uint32_t)CRC_TABLE_WORDS; i += 256u)
        T[i] = crcT[i];
    __syncthreads();
    const uint32_t m = blockIdx.x * (256u / CRC_PARTS) + threadIdx.x / CRC_PARTS, j = threadIdx.x % CRC_PARTS;
    const bool live = m < n_members && !(status && status[m]); // (the same for the four lanes of a member)
    uint32_t c = 0;
    if (live)
        c = crc_part_share(T, T + 8 * 256, out + o_off[m], (uint32_t)(o_off[m + 1] - o_off[m]), j);
    c ^= (uint32_t)__shfl_xor((int)c, 1);
    c ^= (uint32_t)__shfl_xor((int)c, 2);
    if (live && j == 0 && ~c != rd32(comp + m_off[m + 1] - 8)) {
        atomicAdd(&err[0], 1u);
        if (atomicMin(&err[1], m) > m)
            err[2] = (unsigned int)E_CRC;
    }
}
#endif

} // namespace kmm_gz
