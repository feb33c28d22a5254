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
// on the one before), lanes cannot share one; but a batch holds tens of thousands of members (3 GB of FASTQ = 47 000),
// so every lane of every wavefront gets a stream of its own and the chip hides each lane's memory latency behind the
// other lanes' — the same latency-bound-per-lane, throughput-by-parallelism regime as the direct probe kernel.  A lane
// keeps its two Huffman tables (literal/length: 10-bit primary + subtables; distance: 8-bit primary + subtables) in a
// 12 KB scratch area of its own in HBM (L2 / MALL resident while in use), reads its input through a 64-bit bit buffer
// refilled by 4-byte loads that are requested one refill ahead, writes literals as they come and copies matches in
// 8-byte pieces where the distance allows.  Stored, fixed and dynamic blocks (RFC 1951); every access is bounds-checked
// against the member's ISIZE / compressed size, so a damaged member ends in an error code, never in a stray access; the
// CRC32 of the output is checked on the device too (slicing-by-8, the tables in LDS).
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
// LDS on the GPU (576 entries = 1152 bytes per lane, 72 KB per wavefront: two wavefronts per CU), codes longer than that go
// through a link to a SECONDARY table in the lane's scratch area in HBM (rare: the frequent symbols have the short codes).
//   direct entry   bits 0..3 code length (subtables: length beyond the primary bits; 0 = no code), bits 4..12 symbol
//   link           bit 15, bits 4..14 offset of the subtable in the secondary table, bits 0..3 its index bits
constexpr int LIT_PB = 9, DIST_PB = 6;
constexpr int PRIM_LIT = 1 << LIT_PB, PRIM_DIST = 1 << DIST_PB, PRIM_WORDS = PRIM_LIT + PRIM_DIST; // uint16 per lane (LDS)
constexpr int SEC_LIT = 1024, SEC_DIST = 1024, SEC_WORDS = SEC_LIT + SEC_DIST;                     // uint16 per lane (HBM)
constexpr uint32_t LINK = 0x8000u;

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

// Decoding tables of a canonical Huffman code (RFC 1951 3.2.2) from its code lengths: prim[0 .. 2^pb), subtables in
// sec[0 .. sec_cap).  Two sweeps over the symbols, no temporary arrays: the primary entries of the prefixes that need a
// subtable first hold the longest code length under them, then the link.  Returns OK or E_TABLE.  allow_incomplete: a
// distance code may consist of one code.
KMM_HD inline int build_table(const uint8_t *lens, int n, int pb, uint16_t *prim, uint16_t *sec, int sec_cap, bool allow_incomplete)
{
    int count[16];
    for (int i = 0; i < 16; ++i)
        count[i] = 0;
    for (int i = 0; i < n; ++i)
        count[lens[i] & 15]++;
    if (count[0] == n)
        return E_TABLE;
    int left = 1; // Kraft: code space still free
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0)
            return E_TABLE;
    }
    if (left > 0 && !allow_incomplete)
        return E_TABLE;
    uint32_t first[16];
    {
        uint32_t code = 0;
        first[0] = 0;
        for (int l = 1; l <= 15; ++l) {
            code = (code + (l > 1 ? (uint32_t)count[l - 1] : 0u)) << 1;
            first[l] = code;
        }
    }
    const uint32_t psize = 1u << pb, pmask = psize - 1u;
    for (uint32_t i = 0; i < psize; ++i)
        prim[i] = 0;
    uint32_t nx[16];
    bool any_long = false;
    for (int l = 0; l < 16; ++l)
        nx[l] = first[l];
    for (int s = 0; s < n; ++s) { // sweep 1: longest code under every primary index
        const int l = lens[s] & 15;
        if (!l)
            continue;
        const uint32_t r = rev_bits(nx[l]++, l);
        if (l > pb) {
            any_long = true;
            if ((uint32_t)l > prim[r & pmask])
                prim[r & pmask] = (uint16_t)l;
        }
    }
    if (any_long) {
        uint32_t total = 0;
        for (uint32_t i = 0; i < psize; ++i)
            if (prim[i]) {
                const uint32_t sb = (uint32_t)prim[i] - (uint32_t)pb;
                if (!sec || total + (1u << sb) > (uint32_t)sec_cap || total >= 2048u)
                    return E_TABLE;
                prim[i] = (uint16_t)(LINK | (total << 4) | sb);
                total += 1u << sb;
            }
        for (uint32_t i = 0; i < total; ++i)
            sec[i] = 0;
    }
    for (int l = 0; l < 16; ++l)
        nx[l] = first[l];
    for (int s = 0; s < n; ++s) { // sweep 2: the entries
        const int l = lens[s] & 15;
        if (!l)
            continue;
        const uint32_t r = rev_bits(nx[l]++, l);
        if (l <= pb) {
            const uint16_t e = (uint16_t)(((uint32_t)s << 4) | (uint32_t)l);
            for (uint32_t i = r; i < psize; i += 1u << l)
                prim[i] = e;
        } else {
            const uint32_t link = prim[r & pmask], off = (link >> 4) & 0x7FFu, sb = link & 15u;
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
    uint64_t nlo, nhi;   // the block behind it (bytes [pos - 16, pos))
};

KMM_HD inline void load16(const uint8_t *in, uint32_t n, uint32_t pos, uint64_t &lo, uint64_t &hi)
{
    if (pos + 16u <= n) {
        memcpy(&lo, in + pos, 8); // (unaligned loads)
        memcpy(&hi, in + pos + 8, 8);
        return;
    }
    lo = 0;
    hi = 0;
    for (uint32_t j = 0; j < 16u; ++j)
        if (pos + j < n) {
            if (j < 8u)
                lo |= (uint64_t)in[pos + j] << (8u * j);
            else
                hi |= (uint64_t)in[pos + j] << (8u * (j - 8u));
        }
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
    load16(b.in, b.n, at + 16u, b.nlo, b.nhi);
    b.pos = at + 32u;
}

KMM_HD inline void bits_refill(Bits &b)
{
    if (b.cnt <= 32) {
        if (b.fw == 0) {
            b.flo = b.nlo;
            b.fhi = b.nhi;
            b.fw = 4;
            load16(b.in, b.n, b.pos, b.nlo, b.nhi);
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

// bytes of the input the decoder has really used (the buffer, the open block and the block behind it hold bytes it has not)
KMM_HD inline uint32_t bits_consumed_bytes(const Bits &b)
{
    return b.pos - 16u - 4u * (uint32_t)b.fw - ((uint32_t)b.cnt >> 3);
}

// n <= 16 bytes of the 16 in (lo, hi) -> q
KMM_HD inline void store_upto16(uint8_t *q, uint64_t lo, uint64_t hi, uint32_t n)
{
    if (n >= 8u) {
        memcpy(q, &lo, 8);
        if (n == 16u) {
            memcpy(q + 8, &hi, 8);
        } else {
            for (uint32_t j = 8; j < n; ++j)
                q[j] = (uint8_t)(hi >> (8u * (j - 8u)));
        }
    } else {
        for (uint32_t j = 0; j < n; ++j)
            q[j] = (uint8_t)(lo >> (8u * j));
    }
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
// History (profiles/r05/bgzf_e2e_*.txt, bgzf_v*_kernel_stats.csv; 3.26 GB of FASTQ in 47 000 members, kernel time for 23 000
// members = 1.5 GB): v1, a plain symbol loop, the match copy inside it, all tables in HBM — 2-3 dependent trips per
// symbol: ~75 ms; v2, the same as a one-access-per-turn state machine: 74 ms (as many trips, just tidier); v3, primary
// tables in LDS and up to six literals per turn, one trip per match: 48 ms; v4, this one.
constexpr int DECODE_RUN = 1024;    // symbols per phase A
constexpr int LIST_CAP = 512;       // matches per phase A (8 bytes each)
enum State { S_HDR = 0, S_SYM, S_DONE };
constexpr int SCRATCH_BYTES = LIST_CAP * 8 + SEC_WORDS * 2; // per lane, in HBM: the match list, then the subtables

// the bytes of one match: len bytes from `dist` behind q (q = out + position); every load before the first dependent store
KMM_HD inline void copy_match(uint8_t *q, uint32_t len, uint32_t dist)
{
    const uint8_t *s = q - dist;
    if (dist >= len || dist >= 16u) {
        // 16-byte pieces; with dist >= len none of them reads what this match writes: two pieces per trip; with
        // 16 <= dist < len a piece may read the piece before it: one piece per trip (program order does the rest)
        const bool indep = dist >= len;
        while (len) {
            uint64_t a0, a1, b0 = 0, b1 = 0;
            memcpy(&a0, s, 8);
            memcpy(&a1, s + 8, 8);
            const bool two = indep && len > 16u;
            if (two) {
                memcpy(&b0, s + 16, 8);
                memcpy(&b1, s + 24, 8);
            }
            uint32_t n = len < 16u ? len : 16u;
            store_upto16(q, a0, a1, n);
            q += n;
            s += n;
            len -= n;
            if (two) {
                n = len < 16u ? len : 16u;
                store_upto16(q, b0, b1, n);
                q += n;
                s += n;
                len -= n;
            }
        }
        return;
    }
    // distance 1 .. 15 < len: the last `dist` bytes repeat; the pattern is built once, pc = the largest multiple of dist within 16
    uint64_t d0, d1, p0 = 0, p1 = 0;
    memcpy(&d0, s, 8);
    memcpy(&d1, s + 8, 8);
    uint32_t k2 = 0;
    for (uint32_t j = 0; j < 16u; ++j) {
        const uint64_t byte = k2 < 8u ? (d0 >> (8u * k2)) & 0xFFull : (d1 >> (8u * (k2 - 8u))) & 0xFFull;
        if (j < 8u)
            p0 |= byte << (8u * j);
        else
            p1 |= byte << (8u * (j - 8u));
        if (++k2 == dist)
            k2 = 0;
    }
    const uint32_t pc = 16u / dist * dist;
    while (len) {
        const uint32_t n = len < pc ? len : pc;
        store_upto16(q, p0, p1, n);
        q += n;
        len -= n;
    }
}

KMM_HD inline int inflate_stream(const uint8_t *in, uint32_t n_in, uint8_t *out, uint32_t n_out, uint16_t *prim, uint16_t *sec,
                                 uint64_t *list)
{
    // RFC 1951 3.2.5: length codes 257..285, distance codes 0..29
    const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint16_t *lit = prim, *dst = prim + PRIM_LIT, *lit2 = sec, *dst2 = sec ? sec + SEC_LIT : nullptr;
    Bits b;
    b.in = in;
    b.n = n_in;
    bits_start(b, 0);
    uint32_t o = 0, final = 0;
    int state = S_HDR;
    while (state != S_DONE) {
        // ---- phase A: decode; literals are stored, matches written down
        uint32_t n_list = 0;
        for (int t = 0; t < DECODE_RUN && state != S_DONE && n_list < (uint32_t)LIST_CAP; ++t) {
            if (state == S_HDR) {
                bits_refill(b);
                if ((((uint32_t)b.buf >> 1) & 3u) == 0u && n_list)
                    break; // (a stored block's bytes are copied right here: the matches written down so far go first)
                final = bits_take(b, 1);
                const uint32_t type = bits_take(b, 2);
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
                        out[o + j] = in[at + j];
                    o += slen;
                    bits_start(b, at + slen);
                    if (bits_consumed_bytes(b) > n_in)
                        return E_INPUT;
                    state = final ? S_DONE : S_HDR;
                    continue;
                }
                uint8_t lens[320];
                int hlit = 288, hdist = 30;
                if (type == 1u) { // fixed code (RFC 1951 3.2.6)
                    for (int i = 0; i < 144; ++i) lens[i] = 8;
                    for (int i = 144; i < 256; ++i) lens[i] = 9;
                    for (int i = 256; i < 280; ++i) lens[i] = 7;
                    for (int i = 280; i < 288; ++i) lens[i] = 8;
                    for (int i = 0; i < 30; ++i) lens[288 + i] = 5;
                } else {
                    hlit = (int)bits_take(b, 5) + 257;
                    hdist = (int)bits_take(b, 5) + 1;
                    const int hclen = (int)bits_take(b, 4) + 4;
                    if (hlit > 286 || hdist > 30)
                        return E_CODELEN;
                    uint8_t cl[19];
                    for (int i = 0; i < 19; ++i)
                        cl[i] = 0;
                    for (int i = 0; i < hclen; ++i) {
                        bits_refill(b);
                        cl[cl_order[i]] = (uint8_t)bits_take(b, 3);
                    }
                    // the code-length code (codes of at most 7 bits): a 7-bit table in the literal table's place (rebuilt below)
                    if (build_table(cl, 19, 7, lit, nullptr, 0, false) != OK)
                        return E_CODELEN;
                    int i = 0;
                    while (i < hlit + hdist) {
                        bits_refill(b);
                        const uint32_t ce = lit[(uint32_t)b.buf & 127u];
                        const int cl_len = (int)(ce & 15u);
                        if (!cl_len)
                            return E_CODELEN;
                        bits_take(b, cl_len);
                        const uint32_t sym = ce >> 4;
                        if (sym < 16u) {
                            lens[i++] = (uint8_t)sym;
                        } else {
                            int rep;
                            uint8_t v = 0;
                            if (sym == 16u) {
                                if (i == 0)
                                    return E_CODELEN;
                                v = lens[i - 1];
                                rep = 3 + (int)bits_take(b, 2);
                            } else if (sym == 17u) {
                                rep = 3 + (int)bits_take(b, 3);
                            } else {
                                rep = 11 + (int)bits_take(b, 7);
                            }
                            if (i + rep > hlit + hdist)
                                return E_CODELEN;
                            for (int j = 0; j < rep; ++j)
                                lens[i++] = v;
                        }
                    }
                    if (lens[256] == 0)
                        return E_TABLE; // no end-of-block code
                }
                if (build_table(lens, hlit, LIT_PB, lit, lit2, SEC_LIT, false) != OK ||
                    build_table(lens + hlit, hdist, DIST_PB, dst, dst2, SEC_DIST, true) != OK)
                    return E_TABLE;
                state = S_SYM;
                continue;
            }
            // one symbol
            bits_refill(b);
            uint32_t e = lit[(uint32_t)b.buf & (uint32_t)(PRIM_LIT - 1)];
            if (e & LINK) { // a code longer than the primary index: its subtable (HBM)
                e = lit2[((e >> 4) & 0x7FFu) + (((uint32_t)b.buf >> LIT_PB) & ((1u << (e & 15u)) - 1u))];
                if (!(e & 15u))
                    return E_SYMBOL;
                bits_take(b, LIT_PB);
            }
            const int l = (int)(e & 15u);
            if (!l)
                return E_SYMBOL;
            bits_take(b, l);
            const uint32_t sym = e >> 4;
            if (sym < 256u) {
                if (o >= n_out)
                    return E_OUTPUT;
                out[o++] = (uint8_t)sym;
                continue;
            }
            if (sym == 256u) {
                if (bits_consumed_bytes(b) > n_in)
                    return E_INPUT;
                state = final ? S_DONE : S_HDR;
                continue;
            }
            if (sym > 285u)
                return E_SYMBOL;
            const uint32_t len = (uint32_t)len_base[sym - 257u] + bits_take(b, len_extra[sym - 257u]);
            bits_refill(b);
            uint32_t d = dst[(uint32_t)b.buf & (uint32_t)(PRIM_DIST - 1)];
            if (d & LINK) {
                d = dst2[((d >> 4) & 0x7FFu) + (((uint32_t)b.buf >> DIST_PB) & ((1u << (d & 15u)) - 1u))];
                if (!(d & 15u))
                    return E_SYMBOL;
                bits_take(b, DIST_PB);
            }
            const int dl = (int)(d & 15u);
            if (!dl)
                return E_SYMBOL;
            bits_take(b, dl);
            const uint32_t dsym = d >> 4;
            if (dsym > 29u)
                return E_SYMBOL;
            const int de = dist_extra[dsym];
            uint32_t dist = dist_base[dsym];
            if (de)
                dist += bits_take(b, de); // (a refill leaves >= 33 bits: 15 + 13 fit)
            if (dist > o)
                return E_DISTANCE;
            if (o + len > n_out)
                return E_OUTPUT;
            list[n_list++] = (uint64_t)o | ((uint64_t)len << 32) | ((uint64_t)dist << 41);
            o += len;
        }
        // ---- phase B: the matches, in order; up to four per step when none of them can depend on another.  The list lies in
        // HBM: the four entries a step looks at were requested during the step before (entries beyond the end read as an
        // entry that always goes by itself), so a step costs ONE round trip — its sources'.
        uint32_t i = 0;
        uint64_t ent[4];
        for (int x = 0; x < 4; ++x)
            ent[x] = (uint32_t)x < n_list ? list[x] : ~0ull;
        while (i < n_list) {
            const uint32_t o0 = (uint32_t)ent[0], l0 = (uint32_t)(ent[0] >> 32) & 0x1FFu, d0 = (uint32_t)(ent[0] >> 41);
            uint32_t g = 1;
            const bool simple = l0 <= 16u && d0 >= l0;
            if (simple) {
                for (; g < 4u; ++g) { // (an entry behind the end has length 511: it ends the group)
                    const uint32_t og = (uint32_t)ent[g], lg = (uint32_t)(ent[g] >> 32) & 0x1FFu, dg = (uint32_t)(ent[g] >> 41);
                    if (lg > 16u || dg < lg || og - dg + lg > o0)
                        break;
                }
            }
            // the next step's entries, and this step's sources: all loads leave before the first is used
            uint64_t nxt[4];
            for (int x = 0; x < 4; ++x)
                nxt[x] = i + g + (uint32_t)x < n_list ? list[i + g + (uint32_t)x] : ~0ull;
            if (simple) {
                uint64_t lo[4], hi[4];
                for (uint32_t x = 0; x < 4u; ++x) {
                    lo[x] = hi[x] = 0;
                    if (x < g) {
                        const uint8_t *src = out + (uint32_t)ent[x] - (uint32_t)(ent[x] >> 41);
                        memcpy(&lo[x], src, 8);
                        memcpy(&hi[x], src + 8, 8);
                    }
                }
                for (uint32_t x = 0; x < g; ++x)
                    store_upto16(out + (uint32_t)ent[x], lo[x], hi[x], (uint32_t)(ent[x] >> 32) & 0x1FFu);
            } else { // long, or repeating a pattern: by itself
                copy_match(out + o0, l0, d0);
            }
            for (int x = 0; x < 4; ++x)
                ent[x] = nxt[x];
            i += g;
        }
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

// crc over p[0, n) with the 8 x 256 tables T (flat: T[k * 256 + b])
KMM_HD inline uint32_t crc32_sliced(const uint32_t *T, const uint8_t *p, uint32_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    uint32_t i = 0;
    for (; i + 8u <= n; i += 8u) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        const uint32_t lo = (uint32_t)w ^ c, hi = (uint32_t)(w >> 32);
        c = T[7 * 256 + (lo & 0xFFu)] ^ T[6 * 256 + ((lo >> 8) & 0xFFu)] ^ T[5 * 256 + ((lo >> 16) & 0xFFu)] ^ T[4 * 256 + (lo >> 24)] ^
            T[3 * 256 + (hi & 0xFFu)] ^ T[2 * 256 + ((hi >> 8) & 0xFFu)] ^ T[1 * 256 + ((hi >> 16) & 0xFFu)] ^ T[0 * 256 + (hi >> 24)];
    }
    for (; i < n; ++i)
        c = (c >> 8) ^ T[(c ^ p[i]) & 0xFFu];
    return ~c;
}

// One BGZF member at m (its total size msize from the header) -> out[0, n_out), n_out = the trailer's ISIZE as the caller
// planned it.  crcT: the sliced CRC tables.
KMM_HD inline int inflate_bgzf_member(const uint8_t *m, uint32_t msize, uint8_t *out, uint32_t n_out, uint16_t *prim, uint16_t *sec,
                                      uint64_t *list, const uint32_t *crcT)
{
    if (bgzf_member_size(m, msize) != msize)
        return E_HEADER;
    const uint32_t xlen = rd16(m + 10);
    const uint8_t *payload = m + 12 + xlen;
    const uint32_t plen = msize - 12u - xlen - 8u;
    if (rd32(m + msize - 4) != n_out)
        return E_HEADER;
    const int rc = inflate_stream(payload, plen, out, n_out, prim, sec, list);
    if (rc != OK)
        return rc;
    return crc32_sliced(crcT, out, n_out) == rd32(m + msize - 8) ? OK : E_CRC;
}

#if defined(__HIPCC__)
// One thread per member (see the head of the file).  m_off[i] / o_off[i]: where member i starts in comp / its bytes in out
// (n_members + 1 entries each); tabs: SCRATCH_BYTES of scratch per thread of the grid (subtables, match list); crcT: the 8 x 256 CRC tables
// (made once per handle by the host); err: [0] members in error, [1] the first of them (atomic minimum), [2] its error code.
__global__ void __launch_bounds__(64) k_inflate_bgzf(const uint8_t *__restrict__ comp, const unsigned long long *__restrict__ m_off,
                                                     const unsigned long long *__restrict__ o_off, uint8_t *__restrict__ out,
                                                     uint32_t n_members, uint8_t *__restrict__ tabs, const uint32_t *__restrict__ crcT,
                                                     unsigned int *__restrict__ err)
{
    __shared__ uint16_t s_prim[64 * PRIM_WORDS]; // 72 KB: two wavefronts per CU
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x, stride = gridDim.x * 64u;
    uint16_t *prim = s_prim + threadIdx.x * PRIM_WORDS;
    uint64_t *list = reinterpret_cast<uint64_t *>(tabs + (size_t)slot * SCRATCH_BYTES);
    uint16_t *sec = reinterpret_cast<uint16_t *>(list + LIST_CAP);
    for (uint32_t m = slot; m < n_members; m += stride) {
        const unsigned long long a = m_off[m], b = m_off[m + 1], oa = o_off[m], ob = o_off[m + 1];
        const int rc = inflate_bgzf_member(comp + a, (uint32_t)(b - a), out + oa, (uint32_t)(ob - oa), prim, sec, list, crcT);
        if (rc != OK) {
            atomicAdd(&err[0], 1u);
            if (atomicMin(&err[1], m) > m)
                err[2] = (unsigned int)rc; // (the code of the lowest member seen so far; a later, lower member overwrites it)
        }
    }
}
#endif

} // namespace kmm_gz
