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

constexpr int LIT_PB = 10, DIST_PB = 8;            // primary table bits
constexpr int LIT_CAP = 2048, DIST_CAP = 1024;     // table entries incl. subtables (checked while the tables are built)
constexpr int TAB_WORDS = LIT_CAP + DIST_CAP;      // uint32 entries of scratch per member
constexpr uint32_t LINK = 0x80000000u;             // entry: bit 31 = link to a subtable (bits 8..23 offset, 0..4 index bits);
                                                   // else bits 0..4 = code length (0: no code), bits 8..23 = symbol

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
#endif
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

KMM_HD inline uint32_t rd16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
KMM_HD inline uint32_t rd32(const uint8_t *p) { return rd16(p) | (rd16(p + 2) << 16); }

// Decoding table of a canonical Huffman code (RFC 1951 3.2.2) from its code lengths, into t[0 .. cap).  Three sweeps over
// the symbols, no temporary arrays: the primary entries of the prefixes that need a subtable first hold the longest code
// length under them, then the link.  Returns OK or E_TABLE.  allow_incomplete: a distance code may consist of one code.
KMM_HD inline int build_table(const uint8_t *lens, int n, int pb, uint32_t *t, int cap, bool allow_incomplete)
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
        t[i] = 0u;
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
            if ((uint32_t)l > t[r & pmask])
                t[r & pmask] = (uint32_t)l;
        }
    }
    uint32_t total = psize;
    if (any_long) {
        for (uint32_t i = 0; i < psize; ++i)
            if (t[i]) {
                const uint32_t sb = t[i] - (uint32_t)pb;
                if (total + (1u << sb) > (uint32_t)cap)
                    return E_TABLE;
                t[i] = LINK | (total << 8) | sb;
                total += 1u << sb;
            }
        for (uint32_t i = psize; i < total; ++i)
            t[i] = 0u;
    }
    for (int l = 0; l < 16; ++l)
        nx[l] = first[l];
    for (int s = 0; s < n; ++s) { // sweep 2: the entries
        const int l = lens[s] & 15;
        if (!l)
            continue;
        const uint32_t r = rev_bits(nx[l]++, l);
        if (l <= pb) {
            const uint32_t e = ((uint32_t)s << 8) | (uint32_t)l;
            for (uint32_t i = r; i < psize; i += 1u << l)
                t[i] = e;
        } else {
            const uint32_t link = t[r & pmask], off = (link >> 8) & 0xFFFFu, sb = link & 31u;
            const uint32_t e = ((uint32_t)s << 8) | (uint32_t)(l - pb);
            for (uint32_t i = r >> pb; i < (1u << sb); i += 1u << (l - pb))
                t[off + i] = e;
        }
    }
    return OK;
}

// bit reader over in[0, n): 64-bit buffer, refilled four bytes at a time; the next four bytes are requested one refill
// ahead (`ahead`), so that the load's latency lies behind the symbols decoded in between
struct Bits {
    const uint8_t *in;
    uint32_t n, pos; // next byte not yet requested
    uint64_t buf;
    int cnt;         // valid bits in buf
    uint32_t ahead;  // bytes [pos - 4, pos) when ahead_ok
    bool ahead_ok;
};

KMM_HD inline uint32_t load4(const uint8_t *in, uint32_t n, uint32_t pos)
{
    if (pos + 4u <= n) {
        uint32_t w;
        memcpy(&w, in + pos, 4); // (unaligned 4-byte load)
        return w;
    }
    uint32_t w = 0;
    for (uint32_t j = 0; j < 4u; ++j)
        if (pos + j < n)
            w |= (uint32_t)in[pos + j] << (8 * j);
    return w; // (bytes behind the end read as zero; `consumed` is checked against n at the end of every block)
}

KMM_HD inline void bits_init(Bits &b, const uint8_t *in, uint32_t n)
{
    b.in = in;
    b.n = n;
    b.buf = (uint64_t)load4(in, n, 0) | ((uint64_t)load4(in, n, 4) << 32);
    b.cnt = 64;
    b.pos = 12;
    b.ahead = load4(in, n, 8);
    b.ahead_ok = true;
}

KMM_HD inline void bits_refill(Bits &b)
{
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)b.ahead << b.cnt;
        b.cnt += 32;
        b.ahead = load4(b.in, b.n, b.pos);
        b.pos += 4;
    }
}

KMM_HD inline uint32_t bits_take(Bits &b, int k) // k <= 16 bits, after a refill
{
    const uint32_t v = (uint32_t)b.buf & ((1u << k) - 1u);
    b.buf >>= k;
    b.cnt -= k;
    return v;
}

// bytes of the input the decoder has really used (the buffer and the look-ahead hold bytes it has not)
KMM_HD inline uint32_t bits_consumed_bytes(const Bits &b)
{
    const uint32_t held_bits = (uint32_t)b.cnt + 32u; // buffer + look-ahead word
    return b.pos - held_bits / 8u;                   // (rounded towards the byte that holds the next bit)
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
// out[n_out)).  tab: TAB_WORDS uint32 of scratch.  Returns OK or an Err.
//
// The decoder is a STATE MACHINE that does at most one dependent memory access per turn of its loop: look up a symbol
// (S_SYM, S_SUB: the link's subtable), look up a distance (S_DIST, S_DSUB), copy one 16-byte piece of a match (S_COPY;
// S_PAT / S_FILL for distances below 16, where the piece is built once from the repeating pattern and then stored without
// further loads), read a block header and build its tables (S_HDR).  On the GPU every lane of a wavefront decodes a member of
// its own and the lanes are in different states: with the loads of all states issued side by side at the top of the turn a
// turn costs ONE memory round trip whatever the mix, and a lane with a 258-byte match holds the others up for one piece
// at a time, not for the whole copy (the first version looped over the match inside the symbol loop: every turn cost the
// longest copy among the 64 lanes, 14 GB/s of FASTQ per GPU instead of ... — profiles/r05/bgzf_e2e*.txt).
enum State { S_HDR = 0, S_SYM, S_SUB, S_DIST, S_DSUB, S_COPY, S_PAT, S_FILL, S_DONE };

KMM_HD inline int inflate_stream(const uint8_t *in, uint32_t n_in, uint8_t *out, uint32_t n_out, uint32_t *tab)
{
    // RFC 1951 3.2.5: length codes 257..285, distance codes 0..29
    const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint32_t *lit = tab, *dst = tab + LIT_CAP;
    Bits b;
    bits_init(b, in, n_in);
    uint32_t o = 0, final = 0, len = 0, dist = 0, rem = 0, sub = 0, pc = 0;
    uint64_t p0 = 0, p1 = 0; // the repeating pattern of a match with a distance below 16
    int state = S_HDR, rc = OK;
    while (state != S_DONE) {
        // ---- what this turn reads
        const uint32_t *ta = nullptr;
        const uint8_t *da = nullptr;
        if (state == S_SYM) {
            bits_refill(b);
            ta = lit + ((uint32_t)b.buf & ((1u << LIT_PB) - 1u));
        } else if (state == S_DIST) {
            bits_refill(b);
            ta = dst + ((uint32_t)b.buf & ((1u << DIST_PB) - 1u));
        } else if (state == S_SUB) {
            ta = lit + sub;
        } else if (state == S_DSUB) {
            ta = dst + sub;
        } else if (state == S_COPY || state == S_PAT) {
            da = out + o - dist;
        }
        const uint32_t e = ta ? *ta : 0u;
        uint64_t d0 = 0, d1 = 0;
        if (da) {
            memcpy(&d0, da, 8);
            memcpy(&d1, da + 8, 8);
        }
        // ---- what it does with it
        switch (state) {
        case S_HDR: {
            bits_refill(b);
            final = bits_take(b, 1);
            const uint32_t type = bits_take(b, 2);
            if (type == 3u) {
                rc = E_BTYPE;
                break;
            }
            if (type == 0u) { // stored: to the next byte boundary, LEN, NLEN, LEN bytes
                bits_take(b, b.cnt & 7);
                bits_refill(b);
                const uint32_t slen = bits_take(b, 16);
                bits_refill(b);
                const uint32_t nlen = bits_take(b, 16);
                if ((slen ^ nlen) != 0xFFFFu) {
                    rc = E_STORED;
                    break;
                }
                uint32_t at = bits_consumed_bytes(b); // (on a byte boundary: the bytes follow in the input as they are)
                if (at + slen > n_in) {
                    rc = E_STORED;
                    break;
                }
                if (o + slen > n_out) {
                    rc = E_OUTPUT;
                    break;
                }
                for (uint32_t j = 0; j < slen; ++j)
                    out[o + j] = in[at + j];
                o += slen;
                at += slen;
                b.buf = (uint64_t)load4(in, n_in, at) | ((uint64_t)load4(in, n_in, at + 4) << 32); // the bit reader behind the block
                b.cnt = 64;
                b.ahead = load4(in, n_in, at + 8);
                b.pos = at + 12;
                if (bits_consumed_bytes(b) > n_in)
                    rc = E_INPUT;
                state = final ? S_DONE : S_HDR;
                break;
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
                if (hlit > 286 || hdist > 30) {
                    rc = E_CODELEN;
                    break;
                }
                uint8_t cl[19];
                for (int i = 0; i < 19; ++i)
                    cl[i] = 0;
                for (int i = 0; i < hclen; ++i) {
                    bits_refill(b);
                    cl[cl_order[i]] = (uint8_t)bits_take(b, 3);
                }
                // the code-length code: 7-bit table in the distance table's scratch (rebuilt below)
                if (build_table(cl, 19, 7, dst, DIST_CAP, false) != OK) {
                    rc = E_CODELEN;
                    break;
                }
                int i = 0;
                while (i < hlit + hdist && rc == OK) {
                    bits_refill(b);
                    const uint32_t ce = dst[(uint32_t)b.buf & 127u];
                    const int cl_len = (int)(ce & 31u);
                    if (!cl_len) {
                        rc = E_CODELEN;
                        break;
                    }
                    bits_take(b, cl_len);
                    const uint32_t sym = (ce >> 8) & 0xFFFFu;
                    if (sym < 16u) {
                        lens[i++] = (uint8_t)sym;
                    } else {
                        int rep;
                        uint8_t v = 0;
                        if (sym == 16u) {
                            if (i == 0) {
                                rc = E_CODELEN;
                                break;
                            }
                            v = lens[i - 1];
                            rep = 3 + (int)bits_take(b, 2);
                        } else if (sym == 17u) {
                            rep = 3 + (int)bits_take(b, 3);
                        } else {
                            rep = 11 + (int)bits_take(b, 7);
                        }
                        if (i + rep > hlit + hdist) {
                            rc = E_CODELEN;
                            break;
                        }
                        for (int j = 0; j < rep; ++j)
                            lens[i++] = v;
                    }
                }
                if (rc != OK)
                    break;
                if (lens[256] == 0) {
                    rc = E_TABLE; // no end-of-block code
                    break;
                }
            }
            if (build_table(lens, hlit, LIT_PB, lit, LIT_CAP, false) != OK || build_table(lens + hlit, hdist, DIST_PB, dst, DIST_CAP, true) != OK) {
                rc = E_TABLE;
                break;
            }
            state = S_SYM;
            break;
        }
        case S_SYM:
        case S_SUB: {
            if (state == S_SYM && (e & LINK)) { // a code longer than the primary table's index: its subtable next turn
                sub = ((e >> 8) & 0xFFFFu) + (((uint32_t)b.buf >> LIT_PB) & ((1u << (e & 31u)) - 1u));
                state = S_SUB;
                break;
            }
            const int l = (int)(e & 31u);
            if (!l) {
                rc = E_SYMBOL;
                break;
            }
            if (state == S_SUB)
                bits_take(b, LIT_PB);
            bits_take(b, l);
            const uint32_t sym = (e >> 8) & 0xFFFFu;
            state = S_SYM;
            if (sym < 256u) {
                if (o >= n_out)
                    rc = E_OUTPUT;
                else
                    out[o++] = (uint8_t)sym;
            } else if (sym == 256u) {
                if (bits_consumed_bytes(b) > n_in)
                    rc = E_INPUT;
                state = final ? S_DONE : S_HDR;
            } else if (sym > 285u) {
                rc = E_SYMBOL;
            } else {
                len = (uint32_t)len_base[sym - 257u] + bits_take(b, len_extra[sym - 257u]);
                state = S_DIST;
            }
            break;
        }
        case S_DIST:
        case S_DSUB: {
            if (state == S_DIST && (e & LINK)) {
                sub = ((e >> 8) & 0xFFFFu) + (((uint32_t)b.buf >> DIST_PB) & ((1u << (e & 31u)) - 1u));
                state = S_DSUB;
                break;
            }
            const int dl = (int)(e & 31u);
            if (!dl) {
                rc = E_SYMBOL;
                break;
            }
            if (state == S_DSUB)
                bits_take(b, DIST_PB);
            bits_take(b, dl);
            const uint32_t dsym = (e >> 8) & 0xFFFFu;
            if (dsym > 29u) {
                rc = E_SYMBOL;
                break;
            }
            const int de = dist_extra[dsym];
            dist = dist_base[dsym];
            if (de)
                dist += bits_take(b, de); // (a refill leaves >= 33 bits: 15 + 13 fit)
            if (dist > o) {
                rc = E_DISTANCE;
                break;
            }
            if (o + len > n_out) {
                rc = E_OUTPUT;
                break;
            }
            rem = len;
            state = dist >= 16u ? S_COPY : S_PAT;
            break;
        }
        case S_COPY: { // 16 bytes that lie wholly behind the write position
            const uint32_t n = rem < 16u ? rem : 16u;
            store_upto16(out + o, d0, d1, n);
            o += n;
            rem -= n;
            if (!rem)
                state = S_SYM;
            break;
        }
        case S_PAT: { // distance 1 .. 15: the last `dist` bytes repeat; pc = the largest multiple of dist within 16
            uint64_t q0 = 0, q1 = 0;
            uint32_t k2 = 0;
            for (uint32_t j = 0; j < 16u; ++j) {
                const uint64_t byte = k2 < 8u ? (d0 >> (8u * k2)) & 0xFFull : (d1 >> (8u * (k2 - 8u))) & 0xFFull;
                if (j < 8u)
                    q0 |= byte << (8u * j);
                else
                    q1 |= byte << (8u * (j - 8u));
                if (++k2 == dist)
                    k2 = 0;
            }
            p0 = q0;
            p1 = q1;
            pc = 16u / dist * dist;
            state = S_FILL;
        }
            [[fallthrough]]; // the first piece right away
        case S_FILL: {
            const uint32_t n = rem < pc ? rem : pc;
            store_upto16(out + o, p0, p1, n);
            o += n;
            rem -= n;
            if (!rem)
                state = S_SYM;
            break;
        }
        default:
            rc = E_SYMBOL;
            break;
        }
        if (rc != OK)
            return rc;
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
KMM_HD inline int inflate_bgzf_member(const uint8_t *m, uint32_t msize, uint8_t *out, uint32_t n_out, uint32_t *tab, const uint32_t *crcT)
{
    if (bgzf_member_size(m, msize) != msize)
        return E_HEADER;
    const uint32_t xlen = rd16(m + 10);
    const uint8_t *payload = m + 12 + xlen;
    const uint32_t plen = msize - 12u - xlen - 8u;
    if (rd32(m + msize - 4) != n_out)
        return E_HEADER;
    const int rc = inflate_stream(payload, plen, out, n_out, tab);
    if (rc != OK)
        return rc;
    return crc32_sliced(crcT, out, n_out) == rd32(m + msize - 8) ? OK : E_CRC;
}

#if defined(__HIPCC__)
// One thread per member (see the head of the file).  m_off[i] / o_off[i]: where member i starts in comp / its bytes in out
// (n_members + 1 entries each); tabs: TAB_WORDS words of scratch per thread of the grid; err: [0] members in error,
// [1] the first of them (atomic minimum), [2 + ...] unused, [2] its error code.
__global__ void __launch_bounds__(64) k_inflate_bgzf(const uint8_t *__restrict__ comp, const unsigned long long *__restrict__ m_off,
                                                     const unsigned long long *__restrict__ o_off, uint8_t *__restrict__ out,
                                                     uint32_t n_members, uint32_t *__restrict__ tabs, unsigned int *__restrict__ err)
{
    __shared__ uint32_t crcT[8 * 256];
    for (uint32_t i = threadIdx.x; i < 8u * 256u; i += 64u)
        crcT[i] = crc_table_entry((int)(i >> 8), i & 255u);
    __syncthreads();
    const uint32_t slot = blockIdx.x * 64u + threadIdx.x, stride = gridDim.x * 64u;
    uint32_t *tab = tabs + (size_t)slot * TAB_WORDS;
    for (uint32_t m = slot; m < n_members; m += stride) {
        const unsigned long long a = m_off[m], b = m_off[m + 1], oa = o_off[m], ob = o_off[m + 1];
        const int rc = inflate_bgzf_member(comp + a, (uint32_t)(b - a), out + oa, (uint32_t)(ob - oa), tab, crcT);
        if (rc != OK) {
            atomicAdd(&err[0], 1u);
            if (atomicMin(&err[1], m) > m)
                err[2] = (unsigned int)rc; // (the code of the lowest member seen so far; a later, lower member overwrites it)
        }
    }
}
#endif

} // namespace kmm_gz
