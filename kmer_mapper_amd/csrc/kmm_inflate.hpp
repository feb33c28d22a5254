// kmm_inflate.hpp — part of libkmm_io.so (plain C++17, host only): a deflate decoder that can start in the MIDDLE of a
// stream, for inflating ONE gzip member on many cores.
//
// Why: `gzip reads.fq` writes a single deflate stream; zlib inflates it on one core at ~0.36 GB/s, below the 16-thread
// CPU mapper of the reference and two orders of magnitude below the GPU.  The reference already reached for a faster
// inflater (igzip, kmer_mapper/util.py:78-101).  A deflate stream cannot be split at known places, but it can be
// decoded from any BLOCK boundary if the 32 KiB of history before it are treated as unknowns (the two-pass scheme of
// pugz / rapidgzip, restated here from the published idea, no code taken):
//   * find_block: the first bit position in a range where a dynamic-Huffman block header parses — HLIT / HDIST in
//     range, the code-length code a complete prefix code, the literal/length code complete with an end-of-block code,
//     the distance code complete (or a single code) — and from which the data decodes without an invalid symbol;
//   * decode with MARKERS: the output is 16-bit symbols, 0..255 = a known byte, 0x8000 + i = "byte i of the unknown
//     32 KiB window"; the output buffer starts with the 32768 markers themselves, so a back-reference into the
//     unknown history is an ordinary copy and markers propagate through later copies by themselves;
//   * once the real window is known (the end of the previous chunk, resolved first, chunk after chunk — 32 KiB each),
//     every chunk's symbols are replaced by bytes in parallel, and the CRC32s of the chunks are combined.
// The decoder is table driven (10-bit primary table + subtables for the literal/length code, 9-bit for distances),
// refills its 64-bit bit buffer with one unaligned load, and handles stored and fixed-Huffman blocks too.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace kmm_inflate {

constexpr int LIT_PB = 10, DIST_PB = 9;       // primary table bits
constexpr uint32_t WINDOW = 32768;
constexpr uint16_t MARK = 0x8000;             // symbol MARK + i = byte i of the unknown window

struct Huff {
    // entry: bit 31 = link to a subtable (bits 8..23 = its offset, bits 0..4 = its index bits);
    // else bits 0..4 = code length (0 = invalid code), bits 8..23 = symbol
    std::vector<uint32_t> t;
    int pb = 0;
};

inline uint32_t rev_bits(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

// Canonical Huffman code from code lengths (RFC 1951 3.2.2).  Returns 0 ok / complete, 1 incomplete, -1 over-subscribed
// or no code at all.  `allow_incomplete`: an incomplete code still gets a table (distance codes with one code).
inline int build_huff(const uint8_t *lens, int n, int pb, Huff &h, bool allow_incomplete)
{
    int count[16] = {0};
    for (int i = 0; i < n; ++i)
        count[lens[i]]++;
    if (count[0] == n)
        return -1;
    int left = 1; // Kraft: codes still available
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0)
            return -1;
    }
    if (left > 0 && !allow_incomplete)
        return 1;
    uint32_t next[16];
    {
        uint32_t code = 0;
        next[0] = 0;
        for (int l = 1; l <= 15; ++l) { // (unused symbols, length 0, take no code)
            code = (code + (l > 1 ? (uint32_t)count[l - 1] : 0u)) << 1;
            next[l] = code;
        }
    }
    h.pb = pb;
    const uint32_t psize = 1u << pb, pmask = psize - 1u;
    // longest code under every primary index that needs a subtable
    std::vector<uint8_t> sub_len(psize, 0);
    std::vector<uint32_t> codes((size_t)n);
    {
        uint32_t nx[16];
        memcpy(nx, next, sizeof nx);
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l)
                continue;
            const uint32_t r = rev_bits(nx[l]++, l);
            codes[(size_t)s] = r;
            if (l > pb && (uint8_t)l > sub_len[r & pmask])
                sub_len[r & pmask] = (uint8_t)l;
        }
    }
    size_t total = psize;
    std::vector<uint32_t> sub_off(psize, 0);
    for (uint32_t i = 0; i < psize; ++i)
        if (sub_len[i]) {
            sub_off[i] = (uint32_t)total;
            total += (size_t)1 << (sub_len[i] - pb);
        }
    h.t.assign(total, 0u);
    for (uint32_t i = 0; i < psize; ++i)
        if (sub_len[i])
            h.t[i] = 0x80000000u | (sub_off[i] << 8) | (uint32_t)(sub_len[i] - pb);
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l)
            continue;
        const uint32_t r = codes[(size_t)s];
        if (l <= pb) {
            const uint32_t e = ((uint32_t)s << 8) | (uint32_t)l;
            for (uint32_t i = r; i < psize; i += 1u << l)
                h.t[i] = e;
        } else {
            const uint32_t pi = r & pmask, sb = (uint32_t)(sub_len[pi] - pb);
            const uint32_t e = ((uint32_t)s << 8) | (uint32_t)(l - pb);
            for (uint32_t i = r >> pb; i < (1u << sb); i += 1u << (l - pb))
                h.t[sub_off[pi] + i] = e;
        }
    }
    return left > 0 ? 1 : 0;
}

struct Bits {
    const uint8_t *p = nullptr;
    size_t n = 0;      // bytes available
    size_t pos = 0;    // next byte to load
    uint64_t buf = 0;
    int cnt = 0;       // valid bits in buf
    bool over = false; // bits were requested beyond the end of the input

    void seek_bit(uint64_t bit)
    {
        pos = (size_t)(bit >> 3);
        buf = 0;
        cnt = 0;
        over = false;
        const int skip = (int)(bit & 7);
        if (skip) {
            refill();
            buf >>= skip;
            cnt -= skip;
        }
    }
    uint64_t bit_pos() const { return (uint64_t)pos * 8 - (uint64_t)cnt; }
    inline void refill()
    {
        if (pos + 8 <= n) {
            uint64_t v;
            memcpy(&v, p + pos, 8);
            buf |= v << cnt;
            pos += (size_t)((63 - cnt) >> 3);
            cnt |= 56;
        } else {
            while (cnt <= 56 && pos < n) {
                buf |= (uint64_t)p[pos++] << cnt;
                cnt += 8;
            }
        }
    }
    inline uint32_t peek(int nb) const { return (uint32_t)(buf & ((1ull << nb) - 1ull)); }
    inline void drop(int nb)
    {
        if (nb > cnt) {
            over = true;
            buf = 0;
            cnt = 0;
            return;
        }
        buf >>= nb;
        cnt -= nb;
    }
    inline uint32_t take(int nb)
    {
        if (cnt < nb)
            refill();
        const uint32_t v = peek(nb);
        drop(nb);
        return v;
    }
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Reads a dynamic block's code definitions (the 3 header bits already taken).  false: not a valid header.
inline bool read_dynamic_header(Bits &b, Huff &lit, Huff &dist)
{
    b.refill();
    const uint32_t hlit = b.peek(5) + 257;
    b.drop(5);
    const uint32_t hdist = b.peek(5) + 1;
    b.drop(5);
    const uint32_t hclen = b.peek(4) + 4;
    b.drop(4);
    if (hlit > 286 || hdist > 30)
        return false;
    uint8_t cl[19] = {0};
    for (uint32_t i = 0; i < hclen; ++i)
        cl[CL_ORDER[i]] = (uint8_t)b.take(3);
    Huff clh;
    if (build_huff(cl, 19, 7, clh, false) != 0)
        return false;
    uint8_t lens[286 + 30 + 138];
    uint32_t i = 0;
    const uint32_t tot = hlit + hdist;
    while (i < tot) {
        b.refill();
        const uint32_t e = clh.t[b.peek(7)];
        const int l = (int)(e & 31u);
        if (!l)
            return false;
        b.drop(l);
        const uint32_t sym = (e >> 8) & 0xFFFFu;
        if (sym < 16) {
            lens[i++] = (uint8_t)sym;
        } else {
            uint32_t rep, val = 0;
            if (sym == 16) {
                if (!i)
                    return false;
                val = lens[i - 1];
                rep = 3 + b.peek(2);
                b.drop(2);
            } else if (sym == 17) {
                rep = 3 + b.peek(3);
                b.drop(3);
            } else {
                rep = 11 + b.peek(7);
                b.drop(7);
            }
            if (i + rep > tot)
                return false;
            while (rep--)
                lens[i++] = (uint8_t)val;
        }
        if (b.over)
            return false;
    }
    if (!lens[256])
        return false; // no end-of-block code
    if (build_huff(lens, (int)hlit, LIT_PB, lit, false) != 0)
        return false;
    // distance code: complete, or incomplete with at most one code (RFC 1951 3.2.7), or none (a literal-only block)
    int n_dist = 0;
    for (uint32_t k = 0; k < hdist; ++k)
        n_dist += lens[hlit + k] != 0;
    if (n_dist == 0) {
        dist.t.assign((size_t)1 << DIST_PB, 0u);
        dist.pb = DIST_PB;
        return true;
    }
    const int rc = build_huff(lens + hlit, (int)hdist, DIST_PB, dist, n_dist == 1);
    return rc == 0 || (rc == 1 && n_dist == 1);
}

inline void fixed_tables(Huff &lit, Huff &dist)
{
    uint8_t l[288], d[30];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    for (int i = 0; i < 30; ++i) d[i] = 5;
    build_huff(l, 288, LIT_PB, lit, false);
    build_huff(d, 30, DIST_PB, dist, true);
}

// One chunk of a deflate stream, decoded into 16-bit symbols behind a prefix of WINDOW markers.
struct Chunk {
    std::vector<uint16_t> sym;   // [0, WINDOW) the markers; the chunk's output follows
    size_t n_out = 0;            // symbols of output (sym.size() may be larger: capacity)
    uint64_t start_bit = 0, end_bit = 0;
    bool final_seen = false;     // the block with BFINAL ended at end_bit
    bool failed = false;
    std::string err;

    void init(uint64_t at, size_t expect_out)
    {
        start_bit = end_bit = at;
        final_seen = failed = false;
        err.clear();
        n_out = 0;
        if (sym.size() < WINDOW + expect_out + 1024) // (kept from wave to wave: no re-allocation, no zero-fill)
            sym.resize(WINDOW + expect_out + 1024);
        for (uint32_t i = 0; i < WINDOW; ++i)
            sym[i] = (uint16_t)(MARK + i);
    }
};

// Decodes blocks from c.end_bit on until a block ends at a bit position >= stop_bit, or the final block has ended, or
// the input is exhausted / invalid (c.failed).  Resumable: call again with a later stop_bit.
// max_out: give up (failed) when a chunk's output exceeds this many symbols (a wrong start position can decode
// garbage for a long time; a real chunk expands by a bounded factor).
inline void decode_chunk(const uint8_t *data, size_t n, Chunk &c, uint64_t stop_bit, size_t max_out)
{
    Bits b;
    b.p = data;
    b.n = n;
    b.seek_bit(c.end_bit);
    Huff lit, dist;
    size_t o = WINDOW + c.n_out;
    auto fail = [&](const char *why) {
        c.failed = true;
        c.err = why;
        c.n_out = o - WINDOW;
    };
    while (!c.final_seen) {
        if (b.bit_pos() >= stop_bit)
            break;
        b.refill();
        if (b.cnt < 3)
            return fail("compressed data ended inside a deflate stream");
        const uint32_t bfinal = b.peek(1);
        b.drop(1);
        const uint32_t btype = b.peek(2);
        b.drop(2);
        if (btype == 3)
            return fail("invalid block type");
        if (btype == 0) {
            b.drop(b.cnt & 7); // to the byte boundary
            b.refill();
            if (b.cnt < 32)
                return fail("compressed data ended inside a stored block");
            const uint32_t len = b.peek(16);
            b.drop(16);
            const uint32_t nlen = b.peek(16);
            b.drop(16);
            if ((len ^ 0xFFFFu) != nlen)
                return fail("invalid stored block lengths");
            // the bytes of the block: first those already in the bit buffer, then straight from the input
            if (o + len + 512 > c.sym.size())
                c.sym.resize((o + len) * 2 + 1024);
            uint32_t left = len;
            while (left && b.cnt >= 8) {
                c.sym[o++] = (uint16_t)b.peek(8);
                b.drop(8);
                --left;
            }
            if (left) {
                if (b.pos + left > n)
                    return fail("compressed data ended inside a stored block");
                for (uint32_t i = 0; i < left; ++i)
                    c.sym[o++] = data[b.pos + i];
                b.pos += left;
                b.buf = 0; // (the bits above cnt are a preview of the bytes at the old position)
                b.cnt = 0;
            }
        } else {
            if (btype == 1) {
                fixed_tables(lit, dist);
            } else if (!read_dynamic_header(b, lit, dist) || b.over) {
                return fail("invalid code lengths set");
            }
            const uint32_t *lt = lit.t.data(), *dt = dist.t.data();
            uint16_t *out = c.sym.data();
            size_t cap = c.sym.size();
            bool eob = false;
            // Fast loop: while 8 input bytes can be loaded at once the refill leaves at least 56 valid bits, and one
            // iteration takes at most 15 + 15 (two literals) or 15 + 5 + 15 + 13 = 48 bits: no underflow checks inside.
            {
                uint64_t buf = b.buf;
                int cnt = b.cnt;
                size_t pos = b.pos;
                const char *why = nullptr;
                while (pos + 8 <= n) {
                    if (o + 300 > cap) {
                        if (o - WINDOW > max_out) {
                            why = "output larger than any real chunk's";
                            break;
                        }
                        c.sym.resize(cap * 2);
                        out = c.sym.data();
                        cap = c.sym.size();
                    }
                    {
                        uint64_t v;
                        memcpy(&v, data + pos, 8);
                        buf |= v << cnt;
                        pos += (size_t)((63 - cnt) >> 3);
                        cnt |= 56;
                    }
                    uint32_t e = lt[buf & ((1u << LIT_PB) - 1u)];
                    if (e & 0x80000000u) {
                        e = lt[((e >> 8) & 0xFFFFu) + ((uint32_t)(buf >> LIT_PB) & ((1u << (e & 31u)) - 1u))];
                        buf >>= LIT_PB;
                        cnt -= LIT_PB;
                    }
                    uint32_t l = e & 31u;
                    if (!l) {
                        why = "invalid literal/length code";
                        break;
                    }
                    buf >>= l;
                    cnt -= (int)l;
                    uint32_t sy = (e >> 8) & 0xFFFFu;
                    if (sy < 256) {
                        out[o++] = (uint16_t)sy;
                        e = lt[buf & ((1u << LIT_PB) - 1u)]; // a second literal from the same refill
                        if (!(e & 0x80000000u) && ((e >> 8) & 0xFFFFu) < 256 && (e & 31u)) {
                            buf >>= (e & 31u);
                            cnt -= (int)(e & 31u);
                            out[o++] = (uint16_t)((e >> 8) & 0xFFFFu);
                        }
                        continue;
                    }
                    if (sy == 256) {
                        eob = true;
                        break;
                    }
                    sy -= 257;
                    if (sy >= 29) {
                        why = "invalid literal/length code";
                        break;
                    }
                    const uint32_t len = LEN_BASE[sy] + (uint32_t)(buf & ((1u << LEN_EXTRA[sy]) - 1u));
                    buf >>= LEN_EXTRA[sy];
                    cnt -= LEN_EXTRA[sy];
                    e = dt[buf & ((1u << DIST_PB) - 1u)];
                    if (e & 0x80000000u) {
                        e = dt[((e >> 8) & 0xFFFFu) + ((uint32_t)(buf >> DIST_PB) & ((1u << (e & 31u)) - 1u))];
                        buf >>= DIST_PB;
                        cnt -= DIST_PB;
                    }
                    l = e & 31u;
                    const uint32_t ds = (e >> 8) & 0xFFFFu;
                    if (!l || ds >= 30) {
                        why = "invalid distance code";
                        break;
                    }
                    buf >>= l;
                    cnt -= (int)l;
                    const uint32_t dd = DIST_BASE[ds] + (uint32_t)(buf & ((1u << DIST_EXTRA[ds]) - 1u));
                    buf >>= DIST_EXTRA[ds];
                    cnt -= DIST_EXTRA[ds];
                    // (dd <= 32768 <= o: the marker prefix makes every distance valid)
                    const uint16_t *src = out + o - dd;
                    uint16_t *dst = out + o;
                    if (dd >= 4) {
                        for (uint32_t i = 0; i < len; i += 4)
                            memcpy(dst + i, src + i, 8); // (may write up to 3 symbols past len: room is guaranteed)
                    } else {
                        for (uint32_t i = 0; i < len; ++i)
                            dst[i] = src[i];
                    }
                    o += len;
                }
                b.buf = cnt < 64 ? buf & ((1ull << cnt) - 1ull) : buf; // (drop the preview bits above cnt)
                b.cnt = cnt;
                b.pos = pos;
                if (why)
                    return fail(why);
            }
            // Careful loop: the last bytes of the input (every drop checks for underflow).
            while (!eob) {
                if (o + 300 > cap) {
                    if (o - WINDOW > max_out)
                        return fail("output larger than any real chunk's");
                    c.sym.resize(cap * 2);
                    out = c.sym.data();
                    cap = c.sym.size();
                }
                b.refill();
                uint32_t e = lt[b.buf & ((1u << LIT_PB) - 1u)];
                if (e & 0x80000000u) {
                    const uint32_t sb = e & 31u;
                    e = lt[((e >> 8) & 0xFFFFu) + ((uint32_t)(b.buf >> LIT_PB) & ((1u << sb) - 1u))];
                    b.drop(LIT_PB);
                }
                uint32_t l = e & 31u;
                if (!l)
                    return fail("invalid literal/length code");
                b.drop((int)l);
                uint32_t sy = (e >> 8) & 0xFFFFu;
                if (b.over)
                    return fail("compressed data ended inside a deflate block");
                if (sy < 256) {
                    out[o++] = (uint16_t)sy;
                    continue;
                }
                if (sy == 256)
                    break;
                sy -= 257;
                if (sy >= 29)
                    return fail("invalid literal/length code");
                const uint32_t len = LEN_BASE[sy] + b.peek(LEN_EXTRA[sy]);
                b.drop(LEN_EXTRA[sy]);
                b.refill();
                e = dt[b.buf & ((1u << DIST_PB) - 1u)];
                if (e & 0x80000000u) {
                    const uint32_t sb = e & 31u;
                    e = dt[((e >> 8) & 0xFFFFu) + ((uint32_t)(b.buf >> DIST_PB) & ((1u << sb) - 1u))];
                    b.drop(DIST_PB);
                }
                l = e & 31u;
                if (!l)
                    return fail("invalid distance code");
                b.drop((int)l);
                const uint32_t ds = (e >> 8) & 0xFFFFu;
                if (ds >= 30)
                    return fail("invalid distance code");
                const uint32_t dd = DIST_BASE[ds] + b.peek(DIST_EXTRA[ds]);
                b.drop(DIST_EXTRA[ds]);
                if (b.over)
                    return fail("compressed data ended inside a deflate block");
                for (uint32_t i = 0; i < len; ++i)
                    out[o + i] = out[o + i - dd];
                o += len;
            }
            if (b.over)
                return fail("compressed data ended inside a deflate block");
        }
        c.end_bit = b.bit_pos();
        c.n_out = o - WINDOW;
        if (bfinal)
            c.final_seen = true;
    }
    c.n_out = o - WINDOW;
}

// First bit position in [from_bit, to_bit) where a non-final dynamic-Huffman block header parses.  npos: none.
constexpr uint64_t NPOS = ~0ull;
inline uint64_t find_block(const uint8_t *data, size_t n, uint64_t from_bit, uint64_t to_bit)
{
    Huff lit, dist;
    Bits b;
    b.p = data;
    b.n = n;
    for (uint64_t at = from_bit; at < to_bit; ++at) {
        // cheap rejection on the next 13 bits before anything is built: BFINAL = 0, BTYPE = 2 (binary 10, LSB first),
        // HLIT <= 29
        const size_t byte = (size_t)(at >> 3);
        if (byte + 4 > n)
            return NPOS;
        uint32_t w;
        memcpy(&w, data + byte, 4);
        w >>= (at & 7);
        if ((w & 7u) != 4u)
            continue;
        if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u)
            continue;
        b.seek_bit(at + 3);
        if (read_dynamic_header(b, lit, dist) && !b.over)
            return at;
    }
    return NPOS;
}

// symbols -> bytes; window = the WINDOW bytes before the chunk (only its last bytes may be real history: the caller
// compares the returned lowest window index any marker used against what it has).  Returns WINDOW if no marker occurred.
inline uint32_t resolve(const uint16_t *sym, size_t n, const uint8_t *window, uint8_t *out)
{
    uint32_t lowest = WINDOW;
    for (size_t i = 0; i < n; ++i) {
        const uint16_t s = sym[i];
        if (s < 256) {
            out[i] = (uint8_t)s;
        } else {
            const uint32_t w = s & 0x7FFFu;
            lowest = w < lowest ? w : lowest;
            out[i] = window[w];
        }
    }
    return lowest;
}

} // namespace kmm_inflate
