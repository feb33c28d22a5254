// kmm_io.cpp — libkmm_io.so: host-side read-file input for the GPU mapper (plain g++, no HIP).
//
// What it replaces: `bnp.open(args.reads).read_chunks(...)` feeding the mapper in the reference
// (kmer_mapper/command_line_interface.py:102-103,109-111; ".fa, .fq, .fa.gz, or fq.gz", Readme.md:11) and the igzip
// reader the reference meant to use (kmer_mapper/util.py:78-101).  The GPU maps a 600 MB batch in ~10 ms, so the
// host's job is to put file bytes into the PINNED staging buffer of the next map call as fast as the cores allow:
//   * BGZF (.gz made of independent members with their sizes up front, what bgzip / htslib write): the members
//     that fit the caller's buffer are planned from their headers and trailers, then inflated IN PARALLEL, each
//     straight to its final place in the caller's buffer (no intermediate copy), CRC32 and ISIZE checked;
//   * plain gzip (one or several concatenated members): one deflate stream cannot be split, so a read-ahead
//     thread inflates it in large pieces while the GPU works on the previous batch;
//   * uncompressed files: parallel pread into the buffer (a single thread copies the page cache at ~8 GB/s).
// Inflate engine: libdeflate when the shared library is present (dlopen, prototypes declared here: ~3x zlib's
// rate), else zlib.  C ABI, no exceptions across it; errors as negative return values + kmm_io_error().
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kmm_inflate.hpp"

namespace {

thread_local std::string g_err;

// ---- libdeflate through dlopen (optional) --------------------------------------------------------------------
struct Deflate {
    void *lib = nullptr;
    void *(*alloc)() = nullptr;
    void (*free_)(void *) = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
    Deflate()
    {
        if (getenv("KMM_IO_NO_LIBDEFLATE"))
            return;
        for (const char *name : {"libdeflate.so.0", "libdeflate.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib)
                break;
        }
        if (!lib)
            return;
        alloc = (void *(*)())dlsym(lib, "libdeflate_alloc_decompressor");
        free_ = (void (*)(void *))dlsym(lib, "libdeflate_free_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(lib, "libdeflate_deflate_decompress");
        crc32 = (uint32_t(*)(uint32_t, const void *, size_t))dlsym(lib, "libdeflate_crc32");
        ok = alloc && free_ && decompress && crc32;
    }
};
Deflate &deflate_lib()
{
    static Deflate d;
    return d;
}

// ---- a small pool of worker threads ---------------------------------------------------------------------------
class Pool {
  public:
    explicit Pool(int n)
    {
        for (int i = 0; i < n; ++i)
            workers_.emplace_back([this] { run(); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_)
            t.join();
    }
    // runs fn(i) for i in [0, n) on the pool and returns when all are done
    void parallel_for(int64_t n, const std::function<void(int64_t)> &fn)
    {
        if (n <= 0)
            return;
        std::unique_lock<std::mutex> g(m_);
        fn_ = &fn;
        next_ = 0;
        end_ = n;
        pending_ = n;
        cv_.notify_all();
        done_cv_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
    int size() const { return (int)workers_.size(); }

  private:
    void run()
    {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return stop_ || (fn_ && next_ < end_); });
            if (stop_)
                return;
            const int64_t i = next_++;
            const std::function<void(int64_t)> *fn = fn_;
            g.unlock();
            (*fn)(i);
            g.lock();
            if (--pending_ == 0)
                done_cv_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int64_t)> *fn_ = nullptr;
    int64_t next_ = 0, end_ = 0, pending_ = 0;
    bool stop_ = false;
};

uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// total size of the BGZF member starting at p (n bytes available), or 0 if it is not one
size_t bgzf_member_size(const uint8_t *p, size_t n)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4)
        return 0;
    const unsigned xlen = rd16(p + 10);
    if (xlen < 6 || p[12] != 'B' || p[13] != 'C' || rd16(p + 14) != 2)
        return 0;
    const size_t ms = (size_t)rd16(p + 16) + 1;
    if (ms < 12 + (size_t)xlen + 8) // header + extra field + CRC32 + ISIZE: the planner reads the trailer at p + ms - 4
        return 0;
    return ms;
}

struct Member {
    size_t src_off, src_len; // whole member in the file
    size_t dst_off, dst_len; // where its inflated bytes go
};

// inflated bytes on their way from the read-ahead thread to the reader (uninitialised storage: no zero-fill)
struct Piece {
    std::unique_ptr<uint8_t[]> p;
    size_t n = 0;
    Piece() = default;
    explicit Piece(size_t size) : p(new uint8_t[size ? size : 1]), n(size) {}
    uint8_t *data() { return p.get(); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
};

} // namespace

struct kmm_io {
    int fd = -1;
    size_t size = 0;
    const uint8_t *map = nullptr; // compressed file, mapped read-only (gz modes)
    int kind = 0;                 // 0 plain file, 1 BGZF, 2 gzip stream
    size_t pos = 0;               // next unread byte of the file (plain: of the uncompressed stream)
    Pool *pool = nullptr;
    std::string err;
    // BGZF: inflated bytes of a member that did not fit the caller's buffer whole
    std::vector<uint8_t> carry;
    size_t carry_pos = 0;
    // gzip stream: read-ahead thread
    std::thread ahead;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Piece> ready;
    size_t ready_bytes = 0, front_pos = 0;
    bool ahead_done = false, ahead_stop = false;
    std::string ahead_err;
};

namespace {

int fail(kmm_io *h, const std::string &msg)
{
    if (h)
        h->err = msg;
    g_err = msg;
    return -1;
}

// one BGZF member -> dst (exactly dst_len bytes); false on any inconsistency
bool inflate_member(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_len, std::string *why)
{
    const unsigned xlen = rd16(src + 10);
    if (src_len < 12 + (size_t)xlen + 8) {
        *why = "truncated BGZF member";
        return false;
    }
    const uint8_t *payload = src + 12 + xlen;
    const size_t payload_len = src_len - 12 - xlen - 8;
    const uint32_t crc = rd32(src + src_len - 8), isize = rd32(src + src_len - 4);
    if (isize != dst_len) {
        *why = "BGZF member's ISIZE changed under the reader";
        return false;
    }
    Deflate &dl = deflate_lib();
    if (dl.ok) {
        // one decompressor per worker thread, freed when the thread ends (every kmm_io_open has its own pool)
        struct Dec {
            void *p = nullptr;
            ~Dec()
            {
                if (p)
                    deflate_lib().free_(p);
            }
        };
        static thread_local Dec dec_holder;
        if (!dec_holder.p)
            dec_holder.p = dl.alloc();
        void *dec = dec_holder.p;
        size_t got = 0;
        if (!dec || dl.decompress(dec, payload, payload_len, dst, dst_len, &got) != 0 || got != dst_len) {
            *why = "corrupt BGZF member (inflate failed or size differs from the trailer)";
            return false;
        }
        if (dl.crc32(0, dst, dst_len) != crc) {
            *why = "corrupt BGZF member: CRC32 of the inflated bytes differs from the trailer";
            return false;
        }
        return true;
    }
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) {
        *why = "inflateInit2 failed";
        return false;
    }
    z.next_in = const_cast<uint8_t *>(payload);
    z.avail_in = (uInt)payload_len;
    z.next_out = dst;
    z.avail_out = (uInt)dst_len;
    const int rc = inflate(&z, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && z.total_out == dst_len;
    inflateEnd(&z);
    if (!ok) {
        *why = "corrupt BGZF member (inflate failed or size differs from the trailer)";
        return false;
    }
    if ((uint32_t)::crc32(0L, dst, (uInt)dst_len) != crc) {
        *why = "corrupt BGZF member: CRC32 of the inflated bytes differs from the trailer";
        return false;
    }
    return true;
}

int64_t read_bgzf(kmm_io *h, uint8_t *dst, int64_t n)
{
    int64_t done = 0;
    if (h->carry_pos < h->carry.size()) { // the rest of a member that was larger than an earlier buffer
        const size_t take = std::min<size_t>(h->carry.size() - h->carry_pos, (size_t)n);
        memcpy(dst, h->carry.data() + h->carry_pos, take);
        h->carry_pos += take;
        done = (int64_t)take;
        if (h->carry_pos < h->carry.size())
            return done;
        h->carry.clear();
        h->carry_pos = 0;
    }
    // plan: the members that fit the rest of the buffer whole
    std::vector<Member> plan;
    size_t p = h->pos, out = (size_t)done;
    while (p < h->size) {
        const size_t ms = bgzf_member_size(h->map + p, h->size - p);
        if (!ms || p + ms > h->size)
            return fail(h, p + 18 > h->size || (ms && p + ms > h->size) ? "BGZF file is truncated (member reaches beyond the end of the file)"
                                                                        : "not a BGZF member at byte " + std::to_string(p));
        const size_t isize = rd32(h->map + p + ms - 4);
        // (deflate expands at most ~1032 : 1: an ISIZE beyond that is damage, and must not size a buffer — a fuzzed
        // file made the carry buffer below 4 GB)
        if (isize > ms * 1032 + 64)
            return fail(h, "corrupt BGZF member: ISIZE " + std::to_string(isize) + " is impossible for " + std::to_string(ms) +
                               " compressed bytes, at byte " + std::to_string(p));
        if (out + isize > (size_t)n)
            break;
        plan.push_back({p, ms, out, isize});
        out += isize;
        p += ms;
    }
    if (out == 0 && p < h->size) {
        // nothing planned holds a byte (no member before p, or only empty ones — a BGZF end-of-file marker in the middle
        // of a concatenated file) and the member at p does not fit: inflate it aside and hand out what fits.  (0 is
        // returned at the end of the file only.)
        const size_t ms = bgzf_member_size(h->map + p, h->size - p);
        const size_t isize = rd32(h->map + p + ms - 4);
        h->carry.resize(isize);
        std::string why;
        if (!inflate_member(h->map + p, ms, h->carry.data(), isize, &why))
            return fail(h, why);
        h->pos = p + ms;
        const size_t take = std::min<size_t>(isize, (size_t)n);
        memcpy(dst, h->carry.data(), take);
        h->carry_pos = take;
        if (take == isize) {
            h->carry.clear();
            h->carry_pos = 0;
        }
        return (int64_t)take;
    }
    std::atomic<bool> bad(false);
    std::mutex em;
    std::string first_err;
    // tasks of ~16 members each: a member is 64 KiB at most, a task a few hundred microseconds
    const int64_t per = 16, n_tasks = ((int64_t)plan.size() + per - 1) / per;
    std::function<void(int64_t)> fn = [&](int64_t t) {
        for (int64_t i = t * per; i < std::min<int64_t>((t + 1) * per, (int64_t)plan.size()) && !bad.load(); ++i) {
            const Member &mb = plan[(size_t)i];
            std::string why;
            if (mb.dst_len && !inflate_member(h->map + mb.src_off, mb.src_len, dst + mb.dst_off, mb.dst_len, &why)) {
                bad.store(true);
                std::lock_guard<std::mutex> g(em);
                if (first_err.empty())
                    first_err = why + " at byte " + std::to_string(mb.src_off);
            }
        }
    };
    h->pool->parallel_for(n_tasks, fn);
    if (bad.load())
        return fail(h, first_err);
    h->pos = p;
    return (int64_t)out;
}

// gzip stream: inflated by a read-ahead thread in pieces of 8 MiB, at most 8 pieces ahead
void ahead_main(kmm_io *h)
{
    const size_t piece = (size_t)8 << 20, max_ahead = (size_t)64 << 20;
    z_stream z;
    memset(&z, 0, sizeof z);
    bool open = false, fed = false;
    size_t p = 0;
    std::string err;
    auto push = [&](Piece &&v) {
        std::unique_lock<std::mutex> g(h->m);
        h->cv.wait(g, [&] { return h->ahead_stop || h->ready_bytes < max_ahead; });
        if (h->ahead_stop)
            return false;
        h->ready_bytes += v.size();
        h->ready.push_back(std::move(v));
        h->cv.notify_all();
        return true;
    };
    bool any_member = false;
    while (p < h->size || open) {
        if (!open) {
            if (any_member) {
                // bytes behind a member's end marker: zero padding is skipped (gzip.open — the reference's reader through
                // bnp.open — does the same), another member must start with the gzip magic, anything else is an error
                while (p < h->size && h->map[p] == 0)
                    ++p;
                if (p >= h->size)
                    break;
                if (p + 2 > h->size || h->map[p] != 0x1f || h->map[p + 1] != 0x8b) {
                    err = "trailing bytes after the gzip stream are neither zero padding nor another gzip member (at byte " +
                          std::to_string(p) + ")";
                    break;
                }
            }
            memset(&z, 0, sizeof z);
            if (inflateInit2(&z, 31) != Z_OK) {
                err = "inflateInit2 failed";
                break;
            }
            open = true;
            fed = false;
        }
        if (p >= h->size) {
            if (fed)
                err = "compressed file ended before the end-of-stream marker was reached";
            inflateEnd(&z);
            open = false;
            break;
        }
        const size_t chunk = std::min<size_t>(h->size - p, (size_t)1 << 30);
        z.next_in = const_cast<uint8_t *>(h->map + p);
        z.avail_in = (uInt)chunk;
        int rc = Z_OK;
        while (z.avail_in && rc != Z_STREAM_END) {
            Piece out(piece);
            z.next_out = out.data();
            z.avail_out = (uInt)piece;
            rc = inflate(&z, Z_NO_FLUSH);
            fed = true;
            if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
                err = std::string("corrupt gzip stream: ") + (z.msg ? z.msg : "inflate error");
                break;
            }
            out.n = piece - z.avail_out;
            if (!out.empty() && !push(std::move(out))) {
                err = "stopped";
                break;
            }
            if (rc == Z_BUF_ERROR && z.avail_in == 0)
                break;
        }
        p += chunk - z.avail_in;
        if (!err.empty())
            break;
        if (rc == Z_STREAM_END) { // next member of a concatenated file (zlib has checked CRC32 and ISIZE)
            inflateEnd(&z);
            open = false;
            any_member = true;
        }
    }
    if (open)
        inflateEnd(&z);
    std::lock_guard<std::mutex> g(h->m);
    if (err != "stopped")
        h->ahead_err = err;
    h->ahead_done = true;
    h->cv.notify_all();
}

// ---- one gzip member on many cores (kmm_inflate.hpp) ---------------------------------------------------------------
// end of the gzip member header at p (RFC 1952), or 0 if it is not one / truncated
size_t gzip_header_end(const uint8_t *d, size_t n, size_t p)
{
    if (p + 10 > n || d[p] != 0x1f || d[p + 1] != 0x8b || d[p + 2] != 8)
        return 0;
    const unsigned flg = d[p + 3];
    size_t q = p + 10;
    if (flg & 4) { // FEXTRA
        if (q + 2 > n)
            return 0;
        q += 2 + (size_t)rd16(d + q);
    }
    for (unsigned bit : {8u, 16u}) // FNAME, FCOMMENT: zero-terminated
        if (flg & bit) {
            while (q < n && d[q])
                ++q;
            ++q;
        }
    if (flg & 2) // FHCRC
        q += 2;
    return q <= n ? q : 0;
}

// The stream is cut into chunks of `csize` compressed bytes; a wave = as many chunks as there are workers.  Every
// chunk but the wave's first starts at the first dynamic-block header found in its byte range and is decoded with
// markers for the unknown history; the wave's first chunk starts where the previous wave ended.  A chunk is accepted if
// the chunk before it stopped exactly at its start (a block boundary); else it was a false start and the chunk
// before it decodes on through its range.  Then the 32 KiB windows are resolved in order, the chunks in parallel.
void ahead_main_parallel(kmm_io *h)
{
    using namespace kmm_inflate;
    const size_t max_ahead = (size_t)512 << 20;
    const uint8_t *d = h->map;
    const size_t n = h->size;
    const int T = h->pool->size();
    std::string err;
    auto push = [&](Piece &&v) {
        if (v.empty())
            return true;
        std::unique_lock<std::mutex> g(h->m);
        h->cv.wait(g, [&] { return h->ahead_stop || h->ready_bytes < max_ahead; });
        if (h->ahead_stop)
            return false;
        h->ready_bytes += v.size();
        h->ready.push_back(std::move(v));
        h->cv.notify_all();
        return true;
    };
    std::vector<Chunk> chunks((size_t)T);
    std::vector<uint64_t> starts((size_t)T);
    std::vector<Piece> bytes((size_t)T);
    std::vector<std::vector<uint8_t>> windows((size_t)T + 1, std::vector<uint8_t>(WINDOW, 0));
    std::vector<uint32_t> crcs((size_t)T), lowest((size_t)T);
    size_t p = 0;
    bool any_member = false, stopped = false;
    // A file of many SMALL plain-gzip members (cat of small .gz files): a wave planned over the rest of the FILE puts T - 1
    // speculative chunks into the members behind this one, all thrown away when chunk 0 reaches BFINAL — T-fold work.
    // So once a member has ended inside the first chunk of its first wave, the next member's first wave is that one chunk
    // alone, decoded in order; a member that does not end in it goes on in parallel waves as before.
    int small_run = 0;
    while (p < n && err.empty() && !stopped) {
        if (any_member) { // behind a member: zero padding is skipped, another member must start with the magic
            while (p < n && d[p] == 0)
                ++p;
            if (p >= n)
                break;
            if (p + 2 > n || d[p] != 0x1f || d[p + 1] != 0x8b) {
                err = "trailing bytes after the gzip stream are neither zero padding nor another gzip member (at byte " +
                      std::to_string(p) + ")";
                break;
            }
        }
        const size_t body = gzip_header_end(d, n, p);
        if (!body) {
            err = p + 18 > n ? "compressed file ended before the end-of-stream marker was reached"
                             : "corrupt gzip stream: incorrect header check";
            break;
        }
        uint64_t cur_bit = (uint64_t)body * 8;
        uint32_t crc = 0;
        uint64_t total_out = 0;
        size_t hist = 0; // real bytes at the end of windows[0]
        bool final = false;
        int barren = 0;  // waves in a row whose byte ranges held no dynamic block start (stored blocks: incompressible data)
        int waves = 0;
        while (!final && err.empty() && !stopped) {
            const size_t cur_byte = (size_t)(cur_bit >> 3), left = n - cur_byte;
            size_t csize = left / (size_t)(2 * T);
            csize = std::min<size_t>(std::max<size_t>(csize, (size_t)256 << 10), (size_t)2 << 20);
            if (const char *env = getenv("KMM_IO_GZIP_CHUNK")) // (tests: many chunks in a small file)
                csize = std::max<size_t>((size_t)strtoull(env, nullptr, 10), 1024);
            int n_ch = (int)std::min<size_t>((size_t)T, (left + csize - 1) / csize);
            if (barren >= 2) { // no starting points to be found: one chunk per wave, decoded in order, no search
                csize *= (size_t)n_ch;
                n_ch = 1;
            } else if (waves == 0 && small_run > 0) { // the members before this one were small: see above
                n_ch = 1;
            }
            ++waves;
            const uint64_t wave_end_bit = (uint64_t)std::min(n, cur_byte + (size_t)n_ch * csize) * 8;
            // phase A: where the chunks start
            const bool dbg = getenv("KMM_IO_DEBUG") != nullptr;
            auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            const double t_a = now();
            starts[0] = cur_bit;
            std::function<void(int64_t)> find = [&](int64_t j) {
                if (j == 0)
                    return;
                const uint64_t a = (uint64_t)(cur_byte + (size_t)j * csize) * 8;
                const uint64_t b = std::min<uint64_t>((uint64_t)(cur_byte + (size_t)(j + 1) * csize) * 8, (uint64_t)n * 8);
                starts[(size_t)j] = find_block(d, n, a, b);
            };
            h->pool->parallel_for(n_ch, find);
            // phase B: decode; a chunk stops at the first block boundary at or behind the next chunk's start
            std::function<void(int64_t)> dec = [&](int64_t j) {
                Chunk &c = chunks[(size_t)j];
                if (starts[(size_t)j] == NPOS) {
                    c.failed = true;
                    c.n_out = 0;
                    return;
                }
                uint64_t stop = wave_end_bit;
                for (int x = (int)j + 1; x < n_ch; ++x)
                    if (starts[(size_t)x] != NPOS) {
                        stop = starts[(size_t)x];
                        break;
                    }
                c.init(starts[(size_t)j], csize * 5);
                decode_chunk(d, n, c, stop, j == 0 ? ~(size_t)0 : csize * 40);
            };
            h->pool->parallel_for(n_ch, dec);
            const double t_b = now();
            // accept the chunks that continue each other
            std::vector<int> ok;
            ok.push_back(0);
            Chunk *cur = &chunks[0];
            auto stream_error = [](const std::string &why) {
                return why.rfind("compressed data ended", 0) == 0 ? std::string("compressed file ended before the end-of-stream marker was reached")
                                                                  : "corrupt gzip stream: " + why;
            };
            if (cur->failed && !cur->final_seen) {
                err = stream_error(cur->err);
                break;
            }
            for (int j = 1; j < n_ch && !cur->final_seen; ++j) {
                if (starts[(size_t)j] == NPOS)
                    continue;
                Chunk &c = chunks[(size_t)j];
                if (cur->end_bit == starts[(size_t)j] && !c.failed) {
                    ok.push_back(j);
                    cur = &c;
                    continue;
                }
                // a false start (or one that decoded into an error): the chunk before it goes on through its range
                uint64_t stop = wave_end_bit;
                for (int x = j + 1; x < n_ch; ++x)
                    if (starts[(size_t)x] != NPOS) {
                        stop = starts[(size_t)x];
                        break;
                    }
                if (cur->end_bit < stop) {
                    decode_chunk(d, n, *cur, stop, ~(size_t)0);
                    if (cur->failed) {
                        err = stream_error(cur->err);
                        break;
                    }
                }
            }
            if (!err.empty())
                break;
            barren = (n_ch > 1 && ok.size() == 1) ? barren + 1 : (n_ch > 1 ? 0 : barren);
            if (waves == 1) // did the member end inside the first chunk's byte range?
                small_run = (cur->final_seen && (cur->end_bit >> 3) <= (uint64_t)(cur_byte + csize)) ? small_run + 1 : 0;
            // the windows in order (32 KiB each), then every accepted chunk in parallel
            const double t_c = now();
            for (size_t q = 0; q < ok.size(); ++q) {
                const Chunk &c = chunks[(size_t)ok[q]];
                const std::vector<uint8_t> &w = windows[q];
                std::vector<uint8_t> &wn = windows[q + 1];
                const size_t tail = std::min<size_t>(c.n_out, WINDOW);
                if (tail < WINDOW)
                    memmove(wn.data(), w.data() + tail, WINDOW - tail);
                (void)resolve(c.sym.data() + WINDOW + c.n_out - tail, tail, w.data(), wn.data() + WINDOW - tail);
            }
            std::function<void(int64_t)> res = [&](int64_t q) {
                const Chunk &c = chunks[(size_t)ok[(size_t)q]];
                Piece &out = bytes[(size_t)q];
                out = Piece(c.n_out);
                lowest[(size_t)q] = resolve(c.sym.data() + WINDOW, c.n_out, windows[(size_t)q].data(), out.data());
                Deflate &dl = deflate_lib();
                uint32_t cr = 0;
                if (dl.ok) {
                    cr = dl.crc32(0, out.data(), out.size());
                } else {
                    size_t done = 0;
                    while (done < out.size()) { // (zlib's length argument is 32 bits)
                        const size_t step = std::min<size_t>(out.size() - done, (size_t)1 << 30);
                        cr = (uint32_t)::crc32(cr, out.data() + done, (uInt)step);
                        done += step;
                    }
                }
                crcs[(size_t)q] = cr;
            };
            h->pool->parallel_for((int64_t)ok.size(), res);
            const double t_d = now();
            if (dbg)
                fprintf(stderr, "[kmm_io] wave at byte %zu: %d chunks of %zu bytes, %zu accepted; find+decode %.1f ms, continue %.1f ms, "
                        "resolve %.1f ms\n", cur_byte, n_ch, csize, ok.size(), (t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (t_d - t_c) * 1e3);
            for (size_t q = 0; q < ok.size() && err.empty(); ++q) {
                // a marker may only point at history that exists: the last `hist` bytes of the window
                if (lowest[q] < WINDOW && (size_t)(WINDOW - lowest[q]) > hist)
                    err = "corrupt gzip stream: invalid distance too far back";
                hist = std::min<size_t>(WINDOW, hist + bytes[q].size());
                crc = (uint32_t)crc32_combine(crc, crcs[q], (z_off_t)bytes[q].size());
                total_out += bytes[q].size();
            }
            if (!err.empty())
                break;
            for (size_t q = 0; q < ok.size(); ++q)
                if (!push(std::move(bytes[q]))) {
                    stopped = true;
                    break;
                }
            windows[0].swap(windows[ok.size()]);
            cur_bit = cur->end_bit;
            final = cur->final_seen;
            if (!final && cur_bit >= (uint64_t)n * 8)
                err = "compressed file ended before the end-of-stream marker was reached";
        }
        if (!err.empty() || stopped)
            break;
        const size_t t = (size_t)((cur_bit + 7) >> 3);
        if (t + 8 > n) {
            err = "compressed file ended before the end-of-stream marker was reached";
            break;
        }
        if (rd32(d + t) != crc) {
            err = "corrupt gzip stream: incorrect data check";
            break;
        }
        if (rd32(d + t + 4) != (uint32_t)total_out) {
            err = "corrupt gzip stream: incorrect length check";
            break;
        }
        p = t + 8;
        any_member = true;
    }
    std::lock_guard<std::mutex> g(h->m);
    if (!stopped)
        h->ahead_err = err;
    h->ahead_done = true;
    h->cv.notify_all();
}

int64_t read_gzip(kmm_io *h, uint8_t *dst, int64_t n)
{
    int64_t done = 0;
    std::unique_lock<std::mutex> g(h->m);
    while (done < n) {
        h->cv.wait(g, [&] { return !h->ready.empty() || h->ahead_done; });
        if (h->ready.empty()) {
            if (!h->ahead_err.empty())
                return fail(h, h->ahead_err);
            break; // end of stream
        }
        Piece &v = h->ready.front();
        const size_t take = std::min<size_t>(v.size() - h->front_pos, (size_t)(n - done));
        g.unlock();
        memcpy(dst + done, v.data() + h->front_pos, take);
        g.lock();
        done += (int64_t)take;
        h->front_pos += take;
        if (h->front_pos == v.size()) {
            h->ready_bytes -= v.size();
            h->ready.pop_front();
            h->front_pos = 0;
            h->cv.notify_all();
        }
    }
    return done;
}

int64_t read_plain(kmm_io *h, uint8_t *dst, int64_t n)
{
    const size_t want = std::min<size_t>((size_t)n, h->size - h->pos);
    if (!want)
        return 0;
    const size_t slice = (size_t)8 << 20;
    const int64_t n_tasks = (int64_t)((want + slice - 1) / slice);
    std::atomic<bool> bad(false);
    const size_t base = h->pos;
    std::function<void(int64_t)> fn = [&](int64_t t) {
        size_t off = (size_t)t * slice;
        const size_t end = std::min(want, off + slice);
        while (off < end) {
            const ssize_t r = pread(h->fd, dst + off, end - off, (off_t)(base + off));
            if (r <= 0) {
                bad.store(true);
                return;
            }
            off += (size_t)r;
        }
    };
    h->pool->parallel_for(n_tasks, fn);
    if (bad.load())
        return fail(h, "pread failed (file changed under the reader?)");
    h->pos += want;
    return (int64_t)want;
}

} // namespace

extern "C" {

const char *kmm_io_error(void) { return g_err.c_str(); }

// 1 if libdeflate does the inflating, 0 if zlib
int kmm_io_engine(void) { return deflate_lib().ok ? 1 : 0; }

kmm_io *kmm_io_open(const char *path, int n_threads)
{
    if (!path) {
        g_err = "path is NULL";
        return nullptr;
    }
    kmm_io *h = new kmm_io();
    h->fd = open(path, O_RDONLY);
    struct stat st;
    if (h->fd < 0 || fstat(h->fd, &st) != 0) {
        g_err = std::string("cannot open ") + path;
        if (h->fd >= 0)
            close(h->fd);
        delete h;
        return nullptr;
    }
    h->size = (size_t)st.st_size;
    if (n_threads < 1)
        n_threads = 1;
    uint8_t head[18] = {0};
    const ssize_t got = pread(h->fd, head, sizeof head, 0);
    const bool gz = got >= 2 && head[0] == 0x1f && head[1] == 0x8b;
    if (gz) {
        h->kind = bgzf_member_size(head, (size_t)got) ? 1 : 2;
        if (h->size) {
            void *m = mmap(nullptr, h->size, PROT_READ, MAP_PRIVATE, h->fd, 0);
            if (m == MAP_FAILED) {
                g_err = std::string("mmap failed for ") + path;
                close(h->fd);
                delete h;
                return nullptr;
            }
            h->map = (const uint8_t *)m;
            madvise(m, h->size, MADV_SEQUENTIAL);
        }
    }
    if (h->kind != 2) {
        h->pool = new Pool(n_threads);
    } else if (n_threads >= 2 && !getenv("KMM_IO_GZIP_SERIAL") &&
               h->size >= (getenv("KMM_IO_GZIP_CHUNK") ? (size_t)1 : (size_t)1 << 20)) {
        // one gzip member on many cores: speculative chunks with markers for the unknown history (kmm_inflate.hpp)
        h->pool = new Pool(n_threads);
        h->ahead = std::thread(ahead_main_parallel, h);
    } else {
        h->ahead = std::thread(ahead_main, h);
    }
    return h;
}

// 0 plain file, 1 BGZF, 2 gzip stream
int kmm_io_kind(const kmm_io *h) { return h ? h->kind : -1; }

// Fills dst with up to n bytes of the (inflated) stream; returns the number of bytes, 0 at the end, -1 on error.  BGZF:
// fewer than n bytes come back when the next member does not fit the rest of the buffer.
int64_t kmm_io_read(kmm_io *h, uint8_t *dst, int64_t n)
{
    if (!h || !dst || n < 0) {
        g_err = "bad argument";
        return -1;
    }
    if (n == 0)
        return 0;
    if (h->kind == 1)
        return read_bgzf(h, dst, n);
    if (h->kind == 2)
        return read_gzip(h, dst, n);
    return read_plain(h, dst, n);
}

// plain files only: position the stream (rank byte ranges)
int kmm_io_seek(kmm_io *h, int64_t pos)
{
    if (!h || h->kind != 0 || pos < 0 || (size_t)pos > h->size) {
        g_err = "seek needs an uncompressed file and a position inside it";
        return -1;
    }
    h->pos = (size_t)pos;
    return 0;
}

void kmm_io_close(kmm_io *h)
{
    if (!h)
        return;
    if (h->kind == 2) {
        {
            std::lock_guard<std::mutex> g(h->m);
            h->ahead_stop = true;
        }
        h->cv.notify_all();
        if (h->ahead.joinable())
            h->ahead.join();
    }
    delete h->pool;
    if (h->map)
        munmap(const_cast<uint8_t *>(h->map), h->size);
    if (h->fd >= 0)
        close(h->fd);
    delete h;
}

} // extern "C"
