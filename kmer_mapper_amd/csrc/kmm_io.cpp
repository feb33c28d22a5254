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
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

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
    return (size_t)rd16(p + 16) + 1;
}

struct Member {
    size_t src_off, src_len; // whole member in the file
    size_t dst_off, dst_len; // where its inflated bytes go
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
    std::deque<std::vector<uint8_t>> ready;
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
        static thread_local void *dec = nullptr;
        if (!dec)
            dec = dl.alloc();
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
        if (out + isize > (size_t)n)
            break;
        plan.push_back({p, ms, out, isize});
        out += isize;
        p += ms;
    }
    if (plan.empty() && p < h->size && done == 0) {
        // not even one member fits: inflate it aside and hand out what fits
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
    std::vector<uint8_t> out;
    auto push = [&](std::vector<uint8_t> &&v) {
        std::unique_lock<std::mutex> g(h->m);
        h->cv.wait(g, [&] { return h->ahead_stop || h->ready_bytes < max_ahead; });
        if (h->ahead_stop)
            return false;
        h->ready_bytes += v.size();
        h->ready.push_back(std::move(v));
        h->cv.notify_all();
        return true;
    };
    while (p < h->size || open) {
        if (!open) {
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
            out.resize(piece);
            z.next_out = out.data();
            z.avail_out = (uInt)piece;
            rc = inflate(&z, Z_NO_FLUSH);
            fed = true;
            if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
                err = std::string("corrupt gzip stream: ") + (z.msg ? z.msg : "inflate error");
                break;
            }
            out.resize(piece - z.avail_out);
            if (!out.empty() && !push(std::move(out))) {
                err = "stopped";
                break;
            }
            out = std::vector<uint8_t>();
            if (rc == Z_BUF_ERROR && z.avail_in == 0)
                break;
        }
        p += chunk - z.avail_in;
        if (!err.empty())
            break;
        if (rc == Z_STREAM_END) { // next member of a concatenated file (zlib has checked CRC32 and ISIZE)
            inflateEnd(&z);
            open = false;
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
        std::vector<uint8_t> &v = h->ready.front();
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
    if (h->kind != 2)
        h->pool = new Pool(n_threads);
    else
        h->ahead = std::thread(ahead_main, h);
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
