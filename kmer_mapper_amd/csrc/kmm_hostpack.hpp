// kmm_hostpack.hpp — part of libkmm (host code only, no HIP inside); included by kmm.hip, compiled by itself in tests/test_host.py.
//
// The host cores' share of the read bytes.  Reads that arrive in HOST memory cross the PCIe link as they are: 1.25 bytes per
// k-mer at ~55 GB/s is 44 G k-mers/s per GPU, a quarter of what the kernels map.  The reference
// spends its `-t` worker processes on exactly these bytes (bnp.as_encoded_array + get_kmers, kmer_mapper/util.py:71-75, in
// command_line_interface.py:124-130,168); here the same cores only do the byte -> 2-bit code step, so that the link carries
// 0.31 bytes per k-mer and pass 1 of the radix path reads the form its tiles keep in LDS anyway (16 codes per 32-bit word,
// first base lowest):
//   * flat reads (kmm_map_reads / kmm_map_reads_uniform with host pointers): FlatJob — 4 MiB tasks on a persistent pool;
//   * raw FASTQ / two-line FASTA records (kmm_map_records with a host pointer — the file mapping or the inflater's output,
//     no pinned copy of the raw bytes): RecordsJob — the bytes of the sequence lines go straight to the 2-bit stream and
//     the read-start bitset, in slices claimed in order by the workers: newline census of the slice -> the line number and
//     flat position of its first byte from the slice before it (a chained prefix, like a decoupled look-back scan) -> the
//     slice's lines packed from cache.  Same rules as the device-side compaction (kmm_records.hpp): line index mod period
//     == 1 is the sequence line, '\r' is dropped and breaks the read, `consumed` = the byte behind the last newline whose
//     1-based count is a multiple of the period.
// The default lookup table only (A C G T a c g t -> 0..3, N n -> 0 as command_line_interface.py:41).  A byte outside it, or a
// record line that does not start with '@' / '+' / '>', makes the packer give up: the call takes the ordinary route, whose
// kernels report the offending byte's offset (nothing was mapped here).
#pragma once

#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kmm_hostpack {

// Cores this process may keep busy: its affinity mask, cut by the cgroup's CPU quota where there is one (a container with
// 256 visible CPUs and cpu.max = "1600000 100000" gets 16 cores' worth of time: more busy threads are throttled, not faster).
inline int cpu_budget()
{
    int n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0)
        n = CPU_COUNT(&set);
    if (n < 1)
        n = (int)sysconf(_SC_NPROCESSORS_ONLN);
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2
        char q[64] = {0};
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0)
            quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { // cgroup v1
        if (fscanf(g, "%lld", &quota) != 1)
            quota = -1;
        fclose(g);
        if (FILE *p = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(p, "%lld", &period) != 1)
                period = 0;
            fclose(p);
        }
    }
    // (all but one of the CPUs of the mask: the packing threads hand work to each other in order, and a thread that has to
    // share its CPU with the caller's own threads stalls the ones behind it)
    if (n > 1)
        --n;
    if (quota > 0 && period > 0) {
        const long long c = (quota + period - 1) / period;
        if (c >= 1 && c < n)
            n = (int)c;
    }
    return n < 1 ? 1 : n;
}

// ------------------------------------------------------------------------------------------------------------------
// byte -> code tables
// ------------------------------------------------------------------------------------------------------------------
struct CodeTab {
    uint8_t t[256];
    CodeTab()
    {
        memset(t, 0x80, sizeof t);
        t[(int)'A'] = t[(int)'a'] = 0;
        t[(int)'C'] = t[(int)'c'] = 1;
        t[(int)'G'] = t[(int)'g'] = 2;
        t[(int)'T'] = t[(int)'t'] = 3;
        t[(int)'N'] = t[(int)'n'] = 0;
    }
};
inline const CodeTab &code_tab()
{
    static const CodeTab tab;
    return tab;
}

// n bases -> ceil(n / 4) bytes (the last byte zero-padded); false = a byte that is not a nucleotide
inline bool pack2_scalar(const uint8_t *src, size_t n, uint8_t *dst)
{
    const uint8_t *t = code_tab().t;
    uint32_t bad = 0;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const uint32_t a = t[src[i]], b = t[src[i + 1]], c = t[src[i + 2]], d = t[src[i + 3]];
        bad |= a | b | c | d;
        dst[i >> 2] = (uint8_t)(a | (b << 2) | (c << 4) | (d << 6));
    }
    if (i < n) {
        uint32_t v = 0;
        for (size_t j = i; j < n; ++j) {
            const uint32_t a = t[src[j]];
            bad |= a;
            v |= (a & 3u) << (2 * (j - i));
        }
        dst[i >> 2] = (uint8_t)v;
    }
    return !(bad & 0x80u);
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline bool pack2_avx2(const uint8_t *src, size_t n, uint8_t *dst)
{
    // by the low nibble of the upper-cased byte: A = 0x41 -> 1, C = 0x43 -> 3, G = 0x47 -> 7, T = 0x54 -> 4, N = 0x4E -> 14
    const __m256i expect = _mm256_setr_epi8((char)0xFF, 0x41, (char)0xFF, 0x43, 0x54, (char)0xFF, (char)0xFF, 0x47, (char)0xFF,
                                            (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, (char)0xFF, 0x4E, (char)0xFF, (char)0xFF, 0x41,
                                            (char)0xFF, 0x43, 0x54, (char)0xFF, (char)0xFF, 0x47, (char)0xFF, (char)0xFF, (char)0xFF,
                                            (char)0xFF, (char)0xFF, (char)0xFF, 0x4E, (char)0xFF);
    const __m256i codes = _mm256_setr_epi8(0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i up_mask = _mm256_set1_epi8((char)0xDF), nib_mask = _mm256_set1_epi8(0x0F);
    const __m256i m1 = _mm256_set1_epi16(0x0401);     // b0 + 4 b1 per 16-bit lane
    const __m256i m2 = _mm256_set1_epi32(0x00100001); // + 16 (b2 + 4 b3) per 32-bit lane
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1,
                                          -1, -1, -1, -1, -1, -1);
    __m256i bad = _mm256_setzero_si256();
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i up = _mm256_and_si256(c, up_mask);
        const __m256i nib = _mm256_and_si256(up, nib_mask);
        bad = _mm256_or_si256(bad, _mm256_xor_si256(_mm256_shuffle_epi8(expect, nib), up));
        const __m256i code = _mm256_shuffle_epi8(codes, nib);
        const __m256i w = _mm256_madd_epi16(_mm256_maddubs_epi16(code, m1), m2); // one packed byte per 32-bit lane
        const __m256i p = _mm256_shuffle_epi8(w, pick);
        const uint32_t lo = (uint32_t)_mm256_extract_epi32(p, 0), hi = (uint32_t)_mm256_extract_epi32(p, 4);
        const uint64_t out = (uint64_t)lo | ((uint64_t)hi << 32);
        memcpy(dst + (i >> 2), &out, 8);
    }
    bool ok = _mm256_testz_si256(bad, bad) != 0;
    if (i < n)
        ok = pack2_scalar(src + i, n - i, dst + (i >> 2)) && ok;
    return ok;
}

// AVX-512 VBMI (Zen 4 / 5, Ice Lake on): the 128-entry table is one two-register byte permute, 64 bases per step.
// Entry: the code, or 0x80 for a byte that is no nucleotide (bytes >= 0x80 carry the flag themselves).
struct Tab128 {
    alignas(64) uint8_t t[128];
    Tab128()
    {
        for (int i = 0; i < 128; ++i)
            t[i] = code_tab().t[i];
    }
};
inline const Tab128 &tab128()
{
    static const Tab128 tab;
    return tab;
}

#define KMM_AVX512_TARGET __attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi,bmi2")))

// 64 bytes (the lanes of `live`; the others read as zero and yield code 0) -> 128 bits of codes; *flags collects bit 7
KMM_AVX512_TARGET inline __m128i codes64_avx512(__m512i c, __m512i t0, __m512i t1, __m512i &flags)
{
    const __m512i l = _mm512_permutex2var_epi8(t0, c, t1); // table[c & 127]
    flags = _mm512_or_si512(flags, _mm512_or_si512(l, c));
    const __m512i code = _mm512_and_si512(l, _mm512_set1_epi8(3));
    const __m512i w = _mm512_madd_epi16(_mm512_maddubs_epi16(code, _mm512_set1_epi16(0x0401)), _mm512_set1_epi32(0x00100001));
    return _mm512_cvtepi32_epi8(w); // one packed byte per 32-bit lane -> 16 bytes
}

KMM_AVX512_TARGET inline bool pack2_avx512(const uint8_t *src, size_t n, uint8_t *dst)
{
    const __m512i t0 = _mm512_load_si512(tab128().t), t1 = _mm512_load_si512(tab128().t + 64);
    __m512i flags = _mm512_setzero_si512();
    size_t i = 0;
    static const bool streaming = [] {
        const char *env = getenv("KMM_HOSTPACK_STREAMING_STORES"); // (experiments: 0 = ordinary stores)
        return env ? atoi(env) != 0 : true;
    }();
    if (streaming && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        // streaming stores: the packed codes are read next by the GPU's copy engine, not by this core, and a line that is
        // written whole need not be fetched first (16 threads pack at the host's DRAM rate: every byte not moved counts)
        for (; i + 64 <= n; i += 64) {
            const __m512i c = _mm512_loadu_si512(src + i);
            _mm_stream_si128(reinterpret_cast<__m128i *>(dst + (i >> 2)), codes64_avx512(c, t0, t1, flags));
        }
        _mm_sfence();
    }
    for (; i + 64 <= n; i += 64) {
        const __m512i c = _mm512_loadu_si512(src + i);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + (i >> 2)), codes64_avx512(c, t0, t1, flags));
    }
    bool ok = _mm512_movepi8_mask(flags) == 0;
    if (i < n) {
        const size_t m = n - i;
        const __mmask64 live = ~0ull >> (64 - m);
        __m512i f2 = _mm512_setzero_si512();
        const __m512i c = _mm512_maskz_loadu_epi8(live, src + i);
        const __m128i out = codes64_avx512(c, t0, t1, f2);
        _mm_mask_storeu_epi8(dst + (i >> 2), (__mmask16)((1u << ((m + 3) / 4)) - 1u), out);
        ok = ok && (_mm512_movepi8_mask(f2) & live) == 0;
    }
    return ok;
}
#endif

enum Isa { ISA_SCALAR = 0, ISA_AVX2 = 1, ISA_AVX512 = 2 };
inline Isa isa()
{
#if defined(__x86_64__)
    static const Isa level = [] {
        if (const char *env = getenv("KMM_HOSTPACK_ISA")) // tests: 0 scalar, 1 AVX2, 2 AVX-512
            return (Isa)atoi(env);
        if (__builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
            __builtin_cpu_supports("bmi2"))
            return ISA_AVX512;
        return __builtin_cpu_supports("avx2") ? ISA_AVX2 : ISA_SCALAR;
    }();
    return level;
#else
    return ISA_SCALAR;
#endif
}

inline bool pack2(const uint8_t *src, size_t n, uint8_t *dst)
{
#if defined(__x86_64__)
    const Isa l = isa();
    if (l == ISA_AVX512)
        return pack2_avx512(src, n, dst);
    if (l == ISA_AVX2)
        return pack2_avx2(src, n, dst);
#endif
    return pack2_scalar(src, n, dst);
}

// ------------------------------------------------------------------------------------------------------------------
// A persistent pool: the handle's packing threads sleep between calls (round 4 spawned and joined them per map call).
//
// Placement.  A condition-variable broadcast is a "sync" wake-up: the kernel puts the woken threads on the WAKER's CPU,
// expecting it to sleep — all of them, measured: four workers sharing CPU 5 with the caller for a second, 250 ms per
// 616 MB of FASTQ instead of 50 — and the load balancer leaves threads that run in short slices where they are ("cache
// hot").  So every worker moves itself to a core of its own at the start of a job — worker i to the i-th of n cores
// spread evenly over the physical cores of the process's affinity mask (one hardware thread per core first), starting at a
// core drawn from the process id so that several processes do not pick the same ones — and then hands its affinity mask
// back: a hint that costs two system calls per job, not a pin.  KMM_HOST_PACK_SPREAD=0 switches it off.
// ------------------------------------------------------------------------------------------------------------------
class Workers {
  public:
    // throws std::system_error when a thread cannot be created: the caller (behind the C ABI) catches it
    explicit Workers(int n)
    {
        plan_placement(n);
        for (int i = 0; i < n; ++i)
            threads_.emplace_back([this, i] { run(i); });
    }
    ~Workers()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_)
            t.join();
    }
    int size() const { return (int)threads_.size(); }
    // every worker runs fn(worker index) once; wait() returns when all of them are back
    void start(std::function<void(int)> fn)
    {
        std::lock_guard<std::mutex> g(m_);
        fn_ = std::move(fn);
        running_ = (int)threads_.size();
        ++gen_;
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return running_ == 0; });
    }

  private:
    void plan_placement(int n)
    {
        target_.assign((size_t)n, -1);
        if (const char *env = getenv("KMM_HOST_PACK_SPREAD"))
            if (atoi(env) == 0)
                return;
        CPU_ZERO(&mask_);
        if (sched_getaffinity(0, sizeof mask_, &mask_) != 0)
            return;
        // the mask's CPUs grouped by physical core (package, core id)
        std::vector<std::pair<long, int>> key_cpu;
        for (int c = 0; c < CPU_SETSIZE; ++c) {
            if (!CPU_ISSET(c, &mask_))
                continue;
            long key = c;
            char path[128];
            int core = -1, pkg = 0;
            snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/core_id", c);
            if (FILE *f = fopen(path, "r")) {
                if (fscanf(f, "%d", &core) != 1)
                    core = -1;
                fclose(f);
            }
            snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", c);
            if (FILE *f = fopen(path, "r")) {
                if (fscanf(f, "%d", &pkg) != 1)
                    pkg = 0;
                fclose(f);
            }
            if (core >= 0)
                key = ((long)pkg << 20) | core;
            else
                key = ((long)1 << 40) | c;
            key_cpu.emplace_back(key, c);
        }
        if (key_cpu.size() < 2)
            return;
        std::sort(key_cpu.begin(), key_cpu.end());
        std::vector<std::vector<int>> cores;
        for (size_t i = 0; i < key_cpu.size(); ++i) {
            if (i == 0 || key_cpu[i].first != key_cpu[i - 1].first)
                cores.emplace_back();
            cores.back().push_back(key_cpu[i].second);
        }
        const size_t nc = cores.size();
        const size_t first = (size_t)(((uint64_t)getpid() * 0x9E3779B97F4A7C15ull) >> 33) % nc;
        for (int i = 0; i < n; ++i) {
            // n <= nc: evenly spread cores; more workers than cores: the cores' further hardware threads
            const size_t round = (size_t)i / nc, slot = (size_t)i % nc;
            const size_t step = (size_t)n <= nc ? nc / (size_t)n : 1;
            const std::vector<int> &core = cores[(first + slot * step) % nc];
            target_[(size_t)i] = core[round % core.size()];
        }
        have_mask_ = true;
    }
    void place(int i)
    {
        if (!have_mask_ || target_[(size_t)i] < 0)
            return;
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(target_[(size_t)i], &one);
        if (sched_setaffinity(0, sizeof one, &one) == 0) // (moves the thread there now)
            (void)sched_setaffinity(0, sizeof mask_, &mask_);
    }
    void run(int i)
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [&] { return stop_ || gen_ != seen; });
            if (stop_)
                return;
            seen = gen_;
            g.unlock();
            place(i);
            fn_(i);
            g.lock();
            if (--running_ == 0)
                done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::vector<int> target_;
    cpu_set_t mask_;
    bool have_mask_ = false;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void(int)> fn_;
    uint64_t gen_ = 0;
    int running_ = 0;
    bool stop_ = false;
};

// ------------------------------------------------------------------------------------------------------------------
// Flat reads: src[0, n) -> dst, chunk by chunk (chunk_bases a multiple of 64); the caller copies chunk groups to the device
// as they complete (wait_chunk sleeps: the calling thread must not eat a core of the cgroup's quota by spinning).
// ------------------------------------------------------------------------------------------------------------------
struct FlatJob {
    const uint8_t *src = nullptr;
    uint8_t *dst = nullptr;
    size_t n = 0, chunk = 0, n_chunks = 0;
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    std::vector<std::atomic<uint8_t>> done;
    std::mutex m;
    std::condition_variable cv;

    void prepare(const uint8_t *s, size_t n_bases, uint8_t *d, size_t chunk_bases)
    {
        src = s;
        dst = d;
        n = n_bases;
        chunk = chunk_bases;
        n_chunks = (n + chunk - 1) / chunk;
        next.store(0);
        bad.store(false);
        done = std::vector<std::atomic<uint8_t>>(n_chunks);
        for (auto &f : done)
            f.store(0, std::memory_order_relaxed);
    }
    void run() // one worker
    {
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= n_chunks)
                return;
            const size_t b0 = c * chunk, len = n - b0 < chunk ? n - b0 : chunk;
            if (!bad.load(std::memory_order_relaxed) && !pack2(src + b0, len, dst + b0 / 4))
                bad.store(true);
            {
                std::lock_guard<std::mutex> g(m);
                done[c].store(1, std::memory_order_release);
            }
            cv.notify_all();
        }
    }
    void wait_chunk(size_t c)
    {
        if (done[c].load(std::memory_order_acquire))
            return;
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return done[c].load(std::memory_order_acquire) != 0; });
    }
};

// (round 4's interface, kept for tools and tests: threads of its own, joined by the destructor)
struct Job : FlatJob {
    std::vector<std::thread> threads;
    void start(const uint8_t *s, size_t n_bases, uint8_t *d, size_t chunk_bases, int n_threads)
    {
        prepare(s, n_bases, d, chunk_bases);
        const int nt = (size_t)n_threads < n_chunks ? n_threads : (int)n_chunks;
        for (int t = 0; t < nt; ++t)
            threads.emplace_back([this] { run(); });
    }
    void join()
    {
        for (auto &t : threads)
            t.join();
        threads.clear();
    }
    ~Job() { join(); }
};

// ------------------------------------------------------------------------------------------------------------------
// Raw records -> 2-bit stream + read-start bitset (see the head of the file).
// ------------------------------------------------------------------------------------------------------------------
struct RecordsResult {
    bool ok = false;          // false: a byte without a code / a malformed record line before `consumed` — take the ordinary route
    int64_t consumed = 0;     // bytes of whole records
    int64_t n_records = 0;
    int64_t n_bases = 0;      // flat length of the stream (the bases before `consumed`)
    int64_t uniform_len = 0;  // > 0: every read has this length
};

class RecordsJob {
  public:
    // raw bytes per slice: census and packing meet in the core's caches; one hand-over of (line, flat position) per slice
    // (KMM_HOST_PACK_SLICE_KB: 16 threads of the GPU box's host pack 132 GB/s of FASTQ at 256 KB, 163 at 1 MB:
    // profiles/r05/hostpack_rate.txt)
    static size_t slice_bytes()
    {
        static const size_t v = [] {
            const char *env = getenv("KMM_HOST_PACK_SLICE_KB");
            const long kb = env ? atol(env) : 1024;
            return (size_t)(kb < 4 ? 4 : (kb > 65536 ? 65536 : kb)) << 10;
        }();
        return v;
    }
    // how long a worker spins for the slice before it, in microseconds, before it starts sleeping (KMM_HOST_PACK_SPIN_US)
    static long spin_us()
    {
        static const long v = [] {
            const char *env = getenv("KMM_HOST_PACK_SPIN_US");
            return env ? atol(env) : 300L;
        }();
        return v;
    }

    // codes: >= n / 4 + 512 bytes, 8-byte aligned; start_bits: >= n / 8 + 64 bytes, 4-byte aligned.  Neither needs to be
    // zeroed.  period: 4 (FASTQ) or 2 (two-line FASTA).
    void prepare(const uint8_t *raw, size_t n, int period, uint64_t *codes, uint32_t *start_bits, size_t slice = 0)
    {
        raw_ = raw;
        n_ = n;
        pm_ = (uint32_t)period - 1u;
        hc_ = period == 4 ? (uint8_t)'@' : (uint8_t)'>';
        codes_ = codes;
        bits_ = start_bits;
        SLICE = slice ? (slice < 4096 ? 4096 : slice) : slice_bytes();
        n_slices_ = (n + SLICE - 1) / SLICE;
        desc_ = std::vector<Desc>(n_slices_ + 1);
        desc_[0].line = 0;
        desc_[0].flat = 0;
        desc_[0].ready.store(1, std::memory_order_release);
        next_.store(0);
        watch_ = 0;
        abort_.store(false);
        bad_struct_.store(~0ull);
        bad_base_.store(~0ull);
        n_marks_.store(0);
        n_off_.store(0);
        // the length every read must have for the batch to count as uniform: that of the first record's sequence line
        lguess_ = 0;
        const uint8_t *a = n ? (const uint8_t *)memchr(raw, '\n', n) : nullptr;
        if (a) {
            const uint8_t *b = (const uint8_t *)memchr(a + 1, '\n', n - (size_t)(a + 1 - raw));
            if (b) {
                size_t len = (size_t)(b - (a + 1));
                if (len && b[-1] == '\r')
                    --len;
                lguess_ = len;
            }
        }
    }

    void run() // one worker; returns when no slice is left
    {
        std::vector<uint32_t> nlbuf(SLICE + 1); // (worst case: a slice of newlines)
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= n_slices_)
                return;
            if (abort_.load(std::memory_order_relaxed)) { // (keep the chain alive for whoever waits behind)
                desc_[i + 1].ready.store(1, std::memory_order_release);
                desc_[i].packed.store(1, std::memory_order_release);
                continue;
            }
            const size_t b0 = i * SLICE, b1 = b0 + SLICE < n_ ? b0 + SLICE : n_;
            Census c;
            c.nlpos = nlbuf.data();
            census(raw_ + b0, b1 - b0, c);
            // the line number and the flat position of the slice's first byte, from the slice before it
            // (Spin: the slice before this one is censused at about the same time, and its thread publishes as soon as ITS
            // predecessor has — a ripple of one cache-line hand-over per slice, a microsecond across sockets.  Sleeping early
            // turns every hand-over into a timer wake-up of ~70 us and the whole job runs at that pace: 7 GB/s on 8 threads
            // against 39 on 4, measured.  Only a thread that has waited far longer than any healthy ripple — its predecessor
            // was descheduled, or shares this CPU — gives the CPU away, by sleeping, never by sched_yield: the load balancer
            // does not move threads that run in short slices.)
            if (!desc_[i].ready.load(std::memory_order_acquire)) {
                const auto t_wait = std::chrono::steady_clock::now();
                unsigned spins = 0;
                while (!desc_[i].ready.load(std::memory_order_acquire)) {
                    cpu_relax();
                    if ((++spins & 255u) == 0 &&
                        std::chrono::steady_clock::now() - t_wait > std::chrono::microseconds(spin_us()))
                        std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
            }
            const uint64_t line0 = desc_[i].line, flat0 = desc_[i].flat;
            uint64_t seq = 0;
            for (uint32_t r = 0; r < 4u; ++r)
                if (((line0 + r) & pm_) == 1u)
                    seq += c.by_phase[r];
            const uint64_t flat1 = flat0 + seq;
            // a word two slices share is zeroed by the slice it starts in, before the next slice can know where it is
            if ((flat1 & 31u) && (flat1 & ~(uint64_t)31) >= flat0) {
                codes_[flat1 >> 5] = 0;
                bits_[flat1 >> 5] = 0;
            }
            desc_[i + 1].line = line0 + c.n_nl;
            desc_[i + 1].flat = flat1;
            desc_[i + 1].ready.store(1, std::memory_order_release);
            pack_slice(b0, b1, c.nlpos, c.n_nl, line0, flat0, flat1, c.has_cr);
        }
    }

    // after every worker has returned
    RecordsResult finish()
    {
        RecordsResult r;
        if (abort_.load())
            return r;
        const uint64_t total_lines = desc_[n_slices_].line, flat_total = desc_[n_slices_].flat;
        const uint64_t period = (uint64_t)pm_ + 1u, target = total_lines - total_lines % period;
        // the byte behind newline number `target`: walk back over the (total_lines - target) newlines behind it
        size_t cut = 0;
        if (target) {
            uint64_t skip = total_lines - target;
            size_t p = n_;
            for (;;) {
                while (p > 0 && raw_[p - 1] != '\n')
                    --p;
                // raw_[p - 1] is a newline (p > 0: `target` newlines exist)
                if (skip == 0)
                    break;
                --skip;
                --p;
            }
            cut = p;
        }
        // sequence bytes behind the cut (they were packed too: the line after the cut's header line)
        uint64_t tail_seq = 0;
        {
            size_t p = cut;
            while (p < n_ && raw_[p] != '\n')
                ++p; // header line of the incomplete record
            if (p < n_) {
                ++p;
                while (p < n_ && raw_[p] != '\n') {
                    if (raw_[p] != '\r')
                        ++tail_seq;
                    ++p;
                }
            }
        }
        const uint64_t flat_end = flat_total - tail_seq;
        if (bad_struct_.load() < cut || bad_base_.load() < cut)
            return r; // the ordinary route reports it, with the byte's offset
        // read starts behind the end: counted out, cleared
        uint64_t tail_marks = 0;
        for (uint64_t f = flat_end; f < flat_total; ++f)
            if (bits_[f >> 5] & (1u << (f & 31u))) {
                ++tail_marks;
                bits_[f >> 5] &= ~(1u << (f & 31u));
            }
        // nothing but zeros behind the last base: the rest of its word, and the words pass 1 may load behind it
        {
            const uint64_t w = flat_end >> 5;
            if (flat_end & 31u) {
                codes_[w] &= (1ull << (2u * (flat_end & 31u))) - 1ull;
                bits_[w] &= (1u << (flat_end & 31u)) - 1u;
            } else {
                codes_[w] = 0;
                bits_[w] = 0;
            }
            for (uint64_t j = 1; j <= 40; ++j)
                codes_[w + j] = 0;
            bits_[w + 1] = bits_[w + 2] = 0;
        }
        r.ok = true;
        r.consumed = (int64_t)cut;
        r.n_records = (int64_t)(target / period);
        r.n_bases = (int64_t)flat_end;
        const uint64_t marks = n_marks_.load() - tail_marks;
        if (lguess_ >= 1 && r.n_records > 0 && n_off_.load() == 0 && marks == (uint64_t)r.n_records && tail_marks <= 1 &&
            flat_end == (uint64_t)r.n_records * lguess_)
            r.uniform_len = (int64_t)lguess_;
        return r;
    }

    size_t n_slices() const { return n_slices_; }

    // The calling thread (not a worker): sleeps until the slices [0, upto) are packed and returns the flat position behind
    // them — every 64-bit word of the stream below (that position >> 5) is final and may be copied to the device.
    uint64_t wait_packed_prefix(size_t upto)
    {
        if (upto > n_slices_)
            upto = n_slices_;
        while (watch_ < upto) {
            if (desc_[watch_].packed.load(std::memory_order_acquire))
                ++watch_;
            else
                std::this_thread::sleep_for(std::chrono::microseconds(40));
        }
        return desc_[upto].flat; // (published before slice `upto - 1` was packed)
    }

  private:
    struct alignas(64) Desc {
        std::atomic<uint32_t> ready{0};  // line / flat are known
        std::atomic<uint32_t> packed{0}; // the slice's bases are in the stream
        uint64_t line = 0, flat = 0;     // newlines / sequence bytes before the slice
        Desc() = default;
        Desc(const Desc &o) : ready(o.ready.load()), packed(o.packed.load()), line(o.line), flat(o.flat) {}
    };

    static inline void cpu_relax()
    {
#if defined(__x86_64__)
        _mm_pause();
#endif
    }

    // One pass over the slice: positions of its newlines (nlpos: room for one entry per byte), bytes that are neither '\n'
    // nor '\r' by (newlines before them inside the slice) mod 4, and whether it holds a '\r' at all.
    struct Census {
        uint32_t *nlpos;
        size_t n_nl = 0;
        uint64_t by_phase[4] = {0, 0, 0, 0};
        uint32_t r = 0;
        size_t run = 0;
        bool has_cr = false;
    };

    static inline void census_events(Census &c, size_t base, uint64_t m_nl, uint64_t m_cr)
    {
        if (__builtin_expect(m_cr != 0, 0)) { // (the run's length counts the '\r': taken out here, from the phase it lies in)
            uint64_t m = m_nl | m_cr;
            uint32_t r = c.r;
            while (m) {
                const unsigned b = (unsigned)__builtin_ctzll(m);
                m &= m - 1;
                if ((m_nl >> b) & 1u)
                    r = (r + 1u) & 3u;
                else
                    c.by_phase[r] -= 1;
            }
            c.has_cr = true;
        }
        while (m_nl) {
            const size_t q = base + (unsigned)__builtin_ctzll(m_nl);
            m_nl &= m_nl - 1;
            c.by_phase[c.r] += q - c.run;
            c.run = q + 1;
            c.r = (c.r + 1u) & 3u;
            c.nlpos[c.n_nl++] = (uint32_t)q;
        }
    }

    static void census_scalar(const uint8_t *p, size_t len, size_t from, Census &c)
    {
        for (size_t q = from; q < len; ++q) {
            const uint8_t ch = p[q];
            if (ch == '\n') {
                c.by_phase[c.r] += q - c.run;
                c.run = q + 1;
                c.r = (c.r + 1u) & 3u;
                c.nlpos[c.n_nl++] = (uint32_t)q;
            } else if (ch == '\r') {
                c.by_phase[c.r] -= 1;
                c.has_cr = true;
            }
        }
    }

#if defined(__x86_64__)
    KMM_AVX512_TARGET static size_t census_avx512(const uint8_t *p, size_t len, Census &c)
    {
        const __m512i nl = _mm512_set1_epi8('\n'), cr = _mm512_set1_epi8('\r');
        size_t q = 0;
        for (; q + 64 <= len; q += 64) {
            const __m512i v = _mm512_loadu_si512(p + q);
            const uint64_t m_nl = _mm512_cmpeq_epi8_mask(v, nl), m_cr = _mm512_cmpeq_epi8_mask(v, cr);
            if (m_nl | m_cr)
                census_events(c, q, m_nl, m_cr);
        }
        return q;
    }
    __attribute__((target("avx2"))) static size_t census_avx2(const uint8_t *p, size_t len, Census &c)
    {
        const __m256i nl = _mm256_set1_epi8('\n'), cr = _mm256_set1_epi8('\r');
        size_t q = 0;
        for (; q + 64 <= len; q += 64) {
            const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + q));
            const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + q + 32));
            const uint64_t m_nl = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, nl)) |
                                  ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, nl)) << 32);
            const uint64_t m_cr = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, cr)) |
                                  ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, cr)) << 32);
            if (m_nl | m_cr)
                census_events(c, q, m_nl, m_cr);
        }
        return q;
    }
#endif

    static void census(const uint8_t *p, size_t len, Census &c)
    {
        size_t q = 0;
#if defined(__x86_64__)
        const Isa l = isa();
        if (l == ISA_AVX512)
            q = census_avx512(p, len, c);
        else if (l == ISA_AVX2)
            q = census_avx2(p, len, c);
#endif
        census_scalar(p, len, q, c);
        c.by_phase[c.r] += len - c.run; // the unfinished line at the end of the slice
    }

    // ---- the slice's share of the 2-bit stream: 64-bit words, the first and last of them shared with the neighbours ----
    struct Writer {
        uint64_t *out;         // the word being filled
        uint64_t acc = 0;
        unsigned fill;         // bits of it in use
        uint64_t *shared;      // a word the slice before this one also writes (OR-ed in), or nullptr
        inline void store_word()
        {
            if (out == shared)
                __atomic_fetch_or(out, acc, __ATOMIC_RELAXED);
            else
                *out = acc;
            ++out;
        }
        inline void push(uint64_t w, unsigned nb) // nb in [1, 64]; w has nothing above bit nb
        {
            acc |= w << fill;
            if (fill + nb >= 64u) {
                store_word();
                acc = fill ? w >> (64u - fill) : 0ull;
                fill = fill + nb - 64u;
            } else {
                fill += nb;
            }
        }
        inline void flush() // the last, partial word: zeroed by the slice it starts in, shared with the next slice
        {
            if (fill)
                __atomic_fetch_or(out, acc, __ATOMIC_RELAXED);
        }
    };

    struct PackState {
        Writer w;
        uint64_t flat;     // flat position of the next base
        bool pending;      // the next base starts a read
        uint64_t marks = 0, off = 0, next_mult;
        uint64_t own0 = 0, own1 = 0; // words of the read-start bitset no other slice touches
        bool bad = false;
    };

    inline void mark(PackState &s)
    {
        const uint64_t wi = s.flat >> 5;
        if (wi >= s.own0 && wi < s.own1)
            bits_[wi] |= 1u << (s.flat & 31u);
        else
            __atomic_fetch_or(&bits_[wi], 1u << (s.flat & 31u), __ATOMIC_RELAXED);
        ++s.marks;
        if (lguess_) {
            if (s.flat == s.next_mult) {
                s.next_mult += lguess_;
            } else {
                const uint64_t rem = s.flat % lguess_;
                if (rem)
                    ++s.off;
                s.next_mult = s.flat - rem + lguess_;
            }
        }
        s.pending = false;
    }

    // bases without a '\r' among them
    inline void emit_scalar(PackState &s, const uint8_t *p, size_t len)
    {
        const uint8_t *t = code_tab().t;
        size_t i = 0;
        while (i < len) {
            const size_t m = len - i < 32 ? len - i : 32;
            uint64_t w = 0;
            uint32_t flags = 0;
            for (size_t j = 0; j < m; ++j) {
                const uint32_t c = t[p[i + j]];
                flags |= c;
                w |= (uint64_t)(c & 3u) << (2 * j);
            }
            if (flags & 0x80u)
                s.bad = true;
            s.w.push(w, (unsigned)(2 * m));
            i += m;
        }
    }
#if defined(__x86_64__)
    KMM_AVX512_TARGET inline void emit_avx512(PackState &s, const uint8_t *p, size_t len)
    {
        const __m512i t0 = _mm512_load_si512(tab128().t), t1 = _mm512_load_si512(tab128().t + 64);
        __m512i flags = _mm512_setzero_si512();
        size_t i = 0;
        for (; i + 64 <= len; i += 64) {
            const __m128i o = codes64_avx512(_mm512_loadu_si512(p + i), t0, t1, flags);
            s.w.push((uint64_t)_mm_cvtsi128_si64(o), 64);
            s.w.push((uint64_t)_mm_extract_epi64(o, 1), 64);
        }
        uint64_t bad = _mm512_movepi8_mask(flags);
        if (i < len) {
            const size_t m = len - i;
            const __mmask64 live = ~0ull >> (64 - m);
            __m512i f2 = _mm512_setzero_si512();
            const __m128i o = codes64_avx512(_mm512_maskz_loadu_epi8(live, p + i), t0, t1, f2);
            bad |= _mm512_movepi8_mask(f2) & live;
            if (m <= 32) {
                s.w.push((uint64_t)_mm_cvtsi128_si64(o), (unsigned)(2 * m));
            } else {
                s.w.push((uint64_t)_mm_cvtsi128_si64(o), 64);
                s.w.push((uint64_t)_mm_extract_epi64(o, 1), (unsigned)(2 * (m - 32)));
            }
        }
        if (bad)
            s.bad = true;
    }
#endif
    inline void emit_run(PackState &s, const uint8_t *p, size_t len)
    {
        if (!len)
            return;
        if (s.pending)
            mark(s);
#if defined(__x86_64__)
        if (isa() == ISA_AVX512)
            emit_avx512(s, p, len);
        else
#endif
            emit_scalar(s, p, len);
        s.flat += len;
    }

    // the bytes [p, p + len) of a sequence line (its terminator excluded); ends_line: a '\n' follows them
    inline void emit_seq(PackState &s, const uint8_t *p, size_t len, bool may_have_cr, bool ends_line)
    {
        if (may_have_cr) {
            if (ends_line && len && p[len - 1] == '\r')
                --len; // "\r\n": whatever the '\r' would mark is the next read's first base, marked anyway
            while (len) {
                const uint8_t *c = (const uint8_t *)memchr(p, '\r', len);
                if (!c)
                    break;
                emit_run(s, p, (size_t)(c - p));
                s.pending = true; // no window may span a '\r' (the device parser treats it as a break)
                len -= (size_t)(c - p) + 1;
                p = c + 1;
            }
        }
        emit_run(s, p, len);
    }

    void pack_slice(size_t b0, size_t b1, const uint32_t *nlpos, size_t n_nl, uint64_t line0, uint64_t flat0, uint64_t flat1, bool has_cr)
    {
        PackState s;
        s.w.out = codes_ + (flat0 >> 5);
        s.w.fill = (unsigned)(2u * (flat0 & 31u));
        s.w.shared = s.w.fill ? s.w.out : nullptr;
        s.flat = flat0;
        s.next_mult = lguess_ ? (flat0 + lguess_ - 1) / lguess_ * lguess_ : 0;
        // words of the read-start bitset that begin inside the slice's flat range are this slice's to zero (the last of
        // them, if the next slice shares it, was zeroed before the chain moved on)
        {
            const uint64_t w0 = (flat0 + 31) >> 5, w1 = flat1 >> 5; // whole words [w0, w1)
            if (w1 > w0) {
                memset(bits_ + w0, 0, (size_t)(w1 - w0) * 4);
                s.own0 = w0;
                s.own1 = w1;
            }
        }
        const uint8_t *base = raw_ + b0;
        const size_t len = b1 - b0;
        bool at_line_start = b0 == 0 || raw_[b0 - 1] == '\n';
        // a '\r' as the last byte of the slice before, inside a sequence line that goes on here
        s.pending = !at_line_start && ((line0 & pm_) == 1u) && raw_[b0 - 1] == '\r';
        uint64_t bad_struct = ~0ull;
        size_t pos = 0;
        uint64_t line = line0;
        for (size_t k = 0; k <= n_nl; ++k) {
            const size_t end = k < n_nl ? nlpos[k] : len;
            const uint32_t phase = (uint32_t)line & pm_;
            if (phase == 1u) {
                if (at_line_start)
                    s.pending = true;
                emit_seq(s, base + pos, end - pos, has_cr, k < n_nl);
            } else if (at_line_start && pos < len && !(phase & 1u)) {
                // record structure: a header line starts with '@' / '>', the third line of a FASTQ record with '+'
                const uint8_t want = phase == 0u ? hc_ : (uint8_t)'+';
                if (base[pos] != want && bad_struct == ~0ull)
                    bad_struct = b0 + pos;
            }
            if (k < n_nl) {
                pos = end + 1;
                ++line;
                at_line_start = true;
            }
        }
        s.w.flush();
        desc_[(b0 / SLICE)].packed.store(1, std::memory_order_release);
        if (s.marks)
            n_marks_.fetch_add(s.marks, std::memory_order_relaxed);
        if (s.off)
            n_off_.fetch_add(s.off, std::memory_order_relaxed);
        if (bad_struct != ~0ull)
            atomic_min(bad_struct_, bad_struct);
        if (s.bad)
            atomic_min(bad_base_, (uint64_t)b0); // (somewhere in this slice: the ordinary route finds the byte)
        (void)flat1;
    }

    static void atomic_min(std::atomic<uint64_t> &a, uint64_t v)
    {
        uint64_t cur = a.load(std::memory_order_relaxed);
        while (v < cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {
        }
    }

    const uint8_t *raw_ = nullptr;
    size_t SLICE = (size_t)256 << 10;
    size_t n_ = 0, n_slices_ = 0;
    uint32_t pm_ = 3;
    uint8_t hc_ = '@';
    uint64_t *codes_ = nullptr;
    uint32_t *bits_ = nullptr;
    uint64_t lguess_ = 0;
    std::vector<Desc> desc_;
    std::atomic<size_t> next_{0};
    size_t watch_ = 0;
    std::atomic<bool> abort_{false};
    std::atomic<uint64_t> bad_struct_{~0ull}, bad_base_{~0ull}, n_marks_{0}, n_off_{0};
};

} // namespace kmm_hostpack
