// kmm_hostpack.hpp — part of libkmm (host code only); included by kmm.hip.
// Flat reads that arrive in HOST memory cross the PCIe link as they are: 1.25 bytes per k-mer at ~55 GB/s is 44 G k-mers/s
// per GPU, a quarter of what the kernels map.  With "host_pack_threads" > 0 the call packs them to
// 2 bits per base on the host first — the form pass 1 of the radix path already reads from the records compaction
// (16 codes per 32-bit word, first base lowest) — so that the link carries 0.31 bytes per k-mer.  The reference spends
// its host cores on the same bytes (bnp.as_encoded_array + get_kmers, kmer_mapper/util.py:71-75, in `-t` processes,
// command_line_interface.py:124-130); here they only do the byte -> code step.
//
// The default lookup table only (A C G T a c g t -> 0..3, N n -> 0 as command_line_interface.py:41); a byte outside it
// makes the packer give up and the call takes the ordinary route, where the GPU reports the byte's offset.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include <immintrin.h>

namespace kmm_hostpack {

// n bases -> ceil(n / 4) bytes (the last byte zero-padded); false = a byte that is not a nucleotide
inline bool pack2_scalar(const uint8_t *src, size_t n, uint8_t *dst)
{
    static const struct Tab {
        uint8_t t[256];
        Tab()
        {
            memset(t, 0x80, sizeof t);
            t[(int)'A'] = t[(int)'a'] = 0;
            t[(int)'C'] = t[(int)'c'] = 1;
            t[(int)'G'] = t[(int)'g'] = 2;
            t[(int)'T'] = t[(int)'t'] = 3;
            t[(int)'N'] = t[(int)'n'] = 0;
        }
    } tab;
    uint32_t bad = 0;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const uint32_t a = tab.t[src[i]], b = tab.t[src[i + 1]], c = tab.t[src[i + 2]], d = tab.t[src[i + 3]];
        bad |= a | b | c | d;
        dst[i >> 2] = (uint8_t)(a | (b << 2) | (c << 4) | (d << 6));
    }
    if (i < n) {
        uint32_t v = 0;
        for (size_t j = i; j < n; ++j) {
            const uint32_t a = tab.t[src[j]];
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
#ifndef KMM_HOSTPACK_PREFETCH
#define KMM_HOSTPACK_PREFETCH 1024
#endif
        if (KMM_HOSTPACK_PREFETCH && (i & 63) == 0) // (one core streams faster with its misses requested well ahead)
            _mm_prefetch(reinterpret_cast<const char *>(src + i + KMM_HOSTPACK_PREFETCH), _MM_HINT_NTA);
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
#endif

inline bool pack2(const uint8_t *src, size_t n, uint8_t *dst)
{
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2)
        return pack2_avx2(src, n, dst);
#endif
    return pack2_scalar(src, n, dst);
}

// Packs src[0, n) into dst on n_threads threads, chunk by chunk (chunk_bases a multiple of 64); done[c] is set when chunk c
// is in place, `bad` when a chunk held a byte that is not a nucleotide (the others stop early).
struct Job {
    const uint8_t *src = nullptr;
    uint8_t *dst = nullptr;
    size_t n = 0, chunk = 0, n_chunks = 0;
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    std::vector<std::atomic<uint8_t>> done;
    std::vector<std::thread> threads;

    void start(const uint8_t *s, size_t n_bases, uint8_t *d, size_t chunk_bases, int n_threads)
    {
        src = s;
        dst = d;
        n = n_bases;
        chunk = chunk_bases;
        n_chunks = (n + chunk - 1) / chunk;
        done = std::vector<std::atomic<uint8_t>>(n_chunks);
        for (auto &f : done)
            f.store(0, std::memory_order_relaxed);
        const int nt = (size_t)n_threads < n_chunks ? n_threads : (int)n_chunks;
        for (int t = 0; t < nt; ++t)
            threads.emplace_back([this] { run(); });
    }
    void run()
    {
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= n_chunks)
                return;
            const size_t b0 = c * chunk, len = n - b0 < chunk ? n - b0 : chunk;
            if (!bad.load(std::memory_order_relaxed) && !pack2(src + b0, len, dst + b0 / 4))
                bad.store(true);
            done[c].store(1, std::memory_order_release);
        }
    }
    void wait_chunk(size_t c) const
    {
        while (!done[c].load(std::memory_order_acquire))
            std::this_thread::yield();
    }
    void join()
    {
        for (auto &t : threads)
            t.join();
        threads.clear();
    }
    ~Job() { join(); }
};

} // namespace kmm_hostpack
