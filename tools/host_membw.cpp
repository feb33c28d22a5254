// What bounds the host packer per core?  g++ -O3 -std=c++17 -pthread -march=native -I kmer_mapper_amd/csrc tools/host_membw.cpp -o /tmp/host_membw
//   /tmp/host_membw [threads=16] [MB per thread=192] [NUMA node to bind the process's CPUs to, -1 = none]
// Per thread, a private slice of one large source: (a) a pure AVX-512 read of it, (b) pack2 of it into a private
// destination, (c) pack2 of a 256 KB piece over and over (no DRAM), each with the source on 4 KiB pages and on
// transparent huge pages (madvise).  Prints GB/s of source bytes, all threads together.
#include "kmm_hostpack.hpp"
#ifdef WITH_HIP   // hipcc -DWITH_HIP: the FlatJob lines also with page-locked (hipHostMalloc) source / destination
#include <hip/hip_runtime_api.h>
#endif

#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <sys/mman.h>

using Clock = std::chrono::steady_clock;

__attribute__((target("avx512f,avx512bw"))) static uint64_t read_only(const uint8_t *p, size_t n)
{
    __m512i a = _mm512_setzero_si512(), b = a, c = a, d = a;
    for (size_t i = 0; i + 256 <= n; i += 256) {
        a = _mm512_add_epi64(a, _mm512_loadu_si512(p + i));
        b = _mm512_add_epi64(b, _mm512_loadu_si512(p + i + 64));
        c = _mm512_add_epi64(c, _mm512_loadu_si512(p + i + 128));
        d = _mm512_add_epi64(d, _mm512_loadu_si512(p + i + 192));
    }
    a = _mm512_add_epi64(_mm512_add_epi64(a, b), _mm512_add_epi64(c, d));
    return (uint64_t)_mm512_reduce_add_epi64(a);
}

static uint8_t *big(size_t n, bool huge)
{
    void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED)
        exit(2);
    if (huge)
        madvise(p, n, MADV_HUGEPAGE);
    return static_cast<uint8_t *>(p);
}

static void bind_to_node(int node)
{
    char path[96], text[4096] = {0};
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f || !fgets(text, sizeof text, f)) {
        printf("node %d: no cpulist\n", node);
        return;
    }
    fclose(f);
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    for (char *p = text; *p && *p != '\n';) {
        const long a = strtol(p, &p, 10);
        long b = a;
        if (*p == '-')
            b = strtol(p + 1, &p, 10);
        for (long c = a; c <= b; ++c, ++n)
            CPU_SET((int)c, &set);
        if (*p == ',')
            ++p;
    }
    printf("process bound to the %d CPUs of NUMA node %d: %s\n", n, node, sched_setaffinity(0, sizeof set, &set) == 0 ? "ok" : "FAILED");
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 16;
    const size_t per = (size_t)(argc > 2 ? atoi(argv[2]) : 192) << 20;
    if (argc > 3 && atoi(argv[3]) >= 0)
        bind_to_node(atoi(argv[3]));
    kmm_hostpack::Workers pool(T);
    std::atomic<uint64_t> sink{0};
    for (int huge = 0; huge < 2; ++huge) {
        uint8_t *src = big(per * (size_t)T, huge), *dst = big(per * (size_t)T / 4 + 4096, huge);
        // first touch by the worker that will read it
        pool.start([&](int w) {
            uint8_t *s = src + per * (size_t)w;
            for (size_t i = 0; i < per; ++i)
                s[i] = "ACGT"[(i * 2654435761u >> 13) & 3];
            memset(dst + per / 4 * (size_t)w, 0, per / 4);
        });
        pool.wait();
        for (int mode = 0; mode < 3; ++mode) {
            double best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                const auto t0 = Clock::now();
                pool.start([&](int w) {
                    const uint8_t *s = src + per * (size_t)w;
                    uint8_t *d = dst + per / 4 * (size_t)w;
                    if (mode == 0)
                        sink += read_only(s, per);
                    else if (mode == 1)
                        sink += kmm_hostpack::pack2(s, per, d);
                    else
                        for (size_t done = 0; done < per; done += 256 << 10)
                            sink += kmm_hostpack::pack2(s, 256 << 10, d);
                });
                pool.wait();
                best = std::min(best, std::chrono::duration<double>(Clock::now() - t0).count());
            }
            printf("%-18s %-32s %2d threads: %7.1f GB/s  (%5.1f per thread)\n", huge ? "huge pages (THP)" : "4 KiB pages",
                   mode == 0 ? "read only" : mode == 1 ? "pack2, source in DRAM" : "pack2, 256 KB piece (cache)", T,
                   (double)per * T / best / 1e9, (double)per / best / 1e9);
            fflush(stdout);
        }
        munmap(src, per * (size_t)T);
        munmap(dst, per * (size_t)T / 4 + 4096);
        // the product's job: ONE array written by the calling thread (first touch: its node), 4 MiB tasks handed out by a counter
        const size_t n = per * (size_t)T;
        src = big(n, huge);
        dst = big(n / 4 + 4096, huge);
        for (size_t i = 0; i < n; ++i)
            src[i] = "ACGT"[(i * 2654435761u >> 13) & 3];
        memset(dst, 0, n / 4 + 4096);
        auto flat = [&](const uint8_t *from, uint8_t *to, const char *what) {
            double best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                const auto t0 = Clock::now();
                kmm_hostpack::FlatJob job;
                job.prepare(from, n, to, (size_t)4 << 20);
                pool.start([&job](int) { job.run(); });
                pool.wait();
                best = std::min(best, std::chrono::duration<double>(Clock::now() - t0).count());
            }
            printf("%-18s %-32s %2d threads: %7.1f GB/s  (%5.1f per thread)\n", huge ? "huge pages (THP)" : "4 KiB pages", what, T,
                   (double)n / best / 1e9, (double)n / T / best / 1e9);
            fflush(stdout);
        };
        flat(src, dst, "FlatJob, one array (caller's)");
#ifdef WITH_HIP
        if (!huge) {
            uint8_t *psrc = nullptr, *pdst = nullptr;
            if (hipHostMalloc(reinterpret_cast<void **>(&psrc), n, hipHostMallocDefault) == hipSuccess &&
                hipHostMalloc(reinterpret_cast<void **>(&pdst), 2 * (n / 4) + 4096, hipHostMallocDefault) == hipSuccess) {
                memcpy(psrc, src, n);
                memset(pdst, 0, 2 * (n / 4) + 4096);
                flat(src, pdst, "FlatJob -> page-locked");
                flat(psrc, dst, "FlatJob, page-locked ->");
                flat(psrc, pdst, "FlatJob, page-locked both");
                // the same while a copy engine reads page-locked memory at PCIe rate: another buffer, then the destination itself
                uint8_t *other = nullptr, *dev = nullptr;
                hipStream_t st;
                hipEvent_t ev;
                (void)hipEventCreateWithFlags(&ev, hipEventBlockingSync);
                const size_t other_bytes = n / 4; // as large as the destination, walked through like it
                if (hipHostMalloc(reinterpret_cast<void **>(&other), other_bytes, hipHostMallocDefault) == hipSuccess &&
                    hipMalloc(reinterpret_cast<void **>(&dev), (size_t)256 << 20) == hipSuccess && hipStreamCreate(&st) == hipSuccess) {
                    memset(other, 1, other_bytes);
                    for (int which = 0; which < 3; ++which) {
                        std::atomic<bool> stop{false};
                        std::atomic<uint64_t> copied{0};
                        const auto t0 = Clock::now();
                        std::thread dma([&] {
                            size_t at = 0;
                            while (!stop.load()) {
                                for (int q = 0; q < 8; ++q) {
                                    const size_t walk = at % ((n / 4) - ((size_t)8 << 20));
                                    const uint8_t *from = which == 0 ? other + walk : which == 1 ? pdst + walk : pdst + n / 4 + walk;
                                    (void)hipMemcpyAsync(dev + ((size_t)q << 23), from, (size_t)8 << 20, hipMemcpyHostToDevice, st);
                                    at += (size_t)8 << 20;
                                }
                                (void)hipEventRecord(ev, st); // (a blocking wait: the waiting thread sleeps, it does not spin)
                                (void)hipEventSynchronize(ev);
                                copied += (uint64_t)64 << 20;
                            }
                        });
                        flat(psrc, pdst, which == 0 ? "... while another buffer is copied" : which == 1 ? "... while the destination is copied"
                                                                                     : "... the same allocation's other half");
                        stop = true;
                        dma.join();
                        printf("    (the copy engine moved %.1f GB/s meanwhile)\n",
                               (double)copied.load() / std::chrono::duration<double>(Clock::now() - t0).count() / 1e9);
                    }
                }
                (void)hipHostFree(other);
                (void)hipFree(dev);
            }
            (void)hipHostFree(psrc);
            (void)hipHostFree(pdst);
        }
#endif
        munmap(src, n);
        munmap(dst, n / 4 + 4096);
    }
    return (int)(sink.load() & 1) * 0;
}
