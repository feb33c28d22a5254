// ThreadSanitizer / AddressSanitizer run of the host packer (csrc/kmm_hostpack.hpp; tests/test_host.py builds this twice):
// raw FASTQ records of random read lengths packed by 1 and by several threads — slices handed over by the chained prefix, the
// words two slices share OR-ed in atomically — must give the same 2-bit stream, read-start bitset and counts, with no data
// race and no access outside the buffers.
#include "kmm_hostpack.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

struct Out {
    std::vector<uint64_t> codes;
    std::vector<uint32_t> bits;
    kmm_hostpack::RecordsResult r;
};

static Out pack(kmm_hostpack::Workers &pool, const std::string &raw, size_t slice)
{
    Out o;
    o.codes.assign(raw.size() / 32 + 80, 0xAAAAAAAAAAAAAAAAull); // (garbage: the packer must not rely on zeroed buffers)
    o.bits.assign(raw.size() / 32 + 20, 0x55555555u);
    kmm_hostpack::RecordsJob job;
    job.prepare(reinterpret_cast<const uint8_t *>(raw.data()), raw.size(), 4, o.codes.data(), o.bits.data(), slice);
    pool.start([&job](int) { job.run(); });
    pool.wait();
    o.r = job.finish();
    return o;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    std::mt19937_64 rng(777);
    kmm_hostpack::Workers one(1), many(6);
    for (int r = 0; r < rounds; ++r) {
        std::string raw;
        const int n_rec = 200 + (int)(rng() % 3000);
        const bool uniform = r % 3 == 0;
        for (int i = 0; i < n_rec; ++i) {
            const int len = uniform ? 150 : 1 + (int)(rng() % 400);
            raw += "@r" + std::to_string(i) + (r % 5 == 4 ? "\r\n" : "\n");
            for (int j = 0; j < len; ++j)
                raw += "ACGTacgtN"[rng() % 9];
            raw += r % 5 == 4 ? "\r\n+\r\n" : "\n+\n";
            for (int j = 0; j < len; ++j)
                raw += "@+FI#:,"[rng() % 7];
            raw += r % 5 == 4 ? "\r\n" : "\n";
        }
        if (r % 4 == 1)
            raw += "@cut\nACG"; // an unfinished record at the end
        const size_t slice = (size_t)1 << (10 + rng() % 6); // 1 KiB .. 32 KiB: hundreds of hand-overs
        const Out a = pack(one, raw, slice), b = pack(many, raw, slice);
        if (!a.r.ok || !b.r.ok || a.r.consumed != b.r.consumed || a.r.n_records != b.r.n_records || a.r.n_bases != b.r.n_bases ||
            a.r.uniform_len != b.r.uniform_len) {
            fprintf(stderr, "round %d: results differ (%d %d, %lld %lld)\n", r, (int)a.r.ok, (int)b.r.ok, (long long)a.r.consumed, (long long)b.r.consumed);
            return 1;
        }
        const size_t words = ((size_t)a.r.n_bases + 31) / 32;
        for (size_t w = 0; w < words; ++w) {
            const bool last = w + 1 == words && (a.r.n_bases % 32);
            const uint64_t mask = last ? (~0ull >> (64 - 2 * (a.r.n_bases % 32))) : ~0ull;
            if ((a.codes[w] ^ b.codes[w]) & mask) {
                fprintf(stderr, "round %d: code word %zu differs\n", r, w);
                return 1;
            }
        }
        if (!a.r.uniform_len)
            for (size_t w = 0; w < ((size_t)a.r.n_bases + 31) / 32; ++w) {
                const bool last = w + 1 == ((size_t)a.r.n_bases + 31) / 32 && (a.r.n_bases % 32);
                const uint32_t mask = last ? (~0u >> (32 - a.r.n_bases % 32)) : ~0u;
                if ((a.bits[w] ^ b.bits[w]) & mask) {
                    fprintf(stderr, "round %d: read-start word %zu differs\n", r, w);
                    return 1;
                }
            }
    }
    printf("%d rounds: one thread and six threads agree\n", rounds);
    return 0;
}
