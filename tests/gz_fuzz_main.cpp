// Sanitizer run of the GPU inflater's decoder on the CPU (tests/test_host.py builds this with -fsanitize=address,undefined):
// the same source that runs one lane per member on the GPU (csrc/kmm_gpu_inflate.hpp) inflates deflate streams made by
// zlib — intact ones must come out byte for byte, damaged ones (bytes flipped, truncated, lengthened) must end in an error
// code or in the right bytes, and in no case may the decoder touch memory outside its buffers: after an impossible symbol a
// lane decodes on until the end of its round, on whatever the bits say.
#include "kmm_gpu_inflate.hpp"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <zlib.h>

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &in, int level, int strategy)
{
    z_stream z{};
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy) != Z_OK)
        exit(3);
    std::vector<uint8_t> out(deflateBound(&z, (uLong)in.size()) + 64);
    z.next_in = const_cast<Bytef *>(in.data());
    z.avail_in = (uInt)in.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END)
        exit(3);
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    std::mt19937_64 rng(12345);
    long ok = 0, refused = 0, wrong = 0;
    // the CRC tables as the library makes them (kmm.hip): slicing tables + x^(8 * 2^k)
    std::vector<uint32_t> crcT(kmm_gz::CRC_TABLE_WORDS);
    for (int k = 0; k < 8; ++k)
        for (uint32_t b = 0; b < 256u; ++b)
            crcT[(size_t)k * 256 + b] = kmm_gz::crc_table_entry(k, b);
    for (int k = 0; k < kmm_gz::CRC_SHIFT_WORDS; ++k)
        crcT[8 * 256 + k] = kmm_gz::crc_shift_table_entry(k);
    for (int r = 0; r < rounds; ++r) {
        // FASTQ-like text, runs, noise: every block type and long / overlapping matches
        std::vector<uint8_t> data;
        const size_t n = 1 + rng() % 65000;
        const int kind = (int)(rng() % 4);
        for (size_t i = 0; i < n; ++i) {
            uint8_t c;
            if (kind == 0)
                c = "ACGTACGTN\n@+FFFF:,#"[rng() % 19];
            else if (kind == 1)
                c = (uint8_t)("AB"[(i / (1 + r % 300)) & 1]);
            else if (kind == 2)
                c = (uint8_t)rng();
            else
                c = (uint8_t)("FFFFFFFFFFFFFFFF:"[rng() % 17]);
            data.push_back(c);
        }
        // the CRC the way k_crc_bgzf computes it: four parts, each moved forward by the bytes behind it, against zlib's
        {
            const size_t n_crc = r % 7 == 0 ? (size_t)(r % 67) : n; // (short messages too: empty parts)
            std::vector<uint8_t> msg(data.begin(), data.begin() + (long)n_crc); // exact size: no byte read behind a part
            uint32_t reg = 0;
            for (uint32_t j = 0; j < kmm_gz::CRC_PARTS; ++j)
                reg ^= kmm_gz::crc_part_share(crcT.data(), crcT.data() + 8 * 256, msg.data(), (uint32_t)n_crc, j);
            const uint32_t want = (uint32_t)crc32(0L, msg.data(), (uInt)n_crc);
            if (~reg != want || kmm_gz::crc32_sliced(crcT.data(), msg.data(), (uint32_t)n_crc) != want) {
                fprintf(stderr, "round %d: CRC of %zu bytes by parts %08x, zlib %08x\n", r, n_crc, ~reg, want);
                return 1;
            }
        }
        const int levels[4] = {0, 1, 6, 9}, strategies[4] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_RLE, Z_HUFFMAN_ONLY};
        std::vector<uint8_t> comp = deflate_raw(data, levels[rng() % 4], strategies[rng() % 4]);
        // exact-size heap buffers: one byte too far and the sanitizer says so (the decoder's contract: 16 readable bytes of slack
        // behind the output, none behind the input)
        for (int variant = 0; variant < 6; ++variant) {
            std::vector<uint8_t> c2 = comp;
            if (variant == 1 && !c2.empty())
                c2[rng() % c2.size()] ^= (uint8_t)(1u << (rng() % 8));
            else if (variant == 2 && c2.size() > 2)
                c2.resize(rng() % c2.size());
            else if (variant == 3)
                for (int k = 0; k < 8 && !c2.empty(); ++k)
                    c2[rng() % c2.size()] = (uint8_t)rng();
            else if (variant == 4)
                c2.insert(c2.begin() + (long)(rng() % (c2.size() + 1)), (uint8_t)rng());
            else if (variant == 5 && c2.size() > 8)
                for (size_t k = c2.size() / 2; k < c2.size(); ++k)
                    c2[k] = 0;
            const size_t claimed = variant == 0 ? data.size() : (rng() % 3 ? data.size() : rng() % 70000);
            std::vector<uint8_t> in(c2);                  // exactly c2.size() bytes on the heap
            std::vector<uint8_t> out(claimed + 16);
            std::vector<uint16_t> prim(kmm_gz::PRIM_WORDS), sec(kmm_gz::SEC_WORDS);
            std::vector<uint64_t> list(kmm_gz::LIST_ALLOC);
            const int rc = kmm_gz::inflate_stream(in.data(), (uint32_t)in.size(), out.data(), (uint32_t)claimed, prim.data(), sec.data(), list.data());
            if (rc == kmm_gz::OK) {
                const bool same = claimed == data.size() && std::equal(data.begin(), data.end(), out.begin());
                if (variant == 0 && !same) {
                    fprintf(stderr, "round %d: an intact stream came out wrong\n", r);
                    return 1;
                }
                same ? ++ok : ++wrong; // (a damaged stream may still be a valid one for other bytes: the CRC catches that on the GPU)
            } else {
                if (variant == 0) {
                    fprintf(stderr, "round %d: an intact stream was refused (%d)\n", r, rc);
                    return 1;
                }
                ++refused;
            }
        }
    }
    printf("%d rounds: %ld right, %ld refused, %ld valid streams for other bytes\n", rounds, ok, refused, wrong);
    return 0;
}
