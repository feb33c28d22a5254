// ThreadSanitizer run of libkmm_io's readers (csrc/kmm_io.cpp + kmm_inflate.hpp; tests/test_reads_io.py builds this):
// a plain gzip member inflated on several threads (speculative chunks from block boundaries found by search, markers resolved
// in order), many small members one behind the other (decoded one chunk at a time), a BGZF file (members in parallel), read
// back in odd-sized pieces and compared with what went in.
#include "kmm_io.h"
#include "../kmer_mapper_amd/csrc/kmm_io.cpp"

#include <random>
#include <string>

static std::string gz(const std::string &in, int level, bool bgzf_extra)
{
    z_stream z{};
    if (deflateInit2(&z, level, Z_DEFLATED, bgzf_extra ? -15 : 31, 8, Z_DEFAULT_STRATEGY) != Z_OK)
        exit(3);
    std::string out(deflateBound(&z, (uLong)in.size()) + 64, '\0');
    z.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(in.data()));
    z.avail_in = (uInt)in.size();
    z.next_out = reinterpret_cast<Bytef *>(&out[0]);
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END)
        exit(3);
    out.resize(z.total_out);
    deflateEnd(&z);
    if (!bgzf_extra)
        return out;
    std::string m("\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00\x42\x43\x02\x00", 16);
    const uint32_t bsize = (uint32_t)(18 + out.size() + 8 - 1), crc = (uint32_t)crc32(0, reinterpret_cast<const Bytef *>(in.data()), (uInt)in.size()),
                   isize = (uint32_t)in.size();
    m += (char)(bsize & 255);
    m += (char)(bsize >> 8);
    m += out;
    for (int i = 0; i < 4; ++i)
        m += (char)(crc >> (8 * i));
    for (int i = 0; i < 4; ++i)
        m += (char)(isize >> (8 * i));
    return m;
}

static int check(const char *what, const std::string &file, const std::string &expect, int threads)
{
    const char *path = "/tmp/kmm_io_tsan.gz";
    FILE *f = fopen(path, "wb");
    fwrite(file.data(), 1, file.size(), f);
    fclose(f);
    kmm_io_t *h = kmm_io_open(path, threads);
    if (!h) {
        fprintf(stderr, "%s: open failed: %s\n", what, kmm_io_error());
        return 1;
    }
    std::string got;
    std::vector<uint8_t> buf(1234567);
    for (;;) {
        const int64_t n = kmm_io_read(h, buf.data(), (int64_t)buf.size());
        if (n < 0) {
            fprintf(stderr, "%s: read failed: %s\n", what, kmm_io_error());
            return 1;
        }
        if (n == 0)
            break;
        got.append(reinterpret_cast<const char *>(buf.data()), (size_t)n);
    }
    kmm_io_close(h);
    if (got != expect) {
        fprintf(stderr, "%s: %zu bytes read, %zu expected, or different bytes\n", what, got.size(), expect.size());
        return 1;
    }
    printf("%s: %zu bytes, same\n", what, got.size());
    return 0;
}

int main()
{
    std::mt19937_64 rng(4711);
    std::string text;
    for (int i = 0; i < 20000; ++i) {
        text += "@r" + std::to_string(i) + "\n";
        std::string s, q;
        for (int j = 0; j < 150; ++j) {
            s += "ACGT"[rng() % 4];
            q += "FFFFFF:,#"[rng() % 9];
        }
        text += s + "\n+\n" + q + "\n";
    }
    setenv("KMM_IO_GZIP_CHUNK", "65536", 1); // many chunks per wave in a small file
    int rc = 0;
    rc |= check("one gzip member, 4 threads", gz(text, 6, false), text, 4);
    std::string small, bg;
    for (size_t p = 0; p < text.size(); p += 40000)
        small += gz(text.substr(p, 40000), 6, false);
    rc |= check("many small gzip members, 4 threads", small, text, 4);
    for (size_t p = 0; p < text.size(); p += 65280)
        bg += gz(text.substr(p, 65280), 6, true);
    bg += std::string("\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00\x42\x43\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00", 28);
    rc |= check("BGZF, 4 threads", bg, text, 4);
    return rc;
}
