/*
 * kmm_io.h — C ABI of libkmm_io.so: host-side read-file input for the GPU mapper (no HIP inside).
 *
 * Replaces the chunk reader in front of the mapper, `bnp.open(args.reads).read_chunks(...)`
 * (reference kmer_mapper/command_line_interface.py:102-103,109-111; ".fa, .fq, .fa.gz, or fq.gz", Readme.md:11)
 * and the igzip reader the reference meant to use (kmer_mapper/util.py:78-101): file bytes go straight into the
 * caller's (pinned) staging buffer — BGZF members inflated in parallel at their final place, a plain gzip stream
 * by a read-ahead thread, an uncompressed file by parallel pread.  Errors: NULL / -1 and kmm_io_error().
 */
#ifndef KMM_IO_H
#define KMM_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kmm_io kmm_io_t;

/* Opens a read file; the first bytes decide: BGZF (1), gzip stream (2), else a plain file (0).  n_threads:
 * workers for BGZF members / pread slices (a gzip stream always uses one read-ahead thread). */
kmm_io_t *kmm_io_open(const char *path, int n_threads);
int kmm_io_kind(const kmm_io_t *h);
/* Up to n bytes of the (inflated) stream into dst; returns the count, 0 at the end of the stream, -1 on error
 * (truncated file, corrupt member: CRC32 / ISIZE are checked).  May return fewer than n bytes before the end. */
int64_t kmm_io_read(kmm_io_t *h, uint8_t *dst, int64_t n);
/* Plain files only: continue reading at byte `pos` (a rank's byte range). */
int kmm_io_seek(kmm_io_t *h, int64_t pos);
void kmm_io_close(kmm_io_t *h);
/* 1: libdeflate inflates (found at run time), 0: zlib. */
int kmm_io_engine(void);
const char *kmm_io_error(void);

#ifdef __cplusplus
}
#endif
#endif
