#!/usr/bin/env python3
"""Inflate rate of .gz read files on the host cores: BGZF members on 1 / 4 / 16 threads
and ONE plain gzip member on 1 thread (zlib) and on 4 .. 32 threads (speculative chunks with markers,
csrc/kmm_inflate.hpp), then `kmer_mapper map` end to end on both files.
    python tools/gz_inflate_rate.py [n_reads] [out_dir]"""
import gzip
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import gz_io, synthetic as syn                              # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    d = sys.argv[2] if len(sys.argv) > 2 else "/tmp/kmm_gz"
    os.makedirs(d, exist_ok=True)
    L = 150
    index, genome = syn.make_index(1_000_000, seed=1)
    bases, _ = syn.make_reads(genome, n_reads, L, seed=2)
    rec = np.empty((n_reads, 6 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0:6] = np.frombuffer(b"@read\n", dtype=np.uint8)
    rec[:, 6:6 + L] = bases.reshape(n_reads, L)
    rec[:, 6 + L:9 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 9 + L:9 + 2 * L] = np.random.default_rng(3).choice(np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8),
                                                              size=(n_reads, L))
    rec[:, -1] = 10
    data = rec.tobytes()
    del rec
    pb, pg = os.path.join(d, "reads.bgzf.fq.gz"), os.path.join(d, "reads.plain.fq.gz")
    t = time.perf_counter()
    gz_io.write_bgzf(pb, data, level=1)
    with gzip.open(pg, "wb", compresslevel=1) as f:
        f.write(data)
    print("FASTQ %.1f MB -> bgzf %.1f MB, gzip %.1f MB (written in %.1f s)"
          % (len(data) / 1e6, os.path.getsize(pb) / 1e6, os.path.getsize(pg) / 1e6, time.perf_counter() - t), flush=True)
    buf = bytearray(len(data))
    for label, path, nt in (("plain gzip, 1 thread", pg, 1), ("plain gzip, 4 threads", pg, 4), ("plain gzip, 8 threads", pg, 8),
                            ("plain gzip, 16 threads", pg, 16), ("plain gzip, 32 threads", pg, 32),
                            ("bgzf, 1 thread", pb, 1), ("bgzf, 4 threads", pb, 4),
                            ("bgzf, 8 threads", pb, 8), ("bgzf, 16 threads", pb, 16)):
        t = time.perf_counter()
        with gz_io.open_gz(path, nt) as s:
            n = s.readinto(buf)
        dt = time.perf_counter() - t
        assert n == len(data)
        print("%-22s %.2f GB/s of inflated FASTQ" % (label, n / dt / 1e9), flush=True)
    assert bytes(buf) == data
    if "--no-cli" not in sys.argv:
        from kmer_mapper_amd.command_line_interface import run_argument_parser
        idx = os.path.join(d, "index.npz")
        index.to_file(idx)
        for path in (pb, pg):
            run_argument_parser(["map", "-i", idx, "-f", path, "-o", os.path.join(d, "warm"), "-c", "50000000"])
            t = time.perf_counter()
            run_argument_parser(["map", "-i", idx, "-f", path, "-o", os.path.join(d, "out"), "-c", "50000000"])
            dt = time.perf_counter() - t
            print("kmer_mapper map on %s: %.2f s = %.2f GB/s of inflated FASTQ, %.1f M reads/s"
                  % (os.path.basename(path), dt, len(data) / dt / 1e9, n_reads / dt / 1e6), flush=True)


if __name__ == "__main__":
    main()
