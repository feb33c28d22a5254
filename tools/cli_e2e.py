#!/usr/bin/env python3
"""End-to-end `kmer_mapper map` timing on a synthetic FASTQ / 2-line FASTA file (host parse + H2D + GPU)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import synthetic as syn                     # noqa: E402
from kmer_mapper_amd.command_line_interface import run_argument_parser  # noqa: E402


def write_fastq_fast(path, bases, n_reads, L):
    """Vectorised FASTQ writer: @r\\n<seq>\\n+\\n<qual>\\n with fixed-width records."""
    rec = np.empty((n_reads, 3 + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3:3 + L] = bases.reshape(n_reads, L)
    rec[:, 3 + L:3 + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 3 + L + 3:3 + L + 3 + L] = ord("I")
    rec[:, -1] = 10
    rec.tofile(path)


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    out_dir = sys.argv[2] if len(sys.argv) > 2 else "/tmp/kmm_e2e"
    os.makedirs(out_dir, exist_ok=True)
    index, genome = syn.make_index(1_000_000, seed=1)
    idx_path = os.path.join(out_dir, "index.npz")
    index.to_file(idx_path)
    bases, _ = syn.make_reads(genome, n_reads, 150, seed=2)
    fq = os.path.join(out_dir, "reads.fq")
    write_fastq_fast(fq, bases, n_reads, 150)
    size = os.path.getsize(fq)
    run_argument_parser(["map", "-i", idx_path, "-f", fq, "-o", os.path.join(out_dir, "warm"), "-c", "50000000"])
    for chunk in (2_500_000, 10_000_000, 100_000_000):
        t0 = time.perf_counter()
        run_argument_parser(["map", "-i", idx_path, "-f", fq, "-o", os.path.join(out_dir, "out"),
                             "-c", str(chunk)])
        dt = time.perf_counter() - t0
        print("E2E chunk=%d: %.2f s, %.1f MB/s of FASTQ, %.2f M reads/s, %.1f M k-mers/s"
              % (chunk, dt, size / dt / 1e6, n_reads / dt / 1e6, n_reads * 120 / dt / 1e6), flush=True)


def gz_case(out_dir, idx_path, n_reads=1_000_000):
    import gzip
    import shutil
    fq = os.path.join(out_dir, "reads.fq")
    small = os.path.join(out_dir, "small.fq")
    with open(fq, "rb") as f, open(small, "wb") as g:
        g.write(f.read(307 * n_reads))
    gz = small + ".gz"
    t0 = time.perf_counter()
    with open(small, "rb") as f, gzip.open(gz, "wb", compresslevel=1) as g:
        shutil.copyfileobj(f, g, 1 << 24)
    print("gzip -1 of %d MB took %.1f s" % (os.path.getsize(small) >> 20, time.perf_counter() - t0), flush=True)
    for extra in ([], ["--host-parser"]):
        t0 = time.perf_counter()
        run_argument_parser(["map", "-i", idx_path, "-f", gz, "-o", os.path.join(out_dir, "outgz"),
                             "-c", "10000000"] + extra)
        dt = time.perf_counter() - t0
        print("E2E gz %s: %.2f s, %.1f MB/s of inflated FASTQ, %.2f M reads/s"
              % (extra or "gpu-parser", dt, os.path.getsize(small) / dt / 1e6, n_reads / dt / 1e6), flush=True)


if __name__ == "__main__":
    main()
    d = sys.argv[2] if len(sys.argv) > 2 else "/tmp/kmm_e2e"
    gz_case(d, os.path.join(d, "index.npz"))
