#!/usr/bin/env python3
"""`kmer_mapper map` on a BGZF-compressed FASTQ, members inflated on the GPU (kmm_map_bgzf) against the host inflater.
    python tools/bgzf_e2e.py [n_reads=10000000] [n_index=100000000] [out_dir=/tmp/kmm_bgzf]
FASTQ with read-name-like headers and a skewed 10-letter quality alphabet (compresses ~3.5x at bgzip's default level), written
as BGZF members of 0xFF00 bytes by a process pool.  Prints GB/s of FASTQ and k-mers/s for the library loop, the CLI with
the GPU inflater and the CLI with the host inflater; the three count vectors must be equal."""
import argparse
import multiprocessing as mp
import os
import struct
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import synthetic as syn                     # noqa: E402

_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _member(chunk):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = c.compress(chunk) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload +
            struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def _compress_range(args):
    path, lo, hi = args
    with open(path, "rb") as f:
        f.seek(lo)
        data = f.read(hi - lo)
    return b"".join(_member(data[p:p + 0xFF00]) for p in range(0, len(data), 0xFF00))


def make_fastq(path, bases, n_reads, L, seed=9):
    rng = np.random.default_rng(seed)
    hdr = b"@SRR0000001."
    W = len(hdr) + 9 + 1
    rec = np.empty((n_reads, W + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :len(hdr)] = np.frombuffer(hdr, dtype=np.uint8)
    idx = np.arange(n_reads, dtype=np.int64)
    for d in range(9):
        rec[:, len(hdr) + 8 - d] = (idx // 10 ** d % 10 + 48).astype(np.uint8)
    rec[:, W - 1] = 10
    rec[:, W:W + L] = bases.reshape(n_reads, L)
    rec[:, W + L:W + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    # quality strings: by default a skewed 4-letter alphabet (what binned NovaSeq qualities look like); KMM_E2E_QUAL=full41
    # draws from all 41 Phred+33 values with a distribution that falls off from 'I' (older instruments: longer Huffman codes,
    # more literals), KMM_E2E_QUAL=runs makes long runs of one value
    mode = os.environ.get("KMM_E2E_QUAL", "")
    if mode == "full41":
        p41 = 0.85 ** np.arange(41)
        q = (73 - rng.choice(41, size=(n_reads, L), p=p41 / p41.sum())).astype(np.uint8)
    elif mode == "runs":
        q = np.repeat(rng.choice(np.frombuffer(b"FFFFFF:,", dtype=np.uint8), size=(n_reads, L // 10)), 10, axis=1)
    else:
        q = rng.choice(np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8), size=(n_reads, L))
    rec[:, W + L + 3:W + L + 3 + L] = q
    rec[:, -1] = 10
    rec.tofile(path)
    return rec.shape[1]


def main():
    import logging
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format='%(asctime)s %(levelname)s: %(message)s')
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    n_index = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    out_dir = sys.argv[3] if len(sys.argv) > 3 else "/tmp/kmm_bgzf"
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    index, genome = syn.make_index(n_index, seed=1, gpu_builder=True)
    bases, offs = syn.make_reads(genome, n_reads, 150, seed=2)
    fq = os.path.join(out_dir, "reads.fq")
    rec_len = make_fastq(fq, bases, n_reads, 150)
    size = os.path.getsize(fq)
    gz = fq + ".gz"
    step = 0xFF00 * 256
    with mp.Pool(min(16, os.cpu_count() or 1)) as pool, open(gz, "wb") as g:
        for piece in pool.imap(_compress_range, [(fq, lo, min(lo + step, size)) for lo in range(0, size, step)]):
            g.write(piece)
        g.write(_EOF)
    csize = os.path.getsize(gz)
    print("setup %.1f s: %d reads, FASTQ %.2f GB -> BGZF %.2f GB (ratio %.2f), %d-entry index"
          % (time.time() - t0, n_reads, size / 1e9, csize / 1e9, size / csize, len(index._kmers)), flush=True)
    from kmer_mapper_amd import _lib
    from kmer_mapper_amd.engine import DeviceIndex
    comp = np.memmap(gz, dtype=np.uint8, mode="r")
    mx = index.max_node_id()
    with DeviceIndex.from_index(index, mx) as dev:
        for window in (448 << 20, 224 << 20, 112 << 20):
            for rep in range(2):
                dev.reset()
                t = time.perf_counter()
                pos, recs = 0, 0
                while pos < csize:
                    end = min(pos + window, csize)
                    used, n = dev.map_bgzf(comp[pos:end], fmt=_lib.FORMAT_FASTQ, k=31, first=pos == 0, last=end == csize)
                    pos += used
                    recs += n
                lib_counts = dev.get_node_counts()
                dt = time.perf_counter() - t
            assert recs == n_reads
            print("library loop, compressed windows of %d MB: %.3f s, %.1f GB/s of FASTQ, %.1f G k-mers/s"
                  % (window >> 20, dt, size / dt / 1e9, n_reads * 120 / dt / 1e9), flush=True)
    from kmer_mapper_amd.command_line_interface import map_bnp

    def cli():
        ns = argparse.Namespace(kmer_index=index, index_bundle=None, reads=gz, kmer_size=31, n_threads=16, chunk_size=2_500_000,
                                output_file=None, debug=None, max_hits_per_kmer=1000, gpu=True, gpu_hash_map_size=0,
                                map_reverse_complements=False, apply_max_hits_per_kmer=False, host_parser=False, device=0)
        t = time.perf_counter()
        c = map_bnp(ns)
        return c, time.perf_counter() - t

    time.sleep(4) # (see below: the library loop's handle has just been closed)
    cli()
    time.sleep(4)
    got, dt = cli()
    print("CLI, members inflated on the GPU: %.2f s end to end, %.1f GB/s of FASTQ, %.1f G k-mers/s" % (dt, size / dt / 1e9, n_reads * 120 / dt / 1e9), flush=True)
    os.environ["KMM_CLI_NO_GPU_INFLATE"] = "1"
    # (a handle that has just been closed leaves the driver ~25 GB of VRAM to wipe; the next handle's first large hipMalloc
    # waits for that — 3 s on some boxes — which is no property of either route: give it the time, and one untimed run)
    time.sleep(4)
    cli()
    time.sleep(4)
    host, dth = cli()
    print("CLI, members inflated on the host: %.2f s end to end, %.1f GB/s of FASTQ, %.1f G k-mers/s" % (dth, size / dth / 1e9, n_reads * 120 / dth / 1e9), flush=True)
    print("counts: library loop == CLI (GPU inflater) == CLI (host inflater): %s" % (np.array_equal(lib_counts, got) and np.array_equal(got, host)), flush=True)
    os.remove(fq)
    os.remove(gz)
    if not (np.array_equal(lib_counts, got) and np.array_equal(got, host)):
        sys.exit(1)


if __name__ == "__main__":
    main()
