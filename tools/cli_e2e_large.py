#!/usr/bin/env python3
"""`kmer_mapper map` end to end on a >= 3 GB plain FASTQ against the 100 M-k-mer index: the CLI
accumulates its -c chunks into GPU batches large enough for the radix path; the log must say `path_taken: radix`.
    python tools/cli_e2e_large.py [n_reads=10000000] [n_index=100000000] [out_dir=/tmp/kmm_e2e_large]
Prints the end-to-end rate (file bytes -> node counts on the host) and compares the counts with the oracle."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import synthetic as syn                     # noqa: E402
from kmer_mapper_amd.command_line_interface import map_bnp        # noqa: E402
from tools.cli_e2e import write_fastq_fast                        # noqa: E402


def _node_cpus(node):
    from kmer_mapper_amd.distributed import _parse_cpulist
    with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
        return _parse_cpulist(f.read())


def page_nodes(path):
    """On which NUMA nodes do the file's page-cache pages lie (sampled)?"""
    from kmer_mapper_amd.reads_io import MmapChunker
    c = MmapChunker(path, 1 << 20)
    try:
        return c.page_nodes(64)
    finally:
        c.close()


def main():
    import logging
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format='%(asctime)s %(levelname)s: %(message)s')
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    n_index = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    out_dir = sys.argv[3] if len(sys.argv) > 3 else "/tmp/kmm_e2e_large"
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    index, genome = syn.make_index(n_index, seed=1, gpu_builder=True)
    bases, offs = syn.make_reads(genome, n_reads, 150, seed=2)
    fq = os.path.join(out_dir, "reads.fq")
    # KMM_E2E_WRITE_NODE=n: the file is WRITTEN by a process bound to node n's CPUs, so its page-cache pages lie there (first
    # touch) — the controlled form of what otherwise depends on where the writer happened to run
    write_node = os.environ.get("KMM_E2E_WRITE_NODE")
    before = os.sched_getaffinity(0)
    if write_node is not None:
        os.sched_setaffinity(0, _node_cpus(int(write_node)) & before)
    write_fastq_fast(fq, bases, n_reads, 150)
    if write_node is not None:
        os.sched_setaffinity(0, before)
    size = os.path.getsize(fq)
    try:
        print("page-cache pages of the FASTQ by NUMA node (sampled): %s" % page_nodes(fq), flush=True)
    except Exception as exc:                                  # noqa: BLE001 - a label for the measurement, nothing more
        print("page-cache placement unknown: %s" % exc, flush=True)
    print("setup %.1f s: %d-entry index, %d reads, FASTQ of %.2f GB" % (time.time() - t0, len(index._kmers), n_reads, size / 1e9),
          flush=True)

    def run(chunk, extra=None):
        args = argparse.Namespace(kmer_index=index, index_bundle=None, reads=fq, kmer_size=31, n_threads=16,
                                  chunk_size=chunk, output_file=None, debug=None, max_hits_per_kmer=1000, gpu=True,
                                  gpu_hash_map_size=0, map_reverse_complements=False, apply_max_hits_per_kmer=False,
                                  host_parser=False, device=0)
        for k, v in (extra or {}).items():
            setattr(args, k, v)
        t = time.perf_counter()
        counts = map_bnp(args)
        return counts, time.perf_counter() - t

    run(2_500_000)                                            # warm (page cache, library, index upload path)
    got, dt = run(2_500_000)                                  # the reference's default -c
    print("E2E -c 2500000 (default): %.2f s, %.2f GB/s of FASTQ, %.2f M reads/s, %.2f G k-mers/s"
          % (dt, size / dt / 1e9, n_reads / dt / 1e6, n_reads * 120 / dt / 1e9), flush=True)
    os.environ["KMM_CLI_NO_BATCHING"] = "1"
    _, dt1 = run(2_500_000)
    del os.environ["KMM_CLI_NO_BATCHING"]
    print("E2E -c 2500000 without batch accumulation (every chunk its own direct-path call): %.2f s, %.2f G k-mers/s"
          % (dt1, n_reads * 120 / dt1 / 1e9), flush=True)
    from oracle import oracle
    expect, n = oracle.map_reads(index, index.max_node_id(), bases, offs, 31, n_threads=16)
    print("counts vs oracle on all %d k-mers: %s" % (n, "BIT-EXACT" if np.array_equal(got, expect) else "MISMATCH"), flush=True)
    os.remove(fq)
    if not np.array_equal(got, expect):
        sys.exit(1)


if __name__ == "__main__":
    main()
