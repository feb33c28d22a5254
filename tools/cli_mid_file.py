#!/usr/bin/env python3
"""`kmer_mapper map` on a FASTQ just BELOW the size from which the CLI accumulates its -c chunks into radix-path batches
(12 x radix_min_units bytes): as the CLI maps it by default (chunk by chunk, direct path) and as ONE call (-c = the file's size).
    python tools/cli_mid_file.py [n_reads=2200000] [n_index=100000000]"""
import argparse
import logging
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import synthetic as syn                     # noqa: E402
from kmer_mapper_amd.command_line_interface import map_bnp        # noqa: E402
from tools.cli_e2e import write_fastq_fast                        # noqa: E402


def main():
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format='%(asctime)s %(levelname)s: %(message)s')
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_200_000
    n_index = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    index, genome = syn.make_index(n_index, seed=1, gpu_builder=True)
    bases, offs = syn.make_reads(genome, n_reads, 150, seed=2)
    fq = "/tmp/kmm_mid.fq"
    write_fastq_fast(fq, bases, n_reads, 150)
    size = os.path.getsize(fq)
    print("FASTQ of %.0f MB, %d reads" % (size / 1e6, n_reads), flush=True)
    results = []
    for what, chunk in (("default -c 2500000", 2_500_000), ("one call (-c = file size)", size + (1 << 20))):
        for rep in range(2):
            ns = argparse.Namespace(kmer_index=index, index_bundle=None, reads=fq, kmer_size=31, n_threads=16, chunk_size=chunk,
                                    output_file=None, debug=None, max_hits_per_kmer=1000, gpu=True, gpu_hash_map_size=0,
                                    map_reverse_complements=False)
            t = time.perf_counter()
            got = map_bnp(ns)
            print("%s, run %d: %.3f s end to end" % (what, rep, time.perf_counter() - t), flush=True)
            results.append(got)
    print("counts equal across all runs:", all(np.array_equal(results[0], r) for r in results[1:]), flush=True)
    os.remove(fq)


if __name__ == "__main__":
    main()
