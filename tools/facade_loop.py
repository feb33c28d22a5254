#!/usr/bin/env python3
"""The reference's per-chunk calling pattern through the drop-in façade (command_line_interface.py:51: one
map_kmers_to_graph_index per 2.5 MB chunk, the results summed) at the 100 M-k-mer index: call for call as the reference does
it (a fresh 400 MB vector to the host per chunk) against mapper.NodeCountAccumulator (one vector in HBM, fetched once).
    python tools/facade_loop.py [n_index=100000000] [n_chunks=40]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import mapper, synthetic as syn             # noqa: E402
from kmer_mapper_amd.engine import extract_kmers                  # noqa: E402


def main():
    n_index = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    index, genome = syn.make_index(n_index, seed=1, gpu_builder=True)
    mx = index.max_node_id()
    reads_per_chunk = 8_000            # a 2.5 MB FASTQ chunk (-c 2500000, command_line_interface.py:169) holds ~8 k reads of 150 bp
    bases, offs = syn.make_reads(genome, reads_per_chunk * 4, 150, seed=2)
    chunks = [extract_kmers(bases[offs[i * reads_per_chunk]:offs[(i + 1) * reads_per_chunk]],
                            offs[i * reads_per_chunk:(i + 1) * reads_per_chunk + 1] - offs[i * reads_per_chunk], 31) for i in range(4)]
    n_k = sum(len(chunks[i & 3]) for i in range(n_chunks))
    mapper.map_kmers_to_graph_index(index, mx, chunks[0])          # index upload, warm-up
    t0 = time.perf_counter()
    total = np.zeros(mx + 1, dtype=np.uint32)
    for i in range(n_chunks):
        total += mapper.map_kmers_to_graph_index(index, mx, chunks[i & 3])
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    with mapper.NodeCountAccumulator(index, mx) as acc:
        for i in range(n_chunks):
            mapper.map_kmers_to_graph_index(index, mx, chunks[i & 3], accumulate_into=acc)
        got = acc.node_counts()
    t_acc = time.perf_counter() - t0
    print("index %d k-mers (count vector %.0f MB), %d chunks of %d k-mers" % (n_index, 4 * (mx + 1) / 1e6, n_chunks, len(chunks[0])))
    print("reference-shaped loop (reset + map + fetch per chunk, summed on the host): %.3f s = %.1f ms per chunk, %.2f G k-mers/s"
          % (t_ref, t_ref / n_chunks * 1e3, n_k / t_ref / 1e9))
    print("NodeCountAccumulator (one vector in HBM, one fetch):                        %.3f s = %.1f ms per chunk, %.2f G k-mers/s"
          % (t_acc, t_acc / n_chunks * 1e3, n_k / t_acc / 1e9))
    print("equal: %s" % np.array_equal(got, total))
    if not np.array_equal(got, total):
        sys.exit(1)


if __name__ == "__main__":
    main()
