"""Regenerates tests/golden/kmer_index_small.npz — a Kmer Index file with the key set and on-disk dtypes of
graph_kmer_index.KmerIndex.to_file as SURVEY.md section 8(c) lists them [UPSTREAM-UNVERIFIED: graph_kmer_index is an
un-vendored dependency of the reference; the reference only shows the read side, kmer_mapper/util.py:56-62:
`KmerIndex.from_file(path)` -> `convert_to_int32()` -> `remove_ref_offsets()`]:

    hashes_to_index int64[modulo], n_kmers int64[modulo], nodes int64[N], ref_offsets int64[N], kmers uint64[N],
    modulo int64 scalar, frequencies uint16[N], allele_frequencies float32[N]

The arrays come from the ORACLE's builder (oracle/kmm_oracle.c oracle_build_index), widened to the on-disk dtypes; the
file also carries query k-mers and the expected node counts (oracle_map_kmers, mapper.pyx:53-69) so that the GPU box
checks frozen numbers.          python tests/golden/make_index_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402


def main():
    rng = np.random.default_rng(20261004)
    modulo, n_nodes = 2003, 600
    kmers = rng.integers(0, 2 ** 62, size=900, dtype=np.uint64)
    kmers = np.concatenate([kmers, kmers[:60],                                       # k-mers under two nodes
                            np.repeat(kmers[5], 1200),                               # frequency 1202 > 1000: filtered
                            np.uint64(modulo) * rng.integers(1, 2 ** 40, size=40, dtype=np.uint64) + np.uint64(17)])  # one bucket
    kmers = kmers[rng.permutation(len(kmers))]
    nodes = rng.integers(0, n_nodes, size=len(kmers))
    ix = oracle.build_index(kmers, nodes, modulo)
    mx = ix.max_node_id()
    absent = rng.integers(0, 2 ** 62, size=500, dtype=np.uint64)
    query = np.concatenate([kmers[rng.integers(0, len(kmers), size=3000)], absent,
                            ix._kmers[:50] + np.uint64(modulo)])                     # same bucket, other k-mer
    query = query[rng.permutation(len(query))]
    out = dict(
        hashes_to_index=ix._hashes_to_index.astype(np.int64), n_kmers=ix._n_kmers.astype(np.int64),
        nodes=ix._nodes.astype(np.int64), ref_offsets=rng.integers(0, 10 ** 9, size=len(kmers)).astype(np.int64),
        kmers=ix._kmers, modulo=np.int64(modulo), frequencies=ix._frequencies,
        allele_frequencies=rng.random(len(kmers)).astype(np.float32),
        # expectations (not part of the upstream key set)
        test_query_kmers=query, test_max_node_id=np.int64(mx),
        test_expected_counts=oracle.map_kmers(ix, mx, query),
        test_expected_counts_maxfreq_65535=oracle.map_kmers(ix, mx, query, 65535),
        test_expected_in_index=oracle.in_index(ix, query))
    path = os.path.join(ROOT, "tests", "golden", "kmer_index_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
