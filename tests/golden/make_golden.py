"""Regenerates tests/golden/golden_small.npz — regression vectors for the hot path.

The reference itself cannot be imported here (its mapper.pyx imports un-vendored packages at
module level and bionumpy/graph_kmer_index are absent), so these vectors are produced by the CPU
oracle (oracle/kmm_oracle.c) AFTER it has been pinned against tests/golden/reference_vectors.json
(the reference's own known answers).  They freeze inputs + expected outputs so that the GPU box,
where /root/reference does not exist, checks the same numbers.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from kmer_mapper_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    k = 31
    index, genome = syn.make_index(400, k=k, seed=11)
    mx = index.max_node_id()
    b1, o1 = syn.make_reads(genome, 300, 150, seed=12)
    b2, o2 = syn.make_ragged_reads(genome, 200, 0, 260, seed=13)
    out = dict(
        k=np.int64(k), max_node_id=np.int64(mx), modulo=np.int64(index._modulo),
        hashes_to_index=index._hashes_to_index, n_kmers=index._n_kmers, nodes=index._nodes,
        kmers=index._kmers, frequencies=index._frequencies,
        uniform_bases=b1, uniform_offsets=o1, ragged_bases=b2, ragged_offsets=o2)
    for name, (b, o) in dict(uniform=(b1, o1), ragged=(b2, o2)).items():
        km = oracle.extract(b, o, k)
        out[name + "_kmers"] = km
        out[name + "_counts"] = oracle.map_kmers(index, mx, km)
        out[name + "_counts_maxfreq2"] = oracle.map_kmers(index, mx, km, 2)
        out[name + "_counts_nofilter"] = oracle.map_kmers(index, mx, km, 65535)
        rc = oracle.map_kmers(index, mx, oracle.revcomp(km, k), out=oracle.map_kmers(index, mx, km))
        out[name + "_counts_revcomp"] = rc
        out[name + "_in_index"] = oracle.in_index(index, km)
    # the ragged reads once more as a raw FASTQ chunk (CRLF on every third record, '@' / '+' leading
    # quality strings) for the GPU record parser; expected counts = ragged_counts
    rec = []
    for i in range(len(o2) - 1):
        seq = b2[o2[i]:o2[i + 1]].tobytes()
        eol = b"\r\n" if i % 3 == 0 else b"\n"
        qual = (b"@+I" * len(seq))[:len(seq)]
        rec.append(b"@r%d" % i + eol + seq + eol + b"+" + eol + qual + eol)
    out["ragged_fastq"] = np.frombuffer(b"".join(rec), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "golden_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
