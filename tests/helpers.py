import json
import os
import types

import numpy as np

from kmer_mapper_amd.kmer_index import KmerIndex
from kmer_mapper_amd.util import ReadBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def reference_vectors():
    with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
        return json.load(f)


def index_from_vector(v):
    """The index of a reference vector, built by the ORACLE's builder (oracle_build_index: what
    FlatKmers -> KmerIndex.from_flat_kmers(modulo) -> convert_to_int32 gives, reference tests/test_mapping.py:36-38)."""
    from oracle import oracle
    return oracle.build_index(np.array(v["index_kmers"], dtype=np.uint64),
                              np.array(v["index_nodes"], dtype=np.int64), v["modulo"])


def golden_small():
    d = np.load(os.path.join(GOLDEN, "golden_small.npz"))
    index = types.SimpleNamespace(      # duck-typed, like the reference accepts (mapper.pyx:22-29)
        _hashes_to_index=d["hashes_to_index"], _n_kmers=d["n_kmers"], _nodes=d["nodes"],
        _kmers=d["kmers"], _frequencies=d["frequencies"], _modulo=int(d["modulo"]))
    return d, index, int(d["max_node_id"]), int(d["k"])


def batch(reads):
    return ReadBatch.from_strings(reads)
