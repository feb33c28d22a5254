"""KmerIndex — the Kmer Index (.npz) surface the hot path consumes.

Stand-alone restatement of the parts of graph_kmer_index.KmerIndex that kmer_mapper touches
(kmer_mapper/util.py:38-68, kmer_mapper/mapper.pyx:22-29, tests/test_mapping.py:33-38).
graph_kmer_index itself is an un-vendored dependency of the reference; the .npz key names and the
from_flat_kmers construction are taken from the public project and are UPSTREAM-UNVERIFIED here
(SURVEY.md §8c).  The invariants the lookup relies on are exactly those of mapper.pyx:53-69:
entries grouped by `kmer % modulo`; hashes_to_index[h] = first entry of bucket h; n_kmers[h] =
bucket length; frequencies[l] = number of index entries holding the same k-mer as entry l.
"""
import numpy as np

_NPZ_KEYS = ("hashes_to_index", "n_kmers", "nodes", "ref_offsets", "kmers", "modulo", "frequencies",
             "allele_frequencies")


class KmerIndex:
    def __init__(self, hashes_to_index, n_kmers, nodes, kmers, modulo, frequencies,
                 ref_offsets=None, allele_frequencies=None):
        self._hashes_to_index = hashes_to_index
        self._n_kmers = n_kmers
        self._nodes = nodes
        self._kmers = kmers
        self._modulo = int(modulo)
        self._frequencies = frequencies
        self._ref_offsets = ref_offsets
        self._allele_frequencies = allele_frequencies

    # -- what kmer_mapper calls (util.py:42-43,60-62; command_line_interface.py:51,79,117) -------
    def convert_to_int32(self):
        self._hashes_to_index = np.ascontiguousarray(self._hashes_to_index, dtype=np.int32)
        self._n_kmers = np.ascontiguousarray(self._n_kmers, dtype=np.int32)
        self._nodes = np.ascontiguousarray(self._nodes, dtype=np.int32)
        self._kmers = np.ascontiguousarray(self._kmers, dtype=np.uint64)
        self._frequencies = np.ascontiguousarray(self._frequencies, dtype=np.uint16)

    def remove_ref_offsets(self):
        self._ref_offsets = None

    def max_node_id(self):
        return int(self._nodes.max()) if len(self._nodes) else 0

    # -- file surface ----------------------------------------------------------------------------
    @classmethod
    def from_file(cls, path):
        data = np.load(path)
        get = lambda k: data[k] if k in data.files else None
        return cls(data["hashes_to_index"], data["n_kmers"], data["nodes"], data["kmers"],
                   int(data["modulo"]), data["frequencies"], get("ref_offsets"),
                   get("allele_frequencies"))

    def to_file(self, path):
        d = dict(hashes_to_index=self._hashes_to_index, n_kmers=self._n_kmers, nodes=self._nodes,
                 kmers=self._kmers, modulo=np.int64(self._modulo), frequencies=self._frequencies)
        if self._ref_offsets is not None:
            d["ref_offsets"] = self._ref_offsets
        if self._allele_frequencies is not None:
            d["allele_frequencies"] = self._allele_frequencies
        np.savez(path, **d)

    # -- construction (tests/test_mapping.py:36-38: FlatKmers -> from_flat_kmers(modulo)) --------
    @classmethod
    def from_flat_kmers_gpu(cls, kmers, nodes, modulo, device=0):
        """Same index, built by the hand-written counting sort on the GPU (kmm_build_index)."""
        from .engine import build_index
        h2i, nk, ko, no, fo = build_index(kmers, nodes, modulo, device=device)
        return cls(h2i, nk, no, ko, int(modulo), fo)

    @classmethod
    def from_flat_kmers(cls, kmers, nodes, modulo, ref_offsets=None):
        kmers = np.asarray(kmers, dtype=np.uint64)
        nodes = np.asarray(nodes)
        modulo = int(modulo)
        hashes = kmers % np.uint64(modulo)
        order = np.argsort(hashes, kind="stable")
        hashes, kmers, nodes = hashes[order], kmers[order], nodes[order]
        if ref_offsets is not None:
            ref_offsets = np.asarray(ref_offsets)[order]
        uniq_h, first, cnt = np.unique(hashes, return_index=True, return_counts=True)
        h2i = np.zeros(modulo, dtype=np.int64)
        nk = np.zeros(modulo, dtype=np.int64)
        h2i[uniq_h.astype(np.int64)] = first
        nk[uniq_h.astype(np.int64)] = cnt
        uk, inv, kc = np.unique(kmers, return_inverse=True, return_counts=True)
        freqs = np.minimum(kc[inv], 65535).astype(np.uint16)
        ix = cls(h2i, nk, nodes, kmers, modulo, freqs, ref_offsets)
        ix.convert_to_int32()
        return ix
