"""Drop-in for kmer_mapper/gpu_counter.py — same class and method names; the cucounter/cupy hash
table behind the reference's class is REPLACED by the HIP engine (not wrapped).

Reference behaviour (gpu_counter.py:5-37): count every query k-mer against the index k-mers, then
node_counts = bincount(nodes, weights=count_of(kmer of each entry), minlength=min_nodes), i.e. every
index entry whose k-mer was queried c times adds c to its node — with NO frequency filter and as
float64.  That is mapper.pyx:53-69 with the filter disabled, which is what this class computes
(filter configurable through `max_index_lookup_frequency`; default: none, like the reference).
"""
import numpy as np

from .engine import DeviceIndex
from .kmer_index import KmerIndex

_NO_FILTER = 65535  # frequencies are uint16: nothing exceeds this


class GpuCounter:
    def __init__(self, unique_kmers, kmers, nodes, k, max_index_lookup_frequency=_NO_FILTER):
        self.unique_kmers = unique_kmers
        self.kmers = kmers
        self.nodes = nodes
        self.counter = None
        self.k = k
        self.max_index_lookup_frequency = max_index_lookup_frequency

    @classmethod
    def from_kmers_and_nodes(cls, kmers, nodes, k) -> "GpuCounter":
        unique_kmers = np.unique(kmers)
        return cls(unique_kmers, kmers, nodes, k)

    def initialize_cuda(self, modulo, device=0, per_kmer=True):
        """Builds the device table (gpu_counter.py:13-16).  `modulo` is the hash-table capacity
        (`--gpu-hash-map-size`); 0 picks a prime near 2x the number of entries.  per_kmer: keep the counts per
        index k-mer like cucounter's table does (get_kmer_counts); the node counts are their segmented sum."""
        n = len(self.kmers)
        if not modulo:
            modulo = _next_prime(max(2 * n, 3))
        self._modulo = int(modulo)
        index = KmerIndex.from_flat_kmers_gpu(np.asarray(self.kmers, dtype=np.uint64),
                                              np.asarray(self.nodes), int(modulo), device=device)
        self._max_node = int(np.max(self.nodes)) if n else 0
        self.counter = DeviceIndex.from_index(index, self._max_node, device=device)
        self._per_kmer = bool(per_kmer) and bool(self.counter.get_param("radix_available"))
        if self._per_kmer:
            self.counter.count_kmers_mode(True)

    initialize = initialize_cuda

    def count(self, kmers, count_revcomps=False):
        self.counter.map_kmers(kmers, self.max_index_lookup_frequency, also_revcomp=count_revcomps,
                               k=self.k)

    def get_kmer_counts(self):
        """uint32 count per entry of `self.kmers` (what `counter[self.kmers]` returns in the reference,
        gpu_counter.py:29-34): the number of counted k-mers equal to that entry's k-mer."""
        if not getattr(self, "_per_kmer", False):
            raise RuntimeError("initialize_cuda(..., per_kmer=True) is needed for per-k-mer counts")
        by_entry = self.counter.get_kmer_counts()
        # the device index holds the entries stably sorted by kmer % modulo (from_flat_kmers)
        order = np.argsort(np.asarray(self.kmers, dtype=np.uint64) % np.uint64(self._modulo), kind="stable")
        out = np.empty_like(by_entry)
        out[order] = by_entry
        return out

    def get_node_counts(self, min_nodes=0):
        counts = self.counter.get_node_counts()
        if min_nodes > counts.shape[0]:       # np.bincount(..., minlength=min_nodes)
            counts = np.concatenate([counts, np.zeros(min_nodes - counts.shape[0], counts.dtype)])
        return counts.astype(np.float64)      # np.bincount with weights returns float64


def _next_prime(n):
    def is_prime(x):
        if x < 2:
            return False
        if x % 2 == 0:
            return x == 2
        i = 3
        while i * i <= x:
            if x % i == 0:
                return False
            i += 2
        return True
    while not is_prime(n):
        n += 1
    return n
