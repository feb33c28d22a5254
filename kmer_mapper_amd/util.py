"""Drop-in for the hot-path half of kmer_mapper/util.py.

get_kmer_hashes_from_chunk_sequence(chunk_sequence, kmer_size) -> np.uint64[n]  (util.py:71-75)

The reference receives a bionumpy EncodedRaggedArray; bionumpy is not a dependency here, so a
chunk is the pair the ragged array is made of: flat ASCII bytes + row offsets (`ReadBatch`).
"""
import numpy as np

from .engine import extract_kmers


class ReadBatch:
    """A chunk of reads: `bases` uint8[sum(len)] (ASCII), `offsets` int64[n_reads+1].

    Stands in for bionumpy's `chunk.sequence` (command_line_interface.py:110,
    util.py:72): flat data + row boundaries, reads in file order."""

    def __init__(self, bases, offsets):
        self.bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if self.offsets.ndim != 1 or self.offsets.size < 1 or self.offsets[0] != 0:
            raise ValueError("offsets must be int64[n_reads+1] starting at 0")
        if self.offsets[-1] != self.bases.size:
            raise ValueError("offsets[-1] (%d) != len(bases) (%d)" % (self.offsets[-1], self.bases.size))

    def __len__(self):
        return self.offsets.size - 1

    def n_kmers(self, k):
        """Number of k-mer windows in the batch: sum over reads of max(len - k + 1, 0)."""
        return int(np.maximum(np.diff(self.offsets) - k + 1, 0).sum())

    @property
    def uniform_length(self):
        """Read length if every read has the same length, else None."""
        n = len(self)
        if n == 0:
            return None
        L = int(self.offsets[1])
        if self.offsets[-1] == n * L and (n < 3 or np.array_equal(
                self.offsets, np.arange(n + 1, dtype=np.int64) * L)):
            return L
        return None

    @classmethod
    def from_strings(cls, reads):
        enc = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        offsets = np.zeros(len(enc) + 1, dtype=np.int64)
        np.cumsum([len(e) for e in enc], out=offsets[1:])
        return cls(np.frombuffer(b"".join(enc), dtype=np.uint8), offsets)


def as_read_batch(chunk_sequence):
    if isinstance(chunk_sequence, ReadBatch):
        return chunk_sequence
    if isinstance(chunk_sequence, tuple) and len(chunk_sequence) == 2:
        return ReadBatch(*chunk_sequence)
    if isinstance(chunk_sequence, (list,)):
        return ReadBatch.from_strings(chunk_sequence)
    raise TypeError("chunk_sequence must be a ReadBatch, a (bases, offsets) pair or a list of reads")


def get_kmer_hashes_from_chunk_sequence(chunk_sequence, kmer_size, lut=None, device=0):
    batch = as_read_batch(chunk_sequence)
    return extract_kmers(batch.bases, batch.offsets, kmer_size, lut=lut, device=device)
