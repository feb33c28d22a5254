"""Drop-in for kmer_mapper/mapper.pyx — same names, arguments, return dtypes and shapes,
executed by the HIP kernels behind include/kmm.h (never on the CPU).

  map_kmers_to_graph_index(index, max_node_id, kmers, max_index_lookup_frequency=1000)
      -> np.uint32[max_node_id+1]                       mapper.pyx:19-72
  in_graph_index(index, kmers, max_index_lookup_frequency=1000) -> np.uint8[len(kmers)]
                                                        mapper.pyx:81-130
  in_graph_index_no_memory_maps(...)                    mapper.pyx:137-190 (same result)

`index` is duck-typed exactly like the reference: only _hashes_to_index, _n_kmers, _nodes, _kmers,
_frequencies, _modulo are read (mapper.pyx:22-29); wrong dtypes raise ValueError like Cython's
typed memoryviews do.  The HBM copy of an index is cached per index object so that repeated
calls (one per chunk in map_cpu, command_line_interface.py:51) do not re-upload it.
"""
import numpy as np

from .engine import DeviceIndex

_CACHE = {}      # key -> (DeviceIndex, strong refs to the arrays so ids stay valid)
_CACHE_MAX = 4


def _device_index(index, max_node_id, device=0):
    arrays = (index._hashes_to_index, index._n_kmers, index._nodes, index._kmers,
              index._frequencies)
    key = (tuple(id(a) for a in arrays), int(index._modulo), int(max_node_id), int(device))
    hit = _CACHE.get(key)
    if hit is not None:
        return hit[0]
    dev = DeviceIndex(index._hashes_to_index, index._n_kmers, index._modulo, index._kmers,
                      index._nodes, index._frequencies, max_node_id, device=device)
    while len(_CACHE) >= _CACHE_MAX:
        _, (old, _) = _CACHE.popitem()
        old.close()
    _CACHE[key] = (dev, arrays)
    return dev


def clear_cache():
    while _CACHE:
        _, (old, _) = _CACHE.popitem()
        old.close()


def _check_kmers(kmers):
    kmers = np.asarray(kmers) if not hasattr(kmers, "data_ptr") else kmers
    if isinstance(kmers, np.ndarray):
        if kmers.dtype != np.uint64:
            raise ValueError("Buffer dtype mismatch, expected 'uint64_t' but got '%s'" % kmers.dtype)
        if kmers.ndim != 1:
            raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % kmers.ndim)
        if not kmers.flags.c_contiguous:
            raise ValueError("ndarray is not C-contiguous")
    return kmers


class NodeCountAccumulator:
    """The SUM of many map_kmers_to_graph_index calls without the per-call traffic.  The reference's callers add the
    per-chunk vectors up (command_line_interface.py:51,124-130: one `np.zeros(max_node_id + 1)` per chunk, mapper.pyx:37,
    summed by the pool); reproduced call for call that is a reset, a map and a copy of 4 (max_node_id + 1) bytes to the host
    PER CHUNK — 400 MB at the 100 M index for a 2.5 MB chunk.  An accumulator keeps ONE count vector in HBM:

        with NodeCountAccumulator(index, max_node_id) as acc:
            for kmers in chunks:
                map_kmers_to_graph_index(index, max_node_id, kmers, accumulate_into=acc)     # returns None: nothing moves
            counts = acc.node_counts()                                                       # fetched once

    Bit-identical to summing the per-call results (uint32 additions wrap like the reference's, mapper.pyx:37,68)."""

    def __init__(self, index, max_node_id, device=0):
        self._dev = _device_index(index, max_node_id, device)
        self._dev.reset()

    def map_kmers(self, kmers, max_index_lookup_frequency=1000):
        self._dev.map_kmers(_check_kmers(kmers), max_index_lookup_frequency)

    def node_counts(self):
        return self._dev.get_node_counts()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def map_kmers_to_graph_index(index, max_node_id, kmers, max_index_lookup_frequency=1000, accumulate_into=None):
    """mapper.pyx:19-72.  accumulate_into (extension): a NodeCountAccumulator of the same index — the call adds to its
    vector in HBM and returns None instead of a fresh host array."""
    kmers = _check_kmers(kmers)
    dev = _device_index(index, max_node_id)
    if accumulate_into is not None:
        if accumulate_into._dev is not dev:
            raise ValueError("accumulate_into belongs to another index / max_node_id")
        dev.map_kmers(kmers, max_index_lookup_frequency)
        return None
    dev.reset()                      # mapper.pyx:37 — a fresh zeroed vector per call
    dev.map_kmers(kmers, max_index_lookup_frequency)
    return dev.get_node_counts()


def in_graph_index(index, kmers, max_index_lookup_frequency=1000):
    kmers = _check_kmers(kmers)
    max_node_id = int(np.max(index._nodes)) if len(index._nodes) else 0
    return _device_index(index, max_node_id).in_index(kmers)


in_graph_index_no_memory_maps = in_graph_index
