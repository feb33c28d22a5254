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


def map_kmers_to_graph_index(index, max_node_id, kmers, max_index_lookup_frequency=1000):
    kmers = _check_kmers(kmers)
    dev = _device_index(index, max_node_id)
    dev.reset()                      # mapper.pyx:37 — a fresh zeroed vector per call
    dev.map_kmers(kmers, max_index_lookup_frequency)
    return dev.get_node_counts()


def in_graph_index(index, kmers, max_index_lookup_frequency=1000):
    kmers = _check_kmers(kmers)
    max_node_id = int(np.max(index._nodes)) if len(index._nodes) else 0
    return _device_index(index, max_node_id).in_index(kmers)


in_graph_index_no_memory_maps = in_graph_index
