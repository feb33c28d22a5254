"""ctypes wrapper around oracle/libkmm_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of kmm_oracle.c).  Nothing under kmer_mapper_amd/ imports it.

Each function mirrors one reference entry point:
  map_kmers  -> kmer_mapper/mapper.pyx:19-72   (map_kmers_to_graph_index)
  in_index   -> kmer_mapper/mapper.pyx:81-130  (in_graph_index)
  extract    -> kmer_mapper/util.py:71-75      (get_kmer_hashes_from_chunk_sequence)
  map_reads  -> kmer_mapper/command_line_interface.py:32-56 + :124-130 (map_cpu + additive reduce)
  build_index -> tests/test_mapping.py:36-38 (FlatKmers -> KmerIndex.from_flat_kmers(modulo) -> convert_to_int32;
                 graph_kmer_index itself is un-vendored: the invariants are those mapper.pyx:53-69 reads)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkmm_oracle.so")

_c = ctypes
_P = ctypes.c_void_p


def build(force=False):
    """Compile the C restatement (gcc, seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "kmm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkmm_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def _cpu_signature():
    """Short hash of this host's CPU flags: a -march=native build is only ever loaded on the CPU it was built for."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((ln for ln in f if ln.startswith("flags")), "")
    except OSError:
        flags = ""
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


_native = None          # path of the -march=native build in use, or None (portable build)


def use_native_build():
    """BASELINE.md section 3 / reference setup.py:10-15 compile with -O3 -march=native: build such a copy ON THIS HOST
    (gcc, seconds) and use it from now on; the portable -O3 build stays the fallback when gcc refuses.  Returns the
    flags in use.  Call before the first lib()."""
    global _native, _lib
    so = os.path.join(_HERE, "libkmm_oracle_native_%s.so" % _cpu_signature())
    src = os.path.join(_HERE, "kmm_oracle.c")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-pthread", "-shared", "-o", so, src,
                                   "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _native, _lib = so, None
        return "-O3 -march=native (built on this host)"
    except Exception:
        return "-O3 (portable build; -march=native failed here)"


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_native or _SO)
        L.oracle_map_kmers.argtypes = [_P, _P, _c.c_uint64, _P, _P, _P, _P, _c.c_int64, _c.c_int, _P]
        L.oracle_map_kmers.restype = None
        L.oracle_in_index.argtypes = [_P, _P, _c.c_uint64, _P, _P, _c.c_int64, _P]
        L.oracle_in_index.restype = None
        L.oracle_extract_kmers.argtypes = [_P, _P, _c.c_int64, _c.c_int, _P, _P]
        L.oracle_extract_kmers.restype = _c.c_int64
        L.oracle_revcomp.argtypes = [_c.c_uint64, _c.c_int]
        L.oracle_revcomp.restype = _c.c_uint64
        L.oracle_default_lut.argtypes = [_P]
        L.oracle_default_lut.restype = None
        L.oracle_map_reads.argtypes = [_P, _P, _c.c_uint64, _P, _P, _P, _c.c_int64, _P, _P,
                                       _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int,
                                       _c.c_int64, _P]
        L.oracle_map_reads.restype = _c.c_int64
        L.oracle_build_index.argtypes = [_P, _P, _c.c_int64, _c.c_uint64, _P, _P, _P, _P, _P]
        L.oracle_build_index.restype = _c.c_int
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(_P) if a is not None else None


def _index_arrays(index):
    """Same attribute reads and dtypes as mapper.pyx:22-29 (int32, int32, int32, uint64, uint16)."""
    def chk(a, dt, name):
        a = np.asarray(a)
        if a.dtype != dt:
            raise ValueError("Buffer dtype mismatch for %s: expected %s got %s" % (name, dt, a.dtype))
        return np.ascontiguousarray(a)
    return (chk(index._hashes_to_index, np.int32, "_hashes_to_index"),
            chk(index._n_kmers, np.int32, "_n_kmers"),
            chk(index._nodes, np.int32, "_nodes"),
            chk(index._kmers, np.uint64, "_kmers"),
            chk(index._frequencies, np.uint16, "_frequencies"),
            int(index._modulo))


def map_kmers(index, max_node_id, kmers, max_index_lookup_frequency=1000, out=None):
    h2i, nk, nodes, ikm, fr, mod = _index_arrays(index)
    kmers = np.asarray(kmers)
    if kmers.dtype != np.uint64:
        raise ValueError("Buffer dtype mismatch, expected 'uint64_t' got %s" % kmers.dtype)
    kmers = np.ascontiguousarray(kmers)
    counts = np.zeros(int(max_node_id) + 1, dtype=np.uint32) if out is None else out
    lib().oracle_map_kmers(_ptr(h2i), _ptr(nk), mod, _ptr(ikm), _ptr(nodes), _ptr(fr),
                           _ptr(kmers), kmers.shape[0], int(max_index_lookup_frequency),
                           _ptr(counts))
    return counts


def in_index(index, kmers):
    h2i, nk, nodes, ikm, fr, mod = _index_arrays(index)
    kmers = np.ascontiguousarray(np.asarray(kmers, dtype=np.uint64))
    out = np.zeros(kmers.shape[0], dtype=np.uint8)
    lib().oracle_in_index(_ptr(h2i), _ptr(nk), mod, _ptr(ikm), _ptr(kmers), kmers.shape[0],
                          _ptr(out))
    return out


def default_lut():
    lut = np.zeros(256, dtype=np.uint8)
    lib().oracle_default_lut(_ptr(lut))
    return lut


def extract(bases, read_offsets, k, lut=None):
    bases = np.ascontiguousarray(np.asarray(bases, dtype=np.uint8))
    offs = np.ascontiguousarray(np.asarray(read_offsets, dtype=np.int64))
    n_reads = offs.shape[0] - 1
    lens = np.diff(offs)
    n = int(np.maximum(lens - k + 1, 0).sum())
    out = np.empty(n, dtype=np.uint64)
    if lut is not None:
        lut = np.ascontiguousarray(np.asarray(lut, dtype=np.uint8))
    got = lib().oracle_extract_kmers(_ptr(bases), _ptr(offs), n_reads, int(k), _ptr(lut), _ptr(out))
    if got < 0:
        raise ValueError("invalid nucleotide byte at position %d" % (-got - 1))
    assert got == n
    return out


def revcomp(kmers, k):
    L = lib()
    return np.array([L.oracle_revcomp(int(x), int(k)) for x in np.asarray(kmers).ravel()],
                    dtype=np.uint64)


def map_reads(index, max_node_id, bases, read_offsets, k, max_index_lookup_frequency=1000,
              also_revcomp=False, lut=None, n_threads=1, chunk_reads=16384):
    h2i, nk, nodes, ikm, fr, mod = _index_arrays(index)
    bases = np.ascontiguousarray(np.asarray(bases, dtype=np.uint8))
    offs = np.ascontiguousarray(np.asarray(read_offsets, dtype=np.int64))
    counts = np.zeros(int(max_node_id) + 1, dtype=np.uint32)
    if lut is not None:
        lut = np.ascontiguousarray(np.asarray(lut, dtype=np.uint8))
    got = lib().oracle_map_reads(_ptr(h2i), _ptr(nk), mod, _ptr(ikm), _ptr(nodes), _ptr(fr),
                                 int(max_node_id), _ptr(bases), _ptr(offs), offs.shape[0] - 1,
                                 int(k), int(max_index_lookup_frequency), int(bool(also_revcomp)),
                                 _ptr(lut), int(n_threads), int(chunk_reads), _ptr(counts))
    if got < 0:
        raise ValueError("oracle_map_reads failed (%d): invalid nucleotide byte" % got)
    return counts, int(got)


class OracleIndex:
    """What mapper.pyx:22-29 reads of an index (duck-typed there), in the dtypes convert_to_int32 leaves."""

    def __init__(self, h2i, nk, nodes, kmers, modulo, freqs):
        self._hashes_to_index, self._n_kmers, self._nodes = h2i, nk, nodes
        self._kmers, self._modulo, self._frequencies = kmers, int(modulo), freqs

    def max_node_id(self):
        return int(self._nodes.max()) if len(self._nodes) else 0


def build_index(kmers, nodes, modulo):
    """FlatKmers(kmers, nodes) -> from_flat_kmers(modulo) -> convert_to_int32 (tests/test_mapping.py:36-38)."""
    kmers = np.ascontiguousarray(np.asarray(kmers, dtype=np.uint64))
    nodes = np.ascontiguousarray(np.asarray(nodes, dtype=np.int64))
    if kmers.shape != nodes.shape:
        raise ValueError("kmers and nodes differ in length")
    n, modulo = kmers.shape[0], int(modulo)
    h2i = np.empty(modulo, dtype=np.int32)
    nk = np.empty(modulo, dtype=np.int32)
    ko = np.empty(n, dtype=np.uint64)
    no = np.empty(n, dtype=np.int32)
    fo = np.empty(n, dtype=np.uint16)
    rc = lib().oracle_build_index(_ptr(kmers), _ptr(nodes), n, modulo, _ptr(h2i), _ptr(nk), _ptr(ko), _ptr(no), _ptr(fo))
    if rc:
        raise ValueError("oracle_build_index failed (%d)" % rc)
    return OracleIndex(h2i, nk, no, ko, modulo, fo)
