"""ctypes binding of libkmm.so (the C ABI declared in include/kmm.h).

The library is built in-tree by `build()` (hipcc --offload-arch=gfx950) and loaded from
kmer_mapper_amd/libkmm.so.  No fallback exists: a missing library raises at first use.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SO_PATH = os.environ.get("KMM_LIB_PATH") or os.path.join(_HERE, "libkmm.so")   # override: A/B builds
SRC = os.path.join(_HERE, "csrc", "kmm.hip")
INCLUDE = os.path.join(ROOT, "include")

KMM_OK = 0
KMM_ERR_INVALID_ARG = -1
KMM_ERR_HIP = -2
KMM_ERR_INDEX = -3
KMM_ERR_INVALID_BASE = -4
KMM_ERR_NOMEM = -5
KMM_ERR_MALFORMED = -6
KMM_ERR_INTERNAL = -7
FORMAT_FASTA2, FORMAT_FASTQ = 2, 4
FORMAT_FASTA, FORMAT_LAST_CHUNK, FORMAT_NEW_STREAM = 1, 0x100, 0x400      # multi-line FASTA (unwrapped on the GPU); flag: the chunk ends the file

# kernel ids of kmm_get_timing (include/kmm.h)
(KERNEL_MAP_READS, KERNEL_MAP_KMERS, KERNEL_RX_P1, KERNEL_RX_SCAN, KERNEL_RX_P2, KERNEL_RX_P3,
 KERNEL_RX_FLUSH) = range(7)
KERNEL_NAMES = ("k_map_reads", "k_map_kmers", "k_rx_p1", "k_rx_scan", "k_rx_p2", "k_rx_p3", "k_rx_flush")

_c = ctypes
_P = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/kmm.h declares
SIGNATURES = {
    "kmm_version": (_c.c_char_p, []),
    "kmm_last_error": (_c.c_char_p, []),
    "kmm_device_count": (_c.c_int, [_P]),
    "kmm_device_pci_bus_id": (_c.c_int, [_c.c_int, _c.c_char_p, _c.c_int]),
    "kmm_host_alloc": (_c.c_int, [_c.c_size_t, _P]),
    "kmm_host_free": (_c.c_int, [_P]),
    "kmm_host_reserve": (_c.c_int, [_c.c_int64]),
    "kmm_host_reserve_buffer": (_c.c_int, [_c.c_int64]),
    "kmm_index_create": (_c.c_int, [_P, _P, _c.c_uint64, _P, _P, _P, _c.c_int64, _c.c_int64,
                                    _c.c_int, _P]),
    "kmm_index_destroy": (None, [_P]),
    "kmm_reset_counts": (_c.c_int, [_P]),
    "kmm_bind_counts": (_c.c_int, [_P, _P]),
    "kmm_counts_device_ptr": (_c.c_int, [_P, _P]),
    "kmm_get_node_counts": (_c.c_int, [_P, _P]),
    "kmm_synchronize": (_c.c_int, [_P]),
    "kmm_get_kmer_counts": (_c.c_int, [_P, _P]),
    "kmm_reduce_counts": (_c.c_int, [_P, _c.c_int, _c.c_int]),
    "kmm_comm_get_unique_id": (_c.c_int, [_P]),
    "kmm_comm_init_rank": (_c.c_int, [_P, _P, _c.c_int, _c.c_int]),
    "kmm_comm_reduce_counts": (_c.c_int, [_P, _c.c_int]),
    "kmm_comm_destroy": (_c.c_int, [_P]),
    "kmm_map_kmers": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int]),
    "kmm_map_reads": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P]),
    "kmm_map_reads_uniform": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                                         _c.c_int, _P]),
    "kmm_map_records": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    "kmm_map_bgzf": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    "kmm_map_bgzf_hint_next": (_c.c_int, [_P, _P, _c.c_int64]),
    "kmm_map_packed": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int, _c.c_int, _c.c_int]),
    "kmm_extract_kmers": (_c.c_int, [_c.c_int, _P, _P, _c.c_int64, _c.c_int, _P, _P, _c.c_int64]),
    "kmm_build_index": (_c.c_int, [_c.c_int, _P, _P, _c.c_int64, _c.c_uint64, _P, _P, _P, _P, _P]),
    "kmm_in_index": (_c.c_int, [_P, _P, _c.c_int64, _P]),
    "kmm_set_timing": (_c.c_int, [_P, _c.c_int]),
    "kmm_get_stats": (_c.c_int, [_P, _c.c_int, _P, _P]),
    "kmm_get_timing": (_c.c_int, [_P, _c.c_int, _P, _P]),
    "kmm_set_param": (_c.c_int, [_P, _c.c_char_p, _c.c_int64]),
    "kmm_get_param": (_c.c_int, [_P, _c.c_char_p, _P]),
}


def build(force=False, verbose=False):
    """Compile csrc/kmm.hip for gfx950 into kmer_mapper_amd/libkmm.so (cross-compiles on CPU)."""
    csrc = os.path.dirname(SRC)
    deps = [os.path.join(INCLUDE, "kmm.h")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f != "kmm_io.cpp"]
    if (not force and os.path.exists(SO_PATH)
            and os.path.getmtime(SO_PATH) >= max(os.path.getmtime(d) for d in deps)):
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + INCLUDE, "-o", SO_PATH, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    from . import _io
    _io.build(force=True)               # the host-side reader library (g++), same build step
    return SO_PATH


_lib = None


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64 (SONAME libamdhip64.so.7, the name libkmm.so links against); if libkmm.so were
    loaded first it would pull /opt/rocm's copy and torch would then find "No HIP GPUs".  Importing
    torch first makes the dynamic loader resolve libkmm.so's dependency to the copy torch already
    mapped.  Without torch installed (or with KMM_NO_TORCH=1) /opt/rocm's runtime is used."""
    if os.environ.get("KMM_NO_TORCH") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib():
    """The loaded library; raises RuntimeError if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "kmer_mapper_amd: %s is missing — build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "There is no CPU fallback." % SO_PATH)
        _share_hip_runtime_with_torch()
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("KMM_LIB_PATH") and not hasattr(L, name):
                continue           # A/B runs against an older build may lack newer entry points
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class KmmError(RuntimeError):
    pass


def check(rc):
    """Map a KMM_ERR_* code to the exception the reference raises in the same situation."""
    if rc == KMM_OK:
        return
    msg = lib().kmm_last_error().decode("utf-8", "replace")
    if rc in (KMM_ERR_INVALID_ARG, KMM_ERR_INDEX, KMM_ERR_INVALID_BASE, KMM_ERR_MALFORMED):
        raise ValueError(msg)       # Cython buffer / bionumpy encoding errors are ValueError-like
    if rc == KMM_ERR_NOMEM:
        raise MemoryError(msg)
    raise KmmError(msg)


def pinned_array(n, dtype):
    """A numpy array of n elements in page-locked host memory (kmm_host_alloc), freed with the array: device -> host copies
    into it run at the link's rate (a 400 MB count vector: 7 ms instead of ~25 into pageable memory)."""
    import weakref
    import numpy as np
    dt = np.dtype(dtype)
    nbytes = max(int(n) * dt.itemsize, 1)
    p = _c.c_void_p()
    check(lib().kmm_host_alloc(nbytes, _c.byref(p)))
    buf = (_c.c_uint8 * nbytes).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dt, count=int(n))
    weakref.finalize(buf, lib().kmm_host_free, _c.c_void_p(p.value))
    return arr


def device_count():
    n = _c.c_int(0)
    check(lib().kmm_device_count(_c.byref(n)))
    return n.value
