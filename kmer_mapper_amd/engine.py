"""DeviceIndex — a Kmer Index resident in HBM plus its uint32 node-count vector.

Thin object wrapper over the C ABI (include/kmm.h).  Everything the reference does per chunk in
map_cpu (kmer_mapper/command_line_interface.py:32-56) happens inside `map_reads`; the
operator-level `map_kmers` is the drop-in for mapper.pyx:19-72.
"""
import ctypes

import numpy as np

from . import _lib

_P = ctypes.c_void_p


def _is_torch_tensor(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class _Arg:
    """A borrowed pointer for one C call: numpy array (host) or torch tensor (host or device)."""

    def __init__(self, x, dtype, name):
        self.keep = None
        if x is None:
            self.ptr, self.n = None, 0
            return
        if _is_torch_tensor(x):
            import torch
            want = {np.uint8: torch.uint8, np.int64: torch.int64, np.int32: torch.int32,
                    np.uint64: getattr(torch, "uint64", None), np.uint16: getattr(torch, "uint16", None),
                    np.uint32: getattr(torch, "uint32", None)}[dtype]
            if x.dtype != want and not (dtype is np.uint64 and x.dtype == torch.int64) \
                    and not (dtype is np.uint32 and x.dtype == torch.int32):
                raise ValueError("Buffer dtype mismatch for %s: expected %s got %s"
                                 % (name, np.dtype(dtype), x.dtype))
            if not x.is_contiguous():
                raise ValueError("%s: ndarray is not C-contiguous" % name)
            self.keep, self.ptr, self.n = x, _P(x.data_ptr()), x.numel()
            return
        a = np.asarray(x)
        if a.dtype != np.dtype(dtype):
            # the reference's typed memoryviews reject other dtypes (mapper.pyx:19,22-28)
            raise ValueError("Buffer dtype mismatch for %s: expected '%s' but got '%s'"
                             % (name, np.dtype(dtype), a.dtype))
        if not a.flags.c_contiguous:
            raise ValueError("%s: ndarray is not C-contiguous" % name)
        self.keep, self.ptr, self.n = a, a.ctypes.data_as(_P), a.size


class DeviceIndex:
    """The five index arrays of graph_kmer_index.KmerIndex (mapper.pyx:22-29) repacked in HBM."""

    def __init__(self, hashes_to_index, n_kmers, modulo, kmers, nodes, frequencies, max_node_id,
                 device=0):
        L = _lib.lib()
        h2i = _Arg(hashes_to_index, np.int32, "hashes_to_index")
        nk = _Arg(n_kmers, np.int32, "n_kmers")
        km = _Arg(kmers, np.uint64, "kmers")
        nd = _Arg(nodes, np.int32, "nodes")
        fr = _Arg(frequencies, np.uint16, "frequencies")
        modulo = int(modulo)
        if h2i.n != modulo or nk.n != modulo:
            raise ValueError("hashes_to_index / n_kmers must have `modulo`=%d entries (got %d, %d)"
                             % (modulo, h2i.n, nk.n))
        if not (km.n == nd.n == fr.n):
            raise ValueError("kmers / nodes / frequencies differ in length")
        self._h = _P()
        self.max_node_id = int(max_node_id)
        self.modulo = modulo
        self.n_entries = km.n
        self.device = int(device)
        _lib.check(L.kmm_index_create(h2i.ptr, nk.ptr, modulo, km.ptr, nd.ptr, fr.ptr, km.n,
                                      self.max_node_id, self.device, ctypes.byref(self._h)))
        self._bound = None

    @classmethod
    def from_index(cls, index, max_node_id=None, device=0):
        """From any object with the attributes mapper.pyx:22-29 reads (duck-typed, like the reference)."""
        if max_node_id is None:
            max_node_id = index.max_node_id()
        return cls(index._hashes_to_index, index._n_kmers, index._modulo, index._kmers,
                   index._nodes, index._frequencies, max_node_id, device=device)

    # -- lifetime --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().kmm_index_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- counts ----------------------------------------------------------------------------------
    def reset(self):
        _lib.check(_lib.lib().kmm_reset_counts(self._h))

    def bind_counts(self, tensor):
        """Accumulate into a caller-owned device tensor of max_node_id+1 32-bit ints (for RCCL)."""
        if tensor is None:
            _lib.check(_lib.lib().kmm_bind_counts(self._h, None))
            self._bound = None
            return
        if tensor.numel() != self.max_node_id + 1 or tensor.element_size() != 4:
            raise ValueError("bind_counts needs %d 32-bit elements" % (self.max_node_id + 1))
        _lib.check(_lib.lib().kmm_bind_counts(self._h, _P(tensor.data_ptr())))
        self._bound = tensor

    def synchronize(self):
        _lib.check(_lib.lib().kmm_synchronize(self._h))

    def get_node_counts(self, out=None, pinned=False):
        """The count vector on the host.  pinned: into a fresh page-locked array (freed with it) — the copy of a large
        vector then runs at the link's rate."""
        if out is None:
            out = (_lib.pinned_array(self.max_node_id + 1, np.uint32) if pinned
                   else np.empty(self.max_node_id + 1, dtype=np.uint32))
        _lib.check(_lib.lib().kmm_get_node_counts(self._h, out.ctypes.data_as(_P)))
        return out

    def count_kmers_mode(self, on=True):
        """Per-k-mer counting mode (gpu_counter.py:23-37, command_line_interface.py:46-49): every batch takes the
        radix path, hits are kept per index entry (get_kmer_counts) and summed into the node counts."""
        self.set_param("count_kmers", int(bool(on)))

    def get_kmer_counts(self, out=None):
        """uint32[n_entries]: how many mapped k-mers matched each index entry, in the entry order given to the
        constructor (needs count_kmers_mode() before mapping)."""
        if out is None:
            out = np.empty(self.n_entries, dtype=np.uint32)
        dst = _Arg(out, np.uint32, "out") if not _is_torch_tensor(out) else None
        ptr = dst.ptr if dst is not None else _P(out.data_ptr())
        _lib.check(_lib.lib().kmm_get_kmer_counts(self._h, ptr))
        return out

    # -- multi-GPU: one process per GPU, RCCL behind the C ABI -------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes created on rank 0 (kmm_comm_get_unique_id); hand them to every rank."""
        buf = (ctypes.c_uint8 * 128)()
        _lib.check(_lib.lib().kmm_comm_get_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, n_ranks, rank):
        """Collective: join the communicator of the job with this handle."""
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(_lib.lib().kmm_comm_init_rank(self._h, buf, int(n_ranks), int(rank)))

    def comm_reduce_counts(self, root=0):
        """Collective: sum of the ranks' count vectors, in place (root = -1: on every rank).  Replaces the additive
        reduce of command_line_interface.py:124-130."""
        _lib.check(_lib.lib().kmm_comm_reduce_counts(self._h, int(root)))

    # -- the hot path ----------------------------------------------------------------------------
    def map_kmers(self, kmers, max_index_lookup_frequency=1000, also_revcomp=False, k=31):
        a = _Arg(kmers, np.uint64, "kmers")
        _lib.check(_lib.lib().kmm_map_kmers(self._h, a.ptr, a.n, int(max_index_lookup_frequency),
                                            int(bool(also_revcomp)), int(k)))

    def map_reads(self, bases, read_offsets, k=31, max_index_lookup_frequency=1000,
                  also_revcomp=False, lut=None):
        b = _Arg(bases, np.uint8, "bases")
        o = _Arg(read_offsets, np.int64, "read_offsets")
        t = _Arg(lut, np.uint8, "lut")
        if lut is not None and t.n != 256:
            raise ValueError("lut must have 256 entries")
        if o.n < 1:
            raise ValueError("read_offsets needs n_reads+1 entries")
        _lib.check(_lib.lib().kmm_map_reads(self._h, b.ptr, o.ptr, o.n - 1, int(k),
                                            int(max_index_lookup_frequency),
                                            int(bool(also_revcomp)), t.ptr))

    def map_reads_uniform(self, bases, n_reads, read_len, k=31, max_index_lookup_frequency=1000,
                          also_revcomp=False, lut=None):
        b = _Arg(bases, np.uint8, "bases")
        t = _Arg(lut, np.uint8, "lut")
        if b.n < int(n_reads) * int(read_len):
            raise ValueError("bases holds %d bytes, need n_reads*read_len=%d"
                             % (b.n, int(n_reads) * int(read_len)))
        _lib.check(_lib.lib().kmm_map_reads_uniform(self._h, b.ptr, int(n_reads), int(read_len),
                                                    int(k), int(max_index_lookup_frequency),
                                                    int(bool(also_revcomp)), t.ptr))

    def map_records(self, raw, n_bytes=None, fmt=_lib.FORMAT_FASTQ, k=31, max_index_lookup_frequency=1000,
                    also_revcomp=False, lut=None):
        """Map a raw FASTQ (fmt=4) / two-line FASTA (fmt=2) chunk parsed on the GPU.
        Returns (consumed_bytes, n_records); the caller carries raw[consumed:] to the next chunk."""
        b = _Arg(raw, np.uint8, "raw")
        t = _Arg(lut, np.uint8, "lut")
        n = b.n if n_bytes is None else int(n_bytes)
        if n > b.n:
            raise ValueError("n_bytes exceeds the buffer")
        consumed = ctypes.c_int64(0)
        n_rec = ctypes.c_int64(0)
        _lib.check(_lib.lib().kmm_map_records(self._h, b.ptr, n, int(fmt), int(k),
                                              int(max_index_lookup_frequency), int(bool(also_revcomp)),
                                              t.ptr, ctypes.byref(consumed), ctypes.byref(n_rec)))
        return consumed.value, n_rec.value

    def map_bgzf(self, comp, n_bytes=None, fmt=_lib.FORMAT_FASTQ, k=31, max_index_lookup_frequency=1000, also_revcomp=False,
                 lut=None, first=False, last=False, head_skip=0, tail_stop=None, next_chunk=None):
        """Map a chunk of a BGZF-compressed FASTQ (fmt=4) / two-line FASTA (fmt=2) file, inflated on the GPU (kmm_map_bgzf).
        `comp` starts at a member boundary; returns (compressed bytes used, records mapped): continue at comp[used:].  The
        handle carries the inflated bytes behind the last complete record to the next call; first / last mark the file's
        first / last chunk.  A rank's share of a file (bgzf_ranges.rank_member_range): head_skip = inflated bytes of the
        FIRST chunk's first member that belong to the rank before; tail_stop = how many inflated bytes of the LAST chunk's
        last member are this rank's (None: all).  next_chunk: see below."""
        if next_chunk is not None and len(next_chunk):
            # the bytes that follow `comp` in the caller's memory (a view of the same file mapping): staged under this chunk's
            # inflate kernel (kmm_map_bgzf_hint_next); the next call passes comp[used:] + next_chunk as one view
            nx = _Arg(next_chunk, np.uint8, "next_chunk")
            _lib.check(_lib.lib().kmm_map_bgzf_hint_next(self._h, nx.ptr, nx.n))
        if head_skip:
            self.set_param("bgzf_head_skip", int(head_skip))
        if tail_stop is not None:
            self.set_param("bgzf_tail_stop", int(tail_stop))
        b = _Arg(comp, np.uint8, "comp")
        t = _Arg(lut, np.uint8, "lut")
        n = b.n if n_bytes is None else int(n_bytes)
        if n > b.n:
            raise ValueError("n_bytes exceeds the buffer")
        used = ctypes.c_int64(0)
        n_rec = ctypes.c_int64(0)
        flags = (_lib.FORMAT_NEW_STREAM if first else 0) | (_lib.FORMAT_LAST_CHUNK if last else 0)
        _lib.check(_lib.lib().kmm_map_bgzf(self._h, b.ptr, n, int(fmt) | flags, int(k), int(max_index_lookup_frequency),
                                           int(bool(also_revcomp)), t.ptr, ctypes.byref(used), ctypes.byref(n_rec)))
        return used.value, n_rec.value

    def map_packed(self, codes, n_bases, n_reads, read_len=0, read_starts=None, k=31, max_index_lookup_frequency=1000,
                   also_revcomp=False):
        """Reads held as 2-bit codes (uint32 words, 16 codes per word, first base lowest): kmm_map_packed.  read_len > 0:
        n_reads reads of one length; else `read_starts` = uint32 bitset over the base positions."""
        c = _Arg(codes, np.uint32, "codes")
        st = _Arg(read_starts, np.uint32, "read_starts")
        if c.n < (int(n_bases) + 15) // 16:
            raise ValueError("codes holds %d words, need %d" % (c.n, (int(n_bases) + 15) // 16))
        if not read_len and st.n < int(n_bases) // 32 + 1:
            raise ValueError("read_starts needs n_bases / 32 + 1 words")
        _lib.check(_lib.lib().kmm_map_packed(self._h, c.ptr, int(n_bases), int(n_reads), int(read_len), st.ptr, int(k),
                                             int(max_index_lookup_frequency), int(bool(also_revcomp))))

    def in_index(self, kmers):
        a = _Arg(kmers, np.uint64, "kmers")
        out = np.zeros(a.n, dtype=np.uint8)
        _lib.check(_lib.lib().kmm_in_index(self._h, a.ptr, a.n, out.ctypes.data_as(_P)))
        return out

    # -- measurement -----------------------------------------------------------------------------
    def set_timing(self, on=True):
        _lib.check(_lib.lib().kmm_set_timing(self._h, int(bool(on))))

    def get_timing(self, kernel_id=None):
        """(milliseconds, launches) of one kernel id since the last call, or a dict over all ids."""
        if kernel_id is None:
            return {name: self.get_timing(i) for i, name in enumerate(_lib.KERNEL_NAMES)}
        ms = ctypes.c_double(0.0)
        n = ctypes.c_int64(0)
        _lib.check(_lib.lib().kmm_get_timing(self._h, int(kernel_id), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def get_stats(self, reset=False):
        """(k-mer lookups performed, count increments) since creation / the last reset."""
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _lib.check(_lib.lib().kmm_get_stats(self._h, int(bool(reset)), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def set_param(self, name, value):
        """Tuning knobs of include/kmm.h: "path" (0 auto, 1 direct, 2 radix), "part_shift", "radix_min_units",
        "count_kmers"."""
        _lib.check(_lib.lib().kmm_set_param(self._h, name.encode(), int(value)))

    def get_param(self, name):
        v = ctypes.c_int64(0)
        _lib.check(_lib.lib().kmm_get_param(self._h, name.encode(), ctypes.byref(v)))
        return v.value


def extract_kmers(bases, read_offsets, k, lut=None, device=0, out=None):
    """Operator form of util.py:71-75 on the GPU: flat uint64 k-mers in (read, offset) order.
    bases / read_offsets may be numpy arrays or torch tensors (host or device).  With `out` (a
    uint64 numpy array or an int64/uint64 torch tensor of the right length, host or device) the k-mers
    are written there; otherwise a numpy array is returned."""
    b = _Arg(bases, np.uint8, "bases")
    o = _Arg(read_offsets if _is_torch_tensor(read_offsets)
             else np.ascontiguousarray(np.asarray(read_offsets, dtype=np.int64)), np.int64, "read_offsets")
    t = _Arg(lut, np.uint8, "lut")
    if out is None:
        offs = o.keep.cpu().numpy() if _is_torch_tensor(o.keep) else o.keep
        n_out = int(np.maximum(np.diff(offs) - int(k) + 1, 0).sum())
        out = np.empty(n_out, dtype=np.uint64)
    dst = _Arg(out, np.uint64, "out")
    _lib.check(_lib.lib().kmm_extract_kmers(int(device), b.ptr, o.ptr, o.n - 1, int(k), t.ptr, dst.ptr, dst.n))
    return out


def build_index_device(kmers, nodes, modulo, device=0):
    """kmm_build_index on torch DEVICE tensors (kmers int64/uint64 bit patterns, nodes int32), outputs left on the
    device: (hashes_to_index int32[M], n_kmers int32[M], kmers int64[n], nodes int32[n], frequencies uint16[n]).
    For indexes too large to round-trip through host memory quickly (a 10^9-k-mer index is 30 GB of arrays)."""
    import torch
    n, M = kmers.numel(), int(modulo)
    assert kmers.is_cuda and nodes.is_cuda and nodes.dtype == torch.int32 and kmers.element_size() == 8
    dev = kmers.device
    h2i = torch.empty(M, dtype=torch.int32, device=dev)
    nk = torch.empty(M, dtype=torch.int32, device=dev)
    ko = torch.empty(n, dtype=torch.int64, device=dev)
    no = torch.empty(n, dtype=torch.int32, device=dev)
    fo = torch.empty(n, dtype=torch.uint16, device=dev)
    p = lambda t: _P(t.data_ptr())
    _lib.check(_lib.lib().kmm_build_index(int(device), p(kmers), p(nodes), n, M, p(h2i), p(nk), p(ko), p(no), p(fo)))
    return h2i, nk, ko, no, fo


def build_index(kmers, nodes, modulo, device=0):
    """GPU counterpart of graph_kmer_index's KmerIndex.from_flat_kmers (tests/test_mapping.py:36-38):
    returns (hashes_to_index int32[M], n_kmers int32[M], kmers uint64[n], nodes int32[n],
    frequencies uint16[n]) — bit-identical to the stable-sort numpy construction."""
    km = np.ascontiguousarray(np.asarray(kmers, dtype=np.uint64))
    nd_in = np.asarray(nodes)
    if nd_in.size and (nd_in.min() < 0 or nd_in.max() > 2 ** 31 - 1):
        raise ValueError("node ids must fit int32")
    nd = np.ascontiguousarray(nd_in, dtype=np.int32)
    if km.shape != nd.shape or km.ndim != 1:
        raise ValueError("kmers and nodes must be 1-D arrays of the same length")
    M, n = int(modulo), km.shape[0]
    h2i = np.empty(M, dtype=np.int32)
    nk = np.empty(M, dtype=np.int32)
    ko = np.empty(n, dtype=np.uint64)
    no = np.empty(n, dtype=np.int32)
    fo = np.empty(n, dtype=np.uint16)
    p = lambda a: a.ctypes.data_as(_P)
    _lib.check(_lib.lib().kmm_build_index(int(device), p(km), p(nd), n, M, p(h2i), p(nk), p(ko), p(no), p(fo)))
    return h2i, nk, ko, no, fo
