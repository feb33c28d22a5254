"""Chunked FASTA / FASTQ (+ .gz) reader feeding the GPU — the job `bnp.open(args.reads).read_chunks(
min_chunk_size=args.chunk_size)` does in the reference (kmer_mapper/command_line_interface.py:102-103,
109-111; Readme.md:11: ".fa, .fq, .fa.gz, or fq.gz").  bionumpy is not a dependency here.

A chunk is cut after the last complete record inside ~chunk_size bytes of file (chunk_size counts FILE
bytes, command_line_interface.py:169-170) and comes out as a ReadBatch: flat sequence bytes + int64 read
offsets, exactly what kmm_map_reads takes.  Parsing is vectorised numpy over the raw byte buffer (newline
scan, header/sequence line masks); no per-read Python loop.
"""
import gzip
import queue
import threading

import numpy as np

from .util import ReadBatch

_NL = 10
_CR = 13


def _open(path):
    if str(path).endswith(".gz"):
        from .gz_io import open_gz
        return open_gz(path)            # BGZF members are inflated on several cores, plain gzip on one
    from . import _io
    if _io.available():
        return _io.NativeStream(path)   # parallel pread into the caller's (pinned) buffer
    return open(path, "rb", buffering=0)


def _detect_format(first_byte, path):
    if first_byte == ord(">"):
        return "fasta"
    if first_byte == ord("@"):
        return "fastq"
    name = str(path).lower()
    for ext in (".gz",):
        if name.endswith(ext):
            name = name[: -len(ext)]
    if name.endswith((".fq", ".fastq")):
        return "fastq"
    if name.endswith((".fa", ".fasta", ".fna")):
        return "fasta"
    raise ValueError("cannot tell FASTA from FASTQ: %s starts with byte %r" % (path, first_byte))


def _strip_cr(buf, starts, ends):
    """Line ends without a trailing carriage return (Windows line endings)."""
    nonempty = ends > starts
    last = np.where(nonempty, buf[np.maximum(ends - 1, 0)], 0)
    return ends - (last == _CR)


def _gather_lines(buf, starts, ends):
    """Concatenate buf[starts[i]:ends[i]] for all i -> (flat bytes, lengths)."""
    lens = (ends - starts).astype(np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.uint8), lens
    out_off = np.zeros(lens.shape[0], dtype=np.int64)
    np.cumsum(lens[:-1], out=out_off[1:])
    # index of every output byte in buf: start of its line + position inside the line
    idx = np.repeat(starts - out_off, lens) + np.arange(total, dtype=np.int64)
    return buf[idx], lens


def parse_fastq_block(buf):
    """buf: uint8 array holding whole 4-line records.  Returns ReadBatch."""
    nl = np.flatnonzero(buf == _NL)
    n_lines = nl.shape[0]
    if n_lines % 4:
        raise ValueError("FASTQ block does not hold a whole number of 4-line records")
    n = n_lines // 4
    line_start = np.empty(n_lines, dtype=np.int64)
    line_start[0:1] = 0
    line_start[1:] = nl[:-1] + 1
    s = line_start[1::4]
    e = _strip_cr(buf, s, nl[1::4])
    if n and not (np.all(buf[line_start[0::4]] == ord("@")) and np.all(buf[line_start[2::4]] == ord("+"))):
        raise ValueError("malformed FASTQ: record does not start with '@' / third line is not '+'")
    bases, lens = _gather_lines(buf, s, e)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    return ReadBatch(bases, offsets)


def parse_fasta_block(buf):
    """buf: uint8 array of whole FASTA records (header line + one or more sequence lines)."""
    if buf.shape[0] == 0:
        return ReadBatch(np.zeros(0, np.uint8), np.zeros(1, np.int64))
    if buf[-1] != _NL:
        buf = np.concatenate([buf, np.array([_NL], dtype=np.uint8)])
    nl = np.flatnonzero(buf == _NL)
    line_start = np.empty(nl.shape[0], dtype=np.int64)
    line_start[0:1] = 0
    line_start[1:] = nl[:-1] + 1
    is_header = buf[line_start] == ord(">")
    if not is_header[0]:
        raise ValueError("malformed FASTA: block does not start with '>'")
    rec_of_line = np.cumsum(is_header) - 1
    seq = ~is_header
    s = line_start[seq]
    e = _strip_cr(buf, s, nl[seq])
    bases, lens = _gather_lines(buf, s, e)
    n = int(is_header.sum())
    per_read = np.bincount(rec_of_line[seq], weights=lens, minlength=n).astype(np.int64)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(per_read, out=offsets[1:])
    return ReadBatch(bases, offsets)


def _last_record_boundary(buf, fmt, at_eof):
    """Number of leading bytes of buf that form whole records."""
    if at_eof:
        return buf.shape[0]
    nl = np.flatnonzero(buf == _NL)
    if fmt == "fastq":
        whole = (nl.shape[0] // 4) * 4
        return int(nl[whole - 1]) + 1 if whole else 0
    # FASTA: the last record is complete only once the next header has been seen
    line_start = np.empty(nl.shape[0] + 1, dtype=np.int64)
    line_start[0] = 0
    line_start[1:] = nl + 1
    line_start = line_start[line_start < buf.shape[0]]
    headers = line_start[buf[line_start] == ord(">")]
    return int(headers[-1]) if headers.shape[0] > 1 else 0


def read_chunks(path, min_chunk_size=2_500_000, byte_range=None, owned=None):
    """Yield ReadBatch objects of ~min_chunk_size file bytes each, in file order.  byte_range = (lo, hi):
    only that part of an uncompressed file (both ends record starts, rank_byte_range).  owned(i) -> bool:
    chunks this rank does not own are cut at the record boundary (a look at their last lines) but not parsed,
    and come out as None (ranks sharing one .gz stream)."""
    f = _open(path)
    left = None
    if byte_range is not None:
        f.seek(byte_range[0])
        left = byte_range[1] - byte_range[0]
    try:
        carry = np.zeros(0, dtype=np.uint8)
        fmt = None
        n_chunks = 0
        while True:
            want = int(min_chunk_size) if left is None else min(int(min_chunk_size), left)
            raw = f.read(want) if want else b""
            if left is not None:
                left -= len(raw)
            at_eof = len(raw) == 0
            if at_eof and carry.shape[0] == 0:
                return
            block = np.frombuffer(raw, dtype=np.uint8)
            buf = np.concatenate([carry, block]) if carry.shape[0] else block
            if fmt is None:
                if buf.shape[0] == 0:
                    return
                fmt = _detect_format(int(buf[0]), path)
            if at_eof and fmt == "fastq" and buf[-1] != _NL:
                buf = np.concatenate([buf, np.array([_NL], dtype=np.uint8)])
            mine = owned is None or owned(n_chunks)
            if mine:
                cut = _last_record_boundary(buf, fmt, at_eof)
            else:
                cut = buf.shape[0] if at_eof else last_record_start(buf, fmt)
            if cut:
                part = buf[:cut]
                n_chunks += 1
                if not mine:
                    yield None
                else:
                    yield parse_fastq_block(part) if fmt == "fastq" else parse_fasta_block(part)
            carry = buf[cut:].copy()
            if at_eof:
                return
    finally:
        f.close()


def prefetch(iterator, depth=3):
    """Run `iterator` in a background thread so that parsing/inflating overlaps the GPU."""
    q = queue.Queue(maxsize=depth)
    _END = object()

    def work():
        try:
            for item in iterator:
                q.put(item)
            q.put(_END)
        except BaseException as exc:      # surfaced in the consumer
            q.put(exc)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    while True:
        item = q.get()
        if item is _END:
            return
        if isinstance(item, BaseException):
            raise item
        yield item


def write_fasta(path, batch, gz=False, line_width=None):
    """Test helper: write a ReadBatch as FASTA."""
    opener = gzip.open if gz else open
    with opener(path, "wb") as f:
        for i in range(len(batch)):
            seq = batch.bases[batch.offsets[i]:batch.offsets[i + 1]].tobytes()
            f.write(b">r%d\n" % i)
            if line_width:
                for j in range(0, max(len(seq), 1), line_width):
                    f.write(seq[j:j + line_width] + b"\n")
            else:
                f.write(seq + b"\n")


def write_fastq(path, batch, gz=False):
    opener = gzip.open if gz else open
    with opener(path, "wb") as f:
        for i in range(len(batch)):
            seq = batch.bases[batch.offsets[i]:batch.offsets[i + 1]].tobytes()
            f.write(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * len(seq) + b"\n")


# ------------------------------------------------------------------------------------------------
# Byte-range sharding of a read file over ranks (multi-GPU file input).  The reference hands each chunk to one
# worker (command_line_interface.py:109-111); here rank g owns the records that START inside
# [g*size/G, (g+1)*size/G), found by re-synchronising to the record structure at both ends of the range, so
# that no rank ever reads (or scans) bytes outside its own range.
# ------------------------------------------------------------------------------------------------
def _is_record_start(buf, starts, i, fmt, at_eof):
    """Is line i (starts[i] = offset of its first byte in buf; starts[-1] = len(buf) sentinel when the buffer ends
    with a newline) the first line of a record?  Returns True / False, or None if the buffer holds too few lines
    after it to tell."""
    n_lines = len(starts) - 1                      # complete lines in buf
    if i >= n_lines:
        return None if not at_eof else False
    if fmt != "fastq":                             # FASTA: '>' never starts a sequence line
        return bool(buf[starts[i]] == ord(">"))
    if buf[starts[i]] != ord("@"):
        return False
    if i + 4 > n_lines:                            # need the whole record: header, sequence, '+', quality
        return None if not at_eof else False

    def line_len(j):
        e = starts[j + 1] - 1                      # position of the newline
        if e > starts[j] and buf[e - 1] == _CR:
            e -= 1
        return e - starts[j]
    # a quality line may start with '@' too, but then the line two below it is a sequence, never '+'
    if buf[starts[i + 2]] != ord("+") or line_len(i + 1) != line_len(i + 3):
        return False
    if i + 4 < n_lines and buf[starts[i + 4]] != ord("@"):
        return False
    return True


def find_record_start(path, pos, fmt, size=None, window=1 << 16):
    """Offset of the first record of an UNCOMPRESSED FASTA/FASTQ file that starts at or after byte `pos`
    (`size` if there is none)."""
    import os
    size = os.stat(path).st_size if size is None else size
    if pos <= 0:
        return 0
    if pos >= size:
        return size
    with open(path, "rb", buffering=0) as f:
        while True:
            f.seek(pos - 1)                       # one byte early: `pos` is a line start iff byte pos-1 is a newline
            raw = f.read(window)
            at_eof = pos - 1 + len(raw) >= size
            buf = np.frombuffer(raw, dtype=np.uint8)
            if at_eof and buf[-1] != _NL:
                buf = np.concatenate([buf, np.array([_NL], dtype=np.uint8)])
            starts = (np.flatnonzero(buf == _NL) + 1).astype(np.int64)   # last one: sentinel or an unfinished line
            for i in range(len(starts) - 1):
                r = _is_record_start(buf, starts, i, fmt, at_eof)
                if r is None:
                    break
                if r:
                    return pos - 1 + int(starts[i])
            if at_eof:
                return size
            window *= 2                               # nothing decided yet (long records): look further


def rank_byte_range(path, fmt, rank, world_size):
    """[lo, hi) of the uncompressed file `path` whose records belong to `rank`: both ends are record starts, the
    ranges of ranks 0..world_size-1 partition the file exactly."""
    import os
    size = os.stat(path).st_size
    lo = find_record_start(path, size * rank // world_size, fmt, size)
    hi = find_record_start(path, size * (rank + 1) // world_size, fmt, size)
    return lo, hi


def last_record_start(buf, fmt):
    """Number of leading bytes of the uint8 buffer `buf` (which starts at a record start) that form whole
    records — found by looking at the last few lines only: ranks that merely skip a chunk of a .gz stream need
    nothing else.  FASTA: a record counts as whole once the next header has been seen."""
    n = buf.shape[0]
    back = 1 << 12
    while True:
        lo = max(n - back, 0)
        nl = np.flatnonzero(buf[lo:] == _NL) + lo
        starts = np.concatenate([np.zeros(1 if lo == 0 else 0, dtype=np.int64), nl + 1]).astype(np.int64)
        if starts.shape[0] == 0 or starts[-1] != n:
            starts = np.concatenate([starts, [n]])           # sentinel; the line before it is unfinished ...
            n_lines = len(starts) - 2                          # ... so it is no complete line
        else:
            n_lines = len(starts) - 1
        view = starts[:n_lines + 1]
        if fmt != "fastq":
            cand = starts[(starts > 0) & (starts < n)]          # an unfinished header line is a record start too
            cand = cand[buf[cand] == ord(">")]
            if cand.shape[0]:
                return int(cand[-1])
            if lo == 0:
                return 0
        else:
            for i in range(n_lines - 4, -1, -1):
                if _is_record_start(buf, view, i, "fastq", True):
                    return int(view[i + 4])
            if lo == 0:
                return 0
        back *= 4


def records_cut(buf, fmt, at_eof=False):
    """Number of leading bytes of `buf` that kmm_map_records would consume: the byte after the last newline whose
    1-based count is a multiple of the record's line count (4 for FASTQ, 2 for two-line FASTA) — the rule of the GPU
    record parser (csrc/kmm_records.hpp, k_rec_scan2: target = total - total % period).  Ranks that skip a chunk of
    a shared .gz stream MUST cut it with this rule: the owner advances by the parser's `consumed`, and a rule that
    holds a FASTA record back until the next '>' has been seen (last_record_start) disagrees whenever a chunk ends
    exactly on a record's last newline — the stream positions of the ranks would drift apart."""
    if fmt == "fasta_ml":          # multi-line FASTA: whole records end where the chunk's last header line starts
        if at_eof:
            return int(buf.shape[0])
        nl = np.flatnonzero(buf == _NL)
        starts = nl[nl + 1 < buf.shape[0]] + 1
        hdr = starts[buf[starts] == ord(">")]
        return int(hdr[-1]) if hdr.shape[0] else 0
    period = 4 if fmt == "fastq" else 2
    nl = np.flatnonzero(buf == _NL)
    whole = (nl.shape[0] // period) * period
    return int(nl[whole - 1]) + 1 if whole else 0


# ------------------------------------------------------------------------------------------------
# Raw-chunk reader for the GPU record parser (kmm_map_records): the host only moves bytes.
# ------------------------------------------------------------------------------------------------
def sniff_format(path, probe_bytes=1 << 16):
    """Returns ("fastq" | "fasta", gpu_parsable).  Two-line FASTA and 4-line FASTQ can be parsed on
    the GPU; FASTA whose sequences are wrapped over several lines needs the host parser."""
    with _open(path) as f:
        head = f.read(probe_bytes)
    if not head:
        return "fasta", True
    fmt = _detect_format(head[0], path)
    if fmt == "fastq":
        return fmt, True
    lines = head.split(b"\n")
    complete = lines[:-1] if len(lines) > 1 else lines
    ok = all((ln[:1] == b">") == (i % 2 == 0) for i, ln in enumerate(complete) if ln or i % 2 == 0)
    return fmt, ok


class RawChunker:
    """Feeds successive raw file chunks of ~chunk_size bytes to a consumer that reports how many
    bytes it used (the end of the last complete record); the unused tail is carried over."""

    def __init__(self, path, chunk_size, byte_range=None, pinned=False):
        """byte_range = (lo, hi): only that part of an uncompressed file (both ends record starts,
        rank_byte_range).  pinned: the buffer is page-locked host memory (kmm_host_alloc), so the staging copy of
        kmm_map_records runs at the PCIe link's rate."""
        self.f = _open(path)
        self.chunk_size = int(chunk_size)
        self._pinned = None
        n = self.chunk_size + (1 << 20)
        if pinned:
            import ctypes
            from . import _lib
            p = ctypes.c_void_p()
            _lib.check(_lib.lib().kmm_host_alloc(n, ctypes.byref(p)))
            self._pinned = p
            self.buf = np.frombuffer((ctypes.c_uint8 * n).from_address(p.value), dtype=np.uint8)
        else:
            self.buf = np.empty(n, dtype=np.uint8)
        self.fill = 0
        self.eof = False
        self.left = None
        if byte_range is not None:
            self.f.seek(byte_range[0])
            self.left = byte_range[1] - byte_range[0]

    def next_chunk(self):
        """Returns a uint8 view (valid until the next call) or None at end of input."""
        if self.eof and self.fill == 0:
            return None
        if self.fill == self.buf.shape[0]:          # one record larger than the buffer: grow
            self.buf = np.concatenate([self.buf, np.empty_like(self.buf)])
        while not self.eof and self.fill < min(self.chunk_size, self.buf.shape[0]):
            want = memoryview(self.buf)[self.fill:]
            if self.left is not None:
                want = want[:min(len(want), self.left)]
            got = self.f.readinto(want) if len(want) else 0
            if self.left is not None:
                self.left -= got or 0
            if not got:
                self.eof = True
                if self.fill and self.buf[self.fill - 1] != _NL:     # last line without newline
                    if self.fill == self.buf.shape[0]:
                        self.buf = np.concatenate([self.buf, np.empty(16, np.uint8)])
                    self.buf[self.fill] = _NL
                    self.fill += 1
                break
            self.fill += got
        return self.buf[:self.fill] if self.fill else None

    def consumed(self, n):
        rest = self.fill - n
        if rest:
            self.buf[:rest] = self.buf[n:self.fill]      # numpy handles the overlap
        self.fill = rest

    def close(self):
        self.f.close()
        if self._pinned is not None:
            from . import _lib
            self.buf = np.empty(0, dtype=np.uint8)          # drop the view before the memory goes away
            _lib.lib().kmm_host_free(self._pinned)
            self._pinned = None


class MmapChunker:
    """RawChunker's interface over a FILE MAPPING of an uncompressed read file: next_chunk() hands out views of the page
    cache itself, nothing is copied on the way to kmm_map_records — whose host threads read the sequence lines straight
    from the mapping and pack them to 2 bits per base ("host_pack_threads") — and nothing needs to be page-locked.
    The unused tail of a chunk simply stays where it is: the next chunk starts there."""

    def __init__(self, path, chunk_size, byte_range=None, pinned=False):
        import mmap
        import os
        self.chunk_size = int(chunk_size)
        self._f = open(path, "rb")
        size = os.fstat(self._f.fileno()).st_size
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if size else None
        if self._mm is not None and hasattr(self._mm, "madvise") and hasattr(mmap, "MADV_SEQUENTIAL"):
            self._mm.madvise(mmap.MADV_SEQUENTIAL)
        self._all = np.frombuffer(self._mm, dtype=np.uint8) if self._mm is not None else np.zeros(0, np.uint8)
        self.pos, self.hi = (0, size) if byte_range is None else (int(byte_range[0]), int(byte_range[1]))
        self.eof = False
        self._tail = None            # the file's last bytes + the newline its last line lacks
        self._populators = []

    def populate(self, n_threads=4, limit=16 << 30, piece=64 << 20):
        """Map the range's pages AHEAD of the packer threads (madvise MADV_POPULATE_READ, Linux >= 5.14), from helper
        threads, while the caller does something else (the CLI uploads the index meanwhile): the packer's threads then
        read the mapping without taking a page fault per 64 KiB.  Pieces are handed out in file order, so the first
        batch's pages come first.  Best effort: an older kernel refuses the advice and nothing changes."""
        import ctypes
        import threading
        if self._mm is None or self.hi <= self.pos or self._populators:
            return
        try:
            libc = ctypes.CDLL(None, use_errno=True)
            madvise = libc.madvise
        except (OSError, AttributeError):
            return
        madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        madvise.restype = ctypes.c_int
        base = self._all.ctypes.data                       # page-aligned: the mapping starts at offset 0 of the file
        lo = self.pos & ~4095
        hi = min(self.hi, lo + int(limit))
        pieces = [(a, min(a + piece, hi)) for a in range(lo, hi, piece)]
        nxt = [0]
        lock = threading.Lock()

        import time
        self.populate_t0 = time.perf_counter()
        self.populate_t1 = None                              # when the last helper finished (perf_counter)

        def work():
            try:
                while True:
                    with lock:
                        j = nxt[0]
                        nxt[0] += 1
                    if j >= len(pieces):
                        return
                    a, b = pieces[j]
                    if madvise(base + a, b - a, 22) != 0:    # 22 = MADV_POPULATE_READ (ctypes releases the GIL for the call)
                        return
            finally:
                self.populate_t1 = time.perf_counter()

        for _ in range(max(1, min(int(n_threads), len(pieces)))):
            t = threading.Thread(target=work, daemon=True)
            t.start()
            self._populators.append(t)

    def page_nodes(self, samples=48):
        """{NUMA node: sampled pages} of the range's page-cache pages (get_mempolicy(MPOL_F_NODE | MPOL_F_ADDR), x86-64 syscall
        239, on pages of the mapping): where the file lies decides where its packer threads should run
        (distributed.packer_cpus_near).  {} when the platform does not say."""
        import ctypes
        import platform
        if self._mm is None or self.hi <= self.pos or platform.machine() != "x86_64":
            return {}
        try:
            libc = ctypes.CDLL(None, use_errno=True)
        except OSError:
            return {}
        base = self._all.ctypes.data
        lo, hi = self.pos & ~4095, self.hi
        step = max(4096, ((hi - lo) // max(1, int(samples))) & ~4095)
        out = {}
        for off in range(lo, hi, step):
            _ = int(self._all[off])                            # (mapped before it is asked about)
            node = ctypes.c_int(-1)
            if libc.syscall(239, ctypes.byref(node), None, ctypes.c_ulong(0), ctypes.c_void_p(base + off), ctypes.c_ulong(3)) == 0:
                out[int(node.value)] = out.get(int(node.value), 0) + 1
        return out

    def wait_populated(self):
        for t in self._populators:
            t.join()
        self._populators = []

    def next_chunk(self):
        if self._tail is not None:
            return self._tail if self._tail.shape[0] else None
        if self.pos >= self.hi:
            self.eof = True
            return None
        end = min(self.pos + self.chunk_size, self.hi)
        self.eof = end == self.hi
        view = self._all[self.pos:end]
        if self.eof and view[-1] != _NL:          # a last line without its newline gets one (a copy of this last chunk)
            self._tail = np.concatenate([view, np.array([_NL], dtype=np.uint8)])
            return self._tail
        return view

    def consumed(self, n):
        if self._tail is not None:
            self._tail = self._tail[n:]
            self.pos = self.hi if not self._tail.shape[0] else self.pos
            return
        self.pos += int(n)

    def close(self):
        self.wait_populated()        # (no helper may touch the mapping once it goes)
        self._all = np.zeros(0, np.uint8)
        self._tail = None
        try:
            if self._mm is not None:
                self._mm.close()
        except BufferError:          # a view handed out earlier is still alive somewhere: the mapping goes with it
            pass
        self._f.close()


class PrefetchingRawChunker:
    """RawChunker's interface with TWO buffers and a reader thread: while the consumer (kmm_map_records: staging copy over
    PCIe, record parser, map kernels) works on one chunk, the next one is read — and, for .gz input, inflated by the
    native reader's threads — into the other buffer.  The unused tail of a chunk (at most one record, as a rule) is
    copied in front of the next chunk's bytes, into head room kept for it; anything out of the ordinary (a record longer
    than the head room, a consumer that used nothing) ends the prefetching and continues on one growing buffer exactly
    as RawChunker does.  (`kmer_mapper map` on BGZF input: inflate and the GPU's share no longer take turns.)"""

    HEAD = 4 << 20

    def __init__(self, path, chunk_size, byte_range=None, pinned=False):
        import threading
        self.f = _open(path)
        self.chunk_size = int(chunk_size)
        self._pinned = []
        self._bufs = []
        n = self.HEAD + self.chunk_size + (1 << 20)
        for _ in range(2):
            if pinned:
                import ctypes
                from . import _lib
                p = ctypes.c_void_p()
                _lib.check(_lib.lib().kmm_host_alloc(n, ctypes.byref(p)))
                self._pinned.append(p)
                self._bufs.append(np.frombuffer((ctypes.c_uint8 * n).from_address(p.value), dtype=np.uint8))
            else:
                self._bufs.append(np.empty(n, dtype=np.uint8))
        self.left = None
        if byte_range is not None:
            self.f.seek(byte_range[0])
            self.left = byte_range[1] - byte_range[0]
        self.eof = False                 # the chunk handed out last holds the input's final bytes
        self._src_eof = False            # the reader has seen the end of the input
        self._cur = 0                    # buffer the consumer works on
        self._start = self.HEAD          # its first byte (head room holds the carried-over tail)
        self._end = self.HEAD            # ... one past its last byte
        self._next_fill = 0              # bytes the reader put into the other buffer
        self._next_eof = False
        self._carry = 0                  # carried-over bytes in front of the other buffer's data
        self._fallback = None            # a plain RawChunker-like state once prefetching has ended
        self._err = None
        self._lock = threading.Lock()
        self._thread = None
        self._first = True
        self._handed = False
        self._start_read(self._cur)      # the very first chunk is read in the background too; next_chunk waits for it

    # --- reader thread: fills buffer j from HEAD on, up to chunk_size bytes or the end of the input
    def _read_into(self, j):
        try:
            buf = self._bufs[j]
            fill, eof = 0, self._src_eof
            while not eof and fill < self.chunk_size:
                want = memoryview(buf)[self.HEAD + fill:self.HEAD + self.chunk_size]
                if self.left is not None:
                    want = want[:min(len(want), self.left)]
                got = self.f.readinto(want) if len(want) else 0
                if self.left is not None:
                    self.left -= got or 0
                if not got:
                    eof = True
                    break
                fill += got
            self._next_fill, self._next_eof = fill, eof
            self._src_eof = eof
        except BaseException as exc:     # noqa: BLE001 - handed to the consumer's thread
            self._err = exc

    def _start_read(self, j):
        import threading
        self._thread = threading.Thread(target=self._read_into, args=(j,), daemon=True)
        self._thread.start()

    def _join(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._err is not None:
            err, self._err = self._err, None
            raise err

    def next_chunk(self):
        """Returns a uint8 view (valid until the next call) or None at end of input."""
        if self._fallback is not None:
            return self._fb_next()
        if self._handed:                                 # asked again without consumed(): the consumer could not use
            self._handed = False                         # the chunk (a record longer than it) and wants a longer one
            return self._to_fallback_and_next()
        if self._first:
            self._first = False
            self._join()
            self._end = self.HEAD + self._next_fill
            self.eof = self._next_eof
            self._finish_tail()
            if not self.eof:
                self._start_read(1 - self._cur)
        if self._end == self._start:
            return None if self.eof else self._to_fallback_and_next()
        self._handed = True
        return self._bufs[self._cur][self._start:self._end]

    def _finish_tail(self):
        """A last line without its newline gets one (as RawChunker does)."""
        if self.eof and self._end > self._start and self._bufs[self._cur][self._end - 1] != _NL:
            self._bufs[self._cur][self._end] = _NL       # (1 MiB of slack behind chunk_size)
            self._end += 1

    def consumed(self, n):
        if self._fallback is not None:
            return self._fb_consumed(n)
        self._handed = False
        rest = (self._end - self._start) - n
        if self.eof:                                     # nothing more will come: the rest stays in this buffer
            self._start += n
            return
        if n == 0 or rest > self.HEAD:                   # out of the ordinary: continue on one growing buffer
            self._start += n
            self._to_fallback()
            return
        other = 1 - self._cur
        if rest:
            self._bufs[other][self.HEAD - rest:self.HEAD] = self._bufs[self._cur][self._start + n:self._end]
        self._join()                                     # the other buffer's bytes are in
        self._cur = other
        self._start = self.HEAD - rest
        self._end = self.HEAD + self._next_fill
        self.eof = self._next_eof
        self._finish_tail()
        if not self.eof:
            self._start_read(1 - self._cur)

    # --- fallback: everything that is left, on one growing pageable buffer (RawChunker's loop)
    def _to_fallback(self):
        self._join()
        cur = self._bufs[self._cur][self._start:self._end]
        nxt = self._bufs[1 - self._cur][self.HEAD:self.HEAD + self._next_fill] if not self.eof else cur[:0]
        buf = np.empty(max(2 * (len(cur) + len(nxt)), self.chunk_size + (1 << 20)), dtype=np.uint8)
        buf[:len(cur)] = cur
        buf[len(cur):len(cur) + len(nxt)] = nxt
        self._fallback = {"buf": buf, "fill": len(cur) + len(nxt)}
        self.eof = self._src_eof

    def _to_fallback_and_next(self):
        self._to_fallback()
        return self._fb_next()

    def _fb_next(self):
        st = self._fallback
        if self.eof and st["fill"] == 0:
            return None
        if st["fill"] == st["buf"].shape[0]:
            st["buf"] = np.concatenate([st["buf"], np.empty_like(st["buf"])])
        while not self.eof and st["fill"] < min(self.chunk_size, st["buf"].shape[0]):
            want = memoryview(st["buf"])[st["fill"]:]
            if self.left is not None:
                want = want[:min(len(want), self.left)]
            got = self.f.readinto(want) if len(want) else 0
            if self.left is not None:
                self.left -= got or 0
            if not got:
                self.eof = True
                if st["fill"] and st["buf"][st["fill"] - 1] != _NL:
                    if st["fill"] == st["buf"].shape[0]:
                        st["buf"] = np.concatenate([st["buf"], np.empty(16, np.uint8)])
                    st["buf"][st["fill"]] = _NL
                    st["fill"] += 1
                break
            st["fill"] += got
        return st["buf"][:st["fill"]] if st["fill"] else None

    def _fb_consumed(self, n):
        st = self._fallback
        rest = st["fill"] - n
        if rest:
            st["buf"][:rest] = st["buf"][n:st["fill"]]
        st["fill"] = rest

    def close(self):
        try:
            self._join()
        except BaseException:            # noqa: BLE001 - closing: the consumer's error (if any) is the one to report
            pass
        self.f.close()
        if self._pinned:
            from . import _lib
            self._bufs = []
            for p in self._pinned:
                _lib.lib().kmm_host_free(p)
            self._pinned = []
