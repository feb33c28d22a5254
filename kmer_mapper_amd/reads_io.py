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
        return gzip.open(path, "rb")
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


def read_chunks(path, min_chunk_size=2_500_000):
    """Yield ReadBatch objects of ~min_chunk_size file bytes each, in file order."""
    f = _open(path)
    try:
        carry = np.zeros(0, dtype=np.uint8)
        fmt = None
        while True:
            raw = f.read(int(min_chunk_size))
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
            cut = _last_record_boundary(buf, fmt, at_eof)
            if cut:
                part = buf[:cut]
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

    def __init__(self, path, chunk_size):
        self.f = _open(path)
        self.chunk_size = int(chunk_size)
        self.buf = np.empty(self.chunk_size + (1 << 20), dtype=np.uint8)
        self.fill = 0
        self.eof = False

    def next_chunk(self):
        """Returns a uint8 view (valid until the next call) or None at end of input."""
        if self.eof and self.fill == 0:
            return None
        if self.fill == self.buf.shape[0]:          # one record larger than the buffer: grow
            self.buf = np.concatenate([self.buf, np.empty_like(self.buf)])
        while not self.eof and self.fill < min(self.chunk_size, self.buf.shape[0]):
            got = self.f.readinto(memoryview(self.buf)[self.fill:])
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
