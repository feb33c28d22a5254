"""Inflating .gz read files on several host cores.

The reference reads .fa.gz / .fq.gz through bionumpy (Readme.md:11) and meant to use igzip for it
(kmer_mapper/util.py:78-101, call sites commented out); neither is a dependency here.  A gzip stream made of
independent members whose sizes are stored up front — BGZF, what bgzip / htslib write — can be inflated member
by member on a pool of threads (zlib releases the GIL); anything else is one deflate stream and is inflated by
one thread, in large pieces.  Both come out as a plain forward-only byte stream with read() / readinto().
"""
import os
import struct
import zlib
from collections import deque
from concurrent.futures import ThreadPoolExecutor

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block_size(header):
    """Total size of the BGZF member that starts with the 18 bytes `header`, or 0 if it is not BGZF."""
    if len(header) < 18 or header[:4] != b"\x1f\x8b\x08\x04":
        return 0
    xlen = struct.unpack_from("<H", header, 10)[0]
    if xlen < 6 or header[12:14] != b"BC" or struct.unpack_from("<H", header, 14)[0] != 2:
        return 0
    return struct.unpack_from("<H", header, 16)[0] + 1


def is_bgzf(path):
    with open(path, "rb") as f:
        return _bgzf_block_size(f.read(18)) > 0


def _inflate_member(raw):
    """One whole BGZF member (header + deflate payload + CRC32 + ISIZE) -> bytes."""
    xlen = struct.unpack_from("<H", raw, 10)[0]
    try:
        out = zlib.decompress(raw[12 + xlen:-8], wbits=-15)
    except zlib.error as exc:
        raise ValueError("corrupt BGZF member: %s" % exc) from None
    crc, isize = struct.unpack_from("<II", raw, len(raw) - 8)
    if len(out) != isize:
        raise ValueError("BGZF member inflates to %d bytes, its trailer says otherwise" % len(out))
    if (zlib.crc32(out) & 0xFFFFFFFF) != crc:
        raise ValueError("corrupt BGZF member: CRC32 of the inflated bytes differs from the trailer")
    return out


class _ByteStream:
    """read() / readinto() / close() over a generator of byte pieces."""

    def __init__(self, pieces):
        self._pieces = pieces
        self._cur = memoryview(b"")

    def _refill(self):
        for piece in self._pieces:
            if len(piece):
                self._cur = memoryview(piece)
                return True
        return False

    def readinto(self, b):
        dst = memoryview(b).cast("B")
        done = 0
        while done < len(dst):
            if not len(self._cur) and not self._refill():
                break
            n = min(len(dst) - done, len(self._cur))
            dst[done:done + n] = self._cur[:n]
            self._cur = self._cur[n:]
            done += n
        return done

    def read(self, n=-1):
        if n is None or n < 0:
            parts = [bytes(self._cur)]
            self._cur = memoryview(b"")
            parts.extend(bytes(p) for p in self._pieces)
            return b"".join(parts)
        buf = bytearray(n)
        got = self.readinto(buf)
        return bytes(buf[:got])

    def close(self):
        close = getattr(self._pieces, "close", None)
        if close:
            close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _bgzf_pieces(path, n_threads, group_bytes=1 << 22):
    """Inflate a BGZF file: the member sizes are read from the headers (a seek per member, no inflating), members
    are grouped into tasks of ~group_bytes and inflated on a thread pool, results are yielded in file order."""
    size = os.path.getsize(path)
    pool = ThreadPoolExecutor(max_workers=n_threads)

    def task(off, sizes):
        with open(path, "rb", buffering=0) as g:
            g.seek(off)
            raw = g.read(sum(sizes))
        out, p = [], 0
        for s in sizes:
            out.append(_inflate_member(raw[p:p + s]))
            p += s
        return b"".join(out)

    pending = deque()
    try:
        with open(path, "rb") as f:
            off = 0
            while off < size or pending:
                while off < size and len(pending) < 3 * n_threads:
                    start, sizes, acc = off, [], 0
                    while off < size and acc < group_bytes:
                        f.seek(off)
                        s = _bgzf_block_size(f.read(18))
                        if not s:
                            raise ValueError("%s: not a BGZF member at byte %d" % (path, off))
                        sizes.append(s)
                        acc += s
                        off += s
                    pending.append(pool.submit(task, start, sizes))
                yield pending.popleft().result()
    finally:
        pool.shutdown(wait=False, cancel_futures=True)


def _gzip_pieces(path, piece=1 << 24):
    """One (or several concatenated) plain gzip member(s): a single thread, large reads.  Behind a member's end marker
    zero padding is skipped (as gzip.open does, the reference's reader through bnp.open); anything else must be another
    member."""
    with open(path, "rb", buffering=0) as f:
        d = zlib.decompressobj(wbits=31)
        fed = False                            # has the current member received any input?
        between = False                        # behind a member's end marker, before the next member's first byte
        carry = b""
        while True:
            raw = f.read(piece)
            if not raw:
                if carry:                      # a lone 0x1f at the very end
                    raise ValueError("%s: trailing bytes after the gzip stream are neither zero padding nor another "
                                     "gzip member" % path)
                tail = d.flush()
                if tail:
                    yield tail
                if fed and not d.eof:          # gzip.open raises EOFError here; partial counts must never pass silently
                    raise EOFError("%s: compressed file ended before the end-of-stream marker was reached" % path)
                return
            raw, carry = carry + raw, b""
            while raw:
                if between:
                    raw = raw.lstrip(b"\0")
                    if not raw:
                        break
                    if len(raw) == 1 and raw == b"\x1f":
                        carry = raw            # the magic's second byte comes with the next read
                        break
                    if raw[:2] != b"\x1f\x8b":
                        raise ValueError("%s: trailing bytes after the gzip stream are neither zero padding nor another "
                                         "gzip member" % path)
                    between = False
                try:
                    out = d.decompress(raw)
                except zlib.error as exc:
                    raise ValueError("%s: corrupt gzip stream: %s" % (path, exc)) from None
                fed = True
                if out:
                    yield out
                if d.eof:                      # next member of a concatenated file (zlib has checked CRC32 + ISIZE)
                    raw = d.unused_data
                    d = zlib.decompressobj(wbits=31)
                    fed = False
                    between = True
                else:
                    raw = b""


def open_gz(path, n_threads=None, native=None):
    """Forward-only inflated byte stream of a .gz file; BGZF files use up to n_threads cores (default: the cores
    this process may run on, at most 16 — the reference CLI's -t default).  With libkmm_io.so built (the default)
    the native reader does the work — members inflated by C++ threads straight into the caller's buffer; native=False
    or KMM_IO_PYTHON=1 selects the pure-Python reader below."""
    from . import _io
    if native is None:
        native = _io.available()
    if native:
        return _io.NativeStream(path, n_threads)
    if n_threads is None:
        try:
            n_threads = len(os.sched_getaffinity(0))
        except AttributeError:
            n_threads = os.cpu_count() or 1
        n_threads = max(1, min(16, n_threads))
    if is_bgzf(path) and n_threads > 1:
        return _ByteStream(_bgzf_pieces(path, n_threads))
    return _ByteStream(_gzip_pieces(path))


def write_bgzf(path, data, block=0xFF00, level=6):
    """Write `data` as BGZF (what `bgzip` produces): independent members of at most 64 KiB with their size in the
    header, closed by the empty EOF member."""
    data = memoryview(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data)
    with open(path, "wb") as f:
        for p in range(0, len(data), block):
            chunk = bytes(data[p:p + block])
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            payload = c.compress(chunk) + c.flush()
            bsize = 18 + len(payload) + 8 - 1
            assert bsize < 65536
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize))
            f.write(payload)
            f.write(struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
        f.write(_BGZF_EOF)
