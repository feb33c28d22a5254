"""Member ranges of a BGZF file for the ranks of a job (BASELINE configs[4]: "gzipped .fq input" on 8 GPUs; the reference reads
a .gz through one reader, command_line_interface.py:102): every rank takes the members that start in its share of the
COMPRESSED bytes, resynchronised to the record structure at both ends — a member starts wherever the compressor's 64 KiB
buffer ended, usually inside a record.  The rule is the one of reads_io.find_record_start for plain files, applied to the
inflated bytes of the few members around a boundary (inflated here on the host with zlib: two or three members per
boundary); the ranges of ranks 0..world-1 partition the file's records exactly.  Nothing here walks the whole file: a
boundary is found from its byte position (the next member header whose chain of BSIZE fields holds), so a 100 GB file costs
a rank what a 100 MB one does."""
import struct
import zlib

import numpy as np

from .reads_io import _NL, _is_record_start

_MAGIC = b"\x1f\x8b\x08\x04"


def member_end(buf, p):
    """End offset of the BGZF member that starts at compressed byte p (SAM specification 4.1: gzip header with the BC extra
    subfield, BSIZE = total size - 1); ValueError if there is none."""
    size = len(buf)
    if p + 18 > size or buf[p:p + 4] != _MAGIC:
        raise ValueError("no BGZF member at compressed byte %d" % p)
    xlen = struct.unpack_from("<H", buf, p + 10)[0]
    q = p + 12
    while q + 4 <= p + 12 + xlen:
        si1, si2, slen = struct.unpack_from("<BBH", buf, q)
        if si1 == 66 and si2 == 67 and slen == 2 and q + 6 <= size:
            end = p + struct.unpack_from("<H", buf, q + 4)[0] + 1
            if end > size or end < p + 12 + xlen + 8:
                break
            return end
        q += 4 + slen
    raise ValueError("BGZF member at compressed byte %d has no size / runs past the end of the file" % p)


def member_chain(buf):
    """Compressed offsets of ALL members of the BGZF file in `buf` (bytes-like / mmap): int64[n_members + 1], the last entry the
    file's size (tests and tools; the rank ranges below do not need it)."""
    offs, p = [0], 0
    while p < len(buf):
        p = member_end(buf, p)
        offs.append(p)
    return np.asarray(offs, dtype=np.int64)


def inflation_ratio(buf, lo, hi, n_members=64):
    """Inflated / compressed size over the first `n_members` members of buf[lo:hi] (their ISIZE trailers against their sizes):
    what a caller needs to cut a file into calls of a given INFLATED size without walking its whole member chain.  1.0 if
    there is no member at lo."""
    p, comp, raw = lo, 0, 0
    try:
        for _ in range(n_members):
            if p >= hi:
                break
            e = member_end(buf, p)
            comp += e - p
            raw += struct.unpack_from("<I", buf, e - 4)[0]
            p = e
    except ValueError:
        pass
    return raw / comp if comp and raw else 1.0


def inflate_member(buf, lo, hi):
    """The inflated bytes of the member buf[lo:hi]."""
    xlen = struct.unpack_from("<H", buf, lo + 10)[0]
    return zlib.decompress(bytes(buf[lo + 12 + xlen:hi - 8]), -15)


def _holds(buf, p, depth=3):
    """Does a chain of `depth` members (or fewer, up to the end of the file) start at p?"""
    try:
        for _ in range(depth):
            if p == len(buf):
                return True
            p = member_end(buf, p)
        return True
    except ValueError:
        return False


def member_at_or_after(buf, t):
    """(start of the first member that begins at or after compressed byte t, start of the member before it or None).  Found
    from t itself: the first header within 70 000 bytes in front of t whose chain holds, then along the chain."""
    size = len(buf)
    if t <= 0:
        return 0, None
    p = max(0, t - 70000)
    while True:
        p = buf.find(_MAGIC, p) if p > 0 else 0
        if p < 0 or p >= size:
            return size, None
        if _holds(buf, p):
            break
        p += 1
    prev = None
    while p < t and p < size:
        prev, p = p, member_end(buf, p)
    return p, prev


def _record_start_at_or_after(buf, m, prev, fmt):
    """(start offset of a member, offset in its inflated bytes) of the first record that starts at or after the first
    inflated byte of the member at compressed byte m (prev: the member before it); (file size, 0) if there is none."""
    size = len(buf)
    if m <= 0:
        return 0, 0
    if m >= size:
        return size, 0
    before = inflate_member(buf, prev, m) if prev is not None else b""
    # (an empty member in front — rare — would need the one before it: take the line start on trust then, the rule below
    # still rejects a position that is no record start)
    parts = [before[-1:] if before else b"\n"]          # one byte early: position 0 is a line start iff the byte before is a newline
    members, lens = [], []
    nxt = m
    while True:
        for _ in range(max(2, len(lens))):               # two members first, then twice as many (long records)
            if nxt >= size:
                break
            end = member_end(buf, nxt)
            out = inflate_member(buf, nxt, end)
            parts.append(out)
            members.append(nxt)
            lens.append(len(out))
            nxt = end
        at_eof = nxt >= size
        data = np.frombuffer(b"".join(parts), dtype=np.uint8)
        if at_eof and data.shape[0] and data[-1] != _NL:
            data = np.concatenate([data, np.array([_NL], dtype=np.uint8)])
        starts = (np.flatnonzero(data == _NL) + 1).astype(np.int64)
        for i in range(len(starts) - 1):
            r = _is_record_start(data, starts, i, "fastq" if fmt == "fastq" else "fasta", at_eof)
            if r is None:
                break
            if r:
                found = int(starts[i]) - 1              # relative to the first inflated byte of the member at m
                j = 0
                while j < len(lens) and found >= lens[j]:
                    found -= lens[j]
                    j += 1
                return (members[j], found) if j < len(lens) else (nxt, 0)
        if at_eof:
            return size, 0


def rank_member_range(buf, fmt, rank, world_size):
    """What `rank` of `world_size` maps: (lo, head_skip, hi, tail_stop) — the members in compressed bytes [lo, hi) whole, minus
    the first head_skip inflated bytes of the member at lo, plus the first tail_stop inflated bytes of the member at hi
    (tail_stop 0: none of it).  Boundaries by compressed bytes: rank r starts with the first record that begins in or behind
    the first member at or after byte size * r / world_size."""
    size = len(buf)

    def boundary(r):
        if r <= 0:
            return 0, 0
        if r >= world_size:
            return size, 0
        m, prev = member_at_or_after(buf, size * r // world_size)
        return _record_start_at_or_after(buf, m, prev, fmt)

    (m0, s0), (m1, s1) = boundary(rank), boundary(rank + 1)
    if (m1, s1) < (m0, s0):
        m1, s1 = m0, s0
    return m0, s0, m1, s1
