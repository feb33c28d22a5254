"""Member ranges of a BGZF file for the ranks of a job (BASELINE configs[4]: "gzipped .fq input" on 8 GPUs; the reference reads
a .gz through one reader, command_line_interface.py:102): every rank takes the members that start in its share of the
COMPRESSED bytes, resynchronised to the record structure at both ends — a member starts wherever the compressor's 64 KiB
buffer ended, usually inside a record.  The rule is the one of reads_io.find_record_start for plain files, applied to the
inflated bytes of the few members around a boundary (inflated here on the host with zlib: two or three members per
boundary); the ranges of ranks 0..world-1 partition the file's records exactly."""
import struct
import zlib

import numpy as np

from .reads_io import _NL, _is_record_start


def member_chain(buf):
    """Compressed offsets of the members of the BGZF file in `buf` (bytes-like / mmap): int64[n_members + 1], the last
    entry the file's size.  Walks the BSIZE fields (SAM specification 4.1); ValueError if the chain breaks."""
    size = len(buf)
    offs = [0]
    p = 0
    while p < size:
        if p + 18 > size or buf[p:p + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("no BGZF member at compressed byte %d" % p)
        xlen = struct.unpack_from("<H", buf, p + 10)[0]
        q, bsize = p + 12, None
        while q + 4 <= p + 12 + xlen:
            si1, si2, slen = struct.unpack_from("<BBH", buf, q)
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", buf, q + 4)[0] + 1
                break
            q += 4 + slen
        if bsize is None or p + bsize > size:
            raise ValueError("BGZF member at compressed byte %d has no size / runs past the end of the file" % p)
        p += bsize
        offs.append(p)
    return np.asarray(offs, dtype=np.int64)


def inflate_member(buf, lo, hi):
    """The inflated bytes of the member buf[lo:hi]."""
    xlen = struct.unpack_from("<H", buf, lo + 10)[0]
    return zlib.decompress(bytes(buf[lo + 12 + xlen:hi - 8]), -15)


def _record_start_at_or_after(buf, offs, m, fmt):
    """(member, offset in its inflated bytes) of the first record that starts at or after the first inflated byte of
    member m; (n_members, 0) if there is none."""
    n_members = len(offs) - 1
    if m <= 0:
        return 0, 0
    if m >= n_members:
        return n_members, 0
    prev = inflate_member(buf, int(offs[m - 1]), int(offs[m]))
    k = m - 1
    while not prev and k > 0:                          # (empty members before it: the byte in front of member m lies further back)
        k -= 1
        prev = inflate_member(buf, int(offs[k]), int(offs[k + 1]))
    parts = [prev[-1:] if prev else b"\n"]             # one byte early: position 0 is a line start iff the byte before is a newline
    lens = []
    nxt = m
    while True:
        take = 2 if not lens else max(2, len(lens))    # two members first, then twice as many (long records)
        for _ in range(take):
            if nxt >= n_members:
                break
            out = inflate_member(buf, int(offs[nxt]), int(offs[nxt + 1]))
            parts.append(out)
            lens.append(len(out))
            nxt += 1
        at_eof = nxt >= n_members
        data = np.frombuffer(b"".join(parts), dtype=np.uint8)
        if at_eof and data.shape[0] and data[-1] != _NL:
            data = np.concatenate([data, np.array([_NL], dtype=np.uint8)])
        starts = (np.flatnonzero(data == _NL) + 1).astype(np.int64)
        found = None
        for i in range(len(starts) - 1):
            r = _is_record_start(data, starts, i, "fastq" if fmt == "fastq" else "fasta", at_eof)
            if r is None:
                break
            if r:
                found = int(starts[i]) - 1             # relative to the first byte of member m
                break
        if found is not None:
            j = 0
            while j < len(lens) and found >= lens[j]:
                found -= lens[j]
                j += 1
            if j == len(lens):                          # (exactly at the end of what was inflated)
                return m + j, 0
            return m + j, found
        if at_eof:
            return n_members, 0


def rank_member_range(buf, offs, fmt, rank, world_size):
    """What `rank` of `world_size` maps: (first member, head_skip, end member, tail_stop) — members [first, end) whole, minus
    the first head_skip inflated bytes of member `first`, plus the first tail_stop inflated bytes of member `end` (tail_stop
    0: none of it).  Boundaries by compressed bytes: rank r starts with the first member that begins at or after
    size * r / world_size."""
    n_members = len(offs) - 1
    size = int(offs[-1])

    def boundary(r):
        if r <= 0:
            return 0, 0
        if r >= world_size:
            return n_members, 0
        m = int(np.searchsorted(offs[:-1], size * r // world_size, side="left"))
        return _record_start_at_or_after(buf, offs, m, fmt)

    (m0, s0), (m1, s1) = boundary(rank), boundary(rank + 1)
    if (m1, s1) < (m0, s0):
        m1, s1 = m0, s0
    return m0, s0, m1, s1
