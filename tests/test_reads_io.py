"""CPU tests of the chunked FASTA/FASTQ(+gz) reader."""
import os
import struct
import zlib

import numpy as np
import pytest

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from kmer_mapper_amd import reads_io
from kmer_mapper_amd import synthetic as syn
from kmer_mapper_amd.util import ReadBatch


def _concat(batches):
    batches = list(batches)
    bases = np.concatenate([b.bases for b in batches]) if batches else np.zeros(0, np.uint8)
    lens = np.concatenate([np.diff(b.offsets) for b in batches]) if batches else np.zeros(0, np.int64)
    offs = np.zeros(lens.shape[0] + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    return ReadBatch(bases, offs), len(batches)


@pytest.fixture(scope="module")
def ragged():
    genome = syn.make_genome(5000, seed=3)
    bases, offs = syn.make_ragged_reads(genome, 300, 0, 200, seed=4)
    return ReadBatch(bases, offs)


@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("chunk", [64, 1000, 10 ** 7])
def test_fastq_roundtrip(tmp_path, ragged, gz, chunk):
    p = str(tmp_path / ("r.fq.gz" if gz else "r.fq"))
    reads_io.write_fastq(p, ragged, gz=gz)
    got, n_chunks = _concat(reads_io.read_chunks(p, min_chunk_size=chunk))
    assert np.array_equal(got.offsets, ragged.offsets) and np.array_equal(got.bases, ragged.bases)
    if chunk == 64:
        assert n_chunks > 10


@pytest.mark.parametrize("line_width", [None, 60, 7])
@pytest.mark.parametrize("chunk", [50, 997, 10 ** 7])
def test_fasta_roundtrip(tmp_path, ragged, line_width, chunk):
    p = str(tmp_path / "r.fa")
    reads_io.write_fasta(p, ragged, line_width=line_width)
    got, _ = _concat(reads_io.prefetch(reads_io.read_chunks(p, min_chunk_size=chunk)))
    assert np.array_equal(got.offsets, ragged.offsets) and np.array_equal(got.bases, ragged.bases)


def test_crlf_and_missing_final_newline(tmp_path):
    p = tmp_path / "x.fa"
    p.write_bytes(b">a\r\nACGT\r\nAC\r\n>b\r\nTTT")
    got, _ = _concat(reads_io.read_chunks(str(p), min_chunk_size=5))
    assert got.offsets.tolist() == [0, 6, 9] and got.bases.tobytes() == b"ACGTACTTT"
    q = tmp_path / "x.fq"
    q.write_bytes(b"@a\nACGT\n+\nIIII\n@b\nTT\n+\nII")
    got, _ = _concat(reads_io.read_chunks(str(q), min_chunk_size=9))
    assert got.offsets.tolist() == [0, 4, 6] and got.bases.tobytes() == b"ACGTTT"


def test_malformed_raises(tmp_path):
    p = tmp_path / "bad.fq"
    p.write_bytes(b"@a\nACGT\nX\nIIII\n")
    with pytest.raises(ValueError):
        list(reads_io.read_chunks(str(p)))
    e = tmp_path / "empty.fa"
    e.write_bytes(b"")
    assert list(reads_io.read_chunks(str(e))) == []


def test_sniff_format_and_raw_chunker(tmp_path, ragged):
    fq = str(tmp_path / "a.fq")
    reads_io.write_fastq(fq, ragged)
    assert reads_io.sniff_format(fq) == ("fastq", True)
    fa2 = str(tmp_path / "a.fa")
    reads_io.write_fasta(fa2, ragged)
    assert reads_io.sniff_format(fa2) == ("fasta", True)
    fam = str(tmp_path / "m.fa")
    reads_io.write_fasta(fam, ragged, line_width=50)
    assert reads_io.sniff_format(fam) == ("fasta", False)           # wrapped sequences -> host parser
    gz = str(tmp_path / "a.fq.gz")
    reads_io.write_fastq(gz, ragged, gz=True)
    assert reads_io.sniff_format(gz) == ("fastq", True)

    # the chunker hands out raw bytes; a fake consumer cuts at the last complete 4-line record
    whole = open(fq, "rb").read()
    for path in (fq, gz):
        ch = reads_io.RawChunker(path, 5000)
        seen = bytearray()
        while True:
            buf = ch.next_chunk()
            if buf is None:
                break
            nl = np.flatnonzero(buf == 10)
            n_whole = (nl.shape[0] // 4) * 4
            assert n_whole > 0
            used = int(nl[n_whole - 1]) + 1
            seen += buf[:used].tobytes()
            ch.consumed(used)
        ch.close()
        assert bytes(seen) == whole


def test_raw_chunker_adds_final_newline(tmp_path):
    p = tmp_path / "x.fq"
    p.write_bytes(b"@a\nACGT\n+\nIIII\n@b\nTT\n+\nII")
    ch = reads_io.RawChunker(str(p), 1000)
    buf = ch.next_chunk()
    assert buf.tobytes().endswith(b"II\n")
    ch.consumed(buf.shape[0])
    assert ch.next_chunk() is None


# ---------------------------------------------------------------- byte-range sharding over ranks
def _random_fastq(path, n, rng, crlf=False):
    """Records with 0..300 bases and quality lines full of '@' and '+' (the characters that make naive
    re-synchronisation fail).  Returns (record start offsets, file size)."""
    nlb = b"\r\n" if crlf else b"\n"
    starts, out, pos = [], [], 0
    for i in range(n):
        L = int(rng.integers(0, 300))
        seq = bytes(rng.choice(list(b"ACGT"), size=L))
        qual = bytes(rng.choice(list(b"@+IF#"), size=L))
        rec = b"@r%d x" % i + nlb + seq + nlb + b"+" + nlb + qual + nlb
        starts.append(pos)
        pos += len(rec)
        out.append(rec)
    with open(path, "wb") as f:
        f.write(b"".join(out))
    return np.array(starts), pos


@pytest.mark.parametrize("crlf", [False, True])
def test_rank_byte_ranges_partition_a_fastq_exactly(tmp_path, crlf):
    """Rank g owns exactly the records that start inside [g*size/G, (g+1)*size/G) — for G in {1,2,3,8,37}, with
    '@'-leading quality lines; the ranges tile the file (reference: one chunk -> one worker,
    command_line_interface.py:109-111)."""
    from kmer_mapper_amd import reads_io as rio
    rng = np.random.default_rng(7 + crlf)
    p = str(tmp_path / "t.fq")
    starts, size = _random_fastq(p, 3000, rng, crlf)
    for G in (1, 2, 3, 8, 37):
        prev = 0
        for g in range(G):
            lo, hi = rio.rank_byte_range(p, "fastq", g, G)
            assert lo == prev
            want = starts[(starts >= size * g // G) & (starts < size * (g + 1) // G)]
            assert np.array_equal(starts[(starts >= lo) & (starts < hi)], want), (G, g)
            prev = hi
        assert prev == size
    # every rank's reads, parsed from its own range only, put together = the reads of the whole file
    whole = np.concatenate([b.bases for b in rio.read_chunks(p, 50_000)])
    for G in (3, 8):
        parts = [b.bases for g in range(G)
                 for b in rio.read_chunks(p, 50_000, byte_range=rio.rank_byte_range(p, "fastq", g, G))]
        assert np.array_equal(np.concatenate(parts), whole)
    # the cut a rank makes in a chunk it only skips (shared .gz stream) = the end of the last whole record
    buf = np.fromfile(p, dtype=np.uint8)
    ends = np.append(starts[1:], size)
    for cut in rng.integers(1, size, size=300):
        e = ends[ends <= cut]
        assert rio.last_record_start(buf[:cut], "fastq") == (e[-1] if e.shape[0] else 0)


def test_rank_byte_ranges_fasta_and_shared_gz_stream(tmp_path):
    from kmer_mapper_amd import reads_io as rio
    rng = np.random.default_rng(11)
    recs, starts, pos = [], [], 0
    for i in range(2000):
        L = int(rng.integers(0, 200))
        rec = b">r%d\n" % i + bytes(rng.choice(list(b"ACGT"), size=L)) + b"\n"
        starts.append(pos)
        pos += len(rec)
        recs.append(rec)
    p = str(tmp_path / "t.fa")
    with open(p, "wb") as f:
        f.write(b"".join(recs))
    starts, size = np.array(starts), pos
    for G in (2, 3, 8):
        prev = 0
        for g in range(G):
            lo, hi = rio.rank_byte_range(p, "fasta", g, G)
            assert lo == prev
            assert np.array_equal(starts[(starts >= lo) & (starts < hi)],
                                  starts[(starts >= size * g // G) & (starts < size * (g + 1) // G)])
            prev = hi
        assert prev == size
    # ranks sharing one gzip stream: chunk i -> rank i mod G; skipped chunks come out as None, owned ones parsed
    import gzip
    pz = str(tmp_path / "t.fa.gz")
    with gzip.open(pz, "wb") as f:
        f.write(b"".join(recs))
    whole = [b for b in rio.read_chunks(pz, 20_000)]
    for G in (2, 3):
        got = [None] * len(whole)
        for g in range(G):
            for i, b in enumerate(rio.read_chunks(pz, 20_000, owned=lambda i, g=g: i % G == g)):
                if b is not None:
                    assert i % G == g and got[i] is None
                    got[i] = b
        assert all(np.array_equal(a.bases, b.bases) and np.array_equal(a.offsets, b.offsets)
                   for a, b in zip(whole, got))


def _gpu_parser_consumed(buf, period):
    """Byte-wise restatement of kmm_map_records' `consumed` (csrc/kmm_records.hpp k_rec_scan2): the byte after the
    last newline whose 1-based count is a multiple of `period`."""
    cut = count = 0
    for i, c in enumerate(bytes(buf)):
        if c == 10:
            count += 1
            if count % period == 0:
                cut = i + 1
    return cut


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
def test_ranks_sharing_a_gz_stream_cut_skipped_chunks_like_the_gpu_parser(tmp_path, fmt):
    """the owner of a chunk advances by the GPU parser's `consumed` (newline-count rule); a rank that
    skips the chunk must cut it at the same byte, also when the chunk ends exactly on a record's last newline (where
    a 'next header seen' rule holds the last FASTA record back).  Simulated ranks over RawChunker, G in {2, 3, 8}:
    every record is owned exactly once and all ranks walk the same cuts."""
    import gzip
    from kmer_mapper_amd import reads_io as rio
    rng = np.random.default_rng(31)
    recs = []
    for i in range(1500):
        seq = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 120))))
        recs.append((b">r%d\n" % i + seq + b"\n") if fmt == "fasta" else
                    (b"@r%d\n" % i + seq + b"\n+\n" + b"I" * len(seq) + b"\n"))
    data = b"".join(recs)
    pz = str(tmp_path / ("t.%s.gz" % ("fa" if fmt == "fasta" else "fq")))
    with gzip.open(pz, "wb") as f:
        f.write(data)
    period = 2 if fmt == "fasta" else 4
    # the 3-record example of the finding: a buffer that ends on a record's final newline
    three = b"".join(recs[:3])
    assert rio.records_cut(np.frombuffer(three, np.uint8), fmt) == len(three) == _gpu_parser_consumed(three, period)
    if fmt == "fasta":
        assert rio.last_record_start(np.frombuffer(three, np.uint8), fmt) < len(three)   # the rule that disagreed
    for chunk_size in (997, 4096, len(recs[0]) + len(recs[1])):      # the last one ends chunks on record boundaries
        for G in (2, 3, 8):
            owned, cuts = [], []
            for g in range(G):
                ch = rio.RawChunker(pz, chunk_size)
                mine, my_cuts, i = bytearray(), [], 0
                while True:
                    buf = ch.next_chunk()
                    if buf is None:
                        break
                    if i % G == g:
                        used = _gpu_parser_consumed(buf, period)
                        mine += buf[:used].tobytes()
                    else:
                        used = rio.records_cut(buf, fmt)
                    assert used > 0
                    my_cuts.append(used)
                    ch.consumed(used)
                    i += 1
                ch.close()
                owned.append(bytes(mine))
                cuts.append(my_cuts)
            assert all(c == cuts[0] for c in cuts), (chunk_size, G)
            assert sum(len(o) for o in owned) == len(data)
            # interleave the owners' chunks back in stream order
            pos, parts = [0] * G, []
            for i, used in enumerate(cuts[0]):
                g = i % G
                parts.append(owned[g][pos[g]:pos[g] + used])
                pos[g] += used
            assert b"".join(parts) == data, (chunk_size, G)


# ---------------------------------------------------------------- .gz input inflated on several cores
def test_bgzf_is_inflated_member_parallel_and_plain_gzip_still_works(tmp_path):
    """BGZF (independent members with their size in the header, what bgzip writes) is inflated on a thread pool;
    a plain gzip file — also several concatenated members — by one thread.  Both give the same reads as the
    uncompressed file through read_chunks and RawChunker (reference: '.fa.gz, or fq.gz', Readme.md:11; igzip intent
    kmer_mapper/util.py:78-101)."""
    import gzip
    from kmer_mapper_amd import gz_io, reads_io as rio
    rng = np.random.default_rng(21)
    p = str(tmp_path / "t.fq")
    _random_fastq(p, 4000, rng)
    data = open(p, "rb").read()
    pb, pg, pm = str(tmp_path / "b.fq.gz"), str(tmp_path / "g.fq.gz"), str(tmp_path / "m.fq.gz")
    gz_io.write_bgzf(pb, data, block=20_000)                  # many members
    assert gz_io.is_bgzf(pb) and gzip.open(pb, "rb").read() == data    # ... and a valid gzip file for everyone else
    with gzip.open(pg, "wb") as f:
        f.write(data)
    half = data.index(b"\n@r2000 x") + 1
    with open(pm, "wb") as f:                                  # two concatenated plain members
        f.write(gzip.compress(data[:half]) + gzip.compress(data[half:]))
    assert not gz_io.is_bgzf(pg)
    for path in (pb, pg, pm):
        for n_threads in (1, 4):
            with gz_io.open_gz(path, n_threads) as s:
                got = bytearray()
                while True:
                    piece = s.read(70_001)
                    if not piece:
                        break
                    got += piece
            assert bytes(got) == data, (path, n_threads)
    whole = list(rio.read_chunks(p, 60_000))
    for path in (pb, pm):
        chunks = list(rio.read_chunks(path, 60_000))
        assert np.array_equal(np.concatenate([c.bases for c in chunks]), np.concatenate([c.bases for c in whole]))
        ch = rio.RawChunker(path, 50_000)
        out = bytearray()
        while True:
            buf = ch.next_chunk()
            if buf is None:
                break
            used = buf.shape[0] if ch.eof else rio.last_record_start(buf, "fastq")
            out += buf[:used].tobytes()
            ch.consumed(used)
        ch.close()
        assert bytes(out) == data
    # a truncated plain gzip file must raise like gzip.open does, not yield partial reads
    pt = str(tmp_path / "trunc.fq.gz")
    whole_gz = open(pg, "rb").read()
    for cut in (len(whole_gz) // 2, len(whole_gz) - 4):
        open(pt, "wb").write(whole_gz[:cut])
        with pytest.raises(EOFError):
            with gz_io.open_gz(pt, 1) as s:
                s.read()
        with pytest.raises(EOFError):
            list(rio.read_chunks(pt, 60_000))
    # a BGZF payload of the right length but wrong content: the member's CRC32 objects
    raw_b = bytearray(open(pb, "rb").read())
    first = gz_io._bgzf_block_size(bytes(raw_b[:18]))
    fixed = bytes(raw_b[:first])
    xlen = struct.unpack_from("<H", fixed, 10)[0]
    body = zlib.decompress(fixed[12 + xlen:-8], wbits=-15)
    forged = bytearray(body)
    forged[10] ^= 0x01
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = c.compress(bytes(forged)) + c.flush()
    member = (fixed[:16] + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload + fixed[-8:])   # old CRC, same ISIZE
    with pytest.raises(ValueError, match="CRC32"):
        gz_io._inflate_member(member)
    with pytest.raises(ValueError):
        bad = bytearray(open(pb, "rb").read())
        bad[len(bad) // 2] ^= 0xFF                              # corrupt a member: zlib or the size check objects
        open(pb, "wb").write(bad)
        with gz_io.open_gz(pb, 4) as s:
            s.read()


# ---------------------------------------------------------------- native reader (libkmm_io.so)
def test_native_reader_equals_python_reader_on_every_kind_of_file(tmp_path):
    """libkmm_io.so (C++ threads; libdeflate or zlib) against the pure-Python readers and the raw file: BGZF (members
    inflated in parallel straight into the caller's buffer, also when the buffer is smaller than one member), a plain
    gzip stream, concatenated gzip members, an uncompressed file with seek; zlib engine too (KMM_IO_NO_LIBDEFLATE is
    read when the library is first used, so that leg runs in a child process)."""
    import gzip
    import subprocess
    import sys
    from kmer_mapper_amd import _io, gz_io, reads_io as rio
    _io.build()
    rng = np.random.default_rng(41)
    p = str(tmp_path / "t.fq")
    _random_fastq(p, 6000, rng)
    data = open(p, "rb").read()
    pb, pg, pm = str(tmp_path / "b.fq.gz"), str(tmp_path / "g.fq.gz"), str(tmp_path / "m.fq.gz")
    gz_io.write_bgzf(pb, data, block=30_000)
    with gzip.open(pg, "wb") as f:
        f.write(data)
    with open(pm, "wb") as f:
        f.write(gzip.compress(data[:len(data) // 3]) + gzip.compress(data[len(data) // 3:]))
    for path, kind in ((pb, 1), (pg, 2), (pm, 2), (p, 0)):
        for nt in (1, 5):
            for piece in (7, 4093, 70_001, 1 << 20):          # (7 and 4093: smaller than a BGZF member)
                with _io.NativeStream(path, nt) as s:
                    assert s.kind == kind
                    got = bytearray()
                    while True:
                        b = s.read(piece)
                        if not b:
                            break
                        got += b
                        if piece == 7 and len(got) > 3000:
                            got += s.read()
                            break
                assert bytes(got) == data, (path, nt, piece)
    with _io.NativeStream(p, 3) as s:                          # byte ranges of ranks: seek on a plain file
        s.seek(12345)
        assert s.read(1000) == data[12345:13345]
    # the RawChunker the CLI uses, over the native reader, gives the file back
    for path in (pb, pg, p):
        ch = rio.RawChunker(path, 90_000)
        out = bytearray()
        while True:
            buf = ch.next_chunk()
            if buf is None:
                break
            used = rio.records_cut(buf, "fastq")
            out += buf[:used].tobytes()
            ch.consumed(used)
        ch.close()
        assert bytes(out) == data
    # errors: truncated BGZF, truncated gzip stream, corrupt member
    bad = str(tmp_path / "bad.gz")
    raw_b = open(pb, "rb").read()
    open(bad, "wb").write(raw_b[:len(raw_b) // 2])
    with pytest.raises(EOFError):
        with _io.NativeStream(bad, 2) as s:
            s.read()
    raw_g = open(pg, "rb").read()
    open(bad, "wb").write(raw_g[:len(raw_g) - 5])
    with pytest.raises(EOFError):
        with _io.NativeStream(bad, 2) as s:
            s.read()
    flip = bytearray(raw_b)
    flip[len(flip) // 3] ^= 0x5A
    open(bad, "wb").write(flip)
    with pytest.raises((ValueError, EOFError)):
        with _io.NativeStream(bad, 2) as s:
            s.read()
    # the zlib engine gives the same bytes
    code = ("import sys; sys.path.insert(0, %r); from kmer_mapper_amd import _io; assert _io.lib().kmm_io_engine() == 0; "
            "s = _io.NativeStream(%r, 3); d = s.read(); s.close(); sys.stdout.buffer.write(d)" % (ROOT_DIR, pb))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KMM_IO_NO_LIBDEFLATE="1"), capture_output=True)
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout == data


@pytest.mark.parametrize("chunk_size", [500, 4096, 100_000, 10_000_000])
@pytest.mark.parametrize("trailing_newline", [True, False])
def test_prefetching_chunker_hands_out_the_same_records_as_the_plain_one(tmp_path, chunk_size, trailing_newline):
    """PrefetchingRawChunker (two buffers + a reader thread, what `kmer_mapper map` feeds the GPU parser from) and
    RawChunker cut by the same consumer give the same byte stream: ordinary chunks, records longer than a chunk (the
    consumer uses nothing and asks again), a last line without its newline."""
    rng = np.random.default_rng(5)
    recs = []
    for i in range(2000):
        n = int(rng.integers(20, 300))
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * n + b"\n")
    data = b"".join(recs)
    path = tmp_path / "x.fq"
    path.write_bytes(data if trailing_newline else data[:-1])
    for cls in (reads_io.RawChunker, reads_io.PrefetchingRawChunker):
        ch = cls(str(path), chunk_size)
        out, calls = [], 0
        try:
            while True:
                calls += 1
                assert calls < 100_000
                b = ch.next_chunk()
                if b is None:
                    break
                used = reads_io.records_cut(b, "fastq", ch.eof)
                if used == 0:
                    assert not ch.eof
                    ch.chunk_size *= 2
                    continue
                out.append(bytes(b[:used]))
                ch.consumed(used)
        finally:
            ch.close()
        assert b"".join(out) == data, cls.__name__


def test_prefetching_chunker_reports_a_truncated_gz_in_the_consumers_thread(tmp_path):
    """The reader thread's error (gzip stream that ends early: EOFError, like gzip.open) reaches the caller of
    next_chunk / consumed — `kmer_mapper map` must not return partial counts for a truncated reads.fq.gz."""
    import gzip
    rng = np.random.default_rng(9)
    recs = []
    for i in range(3000):
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 100))
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * 100 + b"\n")
    data = b"".join(recs)
    good = tmp_path / "good.fq.gz"
    with gzip.open(good, "wb", compresslevel=1) as f:
        f.write(data)
    raw = good.read_bytes()
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(raw[:len(raw) * 2 // 3])
    ch = reads_io.PrefetchingRawChunker(str(bad), 50_000)
    with pytest.raises(EOFError):
        try:
            while True:
                b = ch.next_chunk()
                if b is None:
                    break
                used = reads_io.records_cut(b, "fastq", ch.eof)
                assert used
                ch.consumed(used)
        finally:
            ch.close()


def test_native_reader_corner_cases_from_the_round_3_review(tmp_path):
    """ADVICE r3 on csrc/kmm_io.cpp: (1) a BGZF file that starts with an EMPTY member followed by members larger than the
    caller's buffer must not read as end-of-stream; (2) a BGZF header whose BSIZE is smaller than the member's own
    fixed parts is 'not a BGZF member', not an out-of-bounds trailer read; (3) zero padding behind a gzip stream is
    skipped as gzip.open does, other trailing bytes are an error — native reader and pure-Python fallback alike."""
    import gzip
    import struct
    from kmer_mapper_amd import _io, gz_io
    _io.build()
    rng = np.random.default_rng(77)
    data = bytes(rng.integers(65, 91, size=300_000, dtype=np.uint8))
    # (1) empty member first, then 60 kB members, read through a 1000-byte buffer
    p1 = str(tmp_path / "lead_empty.gz")
    gz_io.write_bgzf(p1, data, block=60_000)
    body = open(p1, "rb").read()
    open(p1, "wb").write(gz_io._BGZF_EOF + body)
    with _io.NativeStream(p1, 2) as s:
        buf = bytearray(1000)
        assert s.readinto(buf) == 1000 and bytes(buf) == data[:1000]
        rest = s.read()
    assert bytes(buf) + rest == data
    with gz_io.open_gz(p1, 2, native=False) as s:
        assert s.read() == data
    # (2) BSIZE 0 in an otherwise well-formed BGZF header
    p2 = str(tmp_path / "bad_bsize.gz")
    open(p2, "wb").write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 0) + b"\0" * 64)
    with pytest.raises((ValueError, EOFError, OSError)):
        with _io.NativeStream(p2, 1) as s:      # (a gzip-magic file that is no BGZF member: read as a gzip stream, which fails too)
            s.read()
    # a later member with a broken BSIZE: reported by the planner
    p2b = str(tmp_path / "bad_bsize_later.gz")
    good = body[:gz_io._bgzf_block_size(body[:18])]
    open(p2b, "wb").write(good + b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 3) + b"\0" * 64)
    with pytest.raises(ValueError, match="not a BGZF member"):
        with _io.NativeStream(p2b, 1) as s:
            s.read()
    # (3) trailing zero padding / garbage behind a plain gzip stream
    small = data[:50_000]
    p3, p4, p5 = str(tmp_path / "pad.gz"), str(tmp_path / "junk.gz"), str(tmp_path / "pad_then_member.gz")
    open(p3, "wb").write(gzip.compress(small) + b"\0" * 512)
    open(p4, "wb").write(gzip.compress(small) + b"\0" * 7 + b"junk")
    open(p5, "wb").write(gzip.compress(small) + b"\0" * 100 + gzip.compress(small[::-1]))
    assert gzip.open(p3, "rb").read() == small                       # what the reference's reader does
    for native in (True, False):
        with gz_io.open_gz(p3, 1, native=native) as s:
            assert s.read() == small, native
        with gz_io.open_gz(p5, 1, native=native) as s:
            assert s.read() == small + small[::-1], native
        with pytest.raises(ValueError, match="trailing bytes"):
            with gz_io.open_gz(p4, 1, native=native) as s:
                s.read()


@pytest.mark.parametrize("chunk", [None, 40_000])
def test_one_gzip_member_is_inflated_on_many_threads(tmp_path, monkeypatch, chunk):
    """A plain .gz (ONE deflate stream, what `gzip reads.fq` writes) through the native reader with several threads:
    chunks of the compressed stream are decoded from block boundaries found by search, with markers for the unknown
    32 KiB history, and resolved once the previous chunk's end is known (csrc/kmm_inflate.hpp).  Same bytes as
    gzip.open — the reference's reader through bnp.open (command_line_interface.py:102) — for every compression level,
    stored and fixed-Huffman blocks, several members with padding; truncation, a flipped bit and a wrong CRC raise."""
    import gzip
    import zlib
    from kmer_mapper_amd import _io, gz_io
    _io.build()
    if chunk:                 # many small chunks per wave: starts found by search in every one of them
        monkeypatch.setenv("KMM_IO_GZIP_CHUNK", str(chunk))
    rng = np.random.default_rng(31)
    n_rec, L = 12_000, 150                                     # FASTQ-shaped text, generated without a Python loop
    rec = np.empty((n_rec, 10 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0:10] = np.frombuffer(b"@SRR77.000", dtype=np.uint8)
    rec[:, 7:10] = rng.integers(48, 58, size=(n_rec, 3))
    rec[:, 9] = 10
    rec[:, 10:10 + L] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n_rec, L))
    rec[:, 10 + L:13 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 13 + L:13 + 2 * L] = rng.choice(np.frombuffer(b"FFFFFF:,#@+", dtype=np.uint8), size=(n_rec, L))
    rec[:, -1] = 10
    data = rec.tobytes()
    assert len(data) > 3_000_000
    files = {}
    for lvl in (1, 6):
        files["l%d" % lvl] = (gzip.compress(data, compresslevel=lvl), data)
    files["l9"] = (gzip.compress(data[:1_500_000], compresslevel=9), data[:1_500_000])
    noise = rng.integers(0, 256, size=400_000, dtype=np.uint8).tobytes()
    files["noise_and_text"] = (gzip.compress(noise + data[:1_000_000] + noise, compresslevel=6), noise + data[:1_000_000] + noise)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
    files["fixed"] = (co.compress(data[:1_000_000]) + co.flush(), data[:1_000_000])
    co = zlib.compressobj(0, zlib.DEFLATED, 31)
    files["stored"] = (co.compress(data[:1_000_000]) + co.flush(), data[:1_000_000])
    half = len(data) // 2
    files["members"] = (gzip.compress(data[:half], 6) + b"\0" * 37 + gzip.compress(data[half:], 2) + b"\0" * 5, data)
    # many SMALL members with a large one among them (cat of small .gz files): the small ones are decoded one chunk at a
    # time, in order (no speculative chunks into the members behind them), the large one in parallel waves again
    cuts = [0] + sorted(int(x) for x in rng.integers(1, 600_000, size=40)) + [600_000, 2_400_000, 2_450_000, 2_500_000]
    files["small_members"] = (b"".join(gzip.compress(data[a:b], 6) for a, b in zip(cuts[:-1], cuts[1:])), data[:2_500_000])
    named = bytearray(gzip.compress(data[:2_000_000], 6))
    named[3] |= 8                                                   # FNAME: a zero-terminated name behind the fixed header
    files["with_name"] = (bytes(named[:10]) + b"reads.fq\0" + bytes(named[10:]), data[:2_000_000])
    for name, (blob, expect) in files.items():
        path = str(tmp_path / (name + ".gz"))
        open(path, "wb").write(blob)
        for n_threads in (4, 1):
            with _io.NativeStream(path, n_threads) as s:
                assert s.kind == 2
                got = bytearray()
                while True:
                    piece = s.read(1_234_567)
                    if not piece:
                        break
                    got += piece
            assert bytes(got) == expect, (name, n_threads)
    # errors: truncated (EOFError like gzip.open), corrupted payload, wrong CRC / length in the trailer
    blob = files["l6"][0]
    cases = {"trunc_mid": (blob[: len(blob) // 2], EOFError), "trunc_trailer": (blob[:-5], EOFError)}
    flipped = bytearray(blob)
    flipped[len(blob) // 3] ^= 0x10
    cases["bitflip"] = (bytes(flipped), ValueError)
    badcrc = bytearray(blob)
    badcrc[-6] ^= 0xFF
    cases["crc"] = (bytes(badcrc), ValueError)
    badlen = bytearray(blob)
    badlen[-1] ^= 0x01
    cases["isize"] = (bytes(badlen), ValueError)
    for name, (b2, exc) in cases.items():
        path = str(tmp_path / (name + ".gz"))
        open(path, "wb").write(b2)
        with pytest.raises((exc, ValueError) if name == "bitflip" else exc):
            with _io.NativeStream(path, 4) as s:
                s.read()


@pytest.mark.parametrize("container", ["gzip", "bgzf"])
@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("KMM_FUZZ_SEEDS", "1,2").split(",")])
def test_parallel_inflate_of_damaged_streams_raises_or_matches_zlib(tmp_path, monkeypatch, seed, container):
    """Fuzz of the many-thread decoder of one gzip member (csrc/kmm_inflate.hpp): flipped bits, overwritten runs, cut
    and spliced streams.  Whatever the damage, the reader either raises (ValueError / EOFError, as gzip.open does on
    the reference's side, command_line_interface.py:102) or returns exactly the bytes zlib returns for the same file:
    never other bytes, never a crash, never a hang (the speculative chunk decoders run on data that need not be deflate
    at all — every table walk and back reference is bounds-checked).  Same for a BGZF file (members inflated side by
    side, sizes taken from the members' own headers — which the damage may hit)."""
    import gzip
    import zlib
    from kmer_mapper_amd import _io, gz_io
    _io.build()
    monkeypatch.setenv("KMM_IO_GZIP_CHUNK", "30000")
    rng = np.random.default_rng(700 + seed)
    n_rec, L = 5_000, 150
    rec = np.empty((n_rec, 8 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0:8] = np.frombuffer(b"@r.0000\n", dtype=np.uint8)
    rec[:, 3:7] = rng.integers(48, 58, size=(n_rec, 4))
    rec[:, 8:8 + L] = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=(n_rec, L), p=[.24, .25, .25, .24, .02])
    rec[:, 8 + L:11 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 11 + L:11 + 2 * L] = rng.choice(np.frombuffer(b"FFFF:,#", dtype=np.uint8), size=(n_rec, L))
    rec[:, -1] = 10
    data = rec.tobytes()
    if container == "bgzf":
        pb = str(tmp_path / "whole.bgzf.gz")
        gz_io.write_bgzf(pb, data, block=int(rng.integers(5_000, 60_000)))
        blob = open(pb, "rb").read()
    else:
        blob = gzip.compress(data, compresslevel=int(rng.integers(1, 10)))
    assert len(blob) > 300_000                                   # ten and more chunks of 30 000 bytes

    def zlib_says(b):
        try:
            out, d = bytearray(), zlib.decompressobj(31)
            out += d.decompress(b)
            while d.eof and d.unused_data.lstrip(b"\0"):          # further members behind padding, like gzip.open
                rest = d.unused_data.lstrip(b"\0")
                d = zlib.decompressobj(31)
                out += d.decompress(rest)
            return bytes(out) if d.eof else None
        except zlib.error:
            return None

    path = str(tmp_path / "damaged.gz")
    n_ok = n_raised = 0
    for trial in range(40):
        b = bytearray(blob)
        kind = trial % 5
        pos = int(rng.integers(20, len(b) - 20))
        if kind == 0:                                              # one flipped bit
            b[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                            # a run of random bytes
            n = int(rng.integers(1, 2000))
            b[pos:pos + n] = rng.integers(0, 256, size=len(b[pos:pos + n]), dtype=np.uint8).tobytes()
        elif kind == 2:                                            # cut
            del b[pos:]
        elif kind == 3:                                            # a piece removed from the middle
            del b[pos:pos + int(rng.integers(1, 5000))]
        else:                                                      # a piece of the stream repeated
            n = int(rng.integers(1, 5000))
            b[pos:pos] = b[pos:pos + n]
        open(path, "wb").write(bytes(b))
        expect = zlib_says(bytes(b))
        for n_threads in (4, 1):
            try:
                with _io.NativeStream(path, n_threads) as s:
                    got = bytearray()
                    while True:
                        piece = s.read(777_777)
                        if not piece:
                            break
                        got += piece
            except (ValueError, EOFError):
                # (BGZF: the member sizes come from the BSIZE field of the members' extra headers, which zlib skips
                # unread — damage there makes this reader raise on a file gzip.open still inflates: stricter, never wrong)
                assert expect is None or container == "bgzf", ("the reader raised on a stream zlib inflates", trial, kind, n_threads)
                n_raised += 1
                continue
            assert expect is not None and bytes(got) == expect, ("bytes differ from zlib's", trial, kind, n_threads)
            n_ok += 1
    assert n_raised > 40                                          # (nearly every damage is caught; a flipped bit in a
    #                                                               name-less header field may pass in both readers)


def test_bgzf_member_ranges_partition_the_records_exactly():
    """bgzf_ranges.rank_member_range: the ranks of a job take the BGZF members that start in their share of the compressed
    bytes, resynchronised to the record structure — a member starts wherever the compressor's buffer ended.  For FASTQ
    (quality lines that start with '@' among them) and two-line FASTA, members of 64 KiB, 5 000 and 777 bytes, 1 to 40 ranks:
    the ranges, as positions in the inflated file, follow each other without gap or overlap from 0 to the end, every
    non-empty one starts at a record start and holds whole records."""
    import struct
    import zlib
    from kmer_mapper_amd import bgzf_ranges as br
    rng = np.random.default_rng(5)

    def member(chunk, level=6):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload +
                struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))

    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    reads = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(20, 300)))) for _ in range(3000)]
    fastq = b"".join(b"@r%d\n" % i + r + b"\n+\n" + bytes(rng.choice(np.frombuffer(b"@@FF:,#I", dtype=np.uint8), size=len(r))) + b"\n"
                     for i, r in enumerate(reads))
    fasta = b"".join(b">r%d\n" % i + r + b"\n" for i, r in enumerate(reads))
    for raw, fmt, period, first in ((fastq, "fastq", 4, b"@"), (fasta, "fasta", 2, b">")):
        for block in (0xFF00, 5000, 777):
            comp = b"".join(member(raw[p:p + block]) for p in range(0, len(raw), block)) + eof
            offs = br.member_chain(comp)
            assert offs[0] == 0 and offs[-1] == len(comp)
            sizes = [len(br.inflate_member(comp, int(offs[i]), int(offs[i + 1]))) for i in range(len(offs) - 1)]
            assert sum(sizes) == len(raw)
            cum = np.concatenate([[0], np.cumsum(sizes)])
            # the ratio the CLI cuts its calls by: ISIZE against size over the first members (all of them / the first three)
            assert abs(br.inflation_ratio(comp, 0, len(comp), n_members=10 ** 9) - len(raw) / len(comp)) < 1e-9
            assert abs(br.inflation_ratio(comp, 0, len(comp), n_members=3) - sum(sizes[:3]) / int(offs[3])) < 1e-9
            assert br.inflation_ratio(comp, 1, len(comp)) == 1.0 and br.inflation_ratio(comp, len(comp), len(comp)) == 1.0
            for world in (1, 2, 3, 7, 40):
                spans = []
                where = {int(o): i for i, o in enumerate(offs)}                   # member index by compressed offset
                for r in range(world):
                    lo, s0, hi, s1 = br.rank_member_range(comp, fmt, r, world)
                    m0, m1 = where[lo], where[hi]
                    assert 0 <= s0 < max(sizes[m0] if m0 < len(sizes) else 1, 1) and 0 <= s1
                    spans.append((int(cum[m0]) + s0, int(cum[m1]) + s1))
                assert spans[0][0] == 0 and spans[-1][1] == len(raw), (fmt, block, world)
                assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:])), (fmt, block, world)
                for a, b in spans:
                    if a < b:
                        assert raw[a:a + 1] == first and (a == 0 or raw[a - 1:a] == b"\n")
                        assert raw[a:b].count(b"\n") % period == 0
    with pytest.raises(ValueError):
        br.member_chain(comp[:-40] + b"garbage that is no member header, forty bytes")
    # a boundary is found from its byte position alone: bytes that look like a header inside a member's payload are passed over
    # (stored blocks: the payload holds the decoys verbatim — a bare magic number, and a complete header whose BSIZE leads nowhere)
    decoy = b"\x1f\x8b\x08\x04" + bytes(60) + b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x63\x00" + bytes(120)
    noisy = b"".join(member(decoy * 3 + raw[p:p + 5000], level=0) for p in range(0, 200_000, 5000)) + eof
    assert noisy.count(b"\x1f\x8b\x08\x04") > 3 * (len(noisy) // 6000)
    n_offs = br.member_chain(noisy)
    for t in range(1, len(noisy), 997):
        m, prev = br.member_at_or_after(noisy, t)
        k2 = int(np.searchsorted(n_offs, t, side="left"))
        assert m == n_offs[k2] and (prev is None or prev == n_offs[k2 - 1])


def test_native_readers_under_thread_sanitizer(tmp_path):
    """libkmm_io's readers (csrc/kmm_io.cpp + kmm_inflate.hpp) built with ThreadSanitizer (tests/kmm_io_tsan_main.cpp): one
    plain gzip member inflated on four threads from block boundaries found by search, many small members one behind the other
    (one chunk at a time), a BGZF file (members in parallel), read back in odd-sized pieces: the bytes that went in, no data
    race between the read-ahead thread, the pool and the reader."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "io_tsan")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I" + os.path.join(root, "include"),
           "-I" + os.path.join(root, "kmer_mapper_amd", "csrc"), os.path.join(root, "tests", "kmm_io_tsan_main.cpp"), "-o", exe, "-lz", "-ldl"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("no ThreadSanitizer runtime on this box")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    out = run.stdout + run.stderr
    if "unexpected memory mapping" in out:
        pytest.skip("ThreadSanitizer cannot run in this address-space layout")
    assert run.returncode == 0 and out.count(", same") == 3, out[-3000:]
    assert "WARNING: ThreadSanitizer" not in out, out[-3000:]


def test_mmap_chunker_populated_ahead_hands_out_the_same_bytes(tmp_path):
    """MmapChunker.populate(): helper threads map the file's pages ahead of the packer (the CLI starts them before the index
    upload); the chunks are the same views, close() waits for the helpers, a second populate() is a no-op."""
    p = str(tmp_path / "r.fq")
    genome = syn.make_genome(4000, seed=5)
    bases, offs = syn.make_reads(genome, 2000, 150, seed=6)
    reads_io.write_fastq(p, ReadBatch(bases, offs))
    raw = np.fromfile(p, dtype=np.uint8)
    for byte_range in (None, (4097, raw.shape[0] - 10)):
        c = reads_io.MmapChunker(p, 100_000, byte_range)
        c.populate(n_threads=3, piece=1 << 16)
        c.populate(n_threads=3)
        got = []
        while True:
            v = c.next_chunk()
            if v is None:
                break
            got.append(np.array(v))
            c.consumed(v.shape[0])
        c.wait_populated()
        c.close()
        lo, hi = (0, raw.shape[0]) if byte_range is None else byte_range
        whole = np.concatenate(got)
        assert np.array_equal(whole[:hi - lo], raw[lo:hi])
    empty = str(tmp_path / "empty.fq")
    open(empty, "wb").close()
    c = reads_io.MmapChunker(empty, 1000)
    c.populate()
    assert c.next_chunk() is None
    c.close()


def test_page_nodes_and_where_the_packer_threads_go(tmp_path, monkeypatch):
    """MmapChunker.page_nodes samples the NUMA node of the file's page-cache pages; distributed.packer_cpus_near sends the
    packer threads to that node when it is not the GPU's and holds most of the pages (fake sysfs: no second socket here)."""
    from kmer_mapper_amd import distributed as D
    p = str(tmp_path / "r.fq")
    with open(p, "wb") as f:
        f.write(b"@r\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 20000)
    c = reads_io.MmapChunker(p, 1 << 16)
    nodes = c.page_nodes(16)
    c.close()
    assert nodes == {} or (sum(nodes.values()) >= 16 and all(isinstance(k, int) for k in nodes))
    sysfs = tmp_path / "node"
    (sysfs / "node0").mkdir(parents=True)
    (sysfs / "node1").mkdir()
    (sysfs / "node0" / "cpulist").write_text("0-3\n")
    (sysfs / "node1" / "cpulist").write_text("4-7,12\n")
    monkeypatch.setattr(D, "_BOUND", None)
    assert D.packer_cpus_near({1: 10}, sysfs=str(sysfs)) is None                  # (the rank was never bound: nothing to undo)
    monkeypatch.setattr(D, "_BOUND", {"before": set(range(0, 12)), "numa_node": 0})
    assert D.packer_cpus_near({1: 10}, sysfs=str(sysfs)) == (1, {4, 5, 6, 7})     # (CPU 12 was never this process's)
    assert D.packer_cpus_near({0: 10}, sysfs=str(sysfs)) is None                  # the GPU's own node
    assert D.packer_cpus_near({0: 4, 1: 6}, sysfs=str(sysfs)) is None             # no clear majority
    assert D.packer_cpus_near({}, sysfs=str(sysfs)) is None
    assert D.packer_cpus_near({2: 10}, sysfs=str(sysfs)) is None                  # a node sysfs does not know
    monkeypatch.setattr(D, "_BOUND", {"before": {0, 1}, "numa_node": 0})
    assert D.packer_cpus_near({1: 10}, sysfs=str(sysfs)) is None                  # none of that node's CPUs may be used
