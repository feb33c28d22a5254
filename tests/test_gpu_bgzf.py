"""GPU tests of kmm_map_bgzf (csrc/kmm_gpu_inflate.hpp): BGZF members — the .gz the reference's Readme.md:11 names, as bgzip
writes it — inflated on the GPU, one thread per member, and parsed there; bit-exact against zlib's bytes (through the
oracle's counts on the reads) and refused when damaged."""
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kmm():
    from kmer_mapper_amd import _lib
    assert _lib.device_count() >= 1, "GPU tests need a HIP device"
    import kmer_mapper_amd.engine as engine
    return engine


@pytest.fixture(scope="module")
def syn():
    from kmer_mapper_amd import synthetic
    return synthetic


_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _member(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    payload = c.compress(chunk) + c.flush()
    bsize = 18 + len(payload) + 8 - 1
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + payload +
            struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def _bgzf(data, block=0xFF00, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, eof=True, rng=None):
    out, p = [], 0
    while p < len(data):
        n = block if rng is None else int(rng.integers(1, block + 1))
        out.append(_member(data[p:p + n], level, strategy))
        p += n
    return b"".join(out) + (_EOF if eof else b"")


def _fastq(reads, rng):
    return b"".join(b"@read%d some text\n" % i + r + b"\n+\n" + bytes(rng.choice(np.frombuffer(b"FFFF:,#@+I", dtype=np.uint8), size=len(r))) + b"\n"
                    for i, r in enumerate(reads))


def _feed(dev, comp, fmt, k, step):
    """The caller's loop: windows of `step` compressed bytes, advanced by what each call used."""
    pos, total, size = 0, 0, len(comp)
    buf = np.frombuffer(comp, dtype=np.uint8)
    window = step
    while pos < size:
        end = min(pos + window, size)
        used, n_rec = dev.map_bgzf(buf[pos:end], fmt=fmt, k=k, first=pos == 0, last=end == size)
        if used == 0 and end < size:
            window *= 2
            continue
        assert used > 0
        window = step
        pos += used
        total += n_rec
    return total


def _feed_hinted(dev, comp, fmt, k, step):
    """The CLI's loop (command_line_interface._map_bgzf_file): windows that END at fixed places, each call told which bytes
    follow (kmm_map_bgzf_hint_next: staged under the call's own inflate kernel)."""
    buf = np.frombuffer(comp, dtype=np.uint8)
    size, pos, total = len(comp), 0, 0
    end = min(step, size)
    while pos < size:
        nxt = min(end + step, size)
        used, n_rec = dev.map_bgzf(buf[pos:end], fmt=fmt, k=k, first=pos == 0, last=end == size,
                                   next_chunk=buf[end:nxt] if nxt > end else None)
        assert used > 0 or end < size
        pos += used
        total += n_rec
        if pos < end and end == size:
            continue
        end = nxt
    return total


@pytest.mark.parametrize("level,strategy", [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_FILTERED),
                                            (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY)])
def test_bgzf_members_inflated_on_the_gpu_give_the_oracles_counts(kmm, syn, oracle, level, strategy):
    """FASTQ compressed into BGZF members of every block type (dynamic, fixed, stored; Huffman-only), full-size and random-size
    members, fed whole and in compressed windows that end anywhere (inside members, inside records): the node counts equal
    the oracle's on the reads, the record counts add up, and the carried bytes are gone at the end."""
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(20000, seed=701)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 9000, 0, 260, seed=702)
    reads = [bases[offs[i]:offs[i + 1]].tobytes() for i in range(len(offs) - 1)]
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    rng = np.random.default_rng(703)
    raw = _fastq(reads, rng)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        for path in (2, 0):
            dev.set_param("path", path)
            for comp, step in ((_bgzf(raw, level=level, strategy=strategy), 1 << 30), (_bgzf(raw, 20000, level, strategy, rng=rng), 150_001),
                               (_bgzf(raw, level=level, strategy=strategy, eof=False), 64 * 1024)):
                dev.reset()
                assert _feed(dev, comp, _lib.FORMAT_FASTQ, 31, step) == len(reads)
                assert np.array_equal(dev.get_node_counts(), expect), (path, step)
                assert dev.get_param("bgzf_carry_bytes") == 0
        assert dev.get_param("bgzf_members") > 0
        # the page-locked staging ring wrapped hundreds of times (8 slots of 4 KiB / 64 KiB instead of 16 MiB): same counts
        comp = _bgzf(raw, level=level, strategy=strategy)
        for slot_kb in (4, 64):
            dev.set_param("debug_bgzf_ring_slot_kb", slot_kb)
            dev.reset()
            assert _feed(dev, comp, _lib.FORMAT_FASTQ, 31, 1 << 30) == len(reads)
            assert np.array_equal(dev.get_node_counts(), expect), slot_kb
        dev.set_param("debug_bgzf_ring_slot_kb", 0)
        # windows announced one call ahead: the next window is staged and its member chain walked under this window's kernel
        for comp2, step in ((comp, 150_001), (_bgzf(raw, 20000, level, strategy, rng=rng), 64 * 1024 + 5), (comp, 1 << 30)):
            for slot_kb in (0, 16):
                dev.set_param("debug_bgzf_ring_slot_kb", slot_kb)
                dev.reset()
                before = dev.get_param("bgzf_prestaged_calls")
                assert _feed_hinted(dev, comp2, _lib.FORMAT_FASTQ, 31, step) == len(reads)
                assert np.array_equal(dev.get_node_counts(), expect), (step, slot_kb)
                assert dev.get_param("bgzf_carry_bytes") == 0
                assert (dev.get_param("bgzf_prestaged_calls") > before) == (step < len(comp2)), step
        dev.set_param("debug_bgzf_ring_slot_kb", 0)
        # two BGZF files one behind the other (cat a.fq.gz b.fq.gz: an empty end-of-file member in the middle)
        cut = raw.index(b"\n@read3000 ") + 1
        cat = _bgzf(raw[:cut], level=level, strategy=strategy) + _bgzf(raw[cut:], 30000, level, strategy)
        for step in (1 << 30, 99_991):
            dev.reset()
            assert _feed(dev, cat, _lib.FORMAT_FASTQ, 31, step) == len(reads)
            assert np.array_equal(dev.get_node_counts(), expect), step
        # a hint the next call does not keep to is dropped: an ordinary call follows
        buf = np.frombuffer(comp, dtype=np.uint8)
        dev.reset()
        used, n1 = dev.map_bgzf(buf[:200_000], fmt=_lib.FORMAT_FASTQ, k=31, first=True, last=False, next_chunk=buf[200_000:300_000])
        assert 0 < used <= 200_000
        assert n1 + _feed_rest(dev, buf[used:], _lib.FORMAT_FASTQ, 31) == len(reads)
        assert np.array_equal(dev.get_node_counts(), expect)


def _feed_rest(dev, buf, fmt, k):
    used, n_rec = dev.map_bgzf(buf, fmt=fmt, k=k, first=False, last=True)
    assert used == len(buf)
    return n_rec


@pytest.mark.parametrize("fmt", ["fastq", "fasta"])
def test_ranks_map_their_member_ranges(kmm, syn, oracle, fmt):
    """Several ranks on one BGZF file (kmer_mapper map under torchrun; emulated here on one handle): rank r maps the
    members of bgzf_ranges.rank_member_range with the head of its first member skipped ("bgzf_head_skip") and only the
    head of the member behind its range taken ("bgzf_tail_stop").  Summed over the ranks: the oracle's counts and every
    record once — for members of 64 KiB and of 3 000 bytes (ranges that start and end inside records), 1 to 9 ranks, a
    rank with nothing to do among them."""
    from kmer_mapper_amd import _lib, bgzf_ranges
    index, genome = syn.make_index(20000, seed=731)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 6000, 31, 240, seed=732)
    reads = [bases[offs[i]:offs[i + 1]].tobytes() for i in range(len(offs) - 1)]
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    rng = np.random.default_rng(733)
    raw = _fastq(reads, rng) if fmt == "fastq" else b"".join(b">r%d\n" % i + r + b"\n" for i, r in enumerate(reads))
    kfmt = _lib.FORMAT_FASTQ if fmt == "fastq" else _lib.FORMAT_FASTA2
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        for block, worlds in ((0xFF00, (2, 5, 40)), (3000, (1, 3, 9))):
            comp = _bgzf(raw, block)
            buf = np.frombuffer(comp, dtype=np.uint8)
            for world in worlds:
                dev.reset()
                total = 0
                for r in range(world):
                    lo, s0, hi, s1 = bgzf_ranges.rank_member_range(comp, fmt, r, world)
                    hi = bgzf_ranges.member_end(comp, hi) if s1 > 0 else hi
                    if lo == hi:
                        continue
                    used, n_rec = dev.map_bgzf(buf[lo:hi], fmt=kfmt, k=31, first=True, last=True, head_skip=s0,
                                               tail_stop=s1 if s1 > 0 else None)
                    assert used == hi - lo
                    total += n_rec
                assert total == len(reads), (block, world)
                assert np.array_equal(dev.get_node_counts(), expect), (block, world)
        with pytest.raises(ValueError):                     # a tail beyond the last member's bytes
            dev.map_bgzf(buf, fmt=kfmt, k=31, first=True, last=True, tail_stop=10 ** 6)
        dev.set_param("bgzf_tail_stop", -1)
        dev.reset()


def test_bgzf_two_line_fasta_and_a_last_line_without_newline(kmm, syn, oracle):
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(8000, seed=711)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 5000, 150, seed=712)
    reads = [bases[offs[i]:offs[i + 1]].tobytes() for i in range(len(offs) - 1)]
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    fa = b"".join(b">s%d\n" % i + r + b"\n" for i, r in enumerate(reads))
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        for raw in (fa, fa[:-1]):
            dev.reset()
            assert _feed(dev, _bgzf(raw), _lib.FORMAT_FASTA2, 31, 200_000) == len(reads)
            assert np.array_equal(dev.get_node_counts(), expect)
        # a stream that ends inside a record is refused at its last chunk, and so is one that ends inside a member
        dev.reset()
        with pytest.raises(ValueError, match="no complete record"):
            _feed(dev, _bgzf(fa[:-200] + b"\n>tail"), _lib.FORMAT_FASTA2, 31, 1 << 30)
        dev.reset()
        with pytest.raises(ValueError, match="ends inside a BGZF member"):
            _feed(dev, _bgzf(fa)[:-40], _lib.FORMAT_FASTA2, 31, 1 << 30)
        dev.reset()


def test_damaged_bgzf_members_are_refused_on_the_device(kmm, syn, oracle):
    """CRC32 and ISIZE of every member are checked on the device before anything is mapped; a flipped payload byte, a wrong
    CRC, a wrong ISIZE, a plain gzip member (no BC subfield) all end in ValueError and leave the counts untouched; the handle
    maps a good chunk afterwards.  Random damage: every outcome is a refusal or the counts zlib's bytes give."""
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(8000, seed=721)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 4000, 150, seed=722)
    reads = [bases[offs[i]:offs[i + 1]].tobytes() for i in range(len(offs) - 1)]
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    rng = np.random.default_rng(723)
    raw = _fastq(reads, rng)
    good = _bgzf(raw)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        first_len = struct.unpack_from("<H", good, 16)[0] + 1
        for what, pos, val in (("payload", first_len + 200, None), ("crc", first_len - 8, None), ("isize", first_len - 4, None)):
            bad = bytearray(good)
            bad[pos] ^= 0x5A
            dev.reset()
            with pytest.raises(ValueError, match="corrupt BGZF member|claims|no BGZF member"):
                _feed(dev, bytes(bad), _lib.FORMAT_FASTQ, 31, 1 << 30)
            assert not dev.get_node_counts().any(), what
        import gzip
        dev.reset()
        with pytest.raises(ValueError, match="no BGZF member"):
            _feed(dev, gzip.compress(raw), _lib.FORMAT_FASTQ, 31, 1 << 30)
        dev.reset()
        assert _feed(dev, good, _lib.FORMAT_FASTQ, 31, 1 << 30) == len(reads)
        assert np.array_equal(dev.get_node_counts(), expect)
        n_refused = n_ok = 0
        for trial in range(40):
            bad = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
            dev.reset()
            try:
                _feed(dev, bytes(bad), _lib.FORMAT_FASTQ, 31, 1 << 30)
                got = dev.get_node_counts()
            except ValueError:
                n_refused += 1
                continue
            # accepted: then the bytes are what zlib makes of the file (damage in a header field nobody reads, ...)
            import io
            try:
                ref = gzip.GzipFile(fileobj=io.BytesIO(bytes(bad))).read()
            except Exception:            # noqa: BLE001
                ref = None
            assert ref == raw and np.array_equal(got, expect), trial
            n_ok += 1
        assert n_refused > 20


def test_cli_maps_a_bgzf_file_through_the_gpu_inflater(kmm, syn, oracle, tmp_path, caplog, monkeypatch):
    """`kmer_mapper map -f reads.fq.gz` on a BGZF file: the members are inflated on the GPU (log line), counts = the oracle's;
    with KMM_CLI_NO_GPU_INFLATE the host inflater gives the same."""
    import argparse
    import logging
    from kmer_mapper_amd.command_line_interface import map_bnp
    from kmer_mapper_amd.gz_io import write_bgzf
    index, genome = syn.make_index(8000, seed=731)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 8000, 20, 200, seed=732)
    reads = [bases[offs[i]:offs[i + 1]].tobytes() for i in range(len(offs) - 1)]
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    path = str(tmp_path / "reads.fq.gz")
    write_bgzf(path, _fastq(reads, np.random.default_rng(5)))
    ns = argparse.Namespace(kmer_index=index, index_bundle=None, reads=path, kmer_size=31, n_threads=8, chunk_size=400_000,
                            output_file=None, debug=None, max_hits_per_kmer=1000, gpu=True, gpu_hash_map_size=0,
                            map_reverse_complements=False)
    with caplog.at_level(logging.INFO):
        got = map_bnp(ns)
    assert np.array_equal(got, expect)
    assert "BGZF members inflated on the GPU" in caplog.text
    monkeypatch.setenv("KMM_CLI_NO_GPU_INFLATE", "1")
    assert np.array_equal(map_bnp(ns), expect)
