"""GPU tests of the host cores' share of the read bytes (csrc/kmm_hostpack.hpp behind kmm_set_param "host_pack_threads",
the reference's `-t`, command_line_interface.py:124-130,168): raw FASTQ / two-line FASTA in HOST memory is packed to a 2-bit
stream + read-start bitset by host threads inside kmm_map_records and mapped by pass 1's 2-bit front end; kmm_map_packed
takes such a stream from the caller.  Bit-exact against the oracle and against the device-side record parser."""
import os

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


def _fastq(reads, eol=b"\n", qual=b"I", rng=None):
    out = []
    for i, r in enumerate(reads):
        q = bytes(rng.choice(np.frombuffer(qual, dtype=np.uint8), size=len(r))) if rng is not None else qual[:1] * len(r)
        out.append(b"@r%d x" % i + eol + r + eol + b"+" + eol + q + eol)
    return np.frombuffer(b"".join(out), dtype=np.uint8)


def _reads(bases, offs):
    return [bases[offs[i]:offs[i + 1]].tobytes() for i in range(len(offs) - 1)]


def test_host_packing_is_on_by_default_where_the_process_has_the_cores(kmm, syn, monkeypatch):
    """Library default: min(16, CPU budget) packing threads when the budget (affinity mask but one CPU, cgroup quota) is at
    least 8 — the reference CLI's `-t 16` (command_line_interface.py:168) — else none; KMM_HOST_PACK_THREADS overrides."""
    index, _ = syn.make_index(2000, seed=5)
    monkeypatch.delenv("KMM_HOST_PACK_THREADS", raising=False)
    with kmm.DeviceIndex.from_index(index, index.max_node_id()) as dev:
        budget = dev.get_param("host_cpu_budget")
        assert 1 <= budget <= (os.cpu_count() or 1)
        assert dev.get_param("host_pack_threads") == (min(16, budget) if budget >= 8 else 0)
    monkeypatch.setenv("KMM_HOST_PACK_THREADS", "3")
    with kmm.DeviceIndex.from_index(index, index.max_node_id()) as dev:
        assert dev.get_param("host_pack_threads") == 3


@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
@pytest.mark.parametrize("read_len", [150, 31, 40, 1000, 9])
def test_records_of_one_length_packed_on_the_host(kmm, syn, oracle, eol, read_len):
    """FASTQ whose reads have one length, in host memory: the host threads pack the sequence lines (RecordsJob), find the
    reads uniform, pass 1 takes packed tiles on the 2-bit stream; the counts are the oracle's on the reads themselves
    (command_line_interface.py:102-111 + mapper.pyx:53-69), whole and cut at arbitrary bytes, '@' and '+' among the
    quality bytes, for both line endings; `consumed` and the record count equal the device parser's."""
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(6000, seed=371)
    mx = index.max_node_id()
    n_reads = 7001
    bases, offs = syn.make_reads(genome, n_reads, read_len, seed=372)
    k = 31 if read_len >= 31 else 7
    if k != 31:
        index, genome = syn.make_index(6000, k=k, seed=371, plant=False)
        mx = index.max_node_id()
        bases, offs = syn.make_reads(genome, n_reads, read_len, seed=372)
    expect, _ = oracle.map_reads(index, mx, bases, offs, k)
    raw = _fastq(_reads(bases, offs), eol, qual=b"@+IF5#", rng=np.random.default_rng(3))
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        dev.set_param("host_pack_threads", 5)
        dev.set_param("host_pack_slice_kb", 64)          # (many slices in a test-size chunk)
        used, n_rec = dev.map_records(raw, fmt=_lib.FORMAT_FASTQ, k=k)
        assert (used, n_rec) == (raw.shape[0], n_reads)
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("host_packed_record_calls") == 1
        dev.reset()
        pos, total, calls = 0, 0, 1
        cuts = []
        while pos < raw.shape[0]:
            used, n_rec = dev.map_records(np.ascontiguousarray(raw[pos:pos + 333337]), fmt=_lib.FORMAT_FASTQ, k=k)
            assert used > 0
            cuts.append((used, n_rec))
            pos += used
            total += n_rec
            calls += 1
        assert total == n_reads
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("host_packed_record_calls") == calls and dev.get_param("direct_batches") == 0
        # the device-side parser cuts the same chunks at the same bytes
        dev.set_param("host_pack_threads", 0)
        dev.reset()
        pos = 0
        for used_h, rec_h in cuts:
            assert dev.map_records(np.ascontiguousarray(raw[pos:pos + 333337]), fmt=_lib.FORMAT_FASTQ, k=k) == (used_h, rec_h)
            pos += used_h
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("host_packed_record_calls") == calls


def test_ragged_records_packed_on_the_host(kmm, syn, oracle):
    """Ragged reads (0 .. 260 bases, lower case, N): the read-start bitset crosses with the stream; FASTQ and two-line
    FASTA; also with reverse complements and the frequency filter; the last record incomplete is left to the caller."""
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(20000, seed=301)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 30000, 0, 260, seed=302)
    reads = _reads(bases, offs)
    fq = _fastq(reads, qual=b"@I+", rng=np.random.default_rng(4))
    fa = np.frombuffer(b"".join(b">h%d\n" % i + r + b"\n" for i, r in enumerate(reads)), dtype=np.uint8)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        dev.set_param("host_pack_threads", 7)
        dev.set_param("host_pack_slice_kb", 128)
        n_calls = 0
        for rc, mf in ((False, 1000), (True, 1000), (False, 1)):
            expect, _ = oracle.map_reads(index, mx, bases, offs, 31, max_index_lookup_frequency=mf, also_revcomp=rc, n_threads=4)
            for fmt, raw in ((_lib.FORMAT_FASTQ, fq), (_lib.FORMAT_FASTA2, fa)):
                dev.reset()
                assert dev.map_records(raw, fmt=fmt, max_index_lookup_frequency=mf, also_revcomp=rc) == (raw.shape[0], len(reads))
                assert np.array_equal(dev.get_node_counts(), expect), (rc, mf, fmt)
                n_calls += 1
        assert dev.get_param("host_packed_record_calls") == n_calls
        e2, _ = oracle.map_reads(index, mx, bases[:offs[-2]], offs[:-1], 31, n_threads=4)
        for fmt, raw in ((_lib.FORMAT_FASTQ, fq), (_lib.FORMAT_FASTA2, fa)):
            dev.reset()
            used, n_rec = dev.map_records(np.ascontiguousarray(raw[:-3]), fmt=fmt)
            assert n_rec == len(reads) - 1 and raw[used - 1] == 10
            assert np.array_equal(dev.get_node_counts(), e2)


def test_host_packed_records_edge_cases_and_error_reports(kmm, syn, oracle):
    """Empty sequence lines, reads shorter than k, reads that end at 32-base word boundaries, a '\\r' inside a sequence line
    (dropped, breaks the read — as the device parser treats it); a byte without a code and a malformed record line are
    reported by the ordinary route with their RAW byte offsets (the reference's encoder / reader raise, util.py:72,
    command_line_interface.py:102-111); a caller's lookup table and device buffers never take the host route."""
    import torch
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(3000, k=5, seed=381, plant=False)
    mx = index.max_node_id()
    g = syn.ACGT[genome]
    rng = np.random.default_rng(382)
    reads, pos = [], 0
    for i in range(4000):
        n = int(rng.choice([0, 1, 4, 5, 6, 11, 12, 16, 27, 31, 32, 33, 64, 1020, 1024, 1019, 300]))
        reads.append(g[pos:pos + n].tobytes())
        pos = (pos + n + 7) % (len(g) - 2000)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 5)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        dev.set_param("host_pack_threads", 4)
        dev.set_param("host_pack_slice_kb", 32)
        raw = _fastq(reads)
        assert dev.map_records(raw, fmt=_lib.FORMAT_FASTQ, k=5) == (raw.shape[0], len(reads))
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("host_packed_record_calls") == 1
        # '\r' in the middle of sequence lines: two reads where there was one
        split, s_reads = bytearray(), []
        for i, r in enumerate(reads):
            if len(r) >= 12 and i % 3 == 0:
                split += b"@r\n" + r[:7] + b"\r" + r[7:] + b"\n+\n" + b"I" * (len(r) + 1) + b"\n"
                s_reads += [r[:7], r[7:]]
            else:
                split += b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n"
                s_reads.append(r)
        sb = np.frombuffer(b"".join(s_reads), dtype=np.uint8)
        so = np.concatenate([[0], np.cumsum([len(r) for r in s_reads])]).astype(np.int64)
        e_split, _ = oracle.map_reads(index, mx, sb, so, 5)
        raw_s = np.frombuffer(bytes(split), dtype=np.uint8)
        for threads in (4, 0):
            dev.set_param("host_pack_threads", threads)
            dev.reset()
            assert dev.map_records(raw_s, fmt=_lib.FORMAT_FASTQ, k=5) == (raw_s.shape[0], len(reads))
            assert np.array_equal(dev.get_node_counts(), e_split), threads
        # ... also when the '\r' is the LAST byte of a 4 KiB tile of the device-side compaction / of a host slice (until round 5
        # the device compaction let k-mers span such a '\r')
        long_read = g[100:100 + 9000].tobytes()
        for at in (4095 - 3, 8191 - 3, 4096 - 3):
            one = np.frombuffer(b"@r\n" + long_read[:at] + b"\r" + long_read[at:] + b"\n+\n" + b"I" * (len(long_read) + 1) + b"\n", dtype=np.uint8)
            two = np.frombuffer(long_read, dtype=np.uint8)
            e_two, _ = oracle.map_reads(index, mx, two, np.array([0, at, len(long_read)], dtype=np.int64), 5)
            for threads in (4, 0):
                dev.set_param("host_pack_threads", threads)
                dev.set_param("host_pack_slice_kb", 4)
                dev.reset()
                assert dev.map_records(one, fmt=_lib.FORMAT_FASTQ, k=5) == (one.shape[0], 1)
                assert np.array_equal(dev.get_node_counts(), e_two), (at, threads)
        dev.set_param("host_pack_slice_kb", 32)
        dev.set_param("host_pack_threads", 4)
        n_host = dev.get_param("host_packed_record_calls")
        # errors: the host packer steps back, the device parser reports
        dev.reset()
        dev.map_records(np.frombuffer(b"@r1\nACGTXCGTAC\n+\nIIIIIIIIII\n", dtype=np.uint8), k=5)
        with pytest.raises(ValueError, match="offset 8"):
            dev.get_node_counts()
        dev.reset()
        broken = raw.copy()
        at = int(np.flatnonzero(raw == ord("G"))[5000])
        broken[at] = ord("-")
        dev.map_records(broken, fmt=_lib.FORMAT_FASTQ, k=5)
        with pytest.raises(ValueError, match="offset %d" % at):
            dev.get_node_counts()
        dev.reset()
        dev.map_records(np.frombuffer(b"@r1\nACGT\nACGT\nIIII\n", dtype=np.uint8), k=3)
        with pytest.raises(ValueError, match="record structure"):
            dev.get_node_counts()
        dev.reset()
        assert dev.map_records(np.frombuffer(b"@r1\nACGT", dtype=np.uint8), k=3) == (0, 0)
        assert dev.get_node_counts().sum() == 0
        assert dev.get_param("host_packed_record_calls") == n_host + 1          # (only the chunk without a whole record)
        # a bad byte in the INCOMPLETE last record is the next chunk's business
        dev.reset()
        tail_bad = np.frombuffer(raw.tobytes() + b"@r\nACXT", dtype=np.uint8)
        assert dev.map_records(tail_bad, fmt=_lib.FORMAT_FASTQ, k=5) == (raw.shape[0], len(reads))
        assert np.array_equal(dev.get_node_counts(), expect)
        n_host = dev.get_param("host_packed_record_calls")
        # not taken: a caller's table, bytes that already lie in HBM
        lut = np.full(256, 0xFF, dtype=np.uint8)
        for i, c in enumerate(b"ACGT"):
            lut[c] = lut[c + 32] = i
        dev.reset()
        dev.map_records(raw, fmt=_lib.FORMAT_FASTQ, k=5, lut=lut)
        assert np.array_equal(dev.get_node_counts(), expect)
        dev.reset()
        d_raw = torch.from_numpy(raw.copy()).cuda()
        torch.cuda.synchronize()
        dev.map_records(d_raw, fmt=_lib.FORMAT_FASTQ, k=5)
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("host_packed_record_calls") == n_host


def test_map_packed_takes_the_callers_own_2bit_stream(kmm, syn, oracle):
    """kmm_map_packed: reads the caller holds as 2-bit codes (16 per uint32, first base lowest — util.py:72-73's packing),
    reads of one length or ragged with a read-start bitset, from host and from device memory, short reads (< 16 bases),
    reverse complements; = the oracle on the reads' letters."""
    import torch
    index, genome = syn.make_index(30000, seed=41)
    mx = index.max_node_id()
    code_of = np.zeros(256, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code_of[c] = code_of[c + 32] = i

    def pack(bases):
        c = code_of[bases].astype(np.uint32)
        c = np.concatenate([c, np.zeros((-len(c)) % 16, dtype=np.uint32)]).reshape(-1, 16)
        return (c << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint64).astype(np.uint32)

    for R, L, k in ((20000, 150, 31), (50001, 37, 21), (30000, 9, 5)):
        idx, gen = (index, genome) if k == 31 else syn.make_index(30000, k=k, seed=41, plant=False)
        bases, offs = syn.make_reads(gen, R, L, seed=42 + L)
        codes = pack(bases)
        with kmm.DeviceIndex.from_index(idx, idx.max_node_id()) as d:
            for rc in (False, True):
                expect, _ = oracle.map_reads(idx, idx.max_node_id(), bases, offs, k, also_revcomp=rc, n_threads=4)
                d.reset()
                d.map_packed(codes, R * L, R, read_len=L, k=k, also_revcomp=rc)
                assert np.array_equal(d.get_node_counts(), expect), (R, L, k, rc)
                d.reset()
                d.map_packed(torch.from_numpy(codes.view(np.int32)).cuda(), R * L, R, read_len=L, k=k, also_revcomp=rc)
                assert np.array_equal(d.get_node_counts(), expect), (R, L, k, rc, "device")
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        bases, offs = syn.make_ragged_reads(genome, 40000, 0, 260, seed=45)
        n = int(offs[-1])
        starts = np.zeros(n // 32 + 1, dtype=np.uint32)
        o = offs[:-1][offs[:-1] < n]
        np.bitwise_or.at(starts, o >> 5, (np.uint32(1) << (o & 31).astype(np.uint32)))
        expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
        dev.reset()
        dev.map_packed(pack(bases), n, len(offs) - 1, read_starts=starts)
        assert np.array_equal(dev.get_node_counts(), expect)
        dev.reset()
        dev.map_packed(torch.from_numpy(pack(bases).view(np.int32)).cuda(), n, len(offs) - 1,
                       read_starts=torch.from_numpy(starts.view(np.int32)).cuda())
        assert np.array_equal(dev.get_node_counts(), expect)
        with pytest.raises(ValueError):
            dev.map_packed(pack(bases), n, 10, read_len=150)            # n_reads * read_len != n_bases
        with pytest.raises(ValueError):
            dev.map_packed(pack(bases), n, len(offs) - 1)               # neither a length nor a bitset


def test_cli_maps_a_plain_fastq_from_its_file_mapping(kmm, syn, oracle, tmp_path, monkeypatch, caplog):
    """`kmer_mapper map -t N` on an uncompressed FASTQ: the chunks are views of the file mapping (reads_io.MmapChunker), the
    host threads pack them inside kmm_map_records, `-t` sets how many (command_line_interface.py:168); counts = the
    oracle's; a file without a final newline and one with CRLF line ends included; -t 1 = the device parser."""
    import argparse
    import logging
    from kmer_mapper_amd.command_line_interface import map_bnp
    monkeypatch.setenv("KMM_RX_MIN_UNITS", "100000")          # (a test-size file takes the radix path)
    index, genome = syn.make_index(8000, seed=61)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 12000, 20, 200, seed=62)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    reads = _reads(bases, offs)
    for name, data in (("lf.fq", _fastq(reads).tobytes()), ("nonl.fq", _fastq(reads).tobytes()[:-1]),
                       ("crlf.fq", _fastq(reads, eol=b"\r\n").tobytes()),
                       ("two_line.fa", b"".join(b">h%d\n" % i + r + b"\n" for i, r in enumerate(reads)))):
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        for t in (6, 1):
            ns = argparse.Namespace(kmer_index=index, index_bundle=None, reads=path, kmer_size=31, n_threads=t, chunk_size=400_000,
                                    output_file=None, debug=None, max_hits_per_kmer=1000, gpu=True, gpu_hash_map_size=0,
                                    map_reverse_complements=False)
            with caplog.at_level(logging.INFO):
                caplog.clear()
                got = map_bnp(ns)
            assert np.array_equal(got, expect), (name, t)
            line = [r.getMessage() for r in caplog.records if "path_taken" in r.getMessage()][0]
            packed = int(line.split(";")[1].split()[0])
            assert (packed > 0) == (t > 1), (name, t, line)


def test_cli_sends_its_packer_threads_to_the_node_of_the_files_pages(kmm, syn, oracle, tmp_path, monkeypatch, caplog):
    """A read file whose page-cache pages lie on another NUMA node than the rank's GPU: the thread that makes the packer pool
    takes that node's CPUs for the map phase (distributed.packer_cpus_near) and gets its own back afterwards.  No second socket
    can be asked for here, so the rank is told it was bound to a node the file is NOT on and that the file's node holds half of
    this process's CPUs: the steering runs for real (affinity cut, pool made inside it, affinity restored), the counts are the
    oracle's, and KMM_CLI_NO_PACKER_STEERING leaves the threads where they are."""
    import argparse
    import logging
    import os
    from kmer_mapper_amd import distributed as D
    from kmer_mapper_amd.command_line_interface import map_bnp
    monkeypatch.setenv("KMM_RX_MIN_UNITS", "100000")
    monkeypatch.setenv("KMM_NO_NUMA_BIND", "1")                # (the fake binding below is the only one)
    index, genome = syn.make_index(8000, seed=71)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 12000, 20, 200, seed=72)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    path = str(tmp_path / "r.fq")
    open(path, "wb").write(_fastq(_reads(bases, offs)).tobytes())
    mine = sorted(os.sched_getaffinity(0))
    half = set(mine[:max(1, len(mine) // 2)])
    sysfs = tmp_path / "node"
    for node in range(8):                                      # whichever node the pages are on: it "owns" that half
        (sysfs / ("node%d" % node)).mkdir(parents=True)
        (sysfs / ("node%d" % node) / "cpulist").write_text(",".join(str(c) for c in sorted(half)) + "\n")
    real = D.packer_cpus_near
    seen = []

    def near(page_nodes, **kw):
        got = real(page_nodes, sysfs=str(sysfs))
        seen.append((dict(page_nodes), got))
        return got

    monkeypatch.setattr(D, "packer_cpus_near", near)
    monkeypatch.setattr(D, "_BOUND", {"before": set(mine), "numa_node": 99})
    ns = argparse.Namespace(kmer_index=index, index_bundle=None, reads=path, kmer_size=31, n_threads=6, chunk_size=400_000,
                            output_file=None, debug=None, max_hits_per_kmer=1000, gpu=True, gpu_hash_map_size=0,
                            map_reverse_complements=False)
    with caplog.at_level(logging.INFO):
        caplog.clear()
        got = map_bnp(ns)
    assert np.array_equal(got, expect)
    assert set(os.sched_getaffinity(0)) == set(mine)            # handed back
    if seen and seen[0][0]:                                     # (the platform said where the pages are)
        assert seen[0][1] is not None and seen[0][1][1] == half
        assert any("its packer threads run there" in r.getMessage() for r in caplog.records)
    monkeypatch.setenv("KMM_CLI_NO_PACKER_STEERING", "1")
    n_before = len(seen)
    assert np.array_equal(map_bnp(ns), expect)
    assert len(seen) == n_before


def test_a_large_count_vector_reaches_pageable_memory_through_the_page_locked_ring(kmm, syn, oracle):
    """kmm_get_node_counts into an ordinary numpy array (what the reference's caller gets, mapper.py): vectors of 64 MiB and
    more cross PCIe through the handle's page-locked ring, the packing threads copying the slots out.  Same vector as into
    page-locked memory, as without threads (the runtime's own copy), and as the oracle's — for sizes that end inside a slot,
    at a slot boundary and one element behind it."""
    index, genome = syn.make_index(30000, seed=921)
    real_mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 4000, 31, 200, seed=922)
    expect_small, _ = oracle.map_reads(index, real_mx, bases, offs, 31, n_threads=4)
    slot = (16 << 20) // 4
    for mx in (5 * slot + 12345, 6 * slot - 1, 6 * slot):
        assert mx > real_mx
        with kmm.DeviceIndex.from_index(index, mx) as dev:
            dev.map_reads(bases, offs, 31)
            got = dev.get_node_counts()
            assert got.shape == (mx + 1,)
            assert np.array_equal(got[:real_mx + 1], expect_small) and not got[real_mx + 1:].any()
            assert np.array_equal(dev.get_node_counts(pinned=True), got)
            dev.set_param("host_pack_threads", 0)
            assert np.array_equal(dev.get_node_counts(), got)


def test_page_locked_buffers_reserved_ahead_are_found_by_the_next_handle(kmm, syn, oracle):
    """kmm_host_reserve / kmm_host_reserve_buffer: page-locked staging memory made ahead of the first map call (the CLI does it
    from a helper thread while the index is uploaded) is put on the process-wide shelf and taken by the handle that asks
    next; the results do not depend on where the buffers came from, bad sizes are refused."""
    from kmer_mapper_amd import _lib
    L = _lib.lib()
    _lib.check(L.kmm_host_reserve(64 << 20))
    _lib.check(L.kmm_host_reserve_buffer(16 << 20))          # (one slot of the staging ring)
    _lib.check(L.kmm_host_reserve_buffer(0))
    with pytest.raises(ValueError):
        _lib.check(L.kmm_host_reserve_buffer(-1))
    with pytest.raises(ValueError):
        _lib.check(L.kmm_host_reserve(-5))
    index, genome = syn.make_index(30000, seed=941)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 30000, 40, 200, seed=942)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    raw = _fastq(_reads(bases, offs))
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        dev.set_param("host_pack_threads", 4)
        used, n_rec = dev.map_records(raw, raw.shape[0], _lib.FORMAT_FASTQ, 31)
        assert used == raw.shape[0] and n_rec == len(offs) - 1
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("host_packed_record_calls") == 1
