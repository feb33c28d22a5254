"""CPU tests: host logic, index builder, and that the C-ABI library loads and exports every
symbol include/kmm.h declares (no compute calls — there is no GPU in this tier)."""
import os
import re
import subprocess

import numpy as np
import pytest

from kmer_mapper_amd import _lib
from kmer_mapper_amd import synthetic as syn
from kmer_mapper_amd.kmer_index import KmerIndex
from kmer_mapper_amd.util import ReadBatch, as_read_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "kmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kmm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    L = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), "libkmm.so does not export %s" % s
    assert set(syms) == set(_lib.SIGNATURES), "ctypes table and kmm.h disagree"
    assert L.kmm_version().startswith(b"kmm ")


def test_reader_library_exports_every_declared_symbol():
    """include/kmm_io.h (host-side reader library, g++): same rule as kmm.h."""
    from kmer_mapper_amd import _io
    _io.build()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "kmm_io.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(kmm_io_[a-z_0-9]+)\s*\(", text)))
    L = _io.lib()
    for s in syms:
        assert hasattr(L, s), "libkmm_io.so does not export %s" % s
    assert set(syms) == set(_io.SIGNATURES), "ctypes table and kmm_io.h disagree"


def test_no_gpu_fails_loudly_not_silently():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from kmer_mapper_amd.engine import DeviceIndex
    ix, _ = syn.make_index(20, k=5, plant=False)
    with pytest.raises(_lib.KmmError, match="no HIP device|no CPU fallback"):
        DeviceIndex.from_index(ix)


@pytest.mark.parametrize("host_parser", [False, True])
@pytest.mark.parametrize("name", ["r.fq", "r.fa", "r.fq.gz"])
def test_cli_routes_reach_the_device_and_fail_there_without_a_gpu(tmp_path, name, host_parser):
    """Every route of `kmer_mapper map --gpu` (raw records from the file mapping, a .gz through the reader library, the host
    parser's chunks) runs its host-side preparation — chunker, helper threads, max_node_id — and then fails LOUDLY at the
    device (no CPU fallback): the CPU tier walks the code in front of the first HIP call, so a slip there (an undefined
    name on one route, say) does not wait for the GPU tier to be seen."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    import argparse
    import gzip
    from kmer_mapper_amd.command_line_interface import map_bnp
    index, genome = syn.make_index(300, seed=5)
    bases, offs = syn.make_reads(genome, 50, 60, seed=6)
    text = syn.ACGT_INV[bases] if hasattr(syn, "ACGT_INV") else bases
    seqs = [bytes(np.asarray(text[offs[i]:offs[i + 1]], dtype=np.uint8)) for i in range(50)]
    if name.endswith(".fa"):
        data = b"".join(b">h%d\n" % i + q + b"\n" for i, q in enumerate(seqs))
    else:
        data = b"".join(b"@h%d\n" % i + q + b"\n+\n" + b"I" * len(q) + b"\n" for i, q in enumerate(seqs))
    path = str(tmp_path / name)
    with (gzip.open if name.endswith(".gz") else open)(path, "wb") as f:
        f.write(data)
    ns = argparse.Namespace(kmer_index=index, index_bundle=None, reads=path, kmer_size=31, n_threads=4, chunk_size=2000,
                            output_file=None, debug=None, max_hits_per_kmer=1000, gpu=True, gpu_hash_map_size=0,
                            map_reverse_complements=False, host_parser=host_parser)
    with pytest.raises(_lib.KmmError, match="no HIP device|no CPU fallback"):
        map_bnp(ns)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "kmer_mapper_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no CPU oracle", ""), (f, "mentions oracle")


def test_kmer_index_invariants_and_roundtrip(tmp_path):
    ix, genome = syn.make_index(300, k=31, seed=5)
    M = ix._modulo
    assert ix._hashes_to_index.dtype == np.int32 and ix._n_kmers.dtype == np.int32
    assert ix._nodes.dtype == np.int32 and ix._kmers.dtype == np.uint64
    assert ix._frequencies.dtype == np.uint16
    assert len(ix._hashes_to_index) == M == len(ix._n_kmers)
    assert int(ix._n_kmers.sum()) == len(ix._kmers)
    h = (ix._kmers % np.uint64(M)).astype(np.int64)
    assert np.all(np.diff(h) >= 0)                       # grouped by hash
    for b in np.flatnonzero(ix._n_kmers)[:50]:
        s, c = ix._hashes_to_index[b], ix._n_kmers[b]
        assert np.all(h[s:s + c] == b)
    uk, cnt = np.unique(ix._kmers, return_counts=True)
    assert ix._frequencies.max() == min(cnt.max(), 65535) and cnt.max() >= 1501   # planted hot k-mer
    p = str(tmp_path / "idx.npz")
    ix.to_file(p)
    back = KmerIndex.from_file(p)
    back.convert_to_int32()
    for a in ("_hashes_to_index", "_n_kmers", "_nodes", "_kmers", "_frequencies"):
        assert np.array_equal(getattr(ix, a), getattr(back, a))
    assert back._modulo == M and back.max_node_id() == ix.max_node_id()


def test_read_batch():
    b = ReadBatch.from_strings(["ACGT", "", "TT"])
    assert len(b) == 3 and b.offsets.tolist() == [0, 4, 4, 6]
    assert b.n_kmers(3) == 2 and b.uniform_length is None
    u = ReadBatch.from_strings(["ACGT", "TTTT"])
    assert u.uniform_length == 4
    assert as_read_batch((b.bases, b.offsets)).offsets.tolist() == [0, 4, 4, 6]
    with pytest.raises(ValueError):
        ReadBatch(np.zeros(3, np.uint8), np.array([0, 2], np.int64))


def test_synthetic_hit_rate(oracle):
    ix, genome = syn.make_index(2000, seed=1)
    bases, offs = syn.make_reads(genome, 2000, 150, seed=2)
    counts, n = oracle.map_reads(ix, ix.max_node_id(), bases, offs, 31)
    assert n == 2000 * 120
    assert 0.12 < counts.sum() / n < 0.25               # p ~ 0.18
    assert (bases == ord("N")).sum() > 0 and (bases >= ord("a")).sum() > 0


def test_cli_flags_match_reference(monkeypatch):
    """Flags and defaults of `kmer_mapper map` (reference command_line_interface.py:163-183)."""
    from kmer_mapper_amd import command_line_interface as cli
    seen = {}
    monkeypatch.setattr(cli, "map_bnp", lambda a: seen.update(vars(a)) or "ret")
    # set_defaults(func=map_bnp) captured the original; patch through the parser instead
    import argparse
    real = argparse.ArgumentParser.parse_args

    def fake(self, args=None, namespace=None):
        ns = real(self, args, namespace)
        ns.func = lambda a: seen.update({k: v for k, v in vars(a).items() if k != "func"})
        return ns
    monkeypatch.setattr(argparse.ArgumentParser, "parse_args", fake)
    cli.run_argument_parser(["map", "-i", "x.npz", "-f", "r.fa", "-o", "out"])
    assert seen["kmer_size"] == 31 and seen["n_threads"] == 16 and seen["chunk_size"] == 2500000
    assert seen["max_hits_per_kmer"] == 1000 and seen["gpu"] is False and seen["gpu_hash_map_size"] == 0
    assert seen["map_reverse_complements"] is False and seen["index_bundle"] is None
    cli.run_argument_parser(["map", "-i", "x", "-f", "r", "-o", "o", "-k", "21", "-t", "4", "-c", "100",
                             "-I", "5", "-g", "True", "-s", "7", "-r", "True", "-d", "True", "-b", "bundle"])
    assert (seen["kmer_size"], seen["n_threads"], seen["chunk_size"], seen["max_hits_per_kmer"]) == (21, 4, 100, 5)
    assert seen["gpu"] is True and seen["map_reverse_complements"] is True and seen["index_bundle"] == "bundle"
    with pytest.raises(SystemExit):
        cli.run_argument_parser([])


def test_header_is_plain_c_and_demo_client_links(tmp_path):
    """include/kmm.h must be consumable from C (no C++ / torch types), and a C client must link."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c",
                           os.path.join(inc, "kmm.h")])
    _lib.build()
    exe = str(tmp_path / "kmm_demo")
    libdir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-I" + inc, os.path.join(ROOT, "tools", "kmm_demo.c"),
                           "-o", exe, "-L" + libdir, "-l:libkmm.so", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)


def test_kmer_index_npz_fixture_loads_like_the_reference_reads_it(oracle, golden_dir):
    """The committed Kmer Index file (tests/golden/kmer_index_small.npz: upstream key set, int64 on disk
    [UPSTREAM-UNVERIFIED], written by tests/golden/make_index_fixture.py from the oracle's builder) through the read
    side the reference shows (kmer_mapper/util.py:56-62): from_file -> convert_to_int32 -> remove_ref_offsets; the
    lookup arrays then have the dtypes mapper.pyx:22-29 binds, and the oracle's lookup on them gives the frozen counts."""
    import os
    from kmer_mapper_amd.kmer_index import KmerIndex
    path = os.path.join(golden_dir, "kmer_index_small.npz")
    d = np.load(path)
    assert {"hashes_to_index", "n_kmers", "nodes", "ref_offsets", "kmers", "modulo", "frequencies",
            "allele_frequencies"} <= set(d.files)
    assert d["hashes_to_index"].dtype == np.int64 and d["nodes"].dtype == np.int64 and d["n_kmers"].dtype == np.int64
    ix = KmerIndex.from_file(path)
    assert ix._ref_offsets is not None and ix._allele_frequencies is not None
    with pytest.raises(ValueError, match="dtype mismatch"):
        oracle.map_kmers(ix, int(d["test_max_node_id"]), d["test_query_kmers"])       # int64 tables: like Cython's buffers
    ix.convert_to_int32()
    ix.remove_ref_offsets()
    assert ix._ref_offsets is None
    assert (ix._hashes_to_index.dtype, ix._n_kmers.dtype, ix._nodes.dtype, ix._kmers.dtype, ix._frequencies.dtype) == \
        (np.int32, np.int32, np.int32, np.uint64, np.uint16)
    mx = ix.max_node_id()
    assert mx == int(d["test_max_node_id"]) and ix._modulo == int(d["modulo"])
    q = d["test_query_kmers"]
    assert np.array_equal(oracle.map_kmers(ix, mx, q), d["test_expected_counts"])
    assert np.array_equal(oracle.map_kmers(ix, mx, q, 65535), d["test_expected_counts_maxfreq_65535"])
    assert np.array_equal(oracle.in_index(ix, q), d["test_expected_in_index"])
    assert d["test_expected_counts"].sum() < d["test_expected_counts_maxfreq_65535"].sum()    # the filter bites
    # a file without the optional keys loads too (util.py:60-62 only needs the lookup arrays)
    rebuilt = oracle.build_index(ix._kmers, ix._nodes.astype(np.int64), ix._modulo)
    for name in ("_hashes_to_index", "_n_kmers", "_kmers", "_nodes", "_frequencies"):
        assert np.array_equal(getattr(rebuilt, name), getattr(ix, name)), name


def test_console_script_is_declared_like_the_reference():
    """reference setup.py:31-33 installs `kmer_mapper = kmer_mapper.command_line_interface:main`; pyproject.toml declares
    the same command on this package's CLI module, and that function exists and prints the reference's usage."""
    import importlib
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "pyproject.toml")).read()
    m = re.search(r'^\s*kmer_mapper\s*=\s*"([\w.]+):(\w+)"', text, re.M)
    assert m, "console script missing"
    fn = getattr(importlib.import_module(m.group(1)), m.group(2))
    assert callable(fn)
    from kmer_mapper_amd.command_line_interface import run_argument_parser
    with pytest.raises(SystemExit):
        run_argument_parser([])            # prints help and exits, like the reference (command_line_interface.py:186-188)


def test_host_packer_equals_numpy_packing(tmp_path):
    """csrc/kmm_hostpack.hpp (the host-side 2-bit packing behind kmm_set_param "host_pack_threads"), compiled by itself
    with g++: 16 codes per 32-bit word, first base lowest, A C G T a c g t -> 0..3 and N n -> 0
    (command_line_interface.py:41) — the form pass 1's 2-bit front end reads; the AVX2 body, the scalar tail and the
    threaded job give the bytes numpy gives, the byte behind the last packed one stays untouched, and any byte outside
    the table is reported (the call then takes the ordinary route)."""
    import ctypes
    import subprocess
    src = tmp_path / "shim.cpp"
    src.write_text('#include "kmm_hostpack.hpp"\n'
                   'extern "C" int shim_pack(const uint8_t *s, size_t n, uint8_t *d) { return kmm_hostpack::pack2(s, n, d) ? 1 : 0; }\n'
                   'extern "C" int shim_pack_scalar(const uint8_t *s, size_t n, uint8_t *d) { return kmm_hostpack::pack2_scalar(s, n, d) ? 1 : 0; }\n'
                   'extern "C" int shim_pack_avx2(const uint8_t *s, size_t n, uint8_t *d) { return kmm_hostpack::pack2_avx2(s, n, d) ? 1 : 0; }\n'
                   'extern "C" int shim_pack_avx512(const uint8_t *s, size_t n, uint8_t *d) { return kmm_hostpack::pack2_avx512(s, n, d) ? 1 : 0; }\n'
                   'extern "C" int shim_job(const uint8_t *s, size_t n, uint8_t *d, size_t chunk, int threads) {\n'
                   '    kmm_hostpack::Job j; j.start(s, n, d, chunk, threads);\n'
                   '    for (size_t c = 0; c < j.n_chunks; ++c) j.wait_chunk(c);\n'
                   '    j.join(); return j.bad.load() ? 0 : 1; }\n')
    so = str(tmp_path / "shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread",
                           "-I" + os.path.join(ROOT, "kmer_mapper_amd", "csrc"), str(src), "-o", so])
    lib = ctypes.CDLL(so)
    flags = open("/proc/cpuinfo").read()
    packers = [lib.shim_pack, lib.shim_pack_scalar]            # (dispatching entry, scalar, then every vector body the CPU has)
    packers += [lib.shim_pack_avx2] if " avx2" in flags else []
    packers += [lib.shim_pack_avx512] if "avx512vbmi" in flags and "avx512bw" in flags else []
    for f in packers:
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.shim_job.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)
    code_of = np.full(256, 0xFF, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code_of[c] = code_of[c + 32] = i
    code_of[ord("N")] = code_of[ord("n")] = 0

    def numpy_pack(a):
        c = code_of[a].astype(np.uint8)
        c = np.concatenate([c, np.zeros((-len(c)) % 4, dtype=np.uint8)]).reshape(-1, 4)
        return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)

    for n in (1, 3, 4, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000, 4099, 100_003):
        a = np.ascontiguousarray(alphabet[rng.integers(0, len(alphabet), size=n)])
        want = numpy_pack(a)
        for f in packers:
            out = np.full(len(want) + 8, 0xAA, dtype=np.uint8)
            assert f(a.ctypes.data, n, out.ctypes.data) == 1
            assert np.array_equal(out[:len(want)], want) and (out[len(want):] == 0xAA).all(), n
        for bad in (b"X", b"@", b"\n", b"-", b"\x00", b"\xc1", b"1"):
            b = a.copy()
            b[rng.integers(0, n)] = bad[0]
            out = np.zeros(len(want) + 8, dtype=np.uint8)
            for f in packers:
                assert f(b.ctypes.data, n, out.ctypes.data) == 0, (n, bad)
    n = 3_000_017
    a = np.ascontiguousarray(alphabet[rng.integers(0, len(alphabet), size=n)])
    want = numpy_pack(a)
    for threads, chunk in ((1, 1 << 20), (4, 1 << 18), (7, 64)):
        if chunk == 64:
            a2, want2 = a[:20_000], numpy_pack(a[:20_000])
        else:
            a2, want2 = a, want
        out = np.full(len(want2) + 8, 0xAA, dtype=np.uint8)
        assert lib.shim_job(a2.ctypes.data, len(a2), out.ctypes.data, chunk, threads) == 1
        assert np.array_equal(out[:len(want2)], want2) and (out[len(want2):] == 0xAA).all(), (threads, chunk)
    b = a.copy()
    b[2_000_000] = ord("?")
    out = np.zeros(len(want) + 8, dtype=np.uint8)
    assert lib.shim_job(b.ctypes.data, n, out.ctypes.data, 1 << 18, 4) == 0


# ---------------------------------------------------------------- host records packer (csrc/kmm_hostpack.hpp RecordsJob)
_CODE = {c: i for i, c in enumerate(b"ACGT")}
_CODE.update({c: i for i, c in enumerate(b"acgt")})
_CODE[ord("N")] = _CODE[ord("n")] = 0


def _records_rule(raw, period):
    """Byte-wise restatement of what kmm_map_records makes of a raw chunk (csrc/kmm_records.hpp, include/kmm.h): lines are
    cut at '\\n'; `consumed` = the byte behind the last newline whose 1-based count is a multiple of the period; line
    index mod period == 1 is the sequence line; '\\r' is dropped and breaks the read; header lines start with '@' / '>',
    the third line of a FASTQ record with '+'.  Returns (ok, consumed, records, codes, read starts, uniform length)."""
    hc = ord("@") if period == 4 else ord(">")
    nl = [i for i, c in enumerate(raw) if c == 10]
    target = len(nl) - len(nl) % period
    cut = nl[target - 1] + 1 if target else 0
    flat, marks, pending, ok, line, at_start = [], [], False, True, 0, True
    for i in range(cut):
        c, phase = raw[i], line % period
        if at_start:
            if phase == 0 and c != hc:
                ok = False
            if period == 4 and phase == 2 and c != ord("+"):
                ok = False
            if phase == 1:
                pending = True
        at_start = False
        if c == 10:
            line += 1
            at_start = True
            continue
        if phase == 1:
            if c == 13:
                pending = True
                continue
            if c not in _CODE:
                ok = False
                continue
            if pending:
                marks.append(len(flat))
                pending = False
            flat.append(_CODE[c])
    recs, n, L = target // period, len(flat), 0
    if recs and n and n % recs == 0 and marks == list(range(0, n, n // recs)) and len(marks) == recs:
        L = n // recs
    return ok, cut, recs, flat, marks, L


def test_host_records_packer_equals_the_device_parsers_rules(tmp_path):
    """kmm_hostpack::RecordsJob — what the host threads make of raw FASTQ / two-line FASTA bytes when kmm_map_records is
    handed host memory (the reference's `-t` workers parse and encode the chunks, command_line_interface.py:102-111,
    124-130): the 2-bit stream, the read-start bitset, `consumed`, the record count and the one-length verdict equal the
    byte-wise rule above for LF and CRLF files, ragged and one-length reads, lower case and N, '@'-leading quality lines,
    records cut anywhere, '\\r' inside sequence lines and at slice boundaries, on 1, 3 and 8 threads, with every
    instruction set the packer has; a byte without a code or a malformed record line makes it refuse the chunk; the
    buffers need not be zeroed, and the words behind the last base come back zero."""
    import ctypes
    import subprocess
    import sys
    src = tmp_path / "shim.cpp"
    src.write_text('#include "kmm_hostpack.hpp"\n'
                   'extern "C" int shim_records(const uint8_t *raw, size_t n, int period, int threads, uint64_t *codes, uint32_t *bits, int64_t *out) {\n'
                   '    kmm_hostpack::RecordsJob job; job.prepare(raw, n, period, codes, bits, (size_t)256 << 10);\n'
                   '    kmm_hostpack::Workers w(threads); w.start([&](int) { job.run(); });\n'
                   '    job.wait_packed_prefix(job.n_slices()); w.wait();\n'
                   '    const kmm_hostpack::RecordsResult r = job.finish();\n'
                   '    out[0] = r.ok; out[1] = r.consumed; out[2] = r.n_records; out[3] = r.n_bases; out[4] = r.uniform_len; return 0; }\n'
                   'extern "C" int shim_budget() { return kmm_hostpack::cpu_budget(); }\n')
    so = str(tmp_path / "shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread",
                           "-I" + os.path.join(ROOT, "kmer_mapper_amd", "csrc"), str(src), "-o", so])
    rng = np.random.default_rng(77)

    def fastq(n_rec, lens, eol=b"\n", qual=b"I", alpha=b"ACGT"):
        parts = []
        for i in range(n_rec):
            seq = bytes(rng.choice(np.frombuffer(alpha, dtype=np.uint8), size=lens(i)))
            q = bytes(rng.choice(np.frombuffer(qual, dtype=np.uint8), size=len(seq)))
            parts.append(b"@r%d x" % i + eol + seq + eol + b"+" + eol + q + eol)
        return b"".join(parts)

    big = fastq(6000, lambda i: 150)                        # several 256 KiB slices
    cases = [("one length", big, 4), ("one length crlf", fastq(1500, lambda i: 150, eol=b"\r\n"), 4),
             ("ragged", fastq(2500, lambda i: int(rng.integers(0, 300)), qual=b"@+IF#5", alpha=b"ACGTacgtNn"), 4),
             ("ragged crlf", fastq(1500, lambda i: int(rng.integers(1, 200)), eol=b"\r\n", qual=b"@+I"), 4),
             ("three records", fastq(3, lambda i: 5), 4), ("one long read", fastq(2, lambda i: 700_000), 4),
             ("tiny reads", fastq(20_000, lambda i: int(rng.integers(0, 4))), 4),
             ("cut in the quality line", big[:-100], 4), ("cut in the sequence line", big[:-200], 4),
             ("cut behind the header", big[:len(big) - 320], 4), ("no final newline", big[:-1], 4),
             ("empty", b"", 4), ("no line", b"@abc", 4), ("one line", b"@abc\n", 4)]
    fa = b"".join(b">s%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(0, 500)))) + b"\n"
                  for i in range(2500))
    cases += [("fasta", fa, 2), ("fasta cut", fa[:-3], 2),
              ("fasta one length", b"".join(b">s%d\n" % i + b"ACGTTGCAAC" * 10 + b"\n" for i in range(4000)), 2)]
    for pos, ch in ((1000, b"X"), (len(big) // 2, b"\n"), (len(big) - 50, b"?"), (0, b">"), (307, b"\x00"), (5000, b"\xc1")):
        d = bytearray(big)
        d[pos:pos + 1] = ch
        cases.append(("damaged at %d" % pos, bytes(d), 4))
    d = bytearray(big)
    for pos in rng.integers(0, len(d), size=40):
        d[pos] = 13
    cases.append(("carriage returns anywhere", bytes(d), 4))
    for pos in (262143, 262144, 262145, 524287, 524288):      # the slice boundaries
        d = bytearray(big)
        if d[pos] != 10:
            d[pos] = 13
        cases.append(("carriage return at %d" % pos, bytes(d), 4))
    for it in range(150):                                      # soup with many line ends
        raw = bytes(rng.choice(np.frombuffer(b"ACGT\n\n\r@+>NnX", dtype=np.uint8), size=int(rng.integers(1, 3000))))
        cases += [("soup %d" % it, raw, 4), ("soup %d" % it, raw, 2)]

    def run(isa):
        env = dict(os.environ, KMM_HOSTPACK_ISA=str(isa))
        code = ("import sys, pickle, ctypes, numpy as np\n"
                "cases = pickle.load(open(sys.argv[2], 'rb'))\n"
                "lib = ctypes.CDLL(sys.argv[1])\n"
                "lib.shim_records.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]\n"
                "res = []\n"
                "for name, raw, period in cases:\n"
                "    a = np.frombuffer(raw, dtype=np.uint8); n = len(a)\n"
                "    for threads in ((1, 3, 8) if n > 5000 else (2,)):\n"
                "        codes = np.full(n // 32 + 80, 0xAAAAAAAAAAAAAAAA, dtype=np.uint64); bits = np.full(n // 32 + 20, 0xAAAAAAAA, dtype=np.uint32)\n"
                "        out = np.zeros(8, dtype=np.int64)\n"
                "        lib.shim_records(a.ctypes.data, n, period, threads, codes.ctypes.data, bits.ctypes.data, out.ctypes.data)\n"
                "        res.append((name, period, threads, out.copy(), codes, bits))\n"
                "pickle.dump(res, open(sys.argv[3], 'wb'))\n")
        import pickle
        pickle.dump(cases, open(tmp_path / "cases.pkl", "wb"))
        subprocess.check_call([sys.executable, "-c", code, so, str(tmp_path / "cases.pkl"), str(tmp_path / "res.pkl")], env=env)
        return pickle.load(open(tmp_path / "res.pkl", "rb"))

    expect = {(name, period): _records_rule(raw, period) for name, raw, period in cases}
    n_refused = n_uniform = 0
    for isa in (0, 1, 2):          # scalar, AVX2 census, AVX-512 VBMI (the packer falls back by itself where the CPU lacks it)
        if isa == 2 and "avx512vbmi" not in open("/proc/cpuinfo").read():
            continue
        for name, period, threads, out, codes, bits in run(isa):
            ok, cut, recs, flat, marks, L = expect[(name, period)]
            tag = (isa, name, period, threads)
            if not ok:
                assert out[0] == 0, tag
                n_refused += 1
                continue
            assert out[0] == 1, tag
            assert (out[1], out[2], out[3], out[4]) == (cut, recs, len(flat), L), (tag, out[:5], cut, recs, len(flat), L)
            n_uniform += 1 if L else 0
            n = len(flat)
            idx = np.arange(n)
            c8 = codes.view(np.uint8)
            assert np.array_equal((c8[idx >> 2] >> ((idx & 3) * 2).astype(np.uint8)) & 3, np.array(flat, dtype=np.uint8)), (tag, "codes")
            want = np.zeros(n, dtype=bool)
            want[marks] = True
            assert np.array_equal(((bits[idx >> 5] >> (idx & 31).astype(np.uint32)) & 1).astype(bool), want), (tag, "read starts")
            w = n >> 5
            assert not codes[w + 1:w + 40].any() and not bits[w + 1:w + 3].any(), (tag, "words behind the last base")
            assert int(codes[w]) >> (2 * (n & 31)) == 0 and int(bits[w]) >> (n & 31) == 0, (tag, "bits behind the last base")
    assert n_refused and n_uniform
    lib = ctypes.CDLL(so)
    assert 1 <= lib.shim_budget() <= (os.cpu_count() or 1)


def test_gpu_inflater_decodes_like_zlib_on_the_cpu(tmp_path):
    """csrc/kmm_gpu_inflate.hpp — the decoder every GPU thread runs on one BGZF member (kmm_map_bgzf) — compiled by itself with
    g++ and run on the CPU: deflate streams of every block type and compression setting inflate to zlib's bytes, nothing is
    written behind the output, CRC32 equals zlib's, BGZF member framing is parsed and checked, and damaged streams end in an
    error code (or in the bytes zlib gives) without a stray access."""
    import ctypes
    import struct
    import subprocess
    import zlib
    src = tmp_path / "shim.cpp"
    src.write_text('#include "kmm_gpu_inflate.hpp"\n#include <vector>\n'
                   'static std::vector<uint32_t> g_crc;\n'
                   'static const uint32_t *crcT() { if (g_crc.empty()) { g_crc.resize(2048); for (int k = 0; k < 8; ++k) for (uint32_t b = 0; b < 256; ++b) '
                   'g_crc[k * 256 + b] = kmm_gz::crc_table_entry(k, b); } return g_crc.data(); }\n'
                   'extern "C" int gz_stream(const uint8_t *in, uint32_t n_in, uint8_t *out, uint32_t n_out) {\n'
                   '    std::vector<uint16_t> prim(kmm_gz::PRIM_WORDS), sec(kmm_gz::SEC_WORDS); std::vector<uint64_t> list(kmm_gz::LIST_ALLOC);\n'
                   '    return kmm_gz::inflate_stream(in, n_in, out, n_out, prim.data(), sec.data(), list.data()); }\n'
                   'extern "C" int gz_member(const uint8_t *m, uint32_t msize, uint8_t *out, uint32_t n_out) {\n'
                   '    std::vector<uint16_t> prim(kmm_gz::PRIM_WORDS), sec(kmm_gz::SEC_WORDS); std::vector<uint64_t> list(kmm_gz::LIST_ALLOC);\n'
                   '    return kmm_gz::inflate_bgzf_member(m, msize, out, n_out, prim.data(), sec.data(), list.data(), crcT()); }\n'
                   'extern "C" uint32_t gz_member_size(const uint8_t *p, uint64_t n) { return kmm_gz::bgzf_member_size(p, n); }\n'
                   'extern "C" uint32_t gz_crc(const uint8_t *p, uint32_t n) { return kmm_gz::crc32_sliced(crcT(), p, n); }\n')
    so = str(tmp_path / "shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "kmer_mapper_amd", "csrc"), str(src), "-o", so])
    lib = ctypes.CDLL(so)
    lib.gz_stream.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    lib.gz_member.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    lib.gz_crc.argtypes = [ctypes.c_char_p, ctypes.c_uint32]
    lib.gz_crc.restype = ctypes.c_uint32
    lib.gz_member_size.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
    lib.gz_member_size.restype = ctypes.c_uint32
    rng = np.random.default_rng(91)

    def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
        return c.compress(data) + c.flush()

    def fastq(n):
        return b"".join(b"@SRR1.%d %d/1\n" % (i, i) + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=150)) + b"\n+\n" +
                        bytes(rng.choice(np.frombuffer(b"FFFFFF:,#", dtype=np.uint8), size=150)) + b"\n" for i in range(n))

    datas = [b"", b"a", b"abc" * 1000, bytes(rng.integers(0, 256, size=65000, dtype=np.uint8)), fastq(200), b"A" * 65000,
             bytes(rng.integers(0, 4, size=40000, dtype=np.uint8)), b"".join(bytes([i % 251]) * (i % 300) for i in range(500))[:65000]]
    for di, data in enumerate(datas):
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                for mem in (1, 9):
                    comp = deflate(data, level, strategy, mem)
                    out = np.full(len(data) + 16, 0xAA, dtype=np.uint8)
                    assert lib.gz_stream(comp, len(comp), out.ctypes.data, len(data)) == 0, (di, level, strategy, mem)
                    assert out[:len(data)].tobytes() == data and (out[len(data):] == 0xAA).all(), (di, level, strategy, mem)
    data = fastq(150)                                       # several blocks, empty stored blocks in between
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = b"".join(c.compress(data[i:i + 5000]) + c.flush(zlib.Z_SYNC_FLUSH if (i // 5000) % 2 else zlib.Z_FULL_FLUSH)
                    for i in range(0, len(data), 5000)) + c.flush()
    out = np.zeros(len(data), dtype=np.uint8)
    assert lib.gz_stream(comp, len(comp), out.ctypes.data, len(data)) == 0 and out.tobytes() == data
    for n in (0, 1, 7, 8, 9, 1000, 65280):
        d = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        assert lib.gz_crc(d, n) == (zlib.crc32(d) & 0xFFFFFFFF)
    for data in datas:
        payload = deflate(data)
        bsize = 18 + len(payload) + 8
        m = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + payload +
             struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
        assert lib.gz_member_size(m, len(m)) == len(m) and lib.gz_member_size(m, 17) == 0 and lib.gz_member_size(b"\x1f\x8b\x08\x00" + m[4:], len(m)) == 0
        out = np.zeros(len(data) + 16, dtype=np.uint8)          # (the decoder may READ 16 bytes behind the write position)
        assert lib.gz_member(m, len(m), out.ctypes.data, len(data)) == 0 and out[:len(data)].tobytes() == data
        bad = bytearray(m)
        bad[-8] ^= 1
        assert lib.gz_member(bytes(bad), len(m), out.ctypes.data, len(data)) == 11          # CRC32
        assert lib.gz_member(m, len(m), out.ctypes.data, len(data) + 1) == 1                # ISIZE of the plan != the trailer's
    data = fastq(200)
    comp = deflate(data)
    outcomes = {}
    for it in range(1500):
        b = bytearray(comp)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        if rng.integers(0, 4) == 0:
            b = b[:int(rng.integers(0, len(b)))]
        out = np.zeros(len(data) + 16, dtype=np.uint8)
        rc = lib.gz_stream(bytes(b), len(b), out.ctypes.data, len(data))
        outcomes[rc] = outcomes.get(rc, 0) + 1
        if rc == 0:
            try:
                ref = zlib.decompressobj(-15).decompress(bytes(b))
            except zlib.error:
                ref = None
            assert ref is not None and ref[:len(data)] == out[:len(data)].tobytes(), "accepted a stream zlib refuses"
    assert sum(v for k2, v in outcomes.items() if k2 != 0) > 1000


def test_gpu_inflater_touches_no_byte_outside_its_buffers_under_the_sanitizers(tmp_path):
    """The decoder the GPU runs one lane per member (csrc/kmm_gpu_inflate.hpp), built for the CPU with AddressSanitizer and
    UBSan (tests/gz_fuzz_main.cpp; GPU sanitizers are not to be had): 500 rounds x (an intact stream + five damaged versions
    — a bit flipped, truncated, eight bytes overwritten, a byte inserted, the second half zeroed — with the right or a wrong
    claimed size) in exact-size heap buffers.  After a symbol that cannot be, a lane decodes ON until its round ends (one exit
    from the symbol loop): every access must stay inside the buffers whatever the bits say.  Intact streams come out right,
    nothing crashes, nothing is reported."""
    exe = str(tmp_path / "gz_fuzz")
    src = os.path.join(ROOT, "tests", "gz_fuzz_main.cpp")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "kmer_mapper_amd", "csrc"), src, "-o", exe, "-lz"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and ("asan" in build.stderr or "ubsan" in build.stderr or "sanitize" in build.stderr):
        pytest.skip("no sanitizer runtime on this box: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, "500"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "500 rounds" in run.stdout and "ERROR" not in run.stderr and "runtime error" not in run.stderr, (run.stdout + run.stderr)[-3000:]


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_records_packer_under_the_sanitizers(tmp_path, sanitizer):
    """csrc/kmm_hostpack.hpp RecordsJob with ThreadSanitizer, then with AddressSanitizer + UBSan (tests/hostpack_tsan_main.cpp):
    raw FASTQ of random and of one read length, LF and CRLF, with and without an unfinished last record, slices of 1-32 KiB
    (hundreds of hand-overs along the chained prefix, shared boundary words OR-ed in atomically), buffers full of garbage at the
    start: one thread and six threads give the same stream, bitset and counts; no race, no stray access."""
    exe = str(tmp_path / "hostpack_san")
    src = os.path.join(ROOT, "tests", "hostpack_tsan_main.cpp")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=" + sanitizer, "-I" + os.path.join(ROOT, "kmer_mapper_amd", "csrc"),
           src, "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and ("san" in build.stderr and "cannot find" in build.stderr):
        pytest.skip("no sanitizer runtime on this box: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=600)
    out = run.stdout + run.stderr
    if "ThreadSanitizer" in out and "unexpected memory mapping" in out:
        pytest.skip("ThreadSanitizer cannot run in this address-space layout")
    assert run.returncode == 0 and "40 rounds" in run.stdout, out[-3000:]
    assert "WARNING: ThreadSanitizer" not in out and "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
