"""CPU tests of the chunked FASTA/FASTQ(+gz) reader."""
import numpy as np
import pytest

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
