"""CPU tests: the oracle against the reference's known answers and the committed goldens."""
import numpy as np
import pytest

from tests.helpers import batch, golden_small, index_from_vector, reference_vectors


@pytest.mark.parametrize("v", reference_vectors()["lookup"], ids=lambda v: v["name"])
def test_lookup_reference_vectors(oracle, v):
    index = index_from_vector(v)
    q = np.array(v["query"], dtype=np.uint64)
    got = oracle.map_kmers(index, v["max_node_id"], q, v["max_index_lookup_frequency"])
    assert got.dtype == np.uint32 and got.shape == (v["max_node_id"] + 1,)
    assert got.tolist() == v["expected_node_counts"]
    if "expected_in_index" in v:
        assert oracle.in_index(index, q).tolist() == v["expected_in_index"]


@pytest.mark.parametrize("v", reference_vectors()["string_lookup"], ids=lambda v: v["name"])
def test_oracle_matches_reference_test_mapping(oracle, v):
    """reference tests/test_mapping.py:33-40: four 3-mers as strings (one in mixed case) -> nodes 0..3, modulo 21;
    the lower-case query 'ccg' must resolve to node 2."""
    from kmer_mapper_amd.kmer_index import KmerIndex
    b = batch(v["node_kmers"])
    node_kmers = oracle.extract(b.bases, b.offsets, v["k"])
    assert node_kmers.shape == (len(v["node_kmers"]),)
    index = KmerIndex.from_flat_kmers(node_kmers, np.arange(len(node_kmers), dtype=np.int64), v["modulo"])
    q = batch([v["query"]])
    qk = oracle.extract(q.bases, q.offsets, v["k"])
    counts = oracle.map_kmers(index, v["max_node_id"], qk, v["max_index_lookup_frequency"])
    assert counts.shape == (v["max_node_id"] + 1,) and counts.sum() == 1 and counts[v["expected_node"]] == 1
    counts = oracle.map_kmers(index, v["max_node_id"], node_kmers, v["max_index_lookup_frequency"])
    assert counts[:len(node_kmers)].tolist() == v["expected_counts_of_node_kmers"] and counts.sum() == len(node_kmers)


@pytest.mark.parametrize("v", reference_vectors()["extract"], ids=lambda v: v["name"])
def test_extract_known_answers(oracle, v):
    b = batch(v["reads"])
    got = oracle.extract(b.bases, b.offsets, v["k"])
    assert got.dtype == np.uint64
    assert got.tolist() == v["expected"]


def test_bit_order_identity_of_reference_test_hashing(oracle):
    """reference tests/test_hashing.py:11-26: for codes s = arange(35) % 4 and k = 31,
    get_kmers(s) XOR-complemented per base (legacy ACTG: code ^ 2) equals the reversed
    np.convolve of the reversed complemented sequence with 4**arange(k).  This holds only if the
    first base of a window sits in the lowest two bits."""
    k = 31
    s = (np.arange(35) % 4).astype(np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[s]
    km = oracle.extract(letters, np.array([0, 35], dtype=np.int64), k)
    mask = np.uint64(4 ** k - 1)
    comp = (km ^ np.uint64(0xAAAAAAAAAAAAAAAA)) & mask
    rc = ((s + 2) % 4)[::-1]
    conv = np.convolve(rc.astype(np.uint64), 4 ** np.arange(k).astype(np.uint64), mode="valid")
    assert np.array_equal(comp, conv[::-1])


def test_invalid_base_raises(oracle):
    b = batch(["ACGTXACGT"])
    with pytest.raises(ValueError):
        oracle.extract(b.bases, b.offsets, 3)


def test_revcomp_matches_string_revcomp(oracle):
    rng = np.random.default_rng(0)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for k in (1, 2, 5, 16, 31):
        s = "".join(rng.choice(list("ACGT"), size=k))
        rc = "".join(comp[c] for c in reversed(s))
        a = oracle.extract(*_b([s]), k)
        b = oracle.extract(*_b([rc]), k)
        assert oracle.revcomp(a, k).tolist() == b.tolist()


def _b(reads):
    bb = batch(reads)
    return bb.bases, bb.offsets


def test_goldens_reproduce(oracle):
    d, index, mx, k = golden_small()
    for name in ("uniform", "ragged"):
        km = oracle.extract(d[name + "_bases"], d[name + "_offsets"], k)
        assert np.array_equal(km, d[name + "_kmers"])
        assert np.array_equal(oracle.map_kmers(index, mx, km), d[name + "_counts"])
        assert np.array_equal(oracle.map_kmers(index, mx, km, 2), d[name + "_counts_maxfreq2"])
        assert np.array_equal(oracle.in_index(index, km), d[name + "_in_index"])
        fused, n = oracle.map_reads(index, mx, d[name + "_bases"], d[name + "_offsets"], k,
                                    n_threads=3, chunk_reads=17)
        assert n == km.shape[0]
        assert np.array_equal(fused, d[name + "_counts"])
        fused_rc, _ = oracle.map_reads(index, mx, d[name + "_bases"], d[name + "_offsets"], k,
                                       also_revcomp=True)
        assert np.array_equal(fused_rc, d[name + "_counts_revcomp"])


def test_oracle_matches_pure_python_loop(oracle):
    """Tiny case re-derived with a literal Python transcription of the lookup semantics."""
    from kmer_mapper_amd import synthetic as syn
    index, genome = syn.make_index(50, k=5, seed=3, plant=False, modulo=37)
    bases, offs = syn.make_reads(genome, 20, 12, seed=4)
    km = oracle.extract(bases, offs, 5)
    expect = np.zeros(index.max_node_id() + 1, dtype=np.uint32)
    for q in km.tolist():
        h = q % index._modulo
        for l in range(index._hashes_to_index[h], index._hashes_to_index[h] + index._n_kmers[h]):
            if int(index._kmers[l]) == q and index._frequencies[l] <= 1000:
                expect[index._nodes[l]] += 1
    assert np.array_equal(oracle.map_kmers(index, index.max_node_id(), km), expect)
    assert expect.sum() > 0


def test_dtype_mismatch_raises_like_cython(oracle):
    v = reference_vectors()["lookup"][1]
    index = index_from_vector(v)
    with pytest.raises(ValueError):
        oracle.map_kmers(index, 15, np.array(v["query"], dtype=np.int64))
    index._nodes = index._nodes.astype(np.int64)
    with pytest.raises(ValueError):
        oracle.map_kmers(index, 15, np.array(v["query"], dtype=np.uint64))
