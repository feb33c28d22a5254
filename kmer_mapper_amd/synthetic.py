"""Deterministic synthetic workloads for parity tests and bench.py.

Genome G: Lg = 4*N + k uniform bases (seed 1).  Index: the k-mers at positions p = 4*i (i < N),
node = i (uniform) or i mod 1000 (skewed); 0.1 % of them duplicated under 2-3 further nodes and
one k-mer planted 1500 times (frequency > 1000 -> filtered by the default
max_index_lookup_frequency, mapper.pyx:64-66); modulo = smallest prime >= 2*N.
Reads: R windows G[s:s+L], s uniform (seed 2), 1 % substitutions and 0.05 % 'N' (seed 3),
10 % of the reads lower-cased.  Expected hit rate ~ (N/Lg) * 0.99^k ~ 0.18.

No reference code involved; the reference ships no data generator (its tests need absent files,
tests/test_hashing.py:36, tests/test_reading.py).
"""
import numpy as np

from .kmer_index import KmerIndex

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def next_prime(n):
    n = int(n)
    if n <= 2:
        return 2
    n |= 1
    while True:
        r = int(n ** 0.5) + 1
        if all(n % d for d in range(3, r, 2)):
            return n
        n += 2


def make_genome(n_bases, seed=1):
    """uint8 codes 0..3 (A,C,G,T)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 4, size=int(n_bases), dtype=np.uint8)


def pack_kmers_at(codes, positions, k):
    """First base in the lowest two bits (tests/test_hashing.py:13-26 identity)."""
    positions = np.asarray(positions, dtype=np.int64)
    out = np.zeros(positions.shape[0], dtype=np.uint64)
    for j in range(k):
        out |= codes[positions + j].astype(np.uint64) << np.uint64(2 * j)
    return out


def pack_kmers_strided(codes, n, stride, k):
    """pack_kmers_at(codes, arange(n) * stride, k) without the index arrays: k strided sweeps (a 3e8-k-mer index
    in seconds instead of minutes)."""
    out = np.zeros(n, dtype=np.uint64)
    tmp = np.empty(n, dtype=np.uint64)
    for j in range(k):
        np.copyto(tmp, codes[j:j + (n - 1) * stride + 1:stride], casting="unsafe")
        tmp <<= np.uint64(2 * j)
        out |= tmp
    return out


def pack_kmers_strided_torch(codes, n, stride, k, device=0):
    """Same integers as pack_kmers_strided, packed on the GPU with torch (setup of >= 5e7-k-mer test indexes)."""
    import torch
    g = torch.from_numpy(codes).to("cuda:%d" % device)
    out = torch.zeros(n, dtype=torch.int64, device=g.device)          # k <= 31: 62 bits, no sign involved
    for j in range(k):
        out |= g[j:j + (n - 1) * stride + 1:stride].to(torch.int64) << (2 * j)
    res = out.cpu().numpy().view(np.uint64)
    del g, out
    return res


def make_index(n_kmers, k=31, seed=1, skewed=False, plant=True, modulo=None, gpu_builder=False, device=0):
    """Returns (KmerIndex, genome codes).  gpu_builder: build the index arrays with kmm_build_index
    (bit-identical to the numpy construction, much faster for 1e8 entries)."""
    N = int(n_kmers)
    genome = make_genome(4 * N + k, seed)
    if gpu_builder and N >= 50_000_000:
        kmers = pack_kmers_strided_torch(genome, N, 4, k, device)
    else:
        kmers = pack_kmers_strided(genome, N, 4, k)
    nodes = (np.arange(N, dtype=np.int64) % 1000) if skewed else np.arange(N, dtype=np.int64)
    n_nodes = 1000 if skewed else N
    if plant and N >= 8:
        rng = np.random.Generator(np.random.PCG64(seed + 1000))
        n_dup = max(1, N // 1000)
        dup = rng.choice(N, size=n_dup, replace=False)
        reps = rng.integers(1, 3, size=n_dup)                     # 1-2 extra entries -> 2-3 nodes
        dup_k = np.repeat(kmers[dup], reps)
        dup_n = rng.integers(0, n_nodes, size=dup_k.shape[0])
        hot = kmers[int(rng.integers(0, N))]                      # one k-mer with frequency > 1000
        hot_k = np.full(1500, hot, dtype=np.uint64)
        hot_n = rng.integers(0, n_nodes, size=1500)
        kmers = np.concatenate([kmers, dup_k, hot_k])
        nodes = np.concatenate([nodes, dup_n, hot_n])
    if modulo is None:
        modulo = next_prime(2 * N)
    if gpu_builder:
        index = KmerIndex.from_flat_kmers_gpu(kmers, nodes.astype(np.int64), modulo, device=device)
    else:
        index = KmerIndex.from_flat_kmers(kmers, nodes.astype(np.int64), modulo)
    return index, genome


def make_reads(genome, n_reads, read_len=150, seed=2, sub_rate=0.01, n_rate=0.0005,
               lower_frac=0.1):
    """Returns (bases uint8[n_reads*read_len] ASCII, offsets int64[n_reads+1])."""
    R, L = int(n_reads), int(read_len)
    rng = np.random.Generator(np.random.PCG64(seed))
    starts = rng.integers(0, genome.shape[0] - L + 1, size=R, dtype=np.int64)
    codes = genome[starts[:, None] + np.arange(L, dtype=np.int64)[None, :]]
    rng3 = np.random.Generator(np.random.PCG64(seed + 1))
    sub = rng3.random(size=codes.shape) < sub_rate
    codes = np.where(sub, (codes + rng3.integers(1, 4, size=codes.shape, dtype=np.uint8)) & 3, codes)
    ascii_ = ACGT[codes]
    ascii_[rng3.random(size=codes.shape) < n_rate] = ord("N")
    lower = rng3.random(size=R) < lower_frac
    ascii_[lower] |= 0x20
    offsets = np.arange(R + 1, dtype=np.int64) * L
    return np.ascontiguousarray(ascii_.reshape(-1)), offsets


def make_ragged_reads(genome, n_reads, min_len=0, max_len=300, seed=5, **kw):
    """Reads of varying length (including shorter than k and empty) for the general path."""
    R = int(n_reads)
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = rng.integers(min_len, max_len + 1, size=R, dtype=np.int64)
    bases, _ = make_reads(genome, R, max_len, seed=seed + 1, **kw)
    bases = bases.reshape(R, max_len)
    keep = np.arange(max_len)[None, :] < lens[:, None]
    offsets = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    return np.ascontiguousarray(bases[keep]), offsets


def make_reads_torch(genome_ascii_dev, n_reads, read_len=150, seed=2, sub_rate=0.01,
                     n_rate=0.0005, lower_frac=0.1):
    """Same distribution generated on the GPU with torch (bench-scale batches stay in HBM).
    genome_ascii_dev: uint8 ASCII genome tensor on the device.  Returns uint8[n_reads*read_len]."""
    import torch
    dev = genome_ascii_dev.device
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    R, L = int(n_reads), int(read_len)
    out = torch.empty((R, L), dtype=torch.uint8, device=dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    step = 1 << 20
    ar = torch.arange(L, device=dev, dtype=torch.int64)[None, :]
    for r0 in range(0, R, step):
        r1 = min(R, r0 + step)
        n = r1 - r0
        starts = torch.randint(0, genome_ascii_dev.shape[0] - L + 1, (n,), generator=g, device=dev)
        rd = genome_ascii_dev[starts[:, None] + ar]
        sub = torch.rand((n, L), generator=g, device=dev) < sub_rate
        rnd = lut[torch.randint(0, 4, (n, L), generator=g, device=dev)]
        rd = torch.where(sub, rnd, rd)
        nmask = torch.rand((n, L), generator=g, device=dev) < n_rate
        rd = torch.where(nmask, torch.full_like(rd, ord("N")), rd)
        low = torch.rand((n, 1), generator=g, device=dev) < lower_frac
        rd = torch.where(low, rd | 0x20, rd)
        out[r0:r1] = rd
    return out.reshape(-1)


class DeviceSyntheticIndex:
    """A synthetic Kmer Index whose five arrays live in HBM as torch tensors (duck-typed like the reference's index
    object, mapper.pyx:22-29: DeviceIndex.from_index takes it as it is).  to_host() gives the numpy KmerIndex a
    host-side checker needs."""

    def __init__(self, h2i, nk, kmers, nodes, freqs, modulo, max_node):
        self._hashes_to_index, self._n_kmers, self._kmers, self._nodes, self._frequencies = h2i, nk, kmers, nodes, freqs
        self._modulo = int(modulo)
        self._max_node = int(max_node)

    def max_node_id(self):
        return self._max_node

    def to_host(self):
        c = lambda t: t.cpu().numpy()
        return KmerIndex(c(self._hashes_to_index), c(self._n_kmers), c(self._nodes), c(self._kmers).view(np.uint64),
                         self._modulo, c(self._frequencies))


def make_index_torch(n_kmers, k=31, seed=1, modulo=None, device=0, plant=True):
    """make_index's data model (genome of 4 N + k uniform bases, the k-mers at positions 4 i under node i, 0.1 % of them
    duplicated under 1-2 further nodes, one k-mer planted 1500 times) generated AND built on the GPU — torch's generator,
    so the values differ from make_index's numpy stream; for index sizes whose host-side generation would take
    minutes (BASELINE configs[4]: 10^9 k-mers).  Returns (DeviceSyntheticIndex, genome ASCII uint8 tensor on the device)."""
    import torch
    from .engine import build_index_device
    N = int(n_kmers)
    dev = torch.device("cuda", device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    genome = torch.empty(4 * N + k, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for a in range(0, genome.numel(), step):
        b = min(genome.numel(), a + step)
        genome[a:b] = torch.randint(0, 4, (b - a,), generator=g, device=dev, dtype=torch.uint8)
    kmers = torch.zeros(N, dtype=torch.int64, device=dev)
    for j in range(k):
        kmers |= genome[j:j + (N - 1) * 4 + 1:4].to(torch.int64) << (2 * j)
    nodes = torch.arange(N, dtype=torch.int32, device=dev)
    if plant and N >= 8:
        n_dup = max(1, N // 1000)
        dup = torch.randint(0, N, (n_dup,), generator=g, device=dev)
        reps = torch.randint(1, 3, (n_dup,), generator=g, device=dev)
        dup_k = torch.repeat_interleave(kmers[dup], reps)
        dup_n = torch.randint(0, N, (dup_k.numel(),), generator=g, device=dev, dtype=torch.int32)
        hot = kmers[int(torch.randint(0, N, (1,), generator=g, device=dev).item())]
        hot_k = hot.repeat(1500)
        hot_n = torch.randint(0, N, (1500,), generator=g, device=dev, dtype=torch.int32)
        kmers = torch.cat([kmers, dup_k, hot_k])
        nodes = torch.cat([nodes, dup_n, hot_n])
    if modulo is None:
        modulo = next_prime(2 * N)
    h2i, nk, ko, no, fo = build_index_device(kmers, nodes, modulo, device=device)
    del kmers, nodes
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    for a in range(0, genome.numel(), step):          # codes -> ASCII in place
        b = min(genome.numel(), a + step)
        genome[a:b] = lut[genome[a:b].long()]
    torch.cuda.synchronize(dev)
    return DeviceSyntheticIndex(h2i, nk, ko, no, fo, modulo, N - 1), genome
