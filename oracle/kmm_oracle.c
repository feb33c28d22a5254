/*
 * kmm_oracle.c — CPU restatement of kmer_mapper's k-mer extraction + index lookup hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker / the timed CPU
 * baseline.  The product path (kmer_mapper_amd/) never links, loads or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - lookup half  (oracle_map_kmers, oracle_in_index): pinned by the two known-answer vectors the reference's own
 *     tests hold (tests/test_gpucounter.py:41-48 and tests/test_mapping.py:33-40; data in
 *     tests/golden/reference_vectors.json).  The other vectors of that file (hash collision, multi-node k-mer,
 *     frequency filter) are transcriptions of numbers in SURVEY.md section 8c prose and are NOT pins: they are
 *     regression vectors; for those cases the restatement rests on its line-by-line correspondence with
 *     mapper.pyx:53-69.
 *   - extraction half (oracle_extract_kmers): the arithmetic lives in bionumpy, which is not in
 *     the reference tree nor installed here -> PARITY UNPINNED at that boundary; it follows the
 *     call shape of kmer_mapper/util.py:71-75, the N->A rule of command_line_interface.py:41 and
 *     the bit-order identity of tests/test_hashing.py:11-26, and is checked against hand-derived
 *     known answers.
 *
 *   - index construction (oracle_build_index): graph_kmer_index is un-vendored -> construction UPSTREAM-UNVERIFIED;
 *     pinned by the one index the reference's tests build (tests/test_mapping.py:33-40, modulo 21) and by the
 *     invariants mapper.pyx:53-69 reads.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference root).
 */
#define _POSIX_C_SOURCE 200809L /* pthread_barrier_t under -std=c11 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Lookup: kmer_mapper/mapper.pyx:19-72 (hot loop :53-69).
 *   kmerhash = kmers[i] % modulo                                    (:54)
 *   n_local_hits = n_kmers[kmerhash]; index_position = hashes_to_index[kmerhash]   (:55-56)
 *   for j in range(n_local_hits):                                   (:58)
 *       if index_kmers[l] != kmers[i]: skip                         (:60-62)
 *       if index_frequencies[l] > max_index_lookup_frequency: skip  (:64-66)
 *       node_counts[nodes[l]] += 1                                  (:68)
 * node_counts is uint32 and wraps modulo 2^32 (:37).  The reference allocates a fresh zeroed
 * vector per call; here the caller passes the vector and we ACCUMULATE into it (the caller zeroes
 * it for per-call semantics) so chunked runs can sum exactly like
 * command_line_interface.py:124-130 does.
 * ------------------------------------------------------------------------------------------- */
void oracle_map_kmers(const int32_t *hashes_to_index, const int32_t *n_kmers, uint64_t modulo,
                      const uint64_t *index_kmers, const int32_t *nodes,
                      const uint16_t *frequencies, const uint64_t *kmers, int64_t n,
                      int max_index_lookup_frequency, uint32_t *node_counts)
{
    for (int64_t i = 0; i < n; ++i) {
        uint64_t q = kmers[i];
        uint64_t h = q % modulo;
        int n_local_hits = n_kmers[h];
        int l = hashes_to_index[h];
        for (int j = 0; j < n_local_hits; ++j, ++l) {
            if (index_kmers[l] != q)
                continue;
            if ((int)frequencies[l] > max_index_lookup_frequency)
                continue;
            node_counts[nodes[l]] += 1u;
        }
    }
}

/* Membership mask: kmer_mapper/mapper.pyx:81-130 (loop :112-127): first match wins, no frequency
 * filter (the max_index_lookup_frequency argument is accepted but unused by the reference). */
void oracle_in_index(const int32_t *hashes_to_index, const int32_t *n_kmers, uint64_t modulo,
                     const uint64_t *index_kmers, const uint64_t *kmers, int64_t n, uint8_t *out)
{
    for (int64_t i = 0; i < n; ++i) {
        uint64_t q = kmers[i];
        uint64_t h = q % modulo;
        int n_local_hits = n_kmers[h];
        int64_t l = hashes_to_index[h];
        uint8_t hit = 0;
        for (int j = 0; j < n_local_hits; ++j, ++l) {
            if (index_kmers[l] != q)
                continue;
            hit = 1;
            break;
        }
        out[i] = hit;
    }
}

/* ---------------------------------------------------------------------------------------------
 * Extraction: kmer_mapper/util.py:71-75
 *   bnp.sequence.get_kmers(bnp.as_encoded_array(seq, bnp.DNAEncoding), k).ravel().raw().astype(uint64)
 * Semantics restated: each base -> 2-bit code through `lut` (default A,C,G,T ->
 * 0,1,2,3, case-insensitive; N -> code of A per command_line_interface.py:41); for every read r
 * and offset p in [0, len_r - k]: kmer = sum_{j<k} code[p+j] << (2j)  (first base in the lowest
 * two bits — the only packing under which tests/test_hashing.py:13-26 holds); windows never span
 * reads; output flattened in (read, offset) order.  A byte whose lut entry is 0xFF is not a
 * nucleotide: the reference raises, we return -(position+1) of the first such byte.
 * Returns the number of k-mers written (out may be NULL to only count / validate).
 * ------------------------------------------------------------------------------------------- */
void oracle_default_lut(uint8_t lut[256])
{
    memset(lut, 0xFF, 256);
    lut['A'] = lut['a'] = 0;
    lut['C'] = lut['c'] = 1;
    lut['G'] = lut['g'] = 2;
    lut['T'] = lut['t'] = 3;
    lut['N'] = lut['n'] = 0; /* command_line_interface.py:41: chunk_sequence[== "N"] = "A" */
}

int64_t oracle_extract_kmers(const uint8_t *bases, const int64_t *read_offsets, int64_t n_reads,
                             int k, const uint8_t *lut, uint64_t *out)
{
    uint8_t deflut[256];
    if (!lut) {
        oracle_default_lut(deflut);
        lut = deflut;
    }
    const uint64_t mask = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    int64_t n_out = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        int64_t b = read_offsets[r], e = read_offsets[r + 1];
        for (int64_t p = b; p < e; ++p)
            if (lut[bases[p]] == 0xFF)
                return -(p + 1);
        if (e - b < k)
            continue;
        uint64_t w = 0;
        for (int64_t p = b; p < e; ++p) {
            /* rolling window: drop the oldest base (lowest bits), insert the new one on top */
            w = (w >> 2) | ((uint64_t)lut[bases[p]] << (2 * (k - 1)));
            if (p - b >= k - 1) {
                if (out)
                    out[n_out] = w & mask;
                ++n_out;
            }
        }
    }
    return n_out;
}

/* Reverse complement of a packed k-mer under the A,C,G,T=0,1,2,3 / first-base-lowest layout:
 * complement = 3 - code = bitwise NOT of the 2-bit group; reversing the 2-bit groups of the word
 * and shifting right by 64-2k realigns the k-mer (the operation the reference's `-r` mode
 * delegates to cucounter, gpu_counter.py:23-24). */
uint64_t oracle_revcomp(uint64_t x, int k)
{
    x = ~x;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * k);
}

/* ---------------------------------------------------------------------------------------------
 * Whole map phase on the CPU, the shape of command_line_interface.py:32-56 + :124-130:
 * reads are cut into chunks, each worker thread extracts the k-mers of a chunk (util.py:71-75)
 * and looks them up (mapper.pyx:53-69) into a PRIVATE uint32 vector; the private vectors are
 * summed at the end (additative_shared_array_map_reduce).  `also_revcomp` additionally looks up
 * the reverse complement of every k-mer (`-r`, command_line_interface.py:74).
 * Used as bench.py's cpu_baseline (kind "port") and as the end-to-end checker in tests.
 * Returns the number of k-mers mapped, or a negative value on an invalid byte / alloc failure.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const int32_t *h2i, *nk, *nodes;
    const uint64_t *ikmers;
    const uint16_t *freqs;
    uint64_t modulo;
    const uint8_t *bases;
    const int64_t *offs;
    int64_t n_reads;
    int k, max_freq, also_rc;
    const uint8_t *lut;
    int64_t max_node_id;
    int tid, n_threads;
    int64_t chunk_reads;
    uint32_t *counts; /* private */
    int64_t n_mapped;
    /* parallel sum of the private vectors (every worker adds one slice of all of them) */
    pthread_barrier_t *barrier;
    uint32_t **all_counts;
} oracle_worker_t;

static void oracle_sum_slice(oracle_worker_t *w)
{
    const int64_t n = w->max_node_id + 1;
    const int64_t lo = n * w->tid / w->n_threads, hi = n * (w->tid + 1) / w->n_threads;
    uint32_t *dst = w->all_counts[0];
    for (int t = 1; t < w->n_threads; ++t) {
        const uint32_t *src = w->all_counts[t];
        for (int64_t i = lo; i < hi; ++i)
            dst[i] += src[i];
    }
}

static void *oracle_worker(void *arg)
{
    oracle_worker_t *w = (oracle_worker_t *)arg;
    const int64_t CH = w->chunk_reads;
    uint64_t *buf = NULL;
    int64_t cap = 0;
    w->n_mapped = 0;
    /* static round-robin assignment of chunks to workers */
    for (int64_t c0 = (int64_t)w->tid * CH; c0 < w->n_reads; c0 += (int64_t)w->n_threads * CH) {
        int64_t c1 = c0 + CH < w->n_reads ? c0 + CH : w->n_reads;
        int64_t nb = w->offs[c1] - w->offs[c0];
        if (nb > cap) {
            free(buf);
            cap = nb;
            buf = (uint64_t *)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(uint64_t));
            if (!buf) {
                w->n_mapped = INT64_MIN;
                break;
            }
        }
        int64_t nkm = oracle_extract_kmers(w->bases, w->offs + c0, c1 - c0, w->k, w->lut, buf);
        if (nkm < 0) {
            w->n_mapped = nkm;
            break;
        }
        oracle_map_kmers(w->h2i, w->nk, w->modulo, w->ikmers, w->nodes, w->freqs, buf, nkm,
                         w->max_freq, w->counts);
        if (w->also_rc) {
            for (int64_t i = 0; i < nkm; ++i)
                buf[i] = oracle_revcomp(buf[i], w->k);
            oracle_map_kmers(w->h2i, w->nk, w->modulo, w->ikmers, w->nodes, w->freqs, buf, nkm,
                             w->max_freq, w->counts);
        }
        w->n_mapped += nkm;
    }
    free(buf);
    pthread_barrier_wait(w->barrier); /* every private vector is complete */
    oracle_sum_slice(w);
    return NULL;
}

int64_t oracle_map_reads(const int32_t *hashes_to_index, const int32_t *n_kmers, uint64_t modulo,
                         const uint64_t *index_kmers, const int32_t *nodes,
                         const uint16_t *frequencies, int64_t max_node_id, const uint8_t *bases,
                         const int64_t *read_offsets, int64_t n_reads, int k,
                         int max_index_lookup_frequency, int also_revcomp, const uint8_t *lut,
                         int n_threads, int64_t chunk_reads, uint32_t *node_counts)
{
    uint8_t deflut[256];
    if (!lut) {
        oracle_default_lut(deflut);
        lut = deflut;
    }
    if (n_threads < 1)
        n_threads = 1;
    if (chunk_reads < 1)
        chunk_reads = 16384;
    oracle_worker_t *ws = (oracle_worker_t *)calloc((size_t)n_threads, sizeof(*ws));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(*th));
    uint32_t **all = (uint32_t **)calloc((size_t)n_threads, sizeof(*all));
    pthread_barrier_t barrier;
    if (!ws || !th || !all)
        return INT64_MIN;
    pthread_barrier_init(&barrier, NULL, (unsigned)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        oracle_worker_t *w = &ws[t];
        w->h2i = hashes_to_index; w->nk = n_kmers; w->nodes = nodes; w->ikmers = index_kmers;
        w->freqs = frequencies; w->modulo = modulo; w->bases = bases; w->offs = read_offsets;
        w->n_reads = n_reads; w->k = k; w->max_freq = max_index_lookup_frequency;
        w->also_rc = also_revcomp; w->lut = lut; w->max_node_id = max_node_id;
        w->tid = t; w->n_threads = n_threads; w->chunk_reads = chunk_reads;
        /* thread 0 accumulates straight into the caller's vector */
        w->counts = (t == 0) ? node_counts
                             : (uint32_t *)calloc((size_t)max_node_id + 1, sizeof(uint32_t));
        if (!w->counts)
            return INT64_MIN;
        all[t] = w->counts;
        w->all_counts = all;
        w->barrier = &barrier;
    }
    for (int t = 1; t < n_threads; ++t)
        pthread_create(&th[t], NULL, oracle_worker, &ws[t]);
    oracle_worker(&ws[0]);
    int64_t total = ws[0].n_mapped;
    int64_t err = total < 0 ? total : 0;
    for (int t = 1; t < n_threads; ++t) {
        pthread_join(th[t], NULL);
        if (ws[t].n_mapped < 0) {
            if (!err) err = ws[t].n_mapped;
        } else {
            total += ws[t].n_mapped;
        }
    }
    for (int t = 1; t < n_threads; ++t) /* only now: every worker reads every private vector while it sums its slice */
        free(ws[t].counts);
    pthread_barrier_destroy(&barrier);
    free(all);
    free(ws);
    free(th);
    return err ? err : total;
}
/* ---------------------------------------------------------------------------------------------
 * Index construction: what `FlatKmers(hashes, nodes, ref_offsets)` -> `KmerIndex.from_flat_kmers(flat, modulo=M)` ->
 * `convert_to_int32()` produces in tests/test_mapping.py:33-38 — graph_kmer_index is an un-vendored dependency, so
 * the construction is [UPSTREAM-UNVERIFIED]; what IS fixed by the reference tree are the invariants its lookup loop
 * relies on (kmer_mapper/mapper.pyx:53-69), and this function establishes exactly those:
 *   - entries grouped by kmer % modulo, groups in ascending hash order, the entries of one group in their input
 *     order (a stable sort by hash);
 *   - hashes_to_index[h] = first entry of group h, n_kmers[h] = its length (0 / 0 for an empty bucket);   (:55-56)
 *   - frequencies[l] = number of index entries whose k-mer equals kmers[l], clipped to 65535                (:64)
 * Plain loops: a counting sort over the modulo, then per bucket the equal-key counts (quadratic in the bucket
 * length: oracle sizes only).  Returns 0, or -1 when more than INT32_MAX entries / a node outside int32 is given.
 * ------------------------------------------------------------------------------------------- */
static int oracle_cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

int oracle_build_index(const uint64_t *kmers, const int64_t *nodes, int64_t n, uint64_t modulo,
                       int32_t *hashes_to_index, int32_t *n_kmers, uint64_t *kmers_out, int32_t *nodes_out,
                       uint16_t *frequencies_out)
{
    if (n > INT32_MAX || modulo == 0)
        return -1;
    memset(n_kmers, 0, (size_t)modulo * sizeof(int32_t));
    memset(hashes_to_index, 0, (size_t)modulo * sizeof(int32_t));
    for (int64_t i = 0; i < n; ++i) {
        if (nodes[i] < INT32_MIN || nodes[i] > INT32_MAX)
            return -1;
        n_kmers[kmers[i] % modulo] += 1;
    }
    int32_t *cursor = (int32_t *)malloc((size_t)modulo * sizeof(int32_t));
    if (!cursor)
        return -2;
    int32_t run = 0;
    for (uint64_t h = 0; h < modulo; ++h) {
        cursor[h] = run;
        if (n_kmers[h])
            hashes_to_index[h] = run;
        run += n_kmers[h];
    }
    for (int64_t i = 0; i < n; ++i) { /* input order inside a bucket: stable */
        const int32_t l = cursor[kmers[i] % modulo]++;
        kmers_out[l] = kmers[i];
        nodes_out[l] = (int32_t)nodes[i];
    }
    free(cursor);
    uint64_t *tmp = NULL;
    int32_t tmp_cap = 0;
    for (uint64_t h = 0; h < modulo; ++h) {
        const int32_t s = hashes_to_index[h], c = n_kmers[h];
        if (c <= 64) {
            for (int32_t a = 0; a < c; ++a) {
                uint32_t f = 0;
                for (int32_t b = 0; b < c; ++b) /* equal k-mers share their hash: they all lie in this bucket */
                    f += kmers_out[s + b] == kmers_out[s + a];
                frequencies_out[s + a] = (uint16_t)f;
            }
            continue;
        }
        /* a long bucket (one k-mer under thousands of nodes): sorted copy, the count of a key = its run's length */
        if (c > tmp_cap) {
            free(tmp);
            tmp = (uint64_t *)malloc((size_t)c * sizeof(uint64_t));
            if (!tmp)
                return -2;
            tmp_cap = c;
        }
        memcpy(tmp, kmers_out + s, (size_t)c * sizeof(uint64_t));
        qsort(tmp, (size_t)c, sizeof(uint64_t), oracle_cmp_u64);
        for (int32_t a = 0; a < c; ++a) {
            const uint64_t key = kmers_out[s + a];
            int32_t lo = 0, hi = c; /* first position with tmp[pos] >= key */
            while (lo < hi) {
                const int32_t mid = lo + (hi - lo) / 2;
                if (tmp[mid] < key) lo = mid + 1; else hi = mid;
            }
            int32_t lo2 = lo, hi2 = c; /* first position with tmp[pos] > key */
            while (lo2 < hi2) {
                const int32_t mid = lo2 + (hi2 - lo2) / 2;
                if (tmp[mid] <= key) lo2 = mid + 1; else hi2 = mid;
            }
            const int32_t f = lo2 - lo;
            frequencies_out[s + a] = (uint16_t)(f > 65535 ? 65535 : f);
        }
    }
    free(tmp);
    return 0;
}

#ifdef __cplusplus
}
#endif
