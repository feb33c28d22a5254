/* kmm_demo.c — a plain-C client of libkmm.so: proves the boundary needs nothing but <stdint.h>.
 * Builds the reference's own known-answer index (tests/test_gpucounter.py:41-48: k-mers 1,2,3 ->
 * nodes 10,11,12, hash-table size 2003), maps the query 1,1,1,2,3,1,3 and three short reads.
 *   gcc -std=c11 -Iinclude tools/kmm_demo.c -o /tmp/kmm_demo -Lkmer_mapper_amd -lkmm -Wl,-rpath,$PWD/kmer_mapper_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmm.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != KMM_OK) {                                               \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, kmm_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(void)
{
    enum { M = 2003, N = 3, MAX_NODE = 15 };
    static int32_t h2i[M], nk[M];
    uint64_t kmers[N] = {1, 2, 3};
    int32_t nodes[N] = {10, 11, 12};
    uint16_t freqs[N] = {1, 1, 1};
    for (int i = 0; i < N; ++i) { /* entries already sorted by kmer % M */
        h2i[kmers[i] % M] = i;
        nk[kmers[i] % M] = 1;
    }
    printf("%s\n", kmm_version());
    kmm_index_t *idx = NULL;
    CHECK(kmm_index_create(h2i, nk, M, kmers, nodes, freqs, N, MAX_NODE, 0, &idx));

    uint64_t query[7] = {1, 1, 1, 2, 3, 1, 3};
    uint32_t counts[MAX_NODE + 1];
    CHECK(kmm_map_kmers(idx, query, 7, 1000, 0, 31));
    CHECK(kmm_get_node_counts(idx, counts));
    printf("map_kmers: node 10,11,12 = %u,%u,%u (expected 4,1,2)\n", counts[10], counts[11], counts[12]);
    int ok = counts[10] == 4 && counts[11] == 1 && counts[12] == 2;

    /* k = 2: "CA" = 1 + 0*4 = 1, "GA" = 2, "TA" = 3 -> the same three k-mers from read bytes */
    const char *reads = "CACACAGATACATA"; /* two reads: CACACAGATA | CATA */
    int64_t offsets[3] = {0, 10, 14};
    CHECK(kmm_reset_counts(idx));
    CHECK(kmm_map_reads(idx, (const uint8_t *)reads, offsets, 2, 2, 1000, 0, NULL));
    CHECK(kmm_get_node_counts(idx, counts));
    /* windows: CA AC CA AC CA AG GA AT TA | CA AT TA -> CA x4, GA x1, TA x2 */
    printf("map_reads: node 10,11,12 = %u,%u,%u (expected 4,1,2)\n", counts[10], counts[11], counts[12]);
    ok = ok && counts[10] == 4 && counts[11] == 1 && counts[12] == 2;

    const char *fastq = "@r1\nCACACAGATA\n+\nIIIIIIIIII\n@r2\nCATA\n+\nIIII\n@r3\nCA";
    int64_t consumed = 0, n_rec = 0;
    CHECK(kmm_reset_counts(idx));
    CHECK(kmm_map_records(idx, (const uint8_t *)fastq, (int64_t)strlen(fastq), KMM_FORMAT_FASTQ, 2, 1000, 0,
                          NULL, &consumed, &n_rec));
    CHECK(kmm_get_node_counts(idx, counts));
    printf("map_records: consumed %lld bytes, %lld records; node 10,11,12 = %u,%u,%u (expected 4,1,2)\n",
           (long long)consumed, (long long)n_rec, counts[10], counts[11], counts[12]);
    ok = ok && n_rec == 2 && counts[10] == 4 && counts[11] == 1 && counts[12] == 2;
    kmm_index_destroy(idx);
    printf(ok ? "OK\n" : "MISMATCH\n");
    return ok ? 0 : 2;
}
