/*
 * kmm.h — C ABI of libkmm.so: the MI355X (gfx950) implementation of kmer_mapper's
 * k-mer extraction + Kmer-Index lookup + per-node count accumulation hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; nothing here knows about
 * numpy, torch or Python.  Every entry point names the reference interface it replaces
 * (paths relative to the ivargr/kmer_mapper tree).  INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - All functions return KMM_OK (0) or a negative KMM_ERR_* code and never throw across the
 *     ABI; kmm_last_error() returns a thread-local human-readable message for the last failure.
 *   - Input pointers are BORROWED for the duration of the call.  Every data pointer may be a
 *     host pointer or a device (HBM) pointer on the handle's device; the library detects which
 *     (hipPointerGetAttributes).  Host inputs are staged to HBM by the call (the host buffer is free
 *     again when the call returns); device inputs are used in place by kernels that run after the
 *     call returns: they must already be complete (produced on another stream -> synchronise that
 *     stream first) and must stay valid and unmodified until the next synchronising call.
 *   - One handle = one device + one HIP stream + one uint32 node-count vector in HBM.
 *     map calls are asynchronous on the handle's stream and ACCUMULATE into the count vector
 *     (the reference sums per-chunk vectors, command_line_interface.py:124-130);
 *     kmm_get_node_counts / kmm_synchronize are the synchronisation points.  Calls on one
 *     handle must be serialised by the caller; different handles are independent.
 *   - Errors found by the kernels of a map call (a byte that is not a nucleotide, a malformed record,
 *     decreasing read offsets) are reported by the next synchronising call and then STAY on the handle:
 *     the chunk's valid windows are already counted by then (the reference raises before counting anything
 *     of that chunk), so every later synchronising call fails with the same code until kmm_reset_counts
 *     clears the counts and the error together.  A caller-owned buffer (kmm_bind_counts) must be zeroed by
 *     the caller as well.
 *   - Counts are uint32 and wrap modulo 2^32 exactly like the reference (mapper.pyx:37,68).
 */
#ifndef KMM_H
#define KMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMM_OK 0
#define KMM_ERR_INVALID_ARG (-1)   /* NULL pointer, k out of range, negative size ...            */
#define KMM_ERR_HIP (-2)           /* a HIP runtime call failed (message has file:line)          */
#define KMM_ERR_INDEX (-3)         /* index arrays inconsistent (bucket out of range, bad node)  */
#define KMM_ERR_INVALID_BASE (-4)  /* a read byte is not a nucleotide under the lookup table     */
#define KMM_ERR_NOMEM (-5)
#define KMM_ERR_MALFORMED (-6)     /* raw FASTA/FASTQ chunk does not have the expected line structure */
#define KMM_ERR_INTERNAL (-7)      /* self-check of the radix path failed (k-mers emitted by pass 1 != gathered by
                                      pass 2 != probed by pass 3): counts are NOT returned; sticky until
                                      kmm_reset_counts */

#define KMM_MAX_K 31               /* a k-mer is packed 2 bits/base into a uint64; bionumpy's
                                      get_kmers (util.py:72) is used with k <= 31                */

typedef struct kmm_index kmm_index_t;

/* Library / device probes. */
const char *kmm_version(void);
const char *kmm_last_error(void);
int kmm_device_count(int *n_devices);
/* "0000:5a:00.0" of HIP device `device` (hipDeviceGetPCIBusId): /sys/bus/pci/devices/<id>/numa_node and local_cpulist
 * name the host cores and memory next to that GPU — where a rank's packing threads and page-locked buffers belong
 * (kmer_mapper_amd/distributed.py bind_to_gpu_numa_node; the reference leaves its workers unbound). */
int kmm_device_pci_bus_id(int device, char *out, int out_bytes);

/*
 * kmm_index_create — replaces the typed-memoryview binding of the five index arrays at
 * mapper.pyx:22-29 and the cucounter table construction at gpu_counter.py:13-16.
 * Arrays are the attributes of graph_kmer_index.KmerIndex after convert_to_int32()
 * (util.py:60-62): hashes_to_index int32[modulo], n_kmers int32[modulo], kmers uint64[n_entries],
 * nodes int32[n_entries], frequencies uint16[n_entries].  They are copied to HBM and repacked on the
 * GPU (16-byte bucket records with the single entry of a bucket stored inline, 16-byte {kmer,node,freq}
 * entries, and for small and medium indexes an L2-resident Bloom filter / occupancy bitmap that rejects
 * most absent k-mers without an HBM access; large indexes get 32-byte buckets).  Unlike the
 * reference (no bounds checks, mapper.pyx:17) the arrays are validated: every non-empty bucket
 * must lie inside [0, n_entries) and every node inside [0, max_node_id], else KMM_ERR_INDEX.
 * Environment knobs for experiments, read here: KMM_OCC_MAX_BYTES (largest occupancy bitmap that is still
 * built; 0 forces the wide 32-byte bucket layout), KMM_WIDE_BUCKETS=0 (16-byte buckets without filter),
 * KMM_BLOOM_BYTES (Bloom filter size; 0 = per-bucket bitmap), KMM_BLOOM_MAX_ENTRIES, KMM_OCC_SHIFT.
 */
int kmm_index_create(const int32_t *hashes_to_index, const int32_t *n_kmers, uint64_t modulo,
                     const uint64_t *kmers, const int32_t *nodes, const uint16_t *frequencies,
                     int64_t n_entries, int64_t max_node_id, int device, kmm_index_t **out);
void kmm_index_destroy(kmm_index_t *idx);

/*
 * Node-count vector management (mapper.pyx:37 allocates it per call; gpu_counter.py keeps it
 * inside the counter).  kmm_bind_counts makes the handle accumulate into a caller-owned device
 * buffer of max_node_id+1 uint32 (e.g. a torch tensor that is then reduced with RCCL); the
 * buffer is NOT zeroed by the bind.  kmm_counts_device_ptr returns the current device buffer.
 */
int kmm_reset_counts(kmm_index_t *idx);
int kmm_bind_counts(kmm_index_t *idx, uint32_t *device_counts);
int kmm_counts_device_ptr(kmm_index_t *idx, uint32_t **out);
/* Copies the max_node_id+1 counts to `out` after draining the stream.  out: host memory of any kind (or device memory).  Into
 * ORDINARY (pageable) memory a vector of 64 MiB and more travels through the handle's page-locked staging ring, the
 * "host_pack_threads" threads copying the slots out: the link's rate without a page-locked destination (which costs ~50 ms
 * per GB to make — as much as a whole map phase; configs[2]'s 400 MB vector: ~10 ms instead of 30-40). */
int kmm_get_node_counts(kmm_index_t *idx, uint32_t *out);
int kmm_synchronize(kmm_index_t *idx);

/*
 * The one exchange step of the path: the sum of the per-GPU node-count vectors over RCCL / xGMI.  Replaces the
 * additive reduce of per-chunk vectors, command_line_interface.py:124-130 (shared_memory_wrapper's
 * additative_shared_array_map_reduce).  uint32 addition wraps modulo 2^32 like mapper.pyx:37,68, so the result
 * is bit-exact whatever the reduction order.  RCCL is loaded at first use (dlopen), not linked.
 *
 * One process, several GPUs (one handle per GPU, same index on each):
 *   kmm_reduce_counts(handles, n_gpus, root)  root >= 0: handles[root]'s count vector becomes the sum (the others are
 *                                             unspecified afterwards); root = -1: every vector becomes the sum.
 * One process per GPU (what bench.py and the CLI use under torchrun):
 *   kmm_comm_get_unique_id(id)                on rank 0; the caller hands the 128 bytes to the other ranks by any
 *                                             means (a file, MPI, torch.distributed's store, ...)
 *   kmm_comm_init_rank(idx, id, n_ranks, rank)  collective: every rank joins with its own handle
 *   kmm_comm_reduce_counts(idx, root)         collective, in place on the handle's (library-owned or bound) count
 *                                             vector; root as above
 *   kmm_comm_destroy(idx)                     (kmm_index_destroy does it too)
 * All of them synchronise like kmm_get_node_counts (pending per-entry hits of the radix path are flushed first,
 * deferred device-side errors are reported).
 */
#define KMM_COMM_ID_BYTES 128
int kmm_reduce_counts(kmm_index_t **per_gpu, int n_gpus, int root);
int kmm_comm_get_unique_id(uint8_t id[KMM_COMM_ID_BYTES]);
int kmm_comm_init_rank(kmm_index_t *idx, const uint8_t id[KMM_COMM_ID_BYTES], int n_ranks, int rank);
int kmm_comm_reduce_counts(kmm_index_t *idx, int root);
int kmm_comm_destroy(kmm_index_t *idx);

/*
 * kmm_map_kmers — replaces map_kmers_to_graph_index (mapper.pyx:19-72, loop :53-69) and
 * GpuCounter.count (gpu_counter.py:23-24): for each of the n packed k-mers q,
 * h = q % modulo, scan bucket h, and for every entry with kmer == q and
 * frequency <= max_index_lookup_frequency add 1 to counts[node].  If also_revcomp != 0 the
 * reverse complement of q (k bases) is looked up as well (`-r`,
 * command_line_interface.py:74,180-182).  k is only used for also_revcomp.
 */
int kmm_map_kmers(kmm_index_t *idx, const uint64_t *kmers, int64_t n,
                  int max_index_lookup_frequency, int also_revcomp, int k);

/*
 * kmm_map_reads — the fused path: replaces map_cpu's three steps
 * (command_line_interface.py:41 N->A, :42 get_kmer_hashes_from_chunk_sequence = util.py:71-75,
 * :51 map_kmers_to_graph_index) without materialising the k-mer array.
 * bases: the chunk's flat ASCII read bytes; read_offsets: int64[n_reads+1], read r is
 * bases[read_offsets[r] : read_offsets[r+1]] (read_offsets[0] must be 0; the offsets must be
 * non-decreasing, which is checked on the GPU and reported by the next synchronising call as
 * KMM_ERR_INVALID_ARG).  Every window of k
 * bases inside one read is packed first-base-lowest, 2 bits/base through `lut`
 * (uint8[256]: 0..3 = code, 0xFF = not a nucleotide; NULL = A,C,G,T->0,1,2,3 case-insensitive
 * with N->A) and looked up as in kmm_map_kmers.  A byte with lut 0xFF makes the NEXT
 * synchronising call fail with KMM_ERR_INVALID_BASE (the reference's encoder raises).
 */
int kmm_map_reads(kmm_index_t *idx, const uint8_t *bases, const int64_t *read_offsets,
                  int64_t n_reads, int k, int max_index_lookup_frequency, int also_revcomp,
                  const uint8_t *lut);
/* Same for n_reads reads of identical length read_len stored back to back (no offsets array;
 * the 150 bp short-read case of the reference's Readme.md:11-13). */
int kmm_map_reads_uniform(kmm_index_t *idx, const uint8_t *bases, int64_t n_reads,
                          int64_t read_len, int k, int max_index_lookup_frequency,
                          int also_revcomp, const uint8_t *lut);

/*
 * kmm_map_records — maps a RAW chunk of a FASTQ (format = 4 lines per record) or two-line FASTA
 * (format = 2) file: replaces `bnp.open(reads).read_chunks(...)` + the per-chunk map
 * (command_line_interface.py:102-103,109-111 and :32-56) — record parsing happens on the GPU, the host
 * only reads (and, for .gz, inflates) bytes.  `raw` must start at the first byte of a record.  The
 * library finds the end of the last COMPLETE record inside the chunk, maps every base of the sequence
 * lines up to there exactly like kmm_map_reads, and returns in *consumed how many bytes it used (the
 * caller prepends the remaining raw[consumed:] to the next chunk; at end of file the last line must end
 * with a newline) and in *n_records the number of reads mapped.  '\r' before '\n' is tolerated.
 * The call returns once the chunk is staged and scanned (so *consumed is valid and the host buffer is
 * free); the mapping kernels run asynchronously like every other map call.  Chunks of any size: beyond 2^30
 * bytes the library scans the chunk piece by piece (each piece starts at the end of the previous one's last
 * complete record).  A chunk large enough for the radix path (FASTQ: 2 x kmm_get_param("radix_min_units") bytes) is
 * first compacted on the device — only the bytes of its sequence lines survive, as 2-bit codes — and mapped as ONE
 * batch of flat reads; kmer_mapper map accumulates file chunks to such sizes.
 * A line that should start a record ('@' / '>') or the FASTQ '+' line but does not (e.g. multi-line
 * FASTA) makes the next synchronising call fail with KMM_ERR_MALFORMED.
 */
#define KMM_FORMAT_FASTA2 2
#define KMM_FORMAT_FASTQ 4
/* Multi-line FASTA (sequences wrapped over several lines; `bnp.open` reads those too): the chunk is unwrapped into
 * two-line FASTA on the GPU first.  A record is only known to be complete once the NEXT header line has been seen, so
 * *consumed stops at the start of the chunk's last header line — unless the caller ORs KMM_FORMAT_LAST_CHUNK into
 * `format` (the chunk ends the file: everything is consumed). */
#define KMM_FORMAT_FASTA 1
#define KMM_FORMAT_LAST_CHUNK 0x100
int kmm_map_records(kmm_index_t *idx, const uint8_t *raw, int64_t n_bytes, int format, int k,
                    int max_index_lookup_frequency, int also_revcomp, const uint8_t *lut,
                    int64_t *consumed, int64_t *n_records);

/*
 * kmm_map_bgzf — reads from a BGZF-compressed FASTQ / two-line FASTA file (what bgzip / htslib write: a chain of independent
 * gzip members of at most 64 KiB of data, each with its size in the header), INFLATED ON THE GPU: replaces
 * `bnp.open("reads.fq.gz").read_chunks(...)` + the per-chunk map (command_line_interface.py:102-111; ".fa.gz, or fq.gz",
 * Readme.md:11; the igzip reader of util.py:78-101) without a host-side inflater — the compressed bytes cross PCIe (about a
 * quarter of the raw ones), one GPU thread inflates one member (thousands per chunk; stored, fixed and dynamic blocks;
 * every member's ISIZE and CRC32 checked on the device), and the raw records go to the device-side parser as in
 * kmm_map_records.  The bytes reach HBM through the handle's page-locked staging ring (eight slots of 16 MiB, filled by the
 * "host_pack_threads" threads, emptied by a copy engine); the member chain is walked in the caller's bytes meanwhile.
 * comp: n_comp compressed bytes in HOST memory (a file mapping will do) that start at a member boundary;
 * the call uses the whole members inside (at most 3.5 GiB of inflated bytes) and returns in *consumed_comp how many
 * compressed bytes that was — the caller continues there.  Records do not end where members end: the handle keeps the
 * inflated bytes behind the call's last complete record and puts them in front of the next call's (a STREAM per handle:
 * calls in file order; OR KMM_FORMAT_NEW_STREAM into `format` for the first chunk of a file, KMM_FORMAT_LAST_CHUNK for the
 * last: a final line without newline gets one, and bytes that then still form no record — or compressed bytes behind the
 * last whole member — are KMM_ERR_MALFORMED; a call that stops at its own size limit ignores the flag: the caller comes
 * back with the rest and the same flag).
 * format: KMM_FORMAT_FASTQ or KMM_FORMAT_FASTA2.  A corrupt member (header, Huffman code, distance, ISIZE, CRC32) makes the
 * call fail with KMM_ERR_MALFORMED before anything of the chunk is mapped.  A plain gzip file (no member sizes: `gzip`, not
 * `bgzip`) is refused the same way — inflate it on the host (libkmm_io).  *n_records: reads mapped by this call.
 */
#define KMM_FORMAT_NEW_STREAM 0x400
/* Optional, before a kmm_map_bgzf call whose chunk is followed in the caller's memory by more of the file (a file mapping):
 * comp_next = comp + n_comp of that call, n_next = how many more bytes the NEXT call will bring (the next call then passes
 * comp + *consumed_comp, n_comp - *consumed_comp + n_next).  The library stages and walks those bytes while this chunk's
 * members are being inflated — a chunk's 20 ms over PCIe then no longer stand in front of its 46 ms of kernel.  A hint that
 * the next call does not keep to costs nothing but the copy.  "bgzf_prestaged_calls" (read-only) counts the hints used. */
int kmm_map_bgzf_hint_next(kmm_index_t *idx, const uint8_t *comp_next, int64_t n_next);
int kmm_map_bgzf(kmm_index_t *idx, const uint8_t *comp, int64_t n_comp, int format, int k, int max_index_lookup_frequency,
                 int also_revcomp, const uint8_t *lut, int64_t *consumed_comp, int64_t *n_records);

/*
 * kmm_map_packed — reads the caller already holds as 2-BIT CODES (its own encoder, a .2bit-style store, the output of a
 * host-side packer): the same mapping as kmm_map_reads without the byte -> code step, and a quarter of the bytes over
 * PCIe.  This is the form the library's own host packer produces when kmm_map_reads* / kmm_map_records are handed host
 * memory ("host_pack_threads" below) — the work the reference spends its `-t` worker processes on
 * (bnp.as_encoded_array in util.py:71-72, command_line_interface.py:124-130,168).
 * codes: uint32 words, 16 codes per word, base p of the flat read stream in bits [2 (p & 15), 2 (p & 15) + 2) of word
 * p >> 4 (first base lowest; A,C,G,T = 0,1,2,3 — the packing of util.py:72-73), (n_bases + 15) / 16 words, host or device.
 * read_len > 0: n_reads reads of that length back to back (n_reads * read_len == n_bases), read_starts ignored.
 * read_len == 0: ragged reads; read_starts = bitset over the base positions, bit p & 31 of word p >> 5 set iff a read
 * starts at base p (n_bases / 32 + 1 words, host or device): no k-mer spans a set bit (util.py:72).
 * Served by the radix path at every batch size (KMM_ERR_INVALID_ARG when the index has none: "radix_available").
 */
int kmm_map_packed(kmm_index_t *idx, const uint32_t *codes, int64_t n_bases, int64_t n_reads, int64_t read_len,
                   const uint32_t *read_starts, int k, int max_index_lookup_frequency, int also_revcomp);

/*
 * kmm_host_alloc / kmm_host_free — page-locked host memory (hipHostMalloc) for the caller's read buffers: the
 * staging copy of a map call then runs at the PCIe link's rate (~50 GB/s) instead of the pageable path's.  The
 * reference keeps its chunks in POSIX shared memory (command_line_interface.py:110); this is the GPU counterpart.
 */
int kmm_host_alloc(size_t bytes, void **out);
int kmm_host_free(void *p);
/* Prepares the page-locked staging buffers the library needs for host-resident batches of raw_batch_bytes raw bytes
 * ("host_pack_threads": the packed stream + the read-start bitset) and shelves them for the next handle that asks.  A
 * page-locked allocation costs ~50 ms per GB: a caller that knows its batch size calls this from another thread while
 * the index is created (kmer_mapper map does), and the first map call finds the buffers ready.  Optional. */
int kmm_host_reserve(int64_t raw_batch_bytes);
/* The same for ONE page-locked buffer of `bytes` bytes, for callers that know what the next handle will ask for. */
int kmm_host_reserve_buffer(int64_t bytes);

/*
 * kmm_extract_kmers — replaces get_kmer_hashes_from_chunk_sequence (util.py:71-75) as an
 * operator: writes the packed k-mers of all reads, flattened in (read, offset) order, to `out`
 * (host or device, n_out = sum(max(len_r-k+1,0)) uint64).  Not used by the fused path.
 */
int kmm_extract_kmers(int device, const uint8_t *bases, const int64_t *read_offsets,
                      int64_t n_reads, int k, const uint8_t *lut, uint64_t *out, int64_t n_out);

/*
 * kmm_in_index — replaces in_graph_index / in_graph_index_no_memory_maps
 * (mapper.pyx:81-130,137-190): out[i] = 1 iff some entry of bucket kmers[i] % modulo equals
 * kmers[i] (no frequency filter).  out: uint8[n], host or device.
 */
int kmm_in_index(kmm_index_t *idx, const uint64_t *kmers, int64_t n, uint8_t *out);

/*
 * kmm_build_index — builds the Kmer Index arrays on the GPU from flat (k-mer, node) pairs: replaces
 * graph_kmer_index's KmerIndex.from_flat_kmers(flat_kmers, modulo) (reference call site
 * tests/test_mapping.py:36-38; gpu_counter.py:16 builds its table from the same pairs).  Entries are
 * ordered by kmer % modulo, ties by original position (a stable sort, reproducible bit for bit);
 * hashes_to_index[h] = first entry of bucket h (0 for empty buckets),
 * frequencies[l] = number of entries holding the same k-mer as entry l, clipped to 65535.
 * Outputs: hashes_to_index int32[modulo], n_kmers int32[modulo], kmers_out uint64[n], nodes_out int32[n],
 * frequencies_out uint16[n]; every pointer may be host or device memory.  n, modulo < 2^31.
 */
int kmm_build_index(int device, const uint64_t *kmers, const int32_t *nodes, int64_t n, uint64_t modulo,
                    int32_t *hashes_to_index, int32_t *n_kmers, uint64_t *kmers_out, int32_t *nodes_out,
                    uint16_t *frequencies_out);

/*
 * Measurement hooks (the reference only logs perf_counter deltas,
 * command_line_interface.py:67-78).  With timing on, every launch of a hot-path kernel is
 * bracketed by HIP events on the handle's stream; kmm_get_timing drains the stream and returns,
 * for one kernel id, the summed milliseconds and the number of launches since the last call for
 * that id, then clears them.
 */
#define KMM_KERNEL_MAP_READS 0    /* fused direct kernel (reads -> counts)                    */
#define KMM_KERNEL_MAP_KMERS 1    /* operator kernel (uint64 k-mers -> counts)                */
#define KMM_KERNEL_RX_P1 2        /* radix path, pass 1: reads -> blocks sorted by coarse hash range   */
#define KMM_KERNEL_RX_SCAN 3      /* radix path: directory column scan between pass 1 and pass 2       */
#define KMM_KERNEL_RX_P2 4        /* radix path, pass 2: items sorted by fine hash range               */
#define KMM_KERNEL_RX_P3 5        /* radix path, pass 3: probe of LDS-resident index slices + counting */
#define KMM_KERNEL_RX_FLUSH 6     /* radix path: per-entry hit counts -> node counts                   */
#define KMM_N_KERNELS 7
int kmm_set_timing(kmm_index_t *idx, int enabled);
/* Work done by the map calls of this handle since creation (or the last call with reset != 0):
 * k-mer lookups performed (windows, doubled with also_revcomp) and count increments (index hits that
 * passed the frequency filter) — the numbers the reference logs per chunk
 * (command_line_interface.py:53, util.py:74).  Synchronises like kmm_get_node_counts. */
int kmm_get_stats(kmm_index_t *idx, int reset, uint64_t *n_lookups, uint64_t *n_hits);
int kmm_get_timing(kmm_index_t *idx, int kernel_id, double *kernel_ms, int64_t *n_launches);

/*
 * kmm_get_kmer_counts — per-k-mer counting mode: replaces the table lookup of GpuCounter.get_node_counts
 * (gpu_counter.py:29-34, `counter[index_kmers]`) and the CounterKmerIndex branch of the CLI
 * (command_line_interface.py:46-49,133-138): out[l] = number of mapped k-mers that matched index entry l
 * (entry order of the arrays given to kmm_index_create; the frequency filter of every map call applied) since
 * the mode was switched on or the last kmm_reset_counts.  Needs kmm_set_param(idx, "count_kmers", 1) BEFORE the
 * map calls; the node counts are still maintained (they are the segmented sum of these counts over
 * nodes[l], which is how the reference's GPU counter derives them, gpu_counter.py:37).  out: uint32[n_entries],
 * host or device.  Synchronises.
 */
int kmm_get_kmer_counts(kmm_index_t *idx, uint32_t *out);

/*
 * Tuning knobs, for experiments and benchmarks (defaults are chosen at index creation):
 *   "path"             0 = auto (by batch size), 1 = direct fused kernel (one HBM gather per k-mer),
 *                      2 = radix path (two partition passes by hash range + probe of LDS-resident index
 *                      slices)
 *   "part_shift"       log2 of the number of hash buckets per fine partition of the radix path (0..13)
 *   "radix_min_units"  auto: smallest batch (positions / k-mers) that takes the radix path; default: where the two
 *                      paths break even under a cost model fitted to measurements (~0.38 M reads of 150 bp at a
 *                      100 M-k-mer index)
 *   "radix_grid_per_cu" persistent workgroups per CU of passes 2 and 3 (1 or 2; 2 by default)
 *   "radix_sorted_flush" 1 (default) = per-entry counts are added to the node counts through the node-ordered entry
 *                      list (built when an index has fewer than 8 entries per node on average); 0 = in bucket order
 *   "count_kmers"      1 = per-k-mer counting mode (see kmm_get_kmer_counts)
 *   "radix_filter"     1 (default) = pass 2 drops the k-mers whose bucket is empty (they cannot match: mapper.pyx:55-58)
 *                      wherever a coarse partition's occupancy bitmap fits 64 KB of LDS, at one bit per bucket or — sparse
 *                      tables — per 2 or 4 buckets ("radix_filter_buckets_per_bit", read-only); the fan-out is chosen for it
 *   "radix_packed_tiles" 1 (default) = pass 1 on reads of one length works on tiles of whole reads (no windows across
 *                      read boundaries are computed)
 *   "fine_bits"        experiments: log2 fine partitions per coarse partition of the radix path
 *   "radix_sub_batch_kmers" k-mer slots per sub-batch of the radix path: a larger map call is cut into equal sub-batches,
 *                      each a full run of the passes (the index slices are streamed once per sub-batch); default and
 *                      maximum 2^32 - 2 * 8192 (a coarse partition's k-mers are numbered with 32 bits).  When the batch
 *                      buffers of that size do not fit the free HBM a call takes one sub-batch more, and again (down to
 *                      2^28 slots); the size it ran with is "radix_sub_batch_kmers_effective" (read-only), kept for the
 *                      handle's next 15 calls, after which the caller's value is tried again
 *   "host_pack_threads" the host cores' share of the read bytes — the reference's `-t` (command_line_interface.py:168).  > 0: reads
 *                      that arrive in HOST memory (kmm_map_reads_uniform; kmm_map_reads with host offsets; kmm_map_records
 *                      with FASTQ / two-line FASTA bytes — a file mapping or an inflater's output, pinned or not; default
 *                      lookup table, a batch large enough for the radix path) are packed to 2 bits per base by that many
 *                      threads of a per-handle pool inside the call and cross PCIe at a quarter of their size
 *                      (csrc/kmm_hostpack.hpp: AVX-512 VBMI / AVX2 / scalar; for records the sequence lines go straight
 *                      from the raw bytes to the 2-bit stream + read-start bitset, by the rules of the device parser); a
 *                      byte outside the table or a malformed record sends the call down the ordinary route, which reports
 *                      it with its offset.  0: the bytes cross as they are.  Default: min(16, "host_cpu_budget") when that
 *                      budget — the CPUs of the affinity mask but one, cut by the cgroup's CPU quota — is at least 8, else
 *                      0; environment KMM_HOST_PACK_THREADS overrides it at index creation.
 *                      "host_packed_calls" / "host_packed_record_calls" (read-only) count the calls that took the route;
 *                      "host_pack_slice_kb": raw bytes per slice the records packer hands its threads (0 = default, 1024)
 *   "bgzf_head_skip" / "bgzf_tail_stop"  a RANK'S SHARE of a BGZF file (several processes on one file, kmm_map_bgzf): the next
 *                      call with KMM_FORMAT_NEW_STREAM passes over that many inflated bytes of its first member (they end a
 *                      record of the rank before), the next call with KMM_FORMAT_LAST_CHUNK takes only that many inflated
 *                      bytes of its last member (the rest starts the next rank's first record); each is used once
 *                      (tail: -1 = all, the default).  The boundaries are the caller's business
 *                      (kmer_mapper_amd/bgzf_ranges.py: record-structure resynchronisation on the members around a boundary)
 *   "comm_overlap_slices" kmm_comm_reduce_counts: node ranges whose flush (per-entry hits -> node counts) runs under the
 *                      previous range's RCCL reduce on a second stream (default 8; 1 = flush, then one reduce).  A
 *                      parameter of the JOB: every rank must use the same value — it alone (with the vector's length)
 *                      decides how many collectives a rank issues; a rank that cannot flush by node range (no node-ordered
 *                      entry list for lack of HBM, "count_kmers" mode, "radix_sorted_flush" 0) flushes everything first and
 *                      issues the same reduces.  "comm_sliced_reduces" (read-only) counts the calls that issued them
 *   "debug_records_copy_stream" / "debug_records_skip" / "debug_rx_*"  test hooks of tools/records_overlap_bisect.py (run
 *                      the compaction kernels of kmm_map_records on the copy stream, next to the radix passes; skip one
 *                      of them; directory sums of pass 1) and of the tests ("debug_rx_buffer_limit": a pass-1 buffer
 *                      beyond that many bytes counts as out of memory: the call takes more sub-batches;
 *                      "debug_skew_p2_counter": trips the conservation check; "debug_ring_slot_kb": slot size of the
 *                      page-locked staging ring): not for callers, no effect at 0
 * Read-only (kmm_get_param): "radix_available", "radix_unavailable_reason" (0 available, 1 modulo >= 2^31, 2 slices too
 *   dense for LDS, 3 out of memory, 4 the index's buckets overlap), "n_partitions", "n_coarse_partitions",
 *   "n_fine_per_coarse", "radix_p2_kmers" / "radix_p3_kmers" / "radix_p2_dropped" (the conservation counters every
 *   synchronising call compares: KMM_ERR_INTERNAL), "radix_batches" / "direct_batches" (which path the map calls took),
 *   "radix_view_bytes" / "direct_view_bytes" / "direct_view_resident" (HBM budget: the direct view of an index beyond
 *   16 GiB of it is packed on first use), "radix_p3_keys_in_lds" (entries of a slice pass 3 keeps in LDS: 4096, 4608 or
 *   8192; buckets behind them are walked in HBM), "wide_buckets", "occupancy_filter", "bloom_filter_bytes".
 * Unknown names return KMM_ERR_INVALID_ARG.
 */
int kmm_set_param(kmm_index_t *idx, const char *name, int64_t value);
int kmm_get_param(kmm_index_t *idx, const char *name, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* KMM_H */
