"""`kmer_mapper map …` — the reference's CLI surface (kmer_mapper/command_line_interface.py:155-192)
on the MI355X engine.  Same flags, defaults and output file; the work runs in libkmm.so.

Differences that follow from replacing the engine (all documented in DESIGN.md):
  * there is one engine, the GPU: `-g/--gpu` is accepted and ignored; `-s/--gpu-hash-map-size` is ignored (the
    index's own modulo is the hash table);
  * `-t/--n-threads` keeps its meaning — how many host cores work on the read bytes (the reference: a pool of `-t`
    processes that encode and hash every chunk, :124-130,168) — but the cores do less: they read / inflate the file
    and pack the sequence lines to 2 bits per base (libkmm's host packer, kmm_set_param "host_pack_threads"), so that
    a quarter of the bases' bytes cross PCIe; everything after that happens on the GPU.  Capped by the CPUs the
    process may actually use (affinity mask, cgroup quota);
  * results follow the reference's CPU path (uint32 node counts with the frequency filter of
    mapper.pyx:64-66).  The reference parses `-I/--max-hits-per-kmer` but never forwards it
    (command_line_interface.py:51 passes 3 arguments), so its effective filter is always 1000; the same
    holds here unless `--apply-max-hits-per-kmer` is given;
  * `-r` works (the reference only supports it in its experimental GPU mode, :107);
  * `-b/--index-bundle`, Minimal/Counter index variants (util.py:52-66) are out of scope -> error.
  * launched under torchrun (WORLD_SIZE > 1) every rank maps its own BYTE RANGE of the read file (both ends
    re-synchronised to record starts, reads_io.rank_byte_range: no rank reads or scans another rank's bytes);
    ranks sharing one .gz stream (not seekable) take chunk i mod WORLD_SIZE and cut the chunks they skip by the
    record parser's own rule (newline count, reads_io.records_cut); the count vectors are summed with one RCCL reduce; rank 0 writes the output.
"""
import argparse
import logging
import os
import sys
import time

import numpy as np

from .distributed import chunk_owner
from .engine import DeviceIndex
from .kmer_index import KmerIndex
from . import _lib
from .reads_io import (MmapChunker, RawChunker, PrefetchingRawChunker, prefetch, rank_byte_range, read_chunks, records_cut,
                       sniff_format)


def main():
    run_argument_parser(sys.argv[1:])


def _get_kmer_index_from_args(args):
    """util.py:38-68, restricted to plain KmerIndex (.npz) or an in-memory index object."""
    if args.kmer_index is None:
        if getattr(args, "index_bundle", None) is None:
            logging.error("Either a kmer index (-i) or an index bundle (-b) needs to be specified")
            sys.exit(1)
        logging.error("Index bundles (-b) are not supported by this build: pass the kmer index with -i")
        sys.exit(1)
    if not isinstance(args.kmer_index, (str, os.PathLike)):
        kmer_index = args.kmer_index          # already an index object (util.py:40-44), duck-typed
    else:
        if "minimal" in os.path.basename(str(args.kmer_index)):
            logging.error("MinimalKmerIndex files are not supported by this build")
            sys.exit(1)
        kmer_index = KmerIndex.from_file(args.kmer_index)
    if hasattr(kmer_index, "convert_to_int32"):
        kmer_index.convert_to_int32()
    if hasattr(kmer_index, "remove_ref_offsets"):
        kmer_index.remove_ref_offsets()
    return kmer_index


def map_cpu(args, kmer_index, chunk_sequence):
    """Per-chunk mapper with the reference's name and return value (command_line_interface.py:32-56):
    a fresh uint32 node-count vector for ONE chunk.  `chunk_sequence` is a ReadBatch (the reference
    passes a shared-memory name of the chunk; there is no process pool here).  N->A (:41), k-mer
    extraction (:42) and lookup (:51, default frequency filter 1000) all happen in the fused kernel."""
    from .mapper import _device_index
    from .util import as_read_batch
    k = args["kmer_size"] if isinstance(args, dict) else args.kmer_size
    batch = as_read_batch(chunk_sequence)
    max_node_id = kmer_index.max_node_id() if hasattr(kmer_index, "max_node_id") else int(np.max(kmer_index._nodes))
    dev = _device_index(kmer_index, max_node_id)
    dev.reset()
    dev.map_reads(batch.bases, batch.offsets, k)
    return dev.get_node_counts()


def host_threads(n_threads, world_size=1):
    """`-t` as the number of host threads that work on the read bytes of THIS rank: at most the cores the process may keep
    busy (affinity mask, cgroup quota), shared between the ranks of a node."""
    from . import _io
    return max(1, min(int(n_threads), _io.cpu_budget() // max(int(world_size), 1)))


def map_gpu(index, chunks, k, hash_map_size=0, map_reverse_complements=False,
            max_index_lookup_frequency=1000, device=0, rank=0, world_size=1, before_fetch=None, n_threads=16):
    """command_line_interface.py:59-79 on the HIP engine: chunks -> fused kmm_map_reads calls.
    before_fetch(dev): called with the open handle after the last chunk and before the counts are copied to the
    host (the multi-rank reduce runs there, on the device)."""
    max_node_id = index.max_node_id() if hasattr(index, "max_node_id") else int(np.max(index._nodes))
    dev = DeviceIndex.from_index(index, max_node_id, device=device)
    dev.set_param("host_pack_threads", host_threads(n_threads, world_size) if n_threads > 1 else 0)
    t_start = time.perf_counter()
    n_kmers = 0
    try:
        for i, chunk in enumerate(chunks):
            if chunk is None:          # a chunk of a shared .gz stream that another rank maps
                continue
            t0 = time.perf_counter()
            L = chunk.uniform_length
            if L is not None:
                dev.map_reads_uniform(chunk.bases, len(chunk), L, k, max_index_lookup_frequency,
                                      also_revcomp=map_reverse_complements)
            else:
                dev.map_reads(chunk.bases, chunk.offsets, k, max_index_lookup_frequency,
                              also_revcomp=map_reverse_complements)
            n_kmers += chunk.n_kmers(k)
            logging.debug("GPU: chunk %d (%d reads) submitted in %.5f sec", i, len(chunk),
                          time.perf_counter() - t0)
        if before_fetch is not None:
            before_fetch(dev)
        node_counts = dev.get_node_counts()
    finally:
        dev.close()
    dt = time.perf_counter() - t_start
    logging.info("Time spent only on hashing and counting hashes: %.5f" % dt)
    logging.info("Mapped %d k-mers (%.1f M k-mers/s)" % (n_kmers, n_kmers / max(dt, 1e-9) / 1e6))
    return node_counts


def map_gpu_raw(index, path, chunk_size, fmt, k, map_reverse_complements=False,
                max_index_lookup_frequency=1000, device=0, rank=0, world_size=1, before_fetch=None, n_threads=16):
    """Same job as map_gpu, but the FASTQ / two-line FASTA records are parsed ON THE GPU
    (kmm_map_records): the host only reads (and for .gz inflates) raw bytes."""
    t_index = time.perf_counter()
    # page-locked memory is slow to make (~50 ms per GB): the staging buffers of the host packer are made by a helper thread
    # WHILE the index is uploaded and repacked, not inside the map phase.  (The count vector needs none: kmm_get_node_counts
    # brings a large vector to ordinary memory through the handle's page-locked ring at the link's rate.)
    import threading
    prepared = {}

    def prepare_host_memory():
        try:
            size = os.stat(path).st_size
            if n_threads > 1 and not str(path).endswith(".gz"):
                _lib.check(_lib.lib().kmm_host_reserve(min(size // max(world_size, 1) + (1 << 20), 2 << 30)))
        except Exception as exc:                         # noqa: BLE001 - an optimisation: the map calls allocate what is missing
            logging.debug("host memory was not prepared ahead: %s", exc)

    helper = threading.Thread(target=prepare_host_memory, daemon=True)
    # (.gz input needs 128 MB of it, made in 7 ms by the first call that wants it; behind a helper thread the same allocation
    # came back 100 ms after the index upload it was meant to hide behind: profiles/r05/bgzf_e2e_v6_*.txt)
    if _lib.device_count() > 0 and n_threads > 1 and not str(path).endswith(".gz"):
        helper.start()
    # A plain FASTQ / two-line FASTA is mapped into memory and handed to the packer threads as it lies in the page cache
    # (MmapChunker).  The mapping is made HERE, before the index goes up, and helper threads populate its page tables
    # meanwhile (MADV_POPULATE_READ): the 16 packer threads otherwise take a page fault per 64 KiB of a fresh mapping,
    # all in one address space (profiles/r05/cli_populate_ab.txt).
    seekable = not str(path).endswith(".gz")
    byte_range = rank_byte_range(path, fmt, rank, world_size) if (world_size > 1 and seekable) else None
    early = None
    if (seekable and fmt in ("fastq", "fasta") and n_threads > 1 and _lib.device_count() > 0
            and not os.environ.get("KMM_CLI_NO_MMAP")):
        early = MmapChunker(path, int(chunk_size), byte_range, pinned=True)
        if not os.environ.get("KMM_CLI_NO_POPULATE"):
            early.populate(n_threads=max(1, min(4, host_threads(n_threads, world_size) // 2)))
    # (the scan for the largest node id — 30 ms on one thread for 10^8 entries — runs with the helpers above already at work)
    max_node_id = index.max_node_id() if hasattr(index, "max_node_id") else int(np.max(index._nodes))
    dev = DeviceIndex.from_index(index, max_node_id, device=device)
    logging.info("Index resident in HBM after %.3f sec (max_node_id scan + upload + repack)", time.perf_counter() - t_index)
    # -t: the host cores' share of the work (reference: command_line_interface.py:124-130,168) — reader / inflate threads
    # and the threads that pack the sequence lines to 2 bits per base inside kmm_map_records; -t 1 = no host packing,
    # the raw bytes cross PCIe and the GPU parses them
    n_host = host_threads(n_threads, world_size)
    dev.set_param("host_pack_threads", n_host if n_threads > 1 else 0)
    from . import _io
    _io.set_default_threads(n_host)
    logging.info("%d host thread(s) read and pack the read bytes (-t %d, CPU budget %d)", n_host, n_threads, _io.cpu_budget())
    if byte_range is not None:
        logging.info("Rank %d of %d maps bytes [%d, %d) of %s", rank, world_size, byte_range[0], byte_range[1], path)
    # GPU batches: the reference maps chunk by chunk (-c bytes, command_line_interface.py:109-111,169); here the chunks
    # of a large file are accumulated until one map call holds enough positions for the radix path to run well (a raw
    # FASTQ piece takes it from 2 x radix_min_units bytes on: its sequence lines are compacted into flat reads on the
    # device first, kmm_map_records), at most 2 GiB per call.
    batch_bytes = int(chunk_size)
    try:
        n_file = os.stat(path).st_size * (6.5 if not seekable else 1)
    except OSError:
        n_file = 0
    if dev.get_param("radix_available") and not os.environ.get("KMM_CLI_NO_BATCHING"):
        # (radix_min_units is where the radix path BREAKS EVEN with the direct kernel, in base positions ~ half the
        # FASTQ bytes; a batch twelve times that runs within 20 % of the path's large-batch rate.  Larger batches would
        # run the GPU closer to its large-batch rate, but end to end the host is the bound — reading the file into pinned
        # memory at 8-11 GB/s — and several batches per file let that overlap the copies and kernels: a 3 GB FASTQ took
        # 0.27 s in five batches of 615 MB and 0.62 s as ONE batch, profiles/r04/cli_e2e_large_fastq.txt.)
        want = min(max(int(12 * dev.get_param("radix_min_units")), 256 << 20), 2 << 30)
        share = n_file / max(world_size, 1)
        if share >= want > batch_bytes:
            # equal batches, none below the threshold (a small last batch would take the direct path)
            batch_bytes = min(int(share / int(share // want)) + (1 << 20), 2 << 30) if seekable else want
            logging.info("Chunks of %d bytes are accumulated into GPU batches of %d bytes (radix path)", chunk_size, batch_bytes)
        elif seekable and share > batch_bytes and share >= 2.5 * dev.get_param("radix_min_units"):
            # a file (or a rank's share of one) between the radix path's break-even and the batch size above: ONE call.  Chunk
            # by chunk it is 271 direct-path calls for 675 MB — 39-42 ms of map calls against 7.5 + 4.4 ms of GPU tail as one
            # call packed by the host threads (profiles/r05/cli_mid_sized_file.txt)
            batch_bytes = min(int(share) + (1 << 20), 2 << 30)
            logging.info("Chunks of %d bytes are accumulated into ONE GPU batch of %d bytes (radix path)", chunk_size, batch_bytes)
    # .gz input: two pinned buffers and a reader thread — the next batch is inflated while the GPU works on this one
    # (BGZF 5.7 -> 6.5 GB/s end to end); plain files are read at memory speed and the second pinned buffer costs more
    # than the overlap returns (3 GB FASTQ: 0.30 s with one buffer, 0.37 s with two)
    use_prefetch = not seekable and not os.environ.get("KMM_CLI_NO_PREFETCH")
    # plain FASTQ / two-line FASTA with host packing: the chunks are views of the file mapping (no copy, nothing pinned)
    use_mmap = (seekable and fmt in ("fastq", "fasta") and dev.get_param("host_pack_threads") > 0
                and not os.environ.get("KMM_CLI_NO_MMAP"))
    # BGZF (.gz written by bgzip / htslib: independent members of <= 64 KiB): the compressed bytes go to the GPU as they lie
    # in the file mapping, one GPU thread inflates one member, the records are parsed there too (kmm_map_bgzf) — the host's
    # inflater (10.8 GB/s of FASTQ on 16 cores) is out of the way.  Several ranks: each takes the members that start in its share
    # of the compressed bytes, resynchronised to the records at both ends (bgzf_ranges.py).
    # (Decided BEFORE a chunker is made: the prefetching one starts a reader thread and page-locks two batch buffers — making
    # and freeing those cost this route 100 ms of its map phase until it was noticed.)
    gpu_inflate = (not seekable and fmt in ("fastq", "fasta") and not os.environ.get("KMM_CLI_NO_GPU_INFLATE") and _is_bgzf(path))
    if early is not None and use_mmap and not gpu_inflate:
        chunker = early
        chunker.chunk_size = batch_bytes
    else:
        if early is not None:
            early.close()
        chunker = None if gpu_inflate else (MmapChunker if use_mmap else PrefetchingRawChunker if use_prefetch
                                            else RawChunker)(path, batch_bytes, byte_range, pinned=True)
    steered_from = None
    if chunker is not None and chunker is early and not os.environ.get("KMM_CLI_NO_PACKER_STEERING"):
        # the packer threads are made by the first map call and inherit this thread's CPUs: next to the file's pages
        from .distributed import packer_cpus_near
        try:
            where = chunker.page_nodes()
            near = packer_cpus_near(where)
            if near is not None:
                steered_from = os.sched_getaffinity(0)
                os.sched_setaffinity(0, near[1])
                logging.info("The read file's page-cache pages lie on NUMA node %d (%s): its packer threads run there (%d CPUs), "
                             "the packed stream crosses to the GPU's node", near[0], where, len(near[1]))
        except (OSError, AttributeError) as exc:
            logging.debug("packer threads stay on the GPU's node: %s", exc)
    owns = (lambda i: True) if (world_size == 1 or seekable) else (lambda i: chunk_owner(i, world_size) == rank)
    # FASTQ and two-line FASTA are parsed as they are; FASTA with wrapped sequence lines is unwrapped on the GPU first
    kfmt = {"fastq": _lib.FORMAT_FASTQ, "fasta": _lib.FORMAT_FASTA2, "fasta_ml": _lib.FORMAT_FASTA}[fmt]
    t_start = time.perf_counter()
    n_reads = n_bytes = 0
    if gpu_inflate:
        if helper.ident is not None:
            helper.join()
        return _map_bgzf_file(dev, path, kfmt, k, max_index_lookup_frequency, map_reverse_complements, before_fetch, t_start,
                              counts_out=prepared.get("counts"), rank=rank, world_size=world_size, fmt=fmt)
    try:
        i = 0
        while True:
            buf = chunker.next_chunk()
            if buf is None:
                break
            last = _lib.FORMAT_LAST_CHUNK if (fmt == "fasta_ml" and chunker.eof) else 0
            if owns(i):
                used, n_rec = dev.map_records(buf, buf.shape[0], kfmt | last, k, max_index_lookup_frequency,
                                              also_revcomp=map_reverse_complements)
            else:   # a chunk of a shared .gz stream that another rank maps: only its record boundary is needed,
                    # cut by the SAME rule as the GPU parser's `consumed` (newline count), at end of input too
                used, n_rec = records_cut(buf, fmt, chunker.eof), 0
            if used == 0:
                if chunker.eof:
                    raise ValueError("trailing bytes at end of %s do not form a complete record" % path)
                chunker.chunk_size *= 2          # a record longer than the chunk: read more
                continue
            n_reads += n_rec
            n_bytes += used
            chunker.consumed(used)
            i += 1
        t_calls = time.perf_counter()
        n_lookups, n_hits = dev.get_stats()
        n_radix, n_direct = dev.get_param("radix_batches"), dev.get_param("direct_batches")
        n_host_packed = dev.get_param("host_packed_record_calls")
        if before_fetch is not None:
            before_fetch(dev)
        if helper.is_alive() or helper.ident is not None:
            helper.join()
        t_fetch = time.perf_counter()
        node_counts = dev.get_node_counts(out=prepared.get("counts"))
        # (the reference's timer stops before its get_node_counts, command_line_interface.py:78-79; ours runs on through
        # the fetch, and says what the fetch was)
        if getattr(chunker, "populate_t1", None) is not None:
            logging.info("The file mapping's pages were populated in %.1f ms, done %.1f ms %s the first map call",
                         (chunker.populate_t1 - chunker.populate_t0) * 1e3, abs(t_start - chunker.populate_t1) * 1e3,
                         "before" if chunker.populate_t1 <= t_start else "AFTER")
        logging.info("%.1f ms in the map calls, %.1f ms until the GPU had finished them (= %.1f M k-mers/s up to where the "
                     "reference stops its timer), %.1f ms more until the node counts were on the host",
                     (t_calls - t_start) * 1e3, (t_fetch - t_calls) * 1e3, n_lookups / max(t_fetch - t_start, 1e-9) / 1e6,
                     (time.perf_counter() - t_fetch) * 1e3)
    finally:
        dt = time.perf_counter() - t_start
        chunker.close()
        dev.close()
        if steered_from is not None:                     # (this thread goes back to the GPU's node)
            try:
                os.sched_setaffinity(0, steered_from)
            except OSError:
                pass
    logging.info("Time spent only on hashing and counting hashes: %.5f" % dt)
    logging.info("Mapped %d reads from %d bytes (%.1f MB/s, GPU record parser): %d k-mer lookups "
                 "(%.1f M/s), %d index hits" % (n_reads, n_bytes, n_bytes / max(dt, 1e-9) / 1e6, n_lookups,
                                                  n_lookups / max(dt, 1e-9) / 1e6, n_hits))
    logging.info("path_taken: %s (%d batches on the radix path, %d on the direct path; %d batches packed to 2 bits per base "
                 "by the host threads)"
                 % ("radix" if n_radix and not n_direct else "direct" if n_direct and not n_radix else "mixed", n_radix, n_direct,
                    n_host_packed))
    return node_counts


def _is_bgzf(path):
    """Does the file start with a BGZF member (gzip header with the BC extra subfield, SAM specification 4.1)?"""
    try:
        with open(path, "rb") as f:
            h = f.read(18)
    except OSError:
        return False
    return len(h) == 18 and h[:4] == b"\x1f\x8b\x08\x04" and h[12:14] == b"BC" and h[14:16] == b"\x02\x00"


_BGZF_CALL_INFLATED = 3150 << 20   # inflated bytes per kmm_map_bgzf call: under what a call takes (3.5 GiB; 3.25 GiB for a window staged ahead)


def _map_bgzf_file(dev, path, kfmt, k, max_freq, revcomp, before_fetch, t_start, comp_batch=None, counts_out=None, rank=0,
                   world_size=1, fmt="fastq"):
    """`kmer_mapper map -f reads.fq.gz` for BGZF files: compressed chunks of the file mapping -> kmm_map_bgzf (members inflated
    and records parsed on the GPU; the handle carries the bytes behind a chunk's last complete record to the next chunk).
    Several ranks: each maps its member range (bgzf_ranges.rank_member_range), the first member's head and the last member's
    tail trimmed to the record boundaries the ranks agree on."""
    import mmap
    n_reads = lo = size = 0
    try:
        with open(path, "rb") as f:
            file_size = os.fstat(f.fileno()).st_size
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                lo, size, head_skip, tail_stop = 0, file_size, 0, None
                if world_size > 1:
                    from . import bgzf_ranges
                    lo, s0, hi, s1 = bgzf_ranges.rank_member_range(mm, fmt, rank, world_size)
                    size = bgzf_ranges.member_end(mm, hi) if s1 > 0 else hi
                    head_skip, tail_stop = s0, (s1 if s1 > 0 else None)
                    logging.info("Rank %d of %d maps the BGZF members in compressed bytes [%d, %d) of %d: %d inflated bytes of the "
                                 "first member skipped, %s of the last one taken", rank, world_size, lo, size, file_size, s0,
                                 "all" if tail_stop is None else "%d bytes" % s1)
                if hasattr(mm, "madvise") and hasattr(mmap, "MADV_SEQUENTIAL"):
                    mm.madvise(mmap.MADV_SEQUENTIAL)
                whole = np.frombuffer(mm, dtype=np.uint8)
                # equal windows, none small: one GPU thread inflates one member, a call's time is one member's (~tens of
                # milliseconds) whatever its size — so as FEW calls as their inflated size allows: the compressed bytes of a
                # call follow from the file's own ratio (its first members' ISIZE against their sizes; FASTQ with binned
                # qualities 3.3, with forty quality values 2.1 — one call instead of two for 1.5 GB of it)
                from . import bgzf_ranges as _br
                per_call = int(_BGZF_CALL_INFLATED / max(1.0, _br.inflation_ratio(mm, lo, size)))
                n_calls = max(1, -(-(size - lo) // per_call))
                pos, window = lo, int(comp_batch) if comp_batch else (size - lo) // n_calls + (1 << 16)
                t_calls = time.perf_counter()
                end = min(lo + window, size)             # windows END at fixed places; a window starts where the one before
                while pos < size:                        # it ran out of whole members (at most 64 KiB in front of that end)
                    nxt = min(end + window, size)
                    used, n_rec = dev.map_bgzf(whole[pos:end], fmt=kfmt, k=k, max_index_lookup_frequency=max_freq,
                                               also_revcomp=revcomp, first=pos == lo, last=end == size,
                                               head_skip=head_skip if pos == lo else 0,
                                               tail_stop=tail_stop if end == size else None,
                                               next_chunk=whole[end:nxt] if nxt > end else None)
                    if used == 0 and end == size:
                        raise ValueError("trailing bytes of %s are no complete BGZF member" % path)
                    pos += used
                    n_reads += n_rec
                    if pos < end and end == size:        # the last window held more than one call takes: go on
                        continue
                    end = nxt
                del whole
            finally:
                try:
                    mm.close()
                except BufferError:
                    pass
        n_lookups, n_hits = dev.get_stats()
        n_members = dev.get_param("bgzf_members")
        n_radix, n_direct = dev.get_param("radix_batches"), dev.get_param("direct_batches")
        if before_fetch is not None:
            before_fetch(dev)
        t_fetch = time.perf_counter()
        node_counts = dev.get_node_counts(out=counts_out)
        logging.info("%.0f ms in kmm_map_bgzf, %.0f ms more until the node counts were on the host",
                     (t_fetch - t_calls) * 1e3, (time.perf_counter() - t_fetch) * 1e3)
    finally:
        dt = time.perf_counter() - t_start
        dev.close()
    logging.info("Time spent only on hashing and counting hashes: %.5f" % dt)
    logging.info("Mapped %d reads from %d compressed bytes (%.1f MB/s compressed; %d BGZF members inflated on the GPU): %d k-mer "
                 "lookups (%.1f M/s), %d index hits" % (n_reads, size - lo, (size - lo) / max(dt, 1e-9) / 1e6, n_members, n_lookups,
                                                          n_lookups / max(dt, 1e-9) / 1e6, n_hits))
    logging.info("path_taken: %s (%d batches on the radix path, %d on the direct path; 0 batches packed to 2 bits per base by "
                 "the host threads)" % ("radix" if n_radix and not n_direct else "direct" if n_direct and not n_radix else "mixed",
                                        n_radix, n_direct))
    return node_counts


def map_bnp(args):
    if args.debug:
        logging.info("Will print debug log")
        logging.getLogger().setLevel(logging.DEBUG)

    k = args.kmer_size
    start_time = time.perf_counter()
    kmer_index = _get_kmer_index_from_args(args)

    n_bytes = os.stat(args.reads).st_size
    if str(args.reads).endswith(".gz"):
        n_bytes *= 6.5  # rough estimate for gzipped to give a progress (reference :92-93)
    logging.info("N bytes of reads: %d" % n_bytes)
    logging.info("Approx number of chunks of %d bytes: %d" % (args.chunk_size, int(n_bytes / args.chunk_size)))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    max_freq = args.max_hits_per_kmer if getattr(args, "apply_max_hits_per_kmer", False) else 1000

    device = local_rank if world > 1 else getattr(args, "device", 0)
    backend = os.environ.get("KMM_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 flow on one GPU
    if world > 1 and backend == "gloo":
        device = local_rank % max(_lib.device_count(), 1)
    before_fetch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        from .distributed import init_rccl_comm, reduce_node_counts
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl" and torch.cuda.is_available():
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl")
            else:
                dist.init_process_group("gloo")
        if dist.get_backend() == "nccl":
            # the additive reduce of command_line_interface.py:124-130 as ONE RCCL sum on the device, in place on the
            # handle's count vector (kmm_comm_reduce_counts); torch.distributed only carries the 128-byte communicator id
            def before_fetch(dev):
                t0 = time.perf_counter()
                init_rccl_comm(dev)
                dev.comm_reduce_counts(root=0)
                logging.info("Rank %d: RCCL reduce of the node counts on the device: %.5f sec", rank, time.perf_counter() - t0)
    # the process next to its GPU: reader / packing threads and their page-locked buffers on that NUMA node (16 threads pack
    # 172 GB/s of FASTQ from the GPU's own node, 132 spread over both sockets: profiles/r05/hostpack_rate.txt)
    from .distributed import bind_to_gpu_numa_node
    logging.info("Rank %d: host side bound to its GPU's NUMA node: %s", rank, bind_to_gpu_numa_node(device))
    revcomp = bool(getattr(args, "map_reverse_complements", False))
    fmt, two_line = sniff_format(args.reads)
    if not getattr(args, "host_parser", False):
        if fmt == "fasta" and not two_line:
            fmt = "fasta_ml"           # wrapped sequence lines: unwrapped on the GPU (KMM_FORMAT_FASTA)
        node_counts = map_gpu_raw(kmer_index, args.reads, args.chunk_size, fmt, k, revcomp, max_freq,
                                  device=device, rank=rank, world_size=world, before_fetch=before_fetch,
                                  n_threads=args.n_threads)
    else:
        logging.info("Using the host FASTA/FASTQ parser")
        seekable = not str(args.reads).endswith(".gz")
        if world > 1 and seekable:
            chunks = read_chunks(args.reads, min_chunk_size=args.chunk_size,
                                 byte_range=rank_byte_range(args.reads, fmt, rank, world))
        elif world > 1:
            chunks = read_chunks(args.reads, min_chunk_size=args.chunk_size,
                                 owned=lambda i: chunk_owner(i, world) == rank)
        else:
            chunks = read_chunks(args.reads, min_chunk_size=args.chunk_size)
        chunks = prefetch(chunks)
        node_counts = map_gpu(kmer_index, chunks, k, getattr(args, "gpu_hash_map_size", 0), revcomp,
                              max_freq, device=device, rank=rank, world_size=world, before_fetch=before_fetch,
                              n_threads=args.n_threads)

    if world > 1:
        if before_fetch is None:        # gloo rehearsal on a 1-GPU box: the sum runs on the host copies
            t = torch.from_numpy(node_counts.view(np.int32).copy())
            reduce_node_counts(t, dst=0)
            node_counts = t.numpy().view(np.uint32)
        if rank != 0:
            return node_counts

    if args.output_file is None:
        return node_counts

    np.save(args.output_file, node_counts)
    logging.info("Saved node counts to %s.npy" % args.output_file)
    logging.info("Spent %.3f sec in total mapping kmers using %d threads"
                 % (time.perf_counter() - start_time, args.n_threads))
    return node_counts


def run_argument_parser(args):
    logging.basicConfig(stream=sys.stdout, level=logging.INFO,
                        format='%(asctime)s %(levelname)s: %(message)s')
    parser = argparse.ArgumentParser(
        description='Kmer Mapper',
        prog='kmer_mapper',
        formatter_class=lambda prog: argparse.HelpFormatter(prog, max_help_position=50, width=100))

    subparsers = parser.add_subparsers()
    subparser = subparsers.add_parser("map", help="Map reads to a kmer index")
    subparser.add_argument("-i", "--kmer-index", required=False)
    subparser.add_argument("-b", "--index-bundle", required=False)
    subparser.add_argument("-f", "--reads", required=True, help="Reads in .fa, .fq, .fa.gz, or fq.gz format")
    subparser.add_argument("-k", "--kmer-size", required=False, default=31, type=int)
    subparser.add_argument("-t", "--n-threads", required=False, default=16, type=int,
                           help="Host threads that read / inflate the reads and pack them to 2 bits per base before they "
                                "cross PCIe (1: none, the raw bytes cross). Default 16.")
    subparser.add_argument("-c", "--chunk-size", required=False, type=int, default=2500000,
                           help="N bytes to process in each chunk")
    subparser.add_argument("-o", "--output-file", required=True)
    subparser.add_argument("-d", "--debug", required=False, help="Set to True to print debug log")
    subparser.add_argument("-I", "--max-hits-per-kmer", required=False, default=1000, type=int,
                           help="Ignore kmers that have more than this amount of hits in index")
    subparser.add_argument("-g", "--gpu", default=False, type=bool,
                           help="Accepted for compatibility: this build always maps on the GPU.")
    subparser.add_argument("-s", "--gpu-hash-map-size", default=0, type=int,
                           help="Accepted for compatibility; the index's own modulo is used.")
    subparser.add_argument("-r", "--map-reverse-complements", default=False, type=bool,
                           help="Also count kmers in reverse complement of reads. "
                                "Default False. Not necessary if index contains reverse complements.")
    subparser.add_argument("--apply-max-hits-per-kmer", action="store_true",
                           help="Extension: actually apply -I (the reference parses it but always uses 1000).")
    subparser.add_argument("--host-parser", action="store_true",
                           help="Extension: parse records on the host instead of on the GPU (kmm_map_records; wrapped FASTA is "
                                "unwrapped on the GPU too).")
    subparser.add_argument("--device", default=0, type=int, help="Extension: GPU ordinal (single process).")
    subparser.set_defaults(func=map_bnp)

    if len(args) == 0:
        parser.print_help()
        sys.exit(1)

    args = parser.parse_args(args)
    return args.func(args)


if __name__ == "__main__":
    main()
