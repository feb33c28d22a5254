import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bgzf_e2e, multiprocessing as mp
from kmer_mapper_amd import synthetic as syn, _lib
from kmer_mapper_amd.engine import DeviceIndex
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
index, genome = syn.make_index(10_000_000, seed=1, gpu_builder=True)
bases, offs = syn.make_reads(genome, n_reads, 150, seed=2)
fq = "/tmp/whole.fq"; bgzf_e2e.make_fastq(fq, bases, n_reads, 150); size = os.path.getsize(fq)
step = 0xFF00 * 256
with mp.Pool(16) as pool:
    comp = np.frombuffer(b"".join(pool.imap(bgzf_e2e._compress_range, [(fq, lo, min(lo + step, size)) for lo in range(0, size, step)])) + bgzf_e2e._EOF, dtype=np.uint8)
os.remove(fq)
with DeviceIndex.from_index(index, index.max_node_id()) as dev:
    for threads in (16, 0, 16):
        dev.set_param("host_pack_threads", threads)
        dev.reset()
        t = time.perf_counter()
        used, n = dev.map_bgzf(comp, fmt=_lib.FORMAT_FASTQ, k=31, first=True, last=True)
        c = dev.get_node_counts()
        print("host_pack_threads %d: whole file in one call: %.3f s (%d records)" % (threads, time.perf_counter() - t, n), flush=True)
