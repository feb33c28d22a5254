#!/usr/bin/env python3
"""Rehearsal of the multi-rank CLI flow on a 1-GPU box: run under
   KMM_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/cli_two_rank_rehearsal.py
Every rank maps its own byte range of the plain FASTQ (reads_io.rank_byte_range); for the .gz copy (one stream,
not seekable) chunk i goes to rank i mod WORLD_SIZE.  Counts are summed with one reduce; rank 0 compares with
the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import reads_io, synthetic as syn                      # noqa: E402
from kmer_mapper_amd.command_line_interface import run_argument_parser      # noqa: E402
from kmer_mapper_amd.util import ReadBatch                                  # noqa: E402


def main():
    rank = int(os.environ.get("RANK", "0"))
    d = "/tmp/kmm_rehearsal"
    os.makedirs(d, exist_ok=True)
    index, genome = syn.make_index(20000, seed=5)
    bases, offs = syn.make_ragged_reads(genome, 30000, 20, 220, seed=6)
    idx_path, fq, fa = os.path.join(d, "index.npz"), os.path.join(d, "reads.fq"), os.path.join(d, "reads.fa")
    bgz_fq, bgz_fa = os.path.join(d, "reads_bgzf.fq.gz"), os.path.join(d, "reads_bgzf.fa.gz")
    if rank == 0:
        index.to_file(idx_path)
        reads_io.write_fastq(fq, ReadBatch(bases, offs))
        reads_io.write_fastq(fq + ".gz", ReadBatch(bases, offs), gz=True)
        reads_io.write_fasta(fa + ".gz", ReadBatch(bases, offs), gz=True)   # two-line FASTA through the GPU parser
        # BGZF (what bgzip writes): members inflated on the GPU, every rank its own member range (bgzf_ranges.py); small
        # members here, so that every rank's range starts and ends inside records
        from tools.bgzf_e2e import _member, _EOF
        for src, dst in ((fq, bgz_fq), (None, bgz_fa)):
            if src is None:
                reads_io.write_fasta(fa, ReadBatch(bases, offs))
                src = fa
            raw = open(src, "rb").read()
            with open(dst, "wb") as g:
                for p in range(0, len(raw), 23456):
                    g.write(_member(raw[p:p + 23456]))
                g.write(_EOF)
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("KMM_DIST_BACKEND", "gloo"))
    dist.barrier()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ok = True
    for path in (fq, fq + ".gz", fa + ".gz", bgz_fq, bgz_fa):
        out = os.path.join(d, "out_" + os.path.basename(path).replace(".", "_"))
        run_argument_parser(["map", "-i", idx_path, "-f", path, "-o", out, "-c", "300000"])
        dist.barrier()
        if rank == 0:
            from oracle import oracle
            expect, _ = oracle.map_reads(index, index.max_node_id(), bases, offs, 31, n_threads=4)
            got = np.load(out + ".npy")
            same = bool(np.array_equal(got, expect))
            ok = ok and same
            print("%d-rank CLI rehearsal on %s: %s" % (world, os.path.basename(path),
                                                        "BIT-EXACT" if same else "MISMATCH"), flush=True)
    dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
