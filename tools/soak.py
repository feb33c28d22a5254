#!/usr/bin/env python3
"""Determinism soak: integer counts must be identical on every repetition (a data race in the LDS tile
front end, the aggregation table or the partitioned scatter would show up as a flaky mismatch)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import _lib, synthetic as syn          # noqa: E402
from kmer_mapper_amd.engine import DeviceIndex               # noqa: E402
from oracle import oracle                                     # noqa: E402


def fastq(bases, offs):
    out = []
    for i in range(len(offs) - 1):
        seq = bases[offs[i]:offs[i + 1]].tobytes()
        out.append(b"@r\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
    return np.frombuffer(b"".join(out), dtype=np.uint8)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    index, genome = syn.make_index(200000, seed=31)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 200000, 0, 300, seed=32)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=8)
    raw = fastq(bases[: offs[20000]], offs[:20001])
    expect_raw, _ = oracle.map_reads(index, mx, bases[: offs[20000]], offs[:20001], 31, n_threads=8)
    t0 = time.time()
    for env, label in ((None, "bitmap layout"), ("0", "wide layout")):
        if env is None:
            os.environ.pop("KMM_OCC_MAX_BYTES", None)
        else:
            os.environ["KMM_OCC_MAX_BYTES"] = env
        with DeviceIndex.from_index(index, mx) as dev:
            for path in (1, 2):
                if path == 2 and env is not None:
                    continue
                dev.set_param("path", path)
                bad = 0
                for i in range(reps):
                    dev.reset()
                    dev.map_reads(bases, offs, 31)
                    if not np.array_equal(dev.get_node_counts(), expect):
                        bad += 1
                print("%s, path %d, general reads: %d/%d repetitions bit-exact" % (label, path, reps - bad, reps), flush=True)
            dev.set_param("path", 1)
            bad = 0
            for i in range(reps):
                dev.reset()
                dev.map_records(raw, fmt=_lib.FORMAT_FASTQ)
                if not np.array_equal(dev.get_node_counts(), expect_raw):
                    bad += 1
            print("%s, records mode: %d/%d repetitions bit-exact" % (label, reps - bad, reps), flush=True)
    print("soak done in %.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
