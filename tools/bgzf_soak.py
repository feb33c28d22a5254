#!/usr/bin/env python3
"""Soak of kmm_map_bgzf: random FASTQ files compressed into BGZF members of random sizes, fed in windows of random size with
and without the hint for the next window (kmm_map_bgzf_hint_next: staged under the current window's kernel), random staging
ring slot sizes and thread counts, as one rank or as several member ranges (bgzf_ranges) — every run's counts compared with
the direct kernel's on the same reads.
    python tools/bgzf_soak.py [rounds=60]"""
import os
import struct
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _member(chunk, level):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    payload = c.compress(chunk) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload +
            struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def main():
    from kmer_mapper_amd import _lib, bgzf_ranges, synthetic as syn
    from kmer_mapper_amd.engine import DeviceIndex
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    index, genome = syn.make_index(2_000_000, seed=1, gpu_builder=True)
    mx = index.max_node_id()
    rng = np.random.default_rng(2024)
    fails = 0
    t0 = time.perf_counter()
    with DeviceIndex.from_index(index, mx) as dev, DeviceIndex.from_index(index, mx) as ref:
        ref.set_param("path", 1)
        for r in range(rounds):
            n_reads = int(rng.integers(2_000, 400_000))
            bases, offs = syn.make_ragged_reads(genome, n_reads, 31, 260, seed=1000 + r)
            lens = np.diff(offs)
            # FASTQ without a Python loop per read: header "@<i>\n", sequence, "+\n", quality, "\n"
            parts = []
            q = rng.choice(np.frombuffer(b"FFFFFF:,#@+", dtype=np.uint8), size=int(offs[-1]))
            for i in range(n_reads):
                parts.append(b"@%d\n" % i)
                parts.append(bases[offs[i]:offs[i + 1]].tobytes())
                parts.append(b"\n+\n")
                parts.append(q[offs[i]:offs[i + 1]].tobytes())
                parts.append(b"\n")
            raw = b"".join(parts)
            block = int(rng.choice([0xFF00, 0xFF00, 20000, 3000, 700]))
            level = int(rng.choice([1, 6, 9, 0]))
            comp = b"".join(_member(raw[p:p + block], level) for p in range(0, len(raw), block)) + _EOF
            buf = np.frombuffer(comp, dtype=np.uint8)
            ref.reset()
            ref.map_reads(bases, offs, 31)
            want = ref.get_node_counts()
            threads = int(rng.choice([0, 2, 5, 16]))
            slot_kb = int(rng.choice([0, 0, 4, 64, 1024]))
            world = int(rng.choice([1, 1, 2, 3, 7]))
            step = int(rng.choice([1 << 30, len(comp) // 3 + 1, 200_001, 70_000]))
            hinted = bool(rng.integers(0, 2))
            dev.set_param("host_pack_threads", threads)
            dev.set_param("debug_ring_slot_kb", slot_kb)
            dev.reset()
            total = 0
            for rank in range(world):
                lo, s0, hi, s1 = (0, 0, len(comp), 0) if world == 1 else bgzf_ranges.rank_member_range(comp, "fastq", rank, world)
                if world > 1:
                    hi = bgzf_ranges.member_end(comp, hi) if s1 > 0 else hi
                pos, end = lo, min(lo + step, hi)
                while pos < hi:
                    nxt = min(end + step, hi)
                    used, n_rec = dev.map_bgzf(buf[pos:end], fmt=_lib.FORMAT_FASTQ, k=31, first=pos == lo, last=end == hi,
                                               head_skip=s0 if pos == lo else 0, tail_stop=(s1 if s1 > 0 else None) if end == hi else None,
                                               next_chunk=buf[end:nxt] if (hinted and nxt > end) else None)
                    pos += used
                    total += n_rec
                    if pos < end and end == hi:
                        continue
                    end = nxt
            got = dev.get_node_counts()
            ok = total == n_reads and np.array_equal(got, want) and dev.get_param("bgzf_carry_bytes") == 0
            fails += 0 if ok else 1
            print("round %d: %d reads, members of %d bytes at level %d, %d rank(s), windows of %d%s, %d threads, ring slots of %d KiB: %s"
                  % (r, n_reads, block, level, world, step, " announced ahead" if hinted else "", threads, slot_kb or 16384,
                     "same as the direct kernel" if ok else "DIFFERENT (%d records)" % total), flush=True)
    print("failures: %d of %d (%.1f s)" % (fails, rounds, time.perf_counter() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
