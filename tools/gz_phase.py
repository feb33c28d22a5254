#!/usr/bin/env python3
"""Phase times of the GPU BGZF inflater (kmm_gpu_inflate.hpp built with -DKMM_GZ_TIMERS into a scratch library).
    python tools/gz_phase.py [n_reads=3000000] [grid_waves=0: as the library sizes it] [level=6]
FASTQ as tools/bgzf_e2e.py makes it, compressed into BGZF members of 0xFF00 bytes; prints the kernel time and, per lane,
the 10 ns ticks spent in block headers / symbol decoding / match copies / CRC (mean and maximum over lanes)."""
import ctypes
import os
import struct
import subprocess
import sys
import tempfile
import zlib
import multiprocessing as mp

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
    grid_waves = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import bgzf_e2e
    from kmer_mapper_amd import synthetic as syn
    tmp = tempfile.mkdtemp()
    so = os.path.join(tmp, "gz_phase.so")
    flags = os.environ.get("GZ_PHASE_FLAGS", "").split()
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-DKMM_GZ_TIMERS", "-shared", "-fPIC", "-std=c++17"] + flags +
                          ["-I" + os.path.join(ROOT, "kmer_mapper_amd", "csrc"), os.path.join(ROOT, "tools", "gz_phase.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    rng = np.random.default_rng(3)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n_reads * 150, dtype=np.uint8)]
    fq = os.path.join(tmp, "reads.fq")
    bgzf_e2e.make_fastq(fq, bases, n_reads, 150)
    size = os.path.getsize(fq)
    step = 0xFF00 * 256
    with mp.Pool(min(16, os.cpu_count() or 1)) as pool:
        comp = b"".join(pool.imap(bgzf_e2e._compress_range, [(fq, lo, min(lo + step, size)) for lo in range(0, size, step)]))
    raw = np.fromfile(fq, dtype=np.uint8)
    os.remove(fq)
    m_off, o_off, p, o = [0], [0], 0, 0
    while p < len(comp):
        bsize = struct.unpack_from("<H", comp, p + 16)[0] + 1
        isize = struct.unpack_from("<I", comp, p + bsize - 4)[0]
        p += bsize
        o += isize
        m_off.append(p)
        o_off.append(o)
    assert o == size
    n_members = len(m_off) - 1
    if grid_waves <= 0:
        grid_waves = min((n_members + 63) // 64, 1024)
    grid_threads = grid_waves * 64
    cbuf = np.frombuffer(comp, dtype=np.uint8)
    m = np.array(m_off, dtype=np.uint64)
    oo = np.array(o_off, dtype=np.uint64)
    out = np.zeros(size, dtype=np.uint8)
    timers = np.zeros((grid_threads, 8), dtype=np.uint64)
    err = np.zeros(3, dtype=np.uint32)
    ms = ctypes.c_double(0)
    lib.gz_phase.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64,
                             ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_void_p, ctypes.c_void_p]
    rc = lib.gz_phase(cbuf.ctypes.data, len(comp), m.ctypes.data, oo.ctypes.data, n_members, size, out.ctypes.data, grid_threads, 3,
                      ctypes.byref(ms), timers.ctypes.data, err.ctypes.data)
    assert rc == 0
    same = bool(np.array_equal(out, raw))
    print("%d members, %.1f MB -> %.1f MB, %d wavefronts: kernel %.2f ms = %.1f GB/s out; errors %s; output equal: %s"
          % (n_members, len(comp) / 1e6, size / 1e6, grid_waves, ms.value, size / ms.value / 1e6, err.tolist(), same), flush=True)
    per_lane_members = n_members / grid_threads
    t = timers.astype(np.float64)
    used = t[:, 5] > 0
    names = ["block headers", "symbol decoding", "match copies"]
    for k in range(3):
        print("  %-16s mean %8.2f ms  max %8.2f ms per lane (%.2f members per lane)" % (names[k], t[used, k].mean() / 1e5,
                                                                                   t[used, k].max() / 1e5, per_lane_members))
    print("  per member: %.2f block headers, %.1f rounds of (decode, copy), %.0f matches in %.0f steps of the copy phase"
          % ((timers[used, 4] & 0xFFFFFFFF).sum() / n_members, t[used, 5].sum() / n_members, t[used, 7].sum() / n_members,
             t[used, 6].sum() / n_members))
    print("              %.0f symbols, %.0f literal / length codes and %.0f distance codes through a subtable"
          % ((timers[used, 4] >> 32).sum() / n_members, (timers[used, 3] & 0xFFFFFFFF).sum() / n_members,
             (timers[used, 3] >> 32).sum() / n_members))
    if not same or err[0]:
        sys.exit(1)


if __name__ == "__main__":
    main()
