#!/usr/bin/env python3
"""Rates of the host packer (csrc/kmm_hostpack.hpp) on this host, by thread count, without a GPU: flat reads (FlatJob) and
raw FASTQ records (RecordsJob), compiled with g++ into a scratch library.
    python tools/hostpack_rate.py [n_reads=10000000] [threads=1,2,4,8,16,24,32]
Prints GB/s of input bytes and the k-mers/s they stand for (150 bp reads, k = 31)."""
import ctypes
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = r'''
#include "kmm_hostpack.hpp"
static kmm_hostpack::Workers *g_pool = nullptr;
extern "C" void shim_pool(int threads) { delete g_pool; g_pool = new kmm_hostpack::Workers(threads); }
extern "C" int shim_flat(const uint8_t *s, size_t n, uint8_t *d) {
    kmm_hostpack::FlatJob job; job.prepare(s, n, d, (size_t)4 << 20);
    g_pool->start([&job](int) { job.run(); }); g_pool->wait(); return job.bad.load() ? 0 : 1; }
extern "C" int shim_records(const uint8_t *raw, size_t n, int period, uint64_t *codes, uint32_t *bits, int64_t *out) {
    kmm_hostpack::RecordsJob job; job.prepare(raw, n, period, codes, bits);
    g_pool->start([&job](int) { job.run(); }); g_pool->wait();
    const kmm_hostpack::RecordsResult r = job.finish();
    out[0] = r.ok; out[1] = r.consumed; out[2] = r.n_records; out[3] = r.n_bases; out[4] = r.uniform_len; return 0; }
extern "C" int shim_budget() { return kmm_hostpack::cpu_budget(); }
extern "C" int shim_isa() { return (int)kmm_hostpack::isa(); }
'''


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    threads = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,16,24,32").split(",")]
    L, k = 150, 31
    tmp = tempfile.mkdtemp()
    src = os.path.join(tmp, "shim.cpp")
    open(src, "w").write(SHIM)
    so = os.path.join(tmp, "shim.so")
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I" + os.path.join(ROOT, "kmer_mapper_amd", "csrc"), src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.shim_flat.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.shim_records.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    print("cpu budget %d, instruction set %s" % (lib.shim_budget(), ("scalar", "AVX2", "AVX-512 VBMI")[lib.shim_isa()]), flush=True)
    rng = np.random.default_rng(1)
    seq = rng.integers(0, 4, size=(n_reads, L), dtype=np.uint8)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[seq]
    rec = np.empty((n_reads, 4 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :4] = np.frombuffer(b"@rd\n", dtype=np.uint8)
    rec[:, 4:4 + L] = seq
    rec[:, 4 + L:4 + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 4 + L + 3:4 + L + 3 + L] = ord("F")
    rec[:, -1] = 10
    raw = rec.reshape(-1)
    flat = np.ascontiguousarray(seq.reshape(-1))
    del rec, seq
    n = raw.size
    codes = np.empty(n // 32 + 80, dtype=np.uint64)
    bits = np.empty(n // 32 + 20, dtype=np.uint32)
    dst = np.empty(flat.size // 4 + 64, dtype=np.uint8)
    out = np.zeros(8, dtype=np.int64)
    kmers = n_reads * (L - k + 1)
    only = os.environ.get("HOSTPACK_RATE_ONLY", "")
    for t in threads:
        lib.shim_pool(t)
        best_f = best_r = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            ok = lib.shim_flat(flat.ctypes.data, flat.size, dst.ctypes.data) if only != "records" else 1
            best_f = min(best_f, time.perf_counter() - t0) if only != "records" else 1e9
            assert ok == 1
            t0 = time.perf_counter()
            lib.shim_records(raw.ctypes.data, n, 4, codes.ctypes.data, bits.ctypes.data, out.ctypes.data)
            best_r = min(best_r, time.perf_counter() - t0)
            assert tuple(out[:5]) == (1, n, n_reads, n_reads * L, L), out
        print("threads %3d: flat reads %6.1f GB/s (%6.1f G k-mers/s)   raw FASTQ %6.1f GB/s (%6.1f G k-mers/s)"
              % (t, flat.size / best_f / 1e9, kmers / best_f / 1e9, n / best_r / 1e9, kmers / best_r / 1e9), flush=True)


if __name__ == "__main__":
    main()
