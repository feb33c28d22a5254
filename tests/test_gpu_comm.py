"""The RCCL step behind the C ABI (include/kmm.h: kmm_reduce_counts, kmm_comm_*), replacing the additive reduce
of reference kmer_mapper/command_line_interface.py:124-130.  A 1-GPU box can only form a communicator of one
rank: that still loads RCCL, creates the communicator, runs ncclReduce / ncclAllReduce on the count vector in
place and synchronises; the N > 1 arithmetic (uint32 wrap-around sums) is covered by tests/test_distributed.py."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_reduce_in_place(oracle):
    from kmer_mapper_amd import _lib, synthetic as syn
    from kmer_mapper_amd.engine import DeviceIndex
    assert _lib.device_count() >= 1
    index, genome = syn.make_index(5000, seed=401)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 20000, 150, seed=402)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    with DeviceIndex.from_index(index, mx) as dev:
        uid = DeviceIndex.comm_unique_id()
        assert len(uid) == 128
        dev.comm_init(uid, 1, 0)
        for path in (1, 2):                       # radix path: pending per-entry hits are flushed before the reduce
            dev.reset()
            dev.set_param("path", path)
            dev.map_reads_uniform(bases, 20000, 150, 31)
            dev.comm_reduce_counts(root=0)
            assert np.array_equal(dev.get_node_counts(), expect)
            dev.comm_reduce_counts(root=-1)       # all-reduce form
            assert np.array_equal(dev.get_node_counts(), expect)
        # ONE copy of RCCL in the process: the library takes the one torch has mapped (csrc/kmm_comm.hpp: rccl_load)
        import torch  # noqa: F401  (libtorch_hip needs librccl)
        mapped = {line.split()[-1] for line in open("/proc/self/maps") if "librccl" in line}
        assert len({os.path.realpath(m) for m in mapped}) == 1, mapped
        # single-process form: an array of handles (here of one)
        arr = (ctypes.c_void_p * 1)(dev._h)
        _lib.check(_lib.lib().kmm_reduce_counts(arr, 1, 0))
        assert np.array_equal(dev.get_node_counts(), expect)
        with pytest.raises(ValueError):
            _lib.check(_lib.lib().kmm_reduce_counts(arr, 1, 3))


def test_flush_of_node_ranges_under_the_reduce_of_the_previous_range(oracle):
    """kmm_comm_reduce_counts on the radix path: the per-entry hits are flushed node range by node range (the entries
    are listed in node order) and every range's RCCL reduce is issued on a second stream behind its flush
    (`comm_overlap_slices`, default 8) — same counts as the plain flush + one reduce, for uniform nodes, for a skewed
    node distribution (many entries per node: the ranges are cut at node boundaries) and for slice counts that do not
    divide the vector; accumulated over two jobs on one handle (additivity of command_line_interface.py:124-130)."""
    from kmer_mapper_amd import synthetic as syn
    from kmer_mapper_amd.engine import DeviceIndex
    for skewed, n_index in ((False, 60_000), (True, 60_000)):
        index, genome = syn.make_index(n_index, seed=411, skewed=skewed)
        mx = index.max_node_id()
        bases, offs = syn.make_reads(genome, 30000, 150, seed=412)
        expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
        with DeviceIndex.from_index(index, mx) as dev:
            dev.comm_init(DeviceIndex.comm_unique_id(), 1, 0)
            dev.set_param("path", 2)
            for slices in (8, 1, 7, 64):
                dev.set_param("comm_overlap_slices", slices)
                assert dev.get_param("comm_overlap_slices") == slices
                dev.reset()
                dev.map_reads_uniform(bases, 30000, 150, 31)
                dev.comm_reduce_counts(root=0)
                assert np.array_equal(dev.get_node_counts(), expect), (skewed, slices)
                dev.map_reads_uniform(bases, 30000, 150, 31)        # a second job on top: counts accumulate
                dev.comm_reduce_counts(root=-1)
                assert np.array_equal(dev.get_node_counts(), 2 * expect), (skewed, slices)
            # A rank with no radix batch behind it (nothing mapped at all; only direct-kernel batches) must issue the SAME
            # sequence of reduces as the others: whether the reduce is sliced may not depend on what the rank has mapped.
            dev.set_param("comm_overlap_slices", 8)
            dev.reset()
            dev.comm_reduce_counts(root=0)
            assert not dev.get_node_counts().any()
            dev.set_param("path", 1)
            dev.map_reads_uniform(bases, 30000, 150, 31)
            dev.comm_reduce_counts(root=0)
            assert np.array_equal(dev.get_node_counts(), expect), skewed
            # ... nor on rank-local state: a handle that cannot flush by node range (here: the sorted flush switched off —
            # the same branch a rank takes whose node-ordered entry list did not fit its HBM; per-k-mer counting mode) issues
            # the SAME S range reduces as its peers, behind one full flush (one reduce of n against S of n / S
            # hangs the job)
            dev.set_param("path", 2)
            before = dev.get_param("comm_sliced_reduces")
            for knob in ("radix_sorted_flush", "count_kmers"):
                dev.set_param(knob, 0 if knob == "radix_sorted_flush" else 1)
                dev.reset()
                dev.map_reads_uniform(bases, 30000, 150, 31)
                dev.comm_reduce_counts(root=0)
                assert np.array_equal(dev.get_node_counts(), expect), (skewed, knob)
                dev.set_param(knob, 1 if knob == "radix_sorted_flush" else 0)
            # (the count vector of the skewed index — 1000 nodes — is too short to be cut: one reduce on every rank)
            assert dev.get_param("comm_sliced_reduces") == before + (2 if mx + 1 >= 8 * 1024 else 0)


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_multi_rank_cli_flow_with_the_hip_engine(n_ranks, tmp_path):
    """The N > 1 flow end to end with the HIP engine on every rank (they share the box's one GPU, the sum of the count
    vectors runs over gloo): byte-range sharding of a FASTQ, chunk round-robin of its .gz copy and of a two-line FASTA .gz, member
    ranges of BGZF copies of both (inflated on the GPU, every rank its own range), one reduce, rank 0
    compares with the oracle (tools/cli_two_rank_rehearsal.py).  Replaces the reference's process pool + additive
    reduce (kmer_mapper/command_line_interface.py:109-130)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KMM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29600 + n_ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "cli_two_rank_rehearsal.py")]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    # .fq (byte ranges), .fq.gz and .fa.gz (plain gzip: chunk i to rank i mod N), BGZF .fq.gz and .fa.gz (member ranges, GPU inflate)
    assert out.count("BIT-EXACT") == 5 and "MISMATCH" not in out, out[-2000:]


def _bench_line(args, n_ranks=1, port=29655):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    if n_ranks == 1:
        cmd = [sys.executable, bench] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port), bench] + args
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=500,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line: %r" % r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_has_what_the_contract_names():
    """`python bench.py` at a reduced size: one JSON line with the metric, the roofline object (bound, achieved, peak,
    frac, per-kernel times from HIP events), the CPU baseline leg of the oracle on the same reads (parity checked in the
    same run), and the staged figure."""
    j = _bench_line(["--steps", "3", "--warmup", "1", "--index-kmers", "2000000", "--reads", "600000",
                     "--cpu-sample-reads", "100000"])
    assert j["unit"] == "M k-mers/s" and j["n_gpus"] == 1 and j["steps"] == 3 and j["higher_is_better"] is True
    assert j["value"] > 0 and j["dtype"] == "u64" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert abs(j["value"] - 3 * 600000 * 120 / (j["ms_per_step"] * 3 / 1e3) / 1e6) / j["value"] < 0.02
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = j["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert j["parity_vs_oracle_on_sample"] is True
    assert j["config"]["reads_in_hbm_when_timed"] is True and j["value_incl_h2d"] > 0
    # SURVEY 8(d)'s map-phase figure: by default the reads are packed by the rank's host threads (thread count stated); the
    # plain copy of the ASCII bytes beside it; the count vector's way back to the host; raw FASTQ from host memory
    assert j["value_incl_h2d_plain_copy"] > 0 and "ASCII" in j["config"]["h2d_leg_plain_copy"]["what"]
    if j["config"]["host_cores_of_this_rank"] >= 8:
        assert j["value_incl_h2d_host_packed"] == j["value_incl_h2d"] and "host threads" in j["config"]["h2d_leg"]["what"]
        assert j["config"]["h2d_leg"]["host_pack_threads"] == min(16, j["config"]["host_cores_of_this_rank"])
        rec = j["config"]["records_from_host_memory"]
        assert rec["M_kmers_per_s"] > 0 and rec["host_packed_calls"] >= 2
    assert j["config"]["final_d2h_ms"] > 0 and "numa_node" in j["config"]["numa_binding"]


def test_bench_line_of_two_ranks_carries_the_strong_and_the_staged_leg():
    """The driver's N > 1 invocation (`torch.distributed.run ... bench.py --gpus N --steps K --warmup W`) rehearsed with
    two ranks on this box's one GPU, the reduce over gloo: weak `value` over both ranks, the configs[3] `strong` leg
    (the same reads as ONE job split over the ranks: map, flush, reduce) and the staged leg on every rank."""
    j = _bench_line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--index-kmers", "2000000", "--reads", "600000",
                     "--dist-backend", "gloo"], n_ranks=2)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert abs(j["value"] - 2 * 3 * 600000 * 120 / (j["ms_per_step"] * 3 / 1e3) / 1e6) / j["value"] < 0.02
    assert len(j["config"]["per_rank_map_ms"]) == 2
    s = j["strong"]
    assert s["value"] > 0 and len(s["per_rank_map_ms"]) == 2 and len(s["flush_ms"]) == 2 and s["final_reduce_ms"] >= 0
    assert 0 < s["efficiency_vs_n1"] <= 1.5
    h = j["config"]["h2d_leg"]
    assert len(h["per_rank_read_GB_per_s"]) == 2 and j["value_incl_h2d"] > 0
    assert j["value_incl_h2d_plain_copy"] > 0 and len(j["config"]["h2d_leg_plain_copy"]["per_rank_read_GB_per_s"]) == 2
    # every record of a scaling run stands alone: rank 0 times the CPU baseline and checks parity at N > 1 too
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] >= 1 and j["parity_vs_oracle_on_sample"] is True
    assert "rehearsal" in j["config"]["parallelism"]          # (which reduce ran, and why)
