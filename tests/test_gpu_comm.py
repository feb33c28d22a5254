"""The RCCL step behind the C ABI (include/kmm.h: kmm_reduce_counts, kmm_comm_*), replacing the additive reduce
of reference kmer_mapper/command_line_interface.py:124-130.  A 1-GPU box can only form a communicator of one
rank: that still loads RCCL, creates the communicator, runs ncclReduce / ncclAllReduce on the count vector in
place and synchronises; the N > 1 arithmetic (uint32 wrap-around sums) is covered by tests/test_distributed.py."""
import ctypes

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
        # single-process form: an array of handles (here of one)
        arr = (ctypes.c_void_p * 1)(dev._h)
        _lib.check(_lib.lib().kmm_reduce_counts(arr, 1, 0))
        assert np.array_equal(dev.get_node_counts(), expect)
        with pytest.raises(ValueError):
            _lib.check(_lib.lib().kmm_reduce_counts(arr, 1, 3))
