"""kmer_mapper_amd — MI355X-native drop-in for kmer_mapper's k-mer extraction + index lookup path.

Host orchestration is Python/numpy; all work on the path runs in hand-written HIP kernels
(csrc/kmm.hip) reached through the C ABI of include/kmm.h via ctypes (_lib.py).  There is no CPU
fallback: importing the operators without libkmm.so, or calling them without a GPU, fails loudly.

Reference surface mirrored here (ivargr/kmer_mapper):
  mapper.map_kmers_to_graph_index / in_graph_index      kmer_mapper/mapper.pyx:19-72, :81-130
  util.get_kmer_hashes_from_chunk_sequence              kmer_mapper/util.py:71-75
  gpu_counter.GpuCounter                                kmer_mapper/gpu_counter.py:5-37
  command_line_interface.main / map_bnp / map_gpu       kmer_mapper/command_line_interface.py
"""
__version__ = "0.1.0"
