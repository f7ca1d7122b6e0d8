"""zvec_amd — MI355X (gfx950) native flat / IVF-Flat distance-scan core for zvec.

Package layout (only what the hot path needs):
  csrc/     hand-written HIP kernels + the C ABI (include/zvec_hip.h)
  _lib.py   ctypes binding of the C ABI
  index.py  host-side mirror of the reference's index-operator interface for this path
            (IndexSearcher / IndexStreamer / Context: set_topk, search_impl, result(i) ...)
  dist.py   one-process-per-GPU sharding + RCCL all-gather merge of the per-shard candidates
"""
from .index import (  # noqa: F401
    DocFilter, Gate, HipFlatSearcher, HipShardedIndex, HipFlatStreamer, HipIVFSearcher, HipIVFStreamer, IndexContext, IndexDocument, IndexError_,
    METRIC_L2, METRIC_IP, METRIC_COSINE, metric_from_name, shard_map, container_segments, open_flat_file, open_ivf_file,
)
