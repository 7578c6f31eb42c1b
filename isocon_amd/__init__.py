"""MI355X-native drop-in for IsoCon's all-pairs alignment + nearest-neighbour-graph hot path.

Modules mirror the reference's (ksahlin/IsoCon v0.3.3, modules/*.py) names and signatures:
    nearest_neighbor_graph, edlib_alignment_module, SW_alignment_module, get_best_alignments,
    functions.filter_exon_differences (the step right after the path)
plus store (packed sequence set in HBM), dist (one-process-per-GPU sharding), synth (seeded test data) and _lib
(ctypes binding of libisocon_hip.so, C ABI in include/isocon_hip.h).  No CPU fallback.
"""
__all__ = ["nearest_neighbor_graph", "edlib_alignment_module", "SW_alignment_module", "get_best_alignments",
           "functions", "store", "dist", "synth"]
