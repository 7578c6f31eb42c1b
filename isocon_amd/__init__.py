"""MI355X-native drop-in for IsoCon's all-pairs alignment + nearest-neighbour-graph hot path.

Modules mirror the reference's (ksahlin/IsoCon v0.3.3, modules/*.py) names and signatures:
    nearest_neighbor_graph, edlib_alignment_module, SW_alignment_module, get_best_alignments        (the hot path)
    functions, graphs, partitions, correction_module, isocon_get_candidates, end_invariant_functions, input_output,
    hypothesis_test_module, isocon_statistical_test, ccs_info        (its callers, SURVEY 8(f) f1-f4: both pipeline phases)
plus store (packed sequence set in HBM), dist (one-process-per-GPU sharding), synth (seeded test data) and _lib
(ctypes binding of libisocon_hip.so, C ABI in include/isocon_hip.h).  No CPU fallback.
"""
__all__ = ["nearest_neighbor_graph", "edlib_alignment_module", "SW_alignment_module", "get_best_alignments",
           "functions", "graphs", "partitions", "correction_module", "isocon_get_candidates", "end_invariant_functions",
           "hypothesis_test_module", "isocon_statistical_test", "ccs_info", "input_output", "store", "dist", "synth"]
