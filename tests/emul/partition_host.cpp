// partition_host.cpp -- isocon_partition_ids' host routine (isocon_amd/csrc/partition_host.hpp) compiled for the CPU box, plain or with
// -fsanitize=address,undefined (tests/test_partition_native.py).  Same signature as the C ABI's isocon_partition_ids.
#include "../../isocon_amd/csrc/partition_host.hpp"

extern "C" int emul_partition_ids(uint32_t n, const int32_t *degree, uint64_t n_edges, const uint32_t *edge_a, const uint32_t *edge_b,
                                  const uint32_t *rank, int32_t nbr_tiebreak, uint32_t *out_centre, int64_t *out_weight,
                                  uint64_t *out_member_ptr, uint32_t *out_members, uint32_t *n_parts)
{
    return partition_ids_impl(n, degree, n_edges, edge_a, edge_b, rank, nbr_tiebreak, out_centre, out_weight, out_member_ptr, out_members, n_parts);
}
