// finalize_host.cpp -- the host-side CSR assembly of libisocon_hip.so (isocon_amd/csrc/nn_finalize_host.hpp, what isocon_nn_finalize
// runs) compiled for the CPU box, plain or with -fsanitize=address,undefined (tests/test_finalize_host.py).  Same signature as the
// C ABI's isocon_nn_finalize (include/isocon_hip.h).
#include "../../isocon_amd/csrc/nn_finalize_host.hpp"

extern "C" int emul_nn_finalize(uint32_t n, const int32_t *best, const int32_t *hits, uint64_t n_hits, int32_t *out_best,
                                uint64_t *out_row_ptr, uint32_t *out_cols, uint64_t cols_cap, uint64_t *n_cols_needed)
{
    if ((n && (!best || !out_best)) || !out_row_ptr || (n_hits && !hits) || (cols_cap && !out_cols)) return ISOCON_E_ARG;
    return nn_finalize_impl(n, best, hits, n_hits, out_best, out_row_ptr, out_cols, cols_cap, n_cols_needed);
}
