// nn_finalize.hpp -- the CSR of the nearest-neighbour graph on the device: the hits that attain their endpoint's minimum, bucketed by
// endpoint, every row in the reference's insertion order (ascending offset, the lower neighbour before the upper:
// /root/reference/modules/nearest_neighbor_graph.py:155-178), duplicates dropped.  Same result as the host routine nn_finalize_impl
// (nn_finalize_host.hpp), which stays for small inputs, for rows longer than NN_FIN_MAX_ROW and as the checker of tests/.
#pragma once
#include "nn_list.hpp"

namespace isocon {

static constexpr uint32_t NN_FIN_MAX_ROW = 256;      // longer rows (a read with hundreds of equidistant neighbours): host routine

__device__ __forceinline__ bool nn_fin_valid(const int32_t *__restrict__ hits, unsigned long long h, const int32_t *__restrict__ best, uint32_t n,
                                             uint32_t &e, uint32_t &o)
{
    const int32_t a = hits[h * 3], b = hits[h * 3 + 1], d = hits[h * 3 + 2];
    if (a < 0 || b < 0 || (uint32_t)a >= n || (uint32_t)b >= n || d < 0 || d != best[a]) return false;
    e = (uint32_t)a; o = (uint32_t)b;
    return true;
}

__global__ __launch_bounds__(256) void k_fin_count(const int32_t *__restrict__ hits, unsigned long long n_hits, const int32_t *__restrict__ best, uint32_t n,
                                                    uint32_t *__restrict__ cnt)
{
    const unsigned long long h = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    uint32_t e, o;
    if (h < n_hits && nn_fin_valid(hits, h, best, n, e, o)) atomicAdd(cnt + e, 1u);
}

__global__ __launch_bounds__(256) void k_fin_scatter(const int32_t *__restrict__ hits, unsigned long long n_hits, const int32_t *__restrict__ best, uint32_t n,
                                                      const unsigned long long *__restrict__ start, uint32_t *__restrict__ cursor, uint32_t *__restrict__ nb)
{
    const unsigned long long h = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    uint32_t e, o;
    if (h < n_hits && nn_fin_valid(hits, h, best, n, e, o)) nb[start[e] + atomicAdd(cursor + e, 1u)] = o;
}

// one thread per row: insertion sort by (offset from the row's entry, index), duplicates dropped; len2[i] = entries kept
__global__ __launch_bounds__(256) void k_fin_rows(uint32_t n, const unsigned long long *__restrict__ start, uint32_t *__restrict__ nb, uint32_t *__restrict__ len2,
                                                   uint32_t *__restrict__ too_long)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t *b = nb + start[i];
    const uint32_t len = (uint32_t)(start[i + 1] - start[i]);
    if (len > NN_FIN_MAX_ROW) { atomicOr(too_long, 1u); len2[i] = len; return; }
    auto before = [i](uint32_t x, uint32_t y) {
        const uint32_t ox = x > i ? x - i : i - x, oy = y > i ? y - i : i - y;
        return ox != oy ? ox < oy : x < y;
    };
    for (uint32_t a = 1; a < len; ++a) {
        const uint32_t v = b[a];
        uint32_t c = a;
        while (c > 0 && before(v, b[c - 1])) { b[c] = b[c - 1]; --c; }
        b[c] = v;
    }
    uint32_t w = 0;
    for (uint32_t a = 0; a < len; ++a)
        if (a == 0 || b[a] != b[a - 1]) b[w++] = b[a];
    len2[i] = w;
}

__global__ __launch_bounds__(256) void k_fin_gather(uint32_t n, const unsigned long long *__restrict__ start, const uint32_t *__restrict__ nb,
                                                     const unsigned long long *__restrict__ row_ptr, uint32_t *__restrict__ cols)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t len = (uint32_t)(row_ptr[i + 1] - row_ptr[i]);
    const uint32_t *src = nb + start[i];
    uint32_t *dst = cols + row_ptr[i];
    for (uint32_t a = 0; a < len; ++a) dst[a] = src[a];
}

}  // namespace isocon
