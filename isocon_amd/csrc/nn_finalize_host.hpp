// nn_finalize_host.hpp -- host-side CSR assembly of a nearest-neighbour graph from reduced bounds and hit triples (isocon_nn_finalize;
// the fallback of the device routine in nn_finalize.hpp).  Plain C++: no HIP types, so that tests/emul/finalize_host.cpp can compile it
// with g++ -fsanitize=address,undefined on the CPU box (tests/test_finalize_host.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/isocon_hip.h"

namespace {

inline int nn_finalize_impl(uint32_t n, const int32_t *best, const int32_t *hits, uint64_t n_hits, int32_t *out_best,
                     uint64_t *out_row_ptr, uint32_t *out_cols, uint64_t cols_cap, uint64_t *n_cols_needed)
{
    // Bucket the attaining hits by endpoint (counting sort: rows hold one or two edges on average), then order every
    // row the way the reference inserts (NNG:155-178): ascending offset j, the lower neighbour (i-j) before the upper
    // (i+j); duplicates (a pair reported by more than one phase / rank) dropped.
    std::vector<uint64_t> start((size_t)n + 1, 0);
    std::vector<uint64_t> keep;              // endpoint << 32 | neighbour of the hits that attain their endpoint's minimum
    keep.reserve((size_t)std::min<uint64_t>(n_hits, (uint64_t)4 * n + 1024));
    for (uint64_t h = 0; h < n_hits; ++h) {
        const int32_t e = hits[h * 3], o = hits[h * 3 + 1], d = hits[h * 3 + 2];
        if (e < 0 || o < 0 || (uint32_t)e >= n || (uint32_t)o >= n || d < 0 || d != best[e]) continue;
        ++start[(size_t)e + 1];
        keep.push_back(((uint64_t)(uint32_t)e << 32) | (uint32_t)o);
    }
    for (uint32_t i = 0; i < n; ++i) start[i + 1] += start[i];
    std::vector<uint32_t> nb(start[n]);
    {
        std::vector<uint64_t> fill(start.begin(), start.end() - 1);
        for (const uint64_t k : keep) nb[fill[k >> 32]++] = (uint32_t)k;
    }
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t *b = nb.data() + start[i], *e = nb.data() + start[i + 1];
        const size_t len = (size_t)(e - b);
        auto before = [i](uint32_t x, uint32_t y) {
            const uint32_t ox = x > i ? x - i : i - x, oy = y > i ? y - i : i - y;
            return ox != oy ? ox < oy : x < y;
        };
        if (len == 2) {
            if (b[0] == b[1]) e = b + 1;
            else if (before(b[1], b[0])) std::swap(b[0], b[1]);
        } else if (len > 2) {
            if (len <= 16) {                 // insertion sort: rows are short
                for (size_t a = 1; a < len; ++a) {
                    const uint32_t v = b[a];
                    size_t c = a;
                    while (c > 0 && before(v, b[c - 1])) { b[c] = b[c - 1]; --c; }
                    b[c] = v;
                }
            } else std::sort(b, e, before);
            e = std::unique(b, e);
        }
        const uint64_t cnt = (uint64_t)(e - b);
        // compact in place: rows only shrink, so the write position never overtakes the read position
        if (cnt && nb.data() + total != b) std::copy(b, e, nb.data() + total);
        start[i] = total;
        total += cnt;
    }
    start[n] = total;
    if (n_cols_needed) *n_cols_needed = total;
    if (total > cols_cap) return ISOCON_E_CAPACITY;
    for (uint32_t i = 0; i < n; ++i) {
        out_row_ptr[i] = start[i];
        out_best[i] = start[i + 1] > start[i] ? best[i] : -1;
    }
    out_row_ptr[n] = total;
    if (total) memcpy(out_cols, nb.data(), total * sizeof(uint32_t));
    return ISOCON_OK;
}

}  // namespace
