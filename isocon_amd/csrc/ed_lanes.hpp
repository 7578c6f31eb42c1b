// ed_lanes.hpp -- banded bit-vector edit distance with ONE PAIR PER LANE (64-row band, thresholds up to 63): nothing is shared
// between the lanes of a wave, so a list of unrelated pairs keeps all 64 lanes busy (the tile kernels of ed_band.hpp / nn.hpp
// want 64 partners of one shared sequence).  Same band math (band_core.hpp); the pattern window of a lane is cut out of
// 96-bit registers that hold its bit-planes for the 32 columns of a block (two v_alignbit per plane and column), the text
// base of a column is a sign-extended 1-bit field of the lane's own text words.
//
// Used by the nearest-neighbour search to turn the smallest q-gram bounds of every entry into exact distances before the
// main pass starts (tight thresholds from the first pair on), replacing edlib.align(x, y, "NW", "distance", k)
// (/root/reference/modules/nearest_neighbor_graph.py:104-107) like every other distance kernel here.
#pragma once
#include "band_core.hpp"
#include "common.hpp"
#include "ed_lanes_core.hpp"
#include "nn.hpp"

namespace isocon {

// Pairs (pa[i], pb[i]); 0xffffffff in either = no pair.
//   NN == false: out[i] = distance if <= pk[i] (0 <= pk[i] <= 63), else -1.
//   NN == true : the pair goes through the update rule of the nearest-neighbour search (roles, min_d, best[] by atomicMin,
//                hit list) with k = min(63, max of the two ends' current bounds), exactly like nn_process_tile.
template <bool NN>
__global__ __launch_bounds__(256) void k_ed_lanes(DevStore S, NNParams P, const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb,
                                                   const int32_t *__restrict__ pk, uint64_t n_pairs, int32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t x = 0, y = 0;
    bool valid = i < n_pairs;
    if (valid) {
        x = pa[i]; y = pb[i];
        valid = x != 0xffffffffu && y != 0xffffffffu && x < S.n && y < S.n && (!NN || x != y);
        if (!valid) { x = 0; y = 0; }
    }
    const int32_t m = S.lens[x], n = S.lens[y];
    bool upd_x = false, upd_y = false;
    int32_t k = -1;
    if (NN) {
        upd_x = valid && P.qflag[x] && P.tflag[y];
        upd_y = valid && P.qflag[y] && P.tflag[x];
        int32_t kx = -1, ky = -1;
        if (upd_x) { const int32_t b = load_relaxed_agent(P.best + x); kx = b < m ? b : m; }
        if (upd_y) { const int32_t b = load_relaxed_agent(P.best + y); ky = b < n ? b : n; }
        k = kx > ky ? kx : ky;
        if (k > 63) k = 63;
    } else if (valid) {
        k = pk[i];
        if (k > 63) k = 63;
    }
    const int32_t d = m - n, ad = d < 0 ? -d : d;
    bool run = valid && k >= 0 && ad <= k;
    int32_t r = -1;
    if (run && (m == 0 || n == 0)) { r = ad; run = false; }
    const uint64_t *planes = S.planes;
    const uint32_t nseq = S.n;
    const int32_t nchunks = (int32_t)S.nchunks;
    auto x_lo = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + x) * 2] : 0; };
    auto x_hi = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + x) * 2 + 1] : 0; };
    auto y_lo = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + y) * 2] : 0; };
    auto y_hi = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + y) * 2 + 1] : 0; };
    const int32_t rd = lane_pair_distance(x_lo, x_hi, y_lo, y_hi, m, n, k, run, [](bool b) { return __ballot(b) != 0; });
    if (run) r = rd;
    if (!NN) {
        if (i < n_pairs) out[i] = r;
        return;
    }
    bool hit_x = false, hit_y = false;
    if (r >= P.min_d) {
        if (upd_x && r <= m) hit_x = nn_take(P, x, r);
        if (upd_y && r <= n) hit_y = nn_take(P, y, r);
    }
    nn_append(P, hit_x, (int32_t)x, (int32_t)y, r);
    nn_append(P, hit_y, (int32_t)y, (int32_t)x, r);
}

}  // namespace isocon
