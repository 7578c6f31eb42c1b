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
#include "nn.hpp"

namespace isocon {

// bits i of a 32-bit word at stream offset o that lie inside [0, m)
__device__ __forceinline__ uint32_t lane_valid32(int32_t o, int32_t m)
{
    int32_t lo = -o, hi = m - o;
    if (lo < 0) lo = 0;
    if (hi > 32) hi = 32;
    if (hi <= lo) return 0u;
    const uint32_t upto_hi = hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u);
    return upto_hi & ~((1u << lo) - 1u);        // lo < hi <= 32, so lo < 32
}

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
    int32_t a0 = lane_emin(d, k < 0 ? 0 : k);
    if (a0 < -63) a0 = -63;
    const int32_t nv = -a0;
    int32_t bstar = d - a0;
    if (bstar < 0) bstar = 0;
    if (bstar > 63) bstar = 63;
    BandLane<1> L;
    band_init<1>(L, nv, bstar);

    const uint64_t *planes = S.planes;
    const uint32_t nseq = S.n;
    const int32_t nchunks = (int32_t)S.nchunks;
    auto x_lo = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + x) * 2] : 0; };
    auto x_hi = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + x) * 2 + 1] : 0; };
    auto y_lo = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + y) * 2] : 0; };
    auto y_hi = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + y) * 2 + 1] : 0; };

    // complemented pattern planes, bits [o, o + 96) of the stream: window bit b of column c0 + jj is register bit jj + b
    int32_t o = -nv;
    uint32_t L0, L1, L2, H0, H1, H2;
    {
        const uint64_t tl = ~stream64(x_lo, o), th = ~stream64(x_hi, o);
        L0 = (uint32_t)tl; L1 = (uint32_t)(tl >> 32); H0 = (uint32_t)th; H1 = (uint32_t)(th >> 32);
        L2 = (uint32_t)~stream64(x_lo, o + 64);
        H2 = (uint32_t)~stream64(x_hi, o + 64);
    }
    for (int32_t c0 = 0;; c0 += 32) {
        if (__ballot(run) == 0) break;
        const uint32_t wl = (uint32_t)stream64(y_lo, c0), wh = (uint32_t)stream64(y_hi, c0);       // 32 text bases
        const bool full = c0 + 32 <= n;
        if (o >= 0 && __ballot(run && !full) == 0) {
            // no virtual rows in any window of the block, 32 columns for every running lane
#pragma unroll
            for (int jj = 0; jj < 32; ++jj) {
                uint64_t NL[1], NH[1], VM[1] = {0};
                NL[0] = ((uint64_t)__builtin_amdgcn_alignbit(L2, L1, jj) << 32) | __builtin_amdgcn_alignbit(L1, L0, jj);
                NH[0] = ((uint64_t)__builtin_amdgcn_alignbit(H2, H1, jj) << 32) | __builtin_amdgcn_alignbit(H1, H0, jj);
                const uint32_t slo = (uint32_t)__builtin_amdgcn_sbfe((int)wl, jj, 1);
                const uint32_t shi = (uint32_t)__builtin_amdgcn_sbfe((int)wh, jj, 1);
                band_step<1, false>(L, NL, NH, VM, slo, shi);
            }
        } else {
            const uint32_t V0 = lane_valid32(o, m), V1 = lane_valid32(o + 32, m), V2 = lane_valid32(o + 64, m);
#pragma unroll 1
            for (int jj = 0; jj < 32; ++jj) {
                if (run && c0 + jj < n) {
                    uint64_t NL[1], NH[1], VM[1];
                    NL[0] = ((uint64_t)__builtin_amdgcn_alignbit(L2, L1, jj) << 32) | __builtin_amdgcn_alignbit(L1, L0, jj);
                    NH[0] = ((uint64_t)__builtin_amdgcn_alignbit(H2, H1, jj) << 32) | __builtin_amdgcn_alignbit(H1, H0, jj);
                    VM[0] = ((uint64_t)__builtin_amdgcn_alignbit(V2, V1, jj) << 32) | __builtin_amdgcn_alignbit(V1, V0, jj);
                    const uint32_t slo = (uint32_t)__builtin_amdgcn_sbfe((int)wl, jj, 1);
                    const uint32_t shi = (uint32_t)__builtin_amdgcn_sbfe((int)wh, jj, 1);
                    band_step<1, true>(L, NL, NH, VM, slo, shi);
                }
            }
        }
        if (run) {
            const bool fin = c0 + 32 >= n;
            const int32_t dv = band_diag_value<1>(L, nv, fin ? n : c0 + 32);
            if (fin) { r = dv <= k ? dv : -1; run = false; }
            else if (dv > k) run = false;                 // the value on the final diagonal never decreases
        }
        o += 32;
        L0 = L1; L1 = L2; H0 = H1; H1 = H2;
        L2 = (uint32_t)~stream64(x_lo, o + 64);
        H2 = (uint32_t)~stream64(x_hi, o + 64);
    }
    if (!NN) {
        if (i < n_pairs) out[i] = r;
        return;
    }
    bool hit_x = false, hit_y = false;
    if (r >= P.min_d) {
        if (upd_x && r <= m) { const int32_t old = atomicMin(P.best + x, r); hit_x = r <= old; }
        if (upd_y && r <= n) { const int32_t old = atomicMin(P.best + y, r); hit_y = r <= old; }
    }
    nn_append(P, hit_x, (int32_t)x, (int32_t)y, r);
    nn_append(P, hit_y, (int32_t)y, (int32_t)x, r);
}

}  // namespace isocon
