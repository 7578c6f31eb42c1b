// hw_tiles.hpp -- the tiles of isocon_hw_pairs built on the device.  A tile = one query x up to 64 of its targets (hw.hpp); the pairs of a
// call (millions for the candidate-vs-candidate graph, /root/reference/modules/end_invariant_functions.py:622-681) are bucketed by
// (query, words their own band needs) with a counting sort -- histogram, one scan, scatter -- every bucket cut into runs of 64, and every
// run filed under the class (1, 2, 4, 8 words) its COMMON window needs.  Order inside a bucket is whatever the atomics give: a pair's
// result does not depend on which lanes share its tile.
#pragma once
#include "common.hpp"
#include "hw_core.hpp"

namespace isocon {

enum : uint32_t { HWT_ERR_INDEX = 1u, HWT_ERR_K_NEGATIVE = 2u, HWT_ERR_K_LARGE = 4u, HWT_ERR_BAND = 8u, HWT_ERR_STATUS = 16u, HWT_ERR_TILE_BAND = 32u };
static constexpr uint32_t HWT_NONE = 0xffffffffu;

// counters of one stage, zeroed by the host: [0] error flags, [1] tiles in all, [2..5] tiles per class
struct HwTileCounters { uint32_t flags, n_tiles, cls[4], pad[2]; };

ISO_HD int hwt_words(int32_t rows) { return rows <= 64 ? 1 : rows <= 128 ? 2 : rows <= 256 ? 4 : rows <= 512 ? 8 : 0; }
ISO_HD uint32_t hwt_class(int words) { return words == 1 ? 0u : words == 2 ? 1u : words == 4 ? 2u : 3u; }

// one count per distinct key of the wave: lanes holding the same key are counted by their lowest lane.  Returns the lane's rank among the
// lanes of its key and (in `leader_base`, valid for every lane of the key) the value the counter had before.
__device__ __forceinline__ uint32_t hwt_wave_add(uint32_t *__restrict__ counters, uint32_t key, uint32_t &leader_base)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t rank = 0;
    leader_base = 0;
    uint64_t todo = __ballot(key != HWT_NONE);
    while (todo) {
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
        const uint64_t same = __ballot(key == k0) & todo;
        uint32_t base = 0;
        if (lane == first) base = atomicAdd(counters + k0, (uint32_t)__builtin_popcountll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
        if (key == k0) { leader_base = base; rank = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull)); }
        todo &= ~same;
    }
    return rank;
}

// STAGE 0 (LOCATE): every pair that needs a kernel gets key = 4 query + class of its own band; the others (distance > k by the lengths
// alone, an empty sequence) none; out5 / he get their "no hit" rows.  STAGE 1 (START + TRACE): the hits of he.
template <int STAGE>
__global__ __launch_bounds__(256) void k_hwt_keys(DevStore S, const uint32_t *__restrict__ q, const uint32_t *__restrict__ t, const int32_t *__restrict__ k,
                                                   unsigned long long n_pairs, int32_t *__restrict__ he, int32_t *__restrict__ out5, uint32_t *__restrict__ key,
                                                   uint32_t *__restrict__ hist, HwTileCounters *__restrict__ ctr)
{
    const unsigned long long p = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    uint32_t kk = HWT_NONE;
    if (p < n_pairs) {
        const uint32_t qq = q[p], tt = t[p];
        const int32_t kv = k[p];
        if (STAGE == 0) {
            uint32_t err = 0;
            if (qq >= S.n || tt >= S.n) err = HWT_ERR_INDEX;
            else if (kv < 0) err = HWT_ERR_K_NEGATIVE;
            else if (kv > (1 << 20)) err = HWT_ERR_K_LARGE;
            else {
                const int32_t lq = S.lens[qq], lt = S.lens[tt], delta = lt - lq;
                if (!(delta < -kv || lq == 0 || lt == 0)) {
                    const int w = hwt_words(hw_locate_rows(delta, kv));
                    if (!w) err = HWT_ERR_BAND;
                    else kk = qq * 4u + hwt_class(w);
                }
            }
            if (err) atomicOr(&ctr->flags, err);
            he[p * 2] = -1; he[p * 2 + 1] = -1;
            int32_t *o = out5 + p * 5;
            o[0] = -1; o[1] = -1; o[2] = -1; o[3] = 0; o[4] = 0;
        } else {
            const int32_t h = he[p * 2];
            if (h < -1) atomicOr(&ctr->flags, (uint32_t)HWT_ERR_STATUS);
            else if (h >= 0) kk = qq * 4u + hwt_class(hwt_words(2 * kv + 1));
        }
        key[p] = kk;
    }
    uint32_t base;
    (void)hwt_wave_add(hist, kk, base);
}

// tile_base[key] = tiles before the key's first (a key with c pairs has ceil(c / 64) tiles), tile_base[n_keys] = tiles in all: the host's
// device_exscan<6> over the histogram.

// lane_pair (pre-set to "empty") and tile_q
__global__ __launch_bounds__(256) void k_hwt_scatter(const uint32_t *__restrict__ key, unsigned long long n_pairs, const uint32_t *__restrict__ tile_base,
                                                      uint32_t *__restrict__ cursor, uint32_t *__restrict__ tile_q, uint32_t *__restrict__ lane_pair)
{
    const unsigned long long p = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    const uint32_t kk = p < n_pairs ? key[p] : HWT_NONE;
    uint32_t base;
    const uint32_t rank = hwt_wave_add(cursor, kk, base);
    if (kk == HWT_NONE) return;
    const uint32_t pos = base + rank, tile = tile_base[kk] + (pos >> 6);
    lane_pair[(size_t)tile * 64 + (pos & 63u)] = (uint32_t)p;
    if ((pos & 63u) == 0) tile_q[tile] = kk >> 2;
}

// the class of every tile from its lanes' extremes, tiles appended to their class's list (cls_tiles[class * max_tiles + i])
template <int STAGE>
__global__ __launch_bounds__(256) void k_hwt_classes(DevStore S, const uint32_t *__restrict__ tile_q, const uint32_t *__restrict__ lane_pair,
                                                      const uint32_t *__restrict__ t, const int32_t *__restrict__ k, uint32_t max_tiles,
                                                      const uint32_t *__restrict__ n_tiles, uint32_t *__restrict__ cls_tiles, HwTileCounters *__restrict__ ctr)
{
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (tile >= *n_tiles) return;
    const uint32_t pair = lane_pair[(size_t)tile * 64 + lane];
    const bool has = pair != HWT_NONE;
    const int32_t P = S.lens[tile_q[tile]];
    const int32_t kv = has ? k[pair] : 0;
    const int32_t delta = has ? S.lens[t[pair]] - P : -(1 << 30);
    const int32_t dmax = wave_max_i32(delta), kmax = wave_max_i32(kv);
    const int w = hwt_words(STAGE == 0 ? hw_locate_rows(dmax, kmax) : 2 * kmax + 1);
    if (lane == 0) {
        if (!w) atomicOr(&ctr->flags, (uint32_t)HWT_ERR_TILE_BAND);
        else {
            const uint32_t c = hwt_class(w);
            cls_tiles[(size_t)c * max_tiles + atomicAdd(&ctr->cls[c], 1u)] = tile;
        }
    }
}

}  // namespace isocon
