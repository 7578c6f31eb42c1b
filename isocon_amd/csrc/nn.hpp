// nn.hpp -- nearest-neighbour search kernels built on the banded edit-distance tile.
//
// Replaces the adaptive loops of /root/reference/modules/nearest_neighbor_graph.py:110-198 (1-set) and :341-424
// (2-set).  SURVEY.md App. C: for every query the loop's result is the arg-min SET of edit distances over the
// admissible neighbours (restricted to d <= len(query)), so any evaluation order / pruning that certifies the
// minimum reproduces it.  Here:
//   * best[i] holds the smallest distance found so far for query i (atomicMin, agent scope);
//   * a pair (s, p) is evaluated once with threshold k = min(kcap, max(k_s, k_p)), k_x = min(best[x], len(x)) for
//     every endpoint x that acts as a query against the other endpoint (both in the 1-set graph: distances are
//     symmetric, so each unordered pair is aligned once and scattered to both rows);
//   * every result d <= best[x] (non-strict, evaluated atomically) is appended to the hit list; the final
//     arg-min set of x is { hits of x with d == final best[x] }.
#pragma once
#include "ed_band.hpp"

namespace isocon {

static constexpr int32_t NN_INF = 0x3fffffff;
static constexpr int NN_COUNTER_SLOTS = 32;   // counters are striped to keep atomics off one address

struct NNParams {
    int32_t *best;
    const uint8_t *qflag;          // 1: entry acts as a query
    const uint8_t *tflag;          // 1: entry acts as a neighbour candidate
    int32_t *hits;                 // (endpoint, neighbour, distance) triples; distance -2 = pair needs a re-run
    unsigned long long *hit_count; // single counter
    unsigned long long *stats;     // [NN_COUNTER_SLOTS][4]: pairs, columns, tiles, spare
    uint64_t hits_cap;
    int32_t kcap;
    int32_t min_d;                 // 1: only positive distances admitted (1-set, NNG:157), 0: 2-set (NNG:388)
    uint32_t depth;                // largest admissible sorted-order offset (NNG:190)
};

__device__ __forceinline__ void nn_append(const NNParams &P, bool want, int32_t e, int32_t o, int32_t d)
{
    const unsigned long long mask = __ballot(want);
    if (!mask) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(P.hit_count, (unsigned long long)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (want) {
        const unsigned long long idx = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        if (idx < P.hits_cap) {
            int32_t *h = P.hits + idx * 3;
            h[0] = e; h[1] = o; h[2] = d;
        }
    }
}

struct WaveAcc { unsigned long long pairs, cols, tiles, live; };

// One tile: shared entry s (wave-uniform) against the lane entries p (p < 0 = empty lane).
template <int W, bool LDS = false>
__device__ __forceinline__ void nn_process_tile(const DevStore &S, const NNParams &P, uint32_t s, int32_t m,
                                                bool s_isq, bool s_ist, int64_t p, WaveAcc &acc, const uint4 *tab = nullptr)
{
    const bool valid = p >= 0 && p < (int64_t)S.n && (uint32_t)p != s;
    const uint32_t tid = valid ? (uint32_t)p : s;
    const int32_t n_t = S.lens[tid];
    const bool upd_s = valid && s_isq && P.tflag[tid];
    const bool upd_l = valid && s_ist && P.qflag[tid];
    int32_t bs = NN_INF;
    if (s_isq) bs = uniform_i32(load_relaxed_agent(P.best + s));
    int32_t ks = -1, kl = -1;
    if (upd_s) ks = bs < m ? bs : m;
    if (upd_l) { const int32_t bl = load_relaxed_agent(P.best + tid); kl = bl < n_t ? bl : n_t; }
    int32_t k = ks > kl ? ks : kl;
    if (k > P.kcap) k = P.kcap;
    // A tile whose lanes differ much in length cannot certify every lane with ONE common window (result -2 for
    // the lanes with the smaller |length difference|).  Those lanes are simply run again among themselves: the lane
    // that fixes the window origin is always certified, so every round retires at least one lane.
    int32_t r = -1;
    bool pending = valid && k >= 0;
    for (int round = 0; round < 64; ++round) {
        TileStats st;
        const int32_t rr = band_tile_run<W, LDS>(S, s, m, tid, n_t, k, pending, &st, tab);
        acc.pairs += round == 0 ? st.lanes_run : 0;
        acc.cols += (unsigned long long)st.lanes_run * st.cols;
        acc.live += st.live_cols;
        if (pending) r = rr;
        pending = pending && rr == -2;
        if (__ballot(pending) == 0) break;
    }
    acc.tiles += 1;
    bool hit_s = false, hit_l = false;
    if (r >= P.min_d) {
        if (upd_s && r <= m) { const int32_t old = atomicMin(P.best + s, r); hit_s = r <= old; }
        if (upd_l && r <= n_t) { const int32_t old = atomicMin(P.best + tid, r); hit_l = r <= old; }
    }
    nn_append(P, hit_s, (int32_t)s, (int32_t)tid, r);
    nn_append(P, hit_l, (int32_t)tid, (int32_t)s, r);
    nn_append(P, r == -2, (int32_t)s, (int32_t)tid, -2);
}

__device__ __forceinline__ void nn_flush_acc(const NNParams &P, const WaveAcc &acc)
{
    if ((threadIdx.x & 63) == 0 && acc.tiles) {
        unsigned long long *c = P.stats + (size_t)(blockIdx.x % NN_COUNTER_SLOTS) * 4;
        atomicAdd(c + 0, acc.pairs);
        atomicAdd(c + 1, acc.cols);
        atomicAdd(c + 2, acc.tiles);
        atomicAdd(c + 3, acc.live);
    }
}

// Implicit upward scan (64*W-row band, scalar-unit window): shared entry q against q+1+64*tile+lane for tile in
// [tile_begin, tile_end), while the length difference stays within kcap.  wpq waves cooperate on one q (1, 2 or 4).
// W = 1: seed pass.  W = 2, 4, 8: the wide-band phase -- only entries flagged as (still unresolved) queries make a
// pair active, every other tile is skipped after its role/threshold loads.
template <int W>
__global__ __launch_bounds__(256) void k_nn_scan_up(DevStore S, NNParams P, uint32_t q_begin, uint32_t q_end, uint32_t q_stride,
                                                     int32_t tile_begin, int32_t tile_end, int32_t wpq)
{
    const int32_t wave = threadIdx.x >> 6;
    const int32_t lane = threadIdx.x & 63;
    const uint64_t q64 = (uint64_t)q_begin + (uint64_t)((blockIdx.x * 4u + (uint32_t)wave) / (uint32_t)wpq) * q_stride;
    const uint32_t q = q64 < (uint64_t)q_end ? (uint32_t)q64 : q_end;
    WaveAcc acc = {0, 0, 0, 0};
    if (q < q_end) {
        const int32_t m = S.lens[q];
        const bool q_isq = P.qflag[q] != 0, q_ist = P.tflag[q] != 0;
        for (int32_t tile = tile_begin + (wave % wpq); tile < tile_end; tile += wpq) {
            const int64_t t0 = (int64_t)q + 1 + (int64_t)tile * 64;
            if (t0 >= (int64_t)S.n || t0 - (int64_t)q > (int64_t)P.depth) break;
            if (S.lens[t0] - m > P.kcap) break;       // lengths ascend: nothing further can be within kcap
            int64_t p = t0 + lane;
            if (p - (int64_t)q > (int64_t)P.depth) p = -1;
            nn_process_tile<W>(S, P, q, m, q_isq, q_ist, p, acc);
        }
    }
    nn_flush_acc(P, acc);
}

// Same scan with the query's window table in LDS (see band_tile_run<1, true>): one workgroup of NWAVES waves per
// entry q; dynamic LDS = 16 B x (maxlen + 192).  This is the main-pass kernel of the 1-set search.
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_nn_scan_lds(DevStore S, NNParams P, uint32_t q_begin, uint32_t q_end, uint32_t q_stride,
                                                              int32_t tile_begin, int32_t tile_end)
{
    extern __shared__ uint4 wtab[];
    const int32_t wave = threadIdx.x >> 6;
    const int32_t lane = threadIdx.x & 63;
    const uint64_t q64 = (uint64_t)q_begin + (uint64_t)blockIdx.x * q_stride;
    const uint32_t q = q64 < (uint64_t)q_end ? (uint32_t)q64 : q_end;
    WaveAcc acc = {0, 0, 0, 0};
    if (q >= q_end) return;
    const int32_t m = S.lens[q];
    {   // anything to do at all?  (uniform)
        const int64_t t0 = (int64_t)q + 1 + (int64_t)tile_begin * 64;
        if (t0 >= (int64_t)S.n || t0 - (int64_t)q > (int64_t)P.depth || S.lens[t0] - m > P.kcap) return;
    }
    // match-mask table: entry e <-> bit offset o = e - 63 of the query; 32 rows o..o+31 per base, rows outside [0, m) = 0
    {
        const uint64_t *planes = S.planes;
        const uint32_t nseq = S.n;
        const int32_t nchunks = (int32_t)S.nchunks;
        auto chunk_lo = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + q) * 2] : 0; };
        auto chunk_hi = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + q) * 2 + 1] : 0; };
        const int32_t entries = m + 192;
        for (int32_t e = threadIdx.x; e < entries; e += NWAVES * 64) {
            const int32_t o = e - 63;
            const uint32_t lo = (uint32_t)stream64(chunk_lo, o), hi = (uint32_t)stream64(chunk_hi, o);
            const int32_t r0 = o < 0 ? -o : 0, r1 = (m - o) < 32 ? (m - o) : 32;       // valid rows [r0, r1)
            uint32_t v = 0;
            if (r1 > r0) v = (r1 >= 32 ? 0xffffffffu : ((1u << r1) - 1u)) & ~(r0 >= 32 ? 0xffffffffu : ((1u << r0) - 1u));
            uint4 w;
            w.x = ~lo & ~hi & v; w.y = lo & ~hi & v; w.z = ~lo & hi & v; w.w = lo & hi & v;
            wtab[e] = w;
        }
    }
    __syncthreads();
    const bool q_isq = P.qflag[q] != 0, q_ist = P.tflag[q] != 0;
    for (int32_t tile = tile_begin + wave; tile < tile_end; tile += NWAVES) {
        const int64_t t0 = (int64_t)q + 1 + (int64_t)tile * 64;
        if (t0 >= (int64_t)S.n || t0 - (int64_t)q > (int64_t)P.depth) break;
        if (S.lens[t0] - m > P.kcap) break;
        int64_t p = t0 + lane;
        if (p - (int64_t)q > (int64_t)P.depth) p = -1;
        nn_process_tile<1, true>(S, P, q, m, q_isq, q_ist, p, acc, wtab);
    }
    nn_flush_acc(P, acc);
}

// planes2[chunk][newpos] = planes[chunk][perm[newpos]]  (16 B per element; used to group similar sequences of equal length)
// il[chunk][id] = the 64 bases of planes[chunk][id] with the two code bits interleaved (x: bases 0..31, y: 32..63)
__device__ __forceinline__ uint64_t spread32(uint32_t v)
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

__global__ __launch_bounds__(256) void k_interleave_planes(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst, size_t total)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const ulonglong2 p = src[i];
        ulonglong2 o;
        o.x = spread32((uint32_t)p.x) | (spread32((uint32_t)p.y) << 1);
        o.y = spread32((uint32_t)(p.x >> 32)) | (spread32((uint32_t)(p.y >> 32)) << 1);
        dst[i] = o;
    }
}

__global__ __launch_bounds__(256) void k_permute_planes(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst,
                                                         const uint32_t *__restrict__ perm, uint32_t n, uint32_t nchunks)
{
    const size_t total = (size_t)n * nchunks;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t c = (uint32_t)(i / n), pos = (uint32_t)(i - (size_t)c * n);
        dst[i] = src[(size_t)c * n + perm[pos]];
    }
}

// Explicit tiles: shared entry tile_shared[t] against the entries lane_ids[64t + lane] (0xffffffff = empty lane).
template <int W>
__global__ __launch_bounds__(256) void k_nn_tiles(DevStore S, NNParams P, const uint32_t *__restrict__ tile_shared,
                                                   const uint32_t *__restrict__ lane_ids, uint32_t n_tiles)
{
    const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int32_t lane = threadIdx.x & 63;
    WaveAcc acc = {0, 0, 0, 0};
    if (t < n_tiles) {
        const uint32_t s = (uint32_t)uniform_i32((int32_t)tile_shared[t]);
        const int32_t m = S.lens[s];
        const uint32_t id = lane_ids[(size_t)t * 64 + lane];
        const int64_t p = id == 0xffffffffu ? -1 : (int64_t)id;
        nn_process_tile<W>(S, P, s, m, P.qflag[s] != 0, P.tflag[s] != 0, p, acc);
    }
    nn_flush_acc(P, acc);
}

}  // namespace isocon
