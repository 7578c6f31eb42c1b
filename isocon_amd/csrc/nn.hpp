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
    // k_nn_scan_refill, few queries ("sparse" launch): one workgroup per LISTED query, which takes its neighbours on BOTH
    // sides (a listed neighbour below it keeps the pair for its own workgroup); of those pairs a rank evaluates the ones
    // whose lower index it owns (own).  q_list == nullptr: one workgroup per entry, upward scan.
    const uint32_t *q_list;
    QMap own;
    // q-gram lower bounds of the main pass' pairs (qgram.hpp; nullptr = none): lb[lb_row[launch slot] + (p - q - 1)]
    const uint8_t *lb;
    const unsigned long long *lb_row;
    const uint32_t *slot_order;    // launch slot handled by workgroup i (nullptr: i itself): entries with the widest windows first
    uint32_t wg_base;              // k_nn_scan_refill: this launch covers the workgroups wg_base, wg_base + 1, ... of the pass (a long pass is cut into bounded launches)
    // k_nn_scan_refill, "listed" launch (nn_list.hpp; nullptr = not listed): workgroup i aligns the entry chunks[i].slot (its table in
    // LDS) with the partners list[chunks[i].begin .. + count) -- survivors of the q-gram bound, collected before the launch
    const struct NNChunk *chunks;
    const uint32_t *list;          // partner | 0x40000000 (the chunk's entry queries it) | 0x80000000 (it queries the chunk's entry)
    // 1: best[] holds FIXED thresholds -- every pair within its query's threshold is a hit and nothing is tightened (the collecting pass
    // over a set whose planes are class-merged images of the sequences, nn_images.inc: nn_phase_a_images)
    int32_t fixed;
};

struct NNChunk { uint32_t slot, count; unsigned long long begin; };

// The update rule for one end of a pair: distance r against the end's bound; true = the pair is a hit of that end.
__device__ __forceinline__ bool nn_take(const NNParams &P, uint32_t e, int32_t r)
{
    const int32_t old = P.fixed ? load_relaxed_agent(P.best + e) : atomicMin(P.best + e, r);
    return r <= old;
}

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
        if (upd_s && r <= m) hit_s = nn_take(P, s, r);
        if (upd_l && r <= n_t) hit_l = nn_take(P, tid, r);
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
__global__ __launch_bounds__(256) void k_nn_scan_up(DevStore S, NNParams P, QMap Q,
                                                     int32_t tile_begin, int32_t tile_end, int32_t wpq)
{
    const int32_t wave = threadIdx.x >> 6;
    const int32_t lane = threadIdx.x & 63;
    const uint32_t q_end = Q.end;
    const uint64_t q64 = Q.entry((blockIdx.x * 4u + (uint32_t)wave) / (uint32_t)wpq);
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
__global__ __launch_bounds__(NWAVES * 64) void k_nn_scan_lds(DevStore S, NNParams P, QMap Q,
                                                              int32_t tile_begin, int32_t tile_end)
{
    extern __shared__ uint4 wtab[];
    const int32_t wave = threadIdx.x >> 6;
    const int32_t lane = threadIdx.x & 63;
    const uint32_t q_end = Q.end;
    const uint64_t q64 = Q.entry(blockIdx.x);
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

// Keeps the hits that can still matter: distance <= best[endpoint] at the end of the launch -- with bounds that only decrease that is
// distance == best[endpoint]; with fixed thresholds (NNParams::fixed) every recorded hit -- and the -2 markers.  In place is not possible (unordered appends), so the survivors go to a second list.
__global__ __launch_bounds__(256) void k_filter_hits(const int32_t *__restrict__ hits, unsigned long long n_hits, const int32_t *__restrict__ best,
                                                      int32_t *__restrict__ out, unsigned long long *out_count)
{
    const int lane = threadIdx.x & 63;
    for (unsigned long long base = (unsigned long long)blockIdx.x * 256; base < n_hits; base += (unsigned long long)gridDim.x * 256) {
        const unsigned long long i = base + threadIdx.x;
        int32_t e = 0, o = 0, d = -1;
        bool keep = false;
        if (i < n_hits) {
            e = hits[i * 3]; o = hits[i * 3 + 1]; d = hits[i * 3 + 2];
            keep = d == -2 || (d >= 0 && d <= best[e]);
        }
        const unsigned long long mask = __ballot(keep);
        if (!mask) continue;
        const int leader = __ffsll((long long)mask) - 1;
        unsigned long long at = 0;
        if (lane == leader) at = atomicAdd(out_count, (unsigned long long)__popcll(mask));
        at = __shfl(at, leader, 64);
        if (keep) {
            int32_t *h = out + (at + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull))) * 3;
            h[0] = e; h[1] = o; h[2] = d;
        }
    }
}

// Main pass with LANE REFILL.  One workgroup per entry q with q's match masks tabulated in LDS, and the 64 lanes of a
// wave as independent pair processors: every lane carries its own neighbour, band origin, column position and table
// address, and a lane whose pair is decided (final-diagonal value above its threshold, or text exhausted) takes the
// next admissible neighbour from the wave's queue at the next 32-column boundary.  No lane waits for the slowest pair
// of a tile, no tile is half empty, and -- the band origin being per lane -- no pair ever needs a second run.
//
// LDS layout (dwords): four planes tw[b * E + e'] = 32-row match mask of base b at bit offset o = e' - (64 W + 31) of q
// (rows outside [0, m) read 0), then 128 W + 32 dwords of 0xffffffff (the "virtual column" plane, code 4).  A lane reads
// (e', code) and (e' + 32, code) with ONE ds_read2_b32.  ds_read_b32 banks are (address / 4) mod 32 per 32-lane half,
// so lanes at unrelated table positions would collide (measured: 17 LDS cycles per read instead of 4).  Therefore every
// lane keeps e' = lane (mod 32) at step 0 of every block: a new pair starts phi = (e0 - lane) mod 32 steps "early",
// and those phi steps are virtual text columns (code 4 -> all-ones Eq), which leave the initial band state untouched
// (D0 = ~0  =>  VP' = VP, VN' = VN) and only bump the top-row counter, pre-compensated by ztop = -phi.
//
// Lane texts come from the nibble store (4 bits per base, 8 bases per dword, row stride text_stride dwords per
// sequence, 32 virtual codes in front of every sequence, zeros behind it): a block is 5 consecutive dwords funnel-
// shifted by the lane's constant nibble phase.
//
// Neighbours are drawn 64 at a time from a workgroup-wide counter, filtered (roles, |length difference| <= threshold and,
// when the launch comes with them, the pair's q-gram lower bound <= threshold: qgram.hpp) with coalesced loads, and queued
// in a small per-wave ring in LDS.  With the bounds the host launches 4 waves per table instead of 8 (nn_main.inc) and may
// hand the workgroups their entries widest window first (P.slot_order).
static constexpr int NN_RING = 96;       // entries per wave (8 B each)
static constexpr int NN_TEXT_PAD_FRONT = 4, NN_TEXT_PAD_BACK = 6;     // dwords around each sequence of the nibble store

struct __attribute__((packed, aligned(4))) TextQuad { uint32_t x, y, z, w; };

// One kernel for every band width: W = 1 is the main pass (64 rows, thresholds up to 63), W = 2, 4, 8 the phase for
// entries whose nearest neighbour is further away (128 / 256 / 512 rows, ONT-like error rates).  A column's match
// vector is 2W dwords of the base's plane (entries e', e'+32, ..., W ds_read2_b32), the band state is W words with a
// carry chain (band_step_eq<W>), the final-diagonal mask is rebuilt from its bit index at the 32-column checks, and
// the column loop is fully unrolled for W = 1 (immediate table offsets) and unrolled by text dword (8 columns) for the
// wide bands to keep the code inside the instruction cache.  LDS: 4 planes x (m + 192 W) dwords + (128 W + 32) dwords
// of ones for the virtual columns.
template <int W>
__device__ __forceinline__ int32_t diag_value_w(const BandLane<W> &L, int32_t nv, int32_t cols, int32_t bstar)
{
    int32_t v = nv + cols - (int32_t)L.ztop;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int32_t b = bstar - 64 * i;
        const uint64_t lm = b <= 0 ? 0 : (b >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << b) - 1));
        v += popc64(L.VP[i] & lm) - popc64(L.VN[i] & lm);
    }
    return v;
}

// HALF (with W = 1): the 32-row band on 32-bit vectors for launches whose pairs all have thresholds <= 31 (the listed launch in its
// narrow mode, nn_list.hpp): one table dword and 12 instead of 22 vector instructions per column.
template <int NWAVES, int W, bool HALF = false>
__global__ __launch_bounds__(NWAVES * 64) void k_nn_scan_refill(DevStore S, NNParams P, const uint32_t *__restrict__ text, uint32_t text_stride,
                                                                      QMap Q, int32_t tile_begin)
{
    static_assert(W >= 1 && W <= 8, "band widths: 64 .. 512 rows");
    static_assert(!HALF || W == 1, "the 32-row band is the narrow form of the 64-row kernel");
    constexpr int ROWS = HALF ? 32 : 64 * W;
    constexpr int UNROLL_COLS = W >= 5 ? 1 : 8;      // 5 words and more: a real loop (VGPRs, instruction cache)
    // the next block's text is requested while this block computes -- except for the 512-row band, whose 16 waves per table leave 128 registers
    // a lane: five of them held a block ahead pushed the kernel into scratch (44 B); there the text is requested when the block is done
    constexpr bool PREFETCH_TEXT = W < 8;
    constexpr int RING = W < 8 ? NN_RING : 128;      // (a power of two there: the slot arithmetic needs no reciprocal held in a register)
    extern __shared__ uint32_t tw[];
    __shared__ uint32_t s_next;
    __shared__ uint32_t s_ring[NWAVES][RING][2];
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    const int32_t wave = threadIdx.x >> 6;
    const int32_t lane = threadIdx.x & 63;
    const bool sparse = P.q_list != nullptr;
    const bool listed = P.chunks != nullptr;
    uint32_t l_count = 0;
    unsigned long long l_begin = 0;
    const uint32_t wg = blockIdx.x + P.wg_base;
    if (listed) { l_count = P.chunks[wg].count; l_begin = P.chunks[wg].begin; }
    const uint32_t slot = listed ? P.chunks[wg].slot : (!sparse && P.slot_order != nullptr) ? P.slot_order[wg] : wg;
    const uint64_t q64 = listed ? (uint64_t)slot : sparse ? (uint64_t)P.q_list[wg] : Q.entry(slot);
    if (q64 >= (listed ? (uint64_t)S.n : (uint64_t)Q.end)) return;
    const uint32_t q = (uint32_t)q64;
    const int32_t m = S.lens[q];
    int64_t pbase = (int64_t)q + 1 + (int64_t)tile_begin * 64;
    if (sparse) {           // first entry whose length is within kcap below (lengths ascend): wave-uniform binary search
        uint32_t lo = 0, hi = q;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (uniform_i32(S.lens[mid]) < m - P.kcap) lo = mid + 1; else hi = mid;
        }
        pbase = lo;
    } else if (!listed && (pbase >= (int64_t)S.n || pbase - (int64_t)q > (int64_t)P.depth || S.lens[pbase] - m > P.kcap)) return;   // uniform
    const bool q_isq = P.qflag[q] != 0, q_ist = P.tflag[q] != 0;
    if (!q_isq && !q_ist) return;
    const bool bounded = P.lb != nullptr && !sparse && !listed;
    const unsigned long long lb_base = bounded ? P.lb_row[slot] : 0ull;
    const int32_t E = (m + 3 * ROWS + 31) & ~31;          // plane length in dwords
    {
        const uint64_t *planes = S.planes;
        const uint32_t nseq = S.n;
        const int32_t nchunks = (int32_t)S.nchunks;
        auto chunk_lo = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + q) * 2] : 0; };
        auto chunk_hi = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + q) * 2 + 1] : 0; };
        for (int32_t e = threadIdx.x; e < E; e += NWAVES * 64) {
            const int32_t o = e - (ROWS + 31);           // entry 32 + (ROWS - 1) <-> bit offset 0
            const uint32_t lo = (uint32_t)stream64(chunk_lo, o), hi = (uint32_t)stream64(chunk_hi, o);
            const int32_t r0 = o < 0 ? -o : 0, r1 = (m - o) < 32 ? (m - o) : 32;       // valid rows [r0, r1)
            uint32_t v = 0;
            if (r1 > r0) v = (r1 >= 32 ? 0xffffffffu : ((1u << r1) - 1u)) & ~(r0 >= 32 ? 0xffffffffu : ((1u << r0) - 1u));
            tw[e] = ~lo & ~hi & v;
            tw[E + e] = lo & ~hi & v;
            tw[2 * E + e] = ~lo & hi & v;
            tw[3 * E + e] = lo & hi & v;
        }
        for (int32_t e = threadIdx.x; e < 2 * ROWS + 32; e += NWAVES * 64) tw[4 * E + e] = 0xffffffffu;
        if (threadIdx.x == 0) s_next = 0;
    }
    __syncthreads();
    const uint32_t tbase0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t *)tw;
    const uint32_t idle_blk = tbase0 + (uint32_t)(lane & 31) * 4u;
    uint32_t plane_bytes;
    asm volatile("v_mov_b32 %0, %1" : "=v"(plane_bytes) : "s"((uint32_t)E * 4u));
    uint32_t(*ring)[2] = s_ring[wave];
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;

    bool run = false, upd_s = false, upd_l = false;
    uint32_t tid = q, blk = idle_blk, nsh = 0;
    int32_t n_t = 0, k_eff = -1, nv = 0, col = 0, bstar = 0;
    uint32_t cur[5] = {0, 0, 0, 0, 0};
    const uint32_t *tp = text;
    BandLane<W> L;
#pragma unroll
    for (int i = 0; i < W; ++i) { L.VP[i] = ~(uint64_t)0; L.VN[i] = 0; }
    L.ztop = 0;
    uint32_t hvp = ~0u, hvn = 0u;          // HALF: the band state
    uint32_t qhead = 0, qcount = 0;
    bool exhausted = false;
    uint32_t n_pairs = 0, n_batches = 0, n_blocks = 0, n_live = 0, n_filtered = 0;

    auto load5 = [](const uint32_t *p, uint32_t (&d)[5]) {
        const TextQuad t4 = *reinterpret_cast<const TextQuad *>(p);
        d[0] = t4.x; d[1] = t4.y; d[2] = t4.z; d[3] = t4.w; d[4] = p[4];
    };
    // the 2W dwords of one column: plane address a (entries e', e'+32, ..) -- second base register beyond 255 dwords
    auto fetch = [](uint32_t a, int off, uint64_t (&EQ)[W]) {
        lds_u32 *p0 = (lds_u32 *)(uintptr_t)a;
        lds_u32 *p1 = (lds_u32 *)(uintptr_t)(a + 1024u);
#pragma unroll
        for (int i = 0; i < W; ++i) {
            lds_u32 *pp = i < 4 ? p0 : p1;
            const int o = off + 64 * (i & 3);
            EQ[i] = ((uint64_t)pp[o + 32] << 32) | pp[o];
        }
    };

    for (;;) {
        const uint64_t freemask = __ballot(!run);
        const uint32_t nfree = (uint32_t)__popcll(freemask);
        while (!exhausted && qcount < nfree && qcount + 64 <= (uint32_t)RING) {
            uint32_t c0 = 0;
            if (lane == 0) c0 = atomicAdd(&s_next, 64u);
            c0 = (uint32_t)uniform_i32((int32_t)c0);
            int64_t p = pbase + (int64_t)c0 + lane;
            uint32_t pid;
            int32_t np;
            bool within, us, ul;
            if (listed) {
                // the survivors of the bound, already filtered by roles, window and bound (k_nn_survivors): one coalesced load
                const bool inr = c0 + (uint32_t)lane < l_count;
                const uint32_t e = inr ? P.list[l_begin + c0 + (uint32_t)lane] : q;
                pid = e & 0x3fffffffu;
                p = (int64_t)pid;
                np = S.lens[pid];
                within = inr;
                us = inr && (e & 0x40000000u) != 0;
                ul = inr && (e & 0x80000000u) != 0;
                if (c0 + 64u >= l_count) exhausted = true;
            } else {
            const int64_t off = p > (int64_t)q ? p - (int64_t)q : (int64_t)q - p;
            const bool inr = p < (int64_t)S.n && (off <= (int64_t)P.depth || (sparse && p < (int64_t)q));
            pid = inr ? (uint32_t)p : q;
            np = S.lens[pid];
            within = inr && np - m <= P.kcap;
            if (__ballot(within) != ~(uint64_t)0) exhausted = true;
            if (sparse) {
                // entries below q that are too far in the order (depth) are skipped, not the end of the window; the pair
                // belongs to this workgroup unless the neighbour is a listed query below q, and to this rank if it owns
                // the pair's lower index
                const uint32_t lowi = pid < q ? pid : q;
                const bool mine = pid != q && off <= (int64_t)P.depth && (pid > q || P.qflag[pid] == 0) &&
                                  P.own.owns(lowi);
                within = within && mine;
            }
            us = within && q_isq && P.tflag[pid];
            ul = within && q_ist && P.qflag[pid];
            }
            int32_t bs = NN_INF;
            if (q_isq) bs = uniform_i32(load_relaxed_agent(P.best + q));
            int32_t ks = -1, kl = -1;
            if (us) ks = bs < m ? bs : m;
            if (ul) { const int32_t bl = load_relaxed_agent(P.best + pid); kl = bl < np ? bl : np; }
            int32_t k = ks > kl ? ks : kl;
            if (k > P.kcap) k = P.kcap;
            const int32_t d = m - np, ad = d < 0 ? -d : d;
            bool accept = within && k >= 0 && ad <= k;
            if (bounded) {
                // the pair's q-gram bound proves d > k: exactly the pairs the band would have abandoned
                const bool cand = accept;
                if (accept) accept = (int32_t)P.lb[lb_base + (unsigned long long)(p - (int64_t)q - 1)] <= k;
                n_filtered += (uint32_t)__popcll(__ballot(cand && !accept));
            }
            const bool triv = accept && (m == 0 || np == 0);
            if (__ballot(triv) != 0) {
                bool hs = false, hl = false;
                if (triv && ad >= P.min_d) {
                    if (us && ad <= m) hs = nn_take(P, q, ad);
                    if (ul && ad <= np) hl = nn_take(P, pid, ad);
                }
                nn_append(P, hs, (int32_t)q, (int32_t)pid, ad);
                nn_append(P, hl, (int32_t)pid, (int32_t)q, ad);
                accept = accept && !triv;
            }
            const uint64_t am = __ballot(accept);
            if (accept) {
                const uint32_t slot = (qhead + qcount + (uint32_t)__popcll(am & lt_mask)) % (uint32_t)RING;
                int32_t a0 = lane_emin(d, k);
                if (a0 < -(ROWS - 1)) a0 = -(ROWS - 1);
                ring[slot][0] = pid | (us ? 0x40000000u : 0u) | (ul ? 0x80000000u : 0u);
                ring[slot][1] = (uint32_t)np | ((uint32_t)k << 14) | ((uint32_t)(-a0) << 23);      // 14 + 9 + 9 bits
            }
            qcount += (uint32_t)__popcll(am);
            n_pairs = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n_pairs + (uint32_t)__popcll(am)));          // (wave-uniform counters: kept in scalar registers)
            if (W < 8) n_batches += 1;          // (the 512-row form has no register to spare for a diagnostic counter)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (nfree && qcount) {
            const uint32_t rank = (uint32_t)__popcll(freemask & lt_mask);
            if (!run && rank < qcount) {
                const uint32_t slot = (qhead + rank) % (uint32_t)RING;
                const uint32_t e0w = ring[slot][0], e1w = ring[slot][1];
                tid = e0w & 0x3fffffffu;
                upd_s = (e0w >> 30) & 1u;
                upd_l = (e0w >> 31) & 1u;
                n_t = (int32_t)(e1w & 0x3fffu);
                k_eff = (int32_t)((e1w >> 14) & 511u);
                nv = (int32_t)(e1w >> 23);
                bstar = m - n_t + nv;                        // in [0, ROWS - 1]
                if (HALF) {
                    hvp = nv <= 0 ? ~0u : (nv >= 32 ? 0u : (~0u << nv));
                    hvn = ~hvp;
                } else {
#pragma unroll
                    for (int i = 0; i < W; ++i) {
                        const int32_t lo = nv - 64 * i;
                        const uint64_t vp = lo <= 0 ? ~(uint64_t)0 : (lo >= 64 ? 0 : (~(uint64_t)0 << lo));
                        L.VP[i] = vp;
                        L.VN[i] = ~vp;
                    }
                }
                const uint32_t e0 = (uint32_t)(ROWS - 1 - nv);
                const uint32_t phi = (e0 - (uint32_t)lane) & 31u;
                L.ztop = 0u - phi;
                col = -(int32_t)phi;
                blk = tbase0 + (32u + e0 - phi) * 4u;
                nsh = 4u * ((32u - phi) & 7u);
                tp = text + (size_t)tid * text_stride + ((32u - phi) >> 3);
                load5(tp, cur);
                run = true;
            }
            const uint32_t taken = nfree < qcount ? nfree : qcount;
            qhead = (qhead + taken) % (uint32_t)RING;
            qcount -= taken;
        }
        const uint64_t runmask = __ballot(run);
        if (runmask == 0) {
            if (exhausted && qcount == 0) break;
            continue;
        }
        uint32_t w0 = __builtin_amdgcn_alignbit(cur[1], cur[0], nsh), w1 = __builtin_amdgcn_alignbit(cur[2], cur[1], nsh),
                 w2 = __builtin_amdgcn_alignbit(cur[3], cur[2], nsh), w3 = __builtin_amdgcn_alignbit(cur[4], cur[3], nsh);
        if (PREFETCH_TEXT && run && col + 32 < n_t) load5(tp + 4, cur);
        if (HALF) {
            uint32_t zreg = 0;
            if (__ballot(run && col + 32 > n_t) == 0) {
#pragma unroll
                for (int jj = 0; jj < 32; ++jj) {
                    const uint32_t code = (uint32_t)__builtin_amdgcn_ubfe(jj < 8 ? w0 : jj < 16 ? w1 : jj < 24 ? w2 : w3, 4 * (jj & 7), 3);
                    lds_u32 *pe = (lds_u32 *)(uintptr_t)(__umul24(code, plane_bytes) + blk);
                    band_step_eq32(hvp, hvn, zreg, pe[jj]);
                }
            } else {
                // a lane's text ends inside this block: its columns behind the end run as virtual columns (band_core.hpp)
                const int32_t rem = n_t - col;
                const uint32_t act = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
#pragma unroll
                for (int jj = 0; jj < 32; ++jj) {
                    const uint32_t code = (uint32_t)__builtin_amdgcn_ubfe(jj < 8 ? w0 : jj < 16 ? w1 : jj < 24 ? w2 : w3, 4 * (jj & 7), 3);
                    lds_u32 *pe = (lds_u32 *)(uintptr_t)(__umul24(code, plane_bytes) + blk);
                    band_step_eq32_tail(hvp, hvn, zreg, pe[jj], (uint32_t)__builtin_amdgcn_sbfe(act, jj, 1));
                }
            }
            L.ztop += (uint32_t)__popc(zreg);
        } else if (W == 1) {
            // 64-row band: all 32 columns unrolled, immediate table offsets; a block in which some lane's text ends runs that lane's
            // columns behind the end as virtual columns (band_core.hpp) -- every block is 32 columns for every lane, no per-column branch
            uint32_t zreg = 0;
            if (__ballot(run && col + 32 > n_t) == 0) {
#pragma unroll
                for (int jj = 0; jj < 32; ++jj) {
                    const uint32_t code = (uint32_t)__builtin_amdgcn_ubfe(jj < 8 ? w0 : jj < 16 ? w1 : jj < 24 ? w2 : w3, 4 * (jj & 7), 3);
                    lds_u32 *pe = (lds_u32 *)(uintptr_t)(__umul24(code, plane_bytes) + blk);
                    band_step_eq64z(L.VP[0], L.VN[0], zreg, ((uint64_t)pe[jj + 32] << 32) | pe[jj]);
                }
            } else {
                const int32_t rem = n_t - col;
                const uint32_t act = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
#pragma unroll
                for (int jj = 0; jj < 32; ++jj) {
                    const uint32_t code = (uint32_t)__builtin_amdgcn_ubfe(jj < 8 ? w0 : jj < 16 ? w1 : jj < 24 ? w2 : w3, 4 * (jj & 7), 3);
                    lds_u32 *pe = (lds_u32 *)(uintptr_t)(__umul24(code, plane_bytes) + blk);
                    band_step_eq64z_tail(L.VP[0], L.VN[0], zreg, ((uint64_t)pe[jj + 32] << 32) | pe[jj], (uint32_t)__builtin_amdgcn_sbfe(act, jj, 1));
                }
            }
            L.ztop += (uint32_t)__popc(zreg);
        } else if (__ballot(run && col + 32 > n_t) == 0) {
            uint32_t a_blk = blk;
#pragma unroll 1
            for (int g = 0; g < 4; ++g) {
                // 8 columns of one text dword; W = 8 keeps this a real loop (unrolled, its 16 dwords of Eq per column
                // in flight for several columns push the kernel over 128 VGPRs and into scratch)
#pragma unroll UNROLL_COLS
                for (int u = 0; u < 8; ++u) {
                    const uint32_t code = (uint32_t)__builtin_amdgcn_ubfe(w0, 4 * u, 3);
                    uint64_t EQ[W];
                    if (UNROLL_COLS == 1) fetch(__umul24(code, plane_bytes) + a_blk + (uint32_t)u * 4u, 0, EQ);
                    else fetch(__umul24(code, plane_bytes) + a_blk, u, EQ);
                    band_step_eq<W>(L, EQ);
                }
                w0 = w1; w1 = w2; w2 = w3;
                a_blk += 32u;
            }
        } else {
            uint32_t a_blk = blk;
#pragma unroll 1
            for (int g = 0; g < 4; ++g) {
#pragma unroll 1
                for (int u = 0; u < 8; ++u) {
                    if (run && col + g * 8 + u < n_t) {
                        const uint32_t code = (w0 >> (4 * u)) & 7u;
                        uint64_t EQ[W];
                        fetch(code * plane_bytes + a_blk + (uint32_t)u * 4u, 0, EQ);
                        band_step_eq<W>(L, EQ);
                    }
                }
                w0 = w1; w1 = w2; w2 = w3;
                a_blk += 32u;
            }
        }
        n_blocks = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n_blocks + 1u));
        if (W < 8) n_live += (uint32_t)__popcll(runmask);
        col += 32;
        const bool fin = run && col >= n_t;
        int32_t dv;
        if (HALF) {
            const uint32_t lm = bstar <= 0 ? 0u : (bstar >= 32 ? ~0u : ((1u << bstar) - 1u));
            dv = nv + col - (int32_t)L.ztop + __popc(hvp & lm) - __popc(hvn & lm);          // (virtual columns behind the end count on both sides)
        } else dv = diag_value_w<W>(L, nv, (W == 1 || !fin) ? col : n_t, bstar);
        int32_t r = -1;
        if (fin) { r = dv <= k_eff ? dv : -1; run = false; }
        else if (run && dv > k_eff) run = false;
        if (__ballot(fin && r >= P.min_d) != 0) {
            bool hs = false, hl = false;
            if (fin && r >= P.min_d) {
                if (upd_s && r <= m) hs = nn_take(P, q, r);
                if (upd_l && r <= n_t) hl = nn_take(P, tid, r);
            }
            nn_append(P, hs, (int32_t)q, (int32_t)tid, r);
            nn_append(P, hl, (int32_t)tid, (int32_t)q, r);
        }
        blk += 128u;
        tp += 4;
        if (!run) blk = idle_blk;
        if (!PREFETCH_TEXT && run) load5(tp, cur);
    }
    WaveAcc acc;
    acc.pairs = n_pairs; acc.tiles = W < 8 ? n_batches : 1u; acc.cols = (unsigned long long)n_blocks * 2048ull;
    acc.live = (unsigned long long)n_live * 32ull;          // (W = 8 does not count its live lanes: 0)
    nn_flush_acc(P, acc);
    if (lane == 0 && n_filtered) atomicAdd(P.stats + (size_t)NN_COUNTER_SLOTS * 4, (unsigned long long)n_filtered);
}

// Nibble store for k_nn_scan_refill: row i = [4 dwords of code 4][ceil(len_i / 8) dwords, base j at bits 4(j%8)..][zeros].
__global__ __launch_bounds__(256) void k_build_nibble_text(DevStore S, uint32_t *__restrict__ text, uint32_t text_stride)
{
    const uint32_t i = blockIdx.x;
    if (i >= S.n) return;
    const int32_t len = S.lens[i];
    uint32_t *row = text + (size_t)i * text_stride;
    for (uint32_t j = threadIdx.x; j < text_stride; j += 256) {
        uint32_t v = 0;
        if (j < (uint32_t)NN_TEXT_PAD_FRONT) v = 0x44444444u;
        else {
            const int32_t c0 = (int32_t)(j - NN_TEXT_PAD_FRONT) * 8;
            if (c0 < len) {
                const size_t at = ((size_t)(c0 >> 6) * S.n + i) * 2;
                const uint32_t lo = (uint32_t)(S.planes[at] >> (c0 & 63)) & 0xffu, hi = (uint32_t)(S.planes[at + 1] >> (c0 & 63)) & 0xffu;
#pragma unroll
                for (int b = 0; b < 8; ++b) v |= (((lo >> b) & 1u) | (((hi >> b) & 1u) << 1)) << (4 * b);
            }
        }
        row[j] = v;
    }
}

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

// planes2[chunk][newpos] = planes[chunk][perm[newpos]]  (16 B per element; used to group similar sequences of equal length)
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
