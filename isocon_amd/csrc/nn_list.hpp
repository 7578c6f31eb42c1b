// nn_list.hpp -- the main pass of the nearest-neighbour search as LISTS: which pairs survive their q-gram bound is decided in one
// streaming pass over the bound matrix, the alignment kernels only see survivors, and every survivor is aligned against the
// match-mask table of whichever of its two ends has more pairs.
//
// Why (C3, 50 000 reads, measured: profiles/r03b_survivor_graph.txt): of the 3.5 10^8 window pairs ~1.1 10^7 survive (bound <= threshold),
// and they are spread very unevenly -- per entry as the lower index: median 22, 75 % 123, 90 % 449, 99 % 3 716, largest 10 904 (reads with
// few errors are near everybody).  A workgroup per entry that walks its whole window inside the alignment kernel (~110 batches of
// dependent loads per entry) spends its life finding 22 pairs and then runs them on 256 lanes; a 48 KB table in LDS pays only
// when hundreds of lanes use it.  Distances are symmetric, so a pair may use either end's table: handing every pair to the end of
// larger degree puts 88 % of the pairs into lists of >= 512 (73 % by lower index).
// Grouping pairs by an end that is not the row they were found in is a sparse transposition (two scattered atomics per pair: measured
// 4.8 ms for 1.3 10^7 pairs); instead k_qgram_mm also writes the TRANSPOSED matrix and a hub score per entry (pairs with a small
// bound, counted in its epilogue), so that ONE pass finds every entry's pairs on both sides and both ends of a pair agree on
// its owner without communicating:
//   1. k_nn_survivors   one wave per entry: its row and its transposed row through the in-kernel admission's tests (roles, window,
//                       threshold from the same best[], bound); the pairs it owns are staged in LDS and leave as chunks of a list
//                       (one k_nn_scan_refill workgroup per chunk: the entry's table in LDS, lanes refilled from the chunk) or, for
//                       entries with few pairs, as flat pairs for the one-pair-per-lane kernel (ed_lanes.hpp: no table);
//   2. the block filter on the chunks (nn_filter.hpp), then the alignment launches on what it leaves (nn_lists.inc, nn_main.inc).
// The pair set is exactly the one the in-kernel admission evaluates, so the graph is unchanged; replaces the window walk of
// /root/reference/modules/nearest_neighbor_graph.py:136-178.
#pragma once
#include "nn.hpp"

namespace isocon {

static constexpr uint32_t NN_LIST_MIN = 256;       // owners with fewer pairs (and what is left of a list below this): one pair per lane
#ifndef ISOCON_NN_LIST_CHUNK          // (experiments build the library with another value: scripts/dev/build_variant.sh)
#define ISOCON_NN_LIST_CHUNK 2048
#endif
static constexpr uint32_t NN_LIST_CHUNK = ISOCON_NN_LIST_CHUNK;    // a wave hands over a chunk as soon as it has staged this many pairs of its entry
static constexpr int NN_STAGE = NN_LIST_CHUNK + 64;
#ifndef ISOCON_SURV_WAVES            // waves (= entries) per workgroup of the list builder.  One: 8.4 KB of LDS per workgroup, so residency is bound by
#define ISOCON_SURV_WAVES 1          // registers (24 waves per CU) instead of LDS (16 with four): 1.06 -> 0.98 ms at C3 (profiles/r05t_chunk_sweep.txt)
#endif
static constexpr int NN_SURV_WAVES = ISOCON_SURV_WAVES;

// counters of the list builder, each on a cache line of its own (they are hot: tens of thousands of atomics per launch)
struct NNPlanTotals {
    unsigned long long n_pairs, pad0[15], n_list, pad1[15], n_small, pad2[15], n_chunks, pad3[15], n_filtered, pad4[15], overflow, pad5[15];
    unsigned long long n_chunks_narrow, pad6[15];          // chunks of the 32-row class (n_chunks counts the 64-row class)
    unsigned long long n_wide_pairs, n_narrow_listed, pad7[14];          // pairs sent to the pair-per-lane kernel for their threshold (class mode 1); pairs in narrow chunks
    // the block filter behind the list builder (nn_filter.hpp): its work queue, the chunks it leaves per class, the pairs it rejected
    unsigned long long f_next, pad8[15], f_chunks, pad9[15], f_chunks_narrow, pad10[15], f_rejected, f_narrow_listed, f_listed, f_rejected2, pad11[12];          // f_listed: pairs in the chunks it leaves (both classes); f_rejected2: by its second pass
    unsigned long long f_tasks, pad12[15], f_task_next, pad13[15];          // tasks of the second pass (nn_filter.hpp) and its queue
};

// Threshold classes of the listed launch.  A pair whose threshold max(k of its two directions) is <= NN_NARROW_K is exact on 32
// diagonals (a path of cost <= k stays inside k + 1 of them), and on 32-bit vectors the table kernel's column is 12 instead of 21
// instructions with one table dword instead of two (nn.hpp, HALF).  The class is decided PER PAIR: the list builder files the pairs of
// an entry under their class, a chunk holds one class, and the host launches k_nn_scan_refill<.., true> over the narrow chunks and
// <.., false> over the others.  (Round 3 decided per LAUNCH -- narrow only when fewer than n / 16 queries had a threshold above 31 --
// so C3, whose median threshold is 32, ran every pair on 64 rows.)
// class_mode: 0 = per pair; -1 = every pair in the 64-row class (A/B runs, tests); +1 = the pairs above NN_NARROW_K leave the lists
// for the one-pair-per-lane kernel, which takes any threshold up to 63 (the round-3 narrow mode, forced).
static constexpr int32_t NN_NARROW_K = 31;

// the matrix rows of both orientations (qgram_mm.hpp) and the hub scores that decide which end owns a pair
struct NNBoundRows {
    const uint8_t *lb;  const unsigned long long *row_off; const uint32_t *row_len;                      // row of slot s: columns p = q(s) + 1 ...
    const uint8_t *lbT; const unsigned long long *offT;    const uint32_t *sloT; const uint32_t *lenT;   // row of entry p: slots sloT[p] ...
    const uint32_t *score;
};

// what the scan needs to know about a partner, in ONE dword (the scan reads one per window pair and side: 7 10^8 at C3, L2 traffic):
// length (14 bits: the table kernel's limit is below), threshold min(best, length, 64) (7 bits: anything above 63 acts as 63),
// roles (2 bits: target, query), hub score >> 5 capped at 511 (9 bits: only orders the two ends of a pair, the same on both sides)
__device__ __forceinline__ uint32_t nn_meta_len(uint32_t w) { return w & 0x3fffu; }
__device__ __forceinline__ int32_t nn_meta_thr(uint32_t w) { return (int32_t)((w >> 14) & 0x7fu); }
__device__ __forceinline__ uint32_t nn_meta_score(uint32_t w) { return w >> 23; }

__global__ __launch_bounds__(256) void k_nn_entry_meta(DevStore S, NNParams P, const uint32_t *__restrict__ score, uint32_t *__restrict__ meta, NNPlanTotals *__restrict__ totals)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x < S.n) {
        const int32_t m = S.lens[x], b = load_relaxed_agent(P.best + x);
        int32_t thr = b < m ? b : m;
        thr = thr < 64 ? thr : 64;
        const uint32_t sc = score[x] >> 5;
        const bool isq = P.qflag[x] != 0;
        meta[x] = ((uint32_t)m & 0x3fffu) | ((uint32_t)thr << 14) | ((P.tflag[x] != 0 ? 1u : 0u) << 21) | ((isq ? 1u : 0u) << 22) | ((sc < 511u ? sc : 511u) << 23);
    }
}

// One wave per entry x.  Its pairs are the columns of its own row (partners above x; only if x is one of the launch slots) and the
// slots of its transposed row (partners below x): the same survival test from both sides (roles, length window, threshold from the
// same best[], bound), and the same ownership rule -- the end with the larger hub score, ties to the lower index -- so every
// surviving pair is kept by exactly one of its two ends.  Kept pairs are staged in LDS; NN_LIST_CHUNK staged pairs become a chunk
// of `list` (one k_nn_scan_refill workgroup with x's table), what is left at the end becomes a last chunk if it has NN_LIST_MIN
// pairs, else flat pairs for the one-pair-per-lane kernel.
__global__ __launch_bounds__(64 * NN_SURV_WAVES) void k_nn_survivors(DevStore S, NNParams P, NNBoundRows B, const uint32_t *__restrict__ meta, QMap Q, uint32_t nq,
                                                       uint32_t *__restrict__ list, unsigned long long list_cap, NNChunk *__restrict__ chunks, unsigned long long chunks_cap,
                                                       uint32_t *__restrict__ pa, uint32_t *__restrict__ pb, unsigned long long small_cap, NNPlanTotals *__restrict__ totals, uint32_t list_min,
                                                       int32_t class_mode)
{
    __shared__ uint32_t stage[NN_SURV_WAVES][NN_STAGE];
    __shared__ uint32_t s_small[NN_SURV_WAVES], s_filtered[NN_SURV_WAVES], s_kept[NN_SURV_WAVES];
    __shared__ unsigned long long s_base;
    const int wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * (uint32_t)NN_SURV_WAVES + (uint32_t)wave;
    const int lane = threadIdx.x & 63;
    const uint32_t mx = x < S.n ? meta[x] : 0u;
    const bool x_isq = (mx & (1u << 22)) != 0, x_ist = (mx & (1u << 21)) != 0;
    // x's own row, if it is a launch slot, and its transposed row
    uint32_t up_len = 0, dn_len = 0, dn_slo = 0;
    unsigned long long up_off = 0, dn_off = 0;
    if (x < S.n && (x_isq || x_ist)) {
        uint32_t s;
        if (Q.slot_of(x, s) && s < nq) {
            up_len = B.row_len[s];
            up_off = B.row_off[s];
        }
        dn_len = B.lenT[x]; dn_slo = B.sloT[x]; dn_off = B.offT[x];
    }
    const int32_t m = (int32_t)nn_meta_len(mx);
    const int32_t kx0 = nn_meta_thr(mx);
    const uint32_t sx = nn_meta_score(mx);
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;
    uint32_t *st = stage[wave];
    // the staging buffer holds both classes: the 64-row class grows from its front (fill), the 32-row class from its back (fill_n)
    uint32_t fill = 0, fill_n = 0, filtered = 0, kept = 0;
    // one class leaves as a chunk: the 64-row chunks in chunks[0 .. chunks_cap), the 32-row chunks in chunks[chunks_cap .. 2 chunks_cap)
    // (merged: what is left of the 32-row class leaves together with the 64-row class, as a 64-row chunk)
    auto emit_chunk = [&](bool narrow_class, bool merged = false) {
        const uint32_t cnt = merged ? fill + fill_n : narrow_class ? fill_n : fill;
        unsigned long long base = 0, ci = 0;
        if (lane == 0) {
            base = atomicAdd(&totals->n_list, (unsigned long long)cnt);
            ci = atomicAdd(narrow_class ? &totals->n_chunks_narrow : &totals->n_chunks, 1ull);
            if (narrow_class) atomicAdd(&totals->n_narrow_listed, (unsigned long long)cnt);
        }
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        ci = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ci >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ci);
        if (base + cnt <= list_cap && ci < chunks_cap) {
            if (narrow_class) { for (uint32_t i = (uint32_t)lane; i < cnt; i += 64) list[base + i] = st[NN_STAGE - 1 - i]; }
            else { for (uint32_t i = (uint32_t)lane; i < cnt; i += 64) list[base + i] = i < fill ? st[i] : st[NN_STAGE - 1 - (i - fill)]; }
            if (lane == 0) { NNChunk ch; ch.slot = x; ch.count = cnt; ch.begin = base; chunks[(narrow_class ? chunks_cap : 0ull) + ci] = ch; }
        } else if (lane == 0) atomicOr(&totals->overflow, 1ull);
        if (narrow_class || merged) fill_n = 0;
        if (!narrow_class) fill = 0;
    };
#ifndef ISOCON_SURV_U
#define ISOCON_SURV_U 8
#endif
    constexpr int U = ISOCON_SURV_U;             // batches of 64 partners per iteration: their loads are independent
    // (the side is a compile-time constant of the loop body: own row = contiguous partners, transposed row = partners through the slot map)
    auto scan_side = [&](auto side_tag) {
        constexpr int side = decltype(side_tag)::value;
        const uint32_t len = side == 0 ? up_len : dn_len;
        const uint8_t *row = side == 0 ? B.lb + up_off : B.lbT + dn_off;
        for (uint32_t c0 = 0; c0 < len; c0 += 64 * U) {
            uint32_t y[U], lbv[U];
            uint32_t my[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t e = c0 + 64u * u + (uint32_t)lane;
                const uint32_t ec = e < len ? e : len - 1;
                y[u] = side == 0 ? x + 1u + ec : (uint32_t)Q.entry(dn_slo + ec);
                my[u] = meta[y[u]];
                lbv[u] = row[ec];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c0 + 64u * u >= len) break;                        // wave-uniform
                const uint32_t e = c0 + 64u * u + (uint32_t)lane;
                const bool inr = e < len;
                const bool xq = inr && x_isq && (my[u] & (1u << 21));        // x queries y
                const bool yq = inr && x_ist && (my[u] & (1u << 22));        // y queries x
                int32_t kx = -1, ky = -1;
                if (xq) kx = kx0;
                if (yq) ky = nn_meta_thr(my[u]);
                int32_t k = kx > ky ? kx : ky;
                if (k > P.kcap) k = P.kcap;
                const int32_t dl = m - (int32_t)nn_meta_len(my[u]), ad = dl < 0 ? -dl : dl;
                const uint32_t sy_u = nn_meta_score(my[u]);
                const bool cand = inr && k >= 0 && ad <= k;
                const bool accept = cand && (int32_t)lbv[u] <= k;
                // owner: larger hub score, ties to the lower index (side 0: x is the lower end)
                const bool mine = accept && (side == 0 ? sy_u <= sx : sx > sy_u);
                if (side == 0) { filtered += (uint32_t)__popcll(__ballot(cand && !accept)); kept += (uint32_t)__popcll(__ballot(accept)); }
                // class mode +1: the pairs with a threshold above the 32-row form's go straight to the pair arrays
                const bool wide = mine && class_mode > 0 && k > NN_NARROW_K;
                const uint64_t wm = __ballot(wide);
                if (wm != 0) {
                    const int first = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(wm));
                    unsigned long long wbase = 0;
                    if (lane == first) {
                        wbase = atomicAdd(&totals->n_small, (unsigned long long)__popcll(wm));
                        atomicAdd(&totals->n_wide_pairs, (unsigned long long)__popcll(wm));
                        if (wbase + (unsigned long long)__popcll(wm) > small_cap) { atomicOr(&totals->overflow, 1ull); wbase = ~0ull; }
                    }
                    wbase = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(wbase >> 32), first) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wbase, first);
                    if (wide && wbase != ~0ull) {
                        const unsigned long long at = wbase + (unsigned long long)__popcll(wm & lt_mask);
                        pa[at] = x; pb[at] = y[u];
                    }
                }
                const bool keep = mine && !wide;
                const bool to_narrow = keep && class_mode >= 0 && k <= NN_NARROW_K;
                const uint64_t am = __ballot(keep && !to_narrow), an = __ballot(to_narrow);
                if ((am | an) == 0) continue;                            // wave-uniform
                const uint32_t word = y[u] | (xq ? 0x40000000u : 0u) | (yq ? 0x80000000u : 0u);
                if (keep && !to_narrow) st[fill + (uint32_t)__popcll(am & lt_mask)] = word;
                if (to_narrow) st[NN_STAGE - 1 - (fill_n + (uint32_t)__popcll(an & lt_mask))] = word;
                fill += (uint32_t)__popcll(am);
                fill_n += (uint32_t)__popcll(an);
                if (fill + fill_n >= NN_LIST_CHUNK) emit_chunk(fill_n > fill);          // the buffer is full: its larger class leaves (>= half a chunk)
            }
        }
    };
    scan_side(std::integral_constant<int, 0>());
    scan_side(std::integral_constant<int, 1>());
    // the end of the entry's pairs: a class with enough pairs leaves as a chunk of its own; too few of the 32-row class join the 64-row
    // class (its kernel takes any threshold) before that class is judged
    if (fill_n >= list_min) emit_chunk(true);
    if (fill + fill_n >= list_min) emit_chunk(false, true);
    // what is left of both classes goes to the pair arrays (the pair-per-lane kernel takes any threshold)
    const uint32_t rest = fill + fill_n;
    // what is left goes to the pair arrays: one allocation per workgroup
    if (lane == 0) { s_small[wave] = rest; s_filtered[wave] = filtered; s_kept[wave] = kept; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0, fsum = 0, ksum = 0;
        for (int w = 0; w < NN_SURV_WAVES; ++w) { tot += s_small[w]; fsum += s_filtered[w]; ksum += s_kept[w]; }
        unsigned long long base = 0;
        if (tot) {
            base = atomicAdd(&totals->n_small, (unsigned long long)tot);
            if (base + tot > small_cap) { atomicOr(&totals->overflow, 1ull); base = ~0ull; }
        }
        s_base = base;
        if (fsum) atomicAdd(&totals->n_filtered, (unsigned long long)fsum);
        if (ksum) atomicAdd(&totals->n_pairs, (unsigned long long)ksum);
    }
    __syncthreads();
    if (rest && s_base != ~0ull) {
        unsigned long long base = s_base;
        for (int w = 0; w < wave; ++w) base += s_small[w];
        for (uint32_t i = (uint32_t)lane; i < rest; i += 64) { pa[base + i] = x; pb[base + i] = (i < fill ? st[i] : st[NN_STAGE - 1 - (i - fill)]) & 0x3fffffffu; }
    }
}

// Largest chunks first (16 size classes, one workgroup): the table launch does not end with a few long workgroups (10.9 -> 10.4 ms at C3).
__global__ __launch_bounds__(1024) void k_nn_sort_chunks(const NNChunk *__restrict__ in, uint32_t n, NNChunk *__restrict__ out)
{
    __shared__ uint32_t hist[16], cursor[16];
    if (threadIdx.x < 16) hist[threadIdx.x] = 0;
    __syncthreads();
    auto cls = [](uint32_t count) { const uint32_t c = count / (NN_LIST_CHUNK / 16 + 1); return 15u - (c < 15u ? c : 15u); };
    for (uint32_t i = threadIdx.x; i < n; i += 1024) atomicAdd(&hist[cls(in[i].count)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t a = 0; for (int b = 0; b < 16; ++b) { cursor[b] = a; a += hist[b]; } }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) { const NNChunk c = in[i]; out[atomicAdd(&cursor[cls(c.count)], 1u)] = c; }
}

}  // namespace isocon
