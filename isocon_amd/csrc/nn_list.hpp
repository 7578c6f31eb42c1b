// nn_list.hpp -- the main pass of the nearest-neighbour search as LISTS: which pairs survive their q-gram bound is decided in one
// streaming pass over the bound matrix, the alignment kernels only see survivors, and every survivor is aligned against the
// match-mask table of whichever of its two ends has more pairs.
//
// Why (C3, 50 000 reads, measured: scripts/dev/survivor_graph.py): of the 3.5 10^8 window pairs ~1.1 10^7 survive (bound <= threshold),
// and they are spread very unevenly -- per entry as the lower index: median 22, 75 % 123, 90 % 449, 99 % 3 716, largest 10 904 (reads with
// few errors are near everybody).  A workgroup per entry that walks its whole window inside the alignment kernel (~110 batches of
// dependent loads per entry) spends its life finding 22 pairs and then runs them on 256 lanes; a 48 KB table in LDS pays only
// when hundreds of lanes use it.  Distances are symmetric, so a pair may use either end's table: handing every pair to the end of
// larger degree puts 88 % of the pairs into lists of >= 512 (73 % by lower index).  Steps:
//   1. k_nn_survivors   one wave per row of the bound matrix: the in-kernel admission's tests (roles, window, threshold, bound) on
//                       the same best[]; survivors go to a flat pair buffer (staged in LDS, one cursor atomic per ~200 pairs), degrees
//                       are counted;
//   2. k_nn_own_count   per pair: owner = the end of larger degree (ties: the lower index); list sizes;
//   3. k_nn_plan        one workgroup: owners with >= NN_LIST_MIN pairs get a LIST, cut into chunks of <= NN_LIST_CHUNK pairs (one
//                       k_nn_scan_refill workgroup per chunk: owner's table in LDS, lanes refilled from the chunk), largest chunks
//                       first; the other pairs go to flat pair arrays for the one-pair-per-lane kernel (ed_lanes.hpp: no table);
//   4. k_nn_own_fill    per pair: into its list / pair array slot;
//   5. the two alignment launches (nn_host.inc).
// The pair set is exactly the one the in-kernel admission evaluates, so the graph is unchanged; replaces the window walk of
// /root/reference/modules/nearest_neighbor_graph.py:136-178.
#pragma once
#include "nn.hpp"

namespace isocon {

static constexpr uint32_t NN_LIST_MIN = 512;       // owners with fewer pairs: their pairs go to the pair-per-lane kernel
static constexpr uint32_t NN_LIST_CHUNK = 4096;    // pairs per table workgroup (512 lanes: eight rounds of refill)
static constexpr int NN_STAGE = 256;               // pairs a wave stages in LDS before it takes room in the pair buffer

struct NNPlanTotals { unsigned long long n_pairs, n_list, n_small, n_chunks, n_filtered, overflow; };

// pairs[i] = (lower entry q, upper entry p | 0x40000000 (q queries p) | 0x80000000 (p queries q))
__global__ __launch_bounds__(256) void k_nn_survivors(DevStore S, NNParams P, const uint32_t *__restrict__ row_len, uint32_t q_begin, uint32_t q_stride, uint32_t nq,
                                                       uint2 *__restrict__ pairs, unsigned long long pairs_cap, uint32_t *__restrict__ deg, NNPlanTotals *__restrict__ totals)
{
    __shared__ uint2 stage[4][NN_STAGE];
    const int wave = threadIdx.x >> 6;
    const uint32_t s = blockIdx.x * 4u + (uint32_t)wave;
    const int lane = threadIdx.x & 63;
    if (s >= nq) return;
    const uint32_t q = q_begin + s * q_stride;
    const bool q_isq = P.qflag[q] != 0, q_ist = P.tflag[q] != 0;
    const uint32_t rl = row_len[s];
    if (rl == 0 || (!q_isq && !q_ist)) return;
    const int32_t m = S.lens[q];
    const unsigned long long lb_base = P.lb_row[s];
    const int32_t bs = q_isq ? load_relaxed_agent(P.best + q) : NN_INF;
    const int32_t ks0 = bs < m ? bs : m;
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;
    uint2 *st = stage[wave];
    uint32_t fill = 0, total = 0, filtered = 0;
    auto flush = [&]() {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&totals->n_pairs, (unsigned long long)fill);
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        if (base + fill <= pairs_cap) {
            for (uint32_t i = (uint32_t)lane; i < fill; i += 64) pairs[base + i] = st[i];
        } else if (lane == 0) atomicOr(&totals->overflow, 1ull);
        fill = 0;
    };
    constexpr int U = 4;                         // batches of 64 columns per iteration: their loads are independent
    for (uint32_t c0 = 0; c0 < rl; c0 += 64 * U) {
        uint32_t pid[U];
        int32_t np[U], bl[U];
        uint32_t fl[U], lbv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = c0 + 64u * u + (uint32_t)lane;
            const uint32_t ec = e < rl ? e : rl - 1;
            pid[u] = q + 1u + ec;
            np[u] = S.lens[pid[u]];
            bl[u] = load_relaxed_agent(P.best + pid[u]);
            fl[u] = (P.tflag[pid[u]] != 0 ? 1u : 0u) | (P.qflag[pid[u]] != 0 ? 2u : 0u);
            lbv[u] = P.lb[lb_base + ec];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = c0 + 64u * u + (uint32_t)lane;
            const bool inr = e < rl;
            const bool us = inr && q_isq && (fl[u] & 1u);
            const bool ul = inr && q_ist && (fl[u] & 2u);
            int32_t ks = -1, kl = -1;
            if (us) ks = ks0;
            if (ul) kl = bl[u] < np[u] ? bl[u] : np[u];
            int32_t k = ks > kl ? ks : kl;
            if (k > P.kcap) k = P.kcap;
            const int32_t dl = m - np[u], ad = dl < 0 ? -dl : dl;
            const bool cand = inr && k >= 0 && ad <= k;
            const bool accept = cand && (int32_t)lbv[u] <= k;
            const uint64_t am = __ballot(accept);
            filtered += (uint32_t)__popcll(__ballot(cand && !accept));
            if (am == 0) continue;                                   // wave-uniform
            if (accept) {
                st[fill + (uint32_t)__popcll(am & lt_mask)] = make_uint2(q, pid[u] | (us ? 0x40000000u : 0u) | (ul ? 0x80000000u : 0u));
                atomicAdd(deg + pid[u], 1u);
            }
            fill += (uint32_t)__popcll(am);
            total += (uint32_t)__popcll(am);
            if (fill > (uint32_t)NN_STAGE - 64u) flush();
        }
    }
    if (fill) flush();
    if (lane == 0) {
        if (total) atomicAdd(deg + q, total);
        if (filtered) atomicAdd(&totals->n_filtered, (unsigned long long)filtered);
    }
}

__device__ __forceinline__ uint32_t nn_pair_owner(uint32_t q, uint32_t p, const uint32_t *__restrict__ deg)
{
    return deg[p] > deg[q] ? p : q;
}

__global__ __launch_bounds__(256) void k_nn_own_count(const uint2 *__restrict__ pairs, unsigned long long n_pairs, const uint32_t *__restrict__ deg, uint32_t *__restrict__ own_cnt)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pairs) return;
    const uint2 pr = pairs[i];
    atomicAdd(own_cnt + nn_pair_owner(pr.x, pr.y & 0x3fffffffu, deg), 1u);
}

// exclusive scan of one value per thread over a workgroup of 1024 threads; *total receives the sum
__device__ __forceinline__ unsigned long long nn_block_exscan(unsigned long long v, unsigned long long *wave_sums, unsigned long long *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = ((unsigned long long)(uint32_t)__shfl_up((int)(x >> 32), o, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wave_sums[wave] = x;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    for (int w = 0; w < 16; ++w) { const unsigned long long t = wave_sums[w]; if (w < wave) before += t; all += t; }
    __syncthreads();
    *total = all;
    return before + x - v;
}

// One workgroup of 1024 threads: thread t plans the entries [t R, (t + 1) R), R = ceil(n / 1024).  dest[x]: index of the first pair
// of owner x in `list` (bit 63 clear) or in the pair arrays (bit 63 set).  Chunks in three buckets, largest first.
__global__ __launch_bounds__(1024) void k_nn_plan(const uint32_t *__restrict__ own_cnt, uint32_t n, unsigned long long *__restrict__ dest,
                                                   NNChunk *__restrict__ chunks, unsigned long long chunks_cap, NNPlanTotals *__restrict__ totals)
{
    __shared__ unsigned long long wave_sums[16];
    const uint32_t t = threadIdx.x;
    const uint32_t R = (n + 1023u) / 1024u;
    const uint32_t r0 = t * R < n ? t * R : n, r1 = (t + 1) * R < n ? (t + 1) * R : n;
    auto bucket = [](uint32_t c) { return c >= 2048u ? 0 : (c >= 1024u ? 1 : 2); };
    unsigned long long nl = 0, ns = 0, nc[3] = {0, 0, 0};
    for (uint32_t x = r0; x < r1; ++x) {
        const uint32_t c = own_cnt[x];
        if (c >= NN_LIST_MIN) {
            nl += c;
            for (uint32_t b = 0; b < c; b += NN_LIST_CHUNK) nc[bucket(c - b < NN_LIST_CHUNK ? c - b : NN_LIST_CHUNK)] += 1;
        } else ns += c;
    }
    unsigned long long tl, ts, tc[3];
    unsigned long long ol = nn_block_exscan(nl, wave_sums, &tl);
    unsigned long long os = nn_block_exscan(ns, wave_sums, &ts);
    unsigned long long oc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) oc[b] = nn_block_exscan(nc[b], wave_sums, &tc[b]);
    oc[1] += tc[0];
    oc[2] += tc[0] + tc[1];
    if (t == 0) { totals->n_list = tl; totals->n_small = ts; totals->n_chunks = tc[0] + tc[1] + tc[2]; }
    for (uint32_t x = r0; x < r1; ++x) {
        const uint32_t c = own_cnt[x];
        if (c >= NN_LIST_MIN) {
            dest[x] = ol;
            for (uint32_t b = 0; b < c; b += NN_LIST_CHUNK) {
                NNChunk ch;
                ch.slot = x; ch.count = c - b < NN_LIST_CHUNK ? c - b : NN_LIST_CHUNK; ch.begin = ol + b;
                const int bk = bucket(ch.count);
                if (oc[bk] < chunks_cap) chunks[oc[bk]] = ch;
                oc[bk] += 1;
            }
            ol += c;
        } else {
            dest[x] = os | ((unsigned long long)1 << 63);
            os += c;
        }
    }
}

// list entry of owner x: partner | 0x40000000 (x queries the partner) | 0x80000000 (the partner queries x)
__global__ __launch_bounds__(256) void k_nn_own_fill(const uint2 *__restrict__ pairs, unsigned long long n_pairs, const uint32_t *__restrict__ deg,
                                                      const unsigned long long *__restrict__ dest, uint32_t *__restrict__ cursor, uint32_t *__restrict__ list,
                                                      uint32_t *__restrict__ pa, uint32_t *__restrict__ pb)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pairs) return;
    const uint2 pr = pairs[i];
    const uint32_t q = pr.x, p = pr.y & 0x3fffffffu;
    const bool q_queries_p = (pr.y & 0x40000000u) != 0, p_queries_q = (pr.y & 0x80000000u) != 0;
    const uint32_t owner = nn_pair_owner(q, p, deg);
    const unsigned long long d = dest[owner];
    const unsigned long long at = (d & ~((unsigned long long)1 << 63)) + atomicAdd(cursor + owner, 1u);
    if (d >> 63) { pa[at] = q; pb[at] = p; }
    else if (owner == q) list[at] = p | (q_queries_p ? 0x40000000u : 0u) | (p_queries_q ? 0x80000000u : 0u);
    else list[at] = q | (p_queries_q ? 0x40000000u : 0u) | (q_queries_p ? 0x80000000u : 0u);
}

}  // namespace isocon
