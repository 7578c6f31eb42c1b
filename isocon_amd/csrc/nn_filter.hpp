// nn_filter.hpp -- the SECOND rejection test of the main pass: a greedy block bound on the pairs that survive the q-gram bound.
//
// What the alignment kernels were doing at C3 (50 000 CCS reads): 99 % of the ~10^7 survivors of the q-gram bound are reads of one
// isoform whose distance lies 12-26 % above their threshold; the banded DP ran 90 % of their columns to find that out.  The q-gram
// bound (qgram_mm.hpp) reaches 0.82 of the true distance on those pairs; this one reaches 0.93-0.97 for a twelfth of a DP's work
// (scripts/dev/second_stage_study*.py, profiles/r06a_second_stage_study.txt: 95.7 % of the non-hits rejected, no hit ever).
//
// The bound.  Let found(p) = "the b-gram y[p, p + b) occurs somewhere in x".  Take an optimal alignment of y to x with d edits.  A
// b-gram of y that contains no edited position -- no substituted or deleted base, no insertion between two of its bases -- is copied
// to x in one piece, i.e. is found.  So every gram that is NOT found contains one of at most d positions of y (a substitution /
// deletion at i, an insertion between i and i + 1: position i), and any family of pairwise DISJOINT unfound grams has at most d
// members.  Greedy, left to right over the grams at positions 0, s, 2s, ... (b a multiple of s): the first unfound gram counts, the
// probes that overlap it are skipped, and so on.  count <= d whatever set is used for "found", as long as it holds every gram of x
// (a superset only weakens the bound): here an exact 65 536-bit set of x's 8-grams in LDS.  count > k  =>  d > k: the pair cannot be
// a hit of either end and leaves the list.  (The maximum over all shifted tilings of disjoint blocks is the same idea without the
// restart behind every error; it rejects 84 % where greedy rejects 99 %.)
//
//   k_build_text2        2-bit texts: row i = the bases of sequence i, 16 per dword (base j at bits 2j, 2j + 1 of dword j / 16), zero padded
//   k_nn_block_filter    one workgroup per chunk of the list builder (nn_list.hpp): the owner's 8-gram set as a bitmap in LDS (bit set =
//                        ABSENT), one partner per lane streaming its text in 64-byte pieces, four probes per dword (b = 8, s = 4);
//                        survivors are compacted in LDS and leave as a chunk of the same class (enough of them AND less than half of
//                        the chunk rejected: a set on which grams decide little), as tasks of the second pass, or -- a handful -- as
//                        flat pairs for the one-pair-per-lane kernel;
//   k_nn_block_filter2   the second pass on the finer grid (s = 2): a wave per 64 surviving pairs of one owner, its own 8 KB bitmap;
//   k_nn_chunks_to_pairs what is left in table chunks when they are too few for a table launch joins the flat pairs.
// The pair set only shrinks by pairs with d > threshold, so the graph is the one of /root/reference/modules/nearest_neighbor_graph.py:134-192.
#pragma once
#include "nn_list.hpp"

namespace isocon {

static constexpr int NNF_BATCH = 16;        // text dwords (256 bases) per piece: one 64-byte line per lane
#ifndef ISOCON_NNF_GROUP             // text dwords (4 probes each) the scheduler may interleave in the first pass (experiments: scripts/dev/build_variant.sh)
#define ISOCON_NNF_GROUP 2
#endif
static constexpr int NNF_GROUP = ISOCON_NNF_GROUP;
static constexpr uint32_t NNF_LIST_MIN = 32;        // with the filter on, the list builder makes chunks of owners with at least this many pairs
static constexpr uint32_t NNF_TABLE_CHUNKS = 1024;  // fewer chunks than this behind the filter: their pairs go to the pair-per-lane kernel (k_nn_chunks_to_pairs)
static constexpr uint32_t NNF_GRID = 256 * 8;       // workgroups of the filter launch (they share a queue of chunks)
static constexpr uint32_t NNF_PASS2_MIN = 16;       // a chunk that keeps at least this many pairs (and too few for a table) sends them through the second pass
static constexpr uint32_t NNF_GRID2 = 256 * 4;      // workgroups of the second pass (four waves each, a task per wave at a time)

// a task of the second pass: up to 64 pairs of one owner, list[begin .. begin + count)
struct NNFTask { uint32_t owner, count; unsigned long long begin; };

__host__ __device__ __forceinline__ uint32_t nnf_text2_stride(int32_t maxlen)
{
    // whole pieces, one more for the prefetch behind the last one, and the dword behind that
    const uint32_t dw = ((uint32_t)maxlen + 15u) / 16u;
    return ((dw + NNF_BATCH - 1) / NNF_BATCH + 1u) * NNF_BATCH + 4u;
}

__device__ __forceinline__ uint32_t nnf_spread16(uint32_t x)
{
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

__global__ __launch_bounds__(256) void k_build_text2(DevStore S, uint32_t *__restrict__ text, uint32_t stride)
{
    const uint32_t i = blockIdx.x;
    if (i >= S.n) return;
    const int32_t len = S.lens[i];
    uint32_t *row = text + (size_t)i * stride;
    for (uint32_t j = threadIdx.x; j < stride; j += 256) {
        uint32_t v = 0;
        const int32_t c0 = (int32_t)j * 16;
        if (c0 < len) {
            const size_t at = ((size_t)(c0 >> 6) * S.n + i) * 2;
            const uint32_t lo = (uint32_t)(S.planes[at] >> (c0 & 63)) & 0xffffu, hi = (uint32_t)(S.planes[at + 1] >> (c0 & 63)) & 0xffffu;
            v = nnf_spread16(lo) | (nnf_spread16(hi) << 1);          // (bases past the end are 0 in the planes)
        }
        row[j] = v;
    }
}

__device__ __forceinline__ unsigned long long nnf_uniform64(unsigned long long v)
{
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// one probe: the 8-gram whose 16-bit code sits in the low half of c (higher bits: anything); 1 if the owner has no such gram
__device__ __forceinline__ uint32_t nnf_absent(const uint32_t *bitmap, uint32_t c)
{
    const uint32_t word = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(bitmap) + ((c >> 3) & 0x1ffcu));
    return (word >> (c & 31u)) & 1u;
}

// the grams of a sequence (all positions) leave an ABSENT bitmap (all ones before); thread `tid` of `nthreads` that share the bitmap
__device__ __forceinline__ void nnf_add_grams(uint32_t *bitmap, const uint32_t *rx, int32_t len, int32_t tid = (int32_t)threadIdx.x, int32_t nthreads = 256)
{
    const int32_t ng = len - 7;
    for (int32_t j = tid; j * 16 < ng; j += nthreads) {
        const uint32_t w0 = rx[j], w1 = rx[j + 1];
        const uint64_t ww = ((uint64_t)w1 << 32) | w0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (j * 16 + t < ng) {
                const uint32_t code = (uint32_t)(ww >> (2 * t)) & 0xffffu;
                atomicAnd(&bitmap[code >> 5], ~(1u << (code & 31u)));
            }
        }
    }
}

// dwords of a text all of whose grams (bases 16 j + 0, S, 2 S ..., eight bases each) lie inside a sequence of `len` bases
template <int S>
__host__ __device__ __forceinline__ int32_t nnf_dwords(int32_t len) { return len >= 24 - S ? (len - (24 - S)) / 16 + 1 : 0; }

// One lane, one partner: the greedy count over its probes (stride S = 4 or 2 bases), at least until every lane of the wave is decided
// (count > k or the end of the text).  Called by all 64 lanes (the loop bound and the early exit are wave-uniform); `row` = the partner's
// text, nd = nnf_dwords<S>(its length).  A counted gram covers the next 8 / S - 1 probes: they are skipped.
template <int S>
__device__ __forceinline__ uint32_t nnf_count(const uint32_t *bitmap, const uint32_t *row, int32_t nd, int32_t k, bool active)
{
    static_assert(S == 4 || S == 2, "probe stride");
    const int32_t nb = wave_max_i32((nd + NNF_BATCH - 1) / NNF_BATCH);
    uint32_t cnt = 0;
    uint32_t r1 = 0, r2 = 0, r3 = 0;          // the last three probes' "counted" bits (S = 4: only r1 matters)
    // a piece = 16 dwords and the first dword of the next one (the last grams of a piece end there)
    auto load17 = [](const uint32_t *p, uint32_t (&d)[17]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 t = reinterpret_cast<const uint4 *>(p)[q];
            d[4 * q] = t.x; d[4 * q + 1] = t.y; d[4 * q + 2] = t.z; d[4 * q + 3] = t.w;
        }
        d[16] = p[16];
    };
    auto piece = [&](const uint32_t (&w)[17], int32_t b) {
        int32_t rem = nd - b * NNF_BATCH;
        rem = rem < 0 ? 0 : rem > 16 ? 16 : rem;
        const uint32_t lm = (1u << rem) - 1u;          // dwords of the piece that count
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const uint32_t lv = (lm >> jj) & 1u;
            uint32_t sum = 0;
#pragma unroll
            for (int t = 0; t < 16 / S; ++t) {
                const int sh = 2 * S * t;
                const uint32_t c = sh == 0 ? w[jj] : sh <= 16 ? w[jj] >> sh : __builtin_amdgcn_alignbit(w[jj + 1], w[jj], sh);
                const uint32_t a = nnf_absent(bitmap, c);
                const uint32_t h = S == 4 ? a & ~r1 & lv : a & ~(r1 | r2 | r3) & lv;
                r3 = r2; r2 = r1; r1 = h;
                sum += h;
            }
            cnt += sum;
            // (without this the scheduler hoists the whole piece's 64 / 128 lookups: 234 registers, two waves per SIMD)
            if ((jj & (S == 4 ? NNF_GROUP - 1 : 0)) == (S == 4 ? NNF_GROUP - 1 : 0)) __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one piece in registers at a time: eight waves per SIMD hide a piece's load behind the other waves' probes (two register sets with the
    // next piece in flight took 124 / 194 registers -- stride 4 / 2 -- i.e. four / two waves, and ran slower)
    uint32_t w[17];
#pragma unroll 1
    for (int32_t b = 0; b < nb; ++b) {
        load17(row + (size_t)b * NNF_BATCH, w);
        piece(w, b);
        if (__ballot(active && (int32_t)cnt <= k) == 0) break;          // wave-uniform: every lane's pair is decided
    }
    return cnt;
}

#ifndef ISOCON_NNF_WAVES          // waves per SIMD the filter is compiled for (register budget 512 / this); experiments: scripts/dev/build_variant.sh
#define ISOCON_NNF_WAVES 4
#endif
__global__ __launch_bounds__(256, ISOCON_NNF_WAVES) void k_nn_block_filter(const uint32_t *__restrict__ text, uint32_t stride, const int32_t *__restrict__ lens, const uint32_t *__restrict__ meta,
                                                          int32_t kcap, uint32_t *__restrict__ list, const NNChunk *__restrict__ chunks_in, unsigned long long cap_in,
                                                          NNChunk *__restrict__ chunks_out, unsigned long long cap_out, uint32_t *__restrict__ pa, uint32_t *__restrict__ pb,
                                                          unsigned long long small_cap, NNPlanTotals *__restrict__ totals, uint32_t list_min, NNFTask *__restrict__ tasks,
                                                          unsigned long long tasks_cap)
{
    __shared__ uint32_t bitmap[2048];
    __shared__ uint32_t pass[NN_STAGE];
    __shared__ uint32_t s_npass;
    __shared__ unsigned long long s_chunk, s_base;
    const int lane = threadIdx.x & 63;
    if (totals->overflow) return;          // (the list builder gave up: its chunk tables have holes)
    const unsigned long long nA = totals->n_chunks < cap_in ? totals->n_chunks : cap_in;
    const unsigned long long nB = totals->n_chunks_narrow < cap_in ? totals->n_chunks_narrow : cap_in;
    for (;;) {
        __syncthreads();          // (the previous chunk's LDS is no longer read)
        if (threadIdx.x == 0) { s_chunk = atomicAdd(&totals->f_next, 1ull); s_npass = 0; }
        for (uint32_t i = threadIdx.x; i < 2048; i += 256) bitmap[i] = 0xffffffffu;
        __syncthreads();
        const unsigned long long c = nnf_uniform64(s_chunk);          // (scalar: the chunk record, its owner and the loop bounds stay out of the vector registers)
        if (c >= nA + nB) break;
        const bool narrow = c >= nA;
        const NNChunk ch = narrow ? chunks_in[cap_in + (c - nA)] : chunks_in[c];
        const uint32_t x = ch.slot;
        nnf_add_grams(bitmap, text + (size_t)x * stride, lens[x]);
        __syncthreads();
        const uint32_t mx = meta[x];
        const int32_t kx0 = nn_meta_thr(mx);
        uint32_t rejected = 0;
        // the threshold of a listed pair, as the list builder computed it (nn_list.hpp)
        auto threshold = [&](uint32_t word) {
            int32_t k = -1;
            if (word & 0x40000000u) k = kx0;
            if (word & 0x80000000u) { const int32_t ky = nn_meta_thr(meta[word & 0x3fffffffu]); k = ky > k ? ky : k; }
            return k > kcap ? kcap : k;
        };
        for (uint32_t i0 = (threadIdx.x >> 6) * 64u; i0 < ch.count; i0 += 256u) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool active = i < ch.count;
            const uint32_t word = active ? list[ch.begin + i] : 0u;
            const uint32_t y = word & 0x3fffffffu;
            const int32_t k = active ? threshold(word) : -1;
            const uint32_t cnt = nnf_count<4>(bitmap, text + (size_t)y * stride, active ? nnf_dwords<4>(lens[y]) : 0, k, active);
            const bool keep = active && (int32_t)cnt <= k;
            rejected += (uint32_t)__popcll(__ballot(active && !keep));
            if (keep) pass[atomicAdd(&s_npass, 1u)] = word;
        }
        __syncthreads();
        const uint32_t np = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_npass);
        // what is left stays a chunk for the table kernels when it is enough for a table AND the test rejected less than half of the chunk (a set on
        // which the grams decide little); otherwise the second pass looks at it, or -- a handful -- it joins the flat pairs as it is
        const bool to_table = np >= list_min && (tasks == nullptr || 2u * np >= ch.count);
        if (threadIdx.x == 0) {
            unsigned long long base = 0;
            if (to_table) {
                const unsigned long long ci = atomicAdd(narrow ? &totals->f_chunks_narrow : &totals->f_chunks, 1ull);
                if (ci < cap_out) { NNChunk o; o.slot = x; o.count = np; o.begin = ch.begin; chunks_out[(narrow ? cap_out : 0ull) + ci] = o; }
                else { atomicOr(&totals->overflow, 1ull); base = ~0ull; }
                if (narrow) atomicAdd(&totals->f_narrow_listed, (unsigned long long)np);
                atomicAdd(&totals->f_listed, (unsigned long long)np);
            } else if (tasks != nullptr && np >= NNF_PASS2_MIN) {
                // the second pass takes them (k_nn_block_filter2): they go back to the chunk's own part of the list, 64 per task
                const uint32_t nt = (np + 63u) / 64u;
                base = atomicAdd(&totals->f_tasks, (unsigned long long)nt);
                if (base + nt > tasks_cap) { atomicOr(&totals->overflow, 1ull); base = ~0ull; }
            } else if (np) {
                base = atomicAdd(&totals->n_small, (unsigned long long)np);
                if (base + np > small_cap) { atomicOr(&totals->overflow, 1ull); base = ~0ull; }
            }
            s_base = base;
        }
        if ((threadIdx.x & 63) == 0 && rejected) atomicAdd(&totals->f_rejected, (unsigned long long)rejected);
        __syncthreads();
        const unsigned long long base = nnf_uniform64(s_base);
        if (base != ~0ull) {
            if (to_table) { for (uint32_t i = threadIdx.x; i < np; i += 256) list[ch.begin + i] = pass[i]; }
            else if (tasks != nullptr && np >= NNF_PASS2_MIN) {
                for (uint32_t i = threadIdx.x; i < np; i += 256) list[ch.begin + i] = pass[i];
                for (uint32_t j = threadIdx.x; j * 64u < np; j += 256) {
                    NNFTask t; t.owner = x; t.count = np - j * 64u < 64u ? np - j * 64u : 64u; t.begin = ch.begin + (unsigned long long)j * 64ull;
                    tasks[base + j] = t;
                }
            }
            else { for (uint32_t i = threadIdx.x; i < np; i += 256) { pa[base + i] = x; pb[base + i] = pass[i] & 0x3fffffffu; } }
        }
    }
}

// The second pass: what the first leaves of a chunk (7 % at C3), on the finer grid -- a probe every 2 bases separates errors that the 4-base
// grid merges (the study: 99.0 % of the non-hits rejected against 95.7 %).  Inside k_nn_block_filter it would run per chunk on a seventh of
// the lanes and costs what it saves (profiles/r06d_filter_second_pass.txt); here a WAVE takes 64 pairs of one owner at a time from a task
// queue, with an 8 KB bitmap of its own (the owner's grams are filed again: 40 LDS atomics per lane), and appends its survivors to the flat
// pairs of the one-pair-per-lane kernel.  Waves of a workgroup never wait for each other (no barrier: a wave only reads the LDS it wrote).
__global__ __launch_bounds__(256, 3) void k_nn_block_filter2(const uint32_t *__restrict__ text, uint32_t stride, const int32_t *__restrict__ lens, const uint32_t *__restrict__ meta,
                                                             int32_t kcap, const uint32_t *__restrict__ list, const NNFTask *__restrict__ tasks, unsigned long long tasks_cap,
                                                             uint32_t *__restrict__ pa, uint32_t *__restrict__ pb, unsigned long long small_cap, NNPlanTotals *__restrict__ totals)
{
    __shared__ uint32_t bitmaps[4][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *bitmap = bitmaps[wave];
    if (totals->overflow) return;
    const unsigned long long n_tasks = totals->f_tasks < tasks_cap ? totals->f_tasks : tasks_cap;
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;
    for (;;) {
        unsigned long long ti = 0;
        if (lane == 0) ti = atomicAdd(&totals->f_task_next, 1ull);
        ti = nnf_uniform64(ti);
        if (ti >= n_tasks) break;
        const NNFTask t = tasks[ti];
        const uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.owner), count = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.count);
        const unsigned long long begin = nnf_uniform64(t.begin);
        for (int i = lane; i < 2048; i += 64) bitmap[i] = 0xffffffffu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        nnf_add_grams(bitmap, text + (size_t)x * stride, lens[x], lane, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int32_t kx0 = nn_meta_thr(meta[x]);
        const bool active = (uint32_t)lane < count;
        const uint32_t word = active ? list[begin + (unsigned long long)lane] : 0u;
        const uint32_t y = word & 0x3fffffffu;
        int32_t k = -1;
        if (active) {
            if (word & 0x40000000u) k = kx0;
            if (word & 0x80000000u) { const int32_t ky = nn_meta_thr(meta[y]); k = ky > k ? ky : k; }
            if (k > kcap) k = kcap;
        }
        const uint32_t cnt = nnf_count<2>(bitmap, text + (size_t)y * stride, active ? nnf_dwords<2>(lens[y]) : 0, k, active);
        const bool keep = active && (int32_t)cnt <= k;
        const uint64_t km = __ballot(keep);
        const uint32_t nk = (uint32_t)__popcll(km);
        unsigned long long base = 0;
        if (lane == 0) {
            if (nk) {
                base = atomicAdd(&totals->n_small, (unsigned long long)nk);
                if (base + nk > small_cap) { atomicOr(&totals->overflow, 1ull); base = ~0ull; }
            }
            if (count > nk) atomicAdd(&totals->f_rejected2, (unsigned long long)(count - nk));
        }
        base = nnf_uniform64(base);
        if (keep && base != ~0ull) {
            const unsigned long long at = base + (unsigned long long)__popcll(km & lt_mask);
            pa[at] = x; pb[at] = y;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // (the next task's bitmap writes behind this task's reads)
    }
}

// Few chunks left (the filter rejected nearly everything): a table launch of a few dozen workgroups is one lone wave per SIMD for the length of
// a whole chunk (1.4 ms for 57 000 pairs at C3), the pair-per-lane kernel takes the same pairs in 0.15 ms.  One workgroup per chunk appends
// its pairs to the flat arrays (the host adds f_listed to its count of them).
__global__ __launch_bounds__(256) void k_nn_chunks_to_pairs(const NNChunk *__restrict__ chunks, const uint32_t *__restrict__ list, uint32_t *__restrict__ pa, uint32_t *__restrict__ pb,
                                                             unsigned long long small_cap, NNPlanTotals *__restrict__ totals)
{
    __shared__ unsigned long long s_base;
    const NNChunk ch = chunks[blockIdx.x];
    if (threadIdx.x == 0) {
        unsigned long long base = atomicAdd(&totals->n_small, (unsigned long long)ch.count);
        if (base + ch.count > small_cap) { atomicOr(&totals->overflow, 1ull); base = ~0ull; }
        s_base = base;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    if (base == ~0ull) return;
    for (uint32_t i = threadIdx.x; i < ch.count; i += 256) { pa[base + i] = ch.slot; pb[base + i] = list[ch.begin + i] & 0x3fffffffu; }
}

// The count for explicit (owner, partner) pairs, one workgroup per pair (tests: isocon_block_bound_pairs) -- the same two device functions.
__global__ __launch_bounds__(256) void k_nn_block_count_pairs(const uint32_t *__restrict__ text, uint32_t stride, const int32_t *__restrict__ lens, const uint32_t *__restrict__ owner,
                                                               const uint32_t *__restrict__ partner, int32_t *__restrict__ out, int32_t probe_stride)
{
    __shared__ uint32_t bitmap[2048];
    for (uint32_t i = threadIdx.x; i < 2048; i += 256) bitmap[i] = 0xffffffffu;
    __syncthreads();
    const uint32_t x = owner[blockIdx.x], y = partner[blockIdx.x];
    nnf_add_grams(bitmap, text + (size_t)x * stride, lens[x]);
    __syncthreads();
    if (threadIdx.x < 64) {
        const bool active = threadIdx.x == 0;
        const uint32_t cnt = probe_stride == 2 ? nnf_count<2>(bitmap, text + (size_t)y * stride, active ? nnf_dwords<2>(lens[y]) : 0, 0x7fffffff, active)
                                               : nnf_count<4>(bitmap, text + (size_t)y * stride, active ? nnf_dwords<4>(lens[y]) : 0, 0x7fffffff, active);
        if (active) out[blockIdx.x] = (int32_t)cnt;
    }
}

}  // namespace isocon
