// hw.hpp -- infix ("HW") edit distance with location and the terminal insertion runs of its path.
//
// Device side of edlib.align(q, t, mode="HW", task="path", k) as consumed by the candidate-vs-candidate graph of the
// statistical-test phase (/root/reference/modules/end_invariant_functions.py:593-620 edlib_traceback, :622-681
// get_all_NN): the query is aligned globally inside the target (free target prefix and suffix); wanted are the
// distance h (<= k), locations[0] = (start, end) and whether the path starts / ends with an insertion run (and how long).
// k is small there (10 + ignore_ends_len) and the two lengths differ by at most 10 + 2*ignore_ends_len, so every path
// of cost <= k stays inside a fixed set of diagonals:
//   phase A  (distance, first end)    rows = query, top row all 0, diagonals j - i in [-k, (m - n) + k]
//   phase B  (start of that end)      reversed query against reversed target[0..end], top row j, diagonals [-k, k];
//                                     the LAST column of the final row that equals h gives the smallest start
//   phase C  (path)                   query against target[start..end], diagonals [-k, k], two decision bits per cell
//                                     (vertical step optimal / horizontal step optimal), then the walk from the end with
//                                     edlib's order: query-only step ('I'), target-only step ('D'), diagonal.
// One wavefront per pair; lane l of block b owns diagonal 64 b + l, rows are processed one at a time:
//   a[d]   = min(prev[d] + (q_i != t_j), prev[d + 1] + 1)              (diagonal and vertical predecessor, one lane shift)
//   cur[d] = min over d' <= d of a[d'] + (d - d')                       (horizontal runs: a prefix-min of a[d'] - d')
// Integer work on 32-bit lanes; both sequences sit in LDS as one byte per base.  HBM traffic = the two packed sequences in
// and 20 B out per pair (+ 16 B per row and block of decision bits for the pairs that reach phase C), so the kernel is
// bound by the dependent chain of cross-lane operations per row (prefix-min: 7 DPP steps; shifts: DPP wave shifts), not by memory.
#pragma once
#include "common.hpp"

namespace isocon {

static constexpr int32_t HW_INF = 1 << 24;
static constexpr int32_t HW_TPAD = 1024;       // bytes around the target in LDS: > 64 * 8 diagonals + k + 2

// Cross-lane moves as DPP modifiers (VALU, no LDS round trip): a lane that has no source keeps `old`.
template <int CTRL, int ROW_MASK, int BANK_MASK> __device__ __forceinline__ int32_t hw_dpp(int32_t old, int32_t v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int32_t hw_min(int32_t a, int32_t b) { return a < b ? a : b; }

// inclusive prefix-min over the 64 lanes: three single shifts inside the 16-lane rows, shifts by 4 and 8 for the upper
// banks, then lane 15 / lane 31 broadcast into the following rows
__device__ __forceinline__ int32_t hw_prefix_min(int32_t v)
{
    constexpr int32_t I = 0x7fffffff;
    int32_t r = hw_min(v, hw_dpp<0x111, 0xf, 0xf>(I, v));                // row_shr:1
    r = hw_min(r, hw_dpp<0x112, 0xf, 0xf>(I, v));                        // row_shr:2
    r = hw_min(r, hw_dpp<0x113, 0xf, 0xf>(I, v));                        // row_shr:3
    r = hw_min(r, hw_dpp<0x114, 0xf, 0xe>(I, r));                        // row_shr:4, banks 1-3
    r = hw_min(r, hw_dpp<0x118, 0xf, 0xc>(I, r));                        // row_shr:8, banks 2-3
    r = hw_min(r, hw_dpp<0x142, 0xa, 0xf>(I, r));                        // row_bcast:15 into rows 1 and 3
    r = hw_min(r, hw_dpp<0x143, 0xc, 0xf>(I, r));                        // row_bcast:31 into rows 2 and 3
    return r;
}

// Rows 0..n of one banded matrix.  Cell (i, j): query base i (1-based; reversed order when qrev), target base j
// (t0 + j - 1, or t0 - (j - 1) when trev), j in [0, m].  Diagonal index d = j - i + off.  cur[b] = last row on return.
// TRACE: trace[(i * NB + b) * 2 + {0,1}] = ballots "vertical step optimal" / "horizontal step optimal" of row i.
// EARLY: every 32 rows the smallest value of the row is compared with kstop -- row minima never decrease from one row to
// the next, so once a row exceeds kstop no end cell can be <= kstop: returns false at once (cur is then meaningless).
template <int NB, bool TRACE, bool EARLY>
__device__ __forceinline__ bool hw_band_rows(const uint8_t *Q, int32_t n, bool qrev, const uint8_t *T, int32_t t0, bool trev, int32_t m,
                                             int32_t off, bool topzero, uint64_t *trace, int lane, int32_t (&cur)[NB], int32_t kstop)
{
    // No per-cell range checks in the loop: columns j < 0 start at "infinity" and stay there by themselves (all their
    // predecessors are columns < 0, values are clamped), columns j > m hold junk that never reaches a column <= m
    // (information only moves to the same or a larger column), and the callers read columns 1..m only.  T is padded on
    // both sides (HW_TPAD bytes), so the base index t0 +- (j - 1) is always inside the buffer.
    int32_t tix[NB];
    const int32_t tdir = trev ? -1 : 1;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int32_t j = lane + 64 * b - off;
        cur[b] = (j >= 0 && j <= m) ? (topzero ? 0 : j) : HW_INF;
        tix[b] = t0 + tdir * j;                                          // cell (i, c) compares target base c - 1; row 1: c - 1 = d - off
    }
    int32_t qi = qrev ? n - 1 : 0;
    const int32_t qdir = qrev ? -1 : 1;
    for (int32_t i = 1; i <= n; ++i) {
        const int32_t qc = Q[qi];                                       // wave-uniform LDS read
        qi += qdir;
        int32_t nw[NB];
        int32_t run = HW_INF;                                           // min of a[d'] - d' over the blocks already done
        int32_t left_in = HW_INF;                                       // new value of the last diagonal of the previous block
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int32_t d = lane + 64 * b;
            const int32_t nxt0 = b + 1 < NB ? __builtin_amdgcn_readfirstlane(cur[b + 1 < NB ? b + 1 : b]) : HW_INF;
            const int32_t up = hw_dpp<0x130, 0xf, 0xf>(nxt0, cur[b]);    // wave_shl:1: lane l takes lane l + 1, lane 63 keeps nxt0
            const int32_t tc = T[tix[b]];
            tix[b] += tdir;
            const int32_t diag = cur[b] + (tc != qc ? 1 : 0);           // (i-1, j-1) is diagonal d of the previous row
            const int32_t a = hw_min(hw_min(diag, up + 1), HW_INF);
            const int32_t pm = hw_min(hw_prefix_min(a - d), run);
            const int32_t v = hw_min(pm + d, HW_INF);
            run = __builtin_amdgcn_readlane(pm, 63);
            if (TRACE) {
                const int32_t left = hw_dpp<0x138, 0xf, 0xf>(left_in, v);   // wave_shr:1: lane l takes lane l - 1, lane 0 keeps left_in
                const uint64_t m_up = __ballot(up + 1 == v);
                const uint64_t m_left = __ballot(left + 1 == v);
                if (lane == 0) {
                    trace[((size_t)i * NB + b) * 2] = m_up;
                    trace[((size_t)i * NB + b) * 2 + 1] = m_left;
                }
                left_in = __builtin_amdgcn_readlane(v, 63);
            }
            nw[b] = v;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) cur[b] = nw[b];
        if (EARLY && (i & 31) == 0) {
            int32_t rowmin = cur[0];
#pragma unroll
            for (int b = 1; b < NB; ++b) rowmin = hw_min(rowmin, cur[b]);
            if (wave_min_i32(rowmin) > kstop) return false;            // wave-uniform (junk columns can only delay this)
        }
    }
    return true;
}

// out[5 p ..] = distance (-1: > k, -3: band does not fit), start, end, leading insertion run, trailing insertion run.
// NBA blocks of 64 diagonals for phase A (max(len(t) - len(q), 0) + 2 k + 1 diagonals), NB for phases B and C (2 k + 1).
// grid = any number of 64-thread blocks (pairs are dealt round-robin); trace: (maxlen + 1) * NB * 2 words per block;
// dynamic LDS = 2 * lds_stride + 2 * HW_TPAD bytes (lds_stride >= maxlen, multiple of 8): query, pad, target, pad.
template <int NBA, int NB>
__global__ __launch_bounds__(64) void k_hw_path(DevStore S, const uint32_t *__restrict__ pq, const uint32_t *__restrict__ pt, const int32_t *__restrict__ pk,
                                                 uint32_t n_pairs, uint64_t *__restrict__ trace_all, uint32_t trace_rows,
                                                 uint32_t lds_stride, int32_t *__restrict__ out)
{
    extern __shared__ uint8_t hw_lds[];
    uint8_t *Q = hw_lds, *T = hw_lds + lds_stride + HW_TPAD;
    const int lane = threadIdx.x;
    uint64_t *trace = trace_all + (size_t)blockIdx.x * trace_rows * NB * 2;
    const uint64_t *planes = S.planes;
    const uint32_t nseq = S.n;
    // pairs are dealt round-robin: blockIdx.x, blockIdx.x + gridDim.x, ... (the loop variable lives in an SGPR, every
    // branch below is wave-uniform by construction)
    for (uint32_t pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        const uint32_t iq = pq[pair], it = pt[pair];
        const int32_t n = uniform_i32(S.lens[iq]), m = uniform_i32(S.lens[it]);
        const int32_t k = uniform_i32(pk[pair]);
        int32_t r_h = -1, r_start = -1, r_end = -1, r_lead = 0, r_trail = 0;
        const int32_t delta = m - n;
        const int32_t width_a = delta + 2 * k + 1, width_b = 2 * k + 1;
        if (n > 0 && m > 0 && k >= 0 && delta >= -k) {
            if (width_a > 64 * NBA || width_b > 64 * NB || n + 1 > (int32_t)trace_rows || n > (int32_t)lds_stride || m > (int32_t)lds_stride) r_h = -3;
            else {
                __syncthreads();                                        // previous pair's readers are done with the LDS
                for (int32_t x = lane; x < n; x += 64) {
                    const uint64_t lo = planes[((size_t)(x >> 6) * nseq + iq) * 2], hi = planes[((size_t)(x >> 6) * nseq + iq) * 2 + 1];
                    Q[x] = (uint8_t)(((lo >> (x & 63)) & 1) | (((hi >> (x & 63)) & 1) << 1));
                }
                for (int32_t x = lane; x < m; x += 64) {
                    const uint64_t lo = planes[((size_t)(x >> 6) * nseq + it) * 2], hi = planes[((size_t)(x >> 6) * nseq + it) * 2 + 1];
                    T[x] = (uint8_t)(((lo >> (x & 63)) & 1) | (((hi >> (x & 63)) & 1) << 1));
                }
                __syncthreads();
                int32_t cur_a[NBA], cur[NB];
                // phase A: distance and first end column
                const bool alive = hw_band_rows<NBA, false, true>(Q, n, false, T, 0, false, m, k, true, nullptr, lane, cur_a, k);
                int32_t h = HW_INF;
#pragma unroll
                for (int b = 0; b < NBA; ++b) {
                    const int32_t j = n + lane + 64 * b - k;
                    if (alive && j >= 1 && j <= m && cur_a[b] < h) h = cur_a[b];
                }
                h = wave_min_i32(h);
                if (h <= k) {
                    int32_t e = HW_INF;
#pragma unroll
                    for (int b = 0; b < NBA; ++b) {
                        const int32_t j = n + lane + 64 * b - k;
                        if (j >= 1 && j <= m && cur_a[b] == h && j < e) e = j;
                    }
                    const int32_t end = wave_min_i32(e) - 1;
                    // phase B: smallest start whose global distance to target[start..end] is h
                    hw_band_rows<NB, false, false>(Q, n, true, T, end, true, end + 1, k, false, nullptr, lane, cur, k);
                    int32_t pl = -1;
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const int32_t j = n + lane + 64 * b - k;
                        if (j >= 1 && j <= end + 1 && cur[b] == h && j > pl) pl = j;
                    }
                    pl = wave_max_i32(pl);
                    if (pl < 1) r_h = -4;                               // cannot happen (the optimum is attained by some start)
                    else {
                        const int32_t start = end - (pl - 1), ms = pl;
                        // phase C: decision bits of the global alignment query vs target[start..end], then the walk
                        hw_band_rows<NB, true, false>(Q, n, false, T, start, false, ms, k, false, trace, lane, cur, k);
                        // the walk: every lane follows the same cells (uniform addresses, one broadcast load per step)
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");      // lane 0's decision words, read by all lanes
                        int32_t trail = 0;
                        int32_t i = n, j = ms;
                        bool at_end = true;
                        while (i > 0 && j > 0) {
                            const int32_t d = j - i + k;
                            const size_t w = ((size_t)i * NB + (size_t)(d >> 6)) * 2;
                            const uint64_t mu = __hip_atomic_load(trace + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const uint64_t ml = __hip_atomic_load(trace + w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((mu >> (d & 63)) & 1) { --i; if (at_end) ++trail; }
                            else {
                                at_end = false;
                                if ((ml >> (d & 63)) & 1) --j;
                                else { --i; --j; }
                            }
                        }
                        r_h = h; r_start = start; r_end = end;
                        r_lead = j == 0 ? i : 0;
                        r_trail = trail;
                    }
                }
            }
        }
        if (lane == 0) {
            int32_t *o = out + (size_t)pair * 5;
            o[0] = r_h; o[1] = r_start; o[2] = r_end; o[3] = r_lead; o[4] = r_trail;
        }
    }
}

}  // namespace isocon
