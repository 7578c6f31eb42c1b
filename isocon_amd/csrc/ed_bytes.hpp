// ed_bytes.hpp -- global edit distance over the sequences' OWN BYTES, for the pairs the 2-bit store cannot serve.
//
// edlib compares whatever characters it is given (/root/reference/modules/edlib_alignment_module.py:111,
// /root/reference/modules/nearest_neighbor_graph.py:105): 'N' matches 'N', 'a' does not match 'A'.  A set with more than four distinct
// symbols keeps, beside its 2-bit planes, the bytes themselves and a flag per sequence that holds a symbol outside the planes' map
// ("exceptional" sequence).  Every pair with an exceptional sequence is aligned here; all other pairs never leave the bit-vector kernels.
//
// One wavefront per pair, the dynamic-programming matrix in strips of 64 rows: lane r owns row i0 + 1 + r of the strip and walks it
// column by column, one step behind lane r - 1 (at step t it is at column jlo + t - r), so the three neighbours of a cell are the lane's
// own last value (left), the value lane r - 1 produced one step ago (up) and the one it produced two steps ago (diagonal): one cross-lane
// move per step (a DPP wave shift, no LDS) carries (value, text symbol) from lane r - 1 to lane r.  Lane 0 is fed from the last row of the strip above, which lives in
// a per-wavefront row buffer in global memory, read and written in coalesced blocks of 64 columns.
//
// Threshold k: only the columns jlo = i0 + 1 - k .. jhi = i0 + 64 + k of a strip are computed -- a superset of Ukkonen's band |i - j| <= k;
// cells outside count as "infinite".  Every alignment of cost <= k stays inside, every alignment inside costs at least the distance: the
// result is the distance when that is <= k, and -1 otherwise (edlib's answer above k).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace isocon {

static constexpr uint32_t EB_INF = 0x7fffffu;          // travels as (value << 8 | symbol) in one dword

struct ByteStore {
    const uint8_t *bytes;          // the sequences, concatenated
    const uint64_t *off;           // n + 1 offsets, relative to `base`
    uint64_t base;
    const int32_t *lens;
};

// row_stride >= longest sequence + 130 dwords; one row per workgroup (= wavefront)
__global__ __launch_bounds__(64) void k_ed_bytes(ByteStore B, const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb, const int32_t *__restrict__ pk,
                                                 unsigned long long n_pairs, uint32_t *__restrict__ rowbuf, uint32_t row_stride, int32_t *__restrict__ out)
{
    const int32_t lane = (int32_t)threadIdx.x;
    uint32_t *row = rowbuf + (size_t)blockIdx.x * row_stride;
    for (unsigned long long p = blockIdx.x; p < n_pairs; p += gridDim.x) {
        const uint32_t a = pa[p], b = pb[p];
        const int32_t m = B.lens[a], n = B.lens[b];
        const int32_t big = m > n ? m : n, ad = m > n ? m - n : n - m;
        int32_t k = pk ? pk[p] : -1;
        if (k < 0 || k > big) k = big;
        if (ad > k) { if (lane == 0) out[p] = -1; continue; }
        if (m == 0 || n == 0) { if (lane == 0) out[p] = big; continue; }          // (big = ad <= k)
        const uint8_t *x = B.bytes + (B.off[a] - B.base), *y = B.bytes + (B.off[b] - B.base);
        // row 0 of the matrix: D[0][j] = j, as far as strip 0 reads it
        {
            const int32_t hi0 = n < 64 + k ? n : 64 + k;
            for (int32_t j = lane; j <= hi0; j += 64) row[j] = (uint32_t)j;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");          // the row buffer is this wavefront's own: a device-scope fence would write the L2 back (gfx950)
        uint32_t res = EB_INF;
        const int32_t n_strips = (m + 63) >> 6;
        for (int32_t s = 0; s < n_strips; ++s) {
            const int32_t i0 = s << 6;
            const int32_t jlo = i0 + 1 - k > 1 ? i0 + 1 - k : 1;
            const int32_t jhi = i0 + 64 + k < n ? i0 + 64 + k : n;
            const int32_t steps = jhi - jlo + 1 + 63;
            const int32_t i = i0 + 1 + lane;
            const bool rowact = i <= m;
            const uint32_t xi = rowact ? x[i - 1] : 0u;
            const bool edge = jlo == 1;                                    // the strip starts at the matrix' first column: D[i][0] = i
            uint32_t val = edge ? (uint32_t)i : EB_INF;                    // D[i][jlo - 1]
            uint32_t upprev = edge ? (uint32_t)(i - 1) : EB_INF;           // D[i - 1][jlo - 1]
            if (lane == 0 && !edge) upprev = row[jlo - 1];
            const bool full = i0 + 64 <= m;                                // the strip's last row feeds another strip
            uint32_t send = 0, blk = 0, oblk = EB_INF;
            for (int32_t t = 0; t < steps; ++t) {
                if ((t & 63) == 0) {
                    const int32_t j = jlo + t + lane;
                    const uint32_t pv = j <= jhi ? row[j] : EB_INF;
                    const uint32_t yv = j <= n ? y[j - 1] : 0u;
                    blk = (pv << 8) | yv;
                }
                const int feed = __builtin_amdgcn_readlane((int)blk, t & 63);
                const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(feed, (int)send, 0x138, 0xf, 0xf, false);      // wave_shr:1: lane r takes lane r - 1, lane 0 the feed
                const uint32_t up = recv >> 8, sym = recv & 255u;
                const int32_t j = jlo + t - lane;
                if (t >= lane && j <= jhi && rowact) {
                    uint32_t nv = upprev + (sym != xi ? 1u : 0u);
                    const uint32_t side = (up < val ? up : val) + 1u;
                    nv = nv < side ? nv : side;
                    nv = nv < EB_INF ? nv : EB_INF;
                    upprev = up;
                    val = nv;
                    if (i == m && j == n) res = nv;
                }
                send = (val << 8) | sym;
                if (full && t >= 63) {
                    // lane 63 finished column jlo + t - 63 of the strip's last row: gathered 64 at a time, written in one piece
                    const uint32_t v63 = (uint32_t)__builtin_amdgcn_readlane((int)val, 63);
                    const int32_t oi = (t - 63) & 63;
                    if (lane == oi) oblk = v63;
                    if (oi == 63 || t == steps - 1) {
                        const int32_t col = jlo + ((t - 63) & ~63) + lane;
                        if (lane <= oi && col <= jhi) row[col] = oblk;
                    }
                }
            }
            if (full) {
                // the columns the next strip reads beyond this one's: not computed, "infinite"
                const int32_t col = jhi + 1 + lane;
                if (col <= n) row[col] = EB_INF;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");          // the row buffer is this wavefront's own: a device-scope fence would write the L2 back (gfx950)
        }
        res = (uint32_t)__shfl((int)res, (m - 1) & 63, 64);
        if (lane == 0) out[p] = res <= (uint32_t)k ? (int32_t)res : -1;
    }
}

// per sequence: how many bytes outside the map?  (one wavefront per 64 bases, as k_pack_planes)
__global__ __launch_bounds__(256) void k_exception_flags(const uint8_t *__restrict__ ascii, const uint64_t *__restrict__ offsets, uint64_t base, uint32_t n,
                                                        uint32_t nchunks, uint32_t *__restrict__ counts, uint32_t sym4, uint64_t w0)
{
    const uint64_t w = w0 + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (uint64_t)nchunks * n) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t chunk = (uint32_t)(w / n), seq = (uint32_t)(w % n);
    const uint64_t off = offsets[seq] - base, len = offsets[seq + 1] - offsets[seq];
    const uint64_t pos = (uint64_t)chunk * 64 + lane;
    bool bad = false;
    if (pos < len) {
        const uint32_t ch = ascii[off + pos];
        bad = ch != (sym4 & 255u) && ch != ((sym4 >> 8) & 255u) && ch != ((sym4 >> 16) & 255u) && ch != (sym4 >> 24);
    }
    const unsigned long long bm = __ballot(bad);
    if (bm != 0 && lane == 0) atomicAdd(counts + seq, (uint32_t)__popcll(bm));
}

}  // namespace isocon
