// ed_band.hpp -- banded bit-vector edit distance, one wavefront = one shared sequence x 64 lane sequences.
// Lane-level math: band_core.hpp.  Replaces edlib.align(..., mode="NW", task="distance", k=K)
// (/root/reference/modules/nearest_neighbor_graph.py:104-107; modules/edlib_alignment_module.py:111).
#pragma once
#include "band_core.hpp"
#include "common.hpp"

namespace isocon {

struct TileStats {
    uint32_t lanes_run;   // lanes that entered the DP
    uint32_t cols;        // columns the wave executed
    uint32_t live_cols;   // sum over 32-column halves of (lanes still undecided at the start of the half) x 32
};

// Runs one tile.  `shared` and `m` must be wave-uniform.  Returns per lane:
//   >= 0 : exact distance (<= k_req);  -1 : distance > k_req (certified);  -2 : undetermined, the tile's common
//   window could not certify k_req for this lane's length difference (caller re-tiles the pair).
//
// Window source (gfx950 issue costs, scripts/ubench/valu_rate2.hip: and/or/xor/not/add/sub/lshr on VGPR operands
// issue in ~2.3 cycles per wave-instruction, anything with an SGPR operand and all VOP3-only integer ops in ~4.2):
//   LDS == false : the pattern window lives in SGPRs and is slid by the scalar unit (any W);
//   LDS == true  : W == 1, the workgroup has tabulated in LDS, for every bit offset o in [-63, m+129) of the shared
//                  sequence, the 32-row match masks of the four bases (tab[o+63] = {Eq_A, Eq_C, Eq_G, Eq_T}, zero on
//                  rows outside the sequence).  A column is ONE ds_read2_b32 at a per-lane address (entry, base) and
//                  (entry + 32, base): the 64-bit Eq vector arrives ready-made and no Eq logic, bit extraction or
//                  virtual-row mask is left on the VALU.  Lane texts come from the interleaved store S.il
//                  (2 bits per base) so that the address is shift + and_or.
template <int W, bool LDS = false>
__device__ __forceinline__ int32_t band_tile_run(const DevStore &S, uint32_t shared, int32_t m, uint32_t tid,
                                                 int32_t n, int32_t k_req, bool active, TileStats *st,
                                                 const uint4 *tab = nullptr)
{
    static_assert(!LDS || W == 1, "the LDS window table is built for the 64-row band");
    const int32_t d = m - n;
    const int32_t ad = d < 0 ? -d : d;
    active = active && k_req >= 0 && ad <= k_req;
    // an empty sequence on either side: the distance is the other length (no DP needed)
    const bool trivial = active && (m == 0 || n == 0);
    active = active && !trivial;
    int32_t a0 = wave_min_i32(active ? lane_emin(d, k_req) : 0);
    if (a0 < -(64 * W - 1)) a0 = -(64 * W - 1);
    a0 = uniform_i32(a0);
    const int32_t n_min = uniform_i32(wave_min_i32(active ? n : 0x7fffffff));
    const int32_t n_max = uniform_i32(wave_max_i32(active ? n : 0));
    if (st) { st->lanes_run = 0; st->cols = 0; st->live_cols = 0; }
    if (n_max == 0) return trivial ? ad : -1;

    const LaneGeom g = lane_geom<W>(d, k_req, a0);
    bool live = active && g.k_eff >= 0;
    const int32_t fail = (active && g.k_eff < k_req) ? -2 : -1;
    const int32_t k_eff = live ? g.k_eff : -1;
    int32_t res = trivial ? ad : fail;
    const int32_t nv = -a0;
    int32_t bstar = g.bstar;
    if (bstar < 0) bstar = 0;
    if (bstar > 64 * W - 1) bstar = 64 * W - 1;

    BandLane<W> L;
    band_init<W>(L, nv, bstar);

    const uint64_t *planes = S.planes;
    const uint32_t nseq = S.n;
    const int32_t nchunks = (int32_t)S.nchunks;
    auto chunk_lo = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + shared) * 2] : 0; };
    auto chunk_hi = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + shared) * 2 + 1] : 0; };

    uint64_t NL[W], NH[W], VM[W], FL = 0, FH = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        if (!LDS) {
            NL[i] = ~stream64(chunk_lo, a0 + 64 * i);
            NH[i] = ~stream64(chunk_hi, a0 + 64 * i);
        }
        VM[i] = valid_word(nv, i);
    }
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    // LDS byte address of the table entry of column 1 (entry index a0 + 63); entries are 16 B
    const uint32_t tbase = LDS ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint4 *)tab + (uint32_t)(a0 + 63) * 16u : 0u;

    const ulonglong2 *P2 = reinterpret_cast<const ulonglong2 *>(LDS ? S.il : planes);
    ulonglong2 tnext = P2[(size_t)tid];
    int32_t cols = 0;
    uint32_t live_cols = 0;
    bool stop = false;
    for (int32_t c = 0; 64 * c < n_max && !stop; ++c) {
        if (!LDS) {
            FL = ~stream64(chunk_lo, a0 + 64 * W + 64 * c);
            FH = ~stream64(chunk_hi, a0 + 64 * W + 64 * c);
        }
        const ulonglong2 tcur = tnext;
        if (64 * (c + 1) < n_max) tnext = P2[(size_t)(c + 1) * nseq + tid];
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int32_t jb = 64 * c + 32 * h;
            if (jb >= n_max) break;
            const int32_t cnt = (n_max - jb) < 32 ? (n_max - jb) : 32;
            // planes: wl / wh = 32 columns of the low / high bit-plane; interleaved: wl = columns 0..15, wh = 16..31
            const uint32_t wl = LDS ? (uint32_t)(h ? tcur.y : tcur.x) : (uint32_t)(h ? (tcur.x >> 32) : tcur.x);
            const uint32_t wh = LDS ? (uint32_t)((h ? tcur.y : tcur.x) >> 32) : (uint32_t)(h ? (tcur.y >> 32) : tcur.y);
            const bool fast = cnt == 32 && jb >= nv && jb + 32 <= n_min;
            live_cols += (uint32_t)__popcll(__ballot(live)) * (uint32_t)cnt;
            if (fast) {
                if constexpr (LDS) {
                    uint32_t blk;      // table address of this 32-column block, in a VGPR (SGPR operands cost issue cycles)
                    asm volatile("v_mov_b32 %0, %1" : "=v"(blk) : "s"(tbase + (uint32_t)jb * 16u));
#pragma unroll 32
                    for (int jj = 0; jj < 32; ++jj) {
                        const uint32_t w2 = jj < 16 ? wl : wh;
                        const int sh = 2 * (jj & 15);
                        const uint32_t t = sh >= 2 ? (w2 >> (sh - 2)) : (w2 << 2);
                        lds_u32 *p = (lds_u32 *)(uintptr_t)((t & 0xCu) | blk);
                        uint64_t EQ[1];
                        EQ[0] = ((uint64_t)p[4 * jj + 128] << 32) | p[4 * jj];
                        band_step_eq<1>(L, EQ);
                    }
                } else {
#pragma unroll 32
                    for (int jj = 0; jj < 32; ++jj) {
                        const uint32_t slo = (uint32_t)__builtin_amdgcn_sbfe((int)wl, jj, 1);
                        const uint32_t shi = (uint32_t)__builtin_amdgcn_sbfe((int)wh, jj, 1);
                        band_step<W, false>(L, NL, NH, VM, slo, shi);
                        window_slide<W>(NL, NH, VM, FL, FH);
                    }
                }
                cols = jb + 32;
                if (live && n == cols) {
                    const int32_t dv = band_diag_value<W>(L, nv, cols);
                    res = dv <= k_eff ? dv : fail;
                    live = false;
                }
            } else {
#pragma unroll 1
                for (int jj = 0; jj < cnt; ++jj) {
                    if constexpr (LDS) {
                        const uint32_t b = ((jj < 16 ? wl : wh) >> (2 * (jj & 15))) & 3u;
                        lds_u32 *p = (lds_u32 *)(uintptr_t)(tbase + (uint32_t)(jb + jj) * 16u + b * 4u);
                        uint64_t EQ[1];
                        EQ[0] = ((uint64_t)p[128] << 32) | p[0];      // already zero on virtual rows
                        band_step_eq<1>(L, EQ);
                    } else {
                        const uint32_t slo = (uint32_t)__builtin_amdgcn_sbfe((int)wl, jj, 1);
                        const uint32_t shi = (uint32_t)__builtin_amdgcn_sbfe((int)wh, jj, 1);
                        band_step<W, true>(L, NL, NH, VM, slo, shi);
                        window_slide<W>(NL, NH, VM, FL, FH);
                    }
                    if (live && n == jb + jj + 1) {
                        const int32_t dv = band_diag_value<W>(L, nv, n);
                        res = dv <= k_eff ? dv : fail;
                        live = false;
                    }
                }
                cols = jb + cnt;
            }
            if (live) {
                // the value on the final diagonal never decreases: above k_eff means the pair is out
                const int32_t dv = band_diag_value<W>(L, nv, cols);
                if (dv > k_eff) live = false;
            }
            if (__ballot(live) == 0) { stop = true; break; }
        }
    }
    if (st) {
        st->lanes_run = (uint32_t)__popcll(__ballot(active && g.k_eff >= 0));
        st->cols = (uint32_t)cols;
        st->live_cols = live_cols;
    }
    return res;
}

// Explicit tile list: tile t = shared sequence tile_shared[t] x 64 lanes lane_ids[64t..] (0xffffffff = empty)
// with per-lane thresholds lane_k.  One wave per tile, 4 tiles per 256-thread block.
template <int W>
__global__ __launch_bounds__(256) void k_ed_band_tiles(DevStore S, const uint32_t *__restrict__ tile_shared,
                                                        const uint32_t *__restrict__ lane_ids,
                                                        const int32_t *__restrict__ lane_k,
                                                        int32_t *__restrict__ out, uint32_t n_tiles)
{
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (wave >= n_tiles) return;
    const uint32_t shared = (uint32_t)uniform_i32((int32_t)tile_shared[wave]);
    const int32_t m = S.lens[shared];
    const uint32_t id = lane_ids[(size_t)wave * 64 + lane];
    const bool valid = id != 0xffffffffu;
    const uint32_t tid = valid ? id : shared;
    const int32_t n = S.lens[tid];
    const int32_t k = valid ? lane_k[(size_t)wave * 64 + lane] : -1;
    // lanes the common window could not certify (-2) are re-run among themselves (see nn_process_tile)
    int32_t r = -1;
    bool pending = valid;
    for (int round = 0; round < 64; ++round) {
        const int32_t rr = band_tile_run<W>(S, shared, m, tid, n, k, pending, nullptr);
        if (pending) r = rr;
        pending = pending && rr == -2;
        if (__ballot(pending) == 0) break;
    }
    out[(size_t)wave * 64 + lane] = r;
}

}  // namespace isocon
