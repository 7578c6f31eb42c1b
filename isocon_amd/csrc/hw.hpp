// hw.hpp -- infix ("HW") edit distance with location and the terminal insertion runs of its path,
// bit-parallel: device side of edlib.align(q, t, mode="HW", task="path", k) as consumed by the candidate-vs-candidate graph
// of the statistical-test phase (/root/reference/modules/end_invariant_functions.py:593-620 edlib_traceback, :622-681
// get_all_NN).  Lane-level math and the three passes (LOCATE, START, TRACE): hw_core.hpp.
//
// One wavefront = one TILE = one shared query (rows; its sliding 64 W-row window is wave-uniform and lives in SGPRs, slid by
// the scalar unit) x up to 64 lane targets (columns; two text bits per lane and column).  The candidate graph asks for every
// candidate against the ~1 200 candidates of its length window, so tiles are full; a column of 64 pairs costs ~45 W vector
// instructions instead of the 64 W x 27 of the cell-per-lane kernel this replaces (5.77 M C3 pairs: 1.73 s -> see DESIGN.md).
//   k_hw_locate<W>  pass LOCATE for all pairs            -> (h or -1, end) per pair
//   k_hw_finish<W>  passes START and TRACE + the walk for the hits -> start, leading / trailing insertion run
// HBM traffic: the packed sequences in, 8 + 20 B per pair out, and for the hits 16 W bytes per column and pair of stored
// VP / HP vectors (written as 512-B lines, read back once by the walk).
#pragma once
#include "common.hpp"
#include "hw_core.hpp"

namespace isocon {

// the 32 text bits of positions p .. p + 31 of one plane of sequence `id` (p may be negative or run past the end: zeros)
__device__ __forceinline__ uint32_t hw_text32(const uint32_t *__restrict__ pw, uint32_t nseq, uint32_t nchunks, uint32_t id, int plane, int32_t p)
{
    if (p <= -32) return 0;
    const int32_t pp = p < 0 ? 0 : p;
    const uint32_t d = (uint32_t)pp >> 5;
    auto dword = [&](uint32_t dd) -> uint32_t {
        const uint32_t c = dd >> 1;
        return c < nchunks ? pw[(((size_t)c * nseq + id) * 2 + (uint32_t)plane) * 2 + (dd & 1u)] : 0u;
    };
    const uint32_t w0 = dword(d), w1 = dword(d + 1);
    const uint32_t s = (uint32_t)pp & 31u;
    const uint32_t v = __builtin_amdgcn_alignbit(w1, w0, s);
    return p < 0 ? v << (uint32_t)(-p) : v;
}

struct HwTileIn {
    const uint32_t *tile_q;      // query of every tile
    const uint32_t *lane_pair;   // [tiles][64] pair index, 0xffffffff = empty lane
    const uint32_t *pt;          // target of every pair
    const int32_t *pk;           // threshold of every pair
    uint32_t n_tiles;
    const uint32_t *tile_list;   // the launch's tiles (indices into tile_q / lane_pair), or nullptr: tiles 0 .. n_tiles - 1
    uint32_t *flags;             // or nullptr: bit HW_FLAG_STATUS is set when a pair ends with an internal status (< -1)
};
static constexpr uint32_t HW_FLAG_STATUS = 16u;

// out_he[2 p] = distance (-1: above k, -3: the tile's band does not fit W words), out_he[2 p + 1] = end
template <int W>
__global__ __launch_bounds__(256) void k_hw_locate(DevStore S, HwTileIn in, int32_t *__restrict__ out_he)
{
    const uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (slot >= in.n_tiles) return;
    const uint32_t tile = in.tile_list ? (uint32_t)uniform_i32((int32_t)in.tile_list[slot]) : slot;
    const uint32_t q = (uint32_t)uniform_i32((int32_t)in.tile_q[tile]);
    const int32_t P = uniform_i32(S.lens[q]);
    const uint32_t pair = in.lane_pair[(size_t)tile * 64 + lane];
    const bool has = pair != 0xffffffffu;
    const uint32_t tid = has ? in.pt[pair] : q;
    const int32_t k = has ? in.pk[pair] : 0;
    const int32_t m = S.lens[tid];
    const bool valid = has && P > 0 && m > 0 && k >= 0 && m - P >= -k;
    const int32_t dmax = uniform_i32(wave_max_i32(valid ? m - P : -(1 << 30)));
    const int32_t kmax = uniform_i32(wave_max_i32(valid ? k : 0));
    const int32_t mmax = uniform_i32(wave_max_i32(valid ? m : 0));
    int32_t r_h = -1, r_end = -1;
    if (dmax > -(1 << 30)) {
        if (hw_locate_rows(dmax, kmax) > 64 * W) r_h = -3;
        else {
            HwTile T;
            T.P = P; T.a0 = hw_locate_a0(dmax, kmax); T.ncols_max = mmax; T.jx = P - kmax > 1 ? P - kmax : 1;
            HwLane ln;
            ln.ncols = valid ? m : 0; ln.k = k; ln.h = 0;
            const uint64_t *planes = S.planes;
            const uint32_t nseq = S.n;
            const int32_t nchunks = (int32_t)S.nchunks;
            auto chunk_lo = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + q) * 2] : 0; };
            auto chunk_hi = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + q) * 2 + 1] : 0; };
            auto plo = [&](int32_t off) { return stream64(chunk_lo, off); };
            auto phi = [&](int32_t off) { return stream64(chunk_hi, off); };
            const ulonglong2 *P2 = reinterpret_cast<const ulonglong2 *>(planes);
            ulonglong2 tcur = {0, 0};
            auto text = [&](int32_t jb, uint32_t &wl, uint32_t &wh) {
                if ((jb & 63) == 0) tcur = P2[(size_t)(jb >> 6) * nseq + tid];       // columns <-> target positions, 64 per load
                wl = (uint32_t)((jb & 32) ? (tcur.x >> 32) : tcur.x);
                wh = (uint32_t)((jb & 32) ? (tcur.y >> 32) : tcur.y);
            };
            auto any_live = [](bool live) { return __ballot(live) != 0; };
            auto nosink = [](int32_t, int, uint64_t, uint64_t) {};
            hw_run<W, HW_LOCATE>(T, ln, plo, phi, text, any_live, nosink);
            if (valid && ln.r_h <= k) { r_h = ln.r_h; r_end = ln.r_end; }
        }
    }
    if (has) { out_he[(size_t)pair * 2] = r_h; out_he[(size_t)pair * 2 + 1] = r_end; }
    if (in.flags && __ballot(has && r_h < -1) != 0 && lane == 0) atomicOr(in.flags, HW_FLAG_STATUS);
}

// Hits only: he[2 p] = h, he[2 p + 1] = end.  out[5 p ..] = h, start, end, leading insertion run, trailing insertion run
// (h < -1: internal status).  grid = any number of 64-thread blocks (tiles are dealt round-robin); trace: per block
// (trace_cols + 1) x 2 W x 64 words (W > 2) or (trace_cols / HW_SEG + 2) x 2 W x 64 (checkpoints, W <= 2).
template <int W>
__global__ __launch_bounds__(64) void k_hw_finish(DevStore S, HwTileIn in, const int32_t *__restrict__ he, uint64_t *__restrict__ trace_all,
                                                   uint32_t trace_cols, int32_t *__restrict__ out)
{
    const int lane = threadIdx.x;
    // the workgroup's store: checkpoints (W <= 2) or every column's vectors (nn_host's hw_finish_store_bytes is the same formula)
    uint64_t *trace = trace_all + (size_t)blockIdx.x * (W <= 2 ? ((size_t)trace_cols / HW_SEG + 2) : ((size_t)trace_cols + 1)) * 2 * W * 64;
    const uint64_t *planes = S.planes;
    const uint32_t *pw = reinterpret_cast<const uint32_t *>(planes);
    const uint32_t nseq = S.n;
    const int32_t nchunks = (int32_t)S.nchunks;
    for (uint32_t slot = blockIdx.x; slot < in.n_tiles; slot += gridDim.x) {
        const uint32_t tile = in.tile_list ? (uint32_t)uniform_i32((int32_t)in.tile_list[slot]) : slot;
        const uint32_t q = (uint32_t)uniform_i32((int32_t)in.tile_q[tile]);
        const int32_t P = uniform_i32(S.lens[q]);
        const uint32_t pair = in.lane_pair[(size_t)tile * 64 + lane];
        const bool has = pair != 0xffffffffu;
        const uint32_t tid = has ? in.pt[pair] : q;
        const int32_t k = has ? in.pk[pair] : 0;
        const int32_t h = has ? he[(size_t)pair * 2] : -1;
        const int32_t end = has ? he[(size_t)pair * 2 + 1] : -1;
        const bool valid = has && h >= 0 && end >= 0;
        const int32_t kmax = uniform_i32(wave_max_i32(valid ? k : 0));
        int32_t r0 = valid ? h : -1, r_start = -1, r_lead = 0, r_trail = 0;
        auto chunk_lo = [&](int32_t ci) -> uint64_t { return ci >= 0 && ci < nchunks ? planes[((size_t)ci * nseq + q) * 2] : 0; };
        auto chunk_hi = [&](int32_t ci) -> uint64_t { return ci >= 0 && ci < nchunks ? planes[((size_t)ci * nseq + q) * 2 + 1] : 0; };
        auto any_live = [](bool live) { return __ballot(live) != 0; };
        auto nosink = [](int32_t, int, uint64_t, uint64_t) {};
        if (2 * kmax + 1 > 64 * W) r0 = valid ? -3 : r0;
        else if (__ballot(valid) != 0) {
            // ---- START: reversed query against the reversed prefix t[0..end] ----
            HwTile T;
            T.P = P; T.a0 = -kmax; T.jx = P - kmax > 1 ? P - kmax : 1;
            HwLane ln;
            int32_t nc = end + 1 < P + kmax ? end + 1 : P + kmax;
            ln.ncols = valid ? nc : 0; ln.k = k; ln.h = h;
            T.ncols_max = uniform_i32(wave_max_i32(ln.ncols));
            {
                auto plo = [&](int32_t off) { return stream64_rev(chunk_lo, P, off); };
                auto phi = [&](int32_t off) { return stream64_rev(chunk_hi, P, off); };
                auto text = [&](int32_t jb, uint32_t &wl, uint32_t &wh) {       // column c <-> target position end - c
                    const int32_t p0 = end - jb - 31;
                    wl = __builtin_bitreverse32(hw_text32(pw, nseq, (uint32_t)nchunks, tid, 0, p0));
                    wh = __builtin_bitreverse32(hw_text32(pw, nseq, (uint32_t)nchunks, tid, 1, p0));
                };
                hw_run<W, HW_START>(T, ln, plo, phi, text, any_live, nosink);
            }
            const bool ok = valid && ln.r_pl >= 1;
            if (valid && !ok) r0 = -4;
            const int32_t start = ok ? end - (ln.r_pl - 1) : 0;
            const int32_t ms = ok ? end - start + 1 : 0;
            const bool packed = hw_packable(W, h, kmax);           // (per lane: its own distance decides)
            const int32_t pshift = kmax - h - 1;
            // ---- TRACE: query against t[start..end] ----
            ln.ncols = ok && ms <= (int32_t)trace_cols ? ms : 0;
            if (ok && ms > (int32_t)trace_cols) r0 = -7;
            T.ncols_max = uniform_i32(wave_max_i32(ln.ncols));
            T.jx = 1;
            if (W <= 2) {
                // One and two words of band (thresholds up to 63): only CHECKPOINTS of the band state leave the registers (every HW_SEG-th
                // column, 1 B per column and word instead of 16), and the walk runs segment by segment from the end on columns recomputed
                // from their checkpoint into LDS (hw_core.hpp).  Measured on the candidate graph of C3 (2.4 M hits): 54 -> see DESIGN.md.
                extern __shared__ uint64_t seg[];          // [HW_SEG][2][W][64]
                T.jx = uniform_i32(wave_min_i32(ln.ncols > 0 ? ln.ncols : 0x7fffffff));
                auto plo = [&](int32_t off) { return stream64(chunk_lo, off); };
                auto phi = [&](int32_t off) { return stream64(chunk_hi, off); };
                auto text = [&](int32_t jb, uint32_t &wl, uint32_t &wh) {       // column c <-> target position start + c
                    wl = hw_text32(pw, nseq, (uint32_t)nchunks, tid, 0, start + jb);
                    wh = hw_text32(pw, nseq, (uint32_t)nchunks, tid, 1, start + jb);
                };
                // (only start == 0 can have a leading insertion run, see below: the other lanes keep nothing)
                const bool need = ln.ncols > 0 && start == 0;
                auto keep = [&](int32_t c, int w, uint64_t vp, uint64_t vn) {
                    if (need) {
                        trace[(((size_t)c * 2) * W + w) * 64 + lane] = vp;
                        trace[(((size_t)c * 2 + 1) * W + w) * 64 + lane] = vn;
                    }
                };
                if (T.ncols_max > 0) hw_run<W, HW_TRACE_CK>(T, ln, plo, phi, text, any_live, keep);
                if (ln.ncols > 0) {
                    if (ln.r_final != h) r0 = -5;
                    else {
                        r_trail = ln.r_trail;
                        r_start = start;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const bool walking = need && r0 >= 0;
                int32_t wi = P, wj = ms;
                bool bad = false;
                for (int32_t sg = T.ncols_max > 0 ? (T.ncols_max - 1) / HW_SEG : -1; sg >= 0; --sg) {
                    const int32_t c0 = sg * HW_SEG;
                    const bool act = walking && !bad && wi > 0 && wj > c0;
                    if (__ballot(act) == 0) continue;                         // wave-uniform
                    BandLane<W> L2;
                    hw_trace_init<W>(T, L2);
                    if (sg > 0 && need) {
#pragma unroll
                        for (int w = 0; w < W; ++w) {
                            L2.VP[w] = trace[(((size_t)sg * 2) * W + w) * 64 + lane];
                            L2.VN[w] = trace[(((size_t)sg * 2 + 1) * W + w) * 64 + lane];
                        }
                    }
                    const uint32_t wl = hw_text32(pw, nseq, (uint32_t)nchunks, tid, 0, start + c0);
                    const uint32_t wh = hw_text32(pw, nseq, (uint32_t)nchunks, tid, 1, start + c0);
                    if (W == 1) {
                        // one word of band: the segment's vectors stay in registers and the walk is an unrolled loop over its columns
                        uint64_t VPs[HW_SEG][W], HPs[HW_SEG][W];
                        hw_trace_segment<W>(T, c0, L2, plo, phi, wl, wh, [&](int jj, int w, uint64_t vp, uint64_t hp) { VPs[jj][w] = vp; HPs[jj][w] = hp; });
                        if (act) bad = !hw_walk_segment_regs<W>(T.a0, c0, wi, wj, VPs, HPs);
                    } else {
                        hw_trace_segment<W>(T, c0, L2, plo, phi, wl, wh, [&](int jj, int w, uint64_t vp, uint64_t hp) {
                            seg[((jj * 2) * W + w) * 64 + lane] = vp;
                            seg[((jj * 2 + 1) * W + w) * 64 + lane] = hp;
                        });
                        if (act) bad = !hw_walk_segment<W>(T.a0, c0, wi, wj, [&](int jj, int which, int w) -> uint64_t { return seg[((jj * 2 + which) * W + w) * 64 + lane]; });
                    }
                }
                if (walking) {
                    if (bad) r0 = -6;
                    else r_lead = wj == 0 ? wi : 0;
                }
            } else {
                auto plo = [&](int32_t off) { return stream64(chunk_lo, off); };
                auto phi = [&](int32_t off) { return stream64(chunk_hi, off); };
                auto text = [&](int32_t jb, uint32_t &wl, uint32_t &wh) {       // column c <-> target position start + c
                    wl = hw_text32(pw, nseq, (uint32_t)nchunks, tid, 0, start + jb);
                    wh = hw_text32(pw, nseq, (uint32_t)nchunks, tid, 1, start + jb);
                };
                // A path that begins with a query-only step could trade it for a diagonal step into t[start - 1] at no extra cost,
                // so with the SMALLEST start of distance h only start == 0 can have a leading insertion run: every other lane
                // needs neither the stored columns nor the walk (the trailing run is read off the last column).
                auto sink = [&](int32_t j, int w, uint64_t vp, uint64_t hp) {
                    if (start == 0) {
                        if (packed) trace[(((size_t)j * 2) * W + w) * 64 + lane] = hw_pack(vp, hp, pshift);
                        else {
                            trace[(((size_t)j * 2) * W + w) * 64 + lane] = vp;
                            trace[(((size_t)j * 2 + 1) * W + w) * 64 + lane] = hp;
                        }
                    }
                };
                hw_run<W, HW_TRACE>(T, ln, plo, phi, text, any_live, sink);
            if (ln.ncols > 0) {
                if (ln.r_final != h) r0 = -5;
                else {
                    r_trail = ln.r_trail;
                    r_start = start;
                }
            }
            // ---- the walk: every lane reads back its OWN stores (same thread, same addresses: program order suffices, no cache
            // maintenance); columns are visited in descending order, so the words of the next PF columns are requested
            // together, on the assumption that the path keeps its diagonal.  Measured at C3's candidate graph (2.4 M hits):
            // the two passes 15 ms, the column stores +22 ms, the walk +18 ms -- both are HBM traffic (16 B per column and hit).
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (ln.ncols > 0 && r0 >= 0 && start == 0) {
                constexpr int PF = W <= 2 ? 8 : 2;
                auto ld = [&](int32_t jj, int which, int w) -> uint64_t {
                    if (packed) return hw_unpack(trace[(((size_t)jj * 2) * W + w) * 64 + lane], which, pshift);
                    return trace[(((size_t)jj * 2 + which) * W + w) * 64 + lane];
                };
                int32_t i = P, j = ms;
                bool bad = false;
                while (i > 0 && j > 0 && !bad) {
                    const int32_t hb0 = i - T.a0 - j, vb0 = hb0 - 1;
                    if (hb0 < 0 || hb0 >= 64 * W) { bad = true; break; }
                    uint64_t pv[PF], ph[PF];
#pragma unroll
                    for (int c = 0; c < PF; ++c) {
                        const int32_t jj = j - c;
                        pv[c] = (jj >= 1 && vb0 >= 0) ? ld(jj, 0, vb0 >> 6) : 0;
                        ph[c] = jj >= 1 ? ld(jj, 1, hb0 >> 6) : 0;
                    }
                    const int32_t j0 = j;
#pragma unroll
                    for (int c = 0; c < PF; ++c) {
                        if (i > 0 && j > 0 && j == j0 - c && !bad) {
                            for (;;) {                                        // query-only steps stay in the column
                                const int32_t hb = i - T.a0 - j, vb = hb - 1;
                                if (hb < 0 || hb >= 64 * W) { bad = true; break; }
                                const uint64_t wv = hb == hb0 ? pv[c] : (vb >= 0 ? ld(j, 0, vb >> 6) : 0);
                                if (vb >= 0 && ((wv >> (vb & 63)) & 1)) {
                                    --i;
                                    if (i == 0) break;
                                    continue;
                                }
                                const uint64_t wh2 = hb == hb0 ? ph[c] : ld(j, 1, hb >> 6);
                                if ((wh2 >> (hb & 63)) & 1) --j;
                                else { --i; --j; }
                                break;
                            }
                        }
                    }
                }
                if (bad) r0 = -6;
                else r_lead = j == 0 ? i : 0;
            }
            }          // (W > 2)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (compiler) the next tile's stores stay behind these loads
        }
        if (has) {
            int32_t *o = out + (size_t)pair * 5;
            const bool fine = r0 >= 0;
            o[0] = r0; o[1] = fine ? r_start : -1; o[2] = fine ? end : -1; o[3] = fine ? r_lead : 0; o[4] = fine ? r_trail : 0;
        }
        if (in.flags && __ballot(has && r0 < -1) != 0 && lane == 0) atomicOr(in.flags, HW_FLAG_STATUS);
    }
}

}  // namespace isocon
