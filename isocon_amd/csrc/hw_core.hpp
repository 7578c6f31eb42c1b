// hw_core.hpp -- lane-level math of the BIT-PARALLEL infix ("HW") alignment kernels (gfx950).
//
// What edlib.align(q, t, mode="HW", task="path", k) is asked for by the candidate-vs-candidate graph of the statistical
// test (/root/reference/modules/end_invariant_functions.py:593-620 edlib_traceback, :622-681 get_all_NN): the distance h
// of q inside t (free target prefix / suffix), locations[0] = (start, end), and the lengths of the insertion runs the path
// starts / ends with.  Three passes of the diagonal-band bit-vector machinery of band_core.hpp, one wavefront = one shared
// QUERY (rows, wave-uniform window) x 64 lane TARGETS (columns):
//   LOCATE  rows = query, columns = target, row 0 all zeros ("virtual" rows <= 0 hold D = 0: VP = VN = 0 there and Eq = 1,
//           which the column step preserves), diagonals j - i in [-k, max(delta) + k].  Every column j >= P - k the value of
//           the LAST row is read off the vertical deltas: h = its minimum, end = the first column attaining it.  A lane is
//           abandoned once top - popcount(VN) > k: that is a lower bound of every cell of the column inside the band, and a
//           path to any later end cell crosses this column inside the band or costs more than k anyway.
//   START   reversed query against the reversed target prefix t[0..end], row 0 = j (ordinary global top row), diagonals
//           [-k, k]: the LAST column of the final row that equals h gives the smallest start (edlib's rule, oracle section 5).
//   TRACE   query against t[start..end], global, diagonals [-k, k]; every column stores its new vertical "+1" vector VP and
//           its horizontal "+1" vector HP; the walk from the end cell follows edlib's order: query-only step ('I') if the
//           vertical step is optimal, else target-only ('D') if the horizontal one is, else the diagonal.
// The same header is compiled by g++ for tests/emul/hw_emul.cpp (CPU unit tests of this math against the oracle).
#pragma once
#include "band_core.hpp"

namespace isocon {

enum { HW_LOCATE = 0, HW_START = 1, HW_TRACE = 2, HW_TRACE_CK = 3 };
// HW_TRACE_CK: the TRACE pass that keeps only CHECKPOINTS -- the band state (VP, VN) after every HW_SEG-th column -- instead of every
// column's VP / HP: the walk then goes segment by segment from the end, each segment's columns recomputed from its checkpoint
// (hw_trace_segment) into a small buffer (LDS on the device) and walked there (hw_walk_segment).  1 byte per column and word instead of 16.
static constexpr int HW_SEG = 8;
static constexpr int32_t HWB_INF = 1 << 28;

struct HwTile {          // wave-uniform
    int32_t P;           // query length (rows)
    int32_t a0;          // band origin (<= 0): the window of column j covers rows a0 + j .. a0 + j + 64 W - 1
    int32_t ncols_max;   // columns the tile runs
    int32_t jx;          // first column (1-based) at which some lane may need a value of the last row
};

struct HwLane {
    int32_t ncols;       // columns of this lane (LOCATE: len(target); START: min(end + 1, P + kmax); TRACE: end - start + 1); 0 = empty lane
    int32_t k;           // LOCATE: threshold of the pair
    int32_t h;           // START / TRACE: the pair's distance
    int32_t r_h, r_end;  // LOCATE: min over columns of D[P][j] (HWB_INF if the lane was abandoned first), 0-based first column attaining it
    int32_t r_pl;        // START: last column j (1-based) with D[P][j] == h, -1 if none
    int32_t r_final, r_trail;   // TRACE: D[P][ncols] and the number of consecutive optimal vertical steps at the end cell
};

// tile geometry from the lanes' extremes (device: wave reductions; emulator: loops)
ISO_HD int32_t hw_locate_a0(int32_t delta_max, int32_t k_max) { return -((delta_max > 0 ? delta_max : 0) + k_max); }
ISO_HD int32_t hw_locate_rows(int32_t delta_max, int32_t k_max) { return (delta_max > 0 ? delta_max : 0) + 2 * k_max + 1; }

ISO_HD uint64_t hw_brev64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
    x = ((x >> 8) & 0x00ff00ff00ff00ffull) | ((x & 0x00ff00ff00ff00ffull) << 8);
    x = ((x >> 16) & 0x0000ffff0000ffffull) | ((x & 0x0000ffff0000ffffull) << 16);
    return (x >> 32) | (x << 32);
#endif
}

ISO_HD uint32_t hw_brev32(uint32_t x) { return (uint32_t)(hw_brev64(x) >> 32); }

// 64 bits of the REVERSED sequence (length P) starting at reversed offset `off`: reversed base i = base P - 1 - i
template <class ChunkFn>
ISO_HD uint64_t stream64_rev(ChunkFn chunk, int32_t P, int32_t off)
{
    return hw_brev64(stream64(chunk, P - 64 - off));
}

// D at window bit b of the column just processed (vectors are stored aligned to the NEXT column's window: VP[r] is the
// vertical delta between window rows r and r + 1 of this column)
template <int W>
ISO_HD int32_t hw_row_value(const BandLane<W> &L, int32_t top, int32_t b)
{
    int32_t v = top;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int32_t bb = b - 64 * i;
        const uint64_t lm = bb <= 0 ? 0 : (bb >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << bb) - 1));
        v += popc64(L.VP[i] & lm) - popc64(L.VN[i] & lm);
    }
    return v;
}

template <int W>
ISO_HD int32_t hw_lower_bound(const BandLane<W> &L, int32_t top)
{
    int32_t v = top;
#pragma unroll
    for (int i = 0; i < W; ++i) v -= popc64(L.VN[i]);
    return v;
}

template <int W>
ISO_HD uint32_t hw_bit(const uint64_t (&V)[W], int32_t r)
{
    uint64_t w = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) w = (r >> 6) == i ? V[i] : w;
    return (uint32_t)(w >> (r & 63)) & 1u;
}

// One pass over one lane of one tile.  plo / phi(off): 64 bits of the query's bit-planes starting at row offset off (the
// caller hands a forward or a reversed stream); text(jb, wl, wh): the lane's target bits of columns jb + 1 .. jb + 32;
// any_live(live): does any lane of the wave still need columns (device: a ballot; emulator: the lane itself);
// sink(j, word, vp_new, hp): TRACE; sink(j / HW_SEG, word, vp_new, vn_new) after every column j that is a multiple of HW_SEG: TRACE_CK
// (there T.jx = the smallest ncols of the tile's lanes: the blocks before it take the unrolled path).
template <int W, int MODE, class PLo, class PHi, class Text, class AnyLive, class Sink>
ISO_HD void hw_run(const HwTile &T, HwLane &ln, PLo plo, PHi phi, Text text, AnyLive any_live, Sink sink)
{
    constexpr bool ZT = MODE == HW_LOCATE;
    const int32_t nv = -T.a0;
    BandLane<W> L;
    uint64_t NL[W], NH[W], VM[W], FL = 0, FH = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint64_t vp = valid_word(nv, i);
        L.VP[i] = vp;
        L.VN[i] = ZT ? 0 : ~vp;
        NL[i] = ~plo(T.a0 + 64 * i);
        NH[i] = ~phi(T.a0 + 64 * i);
        VM[i] = vp;
    }
    L.ztop = 0;
    const int32_t top0 = ZT ? 0 : nv;
    bool live = ln.ncols > 0;
    int32_t h = HWB_INF, end = -1, pl = -1;
    ln.r_final = HWB_INF;
    ln.r_trail = 0;
    for (int32_t jb = 0; jb < T.ncols_max; jb += 32) {
        if ((jb & 63) == 0) {
            FL = ~plo(T.a0 + 64 * W + jb);
            FH = ~phi(T.a0 + 64 * W + jb);
        }
        uint32_t wl = 0, wh = 0;
        text(jb, wl, wh);
        const int32_t cnt = (T.ncols_max - jb) < 32 ? (T.ncols_max - jb) : 32;
        const bool plain = MODE != HW_TRACE && cnt == 32 && jb + 32 < T.jx;      // wave-uniform: no last-row values in this block
        if (plain) {
#pragma unroll
            for (int jj = 0; jj < 32; ++jj) {
                const uint32_t sl = 0u - ((wl >> jj) & 1u), sh = 0u - ((wh >> jj) & 1u);
                const uint64_t sl64 = ((uint64_t)sl << 32) | sl, sh64 = ((uint64_t)sh << 32) | sh;
                uint64_t EQ[W];
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const uint64_t e = (NL[i] ^ sl64) & (NH[i] ^ sh64);
                    EQ[i] = ZT ? (e | ~VM[i]) : (e & VM[i]);
                }
                band_step_eq<W>(L, EQ);
                window_slide<W>(NL, NH, VM, FL, FH);
                if (MODE == HW_TRACE_CK && (jj & (HW_SEG - 1)) == HW_SEG - 1) {
#pragma unroll
                    for (int i = 0; i < W; ++i) sink((jb + jj + 1) / HW_SEG, i, L.VP[i], L.VN[i]);
                }
            }
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int jj = 0; jj < cnt; ++jj) {
                const uint32_t sl = 0u - ((wl >> jj) & 1u), sh = 0u - ((wh >> jj) & 1u);
                const uint64_t sl64 = ((uint64_t)sl << 32) | sl, sh64 = ((uint64_t)sh << 32) | sh;
                uint64_t EQ[W], HP[W];
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const uint64_t e = (NL[i] ^ sl64) & (NH[i] ^ sh64);
                    EQ[i] = ZT ? (e | ~VM[i]) : (e & VM[i]);
                }
                if (MODE == HW_TRACE) band_step_eq_hp<W>(L, EQ, HP);
                else band_step_eq<W>(L, EQ);
                window_slide<W>(NL, NH, VM, FL, FH);
                const int32_t j = jb + jj + 1;
                const int32_t top = top0 + j - (int32_t)L.ztop;
                const int32_t b = T.P - T.a0 - j;              // window bit of the last row in this column
                if (MODE == HW_LOCATE) {
                    if (live && j >= T.P - ln.k && j <= ln.ncols && b >= 0 && b < 64 * W) {
                        const int32_t v = hw_row_value<W>(L, top, b);
                        if (v < h) { h = v; end = j - 1; }
                    }
                } else if (MODE == HW_START) {
                    if (j >= T.jx && j <= ln.ncols && b >= 0 && b < 64 * W) {
                        if (hw_row_value<W>(L, top, b) == ln.h) pl = j;
                    }
                } else {
                    if (MODE == HW_TRACE && j <= ln.ncols) {
#pragma unroll
                        for (int i = 0; i < W; ++i) sink(j, i, L.VP[i], HP[i]);
                    }
                    if (MODE == HW_TRACE_CK && j <= ln.ncols && (j & (HW_SEG - 1)) == 0) {
#pragma unroll
                        for (int i = 0; i < W; ++i) sink(j / HW_SEG, i, L.VP[i], L.VN[i]);
                    }
                    if (j == ln.ncols && b >= 0 && b < 64 * W) {
                        ln.r_final = hw_row_value<W>(L, top, b);
                        int32_t t = 0;
                        for (int32_t r = b - 1; r >= 0; --r) {
                            if (!hw_bit<W>(L.VP, r)) break;
                            ++t;
                        }
                        ln.r_trail = t;
                    }
                }
            }
        }
        if (MODE == HW_LOCATE) {
            if (live) {
                const int32_t cols = jb + cnt;
                if (cols >= ln.ncols) live = false;
                else if (hw_lower_bound<W>(L, top0 + cols - (int32_t)L.ztop) > ln.k) live = false;
            }
            if (!any_live(live)) break;
        }
    }
    ln.r_h = h;
    ln.r_end = end;
    ln.r_pl = pl;
}

// Columns c0 + 1 .. c0 + HW_SEG of the TRACE pass again, from the band state after column c0 (c0 a multiple of HW_SEG; L = the
// checkpoint, or the initial state for c0 = 0: hw_trace_init).  wl / wh: the lane's target bits of these columns (bit jj <-> column
// c0 + 1 + jj).  sink(jj, word, vp_new, hp) like the TRACE pass.
template <int W>
ISO_HD void hw_trace_init(const HwTile &T, BandLane<W> &L)
{
#pragma unroll
    for (int i = 0; i < W; ++i) {
        L.VP[i] = valid_word(-T.a0, i);
        L.VN[i] = ~L.VP[i];
    }
    L.ztop = 0;
}

template <int W, class PLo, class PHi, class Sink>
ISO_HD void hw_trace_segment(const HwTile &T, int32_t c0, BandLane<W> &L, PLo plo, PHi phi, uint32_t wl, uint32_t wh, Sink sink)
{
    uint64_t NL[W], NH[W], VM[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
        NL[i] = ~plo(T.a0 + c0 + 64 * i);
        NH[i] = ~phi(T.a0 + c0 + 64 * i);
        VM[i] = valid_word(-T.a0 - c0, i);
    }
    uint64_t FL = ~plo(T.a0 + c0 + 64 * W), FH = ~phi(T.a0 + c0 + 64 * W);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jj = 0; jj < HW_SEG; ++jj) {
        const uint32_t sl = 0u - ((wl >> jj) & 1u), sh = 0u - ((wh >> jj) & 1u);
        const uint64_t sl64 = ((uint64_t)sl << 32) | sl, sh64 = ((uint64_t)sh << 32) | sh;
        uint64_t EQ[W], HP[W];
#pragma unroll
        for (int i = 0; i < W; ++i) EQ[i] = (NL[i] ^ sl64) & (NH[i] ^ sh64) & VM[i];
        band_step_eq_hp<W>(L, EQ, HP);
        window_slide<W>(NL, NH, VM, FL, FH);
#pragma unroll
        for (int i = 0; i < W; ++i) sink(jj, i, L.VP[i], HP[i]);
    }
}

// The walk inside the segment of columns c0 + 1 .. c0 + HW_SEG: load(jj, which, word) = what hw_trace_segment's sink got for column
// c0 + 1 + jj.  Stops when the path leaves the segment (j == c0) or ends (i == 0); false: off the band (cannot happen on an optimal path).
template <int W, class Load>
ISO_HD bool hw_walk_segment(int32_t a0, int32_t c0, int32_t &i, int32_t &j, Load load)
{
    while (i > 0 && j > c0) {
        const int32_t hb = i - a0 - j, vb = hb - 1;
        if (hb < 0 || hb >= 64 * W) return false;
        if (vb >= 0 && ((load(j - c0 - 1, 0, vb >> 6) >> (vb & 63)) & 1)) { --i; continue; }
        if ((load(j - c0 - 1, 1, hb >> 6) >> (hb & 63)) & 1) --j;
        else { --i; --j; }
    }
    return true;
}

// The same walk with the segment's vectors in REGISTERS (the device's one-word form): the path visits the columns of a segment in
// descending order and never returns to one, so the columns can be an unrolled loop -- a lane takes part in column c0 + 1 + jj while its
// j is that column (query-only steps stay in the column), and VPs / HPs are only ever indexed by compile-time constants.
template <int W>
ISO_HD bool hw_walk_segment_regs(int32_t a0, int32_t c0, int32_t &i, int32_t &j, const uint64_t (&VPs)[HW_SEG][W], const uint64_t (&HPs)[HW_SEG][W])
{
    bool ok = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jj = HW_SEG - 1; jj >= 0; --jj) {
        if (ok && i > 0 && j == c0 + jj + 1) {
            for (;;) {
                const int32_t hb = i - a0 - j, vb = hb - 1;
                if (hb < 0 || hb >= 64 * W) { ok = false; break; }
                if (vb >= 0 && hw_bit<W>(VPs[jj], vb)) {
                    --i;
                    if (i == 0) break;
                    continue;
                }
                if (hw_bit<W>(HPs[jj], hb)) --j;
                else { --i; --j; }
                break;
            }
        }
    }
    return ok;
}

// Packed column store (one word of band, pairs at distance h <= 14): the walk only visits cells whose diagonal offset j - i lies
// in [-h, h], i.e. window bits kmax - h - 1 .. kmax + h of VP and HP -- at most 32 of each, so both fit ONE 64-bit word
// (VP part low, HP part high) instead of two: half the bytes of the column store, which is what the finish kernel is bound by.
ISO_HD bool hw_packable(int W, int32_t h, int32_t kmax) { return W == 1 && h <= 14 && kmax - h - 1 >= 0; }
ISO_HD uint64_t hw_pack(uint64_t vp, uint64_t hp, int32_t shift)
{
    return (uint64_t)(uint32_t)(vp >> shift) | ((uint64_t)(uint32_t)(hp >> shift) << 32);
}
ISO_HD uint64_t hw_unpack(uint64_t word, int which, int32_t shift)
{
    return (uint64_t)(which == 0 ? (uint32_t)word : (uint32_t)(word >> 32)) << shift;
}

// The walk of the TRACE pass.  load(j, which, word): which = 0 the column's new VP (bit r: the vertical step INTO window row
// r + 1 of column j is optimal), 1 its HP (bit r: the horizontal step into window row r is optimal).  Returns the leading
// insertion run (rows left when column 0 is reached).
template <int W, class Load>
ISO_HD int32_t hw_walk(int32_t P, int32_t a0, int32_t ms, Load load)
{
    int32_t i = P, j = ms;
    while (i > 0 && j > 0) {
        const int32_t hb = i - a0 - j, vb = hb - 1;
        if (hb < 0 || hb >= 64 * W) return -1;                 // off the band: cannot happen on an optimal path
        if (vb >= 0 && ((load(j, 0, vb >> 6) >> (vb & 63)) & 1)) { --i; continue; }
        if ((load(j, 1, hb >> 6) >> (hb & 63)) & 1) --j;
        else { --i; --j; }
    }
    return j == 0 ? i : 0;
}

}  // namespace isocon
