// band_core.hpp -- lane-level math of the banded bit-vector edit-distance kernel (gfx950).
//
// Replaces the arithmetic the reference delegates to edlib.align(q, t, mode="NW", task="distance", k=K)
// (/root/reference/modules/nearest_neighbor_graph.py:104-107, modules/edlib_alignment_module.py:111).
//
// Algorithm (not edlib's block scheme): a 64*W-row window of Myers/Hyyro vertical-delta bit-vectors slides
// down one row per text column (diagonal band).  One wavefront handles ONE shared sequence (the "pattern",
// rows) against 64 lane sequences (the "texts", columns); the band origin a0 is wave-uniform so the pattern
// window is a wave-uniform value (SGPRs, slid by the scalar unit) and only the text base differs per lane.
//
//   window at column j (1-based) covers rows a0+j .. a0+j+64W-1; row r <-> pattern base r-1.
//   rows <= 0 are "virtual": D[i][j] = j - i there, which makes row 0 come out as D[0][j] = j without a
//   special case (init: VP = rows >= 1, VN = rows <= 0; Eq masked to 0 on virtual rows).
//   rows > m hold garbage that can never flow upwards.
//   column step (vectors stored after column j are aligned to the window of column j+1):
//       D0 = (((Eq & VP) + VP) ^ VP) | Eq | VN ;  HP = VN | ~(D0 | VP) ;  HN = D0 & VP
//       VP' = HN | ~((D0 >> 1) | HP) ;            VN' = (D0 >> 1) & HP
//   score: D at the window's top row follows a diagonal: top += 1 - D0[bit 0]; the value on the final diagonal
//   (row j + m - n) is top + popcount(VP & low(b*)) - popcount(VN & low(b*)), b* = (m - n) - a0; it is
//   non-decreasing in j and equals D[m][n] at j = n, so "value > k" is a safe early exit.
//
// The same header is compiled by g++ for the wave emulator in tests/emul (CPU unit tests of this math).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ISO_HD __host__ __device__ __forceinline__
#else
#define ISO_HD inline
#endif

namespace isocon {

ISO_HD int popc64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

// a | ~(b | c) on 64 bits.  gfx950 has V_BITOP3_B32 (arbitrary 3-input boolean, truth table = f(0xF0, 0xCC, 0xAA)):
// one 4-cycle-class instruction per 32-bit half instead of v_or + v_bfi.
ISO_HD uint64_t or_nor(uint64_t a, uint64_t b, uint64_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0xF1);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0xF1);
    return ((uint64_t)hi << 32) | lo;
#else
    return a | ~(b | c);
#endif
}

// Geometry of one lane inside a tile whose band origin is a0 (<= 0) and whose window has 64*W rows.
struct LaneGeom {
    int32_t k_eff;   // largest threshold this lane can certify with the tile's window (-1: none)
    int32_t bstar;   // bit index of the final diagonal inside the window
};

// Ukkonen: a path of cost <= k from diagonal 0 to diagonal d = m - n stays on diagonals
// [min(0,d) - x, max(0,d) + x], x = (k - |d|) / 2.
ISO_HD int32_t lane_emin(int32_t d, int32_t k)
{
    const int32_t ad = d < 0 ? -d : d;
    const int32_t x = (k - ad) / 2;
    return (d < 0 ? d : 0) - x;
}

template <int W>
ISO_HD LaneGeom lane_geom(int32_t d, int32_t k, int32_t a0)
{
    LaneGeom g;
    const int32_t ad = d < 0 ? -d : d;
    const int32_t top = a0 + 64 * W - 1;         // highest diagonal covered
    const int32_t dpos = d > 0 ? d : 0, dneg = d < 0 ? d : 0;
    g.bstar = d - a0;
    if (k < ad || top < dpos || a0 > dneg) { g.k_eff = -1; return g; }
    int32_t x = (k - ad) / 2;
    int32_t xmax = top - dpos;
    if (dneg - a0 < xmax) xmax = dneg - a0;
    if (x <= xmax) { g.k_eff = k; return g; }
    g.k_eff = ad + 2 * xmax + 1;                 // largest k with (k-|d|)/2 == xmax
    return g;
}

// Per-lane DP state.
template <int W>
struct BandLane {
    uint64_t VP[W], VN[W];
    uint64_t LM[W];     // low(b*) mask
    uint32_t ztop;      // number of D0 bit-0 hits so far (top = nv + columns - ztop)
};

template <int W>
ISO_HD void band_init(BandLane<W> &L, int32_t nv /* = -a0 */, int32_t bstar)
{
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int32_t lo = nv - 64 * i;          // number of virtual bits in this word
        const uint64_t vp = lo <= 0 ? ~(uint64_t)0 : (lo >= 64 ? 0 : (~(uint64_t)0 << lo));
        L.VP[i] = vp;
        L.VN[i] = ~vp;
        const int32_t b = bstar - 64 * i;
        L.LM[i] = b <= 0 ? 0 : (b >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << b) - 1));
    }
    L.ztop = 0;
}

// 64 W-bit addition x + vp, word by word.  Device, W > 1: the carry travels as a LANE MASK in an SGPR pair through
// v_add_co / v_addc_co (two 4-cycle instructions per word); recomputing it from 64-bit compares (c = s < x, ...) cost two
// 64-bit adds, two 64-bit compares, a mask OR and a select per word -- a third of the column at 6 words.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t band_add_first(uint64_t x, uint64_t vp, uint64_t &cmask)
{
    uint32_t lo, hi;
    asm("v_add_co_u32_e64 %0, %2, %3, %5\n\tv_addc_co_u32_e64 %1, %2, %4, %6, %2"
        : "=&v"(lo), "=&v"(hi), "=&s"(cmask)
        : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "v"((uint32_t)vp), "v"((uint32_t)(vp >> 32)));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t band_add_next(uint64_t x, uint64_t vp, uint64_t &cmask)
{
    uint32_t lo, hi;
    uint64_t cout;
    asm("v_addc_co_u32_e64 %0, %2, %3, %5, %7\n\tv_addc_co_u32_e64 %1, %2, %4, %6, %2"
        : "=&v"(lo), "=&v"(hi), "=&s"(cout)
        : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "v"((uint32_t)vp), "v"((uint32_t)(vp >> 32)), "s"(cmask));
    cmask = cout;
    return ((uint64_t)hi << 32) | lo;
}
#endif

// One text column given the match vectors: EQ[i] bit r = 1 iff the pattern row of window bit r (word i) equals the
// column's text base (0 on virtual rows).
template <int W>
ISO_HD void band_step_eq(BandLane<W> &L, const uint64_t (&EQ)[W])
{
    // Words are processed in order, and word i-1 is finished as soon as word i's D0 is known (its bit 0 is the bit
    // shifted into word i-1): only one word's D0 / HP / HN is live at a time, whatever W is.
    uint64_t carry = 0, d0p = 0, hpp = 0, hnp = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint64_t eq = EQ[i];
        const uint64_t vp = L.VP[i], vn = L.VN[i];
        const uint64_t x = eq & vp;
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t s;
        if (W > 1) s = i == 0 ? band_add_first(x, vp, carry) : band_add_next(x, vp, carry);     // carry: lane mask (SGPR pair)
        else s = x + vp;
#else
        uint64_t s = x + vp;
        if (W > 1) {
            uint64_t c = s < x;
            const uint64_t s2 = s + carry;
            c |= (s2 < s);
            s = s2;
            carry = c;
        }
#endif
        const uint64_t d0 = (s ^ vp) | eq | vn;
        if (i == 0) L.ztop += (uint32_t)d0 & 1u;
        if (i > 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            // two funnel shifts (v_alignbit_b32) instead of a 64-bit shift + v_lshlrev + v_or
            const uint32_t s_lo = __builtin_amdgcn_alignbit((uint32_t)(d0p >> 32), (uint32_t)d0p, 1);
            const uint32_t s_hi = __builtin_amdgcn_alignbit((uint32_t)d0, (uint32_t)(d0p >> 32), 1);
            const uint64_t d0s = ((uint64_t)s_hi << 32) | s_lo;
#else
            const uint64_t d0s = (d0p >> 1) | (d0 << 63);
#endif
            L.VP[i - 1] = or_nor(hnp, d0s, hpp);
            L.VN[i - 1] = d0s & hpp;
        }
        hpp = or_nor(vn, d0, vp);
        hnp = d0 & vp;
        d0p = d0;
    }
    uint64_t d0s = d0p >> 1;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ISOCON_NO_SHIFT_PIN)
    asm("" : "+v"(d0s));
#endif
    L.VP[W - 1] = or_nor(hnp, d0s, hpp);
    L.VN[W - 1] = d0s & hpp;
}

// The 32-row band: thresholds up to 31 need no more (a path of cost <= k stays inside k + 1 diagonals, lane_emin), and on 32-bit
// vectors a column is 10 instructions instead of 19.  zreg collects bit 0 of D0 column by column (shifted in at the top: after c
// columns the c bits sit in the highest c positions of a register that started as 0); the caller adds its popcount to the top-row
// counter once per block of 32 columns instead of an and + add per column.
ISO_HD uint32_t or_nor32(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xF1);
#else
    return a | ~(b | c);
#endif
}

ISO_HD void band_step_eq32(uint32_t &VP, uint32_t &VN, uint32_t &zreg, uint32_t eq)
{
    const uint32_t vp = VP, vn = VN;
    const uint32_t s = (eq & vp) + vp;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t d0 = __builtin_amdgcn_bitop3_b32(s, vp, eq, 0xBE) | vn;          // ((s ^ vp) | eq) | vn
    zreg = __builtin_amdgcn_alignbit(d0, zreg, 1);
#else
    const uint32_t d0 = (s ^ vp) | eq | vn;
    zreg = (zreg >> 1) | (d0 << 31);
#endif
    const uint32_t hp = or_nor32(vn, d0, vp);
    const uint32_t hn = d0 & vp;
    const uint32_t d0s = d0 >> 1;
    VP = or_nor32(hn, d0s, hp);
    VN = d0s & hp;
}

// ... for a block in which a lane's text may end: real = all ones for a text column, 0 behind the end.  A column behind the end
// acts as a "virtual" column (Eq = all ones => D0 = all ones): the band state below the window's last row and the value on the final
// diagonal stay what they were at the text's last column (the top-row counter is bumped through zreg like the column count), so a
// block is always 32 columns for every lane and the 32-column checks read the result afterwards.  One more instruction than
// band_step_eq32 (the caller's bit extract); the two uses of Eq absorb `| ~real` in their 3-input boolean.
ISO_HD void band_step_eq32_tail(uint32_t &VP, uint32_t &VN, uint32_t &zreg, uint32_t eq, uint32_t real)
{
    const uint32_t vp = VP, vn = VN;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t s = __builtin_amdgcn_bitop3_b32(eq, real, vp, 0xA2) + vp;         // ((eq | ~real) & vp) + vp
    const uint32_t t = __builtin_amdgcn_bitop3_b32(s, vp, eq, 0xBE);                 // (s ^ vp) | eq
    const uint32_t d0 = __builtin_amdgcn_bitop3_b32(t, vn, real, 0xFD);              // t | vn | ~real
    zreg = __builtin_amdgcn_alignbit(d0, zreg, 1);
#else
    const uint32_t e2 = eq | ~real;
    const uint32_t s = (e2 & vp) + vp;
    const uint32_t d0 = (s ^ vp) | e2 | vn;
    zreg = (zreg >> 1) | (d0 << 31);
#endif
    const uint32_t hp = or_nor32(vn, d0, vp);
    const uint32_t hn = d0 & vp;
    const uint32_t d0s = d0 >> 1;
    VP = or_nor32(hn, d0s, hp);
    VN = d0s & hp;
}

// The 64-row column step in the same two forms (top-row bits through zreg; `real` = all ones for a text column, 0 behind the end)
ISO_HD void band_step_eq64z(uint64_t &VP, uint64_t &VN, uint32_t &zreg, uint64_t eq)
{
    const uint64_t vp = VP, vn = VN;
    const uint64_t s = (eq & vp) + vp;
    const uint64_t d0 = (s ^ vp) | eq | vn;
#if defined(__HIP_DEVICE_COMPILE__)
    zreg = __builtin_amdgcn_alignbit((uint32_t)d0, zreg, 1);
#else
    zreg = (zreg >> 1) | ((uint32_t)d0 << 31);
#endif
    const uint64_t hp = or_nor(vn, d0, vp);
    const uint64_t hn = d0 & vp;
    uint64_t d0s = d0 >> 1;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ISOCON_NO_SHIFT_PIN)
    asm("" : "+v"(d0s));
#endif
    VP = or_nor(hn, d0s, hp);
    VN = d0s & hp;
}

ISO_HD void band_step_eq64z_tail(uint64_t &VP, uint64_t &VN, uint32_t &zreg, uint64_t eq, uint32_t real)
{
    const uint64_t vp = VP, vn = VN;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t xl = __builtin_amdgcn_bitop3_b32((uint32_t)eq, real, (uint32_t)vp, 0xA2);
    const uint32_t xh = __builtin_amdgcn_bitop3_b32((uint32_t)(eq >> 32), real, (uint32_t)(vp >> 32), 0xA2);
    const uint64_t s = (((uint64_t)xh << 32) | xl) + vp;
    const uint32_t tl = __builtin_amdgcn_bitop3_b32((uint32_t)s, (uint32_t)vp, (uint32_t)eq, 0xBE);
    const uint32_t th = __builtin_amdgcn_bitop3_b32((uint32_t)(s >> 32), (uint32_t)(vp >> 32), (uint32_t)(eq >> 32), 0xBE);
    const uint32_t dl = __builtin_amdgcn_bitop3_b32(tl, (uint32_t)vn, real, 0xFD);
    const uint32_t dh = __builtin_amdgcn_bitop3_b32(th, (uint32_t)(vn >> 32), real, 0xFD);
    const uint64_t d0 = ((uint64_t)dh << 32) | dl;
    zreg = __builtin_amdgcn_alignbit(dl, zreg, 1);
#else
    const uint64_t r64 = ((uint64_t)real << 32) | real;
    const uint64_t e2 = eq | ~r64;
    const uint64_t s = (e2 & vp) + vp;
    const uint64_t d0 = (s ^ vp) | e2 | vn;
    zreg = (zreg >> 1) | ((uint32_t)d0 << 31);
#endif
    const uint64_t hp = or_nor(vn, d0, vp);
    const uint64_t hn = d0 & vp;
    uint64_t d0s = d0 >> 1;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ISOCON_NO_SHIFT_PIN)
    asm("" : "+v"(d0s));
#endif
    VP = or_nor(hn, d0s, hp);
    VN = d0s & hp;
}

ISO_HD int popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

// The same column step that also hands out the horizontal "+1" vector HP of the column (bit r <-> window row r of THIS
// column, i.e. before the slide) -- what a traceback needs next to the new VP (hw_core.hpp).
template <int W>
ISO_HD void band_step_eq_hp(BandLane<W> &L, const uint64_t (&EQ)[W], uint64_t (&HP)[W])
{
    uint64_t carry = 0, d0p = 0, hpp = 0, hnp = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint64_t eq = EQ[i];
        const uint64_t vp = L.VP[i], vn = L.VN[i];
        const uint64_t x = eq & vp;
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t s;
        if (W > 1) s = i == 0 ? band_add_first(x, vp, carry) : band_add_next(x, vp, carry);     // carry: lane mask (SGPR pair)
        else s = x + vp;
#else
        uint64_t s = x + vp;
        if (W > 1) {
            uint64_t c = s < x;
            const uint64_t s2 = s + carry;
            c |= (s2 < s);
            s = s2;
            carry = c;
        }
#endif
        const uint64_t d0 = (s ^ vp) | eq | vn;
        if (i == 0) L.ztop += (uint32_t)d0 & 1u;
        if (i > 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            // two funnel shifts (v_alignbit_b32) instead of a 64-bit shift + v_lshlrev + v_or
            const uint32_t s_lo = __builtin_amdgcn_alignbit((uint32_t)(d0p >> 32), (uint32_t)d0p, 1);
            const uint32_t s_hi = __builtin_amdgcn_alignbit((uint32_t)d0, (uint32_t)(d0p >> 32), 1);
            const uint64_t d0s = ((uint64_t)s_hi << 32) | s_lo;
#else
            const uint64_t d0s = (d0p >> 1) | (d0 << 63);
#endif
            L.VP[i - 1] = or_nor(hnp, d0s, hpp);
            L.VN[i - 1] = d0s & hpp;
        }
        hpp = or_nor(vn, d0, vp);
        HP[i] = hpp;
        hnp = d0 & vp;
        d0p = d0;
    }
    const uint64_t d0s = d0p >> 1;
    L.VP[W - 1] = or_nor(hnp, d0s, hpp);
    L.VN[W - 1] = d0s & hpp;
}

// One text column.  NL/NH = ~pattern bit-planes of the current window (wave-uniform), VM = valid-row mask
// (only read when MASKED), slo/shi = text base bit-planes splat to 32 bits (0 or 0xffffffff).
template <int W, bool MASKED>
ISO_HD void band_step(BandLane<W> &L, const uint64_t (&NL)[W], const uint64_t (&NH)[W], const uint64_t (&VM)[W],
                      uint32_t slo, uint32_t shi)
{
    const uint64_t sl = ((uint64_t)slo << 32) | slo;
    const uint64_t sh = ((uint64_t)shi << 32) | shi;
    uint64_t EQ[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
        // (NL ^ sl) & (NH ^ sh) per 32-bit half as xor + one 3-input boolean a & (b ^ c) (table 0x60): 2 instead of 3 instructions
        const uint32_t tl = (uint32_t)NL[i] ^ slo, th = (uint32_t)(NL[i] >> 32) ^ slo;
        const uint32_t el = __builtin_amdgcn_bitop3_b32(tl, (uint32_t)NH[i], shi, 0x60);
        const uint32_t eh = __builtin_amdgcn_bitop3_b32(th, (uint32_t)(NH[i] >> 32), shi, 0x60);
        EQ[i] = ((uint64_t)eh << 32) | el;
        (void)sl; (void)sh;
#else
        EQ[i] = (NL[i] ^ sl) & (NH[i] ^ sh);
#endif
        if (MASKED) EQ[i] &= VM[i];
    }
    band_step_eq<W>(L, EQ);
}

// D on the final diagonal after `cols` columns.
template <int W>
ISO_HD int32_t band_diag_value(const BandLane<W> &L, int32_t nv, int32_t cols)
{
    int32_t v = nv + cols - (int32_t)L.ztop;
#pragma unroll
    for (int i = 0; i < W; ++i) v += popc64(L.VP[i] & L.LM[i]) - popc64(L.VN[i] & L.LM[i]);
    return v;
}

// Slide the wave-uniform window one row down.  F* hold the next 64 stream bits (already complemented for NL/NH).
template <int W>
ISO_HD void window_slide(uint64_t (&NL)[W], uint64_t (&NH)[W], uint64_t (&VM)[W], uint64_t &FL, uint64_t &FH)
{
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint64_t inl = (i + 1 < W) ? NL[(i + 1 < W) ? i + 1 : i] : FL;
        const uint64_t inh = (i + 1 < W) ? NH[(i + 1 < W) ? i + 1 : i] : FH;
        const uint64_t inv = (i + 1 < W) ? VM[(i + 1 < W) ? i + 1 : i] : ~(uint64_t)0;
        NL[i] = (NL[i] >> 1) | (inl << 63);
        NH[i] = (NH[i] >> 1) | (inh << 63);
        VM[i] = (VM[i] >> 1) | (inv << 63);
    }
    FL >>= 1;
    FH >>= 1;
}

// 64 bits of a sequence's bit-plane stream starting at bit `off` (may be negative: virtual rows read as 0).
// chunk(ci) must return plane word ci of the shared sequence, 0 beyond the end.
template <class ChunkFn>
ISO_HD uint64_t stream64(ChunkFn chunk, int32_t off)
{
    if (off <= -64) return 0;
    if (off < 0) return chunk(0) << (-off);
    const int32_t ci = off >> 6, s = off & 63;
    const uint64_t a = chunk(ci);
    if (s == 0) return a;
    return (a >> s) | (chunk(ci + 1) << (64 - s));
}

ISO_HD uint64_t valid_word(int32_t nv, int i)
{
    const int32_t lo = nv - 64 * i;
    return lo <= 0 ? ~(uint64_t)0 : (lo >= 64 ? 0 : (~(uint64_t)0 << lo));
}

}  // namespace isocon
