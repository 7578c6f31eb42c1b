// ed_lanes_core.hpp -- one pair per lane through the 64-row band (see ed_lanes.hpp): the lane-level routine, shared by the kernel
// and by the CPU emulator of tests/emul (g++, also under UBSan).
#pragma once
#include "band_core.hpp"

namespace isocon {

// (hi:lo) >> s for 0 <= s < 32
ISO_HD uint32_t funnel32(uint32_t hi, uint32_t lo, int s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s);
#else
    return s ? (lo >> s) | (hi << (32 - s)) : lo;
#endif
}

// bit j of w replicated into all 32 bits
ISO_HD uint32_t splat_bit(uint32_t w, int j)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_sbfe((int)w, (uint32_t)j, 1u);
#else
    return 0u - ((w >> j) & 1u);
#endif
}

// bits i of a 32-bit word at stream offset o that lie inside [0, m)
ISO_HD uint32_t lane_valid32(int32_t o, int32_t m)
{
    int32_t lo = -o, hi = m - o;
    if (lo < 0) lo = 0;
    if (hi > 32) hi = 32;
    if (hi <= lo) return 0u;
    const uint32_t upto_hi = hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u);
    return upto_hi & ~((1u << lo) - 1u);        // lo < hi <= 32, so lo < 32
}

// Pattern x (rows, length m) against text y (columns, length n), threshold k <= 63: the distance if it is <= k, else -1.
// run: this lane has a pair with 0 <= k, |m - n| <= k, m > 0, n > 0 (lanes without one go through the motions: `any` is a vote of
// the whole wave on the device -- __ballot(b) != 0 -- and the identity on the host).
// x_lo .. y_hi: chunk index -> 64 bits of the sequence's low / high bit-plane, 0 beyond its end.
//
// Window at 0-based column c covers the pattern positions c - nv .. c - nv + 63 (band_core.hpp); the complemented planes of the
// positions [o, o + 96), o = c0 - nv, sit in three dwords per plane for the 32 columns of a block: window bit b of column c0 + jj is
// register bit jj + b.  Positions outside [0, m) are masked while o < 0 or the text ends inside the block; past the end of the
// pattern the planes read 0 and those rows cannot reach the final diagonal.
template <class XL, class XH, class YL, class YH, class Any>
ISO_HD int32_t lane_pair_distance(XL x_lo, XH x_hi, YL y_lo, YH y_hi, int32_t m, int32_t n, int32_t k, bool run, Any any)
{
    const int32_t d = m - n;
    int32_t a0 = lane_emin(d, k < 0 ? 0 : k);
    if (a0 < -63) a0 = -63;
    const int32_t nv = -a0;
    int32_t bstar = d - a0;
    if (bstar < 0) bstar = 0;
    if (bstar > 63) bstar = 63;
    BandLane<1> L;
    band_init<1>(L, nv, bstar);
    int32_t r = -1;
    int32_t o = -nv;
    uint32_t L0, L1, L2, H0, H1, H2;
    {
        const uint64_t tl = ~stream64(x_lo, o), th = ~stream64(x_hi, o);
        L0 = (uint32_t)tl; L1 = (uint32_t)(tl >> 32); H0 = (uint32_t)th; H1 = (uint32_t)(th >> 32);
        L2 = (uint32_t)~stream64(x_lo, o + 64);
        H2 = (uint32_t)~stream64(x_hi, o + 64);
    }
    for (int32_t c0 = 0;; c0 += 32) {
        if (!any(run)) break;
        const uint32_t wl = (uint32_t)stream64(y_lo, c0), wh = (uint32_t)stream64(y_hi, c0);       // 32 text bases
        const bool full = c0 + 32 <= n;
        if (o >= 0 && !any(run && !full)) {
            // no virtual rows in any window of the block, 32 columns for every running lane
            // (the top-row bit of every column is shifted into zreg and counted once per block, as in the table kernel: band_step_eq64z)
            uint32_t zreg = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int jj = 0; jj < 32; ++jj) {
                const uint32_t sl = splat_bit(wl, jj), sh = splat_bit(wh, jj);
                const uint32_t nl0 = funnel32(L1, L0, jj), nl1 = funnel32(L2, L1, jj), nh0 = funnel32(H1, H0, jj), nh1 = funnel32(H2, H1, jj);
#if defined(__HIP_DEVICE_COMPILE__)
                const uint32_t e0 = __builtin_amdgcn_bitop3_b32(nl0 ^ sl, nh0, sh, 0x60);          // (nl ^ sl) & (nh ^ sh)
                const uint32_t e1 = __builtin_amdgcn_bitop3_b32(nl1 ^ sl, nh1, sh, 0x60);
#else
                const uint32_t e0 = (nl0 ^ sl) & (nh0 ^ sh), e1 = (nl1 ^ sl) & (nh1 ^ sh);
#endif
                band_step_eq64z(L.VP[0], L.VN[0], zreg, ((uint64_t)e1 << 32) | e0);
            }
#if defined(__HIP_DEVICE_COMPILE__)
            L.ztop += (uint32_t)__popc(zreg);
#else
            L.ztop += (uint32_t)__builtin_popcount(zreg);
#endif
        } else {
            const uint32_t V0 = lane_valid32(o, m), V1 = lane_valid32(o + 32, m), V2 = lane_valid32(o + 64, m);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int jj = 0; jj < 32; ++jj) {
                if (run && c0 + jj < n) {
                    uint64_t NL[1], NH[1], VM[1];
                    NL[0] = ((uint64_t)funnel32(L2, L1, jj) << 32) | funnel32(L1, L0, jj);
                    NH[0] = ((uint64_t)funnel32(H2, H1, jj) << 32) | funnel32(H1, H0, jj);
                    VM[0] = ((uint64_t)funnel32(V2, V1, jj) << 32) | funnel32(V1, V0, jj);
                    band_step<1, true>(L, NL, NH, VM, splat_bit(wl, jj), splat_bit(wh, jj));
                }
            }
        }
        if (run) {
            const bool fin = c0 + 32 >= n;
            const int32_t dv = band_diag_value<1>(L, nv, fin ? n : c0 + 32);
            if (fin) { r = dv <= k ? dv : -1; run = false; }
            else if (dv > k) run = false;                 // the value on the final diagonal never decreases
        }
        o += 32;
        L0 = L1; L1 = L2; H0 = H1; H1 = H2;
        L2 = (uint32_t)~stream64(x_lo, o + 64);
        H2 = (uint32_t)~stream64(x_hi, o + 64);
    }
    return r;
}

}  // namespace isocon
