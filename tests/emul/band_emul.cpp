// CPU emulation of one wavefront of the banded edit-distance kernel (isocon_amd/csrc/ed_band.hpp), lane by
// lane, using the SAME lane-level math header (band_core.hpp).  Test infrastructure: lets the not-gpu test
// suite check the band algorithm against the oracle DP in the build container.
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../isocon_amd/csrc/band_core.hpp"

using namespace isocon;

static void pack(const char *s, int len, std::vector<uint64_t> &lo, std::vector<uint64_t> &hi)
{
    int nc = (len + 63) / 64 + 2;
    lo.assign(nc, 0); hi.assign(nc, 0);
    for (int i = 0; i < len; ++i) {
        int code = s[i] == 'A' ? 0 : s[i] == 'C' ? 1 : s[i] == 'G' ? 2 : 3;
        if (code & 1) lo[i >> 6] |= (uint64_t)1 << (i & 63);
        if (code & 2) hi[i >> 6] |= (uint64_t)1 << (i & 63);
    }
}

template <int W>
static int32_t lane_run(const std::vector<uint64_t> &plo, const std::vector<uint64_t> &phi, int m,
                        const std::vector<uint64_t> &tlo, const std::vector<uint64_t> &thi, int n,
                        int k_req, int a0, int n_min, int n_max)
{
    const int d = m - n;
    LaneGeom g = lane_geom<W>(d, k_req, a0);
    if (k_req < 0) return -1;
    if ((d < 0 ? -d : d) > k_req) return -1;
    if (g.k_eff < 0) return -2;
    const int nv = -a0;
    BandLane<W> L;
    band_init<W>(L, nv, g.bstar);
    auto cl = [&](int ci) -> uint64_t { return ci < (int)plo.size() ? plo[ci] : 0; };
    auto ch = [&](int ci) -> uint64_t { return ci < (int)phi.size() ? phi[ci] : 0; };
    uint64_t NL[W], NH[W], VM[W], FL = 0, FH = 0;
    for (int i = 0; i < W; ++i) {
        NL[i] = ~stream64(cl, a0 + 64 * i);
        NH[i] = ~stream64(ch, a0 + 64 * i);
        VM[i] = valid_word(nv, i);
    }
    int32_t res = -1;
    bool captured = false;
    for (int c = 0; 64 * c < n_max; ++c) {
        FL = ~stream64(cl, a0 + 64 * W + 64 * c);
        FH = ~stream64(ch, a0 + 64 * W + 64 * c);
        const uint64_t tl = c < (int)tlo.size() ? tlo[c] : 0, th = c < (int)thi.size() ? thi[c] : 0;
        for (int h = 0; h < 2; ++h) {
            const int jb = 64 * c + 32 * h;
            if (jb >= n_max) break;
            const int cnt = std::min(32, n_max - jb);
            const uint32_t wl = (uint32_t)(tl >> (32 * h)), wh = (uint32_t)(th >> (32 * h));
            const bool fast = cnt == 32 && jb >= nv && jb + 32 <= n_min;
            for (int jj = 0; jj < cnt; ++jj) {
                const uint32_t slo = 0u - ((wl >> jj) & 1u), shi = 0u - ((wh >> jj) & 1u);
                if (fast) band_step<W, false>(L, NL, NH, VM, slo, shi);
                else band_step<W, true>(L, NL, NH, VM, slo, shi);
                window_slide<W>(NL, NH, VM, FL, FH);
                if (!fast && jb + jj + 1 == n && !captured) {
                    captured = true;
                    res = band_diag_value<W>(L, nv, n);
                }
            }
            if (fast && jb + 32 == n && !captured) {  // text ends exactly on a fast half
                captured = true;
                res = band_diag_value<W>(L, nv, n);
            }
        }
    }
    if (!captured) return -3; /* emulator bug guard */
    if (res <= g.k_eff) return res;
    return g.k_eff < k_req ? -2 : -1;
}

template <int W>
static void tile_run(const char *pat, int m, int nl, const char **texts, const int *tlens, const int *ks, int32_t *out)
{
    std::vector<uint64_t> plo, phi;
    pack(pat, m, plo, phi);
    int a0 = 0, n_min = 1 << 30, n_max = 0;
    bool any = false;
    for (int l = 0; l < nl; ++l) {
        const int d = m - tlens[l];
        if (ks[l] < 0 || (d < 0 ? -d : d) > ks[l]) continue;
        any = true;
        a0 = std::min(a0, lane_emin(d, ks[l]));
        n_min = std::min(n_min, tlens[l]);
        n_max = std::max(n_max, tlens[l]);
    }
    if (a0 < -(64 * W - 1)) a0 = -(64 * W - 1);
    for (int l = 0; l < nl; ++l) {
        std::vector<uint64_t> tlo, thi;
        pack(texts[l], tlens[l], tlo, thi);
        if (!any) { out[l] = -1; continue; }
        out[l] = lane_run<W>(plo, phi, m, tlo, thi, tlens[l], ks[l], a0, n_min, n_max);
    }
}

extern "C" void emul_band_tile(int W, const char *pat, int m, int nl, const char **texts, const int *tlens,
                               const int *ks, int32_t *out)
{
    switch (W) {
    case 1: tile_run<1>(pat, m, nl, texts, tlens, ks, out); break;
    case 2: tile_run<2>(pat, m, nl, texts, tlens, ks, out); break;
    case 4: tile_run<4>(pat, m, nl, texts, tlens, ks, out); break;
    case 8: tile_run<8>(pat, m, nl, texts, tlens, ks, out); break;
    default: for (int l = 0; l < nl; ++l) out[l] = -99;
    }
}

// One pair per lane (isocon_amd/csrc/ed_lanes_core.hpp, the routine of k_ed_lanes): pattern x, text y, threshold 0 <= k <= 63.
// Returns what the kernel's lane would: the distance if <= k, else -1 (|m - n| > k and empty sequences are the caller's cases).
#include "../../isocon_amd/csrc/ed_lanes_core.hpp"

extern "C" int32_t emul_ed_lane(const char *x, int m, const char *y, int n, int k)
{
    const int d = m - n, ad = d < 0 ? -d : d;
    if (k < 0 || ad > k) return -1;
    if (m == 0 || n == 0) return ad;
    std::vector<uint64_t> xl, xh, yl, yh;
    pack(x, m, xl, xh);
    pack(y, n, yl, yh);
    auto f = [](const std::vector<uint64_t> &v) { return [&v](int32_t ci) -> uint64_t { return ci >= 0 && ci < (int)v.size() ? v[ci] : 0; }; };
    return lane_pair_distance(f(xl), f(xh), f(yl), f(yh), m, n, k > 63 ? 63 : k, true, [](bool b) { return b; });
}
