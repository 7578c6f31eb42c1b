// CPU emulation of one tile (one query x up to 64 targets) of the bit-parallel infix kernels (isocon_amd/csrc/hw.hpp),
// lane by lane, on the SAME lane-level math header (hw_core.hpp).  Test infrastructure: lets the not-gpu suite check the
// three passes (LOCATE, START, TRACE + walk) against the oracle's full matrices in the build container.
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../isocon_amd/csrc/hw_core.hpp"

using namespace isocon;

static inline int code_of(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }

struct Packed {
    std::vector<uint64_t> lo, hi;
    int len = 0;
    void set(const char *s, int n)
    {
        len = n;
        const int nc = (n + 63) / 64 + 2;
        lo.assign(nc, 0); hi.assign(nc, 0);
        for (int i = 0; i < n; ++i) {
            const int c = code_of(s[i]);
            if (c & 1) lo[i >> 6] |= (uint64_t)1 << (i & 63);
            if (c & 2) hi[i >> 6] |= (uint64_t)1 << (i & 63);
        }
    }
    uint32_t bit(const std::vector<uint64_t> &v, long p) const { return (p < 0 || p >= (long)v.size() * 64) ? 0u : (uint32_t)(v[p >> 6] >> (p & 63)) & 1u; }
};

template <int W>
static void tile(const char *q, int P, int nl, const char **ts, const int *tl, const int *ks, int32_t *out)
{
    Packed Q; Q.set(q, P);
    std::vector<Packed> T(nl);
    for (int l = 0; l < nl; ++l) T[l].set(ts[l], tl[l]);
    auto qlo = [&](int ci) -> uint64_t { return ci >= 0 && ci < (int)Q.lo.size() ? Q.lo[ci] : 0; };
    auto qhi = [&](int ci) -> uint64_t { return ci >= 0 && ci < (int)Q.hi.size() ? Q.hi[ci] : 0; };
    auto f_lo = [&](int32_t off) { return stream64(qlo, off); };
    auto f_hi = [&](int32_t off) { return stream64(qhi, off); };
    auto r_lo = [&](int32_t off) { return stream64_rev(qlo, P, off); };
    auto r_hi = [&](int32_t off) { return stream64_rev(qhi, P, off); };
    auto alone = [](bool live) { return live; };
    auto nosink = [](int32_t, int, uint64_t, uint64_t) {};

    for (int l = 0; l < nl; ++l) { out[5 * l] = -1; out[5 * l + 1] = -1; out[5 * l + 2] = -1; out[5 * l + 3] = 0; out[5 * l + 4] = 0; }
    // ---- LOCATE ----
    int dmax = -(1 << 30), kmax = 0, mmax = 0;
    std::vector<char> valid(nl, 0);
    for (int l = 0; l < nl; ++l) {
        valid[l] = P > 0 && tl[l] > 0 && ks[l] >= 0 && tl[l] - P >= -ks[l];
        if (!valid[l]) continue;
        dmax = std::max(dmax, tl[l] - P); kmax = std::max(kmax, ks[l]); mmax = std::max(mmax, tl[l]);
    }
    if (dmax == -(1 << 30)) return;
    if (hw_locate_rows(dmax, kmax) > 64 * W) { for (int l = 0; l < nl; ++l) out[5 * l] = -3; return; }
    HwTile A; A.P = P; A.a0 = hw_locate_a0(dmax, kmax); A.ncols_max = mmax; A.jx = std::max(1, P - kmax);
    std::vector<int> h(nl, -1), end(nl, -1);
    for (int l = 0; l < nl; ++l) {
        if (!valid[l]) continue;
        HwLane ln; memset(&ln, 0, sizeof ln);
        ln.ncols = tl[l]; ln.k = ks[l];
        auto text = [&](int32_t jb, uint32_t &wl, uint32_t &wh) {
            wl = wh = 0;
            for (int x = 0; x < 32; ++x) { wl |= T[l].bit(T[l].lo, jb + x) << x; wh |= T[l].bit(T[l].hi, jb + x) << x; }
        };
        hw_run<W, HW_LOCATE>(A, ln, f_lo, f_hi, text, alone, nosink);
        if (ln.r_h <= ks[l]) { h[l] = ln.r_h; end[l] = ln.r_end; }
    }
    // ---- START ---- (reversed query against the reversed prefix t[0..end])
    int hk = 0, cmax = 0;
    for (int l = 0; l < nl; ++l) if (h[l] >= 0) hk = std::max(hk, ks[l]);
    if (2 * hk + 1 > 64 * W) { for (int l = 0; l < nl; ++l) out[5 * l] = -3; return; }
    for (int l = 0; l < nl; ++l) if (h[l] >= 0) cmax = std::max(cmax, std::min(end[l] + 1, P + hk));
    HwTile B; B.P = P; B.a0 = -hk; B.ncols_max = cmax; B.jx = std::max(1, P - hk);
    std::vector<int> start(nl, -1);
    for (int l = 0; l < nl; ++l) {
        if (h[l] < 0) continue;
        HwLane ln; memset(&ln, 0, sizeof ln);
        ln.ncols = std::min(end[l] + 1, P + hk); ln.h = h[l];
        const long e = end[l];
        auto text = [&](int32_t jb, uint32_t &wl, uint32_t &wh) {
            wl = wh = 0;
            for (int x = 0; x < 32; ++x) { wl |= T[l].bit(T[l].lo, e - jb - x) << x; wh |= T[l].bit(T[l].hi, e - jb - x) << x; }
        };
        hw_run<W, HW_START>(B, ln, r_lo, r_hi, text, alone, nosink);
        if (ln.r_pl < 1) { out[5 * l] = -4; continue; }
        start[l] = end[l] - (ln.r_pl - 1);
    }
    // ---- TRACE + walk ----
    int smax = 0;
    for (int l = 0; l < nl; ++l) if (start[l] >= 0) smax = std::max(smax, end[l] - start[l] + 1);
    HwTile C; C.P = P; C.a0 = -hk; C.ncols_max = smax; C.jx = 1;
    if (W <= 2) {
        // checkpoints every HW_SEG columns, the walk segment by segment on recomputed columns (what k_hw_finish<1|2> does)
        int smin = 1 << 30;
        for (int l = 0; l < nl; ++l) if (start[l] >= 0) smin = std::min(smin, end[l] - start[l] + 1);
        C.jx = smin;
        for (int l = 0; l < nl; ++l) {
            if (start[l] < 0) continue;
            HwLane ln; memset(&ln, 0, sizeof ln);
            const int ms = end[l] - start[l] + 1;
            ln.ncols = ms; ln.h = h[l];
            const long s0 = start[l];
            auto text = [&](int32_t jb, uint32_t &wl, uint32_t &wh) {
                wl = wh = 0;
                for (int x = 0; x < 32; ++x) { wl |= T[l].bit(T[l].lo, s0 + jb + x) << x; wh |= T[l].bit(T[l].hi, s0 + jb + x) << x; }
            };
            std::vector<uint64_t> ck((size_t)(smax / HW_SEG + 2) * 2 * W, 0);
            auto keep = [&](int32_t c, int w, uint64_t vp, uint64_t vn) { ck[((size_t)c * 2) * W + w] = vp; ck[((size_t)c * 2 + 1) * W + w] = vn; };
            hw_run<W, HW_TRACE_CK>(C, ln, f_lo, f_hi, text, alone, keep);
            if (ln.r_final != h[l]) { out[5 * l] = -5; continue; }
            int32_t lead = 0;
            if (start[l] == 0) {
                int32_t wi = P, wj = ms;
                bool ok = true;
                uint64_t seg[HW_SEG * 2 * W];
                for (int32_t sg = (smax - 1) / HW_SEG; sg >= 0 && ok; --sg) {
                    const int32_t c0 = sg * HW_SEG;
                    if (!(wi > 0 && wj > c0)) continue;
                    BandLane<W> L2;
                    if (sg == 0) hw_trace_init<W>(C, L2);
                    else for (int w = 0; w < W; ++w) { L2.VP[w] = ck[((size_t)sg * 2) * W + w]; L2.VN[w] = ck[((size_t)sg * 2 + 1) * W + w]; }
                    uint32_t wl = 0, wh = 0;
                    for (int x = 0; x < 32; ++x) { wl |= T[l].bit(T[l].lo, s0 + c0 + x) << x; wh |= T[l].bit(T[l].hi, s0 + c0 + x) << x; }
                    uint64_t VPs[HW_SEG][W], HPs[HW_SEG][W];
                    hw_trace_segment<W>(C, c0, L2, f_lo, f_hi, wl, wh, [&](int jj, int w, uint64_t vp, uint64_t hp) {
                        seg[(jj * 2) * W + w] = vp; seg[(jj * 2 + 1) * W + w] = hp;
                        VPs[jj][w] = vp; HPs[jj][w] = hp;
                    });
                    // one word: the walk on registers, as k_hw_finish<1> does; two words: through the buffer (LDS on the device)
                    if (W == 1) ok = hw_walk_segment_regs<W>(C.a0, c0, wi, wj, VPs, HPs);
                    else ok = hw_walk_segment<W>(C.a0, c0, wi, wj, [&](int jj, int which, int w) -> uint64_t { return seg[(jj * 2 + which) * W + w]; });
                }
                lead = !ok ? -1 : (wj == 0 ? wi : 0);
            }
            if (lead < 0) { out[5 * l] = -6; continue; }
            out[5 * l] = h[l]; out[5 * l + 1] = start[l]; out[5 * l + 2] = end[l]; out[5 * l + 3] = lead; out[5 * l + 4] = ln.r_trail;
        }
        return;
    }
    for (int l = 0; l < nl; ++l) {
        if (start[l] < 0) continue;
        HwLane ln; memset(&ln, 0, sizeof ln);
        const int ms = end[l] - start[l] + 1;
        ln.ncols = ms; ln.h = h[l];
        std::vector<uint64_t> tr((size_t)(ms + 1) * 2 * W, 0);
        const long s0 = start[l];
        auto text = [&](int32_t jb, uint32_t &wl, uint32_t &wh) {
            wl = wh = 0;
            for (int x = 0; x < 32; ++x) { wl |= T[l].bit(T[l].lo, s0 + jb + x) << x; wh |= T[l].bit(T[l].hi, s0 + jb + x) << x; }
        };
        const bool packed = hw_packable(W, h[l], hk);
        const int32_t pshift = hk - h[l] - 1;
        auto sink = [&](int32_t j, int w, uint64_t vp, uint64_t hp) {
            if (packed) tr[((size_t)j * 2) * W + w] = hw_pack(vp, hp, pshift);
            else { tr[((size_t)j * 2) * W + w] = vp; tr[((size_t)j * 2 + 1) * W + w] = hp; }
        };
        hw_run<W, HW_TRACE>(C, ln, f_lo, f_hi, text, alone, sink);
        if (ln.r_final != h[l]) { out[5 * l] = -5; continue; }
        auto load = [&](int32_t j, int which, int w) -> uint64_t {
            if (packed) return hw_unpack(tr[((size_t)j * 2) * W + w], which, pshift);
            return tr[((size_t)j * 2 + which) * W + w];
        };
        const int32_t lead = start[l] == 0 ? hw_walk<W>(P, C.a0, ms, load) : 0;     // smallest start > 0: no leading insertion run (hw.hpp)
        if (lead < 0) { out[5 * l] = -6; continue; }
        out[5 * l] = h[l]; out[5 * l + 1] = start[l]; out[5 * l + 2] = end[l]; out[5 * l + 3] = lead; out[5 * l + 4] = ln.r_trail;
    }
}

extern "C" void emul_hw_tile(int W, const char *q, int P, int nl, const char **ts, const int *tl, const int *ks, int32_t *out)
{
    switch (W) {
    case 1: tile<1>(q, P, nl, ts, tl, ks, out); break;
    case 2: tile<2>(q, P, nl, ts, tl, ks, out); break;
    case 4: tile<4>(q, P, nl, ts, tl, ks, out); break;
    case 8: tile<8>(q, P, nl, ts, tl, ks, out); break;
    default: for (int l = 0; l < nl; ++l) out[5 * l] = -99;
    }
}
