// nn_multi.hpp -- the listed launch of the main pass with SEVERAL entry tables per workgroup.
//
// k_nn_scan_refill (nn.hpp) gives a workgroup one chunk (one entry's table in LDS, up to 2048 of its pairs): once the chunk's list is
// exhausted the lanes finish their last pairs one by one and the workgroup's LDS stays reserved until the slowest is done -- measured at
// C3: 83 % of the executed lane-columns live, 3.1 of 6 possible waves per SIMD resident on average.  Here a workgroup of 16 waves (one
// per CU, 4 waves per SIMD) builds the tables of SLOTS chunks at its start (chunks i, i + G, i + 2G of the table sorted by size: a
// large, a medium and a small one) and its lanes take pairs from chunk after chunk: a lane that finishes a pair of the first chunk
// while that chunk's list is exhausted continues with a pair of the second, so only the LAST chunk's drain leaves lanes idle.  What
// was wave-uniform per workgroup (entry, its length, its table) becomes a property of the lane's pair: the queue entry carries the
// slot.  Same pairs, same thresholds, same results; no barrier after the tables are built.
// Band step, table layout, text store, admission rules: nn.hpp (64-row form and, HALF, the 32-row form).
#pragma once
#include "nn.hpp"

namespace isocon {

template <int NWAVES, int SLOTS, bool HALF>
__global__ __launch_bounds__(NWAVES * 64) void k_nn_scan_multi(DevStore S, NNParams P, const uint32_t *__restrict__ text, uint32_t text_stride,
                                                                uint32_t n_chunks, uint32_t plane_dwords)
{
    constexpr int ROWS = HALF ? 32 : 64;
    extern __shared__ uint32_t tw[];                 // SLOTS tables of 4 planes x plane_dwords + the plane of ones (2 ROWS + 32 dwords)
    __shared__ uint32_t s_next[SLOTS];
    __shared__ uint32_t s_ring[NWAVES][NN_RING][2];
    __shared__ uint32_t s_q[SLOTS], s_m[SLOTS], s_count[SLOTS], s_pb[SLOTS], s_flags[SLOTS];
    __shared__ unsigned long long s_begin[SLOTS];
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    const int32_t wave = threadIdx.x >> 6;
    const int32_t lane = threadIdx.x & 63;
    const uint32_t slot_dwords = 4u * plane_dwords + 2u * ROWS + 32u;
    int nslots = 0;
    for (int s = 0; s < SLOTS; ++s) {
        const unsigned long long cid = (unsigned long long)blockIdx.x + (unsigned long long)s * gridDim.x;
        if (cid >= n_chunks) break;
        nslots = s + 1;
        const NNChunk ch = P.chunks[cid];
        const uint32_t q = ch.slot;
        const int32_t m = S.lens[q];
        const int32_t E = (m + 3 * ROWS + 31) & ~31;          // plane length in dwords (<= plane_dwords: the host sized it by the longest entry)
        uint32_t *t = tw + (size_t)s * slot_dwords;
        const uint64_t *planes = S.planes;
        const uint32_t nseq = S.n;
        const int32_t nchunks = (int32_t)S.nchunks;
        auto chunk_lo = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + q) * 2] : 0; };
        auto chunk_hi = [&](int32_t ci) -> uint64_t { return ci < nchunks ? planes[((size_t)ci * nseq + q) * 2 + 1] : 0; };
        for (int32_t e = threadIdx.x; e < E; e += NWAVES * 64) {
            const int32_t o = e - (ROWS + 31);           // entry 32 + (ROWS - 1) <-> bit offset 0
            const uint32_t lo = (uint32_t)stream64(chunk_lo, o), hi = (uint32_t)stream64(chunk_hi, o);
            const int32_t r0 = o < 0 ? -o : 0, r1 = (m - o) < 32 ? (m - o) : 32;       // valid rows [r0, r1)
            uint32_t v = 0;
            if (r1 > r0) v = (r1 >= 32 ? 0xffffffffu : ((1u << r1) - 1u)) & ~(r0 >= 32 ? 0xffffffffu : ((1u << r0) - 1u));
            t[e] = ~lo & ~hi & v;
            t[E + e] = lo & ~hi & v;
            t[2 * E + e] = ~lo & hi & v;
            t[3 * E + e] = lo & hi & v;
        }
        for (int32_t e = threadIdx.x; e < 2 * ROWS + 32; e += NWAVES * 64) t[4 * E + e] = 0xffffffffu;
        if (threadIdx.x == 0) {
            s_next[s] = 0; s_q[s] = q; s_m[s] = (uint32_t)m; s_count[s] = ch.count; s_begin[s] = ch.begin; s_pb[s] = (uint32_t)E * 4u;
            s_flags[s] = (P.qflag[q] != 0 ? 1u : 0u) | (P.tflag[q] != 0 ? 2u : 0u);
        }
    }
    __syncthreads();
    if (nslots == 0) return;
    const uint32_t tbase0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t *)tw;
    const uint32_t idle_blk = tbase0 + (uint32_t)(lane & 31) * 4u;
    uint32_t(*ring)[2] = s_ring[wave];
    const uint64_t lt_mask = ((uint64_t)1 << lane) - 1;

    bool run = false, upd_s = false, upd_l = false;
    uint32_t tid = 0, blk = idle_blk, nsh = 0, q_l = 0, plane_bytes = s_pb[0];
    int32_t n_t = 0, k_eff = -1, nv = 0, col = 0, bstar = 0, m_l = 0;
    uint32_t cur[5] = {0, 0, 0, 0, 0};
    const uint32_t *tp = text;
    uint64_t VP = ~(uint64_t)0, VN = 0;          // 64-row state
    uint32_t hvp = ~0u, hvn = 0u;                // 32-row state
    uint32_t ztop = 0;
    uint32_t qhead = 0, qcount = 0;
    int adm = 0;                                  // the slot this wave draws from
    bool exhausted = false;
    uint32_t n_pairs = 0, n_batches = 0, n_blocks = 0, n_live = 0;

    auto load5 = [](const uint32_t *p, uint32_t (&d)[5]) {
        const TextQuad t4 = *reinterpret_cast<const TextQuad *>(p);
        d[0] = t4.x; d[1] = t4.y; d[2] = t4.z; d[3] = t4.w; d[4] = p[4];
    };

    for (;;) {
        const uint64_t freemask = __ballot(!run);
        const uint32_t nfree = (uint32_t)__popcll(freemask);
        while (!exhausted && qcount < nfree && qcount + 64 <= (uint32_t)NN_RING) {
            uint32_t c0 = 0;
            if (lane == 0) c0 = atomicAdd(&s_next[adm], 64u);
            c0 = (uint32_t)uniform_i32((int32_t)c0);
            const uint32_t l_count = (uint32_t)uniform_i32((int32_t)s_count[adm]);
            if (c0 >= l_count) {                 // this chunk's list is handed out: on to the workgroup's next chunk
                adm += 1;
                if (adm >= nslots) exhausted = true;
                continue;
            }
            const uint32_t q = (uint32_t)uniform_i32((int32_t)s_q[adm]);
            const int32_t m = uniform_i32((int32_t)s_m[adm]);
            const uint32_t fl = (uint32_t)uniform_i32((int32_t)s_flags[adm]);
            const bool q_isq = (fl & 1u) != 0;
            const unsigned long long l_begin = s_begin[adm];
            // the survivors of the bound, already filtered by roles, window and bound (k_nn_survivors): one coalesced load
            const bool inr = c0 + (uint32_t)lane < l_count;
            const uint32_t e = inr ? P.list[l_begin + c0 + (uint32_t)lane] : q;
            const uint32_t pid = e & 0x3fffffffu;
            const int32_t np = S.lens[pid];
            const bool us = inr && (e & 0x40000000u) != 0;
            const bool ul = inr && (e & 0x80000000u) != 0;
            int32_t bs = NN_INF;
            if (q_isq) bs = uniform_i32(load_relaxed_agent(P.best + q));
            int32_t ks = -1, kl = -1;
            if (us) ks = bs < m ? bs : m;
            if (ul) { const int32_t bl = load_relaxed_agent(P.best + pid); kl = bl < np ? bl : np; }
            int32_t k = ks > kl ? ks : kl;
            if (k > P.kcap) k = P.kcap;
            const int32_t d = m - np, ad = d < 0 ? -d : d;
            bool accept = inr && k >= 0 && ad <= k;
            const bool triv = accept && (m == 0 || np == 0);
            if (__ballot(triv) != 0) {
                bool hs = false, hl = false;
                if (triv && ad >= P.min_d) {
                    if (us && ad <= m) { const int32_t old = atomicMin(P.best + q, ad); hs = ad <= old; }
                    if (ul && ad <= np) { const int32_t old = atomicMin(P.best + pid, ad); hl = ad <= old; }
                }
                nn_append(P, hs, (int32_t)q, (int32_t)pid, ad);
                nn_append(P, hl, (int32_t)pid, (int32_t)q, ad);
                accept = accept && !triv;
            }
            const uint64_t am = __ballot(accept);
            if (accept) {
                const uint32_t slot = (qhead + qcount + (uint32_t)__popcll(am & lt_mask)) % (uint32_t)NN_RING;
                int32_t a0 = lane_emin(d, k);
                if (a0 < -(ROWS - 1)) a0 = -(ROWS - 1);
                ring[slot][0] = pid | (us ? 0x40000000u : 0u) | (ul ? 0x80000000u : 0u);
                ring[slot][1] = (uint32_t)np | ((uint32_t)k << 14) | ((uint32_t)(-a0) << 20) | ((uint32_t)adm << 26);      // 14 + 6 + 6 + 2 bits
            }
            qcount += (uint32_t)__popcll(am);
            n_pairs += (uint32_t)__popcll(am);
            n_batches += 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (nfree && qcount) {
            const uint32_t rank = (uint32_t)__popcll(freemask & lt_mask);
            if (!run && rank < qcount) {
                const uint32_t slot = (qhead + rank) % (uint32_t)NN_RING;
                const uint32_t e0w = ring[slot][0], e1w = ring[slot][1];
                tid = e0w & 0x3fffffffu;
                upd_s = (e0w >> 30) & 1u;
                upd_l = (e0w >> 31) & 1u;
                n_t = (int32_t)(e1w & 0x3fffu);
                k_eff = (int32_t)((e1w >> 14) & 63u);
                nv = (int32_t)((e1w >> 20) & 63u);
                const uint32_t sl = e1w >> 26;
                q_l = s_q[sl];
                m_l = (int32_t)s_m[sl];
                plane_bytes = s_pb[sl];
                const uint32_t tb = tbase0 + sl * slot_dwords * 4u;
                bstar = m_l - n_t + nv;                        // in [0, ROWS - 1]
                if (HALF) {
                    hvp = nv <= 0 ? ~0u : (nv >= 32 ? 0u : (~0u << nv));
                    hvn = ~hvp;
                } else {
                    VP = nv <= 0 ? ~(uint64_t)0 : (nv >= 64 ? 0 : (~(uint64_t)0 << nv));
                    VN = ~VP;
                }
                const uint32_t e0 = (uint32_t)(ROWS - 1 - nv);
                const uint32_t phi = (e0 - (uint32_t)lane) & 31u;
                ztop = 0u - phi;
                col = -(int32_t)phi;
                blk = tb + (32u + e0 - phi) * 4u;
                nsh = 4u * ((32u - phi) & 7u);
                tp = text + (size_t)tid * text_stride + ((32u - phi) >> 3);
                load5(tp, cur);
                run = true;
            }
            const uint32_t taken = nfree < qcount ? nfree : qcount;
            qhead = (qhead + taken) % (uint32_t)NN_RING;
            qcount -= taken;
        }
        const uint64_t runmask = __ballot(run);
        if (runmask == 0) {
            if (exhausted && qcount == 0) break;
            continue;
        }
        const uint32_t w0 = __builtin_amdgcn_alignbit(cur[1], cur[0], nsh), w1 = __builtin_amdgcn_alignbit(cur[2], cur[1], nsh),
                       w2 = __builtin_amdgcn_alignbit(cur[3], cur[2], nsh), w3 = __builtin_amdgcn_alignbit(cur[4], cur[3], nsh);
        if (run && col + 32 < n_t) load5(tp + 4, cur);
        // all 32 columns unrolled, immediate table offsets; a block in which some lane's text ends runs that lane's columns behind the
        // end as virtual columns (band_core.hpp): every block is 32 columns for every lane
        uint32_t zreg = 0;
        if (__ballot(run && col + 32 > n_t) == 0) {
#pragma unroll
            for (int jj = 0; jj < 32; ++jj) {
                const uint32_t code = (uint32_t)__builtin_amdgcn_ubfe(jj < 8 ? w0 : jj < 16 ? w1 : jj < 24 ? w2 : w3, 4 * (jj & 7), 3);
                lds_u32 *pe = (lds_u32 *)(uintptr_t)(__umul24(code, plane_bytes) + blk);
                if (HALF) band_step_eq32(hvp, hvn, zreg, pe[jj]);
                else band_step_eq64z(VP, VN, zreg, ((uint64_t)pe[jj + 32] << 32) | pe[jj]);
            }
        } else {
            const int32_t rem = n_t - col;
            const uint32_t act = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
#pragma unroll
            for (int jj = 0; jj < 32; ++jj) {
                const uint32_t code = (uint32_t)__builtin_amdgcn_ubfe(jj < 8 ? w0 : jj < 16 ? w1 : jj < 24 ? w2 : w3, 4 * (jj & 7), 3);
                lds_u32 *pe = (lds_u32 *)(uintptr_t)(__umul24(code, plane_bytes) + blk);
                const uint32_t real = (uint32_t)__builtin_amdgcn_sbfe(act, jj, 1);
                if (HALF) band_step_eq32_tail(hvp, hvn, zreg, pe[jj], real);
                else band_step_eq64z_tail(VP, VN, zreg, ((uint64_t)pe[jj + 32] << 32) | pe[jj], real);
            }
        }
        ztop += (uint32_t)__popc(zreg);
        n_blocks += 1;
        n_live += (uint32_t)__popcll(runmask);
        col += 32;
        const bool fin = run && col >= n_t;
        int32_t dv;
        if (HALF) {
            const uint32_t lm = bstar <= 0 ? 0u : (bstar >= 32 ? ~0u : ((1u << bstar) - 1u));
            dv = nv + col - (int32_t)ztop + __popc(hvp & lm) - __popc(hvn & lm);          // (virtual columns count on both sides)
        } else {
            const uint64_t lm = bstar <= 0 ? 0 : (bstar >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << bstar) - 1));
            dv = nv + col - (int32_t)ztop + popc64(VP & lm) - popc64(VN & lm);
        }
        int32_t r = -1;
        if (fin) { r = dv <= k_eff ? dv : -1; run = false; }
        else if (run && dv > k_eff) run = false;
        if (__ballot(fin && r >= P.min_d) != 0) {
            bool hs = false, hl = false;
            if (fin && r >= P.min_d) {
                if (upd_s && r <= m_l) { const int32_t old = atomicMin(P.best + q_l, r); hs = r <= old; }
                if (upd_l && r <= n_t) { const int32_t old = atomicMin(P.best + tid, r); hl = r <= old; }
            }
            nn_append(P, hs, (int32_t)q_l, (int32_t)tid, r);
            nn_append(P, hl, (int32_t)tid, (int32_t)q_l, r);
        }
        blk += 128u;
        tp += 4;
        if (!run) blk = idle_blk;
    }
    WaveAcc acc;
    acc.pairs = n_pairs; acc.tiles = n_batches; acc.cols = (unsigned long long)n_blocks * 2048ull; acc.live = (unsigned long long)n_live * 32ull;
    nn_flush_acc(P, acc);
}

}  // namespace isocon
