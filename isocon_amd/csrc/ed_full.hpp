// ed_full.hpp -- un-banded global edit distance for arbitrary k: one wavefront per pair, Myers' 64-row blocks
// laid out one per lane and driven as a systolic array (lane l works on text column s-l at step s, the
// horizontal delta and the text base travel one lane down per step).  Last-resort path for pairs whose distance
// exceeds the widest band (k > 511): unrelated sequences in edlib_align_sequences (unbounded k,
// /root/reference/modules/edlib_alignment_module.py:111) and isolated queries in the NN search
// (k = len(seq1), /root/reference/modules/nearest_neighbor_graph.py:129,156).
#pragma once
#include "common.hpp"

namespace isocon {

__device__ __forceinline__ int myers_block_dev(uint64_t Pv, uint64_t Mv, uint64_t Eq, int hin, uint64_t &Pvo, uint64_t &Mvo)
{
    const uint64_t hneg = (uint64_t)(hin < 0);
    const uint64_t Xv = Eq | Mv;
    Eq |= hneg;
    const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    uint64_t Ph = Mv | ~(Xh | Pv);
    uint64_t Mh = Pv & Xh;
    const int hout = (int)(Ph >> 63) - (int)(Mh >> 63);
    Ph = (Ph << 1) | (uint64_t)(hin > 0);
    Mh = (Mh << 1) | hneg;
    Pvo = Mh | ~(Xv | Ph);
    Mvo = Ph & Xv;
    return hout;
}

// grid = n_pairs blocks of 64 threads; dynamic LDS = max text length (bytes) when any pattern needs > 4096 rows.
__global__ __launch_bounds__(64) void k_ed_full(DevStore S, const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb,
                                                 const int32_t *__restrict__ pk, int32_t *__restrict__ out, uint32_t n_pairs)
{
    extern __shared__ int8_t hbound[];
    const uint32_t pair = blockIdx.x;
    if (pair >= n_pairs) return;
    const int lane = threadIdx.x;
    uint32_t ia = pa[pair], ib = pb[pair];
    int32_t m = S.lens[ia], n = S.lens[ib];
    if (m > n) { uint32_t t = ia; ia = ib; ib = t; int32_t t2 = m; m = n; n = t2; }  // pattern = shorter
    ia = (uint32_t)uniform_i32((int32_t)ia);
    ib = (uint32_t)uniform_i32((int32_t)ib);
    m = uniform_i32(m);
    n = uniform_i32(n);
    const int32_t k = pk ? pk[pair] : -1;
    if (m == 0) {
        if (lane == 0) out[pair] = (k < 0 || n <= k) ? n : -1;
        return;
    }
    const uint64_t *planes = S.planes;
    const uint32_t nseq = S.n;
    const int32_t nchunks = (int32_t)S.nchunks;
    const int32_t passes = (m + 4095) >> 12;
    int32_t result = 0;
    for (int32_t pass = 0; pass < passes; ++pass) {
        const int32_t chunk = pass * 64 + lane;
        const int32_t row0 = chunk * 64;
        uint64_t lo = 0, hi = 0;
        if (chunk < nchunks) {
            lo = planes[((size_t)chunk * nseq + ia) * 2];
            hi = planes[((size_t)chunk * nseq + ia) * 2 + 1];
        }
        const int32_t rem = m - row0;
        const uint64_t vmask = rem <= 0 ? 0 : (rem >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << rem) - 1));
        const uint64_t peq0 = ~lo & ~hi & vmask, peq1 = lo & ~hi & vmask, peq2 = ~lo & hi & vmask, peq3 = lo & hi & vmask;
        uint64_t Pv = ~(uint64_t)0, Mv = 0;
        int32_t score = row0 + 64;
        int32_t packed = 1;  // (char << 2) | (hout + 1)
        uint64_t tlo = 0, thi = 0;
        const int32_t steps = n + 63;
        for (int32_t s = 0; s < steps; ++s) {
            if ((s & 63) == 0) {   // wave-uniform: next 64 text bases
                const int32_t tc = s >> 6;
                if (tc < nchunks) {
                    tlo = planes[((size_t)tc * nseq + ib) * 2];
                    thi = planes[((size_t)tc * nseq + ib) * 2 + 1];
                } else { tlo = 0; thi = 0; }
            }
            const int32_t recv = __shfl_up(packed, 1, 64);
            int32_t ch, hin;
            if (lane == 0) {
                ch = (int32_t)(((tlo >> (s & 63)) & 1) | (((thi >> (s & 63)) & 1) << 1));
                hin = (pass == 0) ? 1 : (s < n ? (int32_t)hbound[s] : 0);
            } else {
                ch = recv >> 2;
                hin = (recv & 3) - 1;
            }
            const int32_t col = s - lane;
            int32_t hout = 0;
            if (col >= 0 && col < n) {
                const uint64_t Eq = (ch & 2) ? ((ch & 1) ? peq3 : peq2) : ((ch & 1) ? peq1 : peq0);
                hout = myers_block_dev(Pv, Mv, Eq, hin, Pv, Mv);
                score += hout;
                if (lane == 63 && pass + 1 < passes) hbound[col] = (int8_t)hout;
            }
            packed = (ch << 2) | (hout + 1);
        }
        if (pass == passes - 1) {
            const int32_t lstar = ((m - 1) >> 6) - pass * 64;
            const int32_t bit = (m - 1) & 63;
            const uint64_t above = bit == 63 ? 0 : (~(uint64_t)0 << (bit + 1));
            const int32_t dloc = score - __popcll(Pv & above) + __popcll(Mv & above);
            result = __shfl(dloc, lstar, 64);
        }
        __syncthreads();
    }
    if (lane == 0) out[pair] = (k < 0 || result <= k) ? result : -1;
}

}  // namespace isocon
