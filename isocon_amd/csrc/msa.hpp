// msa.hpp -- consensus correction on a partition's multi-alignment matrix.
//
// Device part of /root/reference/modules/correction_module.py:260-446 once the matrix exists (the matrix itself -- with the
// reference's insertion-slot layout -- is assembled on the host, isocon_amd/functions.py:msa_matrix):
//   k_msa_col_stats  : per column the weighted counts of A, C, G, T, '-' (position frequency matrix, functions.py:526-536),
//                      the majority symbol (first maximum in that order), whether it is unique, and the partition's totals
//                      of the three error classes over the unambiguous columns (correction_module.py:296-307);
//   k_msa_row_correct: one wavefront per read: its correctable positions (unambiguous majority differs), their frequency
//                      own_count / class_total in double precision, the ceil(n/2)-th smallest frequency, and the
//                      replacement of every position at or below it by the majority symbol (:329-402);
//   k_msa_row_lengths / k_msa_strip: the corrected rows without their gap symbols, packed.
// Byte-matrix work, HBM-bound: the matrix is read three times and written once.
#pragma once
#include "common.hpp"

namespace isocon {

static constexpr int MSA_MAX_CAND = 2048;      // correctable positions per read held in LDS (more: the row is run again with
                                               // its list in a global scratch row, k_msa_row_correct<true>)

// scratch accessors: LDS directly; global scratch through device-scope atomics (lanes read what other lanes of the wave wrote)
template <bool GLOBAL> __device__ __forceinline__ void msa_put(double *fq, uint32_t *cl, uint32_t at, double f, uint32_t col)
{
    if (GLOBAL) {
        __hip_atomic_store((unsigned long long *)fq + at, (unsigned long long)__double_as_longlong(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cl + at, col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else { fq[at] = f; cl[at] = col; }
}
template <bool GLOBAL> __device__ __forceinline__ double msa_freq(const double *fq, uint32_t i)
{
    if (GLOBAL) return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)fq + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return fq[i];
}
template <bool GLOBAL> __device__ __forceinline__ uint32_t msa_col(const uint32_t *cl, uint32_t i)
{
    if (GLOBAL) return __hip_atomic_load(cl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return cl[i];
}

__device__ __forceinline__ int msa_sym(uint8_t c)      // A C G T - -> 0..4
{
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
}

// counts[s * ncols + col]; maj[col] = majority symbol index; flags[col] bit 0 = unambiguous
__global__ __launch_bounds__(256) void k_msa_col_stats(const uint8_t *__restrict__ M, uint32_t nr, uint32_t ncols, const int32_t *__restrict__ degree,
                                                        int32_t *__restrict__ counts, uint8_t *__restrict__ maj, uint8_t *__restrict__ flags,
                                                        unsigned long long *__restrict__ class_tot /* ins, del, subs */)
{
    const uint32_t col = blockIdx.x * 256 + threadIdx.x;
    long long ci = 0, cd = 0, cs = 0;
    if (col < ncols) {
        int32_t c[5] = {0, 0, 0, 0, 0};
        for (uint32_t r = 0; r < nr; ++r) {
            const int sidx = msa_sym(M[(size_t)r * ncols + col]);
            const int32_t d = degree[r];
#pragma unroll
            for (int k = 0; k < 5; ++k) c[k] += sidx == k ? d : 0;
        }
        int best = 0, ties = 1;
#pragma unroll
        for (int k = 1; k < 5; ++k) {
            if (c[k] > c[best]) { best = k; ties = 1; }
            else if (c[k] == c[best]) ++ties;
        }
        int32_t tot = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) { counts[(size_t)k * ncols + col] = c[k]; tot += c[k]; }
        maj[col] = (uint8_t)best;
        flags[col] = ties == 1 ? 1 : 0;
        if (ties == 1) {
            if (best == 4) ci = tot - c[4];
            else { cd = c[4]; cs = tot - c[best] - c[4]; }
        }
    }
    // block reduction of the three totals
    __shared__ long long red[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) { ci += __shfl_xor(ci, o, 64); cd += __shfl_xor(cd, o, 64); cs += __shfl_xor(cs, o, 64); }
    if (lane == 0) { red[0][wave] = ci; red[1][wave] = cd; red[2][wave] = cs; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const long long t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (t) atomicAdd(class_tot + threadIdx.x, (unsigned long long)t);
    }
}

// One wavefront per row.  out row = corrected row; n_cand[r] = number of correctable positions.  GLOBAL = false: the
// list lives in LDS, a row with more than MSA_MAX_CAND positions gets n_cand = -1 and is left for the second launch;
// GLOBAL = true: rows come from `row_list` (n_list of them) and keep their lists in g_freq / g_col (ncols entries per row).
template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_msa_row_correct(const uint8_t *__restrict__ M, uint8_t *__restrict__ out, uint32_t nr, uint32_t ncols,
                                                          const int32_t *__restrict__ degree, const int32_t *__restrict__ counts,
                                                          const uint8_t *__restrict__ maj, const uint8_t *__restrict__ flags,
                                                          const unsigned long long *__restrict__ class_tot, int32_t *__restrict__ n_cand,
                                                          const uint32_t *__restrict__ row_list, uint32_t n_list, double *g_freq, uint32_t *g_col)
{
    __shared__ double s_freq[GLOBAL ? 1 : 4][GLOBAL ? 1 : MSA_MAX_CAND];
    __shared__ uint32_t s_col[GLOBAL ? 1 : 4][GLOBAL ? 1 : MSA_MAX_CAND];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * 4 + wave;
    if (slot >= (GLOBAL ? n_list : nr)) return;
    const uint32_t r = GLOBAL ? row_list[slot] : slot;
    const uint32_t cap = GLOBAL ? ncols : (uint32_t)MSA_MAX_CAND;
    const uint8_t *row = M + (size_t)r * ncols;
    uint8_t *orow = out + (size_t)r * ncols;
    const char SYM[5] = {'A', 'C', 'G', 'T', '-'};
    const double d_ins = (double)(class_tot[0] > 0 ? class_tot[0] : 1ull);
    const double d_del = (double)(class_tot[1] > 0 ? class_tot[1] : 1ull);
    const double d_sub = (double)(class_tot[2] > 0 ? class_tot[2] : 1ull);
    const bool single = degree[r] == 1;
    double *fq = GLOBAL ? g_freq + (size_t)slot * ncols : s_freq[wave];
    uint32_t *cl = GLOBAL ? g_col + (size_t)slot * ncols : s_col[wave];
    uint32_t n = 0;                 // candidates so far (wave-uniform)
    bool overflow = false;
    for (uint32_t c0 = 0; c0 < ncols; c0 += 64) {
        const uint32_t col = c0 + lane;
        uint8_t v = '-';
        bool cand = false;
        if (col < ncols) {
            v = row[col];
            orow[col] = v;
            cand = single && flags[col] && v != (uint8_t)SYM[maj[col]];
        }
        const unsigned long long mask = __ballot(cand);
        if (mask) {
            const uint32_t at = n + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (cand) {
                if (at < cap) {
                    const int mj = maj[col];
                    const double own = (double)counts[(size_t)msa_sym(v) * ncols + col];
                    msa_put<GLOBAL>(fq, cl, at, own / (mj == 4 ? d_ins : (v == '-' ? d_del : d_sub)), col);
                }
            }
            n += (uint32_t)__popcll(mask);
            if (n > cap) overflow = true;
        }
    }
    if (overflow) { if (lane == 0) n_cand[r] = -1; return; }
    if (lane == 0) n_cand[r] = (int32_t)n;
    if (n == 0) return;
    if (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // threshold = k-th smallest frequency, k = ceil(n / 2): the value f with  #{< f} < k <= #{<= f}
    const uint32_t k = (n + 1) / 2;
    double thr = 0.0;
    bool have = false;
    for (uint32_t i0 = 0; i0 < n && !have; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool mine = false;
        double f = 0.0;
        if (i < n) {
            f = msa_freq<GLOBAL>(fq, i);
            uint32_t lt = 0, le = 0;
            for (uint32_t j = 0; j < n; ++j) { const double g = msa_freq<GLOBAL>(fq, j); lt += g < f; le += g <= f; }
            mine = lt < k && k <= le;
        }
        const unsigned long long m2 = __ballot(mine);
        if (m2) {
            const int src = __ffsll((long long)m2) - 1;
            thr = __shfl(f, src, 64);
            have = true;
        }
    }
    for (uint32_t i = lane; i < n; i += 64)
        if (msa_freq<GLOBAL>(fq, i) <= thr) { const uint32_t c = msa_col<GLOBAL>(cl, i); orow[c] = (uint8_t)SYM[maj[c]]; }
}

__global__ __launch_bounds__(256) void k_msa_row_lengths(const uint8_t *__restrict__ rows, uint32_t nr, uint32_t ncols, uint32_t *__restrict__ len)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * 4 + wave;
    if (r >= nr) return;
    const uint8_t *row = rows + (size_t)r * ncols;
    uint32_t c = 0;
    for (uint32_t col = lane; col < ncols; col += 64) c += row[col] != '-';
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) len[r] = c;
}

__global__ __launch_bounds__(256) void k_msa_strip(const uint8_t *__restrict__ rows, uint32_t nr, uint32_t ncols, const uint64_t *__restrict__ off,
                                                    uint8_t *__restrict__ packed)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * 4 + wave;
    if (r >= nr) return;
    const uint8_t *row = rows + (size_t)r * ncols;
    uint8_t *dst = packed + off[r];
    uint32_t at = 0;
    for (uint32_t c0 = 0; c0 < ncols; c0 += 64) {
        const uint32_t col = c0 + lane;
        const uint8_t v = col < ncols ? row[col] : (uint8_t)'-';
        const bool keep = v != '-';
        const unsigned long long mask = __ballot(keep);
        if (keep) dst[at + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = v;
        at += (uint32_t)__popcll(mask);
    }
}

}  // namespace isocon
