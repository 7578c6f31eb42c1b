// msa_batch.hpp -- ALL partitions of a correction step in one set of launches (SURVEY.md 8(f) rows f1 + f3): the batched forms of
// msa_build.hpp (matrix from CIGAR ops + packed store) and msa.hpp (column statistics, per-read correction, gap stripping).
// /root/reference/modules/correction_module.py:12-75 loops over the partitions (a Pool task each); later correction steps of a run have
// hundreds to thousands of small partitions, and one build + correct call pair per partition is ~20 host synchronisations each
// (2 751 call pairs over the ten steps of the 50 000-read set: 1.3 s).  Here a row knows its partition (part_of_row) and a partition its
// pieces of the concatenated arrays: slots (insertion-slot arrays, len(centre) + 1 entries), columns (ncols entries) and matrix cells.
#pragma once
#include "msa.hpp"
#include "msa_build.hpp"

namespace isocon {

struct MsaBatch {
    const uint32_t *part_of_row;          // [n_rows]
    const uint32_t *first_row;            // [n_parts + 1]: rows of partition p = first_row[p] .. first_row[p + 1]; the first is its centre
    const uint32_t *Lm;                   // [n_parts] length of the centre
    const uint32_t *slot_base;            // [n_parts + 1] offset of the partition's slot arrays (longest, width, col_slot)
    const uint32_t *ncols;                // [n_parts] columns of the partition's matrix (after k_msab_layout)
    const unsigned long long *m_off;      // [n_parts + 1] first cell of the partition's matrix
    const uint32_t *col_base;             // [n_parts + 1] offset of the partition's column arrays (counts, maj, flags)
    uint32_t n_parts, n_rows;
};

__global__ __launch_bounds__(256) void k_msab_ops_scan(DevStore S, MsaBatch B, const uint32_t *__restrict__ row_ids, const uint32_t *__restrict__ ops,
                                                        const unsigned long long *__restrict__ ops_ptr, uint32_t *__restrict__ longest, uint32_t *__restrict__ bad)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= B.n_rows) return;
    const uint32_t p = B.part_of_row[r];
    if (r == B.first_row[p]) return;          // the centre itself
    const uint32_t Lm = B.Lm[p];
    uint32_t *lg = longest + B.slot_base[p];
    uint32_t t = 0, sp = 0;
    for (unsigned long long k = ops_ptr[r]; k < ops_ptr[r + 1]; ++k) {
        const uint32_t op = ops[k], len = op >> 4, code = op & 15u;
        if (code == 3u) { if (t <= Lm) atomicMax(lg + t, len); sp += len; }
        else { t += len; if (code != 2u) sp += len; }
    }
    if (t != Lm || sp != (uint32_t)S.lens[row_ids[r]]) atomicOr(bad, 1u);
}

// one workgroup of 1024 threads per partition
__global__ __launch_bounds__(1024) void k_msab_layout(MsaBatch B, const uint32_t *__restrict__ longest, uint32_t *__restrict__ width, uint32_t *__restrict__ col_slot,
                                                       uint32_t *__restrict__ ncols_out)
{
    __shared__ unsigned long long wave_sums[16];
    __shared__ unsigned long long carry;
    const uint32_t p = blockIdx.x;
    const uint32_t Lm = B.Lm[p], sb = B.slot_base[p];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base <= Lm; base += 1024u) {
        const uint32_t t = base + threadIdx.x;
        uint32_t w = 0;
        if (t <= Lm) {
            const uint32_t lg = longest[sb + t];
            w = lg > 1u ? lg + 2u : 1u;
            width[sb + t] = w;
        }
        unsigned long long total = 0;
        const unsigned long long step = t <= Lm ? (unsigned long long)w + (t < Lm ? 1ull : 0ull) : 0ull;
        const unsigned long long off = block_exscan_1024(step, wave_sums, &total);
        if (t <= Lm) col_slot[sb + t] = (uint32_t)(carry + off);
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) ncols_out[p] = (uint32_t)carry;
}

// one wave per row; wide records carry the GLOBAL row
__global__ __launch_bounds__(256) void k_msab_fill(DevStore S, MsaBatch B, const uint32_t *__restrict__ row_ids, const uint32_t *__restrict__ ops,
                                                    const unsigned long long *__restrict__ ops_ptr, const uint32_t *__restrict__ longest_all,
                                                    const uint32_t *__restrict__ width_all, const uint32_t *__restrict__ col_slot_all, uint8_t *__restrict__ M_all,
                                                    uint32_t *__restrict__ wide, unsigned long long wide_cap, unsigned long long *__restrict__ wide_count)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * 4u + (uint32_t)wave;
    if (r >= B.n_rows) return;
    const uint32_t p = B.part_of_row[r], r0 = B.first_row[p], Lm = B.Lm[p], sb = B.slot_base[p], n_cols = B.ncols[p];
    const uint32_t *longest = longest_all + sb, *width = width_all + sb, *col_slot = col_slot_all + sb;
    const uint32_t id = row_ids[r];
    uint8_t *row = M_all + B.m_off[p] + (size_t)(r - r0) * n_cols;
    if (r == r0) {
        for (uint32_t t = (uint32_t)lane; t < Lm; t += 64u) row[col_slot[t] + width[t]] = msa_base_char(S, id, t);
        return;
    }
    // The row's ops, 64 at a time: every lane loads one (coalesced), a wave scan gives each op its slot and member position, and the ops are then
    // taken one by one from the lanes' registers -- no chain of dependent global loads (op k + 1 could not be requested before op k had arrived).
    uint32_t t = 0, sp = 0;
    const unsigned long long k_end = ops_ptr[r + 1];
    for (unsigned long long kb = ops_ptr[r]; kb < k_end; kb += 64ull) {
        const uint32_t n_here = (uint32_t)(k_end - kb < 64ull ? k_end - kb : 64ull);
        const uint32_t op_l = (uint32_t)lane < n_here ? ops[kb + (unsigned long long)lane] : 0u;
        const uint32_t len_l = op_l >> 4, code_l = op_l & 15u;
        uint32_t it = code_l == 3u ? 0u : len_l, is = code_l == 2u ? 0u : len_l;          // inclusive scans of the advances in slot / member position
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ot = (uint32_t)__shfl_up((int)it, d, 64), os = (uint32_t)__shfl_up((int)is, d, 64);
            if (lane >= d) { it += ot; is += os; }
        }
        const uint32_t t_l = t + it - (code_l == 3u ? 0u : len_l), sp_l = sp + is - (code_l == 2u ? 0u : len_l);
        // Insertions at wide slots are only LISTED (their place inside the padded longest insertion is decided on the host): every lane files
        // its own op's record, ONE atomic per batch reserves the records (one atomic per record -- 6 10^5 on one address at C3 -- was what the
        // kernel's 3.7 ms were: same-address atomics serialise in L2).
        {
            // (a malformed op stream -- an op that runs past the member row -- ends the row at that op below: the records of the ops behind it
            // are not listed either: the lanes in front of the first op that fails the bound)
            const unsigned long long bad = __ballot((uint32_t)lane < n_here && t_l + (code_l == 3u ? 0u : len_l) > Lm);
            const bool before_bad = bad == 0ull || (uint32_t)lane < (uint32_t)__builtin_ctzll(bad);
            const bool wide_l = before_bad && (uint32_t)lane < n_here && code_l == 3u && t_l <= Lm && longest[t_l] > 1u;
            const unsigned long long wm = __ballot(wide_l);
            if (wm != 0ull) {
                unsigned long long at0 = 0;
                if (lane == 0) at0 = atomicAdd(wide_count, (unsigned long long)__popcll(wm));
                at0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(at0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)at0);
                const unsigned long long at = at0 + (unsigned long long)__popcll(wm & (((unsigned long long)1 << lane) - 1ull));
                if (wide_l && at < wide_cap) {
                    unsigned long long codes = 0;
                    for (uint32_t jj = 0; jj < len_l && jj < 32u; ++jj) {
                        const size_t w = ((size_t)((sp_l + jj) >> 6) * S.n + id) * 2;
                        const uint32_t sh = (sp_l + jj) & 63u;
                        codes |= (((S.planes[w] >> sh) & 1ull) | (((S.planes[w + 1] >> sh) & 1ull) << 1)) << (2u * jj);
                    }
                    uint32_t *e = wide + 8 * at;
                    e[0] = r; e[1] = t_l; e[2] = sp_l; e[3] = len_l; e[4] = (uint32_t)codes; e[5] = (uint32_t)(codes >> 32); e[6] = p; e[7] = 0u;
                }
            }
        }
        for (uint32_t j = 0; j < n_here; ++j) {
            const uint32_t op = (uint32_t)__builtin_amdgcn_readlane((int)op_l, (int)j), len = op >> 4, code = op & 15u;
            const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)t_l, (int)j), spj = (uint32_t)__builtin_amdgcn_readlane((int)sp_l, (int)j);
            if (tj + (code == 3u ? 0u : len) > Lm) return;
            if (code == 3u) {
                if (longest[tj] <= 1u && lane == 0) row[col_slot[tj]] = msa_base_char(S, id, spj);          // (wide slots: listed above)
            } else {
                for (uint32_t i = (uint32_t)lane; i < len; i += 64u)
                    row[col_slot[tj + i] + width[tj + i]] = code == 2u ? (uint8_t)'-' : msa_base_char(S, id, spj + i);
            }
        }
        t += (uint32_t)__shfl((int)it, 63, 64);
        sp += (uint32_t)__shfl((int)is, 63, 64);
    }
}

__global__ __launch_bounds__(256) void k_msab_patch(uint8_t *__restrict__ M_all, MsaBatch B, const uint32_t *__restrict__ patch_row, const uint32_t *__restrict__ patch_col,
                                                     const uint32_t *__restrict__ patch_ptr, const uint8_t *__restrict__ bytes, uint32_t n_patches)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 4u + (uint32_t)wave;
    if (i >= n_patches) return;
    const uint32_t r = patch_row[i], p = B.part_of_row[r];
    uint8_t *dst = M_all + B.m_off[p] + (size_t)(r - B.first_row[p]) * B.ncols[p] + patch_col[i];
    const uint32_t b = patch_ptr[i], e = patch_ptr[i + 1];
    for (uint32_t k = b + (uint32_t)lane; k < e; k += 64u) dst[k - b] = bytes[k];
}

// Column statistics in two launches (round 5; one thread per column walking ALL rows of its partition -- 11 000 at C3 -- took 6.2 ms with 110
// workgroups on the chip):
//   k_msab_col_counts  workgroup b = the 256 columns cbr[3b + 1] .. of partition cbr[3b], rows cbr[3b + 2] .. + MSAB_ROWS_PER_WG (host-built table):
//                      per-column symbol counts of that row chunk, weighted by the rows' degrees, added to counts (zeroed by the host)
//   k_msab_col_finish  workgroup b = the 256 columns cb_col0[b] .. of partition cb_part[b]: majority, tie flag, the partition's error-class totals
static constexpr uint32_t MSAB_ROWS_PER_WG = 256;

__global__ __launch_bounds__(256) void k_msab_col_counts(const uint8_t *__restrict__ M_all, MsaBatch B, const uint32_t *__restrict__ cbr, const int32_t *__restrict__ degree,
                                                          int32_t *__restrict__ counts_all)
{
    const uint32_t p = cbr[3 * blockIdx.x], col = cbr[3 * blockIdx.x + 1] + threadIdx.x, row0 = cbr[3 * blockIdx.x + 2];
    const uint32_t ncols = B.ncols[p], r0 = B.first_row[p], nr = B.first_row[p + 1] - r0;
    if (col >= ncols) return;
    const uint32_t row1 = row0 + MSAB_ROWS_PER_WG < nr ? row0 + MSAB_ROWS_PER_WG : nr;
    const uint8_t *M = M_all + B.m_off[p];
    int32_t c[5] = {0, 0, 0, 0, 0};
#pragma unroll 8
    for (uint32_t r = row0; r < row1; ++r) {
        const int sidx = msa_sym(M[(size_t)r * ncols + col]);
        const int32_t d = degree[r0 + r];
#pragma unroll
        for (int k = 0; k < 5; ++k) c[k] += sidx == k ? d : 0;
    }
    int32_t *counts = counts_all + (size_t)5 * B.col_base[p];
#pragma unroll
    for (int k = 0; k < 5; ++k) if (c[k]) atomicAdd(counts + (size_t)k * ncols + col, c[k]);
}

__global__ __launch_bounds__(256) void k_msab_col_finish(MsaBatch B, const uint32_t *__restrict__ cb_part, const uint32_t *__restrict__ cb_col0,
                                                          const int32_t *__restrict__ counts_all, uint8_t *__restrict__ maj_all,
                                                          uint8_t *__restrict__ flags_all, unsigned long long *__restrict__ class_tot_all)
{
    const uint32_t p = cb_part[blockIdx.x], col = cb_col0[blockIdx.x] + threadIdx.x;
    const uint32_t ncols = B.ncols[p];
    const int32_t *counts = counts_all + (size_t)5 * B.col_base[p];
    long long ci = 0, cd = 0, cs = 0;
    if (col < ncols) {
        int32_t c[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) c[k] = counts[(size_t)k * ncols + col];
        int best = 0, ties = 1;
#pragma unroll
        for (int k = 1; k < 5; ++k) {
            if (c[k] > c[best]) { best = k; ties = 1; }
            else if (c[k] == c[best]) ++ties;
        }
        int32_t tot = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) tot += c[k];
        maj_all[B.col_base[p] + col] = (uint8_t)best;
        flags_all[B.col_base[p] + col] = ties == 1 ? 1 : 0;
        if (ties == 1) {
            if (best == 4) ci = tot - c[4];
            else { cd = c[4]; cs = tot - c[best] - c[4]; }
        }
    }
    __shared__ long long red[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) { ci += __shfl_xor(ci, o, 64); cd += __shfl_xor(cd, o, 64); cs += __shfl_xor(cs, o, 64); }
    if (lane == 0) { red[0][wave] = ci; red[1][wave] = cd; red[2][wave] = cs; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const long long t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (t) atomicAdd(class_tot_all + (size_t)3 * p + threadIdx.x, (unsigned long long)t);
    }
}

// Candidate positions per row the batched kernel keeps in LDS.  With the single-partition kernel's 2 048 a workgroup of four rows took 96 KB:
// ONE workgroup per CU, one wave per SIMD, 4.2 ms for the 50 000 rows of C3 (25-60 candidates each).  1 024: 48 KB, three workgroups per CU;
// a row with more (reads of > 8 kb at ONT error rates) sends its partition through the single-partition entry points, as before.
static constexpr int MSAB_MAX_CAND = 1024;

// one wave per row (the list of correctable positions in LDS; a row with more than MSAB_MAX_CAND gets n_cand = -1: the host corrects that
// row's partition through the single-partition entry points)
__global__ __launch_bounds__(256) void k_msab_row_correct(const uint8_t *__restrict__ M_all, uint8_t *__restrict__ out_all, MsaBatch B, const int32_t *__restrict__ degree,
                                                           const int32_t *__restrict__ counts_all, const uint8_t *__restrict__ maj_all, const uint8_t *__restrict__ flags_all,
                                                           const unsigned long long *__restrict__ class_tot_all, int32_t *__restrict__ n_cand)
{
    __shared__ double s_freq[4][MSAB_MAX_CAND];
    __shared__ uint32_t s_col[4][MSAB_MAX_CAND];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * 4 + wave;
    if (r >= B.n_rows) return;
    const uint32_t p = B.part_of_row[r], ncols = B.ncols[p];
    const size_t cell0 = B.m_off[p] + (size_t)(r - B.first_row[p]) * ncols;
    const uint8_t *row = M_all + cell0;
    uint8_t *orow = out_all + cell0;
    const int32_t *counts = counts_all + (size_t)5 * B.col_base[p];
    const uint8_t *maj = maj_all + B.col_base[p], *flags = flags_all + B.col_base[p];
    const unsigned long long *class_tot = class_tot_all + (size_t)3 * p;
    const char SYM[5] = {'A', 'C', 'G', 'T', '-'};
    const double d_ins = (double)(class_tot[0] > 0 ? class_tot[0] : 1ull);
    const double d_del = (double)(class_tot[1] > 0 ? class_tot[1] : 1ull);
    const double d_sub = (double)(class_tot[2] > 0 ? class_tot[2] : 1ull);
    const bool single = degree[r] == 1;
    double *fq = s_freq[wave];
    uint32_t *cl = s_col[wave];
    uint32_t n = 0;
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
            if (cand && at < (uint32_t)MSAB_MAX_CAND) {
                const int mj = maj[col];
                const double own = (double)counts[(size_t)msa_sym(v) * ncols + col];
                fq[at] = own / (mj == 4 ? d_ins : (v == '-' ? d_del : d_sub));
                cl[at] = col;
            }
            n += (uint32_t)__popcll(mask);
            if (n > (uint32_t)MSAB_MAX_CAND) overflow = true;
        }
    }
    if (overflow) { if (lane == 0) n_cand[r] = -1; return; }
    if (lane == 0) n_cand[r] = (int32_t)n;
    if (n == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t k = (n + 1) / 2;
    double thr = 0.0;
    bool have = false;
    for (uint32_t i0 = 0; i0 < n && !have; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool mine = false;
        double f = 0.0;
        if (i < n) {
            f = fq[i];
            uint32_t lt = 0, le = 0;
            for (uint32_t j = 0; j < n; ++j) { const double g = fq[j]; lt += g < f; le += g <= f; }
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
        if (fq[i] <= thr) { const uint32_t c = cl[i]; orow[c] = (uint8_t)SYM[maj[c]]; }
}

__global__ __launch_bounds__(256) void k_msab_row_lengths(const uint8_t *__restrict__ rows_all, MsaBatch B, uint32_t *__restrict__ len)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * 4 + wave;
    if (r >= B.n_rows) return;
    const uint32_t p = B.part_of_row[r], ncols = B.ncols[p];
    const uint8_t *row = rows_all + B.m_off[p] + (size_t)(r - B.first_row[p]) * ncols;
    uint32_t c = 0;
    for (uint32_t col = lane; col < ncols; col += 64) c += row[col] != '-';
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) len[r] = c;
}

__global__ __launch_bounds__(256) void k_msab_strip(const uint8_t *__restrict__ rows_all, MsaBatch B, const uint64_t *__restrict__ off, uint8_t *__restrict__ packed)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * 4 + wave;
    if (r >= B.n_rows) return;
    const uint32_t p = B.part_of_row[r], ncols = B.ncols[p];
    const uint8_t *row = rows_all + B.m_off[p] + (size_t)(r - B.first_row[p]) * ncols;
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
