// msa_build.hpp -- the multi-alignment matrix of a partition built ON THE DEVICE from the CIGAR ops of its (centre, member) alignments
// and the packed store (SURVEY.md 8(f) rows f1 + f3): what /root/reference/modules/functions.py:543-588 (create_multialignment_matrix),
// :598-631 (position_query_to_alignment) and :679-767 (create_multialignment_format_NEW) assemble from the gapped strings of
// sw_align_sequences.  The gapped strings never exist here: an alignment is its run-length ops (isocon_sg_trace_batch; ~50 per pair
// at 2.5 kb), the bases come from the store's bit-planes.
//
// Layout (the reference's): in front of every centre base t and behind the last one sits a slot of insertion columns -- 1 column, or
// longest + 2 where some member inserts 2 or more characters (functions.py:722-731) -- followed by the base column of t.
//   k_msa_ops_scan     one thread per row: walks the row's ops, atomicMax of the insertion length per slot;
//   k_msa_layout       one workgroup: slot widths and the exclusive prefix sums that give every slot's first column;
//   k_msa_fill         one wave per row: writes the row (the matrix is pre-filled with '-'); single-character insertions go to their
//                      slot's column, the insertions of WIDE slots are only listed (row, slot, position in the member, length, first bases): where
//                      they sit inside the padded longest insertion is decided by get_best_solution (functions.py:635-676, string
//                      heuristics with an alignment tie), which stays on the host and comes back as patches (k_msa_patch).
// ops: len << 4 | code, code 0 '=', 1 'X', 2 'I' (centre base against a gap of the member), 3 'D' (member bases the centre lacks:
// an insertion into the slot in front of the next centre base); the centre is the QUERY of its alignments (isocon_get_candidates.py:47).
#pragma once
#include "common.hpp"

namespace isocon {

__device__ __forceinline__ uint8_t msa_base_char(const DevStore &S, uint32_t id, uint32_t pos)
{
    const size_t w = ((size_t)(pos >> 6) * S.n + id) * 2;
    const uint32_t sh = pos & 63u;
    const uint32_t code = (uint32_t)((S.planes[w] >> sh) & 1ull) | ((uint32_t)((S.planes[w + 1] >> sh) & 1ull) << 1);
    return (uint8_t)("ACGT"[code]);
}

// longest[t] = longest insertion any row has in slot t (t = 0 .. Lm); bad[0] != 0 if a row's ops do not spell the centre / the member
__global__ __launch_bounds__(256) void k_msa_ops_scan(DevStore S, const uint32_t *__restrict__ row_ids, uint32_t n_rows, const uint32_t *__restrict__ ops,
                                                       const unsigned long long *__restrict__ ops_ptr, uint32_t Lm, uint32_t *__restrict__ longest,
                                                       uint32_t *__restrict__ bad)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r == 0 || r >= n_rows) return;          // (row 0 is the centre itself: no ops, no insertions)
    uint32_t t = 0, sp = 0;
    for (unsigned long long k = ops_ptr[r]; k < ops_ptr[r + 1]; ++k) {
        const uint32_t op = ops[k], len = op >> 4, code = op & 15u;
        if (code == 3u) { if (t <= Lm) atomicMax(longest + t, len); sp += len; }
        else { t += len; if (code != 2u) sp += len; }
    }
    if (t != Lm || sp != (uint32_t)S.lens[row_ids[r]]) atomicOr(bad, 1u);
}

// width[t] = 1 or longest + 2; col_slot[t] = first column of slot t (the base column of t follows the slot); totals[0] = columns,
// totals[1] = wide slots.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_msa_layout(const uint32_t *__restrict__ longest, uint32_t Lm, uint32_t *__restrict__ width, uint32_t *__restrict__ col_slot,
                                                      uint32_t *__restrict__ totals)
{
    __shared__ unsigned long long wave_sums[16];
    __shared__ unsigned long long carry;
    __shared__ uint32_t n_wide;
    if (threadIdx.x == 0) { carry = 0; n_wide = 0; }
    __syncthreads();
    for (uint32_t base = 0; base <= Lm; base += 1024u) {
        const uint32_t t = base + threadIdx.x;
        uint32_t w = 0;
        if (t <= Lm) {
            const uint32_t lg = longest[t];
            w = lg > 1u ? lg + 2u : 1u;
            width[t] = w;
            if (lg > 1u) atomicAdd(&n_wide, 1u);
        }
        unsigned long long total = 0;
        const unsigned long long step = t <= Lm ? (unsigned long long)w + (t < Lm ? 1ull : 0ull) : 0ull;      // slot + its base column (the last slot has none)
        const unsigned long long off = block_exscan_1024(step, wave_sums, &total);
        if (t <= Lm) col_slot[t] = (uint32_t)(carry + off);
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = (uint32_t)carry; totals[1] = n_wide; }
}

// One wave per row.  wide[8 i ..] = row, slot, first position in the member, length, the 2-bit codes (A C G T = 0 1 2 3) of its first
// 32 bases (base j at bits 2 j of the 64-bit word lo | hi << 32), two spare words; wide_count: entries appended (may exceed wide_cap:
// the host then calls again with room).
__global__ __launch_bounds__(256) void k_msa_fill(DevStore S, const uint32_t *__restrict__ row_ids, uint32_t n_rows, const uint32_t *__restrict__ ops,
                                                   const unsigned long long *__restrict__ ops_ptr, uint32_t Lm, const uint32_t *__restrict__ longest,
                                                   const uint32_t *__restrict__ width, const uint32_t *__restrict__ col_slot, uint32_t n_cols,
                                                   uint8_t *__restrict__ M, uint32_t *__restrict__ wide, unsigned long long wide_cap,
                                                   unsigned long long *__restrict__ wide_count)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * 4u + (uint32_t)wave;
    if (r >= n_rows) return;
    const uint32_t id = row_ids[r];
    uint8_t *row = M + (size_t)r * n_cols;
    if (r == 0) {          // the centre: its own bases in the base columns
        for (uint32_t t = (uint32_t)lane; t < Lm; t += 64u) row[col_slot[t] + width[t]] = msa_base_char(S, id, t);
        return;
    }
    uint32_t t = 0, sp = 0;
    for (unsigned long long k = ops_ptr[r]; k < ops_ptr[r + 1]; ++k) {
        const uint32_t op = ops[k], len = op >> 4, code = op & 15u;          // (wave-uniform)
        if (t + (code == 3u ? 0u : len) > Lm) return;          // (ops that do not spell the centre were reported by k_msa_ops_scan)
        if (code == 3u) {
            if (longest[t] <= 1u) { if (lane == 0) row[col_slot[t]] = msa_base_char(S, id, sp); }
            else if (lane == 0) {
                const unsigned long long at = atomicAdd(wide_count, 1ull);
                if (at < wide_cap) {
                    unsigned long long codes = 0;
                    for (uint32_t j = 0; j < len && j < 32u; ++j) {
                        const size_t w = ((size_t)((sp + j) >> 6) * S.n + id) * 2;
                        const uint32_t sh = (sp + j) & 63u;
                        codes |= (((S.planes[w] >> sh) & 1ull) | (((S.planes[w + 1] >> sh) & 1ull) << 1)) << (2u * j);
                    }
                    uint32_t *e = wide + 8 * at;
                    e[0] = r; e[1] = t; e[2] = sp; e[3] = len; e[4] = (uint32_t)codes; e[5] = (uint32_t)(codes >> 32); e[6] = 0u; e[7] = 0u;
                }
            }
            sp += len;
        } else {
            for (uint32_t i = (uint32_t)lane; i < len; i += 64u)
                row[col_slot[t + i] + width[t + i]] = code == 2u ? (uint8_t)'-' : msa_base_char(S, id, sp + i);
            t += len;
            if (code != 2u) sp += len;
        }
    }
}

// patches: bytes[ptr[i] .. ptr[i + 1]) go to row patch_row[i] from column patch_col[i] on (one wave per patch)
__global__ __launch_bounds__(256) void k_msa_patch(uint8_t *__restrict__ M, uint32_t n_cols, const uint32_t *__restrict__ patch_row, const uint32_t *__restrict__ patch_col,
                                                    const uint32_t *__restrict__ patch_ptr, const uint8_t *__restrict__ bytes, uint32_t n_patches)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 4u + (uint32_t)wave;
    if (i >= n_patches) return;
    uint8_t *dst = M + (size_t)patch_row[i] * n_cols + patch_col[i];
    const uint32_t b = patch_ptr[i], e = patch_ptr[i + 1];
    for (uint32_t k = b + (uint32_t)lane; k < e; k += 64u) dst[k - b] = bytes[k];
}

}  // namespace isocon
