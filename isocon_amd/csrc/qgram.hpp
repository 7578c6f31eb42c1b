// qgram.hpp -- q-gram count profiles and the lower bound of the edit distance they give; the main pass of the nearest-
// neighbour search asks it before a pair gets a lane (nn.hpp, k_nn_scan_refill).
//
// Lemma (Ukkonen 1992, one-sided form).  G_x(g) = number of occurrences of the q-gram g in x.  One edit operation destroys at
// most q q-gram occurrences of the string it is applied to (the q windows that contain the position; q - 1 for an insertion)
// and creates at most q, hence with S+(x, y) = sum_g max(0, G_x(g) - G_y(g)):  S+(x, y) <= q * ed(x, y) and S+(y, x) <= q * ed(x, y)
// (S+ obeys the triangle inequality).  With L1 = S+(x, y) + S+(y, x) and S+(x, y) - S+(y, x) = |G_x| - |G_y|:
//        ed(x, y) >= ceil( (L1 + | |G_x| - |G_y| |) / (2 q) ).
// Counts saturate at 255 and may be merged into fewer bins (here: hashed): both only shrink S+ (max(0, min(a, c) - min(b, c)) <= max(0, a - b)),
// the identity above holds for the stored vectors, so the bound stays a bound.
//
// q = 8, the 65 536 gram codes hashed into 6144 byte-wide bins (6 KB per sequence).  Measured on C3 (50 k reads of 2.5 kb at 1 %
// errors, thresholds = the final nearest-neighbour distances, median 32; scripts/dev/qgram_filter_estimate.py): same-isoform pairs
// sit at distance ~50, reads of other isoforms inside the +-63 length window are hundreds of edits away.  Share of the pairs the main
// pass would align whose bound exceeds their threshold / whole step in ms (bound kernel + main pass):
//     6-grams, 4096 bins (exact)  89.3 %  61.5 ms (11.5 + 47.7)       8-grams, 4096 bins  91.2 %  56.2 ms (11.5 + 42.4)
//     8-grams, 6144 bins          94.5 %  53.6 ms (17.8 + 33.4)       8-grams, 8192 bins  95.4 %  57.0 ms (24.2 + 30.5)
// (presence bitsets of 4 KB: nothing rejected; 9- and 10-grams: as 8-grams; the bound kernel's cost is proportional to the bins, the
// main pass loses lane balance as its pairs get fewer.)
//
// L1 of byte vectors is v_sad_u8: four bins per lane and instruction with the accumulator as third operand.
#pragma once
#include "common.hpp"

namespace isocon {

static constexpr int QG_Q = 8;
static constexpr int QG_BINS = 6144;      // the 4^q gram codes hashed into this many bins (merging bins keeps the bound a bound)
static constexpr int QG_DWORDS = QG_BINS / 4;
static constexpr int QG_QT = 32;          // entries (rows of the bound matrix) per wave of k_qgram_lb
static constexpr int QG_CHUNK = 16;       // dwords of a profile per step

// prof[i][QG_DWORDS]: byte b of dword e = min(255, occurrences of gram 4e + b in sequence i); gram index = the low code bits of
// its q bases (bits 0..q-1) | the high code bits (bits q..2q-1).  psum[i] = sum of the stored counts.
__global__ __launch_bounds__(256) void k_qgram_profile(DevStore S, uint32_t *__restrict__ prof, uint32_t *__restrict__ psum)
{
    __shared__ uint32_t hist[QG_BINS];
    __shared__ uint32_t s_sum;
    const uint32_t i = blockIdx.x;
    if (i >= S.n) return;
    for (int e = threadIdx.x; e < QG_BINS; e += 256) hist[e] = 0;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const int32_t ngrams = S.lens[i] - QG_Q + 1;
    for (int32_t j = threadIdx.x; j < ngrams; j += 256) {
        const int32_t c = j >> 6, o = j & 63;
        const size_t at = ((size_t)c * S.n + i) * 2;
        uint64_t lo = S.planes[at] >> o, hi = S.planes[at + 1] >> o;
        if (o > 64 - QG_Q) {
            const size_t at2 = ((size_t)(c + 1) * S.n + i) * 2;
            lo |= S.planes[at2] << (64 - o);
            hi |= S.planes[at2 + 1] << (64 - o);
        }
        const uint32_t mask = (1u << QG_Q) - 1u;
        const uint32_t g = ((uint32_t)lo & mask) | (((uint32_t)hi & mask) << QG_Q);
        atomicAdd(&hist[QG_BINS == (1 << (2 * QG_Q)) ? g : ((g * 0x9E3779B1u) >> 7) % (uint32_t)QG_BINS], 1u);
    }
    __syncthreads();
    uint32_t local = 0;
    for (int e = threadIdx.x; e < QG_DWORDS; e += 256) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t c = hist[4 * e + b] < 255u ? hist[4 * e + b] : 255u;
            v |= c << (8 * b);
            local += c;
        }
        prof[(size_t)i * QG_DWORDS + e] = v;
    }
    atomicAdd(&s_sum, local);
    __syncthreads();
    if (threadIdx.x == 0) psum[i] = s_sum;
}

// min over the 64 lanes, in every lane: the DPP row shifts / row broadcasts of gfx9 (VALU only; six ds_bpermute round trips per
// reduction through __shfl_xor otherwise, 32 reductions per tile)
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v0)
{
    const int id = -1;                                          // 0xffffffff: neutral for the unsigned minimum
    auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
    uint32_t v = mn(v0, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v0, 0x111, 0xf, 0xf, false));     // row_shr:1
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v0, 0x112, 0xf, 0xf, false));               // row_shr:2
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v0, 0x113, 0xf, 0xf, false));               // row_shr:3
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x114, 0xf, 0xe, false));                // row_shr:4
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x118, 0xf, 0xc, false));                // row_shr:8
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xa, 0xf, false));                // row_bcast:15
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xc, 0xf, false));                // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Bound matrix in the main pass' own shape: row s belongs to the entry q = q_begin + s * q_stride (launch slot s of
// k_nn_scan_refill), element e to its neighbour p = q + 1 + e, e < row_len[s]; lb[row_off[s] + e] = min(255, bound).
// One wave = QG_QT consecutive rows x 64 neighbours (the lanes): the lane's profile goes through VGPRs QG_CHUNK dwords at a time,
// the rows' profiles are wave-uniform (scalar loads), one accumulator per row.  The four waves of a workgroup take four
// neighbouring lane blocks of the same rows, so the rows' chunks are in the scalar cache for three of them.
//
// rowmin / colmin (optional): the smallest bound of every row / of every neighbour column over the pairs whose roles admit an
// edge (row entry queries the neighbour: qflag[q] && tflag[p]; the neighbour queries the row entry: qflag[p] && tflag[q]), as
// keys (bound << 32 | e) resp. (bound << 32 | q) under atomicMin -- the seeds of the search (ed_lanes.hpp).
__global__ __launch_bounds__(256) void k_qgram_lb(const uint32_t *__restrict__ prof, const uint32_t *__restrict__ psum,
                                                   const unsigned long long *__restrict__ row_off, const uint32_t *__restrict__ row_len,
                                                   uint8_t *__restrict__ lb, uint32_t n, uint32_t q_begin, uint32_t q_stride, uint32_t nq,
                                                   const uint8_t *__restrict__ qflag, const uint8_t *__restrict__ tflag,
                                                   unsigned long long *__restrict__ rowmin, unsigned long long *__restrict__ colmin,
                                                   const uint32_t *__restrict__ chunk_off, uint32_t n_chunks)
{
    const int lane = threadIdx.x & 63;
    // 1-D grid over the workgroups that have work: row blocks in chunks of 8, chunk c owns the ids [chunk_off[c], chunk_off[c + 1]),
    // id - chunk_off[c] = 8 * (group of four lane blocks) + (row block inside the chunk).  The hardware deals consecutive ids to the 8
    // XCDs in turn, so XCD j works through the lane blocks of ONE row block of the chunk (its rows stay in that L2 / scalar cache).
    uint32_t lo = 0, hi = n_chunks;                          // chunk_off[lo] <= id < chunk_off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_off[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    const uint32_t rid = blockIdx.x - chunk_off[lo];
    const uint32_t sb = lo * 8u + (rid & 7u);
    const uint32_t s0 = sb * (uint32_t)QG_QT;
    const uint32_t tb = (rid >> 3) * 4u + (threadIdx.x >> 6);
    if (s0 >= nq) return;
    const uint32_t s_last = s0 + QG_QT - 1 < nq ? s0 + QG_QT - 1 : nq - 1;
    // neighbours of the block: from the first row's first neighbour to the furthest neighbour of any row
    const uint64_t pmin = (uint64_t)q_begin + (uint64_t)s0 * q_stride + 1;
    uint64_t pend;              // exclusive
    {
        const uint32_t sl = s0 + (uint32_t)(lane & (QG_QT - 1));
        const uint32_t s = sl <= s_last ? sl : s_last;
        pend = (uint64_t)q_begin + (uint64_t)s * q_stride + 1 + row_len[s];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t w = ((uint64_t)(uint32_t)__shfl_xor((int)(pend >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)pend, o, 64);
            pend = w > pend ? w : pend;
        }
    }
    const uint64_t p0 = pmin + (uint64_t)tb * 64u;
    if (p0 >= pend) return;                                   // wave-uniform
    const uint64_t p = p0 + (uint64_t)lane;
    const uint32_t pc = p < (uint64_t)n ? (uint32_t)p : n - 1;
    uint32_t acc[QG_QT];
#pragma unroll
    for (int qi = 0; qi < QG_QT; ++qi) acc[qi] = 0;
    // The rows' chunks are scalar loads.  Scalar loads return out of order, the only wait there is is "all of them", and the
    // compiler puts that wait right behind every load.  Hand-placed instead: the chunk of the NEXT row is requested before the
    // 16 v_sad_u8 of the current row and waited for after them (the asm operands only pin that order).
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    static_assert(QG_CHUNK == 16, "one s_load_dwordx16 per row and step");
    // Row addresses are base + row * stride, with no clamp at the last row block (prof is allocated QG_QT * q_stride rows beyond the
    // last sequence; what is read there is never used): a table of 32 clamped addresses cost 64 SGPRs, spilled the row chunks into
    // VGPRs (a v_mov per dword and row) and their addresses into VGPR lanes -- a fifth of the kernel's VALU instructions.
    // The address walks with the loads (one running pointer, kept opaque so that it is not turned back into a table).
    const uint32_t *rp = prof + ((size_t)q_begin + (size_t)s0 * q_stride) * QG_DWORDS;             // row 0 of the block, chunk 0
    const size_t rstride = (size_t)q_stride * QG_DWORDS;
    // rows two at a time, their instructions interleaved: consecutive v_sad_u8 never wait for each other's result
    u32x16 cur0, cur1;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(cur0), "=&s"(cur1) : "s"(rp), "s"(rp + rstride) : "memory");
    // the lane's own profile: 64 contiguous bytes per step (a 16-byte-interleaved layout that makes the wave's loads contiguous
    // was 20 % slower), the next step's bytes requested before this step's arithmetic
    const uint4 *trow = reinterpret_cast<const uint4 *>(prof + (size_t)pc * QG_DWORDS);
    uint4 tn[QG_CHUNK / 4];
#pragma unroll
    for (int j = 0; j < QG_CHUNK / 4; ++j) tn[j] = trow[j];
    for (int c = 0; c < QG_DWORDS / QG_CHUNK; ++c) {
        uint32_t tv[QG_CHUNK];
#pragma unroll
        for (int j = 0; j < QG_CHUNK / 4; ++j) { tv[4 * j] = tn[j].x; tv[4 * j + 1] = tn[j].y; tv[4 * j + 2] = tn[j].z; tv[4 * j + 3] = tn[j].w; }
        if (c + 1 < QG_DWORDS / QG_CHUNK) {
#pragma unroll
            for (int j = 0; j < QG_CHUNK / 4; ++j) tn[j] = trow[(c + 1) * (QG_CHUNK / 4) + j];
        }
#pragma unroll
        for (int qi = 0; qi < QG_QT; qi += 2) {
            // next: the following two rows of this chunk, after the block's last rows the first two rows of the next chunk
            if (qi + 2 < QG_QT) rp += 2 * rstride;
            else rp = rp - (size_t)(QG_QT - 2) * rstride + (c + 1 < QG_DWORDS / QG_CHUNK ? QG_CHUNK : 0);
            asm volatile("" : "+s"(rp));
            u32x16 nxt0, nxt1;
            asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %4, 0x0"
                         : "=&s"(nxt0), "=&s"(nxt1), "+v"(tv[0]) : "s"(rp), "s"(rp + rstride) : "memory");
            uint32_t a = acc[qi], b = acc[qi + 1], a2 = 0, b2 = 0;      // four independent chains
#pragma unroll
            for (int j = 0; j < QG_CHUNK; j += 2) {
                a = __builtin_amdgcn_sad_u8(tv[j], cur0[j], a);
                b = __builtin_amdgcn_sad_u8(tv[j], cur1[j], b);
                a2 = __builtin_amdgcn_sad_u8(tv[j + 1], cur0[j + 1], a2);
                b2 = __builtin_amdgcn_sad_u8(tv[j + 1], cur1[j + 1], b2);
            }
            a += a2; b += b2;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nxt0), "+s"(nxt1), "+v"(a), "+v"(b));
            acc[qi] = a; acc[qi + 1] = b;
            cur0 = nxt0; cur1 = nxt1;
        }
    }
    const bool inside = p < (uint64_t)n;
    const uint32_t sp = psum[pc];
    const bool seeds = rowmin != nullptr;
    const bool p_isq = seeds && inside && qflag[pc] != 0, p_ist = seeds && inside && tflag[pc] != 0;
    unsigned long long cbest = ~0ull;
    // the rows' scalars (length and offset of the row, sum and roles of its entry): lane i fetches row i's, the loop below reads them
    // back with v_readlane (as 32 x 5 scalar loads they were hoisted to the top and spilled)
    uint32_t m_len = 0, m_sum = 0, m_flags = 0, m_off_lo = 0, m_off_hi = 0;
    {
        const uint32_t sl = s0 + (uint32_t)(lane & (QG_QT - 1));
        const uint32_t s = sl <= s_last ? sl : s_last;
        const uint64_t qq = (uint64_t)q_begin + (uint64_t)s * q_stride;
        m_len = row_len[s];
        m_sum = psum[qq];
        const unsigned long long ro = row_off[s];
        m_off_lo = (uint32_t)ro; m_off_hi = (uint32_t)(ro >> 32);
        if (seeds) m_flags = (qflag[qq] != 0 ? 1u : 0u) | (tflag[qq] != 0 ? 2u : 0u);
    }
#pragma unroll
    for (int qi = 0; qi < QG_QT; ++qi) {
        const uint32_t s = s0 + (uint32_t)qi;
        if (s > s_last) break;                                      // wave-uniform
        const uint64_t qq = (uint64_t)q_begin + (uint64_t)s * q_stride;
        const uint64_t e = p - qq - 1;                              // wraps for p <= qq: fails the range test
        const uint32_t r_len = (uint32_t)__builtin_amdgcn_readlane((int)m_len, qi);
        const bool in_row = inside && p > qq && e < (uint64_t)r_len;
        const uint32_t sq = (uint32_t)__builtin_amdgcn_readlane((int)m_sum, qi);
        const uint32_t ds = sq > sp ? sq - sp : sp - sq;
        uint32_t v = (acc[qi] + ds + 2u * QG_Q - 1u) / (2u * QG_Q);
        v = v < 255u ? v : 255u;
        const unsigned long long r_off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)m_off_hi, qi) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane((int)m_off_lo, qi);
        if (in_row) lb[r_off + e] = (uint8_t)v;
        if (seeds) {
            const uint32_t fl = (uint32_t)__builtin_amdgcn_readlane((int)m_flags, qi);
            const bool q_isq = (fl & 1u) != 0, q_ist = (fl & 2u) != 0;
            // row side: smallest (bound, lane) of the wave, then one atomic
            uint32_t key = (in_row && q_isq && p_ist) ? ((v << 6) | (uint32_t)lane) : 0xffffffffu;
            key = wave_min_u32_dpp(key);
            if (key != 0xffffffffu && lane == 0) {
                const uint64_t ee = p0 + (key & 63u) - qq - 1;
                atomicMin(rowmin + s, ((unsigned long long)(key >> 6) << 32) | (unsigned long long)ee);
            }
            if (in_row && p_isq && q_ist) {
                const unsigned long long ck = ((unsigned long long)v << 32) | (unsigned long long)qq;
                cbest = ck < cbest ? ck : cbest;
            }
        }
    }
    if (seeds && cbest != ~0ull) atomicMin(colmin + p, cbest);
}

// Seed pairs from the smallest bounds: entry x with the neighbour of its row minimum (if x owns a row) and with the row entry of
// its column minimum, unless that row's own minimum is this very pair.  pa / pb hold 2 n slots, 0xffffffff = none.
__global__ __launch_bounds__(256) void k_qgram_seed_pairs(const unsigned long long *__restrict__ rowmin, const unsigned long long *__restrict__ colmin,
                                                           uint32_t n, uint32_t q_begin, uint32_t q_stride, uint32_t nq,
                                                           uint32_t *__restrict__ pa, uint32_t *__restrict__ pb)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= n) return;
    uint32_t a0 = 0xffffffffu, b0 = 0xffffffffu, a1 = 0xffffffffu, b1 = 0xffffffffu;
    if (x >= q_begin && (x - q_begin) % q_stride == 0 && (x - q_begin) / q_stride < nq) {
        const unsigned long long kr = rowmin[(x - q_begin) / q_stride];
        if (kr != ~0ull) { a0 = x; b0 = x + 1u + (uint32_t)kr; }
    }
    const unsigned long long kc = colmin[x];
    if (kc != ~0ull) {
        const uint32_t q = (uint32_t)kc;               // a row entry: q = q_begin + s * q_stride by construction
        const unsigned long long kq = rowmin[(q - q_begin) / q_stride];
        if (kq == ~0ull || q + 1u + (uint32_t)kq != x) { a1 = q; b1 = x; }
    }
    pa[2 * (size_t)x] = a0; pb[2 * (size_t)x] = b0;
    pa[2 * (size_t)x + 1] = a1; pb[2 * (size_t)x + 1] = b1;
}

// The same bound for an explicit pair list (one wave per pair; tests and diagnostics: isocon_qgram_bound_pairs).
__global__ __launch_bounds__(256) void k_qgram_lb_pairs(const uint32_t *__restrict__ prof, const uint32_t *__restrict__ psum,
                                                         const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint64_t n_pairs,
                                                         int32_t *__restrict__ out)
{
    const uint64_t pr = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pr >= n_pairs) return;
    const uint32_t x = a[pr], y = b[pr];
    uint32_t acc = 0;
    for (int e = lane; e < QG_DWORDS; e += 64) acc = __builtin_amdgcn_sad_u8(prof[(size_t)x * QG_DWORDS + e], prof[(size_t)y * QG_DWORDS + e], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o, 64);
    if (lane == 0) {
        const uint32_t sx = psum[x], sy = psum[y];
        out[pr] = (int32_t)((acc + (sx > sy ? sx - sy : sy - sx) + 2u * QG_Q - 1u) / (2u * QG_Q));
    }
}

}  // namespace isocon
