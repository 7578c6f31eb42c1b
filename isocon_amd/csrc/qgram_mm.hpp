// qgram_mm.hpp -- q-gram count bounds of the nearest-neighbour search as a tiled contraction on the matrix cores.
//
// Lemma (Ukkonen 1992, one-sided form).  G_x(g) = number of occurrences of the q-gram g in x.  One edit operation destroys at
// most q q-gram occurrences of the string it is applied to and creates at most q, hence with
// S+(x, y) = sum_g max(0, G_x(g) - G_y(g)):  S+(x, y) <= q ed(x, y) and S+(y, x) <= q ed(x, y).  Any map of the count vectors
// that never increases S+ keeps that true.  The maps used here, each applied to both strings alike:
//   * merging bins (the 4^q gram codes are hashed into QG_B0 bins):     max(0, sum a - sum b) <= sum max(0, a - b);
//   * splitting a count into PRESENCE [a > 0] and EXCESS (a - 1)^+:     S+ is unchanged (check the cases a > b = 0, a > b >= 1, a <= b);
//   * merging the excess counts into QG_B1 coarser bins (bin mod QG_B1) and capping them at QG_CAP:  both only shrink S+.
// With A = the stored vector of x, |A| its sum and M(x, y) = sum min(A, B):  S+(x, y) = |A| - M, so
//        ed(x, y) >= ceil( (max(|A|, |B|) - M(x, y)) / q ).
// M is a DOT PRODUCT of thermometer codes: min(a, b) = sum_t [a > t][b > t].  A profile is therefore stored as QM_K = QG_B0 +
// QG_B1 * QG_CAP binary elements (presence bits, then QG_CAP levels of the excess bins), the bound matrix of the main pass is
// (profiles) x (profiles)^T restricted to the length window -- a banded A B^T -- and runs on v_mfma_f32_32x32x64_f8f6f4 (the builtin's
// scale operands are 0: hipcc emits the non-scaled form, same products) with fp4 operands (1.0 = 0x2; sums of at most QM_K ones are exact in f32): 2048 MAC per cycle and SIMD, twice
// the i8 rate, at 4 bits per element (scripts/ubench/mfma_fp4.hip: operand layout checked with asymmetric data, 40 cycles per
// MFMA at the nominal clock).  The v_sad_u8 kernel this replaces moved 308 B per pair and took 17.6 ms at C3.
//
// Parameters (profiles/r03a_qgram_mm_study_mixed.txt, C3, final thresholds; survivors per query / K): 8-grams in 6144 byte bins, what the
// v_sad_u8 kernel used: 396; uniform thermometer 6144 x 3: 456 / 18432; 16384 x 2: 271 / 32768; see DESIGN.md for the mixed designs.
#pragma once
#include "common.hpp"

namespace isocon {

#ifndef ISOCON_QG_Q
#define ISOCON_QG_Q 9
#endif
static constexpr int QG_Q = ISOCON_QG_Q;         // gram length
#ifndef ISOCON_QG_B0                      // (experiments: scripts/dev/build_variant.sh NAME -DISOCON_QG_B0=...)
#define ISOCON_QG_B0 12288
#endif
static constexpr int QG_B0 = ISOCON_QG_B0;       // presence bins (the 4^q gram codes hashed into them).  24 576 until round 6: with the block filter
                                                 // (nn_filter.hpp) behind this bound a survivor costs 0.13 ns instead of 1.1, and the cheaper contraction wins --
                                                 // C3 step 9.04 / 8.55 / 8.46 / 8.33 / 8.43 / 8.67 ms at 20 480 / 16 384 / 14 336 / 12 288 / 10 240 / 8 192 bins
                                                 // (profiles/r06b_b0_sweep.txt)
#ifndef ISOCON_QG_B1
#define ISOCON_QG_B1 2048
#endif
#ifndef ISOCON_QG_CAP
#define ISOCON_QG_CAP 2
#endif
static constexpr int QG_B1 = ISOCON_QG_B1;       // excess bins (presence bin mod QG_B1)
static constexpr int QG_CAP = ISOCON_QG_CAP;     // levels kept of an excess bin
static constexpr int QM_K = QG_B0 + QG_B1 * QG_CAP;     // binary elements per profile
static constexpr int QM_KBE = 128;        // elements per K-block: 64 B per row (4 slots of 16 B = 32 fp4 elements)
static constexpr int QM_ROWB = QM_KBE / 2;
static constexpr int QM_SLOTS = QM_ROWB / 16;
static constexpr int QM_NKB = QM_K / QM_KBE;
static constexpr int QM_TILE = 256;       // rows and columns of the bound matrix per workgroup
static constexpr int QM_STAGES = 4;       // LDS ring: QM_STAGES x 2 operands x 256 rows x 64 B = 128 KB
static constexpr int QM_STAGE_BYTES = 2 * QM_TILE * QM_ROWB;
static constexpr int QM_OUT_STRIDE = QM_TILE + 8;       // bytes per row of the epilogue's byte tile out[q][p] in LDS (66 dwords: the dword writes of a wave hit 64 banks)
static constexpr int QM_OUTT_STRIDE = QM_TILE + 32;     // bytes per row of the transposed byte tile outT[p][q] (72 dwords = 8 mod 64: the quad-transposed dword writes hit 64 banks; rows 16-byte aligned)
static constexpr int QM_EPI_BYTES = QM_TILE * (QM_OUT_STRIDE + QM_OUTT_STRIDE);          // both byte tiles: they take the ring's place after the K loop
static constexpr int QM_META_BYTES = 256 * (4 + 4 + 4 + 4 + 8 + 4) + 3 * 16 * 2 + 256 * (4 + 4 + 8) + 4 * 256 * 4;          // the tile's row and column tables (k_qgram_mm: m_sA .. m_chub)
#ifndef ISOCON_QM_SEEDS
#define ISOCON_QM_SEEDS 4
#endif
static constexpr int QM_SEEDS = ISOCON_QM_SEEDS;             // seed candidates kept per row and per column of the matrix: the smallest bound of every fourth tile
static constexpr uint32_t QM_HUB_BOUND = 36;   // a pair with a bound up to this counts towards its ends' hub scores (nn_list.hpp: which end's table a pair uses)
static constexpr int QM_META_OFF = QM_STAGES * QM_STAGE_BYTES > QM_EPI_BYTES ? QM_STAGES * QM_STAGE_BYTES : QM_EPI_BYTES;          // (16-byte aligned)
static constexpr size_t QM_LDS_BYTES = (size_t)QM_META_OFF + QM_META_BYTES;
static_assert(QG_B0 % 128 == 0 && QG_B1 % 256 == 0 && QG_B0 % QG_B1 == 0, "K-blocks of 128 elements; excess bin = presence bin mod QG_B1");
static_assert(QM_K % QM_KBE == 0 && QM_NKB >= QM_STAGES && QM_STAGES == 4, "whole K-blocks; the K loop's tail is written for a ring of four");
static_assert(QM_META_OFF % 16 == 0 && QM_LDS_BYTES <= 160 * 1024, "the epilogue's byte tiles reuse the ring; the tile's row / column tables sit behind both");

__host__ __device__ __forceinline__ uint32_t qg_bin(uint32_t g)
{
    return (QG_B0 == (1 << (2 * QG_Q))) ? g : ((g * 0x9E3779B1u) >> 7) % (uint32_t)QG_B0;
}

// 8 bits -> 8 fp4 elements (1.0 = 0x2 per set bit)
__device__ __forceinline__ uint32_t qg_spread8(uint32_t x)
{
    uint32_t y = (x | (x << 12)) & 0x000F000Fu;
    y = (y | (y << 6)) & 0x03030303u;
    y = (y | (y << 3)) & 0x11111111u;
    return y << 1;
}

// prof4[kb][row][QM_ROWB]: the 128 elements kb * 128 .. + 127 of every profile, 4 bits each (n_pad rows per K-block);
// psum[i] = sum of the stored vector.  One workgroup per QP_SEQS consecutive sequences: presence bitset + excess counters in LDS, one
// sequence after the other; the rows of a K-block are adjacent in memory, so the workgroup's QP_SEQS rows leave as pieces of
// QP_SEQS x 64 contiguous bytes (one sequence per workgroup wrote 64-byte pieces: half a cache line each, 0.50 ms at C3; four: 0.3x ms).
static constexpr int QP_SEQS = 4;

__global__ __launch_bounds__(256) void k_qgram_profile4(DevStore S, uint8_t *__restrict__ prof4, uint32_t *__restrict__ psum, uint32_t n_pad)
{
    __shared__ uint32_t bits[QP_SEQS][QM_K / 32];            // presence bits, then QG_CAP level bitsets of the excess bins
    __shared__ uint32_t exh[QG_B1];
    __shared__ uint32_t s_sum[QP_SEQS];
    const uint32_t i0 = blockIdx.x * (uint32_t)QP_SEQS;
    if (i0 >= S.n) return;
    const uint32_t cnt = S.n - i0 < (uint32_t)QP_SEQS ? S.n - i0 : (uint32_t)QP_SEQS;
    for (int e = threadIdx.x; e < QP_SEQS * (QM_K / 32); e += 256) (&bits[0][0])[e] = 0;
    if (threadIdx.x < QP_SEQS) s_sum[threadIdx.x] = 0;
    for (uint32_t r = 0; r < cnt; ++r) {
        const uint32_t i = i0 + r;
        for (int e = threadIdx.x; e < QG_B1; e += 256) exh[e] = 0;
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
            const uint32_t bin = qg_bin(((uint32_t)lo & mask) | (((uint32_t)hi & mask) << QG_Q));
            const uint32_t bit = 1u << (bin & 31u);
            const uint32_t old = atomicOr(&bits[r][bin >> 5], bit);
            if (old & bit) atomicAdd(&exh[bin % (uint32_t)QG_B1], 1u);       // every occurrence after the first is excess
        }
        __syncthreads();
        // level bitsets of the excess bins: one ballot per 64 bins and level
        for (int j0 = (threadIdx.x & ~63); j0 < QG_B1; j0 += 256) {
            const uint32_t v = exh[j0 + (threadIdx.x & 63)];
#pragma unroll
            for (int t = 0; t < QG_CAP; ++t) {
                const unsigned long long m = __ballot(v > (uint32_t)t);
                if ((threadIdx.x & 63) == 0) {
                    bits[r][(QG_B0 + t * QG_B1 + j0) / 32] = (uint32_t)m;
                    bits[r][(QG_B0 + t * QG_B1 + j0) / 32 + 1] = (uint32_t)(m >> 32);
                }
            }
        }
        __syncthreads();
    }
    // write: element e = (K-block, sequence, 16-byte slot) in memory order: consecutive threads fill QP_SEQS x 64 contiguous bytes
    uint32_t local[QP_SEQS];
#pragma unroll
    for (int r = 0; r < QP_SEQS; ++r) local[r] = 0;
    for (int e = threadIdx.x; e < QP_SEQS * (QM_K / 32); e += 256) {
        const int kb = e / (QP_SEQS * QM_SLOTS), r = (e / QM_SLOTS) % QP_SEQS, sl = e % QM_SLOTS;
        const uint32_t w = bits[r][kb * QM_SLOTS + sl];
#pragma unroll
        for (int rr = 0; rr < QP_SEQS; ++rr) local[rr] += rr == r ? (uint32_t)__popc(w) : 0u;
        if ((uint32_t)r < cnt) {
            uint4 o;
            o.x = qg_spread8(w & 0xffu); o.y = qg_spread8((w >> 8) & 0xffu); o.z = qg_spread8((w >> 16) & 0xffu); o.w = qg_spread8(w >> 24);
            *reinterpret_cast<uint4 *>(prof4 + ((size_t)kb * n_pad + i0 + (uint32_t)r) * QM_ROWB + (size_t)sl * 16) = o;
        }
    }
#pragma unroll
    for (int r = 0; r < QP_SEQS; ++r) if (local[r]) atomicAdd(&s_sum[r], local[r]);
    __syncthreads();
    if (threadIdx.x < cnt) psum[i0 + threadIdx.x] = s_sum[threadIdx.x];
}

// -DISOCON_QM_TIMELINE (scripts/dev/build_variant.sh; never in the product build): every workgroup of k_qgram_mm records s_memtime at its
// phase boundaries -- start, ring primed, K loop done, E1, E2, E3 -- and the host prints where a tile's time goes (nn_bounds.inc).
#ifdef ISOCON_QM_TIMELINE
__device__ unsigned long long g_qm_timeline[16384 * 8];
#define QM_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384u) g_qm_timeline[blockIdx.x * 8u + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QM_STAMP(k) do { } while (0)
#endif

typedef int qm_v8i __attribute__((ext_vector_type(8)));
typedef float qm_v16f __attribute__((ext_vector_type(16)));

// LDS-DMA of 16 B per lane: LDS destination = lds_dst (wave-uniform byte address) + 16 * lane.  hipcc does not model the
// statement (cdna_hip_programming.md 5.7): the ring below counts its own vmcnt, and the compiler's ds_reads carry no wait for it.
__device__ __forceinline__ void qm_glds16(const uint8_t *gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ---- pieces of the epilogue (scripts/ubench/epilogue_ops.hip checks each of them on the device) ----
typedef unsigned short qm_us2 __attribute__((ext_vector_type(2)));

// dword of lane S of the caller's quad (DPP quad_perm:[S,S,S,S])
template <int S> __device__ __forceinline__ uint32_t qm_quad_bcast(uint32_t w)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, S * 0x55, 0xf, 0xf, true);
}
// minimum / sum over the 16 lanes of a DPP row (row_shr:1, 2, 4, 8): the result is in lane 15 of the row
template <int CTRL> __device__ __forceinline__ uint32_t qm_dpp_min(uint32_t x)
{
    const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, 0xf, 0xf, false);
    return y < x ? y : x;
}
template <int CTRL> __device__ __forceinline__ uint32_t qm_dpp_add(uint32_t x)
{
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t qm_row_min(uint32_t x) { x = qm_dpp_min<0x111>(x); x = qm_dpp_min<0x112>(x); x = qm_dpp_min<0x114>(x); return qm_dpp_min<0x118>(x); }
__device__ __forceinline__ uint32_t qm_row_add(uint32_t x) { x = qm_dpp_add<0x111>(x); x = qm_dpp_add<0x112>(x); x = qm_dpp_add<0x114>(x); return qm_dpp_add<0x118>(x); }

// A chunk of 16 bound bytes (v, byte k = position k of the chunk) -> key = the smallest (bound << 8 | base + k) over the positions k of
// the bit mask `eligible`, hub = the number of positions of `inside` whose bound is <= QM_HUB_BOUND.  Whole masks (the interior of the
// band: almost every chunk; every entry a query and a target) are done four bytes per instruction: "> 36" per byte by a carry-free add, the
// keys as packed 16-bit minima; a chunk without eligible positions (reads against few candidates: most chunks) has no key to find.
__device__ __forceinline__ void qm_chunk_stats(const uint4 v, uint32_t inside, uint32_t eligible, uint32_t base, uint32_t &key, uint32_t &hub)
{
    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
    if (inside == 0xffffu) {
        uint32_t cnt = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t w = wv[d];
            const uint32_t gt = (((w & 0x7f7f7f7fu) + 0x01010101u * (127u - QM_HUB_BOUND)) | w) & 0x80808080u;          // 0x80 per byte > QM_HUB_BOUND
            cnt += (uint32_t)__builtin_popcount(gt ^ 0x80808080u);
        }
        hub += cnt;
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t bd = (wv[k >> 2] >> (8 * (k & 3))) & 0xffu;
            hub += ((inside >> k) & 1u) && bd <= QM_HUB_BOUND ? 1u : 0u;
        }
    }
    if (eligible == 0xffffu) {
        qm_us2 m = {0xffff, 0xffff};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t w = wv[d];
            const uint32_t b4 = base + 4u * (uint32_t)d;
            const uint32_t ke = ((w << 8) & 0xff00ff00u) | (b4 | ((b4 + 2u) << 16)), ko = (w & 0xff00ff00u) | ((b4 + 1u) | ((b4 + 3u) << 16));
            qm_us2 e, o;
            __builtin_memcpy(&e, &ke, 4);
            __builtin_memcpy(&o, &ko, 4);
            m = __builtin_elementwise_min(m, __builtin_elementwise_min(e, o));
        }
        const uint32_t k16 = m.x < m.y ? m.x : m.y;
        key = k16 < key ? k16 : key;
    } else if (eligible != 0u) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t bd = (wv[k >> 2] >> (8 * (k & 3))) & 0xffu;
            const uint32_t cand = (bd << 8) | (base + (uint32_t)k);
            if ((eligible >> k) & 1u) key = cand < key ? cand : key;
        }
    }
}

// One workgroup = one 256 x 256 tile of the bound matrix: columns = the entries p = 256 J .. + 255 (operand A), rows = the launch
// slots s = 256 I .. + 255 of the main pass, i.e. the entries q = Q.entry(s) (operand B).  8 waves, wave (wp, wq) owns
// 128 p x 64 q = 4 x 2 MFMA tiles of 32 x 32 (128 accumulator registers).  The profiles stream through a ring of QM_STAGES K-blocks
// in LDS (LDS-DMA, prefetch distance QM_STAGES - 1, one barrier per K-block); a row of a K-block is 4 slots of 16 B, slot sl of
// tile row r sits at physical slot sl ^ ((r >> 2) & 3), which makes the ds_read_b128 of a fragment (32 rows x one slot, 16 lanes
// per LDS cycle) conflict-free; the swizzle is applied to the global source address, the LDS image stays lane-linear.
// Any k-permutation that is the same for A and B leaves the dot products unchanged, so a lane simply takes 16 consecutive bytes.
//
// Epilogue (28 % of a tile's time before round 6's second half, when a thread took the bytes one at a time and the transposed matrix was
// written a column per thread: profiles/r06m_mm_epilogue.txt): bound = min(255, ceil((max(|A|, |B|) - M) / q)) in float -- v_max, v_sub,
// v_fma, and v_cvt_pk_u8_f32 rounds to nearest, saturates to [0, 255] and packs (exact for every integer |A| - M: ubench/epilogue_ops) --
// into TWO byte tiles in LDS, out[q][p] and, through a 4 x 4 byte transpose over lane quads, outT[p][q]; then, 16 threads per row and 16
// bytes per thread, (E2) 256-byte row pieces into the main pass' layout lb[row_off[s] + (p - q - 1)] -- the host aligns the rows so that
// the address of p is congruent to p mod 16 -- with every row's smallest admissible bound (rowmin) and hub score, (E3) the same from
// outT for the transposed matrix, the columns' smallest admissible bounds (colmin) and hub scores; keys as in k_qgram_seed_pairs.
__global__ __launch_bounds__(512, 2) void k_qgram_mm(const uint8_t *__restrict__ prof4, const uint32_t *__restrict__ psum, uint32_t n, uint32_t n_pad,
                                                      const uint2 *__restrict__ tiles, const unsigned long long *__restrict__ row_off,
                                                      const uint32_t *__restrict__ row_len, uint8_t *__restrict__ lb, QMap Q, uint32_t nq,
                                                      const uint8_t *__restrict__ qflag, const uint8_t *__restrict__ tflag,
                                                      unsigned long long *__restrict__ rowmin, unsigned long long *__restrict__ colmin,
                                                      const unsigned long long *__restrict__ offT, const uint32_t *__restrict__ sloT, const uint32_t *__restrict__ lenT,
                                                      uint8_t *__restrict__ lbT, uint32_t *__restrict__ score)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qm_lds[];
    // tile table entry of this workgroup: ids b, b + 8, b + 16, ... share an XCD (round-robin dispatch), and the 32 of them that
    // are resident together take the 32 tiles of one super-tile (its operands meet in that XCD's L2)
    const uint32_t b = blockIdx.x;
    const uint2 tl = tiles[(((b >> 3) >> 5) * 8u + (b & 7u)) * 32u + ((b >> 3) & 31u)];
    if (tl.x == 0xffffffffu) return;
    const uint32_t I = tl.x, J = tl.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wq = wave >> 1;
    QM_STAMP(0);
    const int r = lane & 31, h = lane >> 5;

    uint8_t *meta = qm_lds + (size_t)QM_META_OFF;
    float *m_sA = reinterpret_cast<float *>(meta);                        // |A| of the 256 columns (as floats: the epilogue's arithmetic)
    float *m_sB = m_sA + 256;                                             // |B| of the 256 rows
    uint32_t *m_q = reinterpret_cast<uint32_t *>(m_sB + 256);             // entry of the row's slot, 0xffffffff = no such slot
    uint32_t *m_len = m_q + 256;                                          // row length
    unsigned long long *m_off = reinterpret_cast<unsigned long long *>(m_len + 256);
    uint32_t *m_flB = reinterpret_cast<uint32_t *>(m_off + 256);          // roles of the row entries: bit 0 query, bit 1 target
    uint16_t *m_tA = reinterpret_cast<uint16_t *>(m_flB + 256);           // per 16 columns: target flags, query flags
    uint16_t *m_qA = m_tA + 16;
    uint16_t *m_tB = m_qA + 16;                                           // per 16 rows: target flags
    uint32_t *m_slo = reinterpret_cast<uint32_t *>(m_tB + 16);            // transposed rows: first slot, number of slots, offset of the first slot
    uint32_t *m_lenT = m_slo + 256;
    unsigned long long *m_offT = reinterpret_cast<unsigned long long *>(m_lenT + 256);
    uint32_t *m_rkey = reinterpret_cast<uint32_t *>(m_offT + 256);        // per row / per column of the tile: smallest admissible (bound << 8 | index), hub count --
    uint32_t *m_rhub = m_rkey + 256;                                      // collected by E2 / E3, sent to rowmin / colmin / score in ONE pass behind them
    uint32_t *m_ckey = m_rhub + 256;
    uint32_t *m_chub = m_ckey + 256;

    // ---- global -> LDS: 2 operands x 256 rows x 4 slots = 2048 chunks of 16 B per K-block, 4 per thread.  The first three K-blocks are
    //      requested before anything else: the tile's row and column tables (dependent loads) arrive while they are under way
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)qm_lds;
    const uint8_t *gsrc[4];
    uint32_t ldst[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it & 1) * 512 + tid;                       // chunk of the operand: row c >> 2, physical slot c & 3
        const int row = c >> 2, ps = c & 3, ls = ps ^ ((row >> 2) & 3);
        uint64_t ent;
        if (it < 2) ent = (uint64_t)J * QM_TILE + row;            // operand A: columns p (n_pad covers every tile)
        else {
            ent = Q.entry(I * QM_TILE + (uint32_t)row);
            if (ent >= n_pad) ent = n_pad - 1;
        }
#ifdef ISOCON_QM_SAMEPANEL          // timing experiment only (wrong bounds): every tile streams the same two operand panels -- what the K loop costs when nothing misses L2
        ent = (uint64_t)row;
#endif
        gsrc[it] = prof4 + ent * QM_ROWB + ls * 16;
        ldst[it] = lds0 + (uint32_t)((it >> 1) * (QM_TILE * QM_ROWB) + ((it & 1) * 512 + wave * 64) * 16);
    }
    const size_t kb_stride = (size_t)n_pad * QM_ROWB;
    auto issue = [&](int kb) {
        const uint32_t st = (uint32_t)(kb % QM_STAGES) * QM_STAGE_BYTES;
#pragma unroll
        for (int it = 0; it < 4; ++it) qm_glds16(gsrc[it] + (size_t)kb * kb_stride, (uint32_t)__builtin_amdgcn_readfirstlane((int)(ldst[it] + st)));
    };
#pragma unroll
    for (int kb = 0; kb < QM_STAGES - 1; ++kb) issue(kb);

    // the tile's tables: waves 0-3 take the columns (entries p), waves 4-7 the rows (slots s); the flag words per 16 columns / rows are wave votes
    if (tid < 256) {
        const uint64_t p = (uint64_t)J * QM_TILE + tid;
        m_sA[tid] = p < n ? (float)psum[p] : 0.f;
        const bool tp = lbT != nullptr && p < n;
        m_slo[tid] = tp ? sloT[p] : 0u;
        m_lenT[tid] = tp ? lenT[p] : 0u;
        m_offT[tid] = tp ? offT[p] : 0ull;
        const bool fl = rowmin != nullptr && p < n;
        const unsigned long long bt = __ballot(fl && tflag[p] != 0), bq = __ballot(fl && qflag[p] != 0);
        if (lane < 4) {
            m_tA[wave * 4 + lane] = (uint16_t)(bt >> (16 * lane));
            m_qA[wave * 4 + lane] = (uint16_t)(bq >> (16 * lane));
        }
    } else {
        const int t = tid - 256;
        const uint32_t s = I * QM_TILE + (uint32_t)t;
        const uint64_t qq = Q.entry(s);
        const bool have = s < nq && qq < n;
        m_q[t] = have ? (uint32_t)qq : 0xffffffffu;
        m_sB[t] = have ? (float)psum[qq] : 0.f;
        m_len[t] = have ? row_len[s] : 0u;
        m_off[t] = have ? row_off[s] : 0ull;
        const bool fl = have && rowmin != nullptr;
        const bool isq = fl && qflag[qq] != 0, ist = fl && tflag[qq] != 0;
        m_flB[t] = (isq ? 1u : 0u) | (ist ? 2u : 0u);
        const unsigned long long bt = __ballot(ist);
        if (lane < 4) m_tB[(wave - 4) * 4 + lane] = (uint16_t)(bt >> (16 * lane));
    }

    qm_v16f acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = qm_v16f{};

    // fragment addresses inside a stage (bytes): row * 64 + physical slot * 16; the K-step's slot pair is (2 ks + h)
    const int swz = (r >> 2) & 3;
    const int offA = (wp * 128 + r) * QM_ROWB, offB = QM_TILE * QM_ROWB + (wq * 64 + r) * QM_ROWB;
    // Software pipeline over the K-blocks.  A K-block is two K-steps (64 elements each) of eight MFMAs; the fragments of K-step ks of
    // K-block kb + 1 are read from LDS right after the MFMAs that consumed the registers they go to (K-step ks of K-block kb), so
    // every fragment read has eight MFMAs to land behind; the "K-block kb + 1 has landed" barrier therefore sits in the MIDDLE of
    // K-block kb.  The four LDS-DMA requests of K-block kb + 3 go between the MFMA groups, one per four MFMAs (issued in a bunch
    // behind the barrier they cost the wave ~100 cycles each in which it issues no MFMA -- and its SIMD partner stands at the same
    // point).  Ring bookkeeping: the requests of K-block kb + 3 overwrite the stage of K-block kb - 1, whose last fragment read
    // lies before the barrier of K-block kb - 1.
    qm_v8i fa[2][4], fb[2][2];
    auto read_frags = [&](int kb, int ks) {
        const uint8_t *stg = qm_lds + (size_t)(kb % QM_STAGES) * QM_STAGE_BYTES;
        const int ps = ((2 * ks + h) ^ swz) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = *reinterpret_cast<const uint4 *>(stg + offA + i * 32 * QM_ROWB + ps);
            fa[ks][i] = qm_v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint4 v = *reinterpret_cast<const uint4 *>(stg + offB + j * 32 * QM_ROWB + ps);
            fb[ks][j] = qm_v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
        }
    };
    auto mfma_group = [&](int g) {          // g = 2 ks + half: four MFMAs
        const int ks = g >> 1;
#pragma unroll
        for (int i = 2 * (g & 1); i < 2 * (g & 1) + 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[ks][i], fb[ks][j], acc[i][j], 4, 4, 0, 0, 0, 0);
    };
    auto dma_piece = [&](int kb3, int g) {
        __builtin_amdgcn_sched_barrier(0);
        qm_glds16(gsrc[g] + (size_t)kb3 * kb_stride, (uint32_t)__builtin_amdgcn_readfirstlane((int)(ldst[g] + (uint32_t)(kb3 % QM_STAGES) * QM_STAGE_BYTES)));
        __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: K-block 0 has landed for everybody, its fragments are requested
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(0, 0);
    read_frags(0, 1);
    QM_STAMP(1);
    // The two waves of a SIMD (w and w + 4) run this loop in lockstep; a wave that is issuing an LDS-DMA request issues no MFMA, so the
    // second half of the workgroup places its requests BEFORE the MFMA groups, the first half behind them: one partner requests while
    // the other feeds the matrix pipe.
    if (wave < 4) {
        for (int kb = 0; kb < QM_NKB - (QM_STAGES - 1); ++kb) {
            mfma_group(0); dma_piece(kb + 3, 0);
            mfma_group(1); dma_piece(kb + 3, 1);
            // K-block kb + 1 has landed: outstanding are its four requests, the four of kb + 2 and the two of kb + 3 issued above
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_frags(kb + 1, 0);
            mfma_group(2); dma_piece(kb + 3, 2);
            mfma_group(3); dma_piece(kb + 3, 3);
            read_frags(kb + 1, 1);
        }
    } else {
        for (int kb = 0; kb < QM_NKB - (QM_STAGES - 1); ++kb) {
            dma_piece(kb + 3, 0); mfma_group(0);
            dma_piece(kb + 3, 1); mfma_group(1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_frags(kb + 1, 0);
            dma_piece(kb + 3, 2); mfma_group(2);
            dma_piece(kb + 3, 3); mfma_group(3);
            read_frags(kb + 1, 1);
        }
    }
    // tail: K-blocks NKB - 3, NKB - 2, NKB - 1 (nothing left to request)
#pragma unroll
    for (int t = 3; t >= 1; --t) {
        const int kb = QM_NKB - t;
        mfma_group(0);
        mfma_group(1);
        if (t > 1) {
            if (t == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_frags(kb + 1, 0);
        }
        mfma_group(2);
        mfma_group(3);
        if (t > 1) read_frags(kb + 1, 1);
    }
    __syncthreads();          // every wave is done with the ring: it becomes the byte tile out[q][p]
    QM_STAMP(2);

    // ---- E1: accumulators -> bound bytes, twice.  C layout: column (B side, q) = lane & 31, row (A side, p) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5):
    //      a lane's four registers 4 g .. 4 g + 3 are one dword of out[q][p .. p + 3]; the four lanes of a quad (q .. q + 3) exchange bytes for outT[p + t][q .. q + 3]
    uint32_t *out32 = reinterpret_cast<uint32_t *>(qm_lds);
    uint32_t *outT32 = reinterpret_cast<uint32_t *>(qm_lds + QM_TILE * QM_OUT_STRIDE);
    {
        const uint32_t t4 = (uint32_t)lane & 3u;
        const uint32_t sel_lo = 0x0c0c0000u | ((4u + t4) << 8) | t4, sel_hi = 0x00000c0cu | ((4u + t4) << 24) | (t4 << 16);
        // (|A| and |B| are non-negative floats: their maximum is the maximum of their bit patterns -- an integer v_max needs no canonicalised inputs)
        const uint32_t *m_sAu = reinterpret_cast<const uint32_t *>(m_sA), *m_sBu = reinterpret_cast<const uint32_t *>(m_sB);
        uint32_t sbf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) sbf[j] = m_sBu[wq * 64 + j * 32 + r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int pl = wp * 128 + i * 32 + 8 * g + 4 * h;
                const uint4 sa = *reinterpret_cast<const uint4 *>(m_sAu + pl);
                const uint32_t sav[4] = {sa.x, sa.y, sa.z, sa.w};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ql = wq * 64 + j * 32 + r;
                    uint32_t w = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // floor((t + q - 1) / q) for t >= 0, 0 below: (t + q - 0.5) / q - 0.5 is at least 0.5 / q away from a rounding boundary for every integer t
                        const float t = __uint_as_float(sav[k] > sbf[j] ? sav[k] : sbf[j]) - acc[i][j][4 * g + k];
                        const float z = __builtin_fmaf(t, 1.0f / (float)QG_Q, (((float)QG_Q - 0.5f) / (float)QG_Q) - 0.5f);
                        w = __builtin_amdgcn_cvt_pk_u8_f32(z, (uint32_t)k, w);
                    }
                    out32[ql * (QM_OUT_STRIDE / 4) + (pl >> 2)] = w;
                    const uint32_t a0 = qm_quad_bcast<0>(w), a1 = qm_quad_bcast<1>(w), a2 = qm_quad_bcast<2>(w), a3 = qm_quad_bcast<3>(w);
                    outT32[(pl + (int)t4) * (QM_OUTT_STRIDE / 4) + ((ql & ~3) >> 2)] = __builtin_amdgcn_perm(a1, a0, sel_lo) | __builtin_amdgcn_perm(a3, a2, sel_hi);
                }
            }
        }
    }
    __syncthreads();
    QM_STAMP(3);

    // ---- E2: rows of the matrix.  16 threads per row (16 columns each), 32 rows per round.  (Offsets inside the tile fit 32 bits: n < 2^30, build_bounds.)
    {
        const int chunk = tid & 15;
        const int32_t p0 = (int32_t)(J * QM_TILE) + chunk * 16;
        const uint32_t tmask = m_tA[chunk];
        const int32_t nn = (int32_t)n - p0;                                    // columns of the chunk below n
        for (int rr = 0; rr < 8; ++rr) {
            const int ql = rr * 32 + (tid >> 4);
            const uint32_t qe = m_q[ql];
            // valid columns of this chunk: q < p <= q + rl, p < n
            int32_t lo = 0, hi = 0;
            if (qe != 0xffffffffu) {
                const int32_t first = (int32_t)qe + 1 - p0, last = first + (int32_t)m_len[ql];        // [first, last)
                lo = first < 0 ? 0 : (first > 16 ? 16 : first);
                hi = last < nn ? last : nn;
                hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
            }
            uint32_t key = 0xffffffffu, hub = 0;
            if (hi > lo) {
                const uint2 v0 = *reinterpret_cast<const uint2 *>(qm_lds + ql * QM_OUT_STRIDE + chunk * 16);
                const uint2 v1 = *reinterpret_cast<const uint2 *>(qm_lds + ql * QM_OUT_STRIDE + chunk * 16 + 8);
                uint4 v; v.x = v0.x; v.y = v0.y; v.z = v1.x; v.w = v1.y;
                // address of column p: row_off + (p - q - 1), congruent to p mod 16 by the host's row alignment
                *reinterpret_cast<uint4 *>(lb + (m_off[ql] + (unsigned long long)(long long)(p0 - (int32_t)qe - 1))) = v;
                const uint32_t inside = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                qm_chunk_stats(v, inside, (m_flB[ql] & 1u) ? (inside & tmask) : 0u, (uint32_t)(chunk * 16), key, hub);
            }
            key = qm_row_min(key);
            hub = qm_row_add(hub);
            if (chunk == 15) { m_rkey[ql] = key; m_rhub[ql] = hub; }
        }
    }
#ifdef ISOCON_QM_TIMELINE
    __syncthreads();
    QM_STAMP(4);
#endif
    // ---- E3: rows of the TRANSPOSED matrix (row p of lbT = the slots whose window holds p, lbT[offT[p] + (slot - sloT[p])], address congruent to the
    //      slot mod 16), from outT in the same shape: 16 threads per row (16 slots each), 32 rows per round; the columns' hub scores and smallest admissible bounds
    const bool columns = lbT != nullptr || colmin != nullptr;
    if (columns) {
        const int chunk = tid & 15;
        const int32_t s0 = (int32_t)(I * QM_TILE) + chunk * 16;
        const uint32_t tmask = m_tB[chunk];
        const uint8_t *outT = qm_lds + QM_TILE * QM_OUT_STRIDE;
        for (int rr = 0; rr < 8; ++rr) {
            const int pl = rr * 32 + (tid >> 4);
            const int32_t slo = (int32_t)m_slo[pl];
            // valid slots of this chunk: slo <= s < slo + lenT
            const int32_t first = slo - s0, last = first + (int32_t)m_lenT[pl];
            const int32_t lo = first < 0 ? 0 : (first > 16 ? 16 : first), hi = last < 0 ? 0 : (last > 16 ? 16 : last);
            uint32_t key = 0xffffffffu, hub = 0;
            if (hi > lo) {
                const uint4 v = *reinterpret_cast<const uint4 *>(outT + pl * QM_OUTT_STRIDE + chunk * 16);
                if (lbT != nullptr) *reinterpret_cast<uint4 *>(lbT + (m_offT[pl] + (unsigned long long)(long long)(s0 - slo))) = v;
                const uint32_t inside = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                const bool p_isq = colmin != nullptr && ((m_qA[pl >> 4] >> (pl & 15)) & 1u) != 0;
                qm_chunk_stats(v, inside, p_isq ? (inside & tmask) : 0u, (uint32_t)(chunk * 16), key, hub);
            }
            key = qm_row_min(key);
            hub = qm_row_add(hub);
            if (chunk == 15) { m_ckey[pl] = key; m_chub[pl] = hub; }
        }
    }
    // ---- E4: one entry of rowmin / colmin / score per row and per column of the tile (the first half of the workgroup the rows, the second the columns)
    __syncthreads();
    if (tid < 256) {
        const uint32_t key = m_rkey[tid], hub = m_rhub[tid], qe = m_q[tid];
        if (rowmin != nullptr && key != 0xffffffffu) {
            const uint64_t e = (uint64_t)J * QM_TILE + (key & 0xffu) - (uint64_t)qe - 1;
            atomicMin(rowmin + ((size_t)I * QM_TILE + (uint32_t)tid) * QM_SEEDS + (J % QM_SEEDS), ((unsigned long long)(key >> 8) << 32) | (unsigned long long)e);
        }
        if (score != nullptr && hub) atomicAdd(score + qe, hub);
    } else if (columns) {
        const int pl = tid - 256;
        const uint32_t key = m_ckey[pl], hub = m_chub[pl];
        const uint64_t p = (uint64_t)J * QM_TILE + (uint64_t)pl;
        if (colmin != nullptr && key != 0xffffffffu)
            atomicMin(colmin + p * QM_SEEDS + (I % QM_SEEDS), ((unsigned long long)(key >> 8) << 32) | (unsigned long long)m_q[key & 0xffu]);
        if (score != nullptr && hub && p < n) atomicAdd(score + p, hub);
    }
#ifdef ISOCON_QM_TIMELINE
    __syncthreads();
    QM_STAMP(5);
    if (threadIdx.x == 0 && blockIdx.x < 16384u) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_qm_timeline[blockIdx.x * 8u + 6] = hw;
    }
#endif
}

// Row layout of the transposed matrix (nn_bounds.inc, build_bounds): for entry p the launch slots whose window holds p -- the slots s with
// q(s) < p <= q(s) + row_len[s], a run because both q(s) and q(s) + row_len[s] ascend with s: sloT[p] = its first slot, lenT[p] = its
// length, padT[p] = the bytes of its storage (from slot sloT & ~15 to the next multiple of 16 behind its last slot).  k_lbt_offsets
// turns the exclusive prefix sums of padT into the address of the first slot.
__global__ __launch_bounds__(256) void k_lbt_rows(QMap Q, uint32_t nq, const uint32_t *__restrict__ row_len, uint32_t n, uint32_t *__restrict__ sloT,
                                                   uint32_t *__restrict__ lenT, uint32_t *__restrict__ padT)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    QMap below = Q;                          // the owned entries below p: the slots 0 .. b - 1
    below.end = Q.end < p ? Q.end : p;
    uint32_t b = below.count();
    if (b > nq) b = nq;
    uint32_t lo = 0, hi = b;                 // first slot whose window reaches p
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (Q.entry(mid) + row_len[mid] < (uint64_t)p) lo = mid + 1; else hi = mid;
    }
    sloT[p] = lo;
    lenT[p] = b - lo;
    padT[p] = b > lo ? ((b + 15u) & ~15u) - (lo & ~15u) : 0u;
}

__global__ __launch_bounds__(256) void k_lbt_offsets(uint32_t n, const uint32_t *__restrict__ sloT, unsigned long long *__restrict__ offT)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p < n) offT[p] += sloT[p] & 15u;
}

// Seed pairs from the smallest bounds: entry x with the neighbours of its row minima (if x owns a row) and with the row entries of
// its column minima (one per class of tiles), unless that row proposes the very pair itself.  The pairs are APPENDED to pa / pb
// (2 QM_SEEDS n slots, pre-filled with 0xffffffff = none; *count = pairs written): a rank of a sharded search owns 1 / N of the rows,
// so most of its row slots would be empty and the one-pair-per-lane kernel would run half-empty waves.
// col_classes (1, 2 or QM_SEEDS): the column minima of the QM_SEEDS tile classes are merged into this many candidates.  One GPU keeps
// all of them; N ranks each see only their own rows of a column, so together they would propose N QM_SEEDS candidates per column --
// N times the seed work of one GPU, constant per rank: with QM_SEEDS / N classes (at least one) the total stays near one GPU's.
__global__ __launch_bounds__(256) void k_qgram_seed_pairs(const unsigned long long *__restrict__ rowmin, const unsigned long long *__restrict__ colmin,
                                                           uint32_t n, QMap Q, uint32_t nq, uint32_t col_classes,
                                                           uint32_t *__restrict__ pa, uint32_t *__restrict__ pb, unsigned long long *__restrict__ count)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t sa[2 * QM_SEEDS], sb[2 * QM_SEEDS];
    uint32_t m = 0;
    if (x < n) {
        uint32_t s = 0u;
        const bool has_row = Q.slot_of(x, s) && s < nq;
        if (has_row) {
#pragma unroll
            for (int c = 0; c < QM_SEEDS; ++c) {
                const unsigned long long kr = rowmin[(size_t)s * QM_SEEDS + c];
                if (kr != ~0ull) { sa[m] = x; sb[m] = x + 1u + (uint32_t)kr; ++m; }
            }
        }
        const uint32_t per = (uint32_t)QM_SEEDS / col_classes;          // tile classes merged into one candidate
        for (uint32_t g = 0; g < col_classes; ++g) {
            unsigned long long kc = ~0ull;
            for (uint32_t c = g; c < g + per * col_classes; c += col_classes) {          // classes g, g + col_classes, ...
                const unsigned long long v = colmin[(size_t)x * QM_SEEDS + c];
                kc = v < kc ? v : kc;
            }
            if (kc != ~0ull) {
                // a row entry: q = Q.entry(its slot) by construction; skipped when it is that row's own candidate for x's class of tiles
                const uint32_t q = (uint32_t)kc;
                uint32_t sq = 0u;
                (void)Q.slot_of(q, sq);
                const unsigned long long kq = rowmin[(size_t)sq * QM_SEEDS + (x / QM_TILE) % QM_SEEDS];
                if (kq == ~0ull || q + 1u + (uint32_t)kq != x) { sa[m] = q; sb[m] = x; ++m; }
            }
        }
    }
    // wave-aggregated append
    uint32_t inc = m;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if ((uint32_t)lane >= d) inc += o;
    }
    const uint32_t wave_total = (uint32_t)__shfl((int)inc, 63, 64);
    if (wave_total == 0) return;
    unsigned long long base = 0;
    if (lane == 63) base = atomicAdd(count, (unsigned long long)wave_total);
    base = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(base >> 32), 63, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 63, 64);
    const unsigned long long at = base + (inc - m);
#pragma unroll
    for (uint32_t i = 0; i < 2 * QM_SEEDS; ++i)
        if (i < m) { pa[at + i] = sa[i]; pb[at + i] = sb[i]; }
}

// The same bound for an explicit pair list (one wave per pair; tests and diagnostics: isocon_qgram_bound_pairs).
__global__ __launch_bounds__(256) void k_qgram_lb_pairs(const uint8_t *__restrict__ prof4, const uint32_t *__restrict__ psum, uint32_t n_pad,
                                                         const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint64_t n_pairs,
                                                         int32_t *__restrict__ out)
{
    const uint64_t pr = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pr >= n_pairs) return;
    const uint32_t x = a[pr], y = b[pr];
    uint32_t m = 0;
    for (int c = lane; c < QM_K / 8; c += 64) {            // dwords of a profile: K-block c / 16, dword c % 16 of its row
        const size_t at = ((size_t)(c / (QM_ROWB / 4)) * n_pad) * QM_ROWB + (size_t)(c % (QM_ROWB / 4)) * 4;
        const uint32_t u = *reinterpret_cast<const uint32_t *>(prof4 + at + (size_t)x * QM_ROWB);
        const uint32_t v = *reinterpret_cast<const uint32_t *>(prof4 + at + (size_t)y * QM_ROWB);
        m += (uint32_t)__popc(u & v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += (uint32_t)__shfl_xor((int)m, o, 64);
    if (lane == 0) {
        const uint32_t sx = psum[x], sy = psum[y];
        out[pr] = (int32_t)(((sx > sy ? sx : sy) - m + (uint32_t)QG_Q - 1u) / (uint32_t)QG_Q);
    }
}

}  // namespace isocon
