// sg.hpp -- semi-global affine-gap alignment with traceback (kernels).
//
// Replaces parasail.sg_trace_scan_16/32 + CIGAR decode as used by parasail_alignment
// (/root/reference/modules/SW_alignment_module.py:64-86); semantics restated in oracle/isocon_oracle.c
// (orc_sg_trace): Gotoh recurrences, free end gaps on both sequences, a gap of length g costs open+(g-1)*ext.
//
// Two forward kernels.  Pairs with a certified band of at most 256 diagonals (the read pairs of the pipeline: their edit distances come
// down as hints) run in k_sg_band, the band's diagonals on the lanes -- see there; everything else in k_sg_forward:
// one wavefront per pair, systolic over text columns.  Lane l owns query rows [l*R, (l+1)*R) and
// works on column s-l at step s; the bottom-row (H, F) of a strip and the text base travel one lane down per step.
// Each cell emits a 4-bit trace code; a lane's R codes of one column are stored as R/2 contiguous bytes at
// ((s*64 + lane) * R/2) -- i.e. indexed by STEP, not by column, so that all 64 lanes of a step write one contiguous
// 32*R-byte block (coalesced HBM writes; this kernel is HBM-write bound: ~ m*n/2 bytes per pair).
// Walk kernels (k_sg_walk for the strips, k_sg_walk_band for the band kernel's tiled trace): one thread per pair follows the codes back
// from the end cell and emits run-length CIGAR ops.
#pragma once
#include <type_traits>
#include "common.hpp"

namespace isocon {

// trace nibble = four raw decision bits, each the SIGN BIT of a score difference (no compare / select instructions:
// on gfx950 `v_cmp` + `v_cndmask ... vcc` pairs cost an order of magnitude more than the arithmetic, scripts/ubench):
//   bit 3  F opened  (F[i][j] = H[i-1][j] - open  rather than F[i-1][j] - ext)
//   bit 2  E opened  (E[i][j] = H[i][j-1] - open  rather than E[i][j-1] - ext)
//   bit 1  H came from a gap (else from the diagonal)
//   bit 0  gap: 1 = E (horizontal, consumes a ref base, 'D'), 0 = F (vertical, consumes a query base, 'I');
//          diagonal: 1 = mismatch, 0 = match
// The tie policies only shift the differences by 0 or 1 (wave-uniform constants), so one kernel serves all of them.
// A lane's 8 nibbles of one column form one dword, row r in bits [31-4r, 28-4r].
static constexpr uint32_t SG_BIT_FOPEN = 8, SG_BIT_EOPEN = 4, SG_BIT_GAP = 2, SG_BIT_X = 1;
static constexpr int32_t SG_NEG = -(1 << 28);

// tie policy bits, identical to oracle/isocon_oracle.c
static constexpr int SG_POL_E_BEFORE_F = 1, SG_POL_OPEN_ON_TIE = 2, SG_POL_COL_FIRST = 4, SG_POL_LAST_MAX = 8, SG_POL_GAP_FIRST = 16;

struct SgParams {
    int32_t match, open, ext, policy;
};

struct SgPair {            // per pair of the batch
    uint32_t a, b;         // query (rows), ref (columns)
    int32_t mismatch;
    int32_t dlo;           // band of diagonals j - i that is computed: [dlo, dhi]  (full matrix: dlo <= -m, dhi >= n)
    uint64_t trace_off;    // byte offset of this pair's trace in the scratch buffer
    uint64_t ops_off;      // first slot of this pair's ops region (capacity m + n + 2)
    uint64_t bound_off;    // first int2 of this pair's pass-boundary row (capacity n)
    int32_t dhi;
    int32_t steps;         // steps per pass (window columns + 63), the same for every pass of the pair
    int32_t mode;          // 0: strips of 512 query rows (k_sg_forward); 1 / 2: the band's diagonals on the lanes, 4 / 2 per lane (k_sg_band<.., 4 / 2>)
    int32_t retry_x;       // > 0: the pair runs in a band NARROWER than its bound asks for (a cheaper class), and this is the half-width X the
                           // bound asks for: k_sg_band runs the pair again in that band if the narrow result does not certify itself
};

static constexpr int SG_BAND_DIAGS = 256;      // k_sg_band<.., 4>: 64 lanes x 4 diagonals
static constexpr int SG_TILE_DWORDS = 16, SG_TILE_LOG = 4;      // k_sg_band's trace: tiles of 16 dwords per lane (one 64-byte line)
static constexpr int SG_BAND_DIAGS_NARROW = 128;      // k_sg_band<.., 2>: 64 lanes x 2 diagonals (half the cells per step, half the trace)

// Column window of one pass (64*R query rows starting at prow0) for the band [dlo, dhi]: every cell of the band lies
// inside, the window starts on a multiple of 64 (text chunks) at least one column left of the band.  Cells of the
// window outside the band are simply computed too; what lies outside the window counts as -infinity.
__host__ __device__ __forceinline__ void sg_window(int32_t prow0, int32_t rows, int32_t dlo, int32_t dhi, int32_t n, int32_t &jlo, int32_t &jhi)
{
    const int64_t lo = (int64_t)prow0 + dlo - 1;
    jlo = lo <= 0 ? 0 : (int32_t)(lo & ~(int64_t)63);
    const int64_t hi = (int64_t)prow0 + rows - 1 + dhi + 1;
    jhi = hi > (int64_t)n - 1 ? n - 1 : (int32_t)hi;
}

// 64 consecutive bits of a bit-plane starting at bit `off` (per-lane address).
__device__ __forceinline__ uint64_t plane_bits64(const uint64_t *planes, uint32_t nseq, int32_t nchunks, uint32_t id, int plane, int32_t off)
{
    auto chunk = [&](int32_t ci) -> uint64_t { return (ci >= 0 && ci < nchunks) ? planes[((size_t)ci * nseq + id) * 2 + plane] : 0; };
    const int32_t ci = off >> 6, sh = off & 63;
    const uint64_t c0 = chunk(ci);
    return sh ? ((c0 >> sh) | (chunk(ci + 1) << (64 - sh))) : c0;
}

// Forward pass.  The query is cut into passes of 64*R rows; inside a pass lane l owns rows [l*R, (l+1)*R) of the
// pass.  The bottom row (H, F) of a pass is parked in `bound` (one int2 per column) and read back by lane 0 of the
// next pass.  Trace layout: code of cell (i, j) lives at  ((pass*steps + j + l) * 64 + l) * R/2 + r/2,
// pass = i / (64R), l = (i % (64R)) / R, r = i % R, steps = n + 63.
// POL0: the three cell-level tie rules are parasail's (policy bits 0, 1, 4 clear); EXT0: gap extension costs 0
// (the reference's alignment call, SW_alignment_module.py:64).  Both are compile-time so the common case drops the
// corresponding subtractions.
// (register budget: four waves per SIMD for the reference's own call -- policy 0, extension 0: 119 registers --, three for the general form,
// whose seven per-cell constants do not fit 128 without scratch)
template <int R, bool POL0, bool EXT0>
__global__ __launch_bounds__(64, (POL0 && EXT0) ? 4 : 3) void k_sg_forward(DevStore S, const SgPair *__restrict__ pairs, SgParams prm,
                                                    uint8_t *__restrict__ trace, int2 *__restrict__ bound_all,
                                                    int32_t *__restrict__ endinfo)
{
    static_assert(R == 8, "strip height (one trace dword per lane and column)");
    const uint32_t pidx = blockIdx.x;
    const int lane = threadIdx.x;
    const SgPair pr = pairs[pidx];
    if (uniform_i32(pr.mode) != 0) return;          // the pair belongs to k_sg_band
    const uint32_t ia = (uint32_t)uniform_i32((int32_t)pr.a), ib = (uint32_t)uniform_i32((int32_t)pr.b);
    const int32_t m = uniform_i32(S.lens[ia]), n = uniform_i32(S.lens[ib]);
    const int32_t match = prm.match, mism = uniform_i32(pr.mismatch), open = prm.open, ext = prm.ext;
    const int32_t policy = prm.policy;
    // tie policies as 0/1 offsets on the differences whose sign bits become the trace (see the nibble comment)
    const int32_t c_open = (policy & SG_POL_OPEN_ON_TIE) ? 1 : 0;   // opened iff open-value >  ext-value (default) / >= (tie -> open)
    const int32_t c_gap = (policy & SG_POL_GAP_FIRST) ? 1 : 0;      // gap iff diag <  best gap (default) / <=
    const int32_t c_ef = (policy & SG_POL_E_BEFORE_F) ? 1 : 0;      // E iff F < E (default) / F <= E
    // The per-cell constants live in VGPRs on purpose: a VALU instruction with an SGPR operand issues in ~4.2 cycles
    // on gfx950, the same instruction on VGPR operands in ~2.3 (scripts/ubench/valu_rate2.hip).
    int32_t open_v, ext_v, mism_v, delta_v, c_open_v, c_gap_v, c_ef_v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(open_v) : "s"(open));
    asm volatile("v_mov_b32 %0, %1" : "=v"(ext_v) : "s"(ext));
    asm volatile("v_mov_b32 %0, %1" : "=v"(mism_v) : "s"(mism));
    asm volatile("v_mov_b32 %0, %1" : "=v"(delta_v) : "s"(match - mism));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c_open_v) : "s"(c_open));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c_gap_v) : "s"(c_gap));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c_ef_v) : "s"(c_ef));
    const uint64_t *planes = S.planes;
    const uint32_t nseq = S.n;
    const int32_t nchunks = (int32_t)S.nchunks;
    // (wave-uniform offsets through the scalar unit: as vector values the two pointers were spilled around every pass of the <.., false, false> form)
    auto uniform_u64 = [](uint64_t v) { return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)v); };
    int2 *bound = bound_all + uniform_u64(pr.bound_off);
    uint8_t *tbase = trace + uniform_u64(pr.trace_off);
    const int32_t steps = uniform_i32(pr.steps);
    const int32_t dlo = uniform_i32(pr.dlo), dhi = uniform_i32(pr.dhi);
    const int32_t passes = (m + 64 * R - 1) / (64 * R);
    int32_t jhi_prev = -1;          // last column the previous pass left a boundary row for

    const int32_t last_i = m - 1;
    const int32_t pstar = last_i / (64 * R), lstar = (last_i % (64 * R)) / R, rstar = last_i % R;
    int32_t rowbest = SG_NEG, rowj_first = -1, rowj_last = -1;      // last query row, over columns
    int32_t colbest = SG_NEG, coli_first = -1, coli_last = -1;      // last ref column, over this lane's rows

#pragma unroll 1
    for (int32_t pass = 0; pass < passes; ++pass) {
        const int32_t row0 = pass * 64 * R + lane * R;
        int32_t jlo, jhi;
        sg_window(pass * 64 * R, 64 * R, dlo, dhi, n, jlo, jhi);
        const int32_t psteps = jhi >= jlo ? jhi - jlo + 1 + 63 : 0;
        const int32_t left0 = jlo == 0 ? 0 : SG_NEG;       // H[.][jlo - 1]: the free leading gap, or outside the window
        const uint64_t qlo = plane_bits64(planes, nseq, nchunks, ia, 0, row0);
        const uint64_t qhi = plane_bits64(planes, nseq, nchunks, ia, 1, row0);
        int32_t Hp[R], Ep[R];   // H[i][j-1], E[i][j-1]
#pragma unroll
        for (int r = 0; r < R; ++r) { Hp[r] = left0; Ep[r] = SG_NEG; }
        int32_t diag_in = left0;     // H[row0-1][j-1]
        int32_t sendH = 0, sendF = SG_NEG, sendC = 0;
        uint64_t tlo = 0, thi = 0;
        int32_t bndH = 0, bndF = SG_NEG;   // lane l: boundary (H, F) of column 64*(s/64) + l
        const bool lane_has_rows = row0 < m;
        // One step of the pass.  INTERIOR (compile-time): every lane has rows and sits on a column strictly inside the window and left of
        // the last ref column, and the pass is not the one with the last query row -- no lane predicate, no boundary-column select, no
        // end-cell candidates; the loop over those steps (all but the first and last ~64 of a full pass) is a separate one, so that
        // the general path's branches and their joins are not paid there.
        auto step = [&](const int32_t s, auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
            const int32_t col0 = jlo + s;     // lane 0's column (jlo is a multiple of 64)
            if ((s & 63) == 0) {   // wave-uniform: next 64 ref bases
                const int32_t tc = col0 >> 6;
                uint64_t a = 0, b = 0;
                if (tc < nchunks) { a = planes[((size_t)tc * nseq + ib) * 2]; b = planes[((size_t)tc * nseq + ib) * 2 + 1]; }
                // into SGPRs right here: a vector load whose result is first read many steps later would otherwise
                // put an `s_waitcnt vmcnt(0)` (which also waits for the trace stores in flight) into every step
                tlo = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)a);
                thi = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)b);
            }
            if (pass > 0 && (s & 63) == 0) {
                // bottom row of the previous pass for the next 64 columns: one coalesced, L1-bypassing load per lane
                // (the values were written by lane 63 of this very wave); lane 0 picks its column by readlane below
                const int32_t col = col0 + lane;
                if (col <= jhi_prev) {           // (columns right of the previous pass' window: outside the band)
                    const int32_t *bp = reinterpret_cast<const int32_t *>(bound + col);
                    bndH = __hip_atomic_load(bp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bndF = __hip_atomic_load(bp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else { bndH = SG_NEG; bndF = SG_NEG; }
                // consume the loads HERE: otherwise the compiler parks an `s_waitcnt vmcnt(0)` in front of the
                // readlanes of EVERY step, which also waits for the trace stores in flight (vmcnt counts stores) and
                // serialises each step on the HBM write latency.
                asm volatile("" : "+v"(bndH), "+v"(bndF));
            }
            // one lane down: DPP wave shifts (lane 0's values are replaced below); as __shfl_up these were ds_bpermute + address
            // instructions and an LDS round trip in every step's dependency chain
            int32_t Hup = __builtin_amdgcn_update_dpp(0, sendH, 0x138, 0xf, 0xf, false);
            int32_t Fup = __builtin_amdgcn_update_dpp(0, sendF, 0x138, 0xf, 0xf, false);
            int32_t ch = __builtin_amdgcn_update_dpp(0, sendC, 0x138, 0xf, 0xf, false);
            const int32_t b_h = __builtin_amdgcn_readlane(bndH, s & 63), b_f = __builtin_amdgcn_readlane(bndF, s & 63);
            if (lane == 0) {
                ch = (int32_t)(((tlo >> (s & 63)) & 1) | (((thi >> (s & 63)) & 1) << 1));
                if (pass == 0) { Hup = 0; Fup = SG_NEG; }
                else { Hup = b_h; Fup = b_f; }
            }
            const int32_t j = col0 - lane;
            const bool act = INTERIOR || (j >= jlo && j <= jhi && lane_has_rows);
            const int32_t diag_next = Hup;
            if (act) {
                const uint64_t slo = (ch & 1) ? ~(uint64_t)0 : 0, shi = (ch & 2) ? ~(uint64_t)0 : 0;
                const uint32_t eq = (uint32_t)(~(qlo ^ slo) & ~(qhi ^ shi));
                int32_t diag = (!INTERIOR && j == jlo) ? left0 : diag_in;   // left boundary column: H[.][-1] = 0 / outside the window
                uint32_t tw = 0;
                int32_t hv_last = SG_NEG;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int32_t Hl = Hp[r], El = Ep[r];
                    const int32_t Fopn = Hup - open_v, Fext = EXT0 ? Fup : Fup - ext_v;
                    const int32_t Eopn = Hl - open_v, Eext = EXT0 ? El : El - ext_v;
                    const int32_t Fv = Fopn > Fext ? Fopn : Fext;
                    const int32_t Ev = Eopn > Eext ? Eopn : Eext;
                    const int32_t e = __builtin_amdgcn_sbfe((int32_t)eq, r, 1);        // -1 where the bases are equal
                    const int32_t Hd = diag + mism_v + (e & delta_v);
                    int32_t Hv, t_f, t_e, t_g, t_x;                                    // decisions = sign bits
                    if (POL0) {
                        const int32_t g = Fv > Ev ? Fv : Ev;
                        Hv = Hd > g ? Hd : g;                   // (one v_max3)
                        t_f = Fext - Fopn;                      // < 0 : F opened
                        t_e = Eext - Eopn;                      // < 0 : E opened
                        t_g = Hd - Hv;                          // < 0 : H came from a gap
                        t_x = Fv - Ev;                          // < 0 : E rather than F
                    } else {
                        const int32_t g = Fv > Ev ? Fv : Ev;
                        Hv = Hd > g ? Hd : g;
                        t_f = Fext - Fopn - c_open_v;
                        t_e = Eext - Eopn - c_open_v;
                        t_g = Hd - g - c_gap_v;
                        t_x = Fv - Ev - c_ef_v;
                    }
                    // bit 0: gap ? t_x : mismatch (= ~e)   ->  (a & b) | (~a & ~c), truth table 0xC5
                    const uint32_t x = __builtin_amdgcn_bitop3_b32((uint32_t)t_g, (uint32_t)t_x, (uint32_t)e, 0xC5);
                    tw = __builtin_amdgcn_alignbit(tw, (uint32_t)t_f, 31);
                    tw = __builtin_amdgcn_alignbit(tw, (uint32_t)t_e, 31);
                    tw = __builtin_amdgcn_alignbit(tw, (uint32_t)t_g, 31);
                    tw = __builtin_amdgcn_alignbit(tw, x, 31);
                    diag = Hl;
                    Hp[r] = Hv;
                    Ep[r] = Ev;
                    Hup = Hv;
                    Fup = Fv;
                    // the cell of the last query row, picked from the VALUE just computed: a select chain over Hp[r] is turned into ONE load
                    // at a dynamic index by the compiler (select of loads -> load of select), which keeps Hp[] out of the registers -- it
                    // then lived in LDS and every step of every pass wrote it there (2 x ds_write_b128 per step: 1.25e8 LDS instructions and
                    // 8.1e8 bank-conflict cycles per 4 096 full-matrix pairs, profiles/r05sz_pmc_summary.txt)
                    if (!INTERIOR) hv_last = (r == rstar) ? Hv : hv_last;
                }
                // end-cell candidates.  Both guards are WAVE-UNIFORM on purpose (scalar branches): left to itself the
                // compiler if-converts these blocks into ~180 predicated instructions that run on every step.
                if (!INTERIOR && pass == pstar) {                      // last query row: lane lstar, row rstar of this pass
                    const int32_t hv = hv_last;
                    if (lane == lstar) {
                        if (hv > rowbest) { rowbest = hv; rowj_first = j; rowj_last = j; }
                        else if (hv == rowbest) rowj_last = j;
                    }
                }
                if (!INTERIOR && col0 >= n - 1) {                      // only then can some lane sit on the last ref column
                    if (j == n - 1) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            if (row0 + r < m) {
                                if (Hp[r] > colbest) { colbest = Hp[r]; coli_first = row0 + r; coli_last = row0 + r; }
                                else if (Hp[r] == colbest) coli_last = row0 + r;
                            }
                        }
                    }
                }
                // one contiguous 32*R-byte block per wave and step
                uint32_t *dst = reinterpret_cast<uint32_t *>(tbase + (((size_t)pass * steps + s) * 64 + lane) * (R / 2));
                dst[0] = tw;
                if (lane == 63 && pass + 1 < passes) { int2 v; v.x = Hup; v.y = Fup; bound[j] = v; }
            }
            diag_in = diag_next;
            sendH = Hup;
            sendF = Fup < SG_NEG ? SG_NEG : Fup;          // (-inf must not drift towards the end of the int32 range over a long pass)
            sendC = ch;
        };
        // interior steps: s in [s_in, s_out): lane 63 inside the window (s - 63 > 0 <=> its column > jlo), lane 0 at most on jhi and left
        // of the last ref column
        int32_t s_in = psteps, s_out = psteps;
        if (pass != pstar && (pass + 1) * 64 * R <= m) {
            s_in = 64;
            s_out = jhi - jlo + 1;                                  // col0 <= jhi
            if (s_out > n - 1 - jlo) s_out = n - 1 - jlo;           // col0 < n - 1
            if (s_out > psteps) s_out = psteps;
            if (s_out < s_in) s_in = s_out = psteps;
        }
        int32_t s = 0;
#pragma unroll 1
        for (; s < psteps && s < s_in; ++s) step(s, std::false_type());
        // (two steps per iteration: the rows' H / E registers alternate between two sets instead of being moved back at the loop edge --
        // 42 v_mov per step in the one-step loop)
#pragma unroll 1
        for (; s + 1 < s_out; s += 2) {
            step(s, std::true_type());
            step(s + 1, std::true_type());
        }
#pragma unroll 1
        for (; s < psteps; ++s) step(s, std::false_type());
        jhi_prev = jhi;
        if (pass + 1 < passes) __threadfence();
    }
    // reduce the last-column candidates over lanes: maximum; smallest (first) / largest (last) row on ties
    int32_t cb = colbest, cf = coli_first, cl = coli_last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t ob = __shfl_xor(cb, o, 64), of = __shfl_xor(cf, o, 64), ol = __shfl_xor(cl, o, 64);
        if (ob > cb) { cb = ob; cf = of; cl = ol; }
        else if (ob == cb) {
            if (of >= 0 && (cf < 0 || of < cf)) cf = of;
            if (ol > cl) cl = ol;
        }
    }
    const int32_t rb = __shfl(rowbest, lstar, 64), rf = __shfl(rowj_first, lstar, 64), rl = __shfl(rowj_last, lstar, 64);
    if (lane == 0) {
        const bool last = (policy & SG_POL_LAST_MAX) != 0, col_first = (policy & SG_POL_COL_FIRST) != 0;
        int32_t score, eq, er;
        // oracle: scan A then scan B, update on v > score (or v >= score when `last`)
        const int32_t rj = last ? rl : rf, ci = last ? cl : cf;
        if (!col_first) {
            score = rb; eq = m - 1; er = rj;
            if (cb > score || (last && cb == score)) { score = cb; eq = ci; er = n - 1; }
        } else {
            score = cb; eq = ci; er = n - 1;
            if (rb > score || (last && rb == score)) { score = rb; eq = m - 1; er = rj; }
        }
        endinfo[(size_t)pidx * 4 + 0] = score;
        endinfo[(size_t)pidx * 4 + 1] = eq;
        endinfo[(size_t)pidx * 4 + 2] = er;
    }
}

// Forward pass for pairs with a certified band of at most 256 diagonals (SW_alignment_module hands the pairs' edit distances
// down: ed = 25 at 2.5 kb means +-77 diagonals).  The strip kernel above computes such a band as 512-row strips of
// 512 + band + 64 columns -- 5x the cells of the band -- because a diagonal band offers only `band` cells of parallelism at
// any time.  Here the band's DIAGONALS sit on the lanes (lane l: diagonals dlo + 4 l .. + 3) and the wavefront sweeps the
// anti-diagonals a = i + j: a diagonal d has a cell on a iff a = d (mod 2), so every step updates two of a lane's four cells.
// Predecessors: diagonal (i-1, j-1) = the same slot two steps ago; up (i-1, j) = slot + 1 and left (i, j-1) = slot - 1, both
// one step ago -- in the lane, or one DPP wave shift away (slot 0 <- lane - 1's slot 3, slot 3 <- lane + 1's slot 0).  A
// slot that has not reached the matrix yet holds H = 0, E = F = -inf, which IS the free boundary row / column its
// neighbours read; beyond the two outermost diagonals lies -inf (0 where that cell is the boundary row / column).
// Match bits: per slot a 64-cell mask ~(q ^ t) of its diagonal, refilled every 128 steps (per-lane unaligned plane fetches).
// Trace: the two nibbles of a step form a byte, four steps a dword (two diagonals per lane: one nibble per step, eight steps a dword):
// (m + n) * 64 bytes per pair instead of ~ m * (512 + band) / 2.  Dword q = a >> 2 (>> 3) of lane l lies at ((q >> 4) * 64 + l) * 16 + (q & 15):
// tiles of 16 dwords per lane, so that the 64 (128) steps a path spends on a lane's diagonals are ONE 64-byte line for the walk
// (step-major -- dword q of all lanes in one 256-byte line -- the walk fetched a line per dword: 4-5 GB at C3, the whole of its time).
// The wave stages a tile in LDS (dword q of all lanes: one conflict-free ds_write) and writes it out as 64 bytes per lane.
// (the body of k_sg_band: one pair, the band [dlo, dlo + 64 DPL); returns the score of the end cell, wave-uniform)
template <bool POL0, bool EXT0, int DPL>
__device__ __forceinline__ int32_t sg_band_run(const DevStore &S, const SgPair &pr, const int32_t dlo, const SgParams &prm, uint8_t *__restrict__ trace,
                                               int32_t *__restrict__ endinfo, const uint32_t pidx, uint32_t *__restrict__ stage)
{
    const int lane = threadIdx.x;
    static_assert(DPL == 4 || DPL == 2, "diagonals per lane: 4 (bands of up to 256 diagonals) or 2 (up to 128: half the cells and half the trace)");
    constexpr int DIAGS = 64 * DPL;                 // diagonals of the band
    constexpr int SPW = 32 / (2 * DPL);             // steps per trace dword: DPL / 2 nibbles per step
    constexpr int SPW_LOG = DPL == 4 ? 2 : 3;
    const uint32_t ia = (uint32_t)uniform_i32((int32_t)pr.a), ib = (uint32_t)uniform_i32((int32_t)pr.b);
    const int32_t m = uniform_i32(S.lens[ia]), n = uniform_i32(S.lens[ib]);
    const int32_t match = prm.match, mism = uniform_i32(pr.mismatch), open = prm.open, ext = prm.ext;
    const int32_t policy = prm.policy;
    const int32_t c_open = (policy & SG_POL_OPEN_ON_TIE) ? 1 : 0, c_gap = (policy & SG_POL_GAP_FIRST) ? 1 : 0, c_ef = (policy & SG_POL_E_BEFORE_F) ? 1 : 0;
    int32_t open_v, ext_v, mism_v, delta_v, c_open_v, c_gap_v, c_ef_v;      // per-cell constants in VGPRs (issue cost, see k_sg_forward)
    asm volatile("v_mov_b32 %0, %1" : "=v"(open_v) : "s"(open));
    asm volatile("v_mov_b32 %0, %1" : "=v"(ext_v) : "s"(ext));
    asm volatile("v_mov_b32 %0, %1" : "=v"(mism_v) : "s"(mism));
    asm volatile("v_mov_b32 %0, %1" : "=v"(delta_v) : "s"(match - mism));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c_open_v) : "s"(c_open));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c_gap_v) : "s"(c_gap));
    asm volatile("v_mov_b32 %0, %1" : "=v"(c_ef_v) : "s"(c_ef));
    const uint64_t *planes = S.planes;
    const uint32_t nseq = S.n;
    const int32_t nchunks = (int32_t)S.nchunks;
    uint32_t *tw_base = reinterpret_cast<uint32_t *>(trace + pr.trace_off);
    // trace dword q of the lane: into the staged tile; the tile leaves when its 16th dword is in (and at the end of the pair)
    auto flush_tile = [&](int32_t tile) {
        uint4 *dst = reinterpret_cast<uint4 *>(tw_base + ((size_t)tile * 64 + lane) * SG_TILE_DWORDS);
#pragma unroll
        for (int c = 0; c < SG_TILE_DWORDS / 4; ++c)
            dst[c] = make_uint4(stage[(4 * c) * 64 + lane], stage[(4 * c + 1) * 64 + lane], stage[(4 * c + 2) * 64 + lane], stage[(4 * c + 3) * 64 + lane]);
    };
    auto put_dword = [&](int32_t q, uint32_t w) {
        stage[(q & (SG_TILE_DWORDS - 1)) * 64 + lane] = w;
        if ((q & (SG_TILE_DWORDS - 1)) == SG_TILE_DWORDS - 1) flush_tile(q >> SG_TILE_LOG);
    };
    const int32_t d0 = dlo + DPL * lane;
    int32_t H[DPL], E[DPL], F[DPL];
    uint32_t Mlo[DPL], Mhi[DPL], Mcur[DPL];
#pragma unroll
    for (int s = 0; s < DPL; ++s) { H[s] = 0; E[s] = SG_NEG; F[s] = SG_NEG; Mlo[s] = Mhi[s] = Mcur[s] = 0; }
    int32_t rowbest = SG_NEG, rowj_first = -1, rowj_last = -1;      // last query row, over this lane's columns (ascending)
    int32_t colbest = SG_NEG, coli_first = -1, coli_last = -1;      // last ref column, over this lane's rows (ascending)
    const int32_t a_end = m + n - 2;
    const int32_t row_from = 2 * (m - 1) + dlo, col_from = 2 * (n - 1) - dlo - (DIAGS - 1);
    // steps at which EVERY lane's two cells lie inside the matrix and off its last row / column: no range checks there
    const int32_t dtop = dlo + DIAGS - 1;
    const int32_t in_from = (dtop > -dlo ? dtop : -dlo) + 1;
    const int32_t in_to = (2 * (m - 1) + dlo < 2 * (n - 1) - dtop ? 2 * (m - 1) + dlo : 2 * (n - 1) - dtop) - 1;      // inclusive
    uint32_t tw = 0;

    // one cell of the recurrences; the 4 decision sign bits come back in t_f, t_e, t_g, t_x
    auto core = [&](int32_t &Hs, int32_t &Es, int32_t &Fs, int32_t Hup, int32_t Fup, int32_t Hl, int32_t El, uint32_t mbits, int32_t kbit,
                    uint32_t &t_f, uint32_t &t_e, uint32_t &t_g, uint32_t &t_x) {
        const int32_t e = __builtin_amdgcn_sbfe((int32_t)mbits, kbit, 1);      // -1 where the bases are equal
        const int32_t Fopn = Hup - open_v, Fext = EXT0 ? Fup : Fup - ext_v;
        const int32_t Eopn = Hl - open_v, Eext = EXT0 ? El : El - ext_v;
        const int32_t Fv = Fopn > Fext ? Fopn : Fext;
        const int32_t Ev = Eopn > Eext ? Eopn : Eext;
        const int32_t Hd = Hs + mism_v + (e & delta_v);
        const int32_t g = Fv > Ev ? Fv : Ev;
        const int32_t Hv = Hd > g ? Hd : g;
        int32_t d_f, d_e, d_g, d_x;
        if (POL0) { d_f = Fext - Fopn; d_e = Eext - Eopn; d_g = Hd - Hv; d_x = Fv - Ev; }
        else { d_f = Fext - Fopn - c_open_v; d_e = Eext - Eopn - c_open_v; d_g = Hd - g - c_gap_v; d_x = Fv - Ev - c_ef_v; }
        t_f = (uint32_t)d_f; t_e = (uint32_t)d_e; t_g = (uint32_t)d_g;
        t_x = __builtin_amdgcn_bitop3_b32((uint32_t)d_g, (uint32_t)d_x, (uint32_t)e, 0xC5);     // gap ? E-rather-than-F : mismatch
        Hs = Hv; Es = Ev; Fs = Fv;
    };
    // the same for a cell that may lie outside the matrix (then nothing happens), with the end-cell candidates
    auto cell = [&](int32_t a, int32_t d, int32_t &Hs, int32_t &Es, int32_t &Fs, int32_t Hup, int32_t Fup, int32_t Hl, int32_t El, uint32_t mbits, int32_t kbit,
                    uint32_t &t_f, uint32_t &t_e, uint32_t &t_g, uint32_t &t_x) {
        const int32_t i = (a - d) >> 1, j = i + d;
        t_f = t_e = t_g = t_x = 0;
        if ((uint32_t)i < (uint32_t)m && (uint32_t)j < (uint32_t)n) {
            core(Hs, Es, Fs, Hup, Fup, Hl, El, mbits, kbit, t_f, t_e, t_g, t_x);
            if (a >= row_from && i == m - 1) {
                if (Hs > rowbest) { rowbest = Hs; rowj_first = j; rowj_last = j; }
                else if (Hs == rowbest) rowj_last = j;
            }
            if (a >= col_from && j == n - 1) {      // (a lane meets the last column on its largest diagonal first: ascending rows)
                if (Hs > colbest) { colbest = Hs; coli_first = i; coli_last = i; }
                else if (Hs == colbest) coli_last = i;
            }
        }
    };
    auto push = [&](uint32_t t_f, uint32_t t_e, uint32_t t_g, uint32_t t_x) {
        tw = __builtin_amdgcn_alignbit(tw, t_f, 31);
        tw = __builtin_amdgcn_alignbit(tw, t_e, 31);
        tw = __builtin_amdgcn_alignbit(tw, t_g, 31);
        tw = __builtin_amdgcn_alignbit(tw, t_x, 31);
    };

    // One step = one anti-diagonal.  PAR = 0: slots 0 and 2, PAR = 1: slots 1 and 3 -- compile-time, and the loops below run an
    // even and an odd step per iteration, so that no register shuffling is left at the control-flow joins (a loop over
    // single steps with a run-time parity cost ~25 v_mov per step).
    // INTERIOR (compile-time): every lane's cells of the step are plain matrix cells off the last row / column -- the loop over
    // those steps (all but ~ DIAGS at either end of a pair) holds no range checks, no second code path to join with (the join cost
    // 6 v_mov per step), and its DPP shifts take their own previous result as the value of the lane without a neighbour (-inf,
    // set once before the loop), so that no register has to be re-initialised before each of them.
    int32_t hl_keep = SG_NEG, el_keep = SG_NEG, hu_keep = SG_NEG, fu_keep = SG_NEG;
    auto refill = [&](int32_t a) {
        if ((a & 127) == 0) {
#pragma unroll
            for (int s = 0; s < DPL; ++s) {
                const int32_t d = d0 + s;
                const int32_t ifirst = (a + ((a - d) & 1) - d) >> 1;            // row of the slot's first cell at or after step a
                const uint64_t ql = plane_bits64(planes, nseq, nchunks, ia, 0, ifirst), qh = plane_bits64(planes, nseq, nchunks, ia, 1, ifirst);
                const uint64_t tl = plane_bits64(planes, nseq, nchunks, ib, 0, ifirst + d), th = plane_bits64(planes, nseq, nchunks, ib, 1, ifirst + d);
                const uint64_t mk = ~(ql ^ tl) & ~(qh ^ th);
                Mlo[s] = (uint32_t)mk; Mhi[s] = (uint32_t)(mk >> 32);
                // -inf must not drift towards the end of the int32 range along a long band edge (ext per step)
                E[s] = E[s] > SG_NEG ? E[s] : SG_NEG;
                F[s] = F[s] > SG_NEG ? F[s] : SG_NEG;
            }
        }
        if ((a & 63) == 0) {
            asm volatile("");                      // keep this a (wave-uniform) branch: as selects it costs 2 DPL/2 v_cndmask on every step
#pragma unroll
            for (int s = 0; s < DPL; ++s) Mcur[s] = (a & 64) ? Mhi[s] : Mlo[s];
        }
    };
    auto step = [&](int32_t a, auto par_tag, auto interior_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        refill(a);
        const int32_t kbit = (a & 63) >> 1;
        uint32_t f0, e0, g0, x0, f1 = 0, e1 = 0, g1 = 0, x1 = 0;
        if (PAR == 0) {
            int32_t Hl, El;
            if constexpr (INTERIOR) {
                Hl = hl_keep = __builtin_amdgcn_update_dpp(hl_keep, H[DPL - 1], 0x138, 0xf, 0xf, false);      // wave_shr:1: lane l takes lane l - 1
                El = el_keep = __builtin_amdgcn_update_dpp(el_keep, E[DPL - 1], 0x138, 0xf, 0xf, false);
            } else {
                const int32_t edge = a + dlo != 0 ? SG_NEG : 0;      // left of the band: -inf, or the boundary column of cell (i, 0)
                Hl = __builtin_amdgcn_update_dpp(edge, H[DPL - 1], 0x138, 0xf, 0xf, false);
                El = __builtin_amdgcn_update_dpp(SG_NEG, E[DPL - 1], 0x138, 0xf, 0xf, false);
            }
            if constexpr (DPL == 4) {
                const int32_t h1 = H[1], e1s = E[1], f1s = F[1], h3 = H[3], f3 = F[3];
                if constexpr (INTERIOR) {
                    core(H[0], E[0], F[0], h1, f1s, Hl, El, Mcur[0], kbit, f0, e0, g0, x0);
                    core(H[2], E[2], F[2], h3, f3, h1, e1s, Mcur[2], kbit, f1, e1, g1, x1);
                } else {
                    cell(a, d0, H[0], E[0], F[0], h1, f1s, Hl, El, Mcur[0], kbit, f0, e0, g0, x0);
                    cell(a, d0 + 2, H[2], E[2], F[2], h3, f3, h1, e1s, Mcur[2], kbit, f1, e1, g1, x1);
                }
            } else {          // two diagonals per lane: slot 0 (up = the lane's slot 1, left = lane - 1's slot 1)
                const int32_t h1 = H[1], f1s = F[1];
                if constexpr (INTERIOR) core(H[0], E[0], F[0], h1, f1s, Hl, El, Mcur[0], kbit, f0, e0, g0, x0);
                else cell(a, d0, H[0], E[0], F[0], h1, f1s, Hl, El, Mcur[0], kbit, f0, e0, g0, x0);
            }
        } else {
            int32_t Hu, Fu;
            if constexpr (INTERIOR) {
                Hu = hu_keep = __builtin_amdgcn_update_dpp(hu_keep, H[0], 0x130, 0xf, 0xf, false);      // wave_shl:1: lane l takes lane l + 1
                Fu = fu_keep = __builtin_amdgcn_update_dpp(fu_keep, F[0], 0x130, 0xf, 0xf, false);
            } else {
                const int32_t edge = a - dlo - (DIAGS - 1) != 0 ? SG_NEG : 0;   // above the band: -inf, or the boundary row of cell (0, j)
                Hu = __builtin_amdgcn_update_dpp(edge, H[0], 0x130, 0xf, 0xf, false);
                Fu = __builtin_amdgcn_update_dpp(SG_NEG, F[0], 0x130, 0xf, 0xf, false);
            }
            if constexpr (DPL == 4) {
                const int32_t h0 = H[0], e0s = E[0], h2 = H[2], e2s = E[2], f2s = F[2];
                if constexpr (INTERIOR) {
                    core(H[1], E[1], F[1], h2, f2s, h0, e0s, Mcur[1], kbit, f0, e0, g0, x0);
                    core(H[3], E[3], F[3], Hu, Fu, h2, e2s, Mcur[3], kbit, f1, e1, g1, x1);
                } else {
                    cell(a, d0 + 1, H[1], E[1], F[1], h2, f2s, h0, e0s, Mcur[1], kbit, f0, e0, g0, x0);
                    cell(a, d0 + 3, H[3], E[3], F[3], Hu, Fu, h2, e2s, Mcur[3], kbit, f1, e1, g1, x1);
                }
            } else {          // slot 1 (up = lane + 1's slot 0, left = the lane's slot 0)
                const int32_t h0 = H[0], e0s = E[0];
                if constexpr (INTERIOR) core(H[1], E[1], F[1], Hu, Fu, h0, e0s, Mcur[1], kbit, f0, e0, g0, x0);
                else cell(a, d0 + 1, H[1], E[1], F[1], Hu, Fu, h0, e0s, Mcur[1], kbit, f0, e0, g0, x0);
            }
        }
        push(f0, e0, g0, x0);
        if constexpr (DPL == 4) push(f1, e1, g1, x1);
        if ((a & (SPW - 1)) == SPW - 1 && a >= 0) put_dword(a >> SPW_LOG, tw);
    };
    // the first step of a pair is the "even" kind: start one anti-diagonal early if dlo is odd (no cell lives there; its
    // byte is shifted out of the trace word before the first store), and the pair's last half may lie beyond a_end
    const int32_t a_start = -(dlo & 1);
    int32_t a = a_start;
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
#pragma unroll 1
    for (; a <= a_end && a < in_from; a += 2) {            // the band enters the matrix
        step(a, T0(), std::false_type());
        step(a + 1, T1(), std::false_type());
    }
#pragma unroll 1
    for (const int32_t in_end = in_to < a_end ? in_to : a_end; a + 1 <= in_end; a += 2) {
        step(a, T0(), std::true_type());
        step(a + 1, T1(), std::true_type());
    }
#pragma unroll 1
    for (; a <= a_end; a += 2) {                           // the band leaves it
        step(a, T0(), std::false_type());
        step(a + 1, T1(), std::false_type());
    }
    const int32_t a_done = a - 1;                 // last step taken (a_end or a_end + 1)
    {
        const int32_t q_done = a_done >> SPW_LOG;
        const bool partial = ((a_done + 1) & (SPW - 1)) != 0;          // the last dword is not full: not stored yet
        if (partial) stage[(q_done & (SG_TILE_DWORDS - 1)) * 64 + lane] = tw << (2 * DPL * (SPW - ((a_done + 1) & (SPW - 1))));
        if (partial || (q_done & (SG_TILE_DWORDS - 1)) != SG_TILE_DWORDS - 1) flush_tile(q_done >> SG_TILE_LOG);          // (else it left with its 16th dword)
    }
    // reduce the candidates over lanes: maximum; smallest (first) / largest (last) column resp. row on ties
    int32_t rb = rowbest, rf = rowj_first, rl = rowj_last, cb = colbest, cf = coli_first, cl = coli_last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t ob = __shfl_xor(rb, o, 64), of = __shfl_xor(rf, o, 64), ol = __shfl_xor(rl, o, 64);
        if (ob > rb) { rb = ob; rf = of; rl = ol; }
        else if (ob == rb) {
            if (of >= 0 && (rf < 0 || of < rf)) rf = of;
            if (ol > rl) rl = ol;
        }
        const int32_t pb = __shfl_xor(cb, o, 64), pf = __shfl_xor(cf, o, 64), pl = __shfl_xor(cl, o, 64);
        if (pb > cb) { cb = pb; cf = pf; cl = pl; }
        else if (pb == cb) {
            if (pf >= 0 && (cf < 0 || pf < cf)) cf = pf;
            if (pl > cl) cl = pl;
        }
    }
    int32_t score = 0;
    if (lane == 0) {
        const bool last = (policy & SG_POL_LAST_MAX) != 0, col_first = (policy & SG_POL_COL_FIRST) != 0;
        int32_t eq, er;
        const int32_t rj = last ? rl : rf, ci = last ? cl : cf;
        if (!col_first) {
            score = rb; eq = m - 1; er = rj;
            if (cb > score || (last && cb == score)) { score = cb; eq = ci; er = n - 1; }
        } else {
            score = cb; eq = ci; er = n - 1;
            if (rb > score || (last && rb == score)) { score = rb; eq = m - 1; er = rj; }
        }
        endinfo[(size_t)pidx * 4 + 0] = score;
        endinfo[(size_t)pidx * 4 + 1] = eq;
        endinfo[(size_t)pidx * 4 + 2] = er;
    }
    return __builtin_amdgcn_readfirstlane(score);
}

// One wave per pair of the kernel's class (SgPair::mode).  A pair with retry_x > 0 runs in a band NARROWER than its bound asks for
// (sg_host.inc: the 128 diagonals around its corridor, where the bound asks for up to 256): the result stands if it certifies itself --
// score > match * (min(m, n) - X - 1) for the half-width X that was run, so nothing outside that band can score as much -- otherwise
// the same wave runs the pair again in the band of its bound (four diagonals per lane; the trace region was sized for that), the
// pair's record says so for the walk kernel, and *n_again counts it.
template <bool POL0, bool EXT0, int DPL = 4>
__global__ __launch_bounds__(64, 4) void k_sg_band(DevStore S, SgPair *__restrict__ pairs, SgParams prm,
                                                   uint8_t *__restrict__ trace, int32_t *__restrict__ endinfo, uint32_t *__restrict__ n_again)
{
    __shared__ uint32_t stage[SG_TILE_DWORDS * 64];
    const uint32_t pidx = blockIdx.x;
    const SgPair pr = pairs[pidx];
    if (uniform_i32(pr.mode) != (DPL == 4 ? 1 : 2)) return;
    const int32_t dlo = uniform_i32(pr.dlo);
    const int32_t score = sg_band_run<POL0, EXT0, DPL>(S, pr, dlo, prm, trace, endinfo, pidx, stage);
    if constexpr (DPL == 2) {
        const int32_t xs = uniform_i32(pr.retry_x);
        if (xs > 0) {
            const int32_t m = uniform_i32(S.lens[pr.a]), n = uniform_i32(S.lens[pr.b]), D = n - m, mn = m < n ? m : n;
            const int32_t corridor_lo = D < 0 ? D : 0, corridor_hi = D > 0 ? D : 0;
            const int32_t x_run = corridor_lo - dlo;
            if (!((long long)score > (long long)prm.match * (mn - x_run - 1))) {
                sg_band_run<POL0, EXT0, 4>(S, pr, corridor_lo - xs, prm, trace, endinfo, pidx, stage);
                if (threadIdx.x == 0) {
                    pairs[pidx].dlo = corridor_lo - xs;
                    pairs[pidx].dhi = corridor_hi + xs;
                    pairs[pidx].mode = 1;
                    atomicAdd(n_again, 1u);
                }
            }
        }
    }
}

// One thread per pair: follow the trace back from the end cell, emit run-length ops (front-to-back order) into the
// pair's ops region [ops_off, ops_off + m + n + 2), right-aligned; res = score,end_q,end_r,matches,mismatches,indels;
// opcount[p] = number of ops.
template <int DUMMY>
__global__ __launch_bounds__(64) void k_sg_walk(DevStore S, const SgPair *__restrict__ pairs, const int32_t *__restrict__ Rs,
                                                 const uint8_t *__restrict__ trace, const int32_t *__restrict__ endinfo,
                                                 uint32_t *__restrict__ ops, uint32_t *__restrict__ opcount,
                                                 int32_t *__restrict__ res, uint32_t n_pairs)
{
    const uint32_t p = blockIdx.x * 64u + threadIdx.x;
    if (p >= n_pairs) return;
    const SgPair pr = pairs[p];
    if (pr.mode != 0) return;                // k_sg_band's trace layouts: k_sg_walk_band<4 | 2>
    const int32_t m = S.lens[pr.a], n = S.lens[pr.b];
    const int32_t R = Rs[p];
    const uint8_t *tb = trace + pr.trace_off;
    const int32_t steps = pr.steps;
    const int32_t score = endinfo[(size_t)p * 4], eq = endinfo[(size_t)p * 4 + 1], er = endinfo[(size_t)p * 4 + 2];
    bool left_window = false;       // the path stepped on a cell that was never computed (cannot happen inside a certified band)
    const uint64_t cap = (uint64_t)m + n + 2;
    uint32_t *region = ops + pr.ops_off;
    uint64_t pos = cap;
    uint32_t run_code = 0xffffffffu, run_len = 0;
    auto emit = [&](uint32_t code, uint32_t cnt) {
        if (cnt == 0) return;
        if (code == run_code) { run_len += cnt; return; }
        if (run_len) region[--pos] = (run_len << 4) | run_code;
        run_code = code; run_len = cnt;
    };
    emit(3, (uint32_t)(n - 1 - er));   // trailing ref bases 'D'
    emit(2, (uint32_t)(m - 1 - eq));   // trailing query bases 'I'
    int32_t i = eq, j = er, where = 0;
    int32_t nmatch = 0, nmis = 0;
    int64_t alen = (int64_t)(n - 1 - er) + (m - 1 - eq);
    while (i >= 0 && j >= 0) {
        const int32_t pass = i / (64 * R), ip = i - pass * 64 * R;
        const int32_t l = ip / R, r = ip - l * R;
        int32_t jlo, jhi;
        sg_window(pass * 64 * R, 64 * R, pr.dlo, pr.dhi, n, jlo, jhi);
        if (j < jlo || j > jhi) { left_window = true; break; }
        const uint32_t word = *reinterpret_cast<const uint32_t *>(tb + (((size_t)pass * steps + (size_t)(j - jlo + l)) * 64 + l) * (size_t)(R / 2));
        const uint32_t tr = (word >> (28 - 4 * r)) & 15u;
        if (where == 0) {
            if (!(tr & SG_BIT_GAP)) {
                if (tr & SG_BIT_X) { emit(1, 1); ++nmis; } else { emit(0, 1); ++nmatch; }
                ++alen; --i; --j;
            } else where = (tr & SG_BIT_X) ? 2 : 1;
        } else if (where == 1) {
            emit(2, 1); ++alen;
            where = (tr & SG_BIT_FOPEN) ? 0 : 1;
            --i;
        } else {
            emit(3, 1); ++alen;
            where = (tr & SG_BIT_EOPEN) ? 0 : 2;
            --j;
        }
    }
    if (i >= 0) { emit(2, (uint32_t)(i + 1)); alen += i + 1; }
    if (j >= 0) { emit(3, (uint32_t)(j + 1)); alen += j + 1; }
    if (run_len) region[--pos] = (run_len << 4) | run_code;
    opcount[p] = (uint32_t)(cap - pos);
    int32_t *o = res + (size_t)p * 6;
    o[0] = left_window ? SG_NEG : score; o[1] = eq; o[2] = er; o[3] = nmatch; o[4] = nmis; o[5] = (int32_t)(alen - nmatch - nmis);
}

// The same walk over k_sg_band's trace (byte of step a = i + j in dword a / 4 of the lane that owns the diagonal): one thread
// per pair, the state machine written with selects (64 pairs in 64 different states share the wave: every branch body is paid
// by all of them), two tiles (two 64-byte lines: 32 dwords) of the diagonal's lane fetched at a time.
__global__ __launch_bounds__(64) void k_sg_walk_band(DevStore S, const SgPair *__restrict__ pairs, const uint8_t *__restrict__ trace,
                                                      const int32_t *__restrict__ endinfo, uint32_t *__restrict__ ops,
                                                      uint32_t *__restrict__ opcount, int32_t *__restrict__ res, uint32_t n_pairs)
{
    const uint32_t p = blockIdx.x * 64u + threadIdx.x;
    if (p >= n_pairs) return;
    const SgPair pr = pairs[p];
    if (pr.mode != 1 && pr.mode != 2) return;
    // k_sg_band<.., DPL>'s layout, DPL = 4 (mode 1) or 2 (mode 2) diagonals per lane: DPL / 2 nibbles per step, 4 or 8 steps per dword.  One
    // launch walks both classes (a launch per class cost the latency of a thread's ~5 000 dependent steps twice)
    const bool four = pr.mode == 1;
    const int32_t DIAGS = four ? 256 : 128, SPW_LOG = four ? 2 : 3, DPL_LOG = four ? 2 : 1;
    const int32_t m = S.lens[pr.a], n = S.lens[pr.b];
    const uint32_t *tb = reinterpret_cast<const uint32_t *>(trace + pr.trace_off);
    const int32_t score = endinfo[(size_t)p * 4], eq = endinfo[(size_t)p * 4 + 1], er = endinfo[(size_t)p * 4 + 2];
    const uint64_t cap = (uint64_t)m + n + 2;
    uint32_t *region = ops + pr.ops_off;
    uint64_t pos = cap;
    uint32_t run_code = 0xffffffffu, run_len = 0;
    auto emit = [&](uint32_t code, uint32_t cnt) {
        if (cnt == 0) return;
        if (code == run_code) { run_len += cnt; return; }
        if (run_len) region[--pos] = (run_len << 4) | run_code;
        run_code = code; run_len = cnt;
    };
    emit(3, (uint32_t)(n - 1 - er));   // trailing ref bases 'D'
    emit(2, (uint32_t)(m - 1 - eq));   // trailing query bases 'I'
    int32_t i = eq, j = er;
    uint32_t where = 0;
    int32_t nmatch = 0, nmis = 0;
    int64_t alen = (int64_t)(n - 1 - er) + (m - 1 - eq);
    bool left_window = false;
    // Trace cache, in LDS: two tiles (32 dwords: 256 / 128 anti-diagonals) of the diagonal's lane AND of the lanes either side of it, per
    // thread.  With one thread per pair nothing hides a dependent HBM / Infinity-Cache load, and a load that ONE of the wave's 64 pairs
    // waits for is waited for by all of them (each pair ran out of its own two tiles, or changed lane at an indel, every ~3rd iteration
    // of the wave: 2.1 of the walk's 2.8 ms at C3).  So refills are collective: when any pair of the wave needs one, every pair takes
    // the six lines around its current position (unconditional addresses, one wait) -- a wave then waits once per >= 64 iterations, an
    // indel that moves a path to the next lane finds that lane's tiles in place.  Row stride 97: the threads spread over the banks.
    __shared__ uint32_t s_cache[64][97];
    uint32_t *my = s_cache[threadIdx.x];
    int32_t c_lane = -(1 << 20), c_top = -(1 << 30);
    uint32_t word = 0;                       // the dword in hand: dword w_q of lane w_lc
    int32_t w_q = -1, w_lc = -1;
    const int32_t dlo = pr.dlo;
    while (i >= 0 && j >= 0) {
        const int32_t sl = (j - i) - dlo;
        if ((uint32_t)sl >= (uint32_t)DIAGS) { left_window = true; break; }
        const int32_t a = i + j, q = a >> SPW_LOG, lc = sl >> DPL_LOG;
        const bool need = lc < c_lane - 1 || lc > c_lane + 1 || q > c_top || q < c_top - 31;
        if (__ballot(need) != 0) {
            const int32_t tile = q >> SG_TILE_LOG, below = tile > 0 ? tile - 1 : 0;
            c_lane = lc; c_top = tile * SG_TILE_DWORDS + SG_TILE_DWORDS - 1;
            uint4 w[3][8];
#pragma unroll
            for (int side = 0; side < 3; ++side) {
                int32_t ln = lc - 1 + side;
                ln = ln < 0 ? 0 : (ln > 63 ? 63 : ln);
                const uint4 *hi = reinterpret_cast<const uint4 *>(tb + ((size_t)tile * 64 + ln) * SG_TILE_DWORDS);
                const uint4 *lo = reinterpret_cast<const uint4 *>(tb + ((size_t)below * 64 + ln) * SG_TILE_DWORDS);
#pragma unroll
                for (int x = 0; x < 4; ++x) { w[side][x] = hi[x]; w[side][4 + x] = lo[x]; }
            }
#pragma unroll
            for (int side = 0; side < 3; ++side) {
                uint32_t *row = my + 32 * side;          // row[c_top - q']: the top dword first
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    row[15 - 4 * x] = w[side][x].x; row[14 - 4 * x] = w[side][x].y; row[13 - 4 * x] = w[side][x].z; row[12 - 4 * x] = w[side][x].w;
                    row[31 - 4 * x] = w[side][4 + x].x; row[30 - 4 * x] = w[side][4 + x].y; row[29 - 4 * x] = w[side][4 + x].z; row[28 - 4 * x] = w[side][4 + x].w;
                }
            }
            w_q = -1;
        }
        if (q != w_q || lc != w_lc) { word = my[32 * (lc - c_lane + 1) + (c_top - q)]; w_q = q; w_lc = lc; }
        const uint32_t x = word >> (four ? 8 * (3 - (a & 3)) + ((sl & 2) ? 0 : 4) : 4 * (7 - (a & 7)));
        // plain matches first: the cells of this diagonal that the dword in hand holds below the current one lie 8 (two diagonals per
        // lane) or 16 bits apart; as many of them as continue the path diagonally over equal bases are taken in one iteration
        uint32_t c = 0;
        if (where == 0) {
            const uint32_t y = x & (four ? 0x00030003u : 0x03030303u);
            const uint32_t held = four ? (uint32_t)((a & 3) >> 1) + 1u : (uint32_t)((a & 7) >> 1) + 1u;
            c = y ? (uint32_t)__builtin_ctz(y) >> (four ? 4 : 3) : held;
            c = c < held ? c : held;
            const uint32_t room = (uint32_t)(i < j ? i : j) + 1u;
            c = c < room ? c : room;
        }
        if (c) {
            if (run_code == 0u) run_len += c;
            else {
                if (run_len) region[--pos] = (run_len << 4) | run_code;
                run_code = 0u; run_len = c;
            }
            i -= (int32_t)c; j -= (int32_t)c; nmatch += (int32_t)c; alen += c;
            continue;
        }
        const uint32_t tr = x & 15u;
        const uint32_t gap = (tr >> 1) & 1u, xbit = tr & 1u, fopen = (tr >> 3) & 1u, eopen = (tr >> 2) & 1u;
        const bool w0 = where == 0, w1 = where == 1;
        const bool prod = w0 ? gap == 0 : true;                                   // this step emits one alignment column
        const uint32_t code = w0 ? xbit : (w1 ? 2u : 3u);                         // 0 '=', 1 'X', 2 'I', 3 'D'
        where = w0 ? (gap ? 1u + xbit : 0u) : (w1 ? (fopen ? 0u : 1u) : (eopen ? 0u : 2u));
        if (prod) {
            if (code == run_code) ++run_len;
            else {
                if (run_len) region[--pos] = (run_len << 4) | run_code;
                run_code = code; run_len = 1;
            }
            i -= code != 3u;
            j -= code != 2u;
            nmatch += code == 0u;
            nmis += code == 1u;
            ++alen;
        }
    }
    if (i >= 0) { emit(2, (uint32_t)(i + 1)); alen += i + 1; }
    if (j >= 0) { emit(3, (uint32_t)(j + 1)); alen += j + 1; }
    if (run_len) region[--pos] = (run_len << 4) | run_code;
    opcount[p] = (uint32_t)(cap - pos);
    int32_t *o = res + (size_t)p * 6;
    o[0] = left_window ? SG_NEG : score; o[1] = eq; o[2] = er; o[3] = nmatch; o[4] = nmis; o[5] = (int32_t)(alen - nmatch - nmis);
}

// Gapped strings (what cigar_to_seq builds in the reference, SW_alignment_module.py:15-56): one block per pair, one op
// per thread (strided); '=' / 'X' copy both bases, 'I' = query base over '-', 'D' = '-' over ref base.
__global__ __launch_bounds__(256) void k_sg_expand(DevStore S, const SgPair *__restrict__ pairs, const uint32_t *__restrict__ ops,
                                                    const uint32_t *__restrict__ opcount, const uint64_t *__restrict__ aln_off,
                                                    uint8_t *__restrict__ aln_a, uint8_t *__restrict__ aln_b, uint32_t n_pairs)
{
    __shared__ uint32_t s_pos[3];   // running (alignment column, query index, ref index) while thread 0 scans the ops
    extern __shared__ uint32_t s_start[];   // per op: alignment column, query index, ref index (3 words)
    const uint32_t p = blockIdx.x;
    if (p >= n_pairs) return;
    const SgPair pr = pairs[p];
    const uint64_t cap = (uint64_t)S.lens[pr.a] + S.lens[pr.b] + 2;
    const uint32_t cnt = opcount[p];
    const uint32_t *po = ops + pr.ops_off + (cap - cnt);
    const uint64_t base = aln_off[p];
    const uint64_t *planes = S.planes;
    const uint32_t nseq = S.n;
    const char lut[4] = {'A', 'C', 'G', 'T'};
    (void)s_pos;
    // chunks of up to 1024 ops: thread 0 prefix-sums, then everyone expands
    uint32_t col = 0, qi = 0, ri = 0;
    for (uint32_t k0 = 0; k0 < cnt; k0 += 1024) {
        const uint32_t kn = (cnt - k0) < 1024 ? (cnt - k0) : 1024;
        if (threadIdx.x == 0) {
            uint32_t c = col, q = qi, r = ri;
            for (uint32_t k = 0; k < kn; ++k) {
                const uint32_t op = po[k0 + k], len = op >> 4, code = op & 15u;
                s_start[3 * k] = c; s_start[3 * k + 1] = q; s_start[3 * k + 2] = r;
                c += len;
                if (code != 3) q += len;
                if (code != 2) r += len;
            }
            s_start[3 * kn] = c; s_start[3 * kn + 1] = q; s_start[3 * kn + 2] = r;
        }
        __syncthreads();
        // one WAVE per op, its lanes on consecutive alignment columns: 64-byte stores, and the lanes of a run read the same plane words (a thread
        // per op wrote its ~40 bytes one by one while three quarters of the workgroup had no op: 4.1 ms per 50 000 pairs)
        for (uint32_t k = threadIdx.x >> 6; k < kn; k += 4) {
            const uint32_t op = po[k0 + k], len = op >> 4, code = op & 15u;
            const uint32_t c0 = s_start[3 * k], q0 = s_start[3 * k + 1], r0 = s_start[3 * k + 2];
            for (uint32_t x = threadIdx.x & 63u; x < len; x += 64u) {
                char ca = '-', cb = '-';
                if (code != 3) {
                    const uint32_t i = q0 + x;
                    const uint64_t lo = planes[((size_t)(i >> 6) * nseq + pr.a) * 2], hi = planes[((size_t)(i >> 6) * nseq + pr.a) * 2 + 1];
                    ca = lut[((lo >> (i & 63)) & 1) | (((hi >> (i & 63)) & 1) << 1)];
                }
                if (code != 2) {
                    const uint32_t j = r0 + x;
                    const uint64_t lo = planes[((size_t)(j >> 6) * nseq + pr.b) * 2], hi = planes[((size_t)(j >> 6) * nseq + pr.b) * 2 + 1];
                    cb = lut[((lo >> (j & 63)) & 1) | (((hi >> (j & 63)) & 1) << 1)];
                }
                aln_a[base + c0 + x] = (uint8_t)ca;
                aln_b[base + c0 + x] = (uint8_t)cb;
            }
        }
        __syncthreads();
        col = s_start[3 * kn]; qi = s_start[3 * kn + 1]; ri = s_start[3 * kn + 2];
        __syncthreads();
    }
}

// dense copy of the right-aligned op regions
__global__ __launch_bounds__(256) void k_sg_compact(const SgPair *__restrict__ pairs, const DevStore S,
                                                     const uint32_t *__restrict__ ops, const uint32_t *__restrict__ opcount,
                                                     const uint64_t *__restrict__ dense_off, uint32_t *__restrict__ dense,
                                                     uint32_t n_pairs)
{
    const uint32_t p = blockIdx.x;
    if (p >= n_pairs) return;
    const SgPair pr = pairs[p];
    const uint64_t cap = (uint64_t)S.lens[pr.a] + S.lens[pr.b] + 2;
    const uint32_t cnt = opcount[p];
    const uint32_t *src = ops + pr.ops_off + (cap - cnt);
    uint32_t *dst = dense + dense_off[p];
    for (uint32_t k = threadIdx.x; k < cnt; k += 256) dst[k] = src[k];
}

}  // namespace isocon
