// sg.hpp -- semi-global affine-gap alignment with traceback (kernels).
//
// Replaces parasail.sg_trace_scan_16/32 + CIGAR decode as used by parasail_alignment
// (/root/reference/modules/SW_alignment_module.py:64-86); semantics restated in oracle/isocon_oracle.c
// (orc_sg_trace): Gotoh recurrences, free end gaps on both sequences, a gap of length g costs open+(g-1)*ext.
//
// Forward kernel: one wavefront per pair, systolic over text columns.  Lane l owns query rows [l*R, (l+1)*R) and
// works on column s-l at step s; the bottom-row (H, F) of a strip and the text base travel one lane down per step.
// Each cell emits a 4-bit trace code; a lane's R codes of one column are stored as R/2 contiguous bytes at
// ((s*64 + lane) * R/2) -- i.e. indexed by STEP, not by column, so that all 64 lanes of a step write one contiguous
// 32*R-byte block (coalesced HBM writes; this kernel is HBM-write bound: ~ m*n/2 bytes per pair).
// Walk kernel: one thread per pair follows the codes back from the end cell and emits run-length CIGAR ops.
#pragma once
#include "common.hpp"

namespace isocon {

// trace nibble: bits 0-1 = source of H (0 diagonal/match, 3 diagonal/mismatch, 1 F = vertical gap = consumes a
// query base = 'I', 2 E = horizontal gap = consumes a ref base = 'D'); bit 2: E came from E (extend); bit 3: F from F.
static constexpr uint32_t SG_SRC_DIAG_EQ = 0, SG_SRC_F = 1, SG_SRC_E = 2, SG_SRC_DIAG_X = 3, SG_E_EXT = 4, SG_F_EXT = 8;
static constexpr int32_t SG_NEG = -(1 << 28);

// tie policy bits, identical to oracle/isocon_oracle.c
static constexpr int SG_POL_E_BEFORE_F = 1, SG_POL_OPEN_ON_TIE = 2, SG_POL_COL_FIRST = 4, SG_POL_LAST_MAX = 8, SG_POL_GAP_FIRST = 16;

struct SgParams {
    int32_t match, open, ext, policy;
};

struct SgPair {            // per pair of the batch
    uint32_t a, b;         // query (rows), ref (columns)
    int32_t mismatch;
    uint32_t pad;
    uint64_t trace_off;    // byte offset of this pair's trace in the scratch buffer
    uint64_t ops_off;      // first slot of this pair's ops region (capacity m + n + 2)
    uint64_t bound_off;    // first int2 of this pair's pass-boundary row (capacity n)
};

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
template <int R, bool GENERAL>
__global__ __launch_bounds__(64, 4) void k_sg_forward(DevStore S, const SgPair *__restrict__ pairs, SgParams prm,
                                                    uint8_t *__restrict__ trace, int2 *__restrict__ bound_all,
                                                    int32_t *__restrict__ endinfo)
{
    static_assert(R == 8 || R == 16, "strip height");
    const uint32_t pidx = blockIdx.x;
    const int lane = threadIdx.x;
    const SgPair pr = pairs[pidx];
    const uint32_t ia = (uint32_t)uniform_i32((int32_t)pr.a), ib = (uint32_t)uniform_i32((int32_t)pr.b);
    const int32_t m = uniform_i32(S.lens[ia]), n = uniform_i32(S.lens[ib]);
    const int32_t match = prm.match, mism = uniform_i32(pr.mismatch), open = prm.open, ext = prm.ext;
    const int32_t policy = prm.policy;
    const uint64_t *planes = S.planes;
    const uint32_t nseq = S.n;
    const int32_t nchunks = (int32_t)S.nchunks;
    int2 *bound = bound_all + pr.bound_off;
    uint8_t *tbase = trace + pr.trace_off;
    const int32_t steps = n + 63;
    const int32_t passes = (m + 64 * R - 1) / (64 * R);

    const int32_t last_i = m - 1;
    const int32_t pstar = last_i / (64 * R), lstar = (last_i % (64 * R)) / R, rstar = last_i % R;
    int32_t rowbest = SG_NEG, rowj_first = -1, rowj_last = -1;      // last query row, over columns
    int32_t colbest = SG_NEG, coli_first = -1, coli_last = -1;      // last ref column, over this lane's rows

#pragma unroll 1
    for (int32_t pass = 0; pass < passes; ++pass) {
        const int32_t row0 = pass * 64 * R + lane * R;
        const uint64_t qlo = plane_bits64(planes, nseq, nchunks, ia, 0, row0);
        const uint64_t qhi = plane_bits64(planes, nseq, nchunks, ia, 1, row0);
        int32_t Hp[R], Ep[R];   // H[i][j-1], E[i][j-1]
#pragma unroll
        for (int r = 0; r < R; ++r) { Hp[r] = 0; Ep[r] = SG_NEG; }
        int32_t diag_in = 0;     // H[row0-1][j-1]
        int32_t sendH = 0, sendFC = SG_NEG * 4;
        uint64_t tlo = 0, thi = 0;
        const bool lane_has_rows = row0 < m;
#pragma unroll 1
        for (int32_t s = 0; s < steps; ++s) {
            if ((s & 63) == 0) {   // wave-uniform: next 64 ref bases
                const int32_t tc = s >> 6;
                if (tc < nchunks) { tlo = planes[((size_t)tc * nseq + ib) * 2]; thi = planes[((size_t)tc * nseq + ib) * 2 + 1]; }
                else { tlo = 0; thi = 0; }
            }
            int32_t Hup = __shfl_up(sendH, 1, 64);
            const int32_t FC = __shfl_up(sendFC, 1, 64);
            int32_t Fup, ch;
            if (lane == 0) {
                ch = (int32_t)(((tlo >> (s & 63)) & 1) | (((thi >> (s & 63)) & 1) << 1));
                if (pass == 0 || s >= n) { Hup = 0; Fup = SG_NEG; }
                else {
                    // bottom row of the previous pass; bypass this CU's L1 (written by lane 63 of this very wave)
                    const int32_t *bp = reinterpret_cast<const int32_t *>(bound + s);
                    Hup = __hip_atomic_load(bp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    Fup = __hip_atomic_load(bp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                Fup = FC >> 2;
                ch = FC & 3;
            }
            const int32_t j = s - lane;
            const bool act = j >= 0 && j < n && lane_has_rows;
            const int32_t diag_next = Hup;
            if (act) {
                const uint64_t slo = (ch & 1) ? ~(uint64_t)0 : 0, shi = (ch & 2) ? ~(uint64_t)0 : 0;
                const uint32_t eq = (uint32_t)(~(qlo ^ slo) & ~(qhi ^ shi));
                int32_t diag = (j == 0) ? 0 : diag_in;   // left boundary column: H[.][-1] = 0
                uint32_t tw[R / 8];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int32_t Hl = Hp[r], El = Ep[r];
                    const int32_t Fopn = Hup - open, Fext = Fup - ext;
                    const int32_t Eopn = Hl - open, Eext = El - ext;
                    bool fext, eext;
                    if (GENERAL && (policy & SG_POL_OPEN_ON_TIE)) { fext = Fopn < Fext; eext = Eopn < Eext; }
                    else { fext = Fopn <= Fext; eext = Eopn <= Eext; }
                    const int32_t Fv = Fopn > Fext ? Fopn : Fext;
                    const int32_t Ev = Eopn > Eext ? Eopn : Eext;
                    const uint32_t eqbit = (eq >> r) & 1u;
                    const int32_t Hd = diag + (eqbit ? match : mism);
                    const int32_t g = Fv > Ev ? Fv : Ev;
                    const int32_t Hv = Hd > g ? Hd : g;
                    const uint32_t dcode = eqbit ? SG_SRC_DIAG_EQ : SG_SRC_DIAG_X;
                    uint32_t src;
                    if (!GENERAL) {
                        src = (Hd >= g) ? dcode : ((Fv >= Ev) ? SG_SRC_F : SG_SRC_E);
                    } else {
                        const bool gap_first = (policy & SG_POL_GAP_FIRST) != 0, e_first = (policy & SG_POL_E_BEFORE_F) != 0;
                        const bool isD = Hd == Hv, isF = Fv == Hv, isE = Ev == Hv;
                        if (!gap_first && isD) src = dcode;
                        else if (e_first) src = isE ? SG_SRC_E : (isF ? SG_SRC_F : dcode);
                        else src = isF ? SG_SRC_F : (isE ? SG_SRC_E : dcode);
                    }
                    const uint32_t nib = src | (eext ? SG_E_EXT : 0u) | (fext ? SG_F_EXT : 0u);
                    if ((r & 7) == 0) tw[r >> 3] = nib; else tw[r >> 3] |= nib << (4 * (r & 7));
                    diag = Hl;
                    Hp[r] = Hv;
                    Ep[r] = Ev;
                    Hup = Hv;
                    Fup = Fv;
                }
                // end-cell candidates: last query row (one lane of one pass), last ref column (every lane, once)
                if (pass == pstar && lane == lstar) {
                    int32_t hv = Hp[0];
#pragma unroll
                    for (int r = 1; r < R; ++r) hv = (r == rstar) ? Hp[r] : hv;
                    if (hv > rowbest) { rowbest = hv; rowj_first = j; rowj_last = j; }
                    else if (hv == rowbest) rowj_last = j;
                }
                if (j == n - 1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (row0 + r < m) {
                            if (Hp[r] > colbest) { colbest = Hp[r]; coli_first = row0 + r; coli_last = row0 + r; }
                            else if (Hp[r] == colbest) coli_last = row0 + r;
                        }
                    }
                }
                // one contiguous 32*R-byte block per wave and step
                uint32_t *dst = reinterpret_cast<uint32_t *>(tbase + (((size_t)pass * steps + s) * 64 + lane) * (R / 2));
#pragma unroll
                for (int w = 0; w < R / 8; ++w) dst[w] = tw[w];
                if (lane == 63 && pass + 1 < passes) { int2 v; v.x = Hup; v.y = Fup; bound[j] = v; }
            }
            diag_in = diag_next;
            sendH = Hup;
            sendFC = (int32_t)(((uint32_t)(Fup < SG_NEG ? SG_NEG : Fup) << 2) | (uint32_t)ch);
        }
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
    const int32_t m = S.lens[pr.a], n = S.lens[pr.b];
    const int32_t R = Rs[p];
    const uint8_t *tb = trace + pr.trace_off;
    const int32_t steps = n + 63;
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
    int32_t i = eq, j = er, where = 0;
    int32_t nmatch = 0, nmis = 0;
    int64_t alen = (int64_t)(n - 1 - er) + (m - 1 - eq);
    while (i >= 0 && j >= 0) {
        const int32_t pass = i / (64 * R), ip = i - pass * 64 * R;
        const int32_t l = ip / R, r = ip - l * R;
        const uint8_t byte = tb[(((size_t)pass * steps + (size_t)(j + l)) * 64 + l) * (size_t)(R / 2) + (r >> 1)];
        const uint32_t tr = (byte >> (4 * (r & 1))) & 15u;
        if (where == 0) {
            const uint32_t src = tr & 3u;
            if (src == SG_SRC_DIAG_EQ) { emit(0, 1); ++nmatch; ++alen; --i; --j; }
            else if (src == SG_SRC_DIAG_X) { emit(1, 1); ++nmis; ++alen; --i; --j; }
            else where = (src == SG_SRC_F) ? 1 : 2;
        } else if (where == 1) {
            emit(2, 1); ++alen;
            where = (tr & SG_F_EXT) ? 1 : 0;
            --i;
        } else {
            emit(3, 1); ++alen;
            where = (tr & SG_E_EXT) ? 2 : 0;
            --j;
        }
    }
    if (i >= 0) { emit(2, (uint32_t)(i + 1)); alen += i + 1; }
    if (j >= 0) { emit(3, (uint32_t)(j + 1)); alen += j + 1; }
    if (run_len) region[--pos] = (run_len << 4) | run_code;
    opcount[p] = (uint32_t)(cap - pos);
    int32_t *o = res + (size_t)p * 6;
    o[0] = score; o[1] = eq; o[2] = er; o[3] = nmatch; o[4] = nmis; o[5] = (int32_t)(alen - nmatch - nmis);
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
        for (uint32_t k = threadIdx.x; k < kn; k += 256) {
            const uint32_t op = po[k0 + k], len = op >> 4, code = op & 15u;
            const uint32_t c0 = s_start[3 * k], q0 = s_start[3 * k + 1], r0 = s_start[3 * k + 2];
            for (uint32_t x = 0; x < len; ++x) {
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
