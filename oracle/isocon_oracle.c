/*
 * isocon_oracle.c -- CPU ORACLE for the IsoCon all-pairs alignment + nearest-neighbour-graph hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library.  The product path (isocon_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" with respect to the third-party native libraries.  The reference
 * (ksahlin/IsoCon v0.3.3, /root/reference) is pure Python; the arithmetic of this path lives in two
 * un-vendored dependencies that are absent from the reference tree and from this image:
 *     edlib    (requirements.txt:1 ">=1.1.2"; docs/version_history.txt:6 "edlib==1.2.1")
 *     parasail (requirements.txt:4 ">=1.1.10"; setup.py:80 ">=1.1.11")
 * and the reference ships no test that pins a single value of this path.
 * What is restated here:
 *   - edit distance: the published definition (unit-cost Levenshtein, global "NW" mode; "-1 iff
 *     distance > k" as consumed at modules/nearest_neighbor_graph.py:156-162,387-395).  Distances are
 *     unique integers, so parity with edlib is algorithm-independent; orc_ed_dp() (textbook DP) is the
 *     authority and orc_ed_bounded() (Myers 1999 bit-vector blocks + Ukkonen band, the algorithm edlib
 *     publishes: Sosic & Sikic 2017) is validated against it.
 *   - nearest-neighbour loops: modules/nearest_neighbor_graph.py:110-198 (1-set) and :341-424 (2-set),
 *     statement by statement, including the seed dictionary, sticky stops and the depth rules.
 *   - semi-global affine alignment with traceback: parasail's published `sg_trace_scan` semantics
 *     (Daily 2016; Gotoh recurrences, free end gaps on both sequences, CIGAR with =,X,I,D) as consumed
 *     at modules/SW_alignment_module.py:64-86.  Scores are unique; CIGARs depend on three tie decisions
 *      which are parameters here (`policy` bits) with parasail's believed behaviour
 *     as policy 0.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * 1. Edit distance
 * ---------------------------------------------------------------------------------------------- */

/* Textbook O(m*n) Levenshtein DP, two rows.  The authority for every distance in the test-suite. */
int32_t orc_ed_dp(const uint8_t *q, int32_t m, const uint8_t *t, int32_t n)
{
    if (m == 0) return n;
    if (n == 0) return m;
    int32_t *row = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1));
    for (int32_t i = 0; i <= m; ++i) row[i] = i;
    for (int32_t j = 1; j <= n; ++j) {
        int32_t diag = row[0];
        row[0] = j;
        const uint8_t c = t[j - 1];
        for (int32_t i = 1; i <= m; ++i) {
            int32_t up = row[i - 1];   /* D[i-1][j]   */
            int32_t left = row[i];     /* D[i][j-1]   */
            int32_t best = diag + (q[i - 1] != c);
            if (up + 1 < best) best = up + 1;
            if (left + 1 < best) best = left + 1;
            diag = left;
            row[i] = best;
        }
    }
    int32_t r = row[m];
    free(row);
    return r;
}

/* One 64-row block of Myers' bit-vector column update.  hin/hout are the horizontal deltas entering
 * at the top / leaving at the bottom of the block (each in {-1,0,+1}). */
static inline int myers_block(uint64_t Pv, uint64_t Mv, uint64_t Eq, int hin, uint64_t *Pvo, uint64_t *Mvo)
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
    *Pvo = Mh | ~(Xv | Ph);
    *Mvo = Ph & Xv;
    return hout;
}

typedef struct {
    int32_t m;
    int32_t nblk;
    int32_t sym_index[256]; /* -1 = symbol absent from the query */
    int32_t nsym;
    uint64_t *peq;          /* [nsym+1][nblk]; last row all-zero (symbols absent from the query) */
} orc_query;

static void orc_query_init(orc_query *Q, const uint8_t *q, int32_t m)
{
    Q->m = m;
    Q->nblk = (m + 63) / 64;
    Q->nsym = 0;
    for (int i = 0; i < 256; ++i) Q->sym_index[i] = -1;
    for (int32_t i = 0; i < m; ++i)
        if (Q->sym_index[q[i]] < 0) Q->sym_index[q[i]] = Q->nsym++;
    Q->peq = (uint64_t *)calloc((size_t)(Q->nsym + 1) * (size_t)(Q->nblk ? Q->nblk : 1), sizeof(uint64_t));
    for (int32_t i = 0; i < m; ++i)
        Q->peq[(size_t)Q->sym_index[q[i]] * Q->nblk + (i >> 6)] |= (uint64_t)1 << (i & 63);
}

static void orc_query_free(orc_query *Q) { free(Q->peq); }

/* Banded block Myers for one fixed k >= 0.  Returns the distance if <= k, else -1.
 * Band: Ukkonen's diagonals e = i - j in [min(0,d) - x, max(0,d) + x], d = m - n, x = (k-|d|)/2,
 * in 64-row block granularity.  Blocks that enter the band are initialised with the "+1 per row"
 * upper bound; values inside the band are therefore upper bounds of the true DP values and exact
 * for every cell whose optimal path stays inside the band -- which includes (m,n) whenever the
 * true distance is <= k. */
static int32_t ed_fixed_k(const orc_query *Q, const uint8_t *t, int32_t n, int32_t k,
                          uint64_t *P, uint64_t *M, int32_t *score)
{
    const int32_t m = Q->m;
    const int32_t d = m - n;
    const int32_t ad = d < 0 ? -d : d;
    if (ad > k) return -1;
    if (m == 0) return n;   /* n <= k here */
    if (n == 0) return m;
    const int32_t x = (k - ad) / 2;
    const int32_t emin = (d < 0 ? d : 0) - x;
    const int32_t emax = (d > 0 ? d : 0) + x;
    const int32_t nblk = Q->nblk;

    int32_t fb = 0;
    int32_t lb = -1;
    for (int32_t j = 1; j <= n; ++j) {
        /* rows of this column that lie in the band */
        int32_t lo = j + emin; if (lo < 1) lo = 1;
        int32_t hi = j + emax; if (hi > m) hi = m;
        if (lo > hi) return -1;          /* band left the matrix: cannot happen for |d| <= k */
        const int32_t want_fb = (lo - 1) >> 6;
        const int32_t want_lb = (hi - 1) >> 6;
        while (lb < want_lb) {           /* block enters the band at its lower edge */
            ++lb;
            P[lb] = ~(uint64_t)0;
            M[lb] = 0;
            score[lb] = (lb == 0 ? 0 : score[lb - 1]) + 64;
            if (lb == 0) score[0] = 64 + (j - 1); /* D[64][j-1] upper bound via D[0][j-1] = j-1 */
        }
        if (fb < want_fb) fb = want_fb;
        const int32_t si = Q->sym_index[t[j - 1]];
        const uint64_t *eq = Q->peq + (size_t)(si < 0 ? Q->nsym : si) * nblk;
        int hin = 1;                     /* true boundary (row 0) or band edge: +1 */
        for (int32_t b = fb; b <= lb; ++b) {
            hin = myers_block(P[b], M[b], eq[b], hin, &P[b], &M[b]);
            score[b] += hin;
        }
        /* early exit on the final diagonal: D[j+d][j] is non-decreasing in j and ends at D[m][n] */
        const int32_t r = j + d;
        if (r >= 1 && r <= m) {
            const int32_t b = (r - 1) >> 6;
            if (b >= fb && b <= lb) {
                const int32_t bit = (r - 1) & 63;
                const uint64_t above = bit == 63 ? 0 : (~(uint64_t)0 << (bit + 1));
                const int32_t val = score[b] - __builtin_popcountll(P[b] & above) + __builtin_popcountll(M[b] & above);
                if (val > k) return -1;
                if (j == n) return val; /* r == m */
            }
        }
    }
    return -1; /* not reached */
}

/* Global edit distance with edlib's k semantics: k >= 0 -> distance if <= k else -1;
 * k < 0 -> unbounded (band doubling from 64, as edlib does). */
int32_t orc_ed_bounded(const uint8_t *q, int32_t m, const uint8_t *t, int32_t n, int32_t k)
{
    orc_query Q;
    orc_query_init(&Q, q, m);
    const int32_t nb = Q.nblk ? Q.nblk : 1;
    uint64_t *P = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nb * 2);
    uint64_t *M = P + nb;
    int32_t *score = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb);
    int32_t r;
    if (k >= 0) {
        r = ed_fixed_k(&Q, t, n, k, P, M, score);
    } else {
        int32_t kk = 64;
        const int32_t kmax = m > n ? m : n;
        for (;;) {
            r = ed_fixed_k(&Q, t, n, kk, P, M, score);
            if (r >= 0 || kk >= kmax) break;
            kk *= 2;
        }
    }
    free(P);
    free(score);
    orc_query_free(&Q);
    return r;
}

/* Batch over an explicit pair list.  seqs = concatenated bytes, off[i]..off[i+1] = sequence i. */
void orc_ed_pairs(const uint8_t *seqs, const int64_t *off, const int32_t *a, const int32_t *b,
                  const int32_t *k, int64_t n_pairs, int32_t *out)
{
    for (int64_t p = 0; p < n_pairs; ++p) {
        const int32_t ia = a[p], ib = b[p];
        out[p] = orc_ed_bounded(seqs + off[ia], (int32_t)(off[ia + 1] - off[ia]),
                                seqs + off[ib], (int32_t)(off[ib + 1] - off[ib]), k ? k[p] : -1);
    }
}

/* ------------------------------------------------------------------------------------------------
 * 2. Nearest-neighbour loops (modules/nearest_neighbor_graph.py)
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    int32_t *idx;   /* neighbour index (into the sorted list) */
    int32_t *ed;
    int32_t len, cap;
} nbr_list;

static void nbr_push(nbr_list *L, int32_t idx, int32_t ed)
{
    if (L->len == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 8;
        L->idx = (int32_t *)realloc(L->idx, sizeof(int32_t) * (size_t)L->cap);
        L->ed = (int32_t *)realloc(L->ed, sizeof(int32_t) * (size_t)L->cap);
    }
    L->idx[L->len] = idx;
    L->ed[L->len] = ed;
    L->len++;
}

/*
 * get_nearest_neighbors (nearest_neighbor_graph.py:110-198).
 *   seqs/off      : the sorted list `seq_to_acc_list_sorted` (n entries)
 *   converged[i]  : 1 iff sequence i is in `has_converged`
 *   start,count   : query range [start, start+count)  (the reference's start_index / len(batch))
 *   depth         : neighbor_search_depth
 * Output (CSR over the `count` queries, insertion order of the inner dict):
 *   row_ptr[count+1], cols/eds (capacity cap).  Returns the number of edges, or -(needed) if cap is
 *   too small.  *n_calls receives the number of edlib_ed() calls the loop made.
 * The seed dictionary `lower_target_edit_distances` (:112,125-129,164-169,180-185) is reproduced.
 */
int64_t orc_nn_1set(const uint8_t *seqs, const int64_t *off, int32_t n, const uint8_t *converged,
                    int32_t start, int32_t count, int64_t depth,
                    int64_t *row_ptr, int32_t *cols, int32_t *eds, int64_t cap, int64_t *n_calls)
{
    int32_t *lower = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int32_t i = 0; i < n; ++i) lower[i] = -1; /* -1 = key absent */
    int64_t calls = 0, edges = 0;
    nbr_list L = {0, 0, 0, 0};
    row_ptr[0] = 0;
    for (int32_t i = start; i < start + count; ++i) {
        L.len = 0;
        const int32_t len1 = (int32_t)(off[i + 1] - off[i]);
        const uint8_t *s1 = seqs + off[i];
        if (!converged[i]) {
            int32_t best = lower[i] >= 0 ? lower[i] : len1;
            int stop_up = 0, stop_down = 0;
            int64_t j = 1;
            for (;;) {
                if (i - j < 0) stop_down = 1;
                if (i + j >= n) stop_up = 1;
                int32_t lo = -1, up = -1;
                if (!stop_down) {
                    lo = (int32_t)(i - j);
                    int32_t dl = len1 - (int32_t)(off[lo + 1] - off[lo]);
                    if (dl < 0) dl = -dl;
                    if (dl > best) stop_down = 1;
                }
                if (!stop_up) {
                    up = (int32_t)(i + j);
                    int32_t dl = len1 - (int32_t)(off[up + 1] - off[up]);
                    if (dl < 0) dl = -dl;
                    if (dl > best) stop_up = 1;
                }
                if (!stop_down) {
                    const int32_t e = orc_ed_bounded(s1, len1, seqs + off[lo], (int32_t)(off[lo + 1] - off[lo]), best);
                    ++calls;
                    if (0 < e && e < best) { best = e; L.len = 0; nbr_push(&L, lo, e); }
                    else if (e == best) nbr_push(&L, lo, e);
                    if (lower[lo] >= 0) { if (0 < e && e < lower[lo]) lower[lo] = e; }
                    else if (0 < e) lower[lo] = e;
                }
                if (!stop_up) {
                    const int32_t e = orc_ed_bounded(s1, len1, seqs + off[up], (int32_t)(off[up + 1] - off[up]), best);
                    ++calls;
                    if (0 < e && e < best) { best = e; L.len = 0; nbr_push(&L, up, e); }
                    else if (e == best) nbr_push(&L, up, e);
                    if (lower[up] >= 0) { if (0 < e && e < lower[up]) lower[up] = e; }
                    else if (0 < e) lower[up] = e;
                }
                if (stop_down && stop_up) break;
                if (j >= depth) break;
                ++j;
            }
        }
        for (int32_t x = 0; x < L.len; ++x) {
            if (edges < cap) { cols[edges] = L.idx[x]; eds[edges] = L.ed[x]; }
            ++edges;
        }
        row_ptr[i - start + 1] = edges;
    }
    free(L.idx);
    free(L.ed);
    free(lower);
    if (n_calls) *n_calls = calls;
    return edges <= cap ? edges : -edges;
}

/*
 * get_nearest_neighbors_2set (nearest_neighbor_graph.py:341-424).
 *   is_target[i] : 1 iff accession i is in `target_accessions` (such entries are skipped as queries and
 *                  get NO row: row_ptr is indexed by position in [start,start+count) and rows of targets
 *                  are marked with row_ptr[r+1] == row_ptr[r] and is_query_out[r] = 0).
 */
int64_t orc_nn_2set(const uint8_t *seqs, const int64_t *off, int32_t n, const uint8_t *is_target,
                    int32_t start, int32_t count, int64_t depth,
                    int64_t *row_ptr, int32_t *cols, int32_t *eds, int64_t cap, int64_t *n_calls)
{
    int64_t calls = 0, edges = 0;
    nbr_list L = {0, 0, 0, 0};
    row_ptr[0] = 0;
    for (int32_t i = start; i < start + count; ++i) {
        L.len = 0;
        if (!is_target[i]) {
            const int32_t len1 = (int32_t)(off[i + 1] - off[i]);
            const uint8_t *s1 = seqs + off[i];
            int32_t best = len1;
            int stop_up = 0, stop_down = 0;
            int64_t processed = 0;
            int64_t j = 1;
            for (;;) {
                if (i - j < 0) stop_down = 1;
                if (i + j >= n) stop_up = 1;
                int32_t lo = -1, up = -1;
                if (!stop_down) {
                    lo = (int32_t)(i - j);
                    int32_t dl = len1 - (int32_t)(off[lo + 1] - off[lo]);
                    if (dl < 0) dl = -dl;
                    if (dl > best) stop_down = 1;
                }
                if (!stop_up) {
                    up = (int32_t)(i + j);
                    int32_t dl = len1 - (int32_t)(off[up + 1] - off[up]);
                    if (dl < 0) dl = -dl;
                    if (dl > best) stop_up = 1;
                }
                if (!stop_down && is_target[lo]) {
                    ++processed;
                    const int32_t e = orc_ed_bounded(s1, len1, seqs + off[lo], (int32_t)(off[lo + 1] - off[lo]), best);
                    ++calls;
                    if (0 <= e && e < best) { best = e; L.len = 0; nbr_push(&L, lo, e); }
                    else if (e == best) nbr_push(&L, lo, e);
                }
                if (!stop_up && is_target[up]) {
                    ++processed;
                    const int32_t e = orc_ed_bounded(s1, len1, seqs + off[up], (int32_t)(off[up + 1] - off[up]), best);
                    ++calls;
                    if (0 <= e && e < best) { best = e; L.len = 0; nbr_push(&L, up, e); }
                    else if (e == best) nbr_push(&L, up, e);
                }
                if (stop_down && stop_up) break;
                if (processed >= depth) break;
                ++j;
            }
        }
        for (int32_t x = 0; x < L.len; ++x) {
            if (edges < cap) { cols[edges] = L.idx[x]; eds[edges] = L.ed[x]; }
            ++edges;
        }
        row_ptr[i - start + 1] = edges;
    }
    free(L.idx);
    free(L.ed);
    if (n_calls) *n_calls = calls;
    return edges <= cap ? edges : -edges;
}

/* ------------------------------------------------------------------------------------------------
 * 3. Semi-global affine alignment with traceback (parasail sg_trace_scan semantics)
 * ---------------------------------------------------------------------------------------------- */

/* trace byte layout (4 bits used) */
#define TR_H_MASK 3u
#define TR_H_DIAG 0u
#define TR_H_F    1u   /* vertical gap  : consumes a query (s1) base, CIGAR 'I' */
#define TR_H_E    2u   /* horizontal gap: consumes a ref   (s2) base, CIGAR 'D' */
#define TR_E_EXT  4u   /* E[i][j] came from E[i][j-1]-ext (else from H[i][j-1]-open) */
#define TR_F_EXT  8u   /* F[i][j] came from F[i-1][j]-ext (else from H[i-1][j]-open) */

/* policy bits (0 = parasail's believed behaviour):
 *   bit0: at H ties between the two gap states prefer E (horizontal) over F (vertical)   [default F]
 *   bit1: at open/extend ties inside a gap prefer "open"                                  [default extend]
 *   bit2: end cell: scan the last column before the last row                              [default row first]
 *   bit3: end cell: among equal maxima take the last instead of the first                 [default first]
 *   bit4: at H ties prefer a gap over the diagonal                                        [default diagonal]
 */
#define POL_E_BEFORE_F   1
#define POL_OPEN_ON_TIE  2
#define POL_COL_FIRST    4
#define POL_LAST_MAX     8
#define POL_GAP_FIRST   16

#define NEG_INF (-(1 << 29))

/*
 * s1 = query (rows i), s2 = reference (columns j), as in parasail.sg_trace_scan_16(s1, s2, open, ext, M)
 * (SW_alignment_module.py:66).  A gap of length g costs open + (g-1)*ext.  match > 0, mismatch <= 0.
 * Outputs:
 *   ops[]      : CIGAR as (len << 4 | code) with code 0 '=', 1 'X', 2 'I', 3 'D' ; maximal runs
 *   *n_ops     : number of ops written (if > ops_cap nothing past the cap is written)
 *   res[0..5]  : score, end_query, end_ref, matches, mismatches, indels  (counts as SW_alignment_module.py:79-81)
 * Returns 0, or -1 on allocation failure.
 */
int32_t orc_sg_trace(const uint8_t *s1, int32_t m, const uint8_t *s2, int32_t n,
                     int32_t match, int32_t mismatch, int32_t open, int32_t ext, int32_t policy,
                     uint32_t *ops, int64_t ops_cap, int64_t *n_ops, int32_t *res)
{
    *n_ops = 0;
    if (m == 0 || n == 0) {
        /* degenerate: everything is an end gap */
        int64_t k = 0;
        if (m > 0) { if (k < ops_cap) ops[k] = ((uint32_t)m << 4) | 2u; ++k; }
        if (n > 0) { if (k < ops_cap) ops[k] = ((uint32_t)n << 4) | 3u; ++k; }
        *n_ops = k;
        res[0] = 0; res[1] = m - 1; res[2] = n - 1; res[3] = 0; res[4] = 0; res[5] = m + n;
        return 0;
    }
    uint8_t *T = (uint8_t *)malloc((size_t)m * (size_t)n);
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1) * 2);
    int32_t *F = H + (n + 1);
    int32_t *lastcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    if (!T || !H || !lastcol) { free(T); free(H); free(lastcol); return -1; }
    for (int32_t j = 0; j <= n; ++j) { H[j] = 0; F[j] = NEG_INF; }
    for (int32_t i = 1; i <= m; ++i) {
        int32_t NH = H[0];  /* H[i-1][0] = 0 */
        int32_t WH = 0;     /* H[i][0]   = 0 */
        int32_t E = NEG_INF;
        H[0] = WH;
        const uint8_t a = s1[i - 1];
        uint8_t *Trow = T + (size_t)(i - 1) * n;
        for (int32_t j = 1; j <= n; ++j) {
            const int32_t NWH = NH;
            NH = H[j];
            const int32_t F_opn = NH - open, F_ext = F[j] - ext;
            const int32_t E_opn = WH - open, E_ext = E - ext;
            uint8_t tr = 0;
            int32_t Fv, Ev;
            if (policy & POL_OPEN_ON_TIE) {
                if (F_opn >= F_ext) Fv = F_opn; else { Fv = F_ext; tr |= TR_F_EXT; }
                if (E_opn >= E_ext) Ev = E_opn; else { Ev = E_ext; tr |= TR_E_EXT; }
            } else {
                if (F_opn > F_ext) Fv = F_opn; else { Fv = F_ext; tr |= TR_F_EXT; }
                if (E_opn > E_ext) Ev = E_opn; else { Ev = E_ext; tr |= TR_E_EXT; }
            }
            F[j] = Fv;
            E = Ev;
            const int32_t Hd = NWH + (a == s2[j - 1] ? match : mismatch);
            int32_t Hv = Hd;
            if (Fv > Hv) Hv = Fv;
            if (Ev > Hv) Hv = Ev;
            uint8_t src;
            const int gap_first = (policy & POL_GAP_FIRST) != 0;
            const int e_first = (policy & POL_E_BEFORE_F) != 0;
            if (!gap_first && Hd == Hv) src = TR_H_DIAG;
            else if (e_first) src = (Ev == Hv) ? TR_H_E : (Fv == Hv) ? TR_H_F : TR_H_DIAG;
            else src = (Fv == Hv) ? TR_H_F : (Ev == Hv) ? TR_H_E : TR_H_DIAG;
            Trow[j - 1] = tr | src;
            WH = Hv;
            H[j] = Hv;
        }
        lastcol[i - 1] = H[n];
    }
    /* end cell over the last row (query consumed) and the last column (ref consumed) */
    int32_t score = NEG_INF, eq = -1, er = -1;
    const int last = (policy & POL_LAST_MAX) != 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int do_row = ((policy & POL_COL_FIRST) != 0) ? (pass == 1) : (pass == 0);
        if (do_row) {
            for (int32_t j = 0; j < n; ++j) {
                const int32_t v = H[j + 1];
                if (v > score || (last && v == score)) { score = v; eq = m - 1; er = j; }
            }
        } else {
            for (int32_t i = 0; i < m; ++i) {
                const int32_t v = lastcol[i];
                if (v > score || (last && v == score)) { score = v; eq = i; er = n - 1; }
            }
        }
    }
    /* traceback, building the CIGAR back-to-front */
    int64_t cap = (int64_t)m + n + 4;
    uint8_t *rev = (uint8_t *)malloc((size_t)cap); /* one code per alignment column, reversed */
    int64_t L = 0;
    for (int32_t k = n - 1; k > er; --k) rev[L++] = 3; /* trailing ref bases: 'D' */
    for (int32_t k = m - 1; k > eq; --k) rev[L++] = 2; /* trailing query bases: 'I' */
    int32_t i = eq, j = er;
    int where = 0; /* 0 = H, 1 = F, 2 = E */
    int32_t nmatch = 0, nmis = 0;
    while (i >= 0 && j >= 0) {
        const uint8_t tr = T[(size_t)i * n + j];
        if (where == 0) {
            const uint8_t src = tr & TR_H_MASK;
            if (src == TR_H_DIAG) {
                if (s1[i] == s2[j]) { rev[L++] = 0; ++nmatch; } else { rev[L++] = 1; ++nmis; }
                --i; --j;
            } else where = (src == TR_H_F) ? 1 : 2;
        } else if (where == 1) {
            rev[L++] = 2; /* query base against a gap */
            where = (tr & TR_F_EXT) ? 1 : 0;
            --i;
        } else {
            rev[L++] = 3; /* ref base against a gap */
            where = (tr & TR_E_EXT) ? 2 : 0;
            --j;
        }
    }
    while (i >= 0) { rev[L++] = 2; --i; } /* leading query bases: 'I' */
    while (j >= 0) { rev[L++] = 3; --j; } /* leading ref bases: 'D' */
    /* run-length encode front-to-back */
    int64_t no = 0;
    int64_t p = L - 1;
    while (p >= 0) {
        const uint8_t code = rev[p];
        uint32_t run = 0;
        while (p >= 0 && rev[p] == code) { ++run; --p; }
        if (no < ops_cap) ops[no] = (run << 4) | code;
        ++no;
    }
    *n_ops = no;
    res[0] = score; res[1] = eq; res[2] = er;
    res[3] = nmatch; res[4] = nmis; res[5] = (int32_t)(L - nmatch - nmis);
    free(rev);
    free(T);
    free(H);
    free(lastcol);
    return 0;
}

/* Score-only variant with plain full tables, used to cross-check orc_sg_trace's rolling arrays. */
int32_t orc_sg_score(const uint8_t *s1, int32_t m, const uint8_t *s2, int32_t n,
                     int32_t match, int32_t mismatch, int32_t open, int32_t ext)
{
    if (m == 0 || n == 0) return 0;
    const size_t W = (size_t)n + 1;
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * W * ((size_t)m + 1) * 3);
    int32_t *E = H + W * ((size_t)m + 1);
    int32_t *F = E + W * ((size_t)m + 1);
    for (int32_t i = 0; i <= m; ++i)
        for (int32_t j = 0; j <= n; ++j) { H[i * W + j] = 0; E[i * W + j] = NEG_INF; F[i * W + j] = NEG_INF; }
    int32_t best = NEG_INF;
    for (int32_t i = 1; i <= m; ++i)
        for (int32_t j = 1; j <= n; ++j) {
            int32_t e1 = H[i * W + j - 1] - open, e2 = E[i * W + j - 1] - ext;
            int32_t f1 = H[(i - 1) * W + j] - open, f2 = F[(i - 1) * W + j] - ext;
            int32_t e = e1 > e2 ? e1 : e2, f = f1 > f2 ? f1 : f2;
            int32_t h = H[(i - 1) * W + j - 1] + (s1[i - 1] == s2[j - 1] ? match : mismatch);
            if (e > h) h = e;
            if (f > h) h = f;
            E[i * W + j] = e; F[i * W + j] = f; H[i * W + j] = h;
            if ((i == m || j == n) && h > best) best = h;
        }
    free(H);
    return best;
}

/* ------------------------------------------------------------------------------------------------
 * 5. Infix ("HW") edit distance with location and path -- edlib.align(q, t, mode="HW", task="path", k)
 *    as consumed at modules/end_invariant_functions.py:593-620 (edlib_traceback) and :661,:668 (get_all_NN).
 *    "PARITY UNPINNED": edlib's source is absent.  Restated from its published semantics (Sosic & Sikic 2017 and
 *    the library's documentation): the query is aligned globally, gaps before and after it in the target are free;
 *    editDistance = min over end columns; endLocations = every end column attaining it, ascending; the start
 *    location of an end location = the smallest start whose global distance to target[start..end] equals the
 *    optimum (edlib searches the reversed problem in prefix mode and keeps the LAST position found, "so that the
 *    alignment does not start with an insertion if it can start with a mismatch"); locations[0] is the pair for
 *    the first end location and the path is the global alignment of the query to that substring.  Which optimal
 *    path is reported is a tie decision: as everywhere in this oracle (tests/golden/shims/edlib.py, min_ed in
 *    isocon_amd/functions.py) the traceback starts at the end and prefers a query-only step ('I'), then a
 *    target-only step ('D'), then the diagonal.
 * ---------------------------------------------------------------------------------------------- */

/* out[0] = distance (-1 if > k, k < 0 = unbounded), out[1] = start, out[2] = end (0-based, inclusive; -1 when no hit).
 * Full matrix, O(n*m) ints: test sizes only. */
int32_t orc_hw_locate(const uint8_t *q, int32_t n, const uint8_t *t, int32_t m, int32_t k, int32_t *out)
{
    out[0] = -1; out[1] = -1; out[2] = -1;
    if (n <= 0 || m <= 0) return -1;
    const size_t W = (size_t)m + 1;
    int32_t *row = (int32_t *)malloc(sizeof(int32_t) * W * 2);
    int32_t *prev = row, *cur = row + W;
    for (int32_t j = 0; j <= m; ++j) prev[j] = 0;                       /* free start in the target */
    for (int32_t i = 1; i <= n; ++i) {
        cur[0] = i;
        for (int32_t j = 1; j <= m; ++j) {
            int32_t v = prev[j - 1] + (q[i - 1] != t[j - 1]);
            if (prev[j] + 1 < v) v = prev[j] + 1;
            if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
            cur[j] = v;
        }
        int32_t *x = prev; prev = cur; cur = x;
    }
    int32_t h = prev[1], end = 0;
    for (int32_t j = 2; j <= m; ++j) if (prev[j] < h) { h = prev[j]; end = j - 1; }     /* first minimum */
    if (k >= 0 && h > k) { free(row); return -1; }
    /* start: reversed query against reversed target[0..end], global in the query, prefix of the reversed target */
    const int32_t mb = end + 1;
    for (int32_t j = 0; j <= mb; ++j) prev[j] = j;
    for (int32_t i = 1; i <= n; ++i) {
        cur[0] = i;
        for (int32_t j = 1; j <= mb; ++j) {
            int32_t v = prev[j - 1] + (q[n - i] != t[end - (j - 1)]);
            if (prev[j] + 1 < v) v = prev[j] + 1;
            if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
            cur[j] = v;
        }
        int32_t *x = prev; prev = cur; cur = x;
    }
    int32_t plast = -1;
    for (int32_t j = 1; j <= mb; ++j) if (prev[j] == h) plast = j - 1;  /* last position with the optimum */
    free(row);
    if (plast < 0) return -2;
    out[0] = h; out[1] = end - plast; out[2] = end;
    return h;
}

/* Global unit-cost alignment path of q against t, run-length ops (len << 4 | code; 0 '=', 1 'X', 2 'I' query only,
 * 3 'D' target only) in forward order; returns the distance, *n_ops = number of ops (-1 if cap is too small). */
int32_t orc_nw_path(const uint8_t *q, int32_t n, const uint8_t *t, int32_t m, uint32_t *ops, int64_t cap, int64_t *n_ops)
{
    const size_t W = (size_t)m + 1;
    int32_t *D = (int32_t *)malloc(sizeof(int32_t) * W * ((size_t)n + 1));
    for (int32_t j = 0; j <= m; ++j) D[j] = j;
    for (int32_t i = 1; i <= n; ++i) {
        D[i * W] = i;
        for (int32_t j = 1; j <= m; ++j) {
            int32_t v = D[(i - 1) * W + j - 1] + (q[i - 1] != t[j - 1]);
            if (D[(i - 1) * W + j] + 1 < v) v = D[(i - 1) * W + j] + 1;
            if (D[i * W + j - 1] + 1 < v) v = D[i * W + j - 1] + 1;
            D[i * W + j] = v;
        }
    }
    const int32_t ed = D[(size_t)n * W + m];
    uint8_t *rev = (uint8_t *)malloc((size_t)n + m + 1);
    int64_t L = 0;
    int32_t i = n, j = m;
    while (i > 0 || j > 0) {
        if (i > 0 && D[(i - 1) * W + j] + 1 == D[i * W + j]) { rev[L++] = 2; --i; }
        else if (j > 0 && D[i * W + j - 1] + 1 == D[i * W + j]) { rev[L++] = 3; --j; }
        else { rev[L++] = q[i - 1] == t[j - 1] ? 0 : 1; --i; --j; }
    }
    int64_t no = 0;
    for (int64_t a = L - 1; a >= 0;) {
        int64_t b = a;
        while (b >= 0 && rev[b] == rev[a]) --b;
        if (no < cap) ops[no] = ((uint32_t)(a - b) << 4) | rev[a];
        ++no;
        a = b;
    }
    *n_ops = no <= cap ? no : -1;
    free(rev);
    free(D);
    return ed;
}

#ifdef __cplusplus
}
#endif
