// isocon_hip.hip -- C ABI of libisocon_hip.so (see include/isocon_hip.h).  Host orchestration + kernel launches.
// gfx950 only.  One process drives one GPU (isocon_init selects it).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <memory>
#include <thread>
#include <vector>

#include "common.hpp"
#include "ed_band.hpp"
#include "ed_full.hpp"
#include "qgram_mm.hpp"
#include "nn.hpp"
#include "nn_list.hpp"
#include "nn_filter.hpp"
#include "nn_finalize.hpp"
#include "nn_finalize_host.hpp"
#include "partition_host.hpp"
#include "sg.hpp"
#include "msa.hpp"
#include "msa_build.hpp"
#include "msa_batch.hpp"
#include "hw.hpp"
#include "hw_tiles.hpp"
#include "ed_lanes.hpp"
#include "ed_bytes.hpp"

namespace isocon {
thread_local std::string g_last_error;
}

using namespace isocon;

// ISOCON_DEBUG_VARIANT = "name[=value],name,...": the ONE environment switch behind which every diagnostic / A-B variant of the kernel
// selection sits (tests force fallbacks and superseded paths through it; DESIGN.md section 8 lists the names).  Unset -- production -- every
// input class has one code path.  variant(name): the name is listed; variant_value(name): its value ("" if listed without one), nullptr if
// not listed (the pointer is good until eight further calls on this thread).
static bool variant_lookup(const char *name, std::string *value)
{
    const char *e = getenv("ISOCON_DEBUG_VARIANT");
    if (!e) return false;
    const size_t ln = strlen(name);
    for (const char *p = e; *p;) {
        const char *q = strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len >= ln && strncmp(p, name, ln) == 0 && (len == ln || p[ln] == '=')) {
            if (value) *value = len > ln ? std::string(p + ln + 1, len - ln - 1) : std::string();
            return true;
        }
        p += len + (q ? 1 : 0);
    }
    return false;
}
static bool variant(const char *name) { return variant_lookup(name, nullptr); }
static const char *variant_value(const char *name)
{
    thread_local std::string ring[8];
    thread_local unsigned at = 0;
    std::string &slot = ring[at++ % 8];
    return variant_lookup(name, &slot) ? slot.c_str() : nullptr;
}

// What the pool's nearest-neighbour slots currently hold, next to the slots themselves (no free-floating state: a store reaches it
// through its pool, and whoever overwrites or releases a slot invalidates the tag in the same place).
// BoundTag: the q-gram bound matrix of the last seed phase (slots SLOT_NN_LB / SLOT_NN_LBROW): the main phase of the SAME store and
// shard that follows uses it instead of computing it again.  Identified by the store's serial number (not its address).
struct BoundTag {
    bool valid = false, has_order = false;
    uint64_t serial = 0;
    QMap map = {0, 0, 0, 0};
    uint32_t depth = 0;
    int32_t kcap = 0;
    unsigned long long lb_total = 0;
};
// MsaTag: the multi-alignment matrix isocon_msa_build_ops left in SLOT_MSA_IN for isocon_msa_correct_built.
struct MsaTag { bool valid = false; uint64_t serial = 0; uint32_t n_rows = 0, n_cols = 0; };
// MsaBatchHost: the matrices isocon_msa_build_ops_batch left on the device for isocon_msa_correct_built_batch, with the host's copy of their layout.
struct MsaBatchHost {
    bool valid = false;
    uint64_t serial = 0;
    uint32_t n_parts = 0, n_rows = 0;
    std::vector<uint32_t> ncols, col_base, first_row;
    std::vector<unsigned long long> m_off;
};
// HeldHits: candidate edges of a sharded search that stay in device memory from phase to phase (SLOT_NN_ACC_HITS), tagged with their store.
struct HeldHits { uint64_t store_serial = 0; uint64_t rows = 0; };

// Grow-only device scratch owned by the store: repeated calls reuse their buffers instead of paying
// hipMalloc/hipFree (tens of ms for the multi-GB trace scratch) every time.
struct ScratchPool {
    struct Slot { void *p = nullptr; size_t cap = 0; };
    Slot slots[160];
    BoundTag bound_tag;
    HeldHits held_hits;
    MsaTag msa_tag;
    MsaBatchHost msab;
    void *get(int idx, size_t bytes)
    {
        Slot &s = slots[idx];
        if (bytes == 0) bytes = 16;
        if (s.cap < bytes) {
            if (s.p) (void)hipFree(s.p);
            s.p = nullptr; s.cap = 0;
            const size_t want = bytes + bytes / 8;
            if (hipMalloc(&s.p, want) == hipSuccess) s.cap = want;
            else if (hipMalloc(&s.p, bytes) == hipSuccess) s.cap = bytes;
            else { s.p = nullptr; (void)hipGetLastError(); }
        }
        return s.p;
    }
    void release()
    {
        for (Slot &s : slots) { if (s.p) (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
        bound_tag = BoundTag();
        held_hits = HeldHits();
        msa_tag = MsaTag();
        msab = MsaBatchHost();
    }
};

enum {
    SLOT_ED_TS = 0, SLOT_ED_IDS, SLOT_ED_K, SLOT_ED_OUT, SLOT_FULL_A, SLOT_FULL_B, SLOT_FULL_K, SLOT_FULL_OUT,
    SLOT_NN_BEST, SLOT_NN_QF, SLOT_NN_TF, SLOT_NN_HITS, SLOT_NN_HITCOUNT, SLOT_NN_STATS, SLOT_NN_TS, SLOT_NN_IDS, SLOT_NN_PLANES2, SLOT_NN_PERM, SLOT_NN_IL, SLOT_NN_IL2, SLOT_NN_HITS2, SLOT_NN_HITCOUNT2, SLOT_NN_QPROF, SLOT_NN_QSUM, SLOT_NN_LB, SLOT_NN_LBROW, SLOT_NN_LBLEN, SLOT_NN_SLOTORDER, SLOT_NN_LBCHUNKS, SLOT_NN_ROWMIN, SLOT_NN_COLMIN, SLOT_NN_SEED_A, SLOT_NN_SEED_B, SLOT_NN_SEED_N, SLOT_NN_LBT, SLOT_NN_LBT_OFF, SLOT_NN_LBT_SLO, SLOT_NN_LBT_LEN, SLOT_NN_LBT_PAD, SLOT_NN_SCORE, SLOT_NN_LDEST, SLOT_NN_FIN_HITS, SLOT_NN_FIN_CNT, SLOT_NN_FIN_START, SLOT_NN_FIN_CUR, SLOT_NN_FIN_NB, SLOT_NN_FIN_LEN2, SLOT_NN_FIN_ROWPTR, SLOT_NN_FIN_COLS, SLOT_NN_FIN_FLAG, SLOT_NN_FIN_BEST, SLOT_NN_ACC_HITS, SLOT_NN_LTOT, SLOT_NN_LCHUNKS, SLOT_NN_LIST, SLOT_NN_LPA, SLOT_NN_LPB, SLOT_NN_TEXT2, SLOT_NN_LTASKS,
    SLOT_SG_PAIRS, SLOT_SG_R, SLOT_SG_TRACE, SLOT_SG_END, SLOT_SG_OPS, SLOT_SG_CNT, SLOT_SG_RES, SLOT_SG_OFF, SLOT_SG_DENSE, SLOT_SG_BOUND, SLOT_SG_AOFF, SLOT_SG_ALNA, SLOT_SG_ALNB,
    SLOT_MSA_IN, SLOT_MSA_OUT, SLOT_MSA_DEG, SLOT_MSA_COUNTS, SLOT_MSA_MAJ, SLOT_MSA_FLAGS, SLOT_MSA_TOT, SLOT_MSA_NCAND, SLOT_MSA_LEN, SLOT_MSA_OFF, SLOT_MSA_PACKED, SLOT_MSA_ROWS, SLOT_MSA_OPS, SLOT_MSA_OPTR, SLOT_MSA_LONGEST, SLOT_MSA_WIDTH, SLOT_MSA_CSLOT, SLOT_MSA_LTOT, SLOT_MSA_WIDE, SLOT_MSA_PROW, SLOT_MSA_PCOL, SLOT_MSA_PPTR, SLOT_MSA_PBYTES, SLOT_MSAB_PART, SLOT_MSAB_FIRST, SLOT_MSAB_LM, SLOT_MSAB_SBASE, SLOT_MSAB_NCOLS, SLOT_MSAB_MOFF, SLOT_MSAB_CBASE, SLOT_MSAB_CBP, SLOT_MSAB_CBC, SLOT_MSAB_CBR,
    SLOT_HW_Q, SLOT_HW_T, SLOT_HW_K, SLOT_HW_OUT, SLOT_HW_TRACE, SLOT_HW_CTR, SLOT_HW_TILEQ, SLOT_HW_LANES, SLOT_HW_PQ, SLOT_HW_KEY, SLOT_HW_HIST, SLOT_HW_CURSOR, SLOT_HW_TBASE, SLOT_HW_CLS,
    SLOT_PACK_ASCII, SLOT_PACK_OFF, SLOT_PACK_BAD, SLOT_PACK_HIST, SLOT_PACK_FLAGS, SLOT_EB_A, SLOT_EB_B, SLOT_EB_K, SLOT_EB_OUT, SLOT_EB_ROWS, SLOT_SCAN_TMP, SLOT_SCAN_SUMS,
    SLOT_COUNT
};
static_assert(SLOT_COUNT <= 160, "ScratchPool::slots too small");

// One pool per process (one process drives one GPU): scratch outlives the individual stores, because the Python
// wrappers create a fresh store per call (the reference's functions are stateless).
static ScratchPool g_scratch;

static uint64_t g_store_serial = 0;

struct isocon_store {
    uint64_t serial = ++g_store_serial;
    DevStore dev;
    // isocon_store_create_ptrs_ex(..., ISOCON_STORE_PRIVATE_SCRATCH): the store gets a scratch pool of its own, released with it --
    // for runs that emulate SEVERAL ranks inside one process (tests/baton_dist.py): a rank's bound matrix, held candidate edges and
    // counters must not be another rank's.  A real rank is a process, and its stores share the process' pool.
    explicit isocon_store(bool private_scratch = false) : own_pool(private_scratch ? new ScratchPool() : nullptr), pool(own_pool ? *own_pool : g_scratch) {}
    std::unique_ptr<ScratchPool> own_pool;
    ScratchPool &pool;
    std::vector<int32_t> lens;   // host copy
    uint64_t device_bytes = 0;
    uint64_t *d_planes = nullptr;
    // The four symbols of the set, code 0 .. 3.  "ACGT" for every shipped data set; a set over another alphabet of at most four
    // symbols (lower case, RNA) is packed under its own map: distances and the NN graph only compare symbols for equality, exactly
    // what edlib does with whatever characters it is given (EAM:111, NNG:105).  The entry points that emit letters or score with the
    // reference's "ACGT" matrix (alignments, consensus) refuse such a store (ISOCON_E_ALPHABET).
    char alphabet[4] = {'A', 'C', 'G', 'T'};
    bool acgt = true;
    // More than four distinct symbols (ACGT + N, mixed case ...): the planes hold the "ACGT" map with code 0 at every other byte, the
    // bytes themselves stay on the device and exc[i] marks the sequences that hold such a byte ("exceptional").  A pair with an
    // exceptional sequence is aligned by k_ed_bytes (ed_bytes.hpp) on the bytes; the bit-vector kernels never see it.
    uint8_t *d_bytes = nullptr;
    uint64_t *d_boff = nullptr;
    uint64_t bytes_base = 0;
    std::vector<uint8_t> exc;
    std::vector<uint32_t> exc_count;          // such bytes per sequence
    uint32_t n_exc = 0;
    bool is_exc(uint32_t i) const { return n_exc != 0 && exc[i] != 0; }
    int32_t *d_lens = nullptr;
    int32_t maxlen = 0;
};

namespace {

// A device buffer: pooled (slot of the store's ScratchPool) when constructed with a pool, private otherwise.
// Pinned staging for host<->device copies.  hipMemcpy on pageable memory pins the user pages on the fly and releases
// them lazily: measured 17-27 ms showing up at the NEXT synchronisation after a 7 MB hit-list download.  One pinned
// buffer per process, copies go through it in pieces.
struct PinnedStage {
    void *p = nullptr;
    size_t cap = 0;
    void *get(size_t bytes)
    {
        if (cap < bytes) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) cap = bytes;
            else { p = nullptr; (void)hipGetLastError(); }
        }
        return p;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
static PinnedStage g_stage;
static constexpr size_t kStageBytes = (size_t)32 << 20;
static constexpr uint64_t kPackWaves = (uint64_t)1 << 25;          // wavefronts per packing launch (one per 64 bases of a sequence): below 2^32 threads
static constexpr size_t kStageMin = (size_t)32 << 10;      // smaller copies: the runtime's own staging path is fine

// Host buffers handed out by isocon_host_alloc are pinned: copies to / from them need no staging.
static std::vector<std::pair<const char *, size_t>> g_pinned;
static bool is_pinned(const void *p, size_t bytes)
{
    for (const auto &r : g_pinned)
        if ((const char *)p >= r.first && (const char *)p + bytes <= r.first + r.second) return true;
    return false;
}

static hipError_t copy_d2h(void *dst, const void *dsrc, size_t bytes)
{
    if (bytes < kStageMin || is_pinned(dst, bytes)) return hipMemcpy(dst, dsrc, bytes, hipMemcpyDeviceToHost);
    char *st = static_cast<char *>(g_stage.get(kStageBytes));
    if (!st) return hipMemcpy(dst, dsrc, bytes, hipMemcpyDeviceToHost);
    for (size_t off = 0; off < bytes; off += kStageBytes) {
        const size_t len = std::min(kStageBytes, bytes - off);
        const hipError_t e = hipMemcpy(st, static_cast<const char *>(dsrc) + off, len, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        memcpy(static_cast<char *>(dst) + off, st, len);
    }
    return hipSuccess;
}

static hipError_t copy_h2d(void *ddst, const void *src, size_t bytes)
{
    if (bytes < kStageMin || is_pinned(src, bytes)) return hipMemcpy(ddst, src, bytes, hipMemcpyHostToDevice);
    char *st = static_cast<char *>(g_stage.get(kStageBytes));
    if (!st) return hipMemcpy(ddst, src, bytes, hipMemcpyHostToDevice);
    for (size_t off = 0; off < bytes; off += kStageBytes) {
        const size_t len = std::min(kStageBytes, bytes - off);
        memcpy(st, static_cast<const char *>(src) + off, len);
        const hipError_t e = hipMemcpy(static_cast<char *>(ddst) + off, st, len, hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

struct DevBuf {
    void *p = nullptr;
    ScratchPool *pool = nullptr;
    int slot = -1;
    DevBuf() {}
    DevBuf(ScratchPool *pl, int sl) : pool(pl), slot(sl) {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p && !pool) (void)hipFree(p); }
    template <class T> T *as() { return static_cast<T *>(p); }
    int alloc(size_t bytes)
    {
        if (pool) {
            p = pool->get(slot, bytes);
            if (!p) { g_last_error = "hipMalloc(scratch slot " + std::to_string(slot) + ", " + std::to_string(bytes) + " B) failed"; return ISOCON_E_HIP; }
            return ISOCON_OK;
        }
        if (p) { (void)hipFree(p); p = nullptr; }
        ISO_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
        return ISOCON_OK;
    }
};

// out[0 .. n] = exclusive prefix sums of f(in[i]) (common.hpp: k_scan_tiles + k_scan_finish), on the null stream
template <int SHIFT, class OUT>
static int device_exscan(ScratchPool *pl, const uint32_t *d_in, uint32_t n, OUT *d_out)
{
    DevBuf d_tmp(pl, SLOT_SCAN_TMP), d_sums(pl, SLOT_SCAN_SUMS);
    const uint32_t tiles = std::max<uint32_t>(1, (n + SCAN_TILE - 1) / SCAN_TILE);
    int rc;
    if ((rc = d_tmp.alloc((size_t)std::max<uint32_t>(n, 1) * 4)) || (rc = d_sums.alloc((size_t)tiles * 8))) return rc;
    hipLaunchKernelGGL((k_scan_tiles<SHIFT>), dim3(tiles), dim3(256), 0, 0, d_in, n, d_tmp.as<uint32_t>(), d_sums.as<unsigned long long>());
    hipLaunchKernelGGL((k_scan_finish<OUT>), dim3(tiles), dim3(256), 0, 0, d_tmp.as<uint32_t>(), n, d_sums.as<unsigned long long>(), tiles, d_out, (unsigned long long *)nullptr);
    ISO_HIP_CHECK(hipGetLastError());
    return ISOCON_OK;
}

struct EventTimer {
    hipEvent_t e0 = nullptr, e1 = nullptr, em = nullptr;
    float total = 0.f;
    float marked_total = 0.f;      // time between a mark() and the stop() that follows it, summed (a second launch inside one start / stop pair,
    bool marked = false;           // timed without a host synchronisation between the two launches)
    bool ok = false;
    EventTimer() { ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventCreate(&em) == hipSuccess; }
    ~EventTimer() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); if (em) (void)hipEventDestroy(em); }
    void start() { if (ok) (void)hipEventRecord(e0, 0); marked = false; }
    void mark() { if (ok) { (void)hipEventRecord(em, 0); marked = true; } }
    float stop()
    {
        float ms = 0.f;
        if (ok) {
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            if (marked) { float part = 0.f; (void)hipEventElapsedTime(&part, em, e1); marked_total += part; marked = false; }
        }
        total += ms;
        return ms;
    }
};

// Packs ASCII sequences into the store's bit-planes: one wavefront per (sequence, 64-base chunk) -- a coalesced 64-byte load,
// the two code bits of every base become the chunk's two words through two ballots.  *first_bad receives the smallest
// (sequence << 32 | position) holding a symbol outside ACGT.
struct PackMap { uint8_t sym[4]; };

// how often every byte value occurs in the uploaded sequences (only looked at when a symbol outside ACGT turned up)
__global__ __launch_bounds__(256) void k_byte_histogram(const uint8_t *__restrict__ ascii, uint64_t total, unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) atomicAdd(&h[ascii[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(hist + threadIdx.x, (unsigned long long)h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_pack_planes(const uint8_t *__restrict__ ascii, const uint64_t *__restrict__ offsets, uint64_t base, uint32_t n,
                                                      uint32_t nchunks, uint64_t *__restrict__ planes, unsigned long long *__restrict__ first_bad, PackMap map,
                                                      uint32_t fold, uint64_t w0)
{
    const uint64_t w = w0 + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);          // (a launch holds fewer than 2^32 threads: w0 = first wavefront of this one)
    if (w >= (uint64_t)nchunks * n) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t chunk = (uint32_t)(w / n), seq = (uint32_t)(w % n);           // consecutive waves: consecutive sequences
    const uint64_t off = offsets[seq] - base, len = offsets[seq + 1] - offsets[seq];
    const uint64_t pos = (uint64_t)chunk * 64 + lane;
    int cd = 0;
    bool bad = false;
    if (pos < len) {
        uint8_t ch = ascii[off + pos];
        if (fold && ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 32);          // fold: lower case takes its upper-case letter's code, any other byte code 0, nothing is reported
        cd = ch == map.sym[0] ? 0 : ch == map.sym[1] ? 1 : ch == map.sym[2] ? 2 : ch == map.sym[3] ? 3 : -1;
        bad = cd < 0 && !fold;
    }
    const unsigned long long lo = __ballot(!bad && (cd & 1)), hi = __ballot(!bad && (cd & 2)), bm = __ballot(bad);
    if (lane == 0) {
        planes[((size_t)chunk * n + seq) * 2] = lo;
        planes[((size_t)chunk * n + seq) * 2 + 1] = hi;
        if (bm) atomicMin(first_bad, ((unsigned long long)seq << 32) | ((unsigned long long)chunk * 64 + (unsigned long long)(__ffsll((long long)bm) - 1)));
    }
}

}  // namespace

extern "C" {

const char *isocon_strerror(int status)
{
    switch (status) {
    case ISOCON_OK: return "ok";
    case ISOCON_E_ARG: return "bad argument";
    case ISOCON_E_ALPHABET: return "sequence contains a symbol outside ACGT";
    case ISOCON_E_HIP: return "HIP runtime error";
    case ISOCON_E_CAPACITY: return "output buffer too small";
    case ISOCON_E_NODEVICE: return "no usable GPU";
    case ISOCON_E_UNSUPPORTED: return "unsupported request";
    default: return "unknown status";
    }
}

const char *isocon_last_error(void) { return g_last_error.c_str(); }

int isocon_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void isocon_release_scratch(void) { g_scratch.release(); g_stage.release(); }

int isocon_init(int device_ordinal)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_last_error = "hipGetDeviceCount: no device"; return ISOCON_E_NODEVICE; }
    if (device_ordinal < 0 || device_ordinal >= n) return ISOCON_E_ARG;
    ISO_HIP_CHECK(hipSetDevice(device_ordinal));
    return ISOCON_OK;
}

// ascii != nullptr: one contiguous buffer addressed by offsets; else the sequences lie at seq_ptrs[i] (offsets still hold the prefix
// sums of their lengths) and are gathered into the two halves of the pinned staging buffer, one half on its way to the device
// while the other is being filled.
static int store_create_impl(const uint8_t *ascii, const uint8_t *const *seq_ptrs, const uint64_t *offsets, uint32_t n, isocon_store **out, bool private_scratch = false)
{
    *out = nullptr;
    int32_t maxlen = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0x3fffffff) return ISOCON_E_ARG;
        maxlen = std::max<int32_t>(maxlen, (int32_t)(offsets[i + 1] - offsets[i]));
    }
    const uint32_t nchunks = (uint32_t)((maxlen + 63) / 64 + 1);
    const uint32_t nn = std::max<uint32_t>(n, 1);
    std::vector<int32_t> lens(nn, 0);
    for (uint32_t i = 0; i < n; ++i) lens[i] = (int32_t)(offsets[i + 1] - offsets[i]);
    const uint64_t base = n ? offsets[0] : 0, total = n ? offsets[n] - offsets[0] : 0;
    isocon_store *st = new isocon_store(private_scratch);
    st->lens = lens;
    st->maxlen = maxlen;
    const size_t pbytes = (size_t)nchunks * nn * 2 * sizeof(uint64_t), lbytes = lens.size() * sizeof(int32_t);
    if (hipMalloc((void **)&st->d_planes, pbytes) != hipSuccess || hipMalloc((void **)&st->d_lens, lbytes) != hipSuccess) {
        g_last_error = "hipMalloc(store) failed";
        isocon_store_destroy(st);
        return ISOCON_E_HIP;
    }
    // The bytes go to the device as they are and are packed there (k_pack_planes: one wavefront per 64 bases, two ballots);
    // the host only checked the offsets.  Scratch (ASCII, offsets, first bad position) comes from the process-wide pool.
    {
        DevBuf d_ascii(&g_scratch, SLOT_PACK_ASCII), d_off(&g_scratch, SLOT_PACK_OFF), d_bad(&g_scratch, SLOT_PACK_BAD);
        uint64_t pack_waves = kPackWaves;
        if (const char *e = variant_value("pack_waves")) pack_waves = std::max<uint64_t>(4, strtoull(e, nullptr, 10)) & ~(uint64_t)3;      // tests: many launches on a small set
        const unsigned long long none = ~0ull;
        int rc = ISOCON_OK;
        if ((rc = d_ascii.alloc(total ? total : 16)) || (rc = d_off.alloc((size_t)(nn + 1) * 8)) || (rc = d_bad.alloc(8))) { isocon_store_destroy(st); return rc; }
        bool up = true;
        if (total && ascii) up = copy_h2d(d_ascii.p, ascii + base, total) == hipSuccess;
        else if (total) {
            char *stg = static_cast<char *>(g_stage.get(kStageBytes));
            hipEvent_t ev[2] = {nullptr, nullptr};
            up = stg != nullptr && hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
            const size_t half = kStageBytes / 2;
            uint64_t done = 0;           // bytes of the concatenation already sent
            bool busy[2] = {false, false};
            // bytes [b0, b1) of the concatenation -> dst (offsets[] are the sequences' prefix sums, relative to `base`)
            auto gather = [&](char *dst, uint64_t b0, uint64_t b1) {
                uint32_t i = (uint32_t)(std::upper_bound(offsets, offsets + n + 1, base + b0) - offsets) - 1;
                uint64_t in_seq = base + b0 - offsets[i];
                while (b0 < b1 && i < n) {
                    const uint64_t len = offsets[i + 1] - offsets[i];
                    const size_t take = (size_t)std::min<uint64_t>(len - in_seq, b1 - b0);
                    if (take) memcpy(dst, seq_ptrs[i] + in_seq, take);
                    dst += take; b0 += take; in_seq += take;
                    if (in_seq == len) { ++i; in_seq = 0; }
                }
            };
            const unsigned hc = std::thread::hardware_concurrency();
            const unsigned n_thr = total >= ((uint64_t)8 << 20) ? std::min(4u, hc ? hc : 1u) : 1u;      // one thread copies ~10 GB/s: 125 MB at C3
            for (int h = 0; up && done < total; h ^= 1) {
                if (busy[h]) up = hipEventSynchronize(ev[h]) == hipSuccess;
                const size_t fill = (size_t)std::min<uint64_t>(half, total - done);
                if (n_thr <= 1) gather(stg + h * half, done, done + fill);
                else {
                    std::thread th[4];
                    const size_t part = (fill + n_thr - 1) / n_thr;
                    for (unsigned t = 0; t < n_thr; ++t) {
                        const size_t a = std::min(fill, t * part), b = std::min(fill, (t + 1) * part);
                        th[t] = std::thread([&, a, b, h] { if (b > a) gather(stg + h * half + a, done + a, done + b); });
                    }
                    for (unsigned t = 0; t < n_thr; ++t) th[t].join();
                }
                up = up && hipMemcpyAsync(static_cast<char *>(d_ascii.p) + done, stg + h * half, fill, hipMemcpyHostToDevice, 0) == hipSuccess &&
                     hipEventRecord(ev[h], 0) == hipSuccess;
                busy[h] = true;
                done += fill;
            }
            up = up && hipStreamSynchronize(0) == hipSuccess;
            if (ev[0]) (void)hipEventDestroy(ev[0]);
            if (ev[1]) (void)hipEventDestroy(ev[1]);
        }
        bool ok = up &&
                  (n == 0 || copy_h2d(d_off.p, offsets, (size_t)(n + 1) * 8) == hipSuccess) &&
                  hipMemcpy(d_bad.p, &none, 8, hipMemcpyHostToDevice) == hipSuccess && copy_h2d(st->d_lens, lens.data(), lbytes) == hipSuccess;
        unsigned long long bad = none;
        if (ok) {
            if (n) {
                const uint64_t waves = (uint64_t)nchunks * n;
                for (uint64_t w0 = 0; w0 < waves; w0 += pack_waves)
                    hipLaunchKernelGGL(k_pack_planes, dim3((unsigned)((std::min<uint64_t>(pack_waves, waves - w0) + 3) / 4)), dim3(256), 0, 0, d_ascii.as<uint8_t>(), d_off.as<uint64_t>(), base, n, nchunks,
                                       st->d_planes, d_bad.as<unsigned long long>(), PackMap{{'A', 'C', 'G', 'T'}}, 0u, w0);
            } else {
                ok = hipMemset(st->d_planes, 0, pbytes) == hipSuccess;
            }
            ok = ok && hipGetLastError() == hipSuccess && hipMemcpy(&bad, d_bad.p, 8, hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (!ok) {
            g_last_error = "isocon_store_create: device copy / packing failed";
            (void)hipGetLastError();
            isocon_store_destroy(st);
            return ISOCON_E_HIP;
        }
        if (bad != none) {
            // A symbol outside ACGT.  edlib takes any characters (EAM:111, NNG:105): if the whole set uses at most four distinct symbols
            // (lower case, RNA ...) it is packed again under its own map -- A, C, G, T keep their codes if present, the other symbols take
            // the free codes in byte order -- and serves the distance / NN entry points; more than four symbols cannot be held in 2 bits.
            DevBuf d_hist(&g_scratch, SLOT_PACK_HIST);
            unsigned long long hist[256];
            bool hok = d_hist.alloc(256 * 8) == ISOCON_OK && hipMemset(d_hist.p, 0, 256 * 8) == hipSuccess;
            if (hok) {
                hipLaunchKernelGGL(k_byte_histogram, dim3(1024), dim3(256), 0, 0, d_ascii.as<uint8_t>(), (uint64_t)total, d_hist.as<unsigned long long>());
                hok = hipGetLastError() == hipSuccess && hipMemcpy(hist, d_hist.p, sizeof(hist), hipMemcpyDeviceToHost) == hipSuccess;
            }
            int distinct = 0;
            for (int c = 0; hok && c < 256; ++c) distinct += hist[c] != 0;
            if (!hok) {
                g_last_error = "isocon_store_create: symbol histogram failed";
                (void)hipGetLastError();
                isocon_store_destroy(st);
                return ISOCON_E_HIP;
            }
            if (distinct > 4) {
                // Planes: the "ACGT" map with lower case folded onto upper case and code 0 at every other byte.  For the pairs of ordinary
                // sequences they are what they always are; for a sequence with other bytes they hold its image under a map that MERGES symbol
                // classes, and merging classes can only turn mismatches into matches: d(image x, image y) <= d(x, y), so every lower bound
                // computed on the planes (q-gram bounds) holds for the sequences themselves.
                // The bytes and their offsets move from the scratch pool into the store, with one count per sequence.
                DevBuf d_flags(&g_scratch, SLOT_PACK_FLAGS);
                st->exc.assign(nn, 0);
                st->exc_count.assign(nn, 0);
                bool eok = d_flags.alloc((size_t)nn * 4) == ISOCON_OK && hipMemset(d_flags.p, 0, (size_t)nn * 4) == hipSuccess &&
                           hipMalloc((void **)&st->d_bytes, total ? total : 16) == hipSuccess && hipMalloc((void **)&st->d_boff, (size_t)(nn + 1) * 8) == hipSuccess;
                if (eok) {
                    const uint64_t waves = (uint64_t)nchunks * n;
                    for (uint64_t w0 = 0; w0 < waves; w0 += pack_waves)
                    hipLaunchKernelGGL(k_pack_planes, dim3((unsigned)((std::min<uint64_t>(pack_waves, waves - w0) + 3) / 4)), dim3(256), 0, 0, d_ascii.as<uint8_t>(), d_off.as<uint64_t>(), base, n, nchunks,
                                       st->d_planes, d_bad.as<unsigned long long>(), PackMap{{'A', 'C', 'G', 'T'}}, 1u, w0);
                    for (uint64_t w0 = 0; w0 < waves; w0 += pack_waves)
                        hipLaunchKernelGGL(k_exception_flags, dim3((unsigned)((std::min<uint64_t>(pack_waves, waves - w0) + 3) / 4)), dim3(256), 0, 0, d_ascii.as<uint8_t>(), d_off.as<uint64_t>(), base, n,
                                           nchunks, d_flags.as<uint32_t>(), (uint32_t)'A' | ((uint32_t)'C' << 8) | ((uint32_t)'G' << 16) | ((uint32_t)'T' << 24), w0);
                    eok = hipGetLastError() == hipSuccess && hipMemcpy(st->exc_count.data(), d_flags.p, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                          hipMemcpy(st->d_bytes, d_ascii.p, total, hipMemcpyDeviceToDevice) == hipSuccess &&
                          hipMemcpy(st->d_boff, d_off.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToDevice) == hipSuccess;
                }
                if (!eok) {
                    g_last_error = "isocon_store_create: keeping the bytes of a set with more than four symbols failed";
                    (void)hipGetLastError();
                    isocon_store_destroy(st);
                    return ISOCON_E_HIP;
                }
                st->bytes_base = base;
                for (uint32_t i = 0; i < n; ++i) { st->exc[i] = st->exc_count[i] != 0; st->n_exc += st->exc[i]; }
                st->acgt = false;
                st->device_bytes += total + (size_t)(nn + 1) * 8;
            } else {
            PackMap map{{0, 0, 0, 0}};
            bool used[4] = {false, false, false, false};
            const char acgt_sym[4] = {'A', 'C', 'G', 'T'};
            for (int k = 0; k < 4; ++k) if (hist[(uint8_t)acgt_sym[k]]) { map.sym[k] = (uint8_t)acgt_sym[k]; used[k] = true; }
            for (int c = 0; c < 256; ++c) {
                if (!hist[c] || c == 'A' || c == 'C' || c == 'G' || c == 'T') continue;
                for (int k = 0; k < 4; ++k) if (!used[k]) { map.sym[k] = (uint8_t)c; used[k] = true; break; }
            }
            // (free codes keep byte 0, which no sequence holds: C strings)  -- a 0 byte in the input would be one of the four symbols
            for (int k = 0; k < 4; ++k) if (!used[k]) { for (int c = 1; c < 256; ++c) if (!hist[c] && c != map.sym[0] && c != map.sym[1] && c != map.sym[2] && c != map.sym[3]) { map.sym[k] = (uint8_t)c; break; } }
            bool rok = hipMemcpy(d_bad.p, &none, 8, hipMemcpyHostToDevice) == hipSuccess;
            if (rok) {
                const uint64_t waves = (uint64_t)nchunks * n;
                for (uint64_t w0 = 0; w0 < waves; w0 += pack_waves)
                    hipLaunchKernelGGL(k_pack_planes, dim3((unsigned)((std::min<uint64_t>(pack_waves, waves - w0) + 3) / 4)), dim3(256), 0, 0, d_ascii.as<uint8_t>(), d_off.as<uint64_t>(), base, n, nchunks,
                                       st->d_planes, d_bad.as<unsigned long long>(), map, 0u, w0);
                rok = hipGetLastError() == hipSuccess && hipMemcpy(&bad, d_bad.p, 8, hipMemcpyDeviceToHost) == hipSuccess && bad == none;
            }
            if (!rok) {
                g_last_error = "isocon_store_create: packing under the set's own alphabet failed";
                (void)hipGetLastError();
                isocon_store_destroy(st);
                return ISOCON_E_HIP;
            }
            for (int k = 0; k < 4; ++k) st->alphabet[k] = (char)map.sym[k];
            st->acgt = false;
            }
        }
    }
    st->device_bytes += pbytes + lbytes;
    st->dev.planes = st->d_planes;
    st->dev.il = nullptr;
    st->dev.lens = st->d_lens;
    st->dev.n = n;
    st->dev.nchunks = nchunks;
    st->lens.resize(n);
    *out = st;
    return ISOCON_OK;
}

int isocon_store_create(const uint8_t *ascii, const uint64_t *offsets, uint32_t n, isocon_store **out)
{
    if (!out || (!ascii && n) || !offsets) return ISOCON_E_ARG;
    return store_create_impl(ascii, nullptr, offsets, n, out);
}

int isocon_store_create_ptrs(const uint8_t *const *seq_ptrs, const uint64_t *seq_lens, uint32_t n, isocon_store **out)
{
    if (!out || (n && (!seq_ptrs || !seq_lens))) return ISOCON_E_ARG;
    std::vector<uint64_t> offsets((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (seq_lens[i] && !seq_ptrs[i]) return ISOCON_E_ARG;
        offsets[i + 1] = offsets[i] + seq_lens[i];
    }
    return store_create_impl(nullptr, seq_ptrs, offsets.data(), n, out);
}

int isocon_store_create_ptrs_ex(const uint8_t *const *seq_ptrs, const uint64_t *seq_lens, uint32_t n, uint32_t flags, isocon_store **out)
{
    if (!out || (n && (!seq_ptrs || !seq_lens)) || (flags & ~(uint32_t)ISOCON_STORE_PRIVATE_SCRATCH)) return ISOCON_E_ARG;
    std::vector<uint64_t> offsets((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (seq_lens[i] && !seq_ptrs[i]) return ISOCON_E_ARG;
        offsets[i + 1] = offsets[i] + seq_lens[i];
    }
    return store_create_impl(nullptr, seq_ptrs, offsets.data(), n, out, (flags & ISOCON_STORE_PRIVATE_SCRATCH) != 0);
}

void *isocon_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    g_pinned.emplace_back((const char *)p, (size_t)(bytes ? bytes : 16));
    return p;
}

void isocon_host_free(void *p)
{
    if (!p) return;
    for (size_t i = 0; i < g_pinned.size(); ++i)
        if (g_pinned[i].first == (const char *)p) { g_pinned.erase(g_pinned.begin() + (long)i); break; }
    (void)hipHostFree(p);
}

void isocon_store_destroy(isocon_store *s)
{
    if (!s) return;
    if (s->d_planes) (void)hipFree(s->d_planes);
    if (s->d_lens) (void)hipFree(s->d_lens);
    if (s->d_bytes) (void)hipFree(s->d_bytes);
    if (s->d_boff) (void)hipFree(s->d_boff);
    if (s->own_pool) s->own_pool->release();
    delete s;
}

uint32_t isocon_store_size(const isocon_store *s) { return s ? s->dev.n : 0; }
uint64_t isocon_store_device_bytes(const isocon_store *s) { return s ? s->device_bytes : 0; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// explicit pair lists
// ---------------------------------------------------------------------------------------------------------------
namespace {

template <int W>
int launch_band_tiles(isocon_store *st, const DevStore &S, const std::vector<uint32_t> &tile_shared, const std::vector<uint32_t> &lane_ids,
                      const std::vector<int32_t> &lane_k, std::vector<int32_t> &out, EventTimer &tm)
{
    const size_t nt = tile_shared.size();
    out.assign(nt * 64, -1);
    if (!nt) return ISOCON_OK;
    DevBuf d_ts(&st->pool, SLOT_ED_TS), d_ids(&st->pool, SLOT_ED_IDS), d_k(&st->pool, SLOT_ED_K), d_out(&st->pool, SLOT_ED_OUT);
    int rc;
    if ((rc = d_ts.alloc(nt * 4)) || (rc = d_ids.alloc(nt * 64 * 4)) || (rc = d_k.alloc(nt * 64 * 4)) || (rc = d_out.alloc(nt * 64 * 4))) return rc;
    ISO_HIP_CHECK(copy_h2d(d_ts.p, tile_shared.data(), nt * 4));
    ISO_HIP_CHECK(copy_h2d(d_ids.p, lane_ids.data(), nt * 64 * 4));
    ISO_HIP_CHECK(copy_h2d(d_k.p, lane_k.data(), nt * 64 * 4));
    tm.start();
    hipLaunchKernelGGL(k_ed_band_tiles<W>, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, 0, S, d_ts.as<uint32_t>(),
                       d_ids.as<uint32_t>(), d_k.as<int32_t>(), d_out.as<int32_t>(), (uint32_t)nt);
    ISO_HIP_CHECK(hipGetLastError());
    tm.stop();
    ISO_HIP_CHECK(copy_d2h(out.data(), d_out.p, nt * 64 * 4));
    return ISOCON_OK;
}

int run_band_stage(int W, isocon_store *st, const DevStore &S, const std::vector<uint32_t> &ts, const std::vector<uint32_t> &ids,
                   const std::vector<int32_t> &ks, std::vector<int32_t> &out, EventTimer &tm)
{
    switch (W) {
    case 1: return launch_band_tiles<1>(st, S, ts, ids, ks, out, tm);
    case 2: return launch_band_tiles<2>(st, S, ts, ids, ks, out, tm);
    case 4: return launch_band_tiles<4>(st, S, ts, ids, ks, out, tm);
    case 8: return launch_band_tiles<8>(st, S, ts, ids, ks, out, tm);
    default: return ISOCON_E_ARG;
    }
}

int run_full(isocon_store *st, const std::vector<uint32_t> &a, const std::vector<uint32_t> &b,
             const std::vector<int32_t> &k, std::vector<int32_t> &out, EventTimer &tm)
{
    const size_t np = a.size();
    out.assign(np, -1);
    if (!np) return ISOCON_OK;
    DevBuf d_a(&st->pool, SLOT_FULL_A), d_b(&st->pool, SLOT_FULL_B), d_k(&st->pool, SLOT_FULL_K), d_out(&st->pool, SLOT_FULL_OUT);
    int rc;
    if ((rc = d_a.alloc(np * 4)) || (rc = d_b.alloc(np * 4)) || (rc = d_k.alloc(np * 4)) || (rc = d_out.alloc(np * 4))) return rc;
    ISO_HIP_CHECK(copy_h2d(d_a.p, a.data(), np * 4));
    ISO_HIP_CHECK(copy_h2d(d_b.p, b.data(), np * 4));
    ISO_HIP_CHECK(copy_h2d(d_k.p, k.data(), np * 4));
    bool multipass = false;
    int32_t maxtext = 0;
    for (size_t p = 0; p < np; ++p) {
        const int32_t la = st->lens[a[p]], lb = st->lens[b[p]];
        if (std::min(la, lb) > 4096) multipass = true;
        maxtext = std::max(maxtext, std::max(la, lb));
    }
    const size_t lds = multipass ? (size_t)((maxtext + 15) & ~15) : 0;
    if (lds > 160 * 1024) { g_last_error = "sequence longer than 163840 bases in the un-banded kernel"; return ISOCON_E_UNSUPPORTED; }
    if (lds > 64 * 1024)
        ISO_HIP_CHECK(hipFuncSetAttribute((const void *)k_ed_full, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    tm.start();
    hipLaunchKernelGGL(k_ed_full, dim3((unsigned)np), dim3(64), lds, 0, st->dev, d_a.as<uint32_t>(), d_b.as<uint32_t>(),
                       d_k.as<int32_t>(), d_out.as<int32_t>(), (uint32_t)np);
    ISO_HIP_CHECK(hipGetLastError());
    tm.stop();
    ISO_HIP_CHECK(copy_d2h(out.data(), d_out.p, np * 4));
    return ISOCON_OK;
}

// k < 0: unbounded.  One wavefront per pair, a grid of at most 8192 that strides over the list (each needs a row buffer).
int run_bytes(isocon_store *st, const std::vector<uint32_t> &a, const std::vector<uint32_t> &b, const std::vector<int32_t> &k, std::vector<int32_t> &out, EventTimer &tm)
{
    const size_t np = a.size();
    out.assign(np, -1);
    if (!np) return ISOCON_OK;
    if (!st->d_bytes) { g_last_error = "internal: byte-wise distances on a store that kept no bytes"; return ISOCON_E_HIP; }
    if (st->maxlen >= (int32_t)EB_INF - 1) { g_last_error = "byte-wise distances: sequences of 8 388 606 bases and more are not supported"; return ISOCON_E_UNSUPPORTED; }
    const uint32_t row_stride = (uint32_t)st->maxlen + 130u;
    size_t waves = std::min<size_t>(np, 8192);
    while (waves > 256 && waves * row_stride * 4 > ((size_t)2 << 30)) waves /= 2;
    DevBuf d_a(&st->pool, SLOT_EB_A), d_b(&st->pool, SLOT_EB_B), d_k(&st->pool, SLOT_EB_K), d_out(&st->pool, SLOT_EB_OUT), d_rows(&st->pool, SLOT_EB_ROWS);
    int rc;
    if ((rc = d_a.alloc(np * 4)) || (rc = d_b.alloc(np * 4)) || (rc = d_k.alloc(np * 4)) || (rc = d_out.alloc(np * 4)) || (rc = d_rows.alloc(waves * row_stride * 4))) return rc;
    ISO_HIP_CHECK(copy_h2d(d_a.p, a.data(), np * 4));
    ISO_HIP_CHECK(copy_h2d(d_b.p, b.data(), np * 4));
    ISO_HIP_CHECK(copy_h2d(d_k.p, k.data(), np * 4));
    tm.start();
    hipLaunchKernelGGL(k_ed_bytes, dim3((unsigned)waves), dim3(64), 0, 0, ByteStore{st->d_bytes, st->d_boff, st->bytes_base, st->d_lens}, d_a.as<uint32_t>(), d_b.as<uint32_t>(),
                       d_k.as<int32_t>(), (unsigned long long)np, d_rows.as<uint32_t>(), row_stride, d_out.as<int32_t>());
    ISO_HIP_CHECK(hipGetLastError());
    tm.stop();
    ISO_HIP_CHECK(copy_d2h(out.data(), d_out.p, np * 4));
    return ISOCON_OK;
}

}  // namespace

namespace isocon {

// Shared by isocon_ed_pairs and the NN fallback: exact bounded/unbounded distances for an explicit pair list.
// Stages: 64-row band (k <= 63), 128, 256, 512 rows, then the un-banded kernel.
// images: distances of the sequences' IMAGES in the planes (a set with more than four symbols: lower bounds of the true distances, see
// isocon_store) -- the pairs of exceptional sequences are not split off to the byte-wise kernel.
int ed_pairs_impl(isocon_store *st, const uint32_t *a, const uint32_t *b, const int32_t *k, uint64_t n_pairs,
                  int32_t *out_ed, float *kernel_ms, uint64_t *full_pairs, bool images)
{
    EventTimer tm;
    const uint32_t n = st->dev.n;
    for (uint64_t p = 0; p < n_pairs; ++p)
        if (a[p] >= n || b[p] >= n) return ISOCON_E_ARG;
    // which side is shared more often?  (edit distance is symmetric)
    bool swap_roles = false;
    {
        std::vector<uint8_t> seen_a(n, 0), seen_b(n, 0);
        size_t da = 0, db = 0;
        for (uint64_t p = 0; p < n_pairs; ++p) {
            if (!seen_a[a[p]]) { seen_a[a[p]] = 1; ++da; }
            if (!seen_b[b[p]]) { seen_b[b[p]] = 1; ++db; }
        }
        swap_roles = db < da;
    }
    std::vector<uint64_t> pending;
    pending.reserve(n_pairs);
    if (st->n_exc && !images) {
        // pairs with a sequence that holds symbols outside the planes' map: on the bytes (ed_bytes.hpp), whatever their threshold
        std::vector<uint64_t> xp;
        for (uint64_t p = 0; p < n_pairs; ++p) (st->exc[a[p]] || st->exc[b[p]] ? xp : pending).push_back(p);
        if (!xp.empty()) {
            std::vector<uint32_t> xa(xp.size()), xb(xp.size());
            std::vector<int32_t> xk(xp.size()), res;
            for (size_t i = 0; i < xp.size(); ++i) { xa[i] = a[xp[i]]; xb[i] = b[xp[i]]; xk[i] = k ? k[xp[i]] : -1; }
            int rc = run_bytes(st, xa, xb, xk, res, tm);
            if (rc) return rc;
            for (size_t i = 0; i < xp.size(); ++i) out_ed[xp[i]] = res[i];
        }
    } else {
        pending.resize(n_pairs);
        std::iota(pending.begin(), pending.end(), 0);
    }
    static const int stages[4] = {1, 2, 4, 8};
    for (int si = 0; si < 4 && !pending.empty(); ++si) {
        const int W = stages[si];
        const int32_t kcap = 64 * W - 1;
        auto sh = [&](uint64_t p) { return swap_roles ? b[p] : a[p]; };
        auto ln = [&](uint64_t p) { return swap_roles ? a[p] : b[p]; };
        std::sort(pending.begin(), pending.end(), [&](uint64_t x, uint64_t y) {
            if (sh(x) != sh(y)) return sh(x) < sh(y);
            const int32_t lx = st->lens[ln(x)], ly = st->lens[ln(y)];
            if (lx != ly) return lx < ly;
            return x < y;
        });
        std::vector<uint64_t> todo = pending, next;
        if (W == 1) {
            // Pairs that share their sequence with fewer than 16 others would leave most lanes of a tile empty: one pair per
            // lane instead (ed_lanes.hpp; ~1.7x the column cost, every lane busy).  ISOCON_DEBUG_VARIANT=ed_lanes=1 / =0: all / none.
            const char *e = variant_value("ed_lanes");
            const size_t min_group = e ? (atoi(e) ? (size_t)-1 : 0) : 16;
            std::vector<uint64_t> lanes_p, tiles_p;
            for (size_t i = 0; i < todo.size();) {
                size_t j = i;
                while (j < todo.size() && sh(todo[j]) == sh(todo[i])) ++j;
                std::vector<uint64_t> &dst = (j - i) < min_group ? lanes_p : tiles_p;
                dst.insert(dst.end(), todo.begin() + i, todo.begin() + j);
                i = j;
            }
            if (!lanes_p.empty()) {
                const size_t np = lanes_p.size();
                std::vector<uint32_t> la(np), lb2(np);
                std::vector<int32_t> lk(np), res(np, -1);
                for (size_t i = 0; i < np; ++i) {
                    const uint64_t p = lanes_p[i];
                    la[i] = a[p]; lb2[i] = b[p];
                    const int32_t kr = k ? k[p] : -1;
                    lk[i] = kr < 0 ? kcap : std::min(kr, kcap);
                }
                DevBuf d_a(&st->pool, SLOT_ED_TS), d_b(&st->pool, SLOT_ED_IDS), d_k(&st->pool, SLOT_ED_K), d_out(&st->pool, SLOT_ED_OUT);
                int rc;
                if ((rc = d_a.alloc(np * 4)) || (rc = d_b.alloc(np * 4)) || (rc = d_k.alloc(np * 4)) || (rc = d_out.alloc(np * 4))) return rc;
                ISO_HIP_CHECK(copy_h2d(d_a.p, la.data(), np * 4));
                ISO_HIP_CHECK(copy_h2d(d_b.p, lb2.data(), np * 4));
                ISO_HIP_CHECK(copy_h2d(d_k.p, lk.data(), np * 4));
                NNParams none;
                memset(&none, 0, sizeof(none));
                tm.start();
                hipLaunchKernelGGL(k_ed_lanes<false>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, 0, st->dev, none, d_a.as<uint32_t>(), d_b.as<uint32_t>(),
                                   d_k.as<int32_t>(), (uint64_t)np, d_out.as<int32_t>());
                ISO_HIP_CHECK(hipGetLastError());
                tm.stop();
                ISO_HIP_CHECK(copy_d2h(res.data(), d_out.p, np * 4));
                for (size_t i = 0; i < np; ++i) {
                    const uint64_t p = lanes_p[i];
                    const int32_t kr = k ? k[p] : -1;
                    if (res[i] >= 0) out_ed[p] = res[i];
                    else if (kr >= 0 && kr <= kcap) out_ed[p] = -1;
                    else next.push_back(p);
                }
            }
            todo.swap(tiles_p);
        }
        for (int round = 0; round < 2 && !todo.empty(); ++round) {
            // round 0: tiles of up to 64 lanes per shared sequence; round 1: one lane per tile for pairs whose
            // tile could not certify the threshold (heterogeneous length differences)
            std::vector<uint32_t> ts, ids;
            std::vector<int32_t> ks;
            std::vector<uint64_t> slot_pair;
            size_t i = 0;
            while (i < todo.size()) {
                size_t j = i;
                const uint32_t s0 = sh(todo[i]);
                const size_t lim = round == 0 ? 64 : 1;
                while (j < todo.size() && j - i < lim && sh(todo[j]) == s0) ++j;
                ts.push_back(s0);
                for (size_t l = 0; l < 64; ++l) {
                    if (i + l < j) {
                        const uint64_t p = todo[i + l];
                        ids.push_back(ln(p));
                        const int32_t kr = k ? k[p] : -1;
                        ks.push_back(kr < 0 ? kcap : std::min(kr, kcap));
                        slot_pair.push_back(p);
                    } else {
                        ids.push_back(0xffffffffu);
                        ks.push_back(-1);
                        slot_pair.push_back(~(uint64_t)0);
                    }
                }
                i = j;
            }
            std::vector<int32_t> res;
            int rc = run_band_stage(W, st, st->dev, ts, ids, ks, res, tm);
            if (rc) return rc;
            std::vector<uint64_t> retry;
            for (size_t sidx = 0; sidx < slot_pair.size(); ++sidx) {
                const uint64_t p = slot_pair[sidx];
                if (p == ~(uint64_t)0) continue;
                const int32_t r = res[sidx];
                const int32_t kr = k ? k[p] : -1;
                if (r >= 0) out_ed[p] = r;
                else if (r == -2) retry.push_back(p);
                else if (kr >= 0 && kr <= kcap) out_ed[p] = -1;
                else next.push_back(p);
            }
            todo.swap(retry);
        }
        if (!todo.empty()) { g_last_error = "internal: single-lane tile undetermined"; return ISOCON_E_HIP; }
        pending.swap(next);
    }
    if (full_pairs) *full_pairs = pending.size();
    if (!pending.empty()) {
        std::vector<uint32_t> fa, fb;
        std::vector<int32_t> fk, res;
        for (uint64_t p : pending) { fa.push_back(a[p]); fb.push_back(b[p]); fk.push_back(k ? k[p] : -1); }
        int rc = run_full(st, fa, fb, fk, res, tm);
        if (rc) return rc;
        for (size_t i = 0; i < pending.size(); ++i) out_ed[pending[i]] = res[i];
    }
    if (kernel_ms) *kernel_ms = tm.total;
    return ISOCON_OK;
}

}  // namespace isocon

// 64-bit digest of the packed set (every plane word and every length, each mixed with its position): two stores have the
// same digest iff -- up to hash collisions -- they hold the same sequences in the same order.  Sharded runs compare it.
__global__ __launch_bounds__(256) void k_store_digest(const uint64_t *__restrict__ planes, size_t n_words, const int32_t *__restrict__ lens, uint32_t n,
                                                      unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words + n; i += stride) {
        unsigned long long v = i < n_words ? planes[i] : (unsigned long long)(uint32_t)lens[i - n_words];
        v ^= (unsigned long long)i * 0xD6E8FEB86659FD93ull;
        v ^= v >> 32; v *= 0x9E3779B97F4A7C15ull; v ^= v >> 29; v *= 0xBF58476D1CE4E5B9ull; v ^= v >> 32;
        acc += v;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

extern "C" int isocon_store_digest(const isocon_store *s, uint64_t *out)
{
    if (!s || !out) return ISOCON_E_ARG;
    DevBuf d_acc;
    int rc;
    if ((rc = d_acc.alloc(8))) return rc;
    ISO_HIP_CHECK(hipMemset(d_acc.p, 0, 8));
    const size_t n_words = (size_t)s->dev.nchunks * std::max<uint32_t>(s->dev.n, 1) * 2;
    hipLaunchKernelGGL(k_store_digest, dim3(1024), dim3(256), 0, 0, s->d_planes, n_words, s->d_lens, s->dev.n, d_acc.as<unsigned long long>());
    ISO_HIP_CHECK(hipGetLastError());
    ISO_HIP_CHECK(hipMemcpy(out, d_acc.p, 8, hipMemcpyDeviceToHost));
    return ISOCON_OK;
}

extern "C" int isocon_ed_pairs(isocon_store *s, const uint32_t *a, const uint32_t *b, const int32_t *k, uint64_t n_pairs,
                               int32_t *out_ed, float *kernel_ms)
{
    if (!s || (n_pairs && (!a || !b || !out_ed))) return ISOCON_E_ARG;
    if (kernel_ms) *kernel_ms = 0.f;
    if (!n_pairs) return ISOCON_OK;
    return ed_pairs_impl(s, a, b, k, n_pairs, out_ed, kernel_ms, nullptr, false);
}

extern "C" int isocon_qgram_params(int32_t *out)
{
    if (out) { out[0] = QG_Q; out[1] = QG_B0; out[2] = QG_B1; out[3] = QG_CAP; }
    return QM_K;
}

// q-gram lower bounds of explicit pairs (qgram_mm.hpp), computed on the 2-bit planes
namespace isocon {
int qgram_bounds_for_pairs(isocon_store *s, const uint32_t *a, const uint32_t *b, uint64_t n_pairs, int32_t *out_bound, float *kernel_ms)
{
    const uint32_t n = s->dev.n;
    const uint32_t n_pad = std::max<uint32_t>(QM_TILE, ((n + QM_TILE - 1) / QM_TILE) * QM_TILE);
    DevBuf d_prof(&s->pool, SLOT_NN_QPROF), d_sum(&s->pool, SLOT_NN_QSUM), d_a(&s->pool, SLOT_ED_TS), d_b(&s->pool, SLOT_ED_IDS), d_out(&s->pool, SLOT_ED_OUT);
    int rc;
    if ((rc = d_prof.alloc((size_t)n_pad * (QM_K / 2))) || (rc = d_sum.alloc((size_t)n * 4)) || (rc = d_a.alloc(n_pairs * 4)) || (rc = d_b.alloc(n_pairs * 4)) ||
        (rc = d_out.alloc(n_pairs * 4)))
        return rc;
    s->pool.bound_tag.valid = false;          // the profile slot is shared with the bound matrix builds
    ISO_HIP_CHECK(copy_h2d(d_a.p, a, n_pairs * 4));
    ISO_HIP_CHECK(copy_h2d(d_b.p, b, n_pairs * 4));
    EventTimer tm;
    tm.start();
    hipLaunchKernelGGL(k_qgram_profile4, dim3((n + QP_SEQS - 1) / QP_SEQS), dim3(256), 0, 0, s->dev, d_prof.as<uint8_t>(), d_sum.as<uint32_t>(), n_pad);
    ISO_HIP_CHECK(hipGetLastError());
    // (a launch holds fewer than 2^32 threads: 2^24 pairs per launch)
    for (uint64_t at = 0; at < n_pairs; at += (uint64_t)1 << 24) {
        const uint64_t cnt = std::min<uint64_t>((uint64_t)1 << 24, n_pairs - at);
        hipLaunchKernelGGL(k_qgram_lb_pairs, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, 0, d_prof.as<uint8_t>(), d_sum.as<uint32_t>(), n_pad, d_a.as<uint32_t>() + at,
                           d_b.as<uint32_t>() + at, cnt, d_out.as<int32_t>() + at);
        ISO_HIP_CHECK(hipGetLastError());
    }
    const float ms = tm.stop();
    if (kernel_ms) *kernel_ms = ms;
    ISO_HIP_CHECK(copy_d2h(out_bound, d_out.p, n_pairs * 4));
    return ISOCON_OK;
}
}  // namespace isocon

// ... exposed for tests: what the main pass of the NN search consults
extern "C" int isocon_qgram_bound_pairs(isocon_store *s, const uint32_t *a, const uint32_t *b, uint64_t n_pairs, int32_t *out_bound)
{
    if (!s || (n_pairs && (!a || !b || !out_bound))) return ISOCON_E_ARG;
    if (!n_pairs) return ISOCON_OK;
    if (s->n_exc) { g_last_error = "q-gram bounds are defined on the 2-bit planes: the set holds more than four distinct symbols"; return ISOCON_E_ALPHABET; }
    const uint32_t n = s->dev.n;
    for (uint64_t i = 0; i < n_pairs; ++i)
        if (a[i] >= n || b[i] >= n) { g_last_error = "pair index out of range"; return ISOCON_E_ARG; }
    return qgram_bounds_for_pairs(s, a, b, n_pairs, out_bound, nullptr);
}

// ... and the block bound behind it (nn_filter.hpp), one workgroup per pair
extern "C" int isocon_block_bound_pairs(isocon_store *s, const uint32_t *owner, const uint32_t *partner, uint64_t n_pairs, int32_t probe_stride, int32_t *out_count)
{
    if (probe_stride != 4 && probe_stride != 2) return ISOCON_E_ARG;
    if (!s || (n_pairs && (!owner || !partner || !out_count))) return ISOCON_E_ARG;
    if (!n_pairs) return ISOCON_OK;
    if (s->n_exc) { g_last_error = "block bounds are defined on the 2-bit planes: the set holds more than four distinct symbols"; return ISOCON_E_ALPHABET; }
    const uint32_t n = s->dev.n;
    if (n_pairs >= ((uint64_t)1 << 31)) return ISOCON_E_ARG;
    for (uint64_t i = 0; i < n_pairs; ++i)
        if (owner[i] >= n || partner[i] >= n) { g_last_error = "pair index out of range"; return ISOCON_E_ARG; }
    const uint32_t stride = nnf_text2_stride(s->maxlen);
    DevBuf d_text(&s->pool, SLOT_NN_TEXT2), d_a(&s->pool, SLOT_ED_TS), d_b(&s->pool, SLOT_ED_IDS), d_out(&s->pool, SLOT_ED_OUT);
    int rc;
    if ((rc = d_text.alloc((size_t)n * stride * 4)) || (rc = d_a.alloc(n_pairs * 4)) || (rc = d_b.alloc(n_pairs * 4)) || (rc = d_out.alloc(n_pairs * 4))) return rc;
    ISO_HIP_CHECK(copy_h2d(d_a.p, owner, n_pairs * 4));
    ISO_HIP_CHECK(copy_h2d(d_b.p, partner, n_pairs * 4));
    hipLaunchKernelGGL(k_build_text2, dim3(n), dim3(256), 0, 0, s->dev, d_text.as<uint32_t>(), stride);
    hipLaunchKernelGGL(k_nn_block_count_pairs, dim3((unsigned)n_pairs), dim3(256), 0, 0, d_text.as<uint32_t>(), stride, s->dev.lens, d_a.as<uint32_t>(), d_b.as<uint32_t>(), d_out.as<int32_t>(), probe_stride);
    ISO_HIP_CHECK(hipGetLastError());
    ISO_HIP_CHECK(copy_d2h(out_count, d_out.p, n_pairs * 4));
    return ISOCON_OK;
}

#include "nn_context.inc"
#include "nn_bounds.inc"
#include "nn_lists.inc"
#include "nn_main.inc"
#include "nn_wide.inc"
#include "nn_images.inc"
#include "nn_entry.inc"
#include "sg_host.inc"
#include "msa_host.inc"
#include "hw_host.inc"

extern "C" int isocon_partition_ids(uint32_t n, const int32_t *degree, uint64_t n_edges, const uint32_t *edge_a, const uint32_t *edge_b,
                                    const uint32_t *rank, int32_t nbr_tiebreak, uint32_t *out_centre, int64_t *out_weight,
                                    uint64_t *out_member_ptr, uint32_t *out_members, uint32_t *n_parts)
{
    if (!out_member_ptr || !n_parts || (n && (!degree || !rank || !out_centre || !out_weight || !out_members)) || (n_edges && (!edge_a || !edge_b))) return ISOCON_E_ARG;
    return partition_ids_impl(n, degree, n_edges, edge_a, edge_b, rank, nbr_tiebreak, out_centre, out_weight, out_member_ptr, out_members, n_parts);
}
