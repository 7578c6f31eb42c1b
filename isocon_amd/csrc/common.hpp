// common.hpp -- device store layout, error plumbing and wave helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/isocon_hip.h"

namespace isocon {

// Packed sequence set in HBM.
//   planes[((chunk * n) + id) * 2 + p] : 64 bases of sequence `id`, bit-plane p (0 = low code bit, 1 = high);
//   A=0 C=1 G=2 T=3; bases past the end are 0; there are nchunks = ceil(maxlen/64)+1 chunks (last all-zero).
//   Consecutive ids are adjacent in memory, so a wave whose lanes hold consecutive sequences loads
//   64 x 16 B = 1 KiB contiguous per chunk.
//   il (optional, built on demand for the LDS-table kernel): same indexing, 16 B per (chunk, id), the two code bits
//   of each base adjacent: word x = bases 0..31, word y = bases 32..63, base j of a word at bits 2j, 2j+1.
struct DevStore {
    const uint64_t *planes;
    const uint64_t *il;
    const int32_t *lens;
    uint32_t n;
    uint32_t nchunks;
};

// Which entries of the length-sorted order a shard owns as rows ("launch slots") of the search: the entries x in [begin, end) with
// (x - begin) mod stride < block, numbered in ascending order -- slot s <-> entry begin + (s / block) * stride + s mod block.
//   one GPU:       begin 0, stride 1, block 1                      (slot = entry)
//   cyclic:        begin r, stride N, block 1                      (rank r of N: r, r + N, ...)
//   block-cyclic:  begin r * B, stride N * B, block B = 256        (blocks of one tile row of the bound matrix dealt round-robin: a
//                  rank's 256-slot tiles are as dense as on one GPU, and an entry's window is a run of consecutive slots)
// block is a power of two (block_log2).  The map is monotone, so an interval of entries is an interval of slots.
struct QMap {
    uint32_t begin, end, stride, block_log2;
    __host__ __device__ __forceinline__ uint32_t block() const { return 1u << block_log2; }
    __host__ __device__ __forceinline__ uint64_t entry(uint32_t s) const
    {
        return (uint64_t)begin + (uint64_t)(s >> block_log2) * stride + (s & (block() - 1u));
    }
    // x owned: its slot in s
    __host__ __device__ __forceinline__ bool slot_of(uint32_t x, uint32_t &s) const
    {
        if (x < begin || x >= end) return false;
        const uint32_t d = x - begin, b = d / stride, r = d - b * stride;
        s = (b << block_log2) + r;
        return r < block();
    }
    __host__ __device__ __forceinline__ bool owns(uint32_t x) const { uint32_t s; return slot_of(x, s); }
    __host__ __device__ __forceinline__ uint32_t count() const
    {
        if (end <= begin) return 0u;
        const uint32_t span = end - begin, full = span / stride, rem = span - full * stride;
        return (full << block_log2) + (rem < block() ? rem : block());
    }
    __host__ __device__ __forceinline__ bool same(const QMap &o) const { return begin == o.begin && end == o.end && stride == o.stride && block_log2 == o.block_log2; }
};

extern thread_local std::string g_last_error;

#define ISO_HIP_CHECK(expr)                                                                    \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                  \
            return ISOCON_E_HIP;                                                               \
        }                                                                                      \
    } while (0)

__device__ __forceinline__ int32_t wave_min_i32(int32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}

__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

// exclusive prefix sum of one value per thread over a workgroup of 1024 threads (16 waves); *total = the sum.  wave_sums: 16 words of LDS
__device__ __forceinline__ unsigned long long block_exscan_1024(unsigned long long s, unsigned long long *wave_sums, unsigned long long *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long inc = s;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    unsigned long long off = inc - s, tot = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        if (w < wave) off += wave_sums[w];
        tot += wave_sums[w];
    }
    *total = tot;
    return off;
}

// Exclusive prefix sum of f(in[i]) over n elements in two launches of ceil(n / 2048) workgroups of 256 threads (8 consecutive elements
// per thread): k_scan_tiles leaves every element's prefix inside its tile in tmp[] and the tile's sum in tile_sum[]; k_scan_finish adds
// the sums of the tiles before and writes out[i], and out[n] = the total.  SHIFT: f(v) = (v + 2^SHIFT - 1) >> SHIFT (SHIFT 0: v itself).
static constexpr uint32_t SCAN_TILE = 2048;

template <int SHIFT>
__global__ __launch_bounds__(256) void k_scan_tiles(const uint32_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ tmp, unsigned long long *__restrict__ tile_sum)
{
    __shared__ unsigned long long wave_sums[4];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8u;
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t x = base + j < n ? in[base + j] : 0u;
        v[j] = SHIFT ? (x + (1u << SHIFT) - 1u) >> SHIFT : x;
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = s;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t off = inc - s;
    for (uint32_t w = 0; w < wave; ++w) off += (uint32_t)wave_sums[w];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (base + j < n) tmp[base + j] = off;
        off += v[j];
    }
    if (threadIdx.x == 255) tile_sum[blockIdx.x] = (unsigned long long)off;
}

template <class OUT>
__global__ __launch_bounds__(256) void k_scan_finish(const uint32_t *__restrict__ tmp, uint32_t n, const unsigned long long *__restrict__ tile_sum, uint32_t n_tiles,
                                                      OUT *__restrict__ out, unsigned long long *__restrict__ total_out)
{
    __shared__ unsigned long long part[4];
    unsigned long long s = 0, all = 0;
    for (uint32_t b = threadIdx.x; b < n_tiles; b += 256u) {
        const unsigned long long t = tile_sum[b];
        if (b < blockIdx.x) s += t;
        all += t;
    }
    // workgroup sums of s (tiles before this one) and, for the last workgroup, of all tiles
    const bool last = blockIdx.x + 1 == gridDim.x;
    unsigned long long r = last ? all : s, r2 = s;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { r += __shfl_xor(r, o, 64); r2 += __shfl_xor(r2, o, 64); }
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = r2;
    __syncthreads();
    const unsigned long long before = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = r;
    __syncthreads();
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8u;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (base + j < n) out[base + j] = (OUT)(before + tmp[base + j]);
    if (last && threadIdx.x == 0) {
        const unsigned long long tot = part[0] + part[1] + part[2] + part[3];
        out[n] = (OUT)tot;
        if (total_out) *total_out = tot;
    }
}

__device__ __forceinline__ int32_t uniform_i32(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int32_t load_relaxed_agent(const int32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace isocon
