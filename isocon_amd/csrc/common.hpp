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

__device__ __forceinline__ int32_t uniform_i32(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int32_t load_relaxed_agent(const int32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace isocon
