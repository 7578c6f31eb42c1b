// Micro-test for the thermometer-code contraction of the q-gram bound (csrc/qgram_mm.hpp): does
// v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands (1.0 = 0x2, 0.0 = 0x0, unit scales) compute exact dot products of
// binary vectors in the lane layout the kernel assumes, and at what rate does it issue?
//   A operand, lane l: 16 bytes = the 32 elements k = 32 (l >> 5) .. + 31 of row (l & 31); B likewise for column (l & 31);
//   C: col = lane & 31 (B side), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (A side).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_fp4 mfma_fp4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_once(const uint4 *a, const uint4 *b, float *c)
{
    const int lane = threadIdx.x;
    const uint4 x = a[lane], y = b[lane];
    v8i A = {(int)x.x, (int)x.y, (int)x.z, (int)x.w, 0, 0, 0, 0};
    v8i B = {(int)y.x, (int)y.y, (int)y.z, (int)y.w, 0, 0, 0, 0};
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) c[r * 64 + lane] = acc[r];
}

// the same with scale operands 0: hipcc then emits the NON-scaled v_mfma_f32_32x32x64_f8f6f4 (no v_mfma_ld_scale half)
__global__ void k_once_ns(const uint4 *a, const uint4 *b, float *c)
{
    const int lane = threadIdx.x;
    const uint4 x = a[lane], y = b[lane];
    v8i A = {(int)x.x, (int)x.y, (int)x.z, (int)x.w, 0, 0, 0, 0};
    v8i B = {(int)y.x, (int)y.y, (int)y.z, (int)y.w, 0, 0, 0, 0};
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, 0, 0, 0);
    for (int r = 0; r < 16; ++r) c[r * 64 + lane] = acc[r];
}

__global__ void k_once_i8(const uint4 *a, const uint4 *b, int *c)
{
    const int lane = threadIdx.x;
    const uint4 x = a[lane], y = b[lane];
    v4i A = {(int)x.x, (int)x.y, (int)x.z, (int)x.w};
    v4i B = {(int)y.x, (int)y.y, (int)y.z, (int)y.w};
    v16i acc = {};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) c[r * 64 + lane] = acc[r];
}

// rate: NACC independent accumulators, back to back
template <int NACC, int FP4>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters)
{
    v8i A = {(int)threadIdx.x, 0x22222222, 0x20202020, 0x02020202, 0, 0, 0, 0};
    v8i B = {0x22222222, (int)(threadIdx.x * 7u), 0x22002200, 0x00220022, 0, 0, 0, 0};
    v4i A4 = {A[0], A[1], A[2], A[3]}, B4 = {B[0], B[1], B[2], B[3]};
    v16f acc[NACC];
    v16i iacc[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = v16f{}; iacc[i] = v16i{}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (FP4 == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[i], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            else if (FP4 == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[i], 4, 4, 0, 0, 0, 0);
            else iacc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, iacc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r] + (float)iacc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int FP4>
void rate(const char *name, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    float *d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<NACC, FP4><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<NACC, FP4><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * NACC * blocks_per_cu;
    const double macs = (double)blocks * 4 * iters * NACC * 32 * 32 * (FP4 ? 64 : 32);
    printf("%-28s acc=%d waves/SIMD=%d  %.3f ms  %.1f cycles per MFMA per SIMD (at 2.4 GHz)  %.2f PMAC/s\n", name, NACC, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / per_simd, macs / (ms * 1e-3) * 1e-15);
    hipFree(d);
}

int main()
{
    // ---- layout check, asymmetric random binary data
    srand(12345);
    std::vector<uint8_t> Abit(32 * 64), Bbit(32 * 64);
    for (auto &v : Abit) v = rand() % 3 == 0;
    for (auto &v : Bbit) v = rand() % 2 == 0;
    std::vector<uint8_t> a(64 * 16, 0), b(64 * 16, 0);
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 32; ++e) {
            const int r = l & 31, k = 32 * (l >> 5) + e;
            if (Abit[r * 64 + k]) a[l * 16 + e / 2] |= (e & 1) ? 0x20 : 0x02;
            if (Bbit[r * 64 + k]) b[l * 16 + e / 2] |= (e & 1) ? 0x20 : 0x02;
        }
    uint4 *da, *db;
    float *dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 16 * 64 * 4);
    hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice);
    k_once<<<1, 64>>>(da, db, dc);
    std::vector<float> c(16 * 64);
    hipMemcpy(c.data(), dc, 16 * 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            int ref = 0;
            for (int k = 0; k < 64; ++k) ref += Abit[row * 64 + k] & Bbit[col * 64 + k];
            if ((int)c[r * 64 + l] != ref) { if (bad < 5) printf("fp4 mismatch lane %d reg %d: got %g want %d\n", l, r, c[r * 64 + l], ref); ++bad; }
        }
    printf("fp4 32x32x64 layout check: %s (%d mismatches of 1024)\n", bad ? "FAILED" : "ok", bad);
    {
        k_once_ns<<<1, 64>>>(da, db, dc);
        hipMemcpy(c.data(), dc, 16 * 64 * 4, hipMemcpyDeviceToHost);
        int badn = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                int ref = 0;
                for (int k = 0; k < 64; ++k) ref += Abit[row * 64 + k] & Bbit[col * 64 + k];
                if ((int)c[r * 64 + l] != ref) { if (badn < 3) printf("non-scaled mismatch lane %d reg %d: got %g want %d\n", l, r, c[r * 64 + l], ref); ++badn; }
            }
        printf("fp4 32x32x64 NON-scaled form (scale operands 0): %s (%d mismatches of 1024)\n", badn ? "DIFFERENT" : "ok, same products", badn);
    }
    // i8 32x32x32: lane l holds k = 16 (l >> 5) .. + 15 of row l & 31
    {
        std::vector<int8_t> ai(64 * 16), bi(64 * 16);
        std::vector<int> Am(32 * 32), Bm(32 * 32);
        for (auto &v : Am) v = rand() % 7 - 3;
        for (auto &v : Bm) v = rand() % 5 - 2;
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 16; ++e) { ai[l * 16 + e] = (int8_t)Am[(l & 31) * 32 + 16 * (l >> 5) + e]; bi[l * 16 + e] = (int8_t)Bm[(l & 31) * 32 + 16 * (l >> 5) + e]; }
        hipMemcpy(da, ai.data(), 1024, hipMemcpyHostToDevice);
        hipMemcpy(db, bi.data(), 1024, hipMemcpyHostToDevice);
        k_once_i8<<<1, 64>>>(da, db, (int *)dc);
        std::vector<int> ci(16 * 64);
        hipMemcpy(ci.data(), dc, 16 * 64 * 4, hipMemcpyDeviceToHost);
        int badi = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                int ref = 0;
                for (int k = 0; k < 32; ++k) ref += Am[row * 32 + k] * Bm[col * 32 + k];
                if (ci[r * 64 + l] != ref) ++badi;
            }
        printf("i8 32x32x32 layout check: %s (%d mismatches of 1024)\n", badi ? "FAILED" : "ok", badi);
    }
    // ---- issue rate
    rate<1, 1>("fp4 32x32x64", 1);
    rate<4, 1>("fp4 32x32x64", 1);
    rate<4, 1>("fp4 32x32x64", 2);
    rate<4, 2>("fp4 32x32x64 non-scaled", 1);
    rate<4, 2>("fp4 32x32x64 non-scaled", 2);
    rate<8, 2>("fp4 32x32x64 non-scaled", 2);
    rate<4, 0>("i8 32x32x32", 1);
    rate<4, 0>("i8 32x32x32", 2);
    return bad != 0;
}
