// Micro-benchmark: issue rate of the integer VALU instructions the bit-vector kernels are made of (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, unsigned long long *clk)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t b = blockIdx.x + 1;
    unsigned long long q0 = a0, q1 = a1;
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    a3 ^= lds[threadIdx.x];
    uint32_t ad = (threadIdx.x & 3) * 4;
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 1) { REP16(asm volatile("v_and_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_and_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_and_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_and_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 2) { REP16(asm volatile("v_mov_b32_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n v_mov_b32_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n v_mov_b32_sdwa %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n v_mov_b32_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 3) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 4) { REP16(asm volatile("v_and_or_b32 %0, %0, 12, %4\n v_and_or_b32 %1, %1, 12, %4\n v_and_or_b32 %2, %2, 12, %4\n v_and_or_b32 %3, %3, 12, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 5) { REP16(asm volatile("v_alignbit_b32 %0, %0, %4, 1\n v_alignbit_b32 %1, %1, %4, 1\n v_alignbit_b32 %2, %2, %4, 1\n v_alignbit_b32 %3, %3, %4, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 6) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 7) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 8) { REP16(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1" : "+v"(q0), "+v"(q1) : : "vcc");) }
        if (OP == 9) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0" : "+v"(q0), "+v"(q1) : : "vcc");) }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)q0 ^ (uint32_t)q1;
}

template <int OP>
void run(const char *name)
{
    const int w = 8, blocks = 256 * w, iters = 2000;
    uint32_t *d; unsigned long long *clk;
    hipMalloc(&d, (size_t)blocks * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    const double insts_per_simd = (double)iters * 64 * w * (OP >= 8 && OP < 10 ? 0.5 : 1.0);
    printf("%-28s %.3f ms  clock %.2f GHz  cycles/wave-instr/SIMD = %.2f\n", name, ms, ghz, ms * 1e-3 * ghz * 1e9 / insts_per_simd);
    hipFree(d); hipFree(clk);
}

int main()
{
    run<0>("v_add_u32_sdwa BYTE_1");
    run<1>("v_and_b32_sdwa BYTE_2");
    run<2>("v_mov_b32_sdwa BYTE_3");
    run<3>("v_lshl_add_u32");
    run<4>("v_and_or_b32");
    run<5>("v_alignbit_b32");
    run<6>("v_add_u32 vv");
    run<7>("v_add_co+v_addc pair");
    run<8>("v_lshrrev_b64");
    run<9>("v_lshl_add_u64");
    return 0;
}
