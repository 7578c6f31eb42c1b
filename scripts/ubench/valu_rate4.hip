// Micro-benchmark: issue rate of v_sad_u8 / v_dot4 / v_bcnt forms on gfx950 (what the q-gram bound kernel could be made of).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate4 valu_rate4.hip ; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, unsigned long long *clk, uint32_t sv)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t b = blockIdx.x + 1, c = threadIdx.x * 11;
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_sad_u8 %0, %4, %5, %0\n v_sad_u8 %1, %4, %5, %1\n v_sad_u8 %2, %4, %5, %2\n v_sad_u8 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 1) { REP16(asm volatile("v_sad_u8 %0, %4, %5, %0\n v_sad_u8 %1, %4, %5, %1\n v_sad_u8 %2, %4, %5, %2\n v_sad_u8 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(sv));) }
        if (OP == 2) { REP16(asm volatile("v_sad_u32 %0, %4, %5, %0\n v_sad_u32 %1, %4, %5, %1\n v_sad_u32 %2, %4, %5, %2\n v_sad_u32 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 3) { REP16(asm volatile("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 4) { REP16(asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %4, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 5) { REP16(asm volatile("v_sad_u16 %0, %4, %5, %0\n v_sad_u16 %1, %4, %5, %1\n v_sad_u16 %2, %4, %5, %2\n v_sad_u16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 6) { REP16(asm volatile("v_pk_sub_u16 %0, %4, %5 clamp\n v_pk_sub_u16 %1, %4, %5 clamp\n v_pk_sub_u16 %2, %4, %5 clamp\n v_pk_sub_u16 %3, %4, %5 clamp" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 7) { REP16(asm volatile("v_min_u32 %0, %4, %0\n v_min_u32 %1, %4, %1\n v_min_u32 %2, %4, %2\n v_min_u32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 8) { REP16(asm volatile("v_dot8_u32_u4 %0, %4, %5, %0\n v_dot8_u32_u4 %1, %4, %5, %1\n v_dot8_u32_u4 %2, %4, %5, %2\n v_dot8_u32_u4 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

template <int OP>
void run(const char *name)
{
    const int w = 8, blocks = 256 * w, iters = 2000;
    uint32_t *d; unsigned long long *clk;
    hipMalloc(&d, (size_t)blocks * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, nullptr, 0x01020304u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, clk, 0x01020304u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    const double insts_per_simd = (double)iters * 64 * w;
    printf("%-28s %.3f ms  clock %.2f GHz  cycles/wave-instr/SIMD = %.2f\n", name, ms, ghz, ms * 1e-3 * ghz * 1e9 / insts_per_simd);
    hipFree(d); hipFree(clk);
}

int main()
{
    run<0>("v_sad_u8 vvv");
    run<1>("v_sad_u8 v,s,v");
    run<2>("v_sad_u32 vvv");
    run<3>("v_dot4_u32_u8 vvv");
    run<4>("v_bcnt_u32_b32");
    run<5>("v_sad_u16 vvv");
    run<6>("v_pk_sub_u16 clamp");
    run<7>("v_min_u32");
    run<8>("v_dot8_u32_u4 vvv");
    return 0;
}
