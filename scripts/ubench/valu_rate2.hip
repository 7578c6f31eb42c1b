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
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 1) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 2) { REP16(asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 3) { REP16(asm volatile("v_xnor_b32 %0, %0, %4\n v_xnor_b32 %1, %1, %4\n v_xnor_b32 %2, %2, %4\n v_xnor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 4) { REP16(asm volatile("v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 5) { REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 6) { REP16(asm volatile("v_mov_b32 %0, s4\n v_mov_b32 %1, s4\n v_mov_b32 %2, s4\n v_mov_b32 %3, s4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 7) { REP16(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 8) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 9) { REP16(asm volatile("v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 10) { REP16(asm volatile("v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 11) { REP16(asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 12) { REP16(asm volatile("v_add_u32 %0, 1, %0\n v_add_u32 %1, 1, %1\n v_add_u32 %2, 1, %2\n v_add_u32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 13) { REP16(asm volatile("v_and_b32 %0, 1, %0\n v_and_b32 %1, 1, %1\n v_and_b32 %2, 1, %2\n v_and_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 14) { REP16(asm volatile("v_and_b32 %0, s4, %0\n v_and_b32 %1, s4, %1\n v_and_b32 %2, s4, %2\n v_and_b32 %3, s4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 15) { REP16(asm volatile("v_xor_b32 %0, 0x12345678, %0\n v_xor_b32 %1, 0x12345678, %1\n v_xor_b32 %2, 0x12345678, %2\n v_xor_b32 %3, 0x12345678, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 16) { REP16(asm volatile("v_bfe_u32 %0, %0, 3, 1\n v_bfe_u32 %1, %1, 3, 1\n v_bfe_u32 %2, %2, 3, 1\n v_bfe_u32 %3, %3, 3, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 17) { REP16(asm volatile("v_bfe_i32 %0, %0, s4, 1\n v_bfe_i32 %1, %1, s4, 1\n v_bfe_i32 %2, %2, s4, 1\n v_bfe_i32 %3, %3, s4, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 18) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 19) { REP16(asm volatile("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 20) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc","s6","s7");) }
        if (OP == 21) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, s[6:7]\n v_cndmask_b32 %1, %1, %4, s[6:7]\n v_cndmask_b32 %2, %2, %4, s[6:7]\n v_cndmask_b32 %3, %3, %4, s[6:7]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc","s6","s7");) }
        if (OP == 22) { REP16(asm volatile("v_cmp_lt_i32 vcc, %0, %4\n v_cmp_lt_i32 vcc, %1, %4\n v_cmp_lt_i32 vcc, %2, %4\n v_cmp_lt_i32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 23) { REP16(asm volatile("v_cmp_lt_i32 s[6:7], %0, %4\n v_cmp_lt_i32 s[6:7], %1, %4\n v_cmp_lt_i32 s[6:7], %2, %4\n v_cmp_lt_i32 s[6:7], %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s6","s7");) }
        if (OP == 24) { REP16(asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 25) { REP16(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 26) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 27) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 28) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 4, %4\n v_lshl_or_b32 %1, %1, 4, %4\n v_lshl_or_b32 %2, %2, 4, %4\n v_lshl_or_b32 %3, %3, 4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 29) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %4\n v_perm_b32 %1, %1, %4, %4\n v_perm_b32 %2, %2, %4, %4\n v_perm_b32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 30) { REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 31) { REP16(asm volatile("v_pk_sub_i16 %0, %0, %4\n v_pk_sub_i16 %1, %1, %4\n v_pk_sub_i16 %2, %2, %4\n v_pk_sub_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 32) { REP16(asm volatile("v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 33) { REP16(asm volatile("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
        if (OP == 34) { REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %4\n v_mbcnt_lo_u32_b32 %1, %1, %4\n v_mbcnt_lo_u32_b32 %2, %2, %4\n v_mbcnt_lo_u32_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4");) }
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
    k<OP><<<blocks, 256>>>(d, 10, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, clk);
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
    run<0>("v_xor_b32 vv");
    run<1>("v_and_b32 vv");
    run<2>("v_or_b32 vv");
    run<3>("v_xnor_b32 vv");
    run<4>("v_not_b32");
    run<5>("v_mov_b32 v,v");
    run<6>("v_mov_b32 v,s");
    run<7>("v_lshrrev_b32 imm");
    run<8>("v_lshlrev_b32 imm");
    run<9>("v_ashrrev_i32 imm");
    run<10>("v_lshrrev_b32 vv");
    run<11>("v_sub_u32");
    run<12>("v_add_u32 v,imm");
    run<13>("v_and_b32 imm");
    run<14>("v_and_b32 s,v (src0 sgpr)");
    run<15>("v_xor_b32 literal");
    run<16>("v_bfe_u32");
    run<17>("v_bfe_i32 v,s,1");
    run<18>("v_add_co_u32 alone");
    run<19>("v_addc_co_u32 alone");
    run<20>("v_cndmask_b32 e32 vcc");
    run<21>("v_cndmask_b32 e64 s[6:7]");
    run<22>("v_cmp_lt_i32 vcc");
    run<23>("v_cmp_lt_i32 e64 s[6:7]");
    run<24>("v_max_i32 vv");
    run<25>("v_min_u32 vv");
    run<26>("v_mul_u32_u24");
    run<27>("v_mad_u32_u24");
    run<28>("v_lshl_or_b32");
    run<29>("v_perm_b32");
    run<30>("v_pk_add_u16");
    run<31>("v_pk_sub_i16");
    run<32>("v_pk_max_i16");
    run<33>("v_popcnt");
    run<34>("v_mbcnt_lo");
    return 0;
}
