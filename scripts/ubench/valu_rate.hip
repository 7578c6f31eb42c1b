// Micro-benchmark: issue rate of the integer VALU instructions the bit-vector kernels are made of (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 ^ 5, a7 = a0 ^ 9;
    uint32_t b = blockIdx.x + 1, c = blockIdx.x * 77 + 5;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 2) { REP16(asm volatile("v_bfi_b32 %0, %0, %4, %5\n v_bfi_b32 %1, %1, %4, %5\n v_bfi_b32 %2, %2, %4, %5\n v_bfi_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 3) { REP16(asm volatile("v_bfe_i32 %0, %0, 3, 1\n v_bfe_i32 %1, %1, 3, 1\n v_bfe_i32 %2, %2, 3, 1\n v_bfe_i32 %3, %3, 3, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 4) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 5) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q0));) }
        if (OP == 6) { REP16(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));) }
        if (OP == 7) { REP16(asm volatile("v_alignbit_b32 %0, %0, %4, 1\n v_alignbit_b32 %1, %1, %4, 1\n v_alignbit_b32 %2, %2, %4, 1\n v_alignbit_b32 %3, %3, %4, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 8) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 9) { REP16(asm volatile("v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %4, %5\n v_or3_b32 %2, %2, %4, %5\n v_or3_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 10) { REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 11) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 12) { REP16(asm volatile("v_xor_b32 %0, %0, s4\n v_xor_b32 %1, %1, s5\n v_xor_b32 %2, %2, s4\n v_xor_b32 %3, %3, s5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s4", "s5");) }
        if (OP == 13) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (OP == 14) { REP16(asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 15) { REP16(asm volatile("v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 16) { REP16(asm volatile("v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %1, %1, %4, %5\n v_max3_i32 %2, %2, %4, %5\n v_max3_i32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 17) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n s_lshr_b64 s[4:5], s[4:5], 1\n v_xor_b32 %1, %1, %4\n s_or_b64 s[6:7], s[6:7], s[4:5]\n v_xor_b32 %2, %2, %4\n s_lshl_b32 s8, s8, 1\n v_xor_b32 %3, %3, %4\n s_add_i32 s9, s9, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s4", "s5", "s6", "s7", "s8", "s9", "scc");) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3);
}

template <int OP>
double run(const char *name, int wpb_blocks, int valu_per_iter)
{
    const int blocks = 256 * wpb_blocks;   // wpb_blocks 256-thread blocks per CU -> that many waves per SIMD
    const int iters = 2000;
    uint32_t *d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: each block = 4 waves on 4 SIMDs; wpb_blocks blocks per CU
    const double insts_per_simd = (double)iters * valu_per_iter * wpb_blocks;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-34s waves/SIMD=%d  %.3f ms  cycles per wave-instr per SIMD (at 2.4 GHz) = %.2f\n", name, wpb_blocks, ms, cycles / insts_per_simd);
    hipFree(d);
    return ms;
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_xor_b32", w, 64);
        run<11>("v_fma_f32", w, 64);
    }
    const int w = 8;
    run<1>("v_and_or_b32", w, 64);
    run<2>("v_bfi_b32", w, 64);
    run<3>("v_bfe_i32", w, 64);
    run<4>("v_add_u32", w, 64);
    run<5>("v_lshl_add_u64", w, 64);
    run<6>("v_lshrrev_b64", w, 64);
    run<7>("v_alignbit_b32", w, 64);
    run<8>("v_add_co/addc pair (per instr)", w, 64);
    run<9>("v_or3_b32", w, 64);
    run<10>("v_pk_add_u16", w, 64);
    run<12>("v_xor_b32 v,v,sgpr", w, 64);
    run<13>("v_cndmask_b32", w, 64);
    run<14>("v_max_i32", w, 64);
    run<15>("v_pk_max_i16", w, 64);
    run<16>("v_max3_i32", w, 64);
    run<17>("v_xor_b32 + 1 SALU each (per VALU)", w, 64);
    return 0;
}
