// Issue cost of the vector instructions the band kernels are made of (gfx950): N independent chains per wave, W waves per SIMD, every
// CU busy.  Prints cycles per instruction per SIMD (at the clock measured with s_memtime) -- 2.0 = the SIMD-32 rate for wave64.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(uint32_t *out, int iters, long long *cycles)
{
    uint32_t a0 = threadIdx.x, a1 = threadIdx.x * 3u, a2 = 7u, a3 = 11u, b0 = blockIdx.x, b1 = 5u, b2 = 9u, b3 = 13u;
    uint64_t q0 = a0, q1 = a1 + 1, q2 = 77, q3 = 99;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {          // v_and_b32, 4 independent chains
            REP64(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
        } else if (KIND == 1) {   // v_bitop3_b32
            REP64(asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0xf1\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0xf1\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0xf1\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xf1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
        } else if (KIND == 2) {   // v_lshrrev_b64
            REP64(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
        } else if (KIND == 3) {   // v_lshl_add_u64
            REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q3));)
        } else if (KIND == 4) {   // v_alignbit_b32
            REP64(asm volatile("v_alignbit_b32 %0, %0, %4, 1\n v_alignbit_b32 %1, %1, %4, 1\n v_alignbit_b32 %2, %2, %4, 1\n v_alignbit_b32 %3, %3, %4, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
        } else if (KIND == 5) {   // v_add_co + v_addc_co pairs (two chains)
            REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");)
        } else if (KIND == 6) {   // v_and_b32, ONE dependent chain
            REP64(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %0, %0, %1\n v_and_b32 %0, %0, %1\n v_and_b32 %0, %0, %1" : "+v"(a0) : "v"(b0));)
        } else if (KIND == 7) {   // v_bitop3_b32, ONE dependent chain
            REP64(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1" : "+v"(a0) : "v"(b0), "v"(b1));)
        } else if (KIND == 8) {   // v_lshrrev_b32 (32-bit shift)
            REP64(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 9) {   // v_mad_u32_u24
            REP64(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
        } else if (KIND == 10) {  // v_bfe_u32
            REP64(asm volatile("v_bfe_u32 %0, %0, 4, 3\n v_bfe_u32 %1, %1, 4, 3\n v_bfe_u32 %2, %2, 4, 3\n v_bfe_u32 %3, %3, 4, 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 11) {  // v_add_u32
            REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
        } else if (KIND == 12) {  // v_bcnt_u32_b32
            REP64(asm volatile("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
        } else if (KIND == 13) {  // v_mul_u32_u24 with a byte of its source selected (SDWA)
            REP64(asm volatile("v_mul_u32_u24_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_u32_u24_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_mul_u32_u24_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_mul_u32_u24_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
        } else if (KIND == 14) {  // v_mul_u32_u24 (VOP2)
            REP64(asm volatile("v_mul_u32_u24_e32 %0, %0, %4\n v_mul_u32_u24_e32 %1, %1, %4\n v_mul_u32_u24_e32 %2, %2, %4\n v_mul_u32_u24_e32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
        } else if (KIND == 15) {  // v_add_u32 with a byte of its source selected (SDWA)
            REP64(asm volatile("v_add_u32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_add_u32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_add_u32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
        } else if (KIND == 16) {  // v_or3_b32
            REP64(asm volatile("v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %4, %5\n v_or3_b32 %2, %2, %4, %5\n v_or3_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
        } else if (KIND == 17) {  // v_lshl_add_u32
            REP64(asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
        } else if (KIND == 18) {  // v_and_b32 with a 32-bit literal
            REP64(asm volatile("v_and_b32 %0, 0x0f0f0f0f, %0\n v_and_b32 %1, 0x0f0f0f0f, %1\n v_and_b32 %2, 0x0f0f0f0f, %2\n v_and_b32 %3, 0x0f0f0f0f, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 19) {  // v_xor3? (v_xad / v_add3)
            REP64(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(q0 + q1 + q2 + q3) + b2 + b3;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = t1 - t0;
}

template <int KIND>
void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 1500;
    uint32_t *d; long long *dc;
    hipMalloc(&d, (size_t)blocks * 256 * 4); hipMalloc(&dc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_issue<KIND><<<blocks, 256>>>(d, 2, dc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_issue<KIND><<<blocks, 256>>>(d, iters, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    const double n_instr = (double)iters * 64 * 4;          // per wave
    // s_memtime / readcyclecounter ticks at a fixed 100 MHz on this part?  report both: wall-clock based (2.4 GHz nominal) and counter based
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f cycles/instr/SIMD at 2.4 GHz nominal (counter: %.2f ticks/instr/wave)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / (n_instr * waves_per_simd), (double)cyc / n_instr);
    hipFree(d); hipFree(dc);
}

int main()
{
    for (int w : {1, 2, 4, 6}) {
        run<0>("v_and_b32 x4 chains", w);
        run<6>("v_and_b32 one dependent chain", w);
        run<1>("v_bitop3_b32 x4 chains", w);
        run<7>("v_bitop3_b32 one dependent chain", w);
        run<2>("v_lshrrev_b64 x4", w);
        run<8>("v_lshrrev_b32 x4", w);
        run<3>("v_lshl_add_u64 x4", w);
        run<5>("v_add_co + v_addc_co x2", w);
        run<4>("v_alignbit_b32 x4", w);
        run<9>("v_mad_u32_u24 x4", w);
        run<10>("v_bfe_u32 x4", w);
        run<11>("v_add_u32 x4", w);
        run<12>("v_bcnt_u32_b32 x4", w);
        run<13>("v_mul_u32_u24_sdwa (byte) x4", w);
        run<14>("v_mul_u32_u24_e32 x4", w);
        run<15>("v_add_u32_sdwa (byte) x4", w);
        run<16>("v_or3_b32 x4", w);
        run<17>("v_lshl_add_u32 x4", w);
        run<18>("v_and_b32 literal x4", w);
        run<19>("v_add3_u32 x4", w);
    }
    return 0;
}
