// instruction-throughput microbenchmarks for gfx950 integer multiply paths (dev tool, not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int MODE> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters, uint32_t seed, unsigned long long* clk) {
    unsigned long long t0 = clock64(), w0 = wall_clock64();
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    float f0 = a, f1 = b, f2 = 1.5f, f3 = 2.5f, f4 = 1.0000001f, f5 = 0.25f;
    double d0 = a, d1 = b, d2 = a + 1.0, d3 = b + 1.0, d4 = 1.5, d5 = 2.5, d6 = 3.5, d7 = 4.5;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 8 independent v_mad_u64_u32 chains
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                               "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");)
        } else if (MODE == 1) {   // 1 dependent chain
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                               "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                               : "+v"(c0) : "v"(a), "v"(b) : "vcc");)
        } else if (MODE == 2) {   // mac = mad + addc, single dependent chain (as in fp_mul now)
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n"
                               "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n"
                               : "+v"(c0), "+v"(h0) : "v"(a), "v"(b) : "vcc");)
        } else if (MODE == 3) {   // mac, two interleaved chains using distinct carry registers (s[..] via vcc and explicit sgpr pair)
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %2, s[20:21], %4, %5, %2\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_addc_co_u32_e64 %3, s[20:21], 0, %3, s[20:21]\n"
                               "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %2, s[20:21], %4, %5, %2\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_addc_co_u32_e64 %3, s[20:21], 0, %3, s[20:21]\n"
                               : "+v"(c0), "+v"(h0), "+v"(c1), "+v"(h1) : "v"(a), "v"(b) : "vcc", "s20", "s21");)
        } else if (MODE == 4) {   // v_mul_lo_u32 + v_mul_hi_u32 independent
            REP16(asm volatile("v_mul_lo_u32 %0, %4, %5\n v_mul_hi_u32 %1, %4, %5\n v_mul_lo_u32 %2, %4, %5\n v_mul_hi_u32 %3, %4, %5\n v_mul_lo_u32 %0, %4, %5\n v_mul_hi_u32 %1, %4, %5\n v_mul_lo_u32 %2, %4, %5\n v_mul_hi_u32 %3, %4, %5\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));)
        } else if (MODE == 5) {   // v_mad_u32_u24 independent (8 per group)
            REP16(asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3\n v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));)
        } else if (MODE == 6) {   // v_add_co_u32 / v_addc chain (8)
            REP16(asm volatile("v_add_co_u32_e32 %0, vcc, %4, %0\n v_addc_co_u32_e32 %1, vcc, %5, %1, vcc\n v_addc_co_u32_e32 %2, vcc, %4, %2, vcc\n v_addc_co_u32_e32 %3, vcc, %5, %3, vcc\n v_add_co_u32_e32 %0, vcc, %4, %0\n v_addc_co_u32_e32 %1, vcc, %5, %1, vcc\n v_addc_co_u32_e32 %2, vcc, %4, %2, vcc\n v_addc_co_u32_e32 %3, vcc, %5, %3, vcc\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : "vcc");)
        } else if (MODE == 7) {   // v_fma_f64 8 independent
            REP16(asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(d0 * 0 + 1.0000001), "v"(d1 * 0 + 0.5));)
        } else if (MODE == 8) {   // v_lshl_add_u64 independent (8)
            REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(c4));)
        } else if (MODE == 9) {   // v_mad_u64_u32 with SGPR carry-out dst (not vcc), 8 independent, to see if vcc matters
            REP16(asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[22:23], %8, %9, %1\n v_mad_u64_u32 %2, s[24:25], %8, %9, %2\n v_mad_u64_u32 %3, s[26:27], %8, %9, %3\n"
                               "v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[22:23], %8, %9, %5\n v_mad_u64_u32 %6, s[24:25], %8, %9, %6\n v_mad_u64_u32 %7, s[26:27], %8, %9, %7\n"
                               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (MODE == 10) {  // v_add_u32 (no carry), 8 independent
            REP16(asm volatile("v_add_u32_e32 %0, %4, %0\n v_add_u32_e32 %1, %5, %1\n v_add_u32_e32 %2, %4, %2\n v_add_u32_e32 %3, %5, %3\n v_add_u32_e32 %0, %5, %0\n v_add_u32_e32 %1, %4, %1\n v_add_u32_e32 %2, %5, %2\n v_add_u32_e32 %3, %4, %3\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));)
        } else if (MODE == 11) {  // v_add3_u32
            REP16(asm volatile("v_add3_u32 %0, %4, %5, %0\n v_add3_u32 %1, %4, %5, %1\n v_add3_u32 %2, %4, %5, %2\n v_add3_u32 %3, %4, %5, %3\n v_add3_u32 %0, %4, %5, %0\n v_add3_u32 %1, %4, %5, %1\n v_add3_u32 %2, %4, %5, %2\n v_add3_u32 %3, %4, %5, %3\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));)
        } else if (MODE == 12) {  // v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f4), "v"(f5));)
        } else if (MODE == 13) {  // v_and_b32
            REP16(asm volatile("v_and_b32_e32 %0, %4, %0\n v_and_b32_e32 %1, %5, %1\n v_and_b32_e32 %2, %4, %2\n v_and_b32_e32 %3, %5, %3\n v_and_b32_e32 %0, %5, %0\n v_and_b32_e32 %1, %4, %1\n v_and_b32_e32 %2, %5, %2\n v_and_b32_e32 %3, %4, %3\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));)
        } else if (MODE == 14) {  // v_pk_fma_f32 (two f32 FMAs per lane)
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4), "v"(d5));)
        } else if (MODE == 15) {  // v_add_f32
            REP16(asm volatile("v_add_f32_e32 %0, %4, %0\n v_add_f32_e32 %1, %5, %1\n v_add_f32_e32 %2, %4, %2\n v_add_f32_e32 %3, %5, %3\n v_add_f32_e32 %0, %5, %0\n v_add_f32_e32 %1, %4, %1\n v_add_f32_e32 %2, %5, %2\n v_add_f32_e32 %3, %4, %3\n"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f4), "v"(f5));)
        } else if (MODE == 16) {  // v_mov_b32
            REP16(asm volatile("v_mov_b32_e32 %0, %4\n v_mov_b32_e32 %1, %5\n v_mov_b32_e32 %2, %4\n v_mov_b32_e32 %3, %5\n v_mov_b32_e32 %0, %5\n v_mov_b32_e32 %1, %4\n v_mov_b32_e32 %2, %5\n v_mov_b32_e32 %3, %4\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));)
        } else if (MODE == 17) {  // v_accvgpr_write / read pairs
            REP16(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n v_accvgpr_read_b32 %0, a1\n v_accvgpr_read_b32 %1, a2\n v_accvgpr_read_b32 %2, a3\n v_accvgpr_read_b32 %3, a0\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : : "a0", "a1", "a2", "a3");)
        } else if (MODE == 18) {  // v_lshl_or_b32 / v_alignbit_b32 / v_bfe_u32 mix (digit conversion ops)
            REP16(asm volatile("v_lshl_or_b32 %0, %4, 4, %0\n v_alignbit_b32 %1, %4, %1, 5\n v_bfe_u32 %2, %2, 3, 28\n v_lshrrev_b32_e32 %3, 1, %3\n v_lshl_or_b32 %0, %5, 4, %0\n v_alignbit_b32 %1, %5, %1, 5\n v_bfe_u32 %2, %2, 1, 28\n v_lshrrev_b32_e32 %3, 1, %3\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));)
        } else if (MODE == 19) {  // v_cndmask_b32 with vcc
            REP16(asm volatile("v_cndmask_b32_e32 %0, %4, %0, vcc\n v_cndmask_b32_e32 %1, %5, %1, vcc\n v_cndmask_b32_e32 %2, %4, %2, vcc\n v_cndmask_b32_e32 %3, %5, %3, vcc\n v_cndmask_b32_e32 %0, %5, %0, vcc\n v_cndmask_b32_e32 %1, %4, %1, vcc\n v_cndmask_b32_e32 %2, %5, %2, vcc\n v_cndmask_b32_e32 %3, %4, %3, vcc\n"
                               : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : "vcc");)
        } else if (MODE == 20) {  // v_mad_u64_u32 with 28-bit operands (as in the digit routines), 8 independent
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                               "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a & 0xFFFFFFF), "v"(b & 0xFFFFFFF) : "vcc");)
        }
    }
    unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (clk && threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(f0 + f1 + f2 + f3) ^ (uint32_t)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) ^ h0 ^ h1 ^ h2 ^ h3 ^ (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int MODE> void run(const char* name, int instr_per_iter, uint32_t* d_out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static unsigned long long* d_clk = nullptr; static unsigned long long* h_clk = nullptr;
    if (!d_clk) { hipMalloc(&d_clk, 256 * 4 * 8 * 16); h_clk = (unsigned long long*)malloc(256 * 4 * 8 * 16); }
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * 4 * wps, iters = 2000;
        hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 10, 1u, (unsigned long long*)nullptr);
        hipEventRecord(e0); hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 2u, d_clk); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h_clk, d_clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0; for (int b = 0; b < blocks; b++) { cyc += (double)h_clk[2 * b]; rt += (double)h_clk[2 * b + 1]; }
        cyc /= blocks; rt /= blocks;
        double mhz = cyc / (rt / 100.0);                       // s_memrealtime ticks at 100 MHz
        double inst = (double)blocks * iters * instr_per_iter;   // wave-instructions
        double per_simd_per_s = inst / 1024.0 / (ms * 1e-3);
        double clk_issue = cyc / ((double)iters * instr_per_iter * wps);   // shader clocks of SIMD issue time per wave-instruction
        printf("%-30s waves/SIMD=%d  %6.2f ms  %.3f Gwave-inst/s/SIMD  %5.2f clk/inst @2.4GHz | in-kernel: %4.0f MHz, %5.2f clk/inst (s_memtime)\n",
               name, wps, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, mhz, clk_issue);
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    run<0>("mad_u64_u32 x8 indep", 128, d);
    run<20>("mad_u64_u32 x8 indep 28-bit ops", 128, d);
    run<1>("mad_u64_u32 dependent", 128, d);
    run<2>("mad+addc dependent (mac)", 128, d);
    run<3>("mac two chains interleaved", 128, d);
    run<4>("mul_lo/mul_hi", 128, d);
    run<5>("mad_u32_u24", 128, d);
    run<6>("add_co/addc chain", 128, d);
    run<7>("fma_f64", 128, d);
    run<8>("lshl_add_u64", 128, d);
    run<9>("mad_u64_u32 sgpr carry", 128, d);
    run<10>("v_add_u32", 128, d);
    run<11>("v_add3_u32", 128, d);
    run<12>("v_fma_f32", 128, d);
    run<15>("v_add_f32", 128, d);
    run<14>("v_pk_fma_f32", 128, d);
    run<13>("v_and_b32", 128, d);
    run<16>("v_mov_b32", 128, d);
    run<17>("v_accvgpr_write/read", 128, d);
    run<18>("lshl_or/alignbit/bfe/lshr", 128, d);
    run<19>("v_cndmask_b32 vcc", 128, d);
    return 0;
}
