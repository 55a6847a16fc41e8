// What does a second wave per SIMD buy for the instruction MIX of the generated kernels? (dev probe; profiles/r05_two_wave_mix.txt)
// Loop body = the digit-form Fp2 product routine (980 multiply-accumulates of 1 281 instructions) + a block of the other classes in the proportions of
// k_miller (64.8 % multiply-accumulates / 16.0 % carry3 / 19.1 % simple2) or k_final (64.3 / 12.9 / 22.8) -- tools/instr_census.py. 140 registers, no LDS:
// the mix without the register pressure that keeps the real kernels at one wave per SIMD. python3 scripts/dbg/gen_mixbench.py writes the include file.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "mixbench_gen.inc"
#define V10(a) "v" #a "0","v" #a "1","v" #a "2","v" #a "3","v" #a "4","v" #a "5","v" #a "6","v" #a "7","v" #a "8","v" #a "9"
#define ALLV "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9",V10(1),V10(2),V10(3),V10(4),V10(5),V10(6),V10(7),V10(8),V10(9),V10(10),V10(11),V10(12),V10(13)
#define A10(a) "a" #a "0","a" #a "1","a" #a "2","a" #a "3","a" #a "4","a" #a "5","a" #a "6","a" #a "7","a" #a "8","a" #a "9"
#define ALLA "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9",A10(1),A10(2)
#define ALLS "s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","vcc","scc"
template <int MIX> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;
    asm volatile(LOAD_CONST ::: ALLS);
    asm volatile("v_mov_b32 v0, %0\n\tv_and_b32 v0, 0x0fffffff, v0" :: "v"(seed) : "v0");
#define INIT(r, k) asm volatile("v_mul_lo_u32 v" #r ", v0, %0\n\tv_and_b32 v" #r ", 0x0fffffff, v" #r :: "s"(k) : "v" #r);
    INIT(1, 3) INIT(2, 5) INIT(3, 7) INIT(4, 11) INIT(5, 13) INIT(6, 17) INIT(7, 19) INIT(8, 23) INIT(9, 29) INIT(10, 31) INIT(11, 37) INIT(12, 41) INIT(13, 3)
    INIT(14, 43) INIT(15, 47) INIT(16, 53) INIT(17, 59) INIT(18, 61) INIT(19, 67) INIT(20, 71) INIT(21, 73) INIT(22, 79) INIT(23, 83) INIT(24, 89) INIT(25, 97) INIT(26, 101) INIT(27, 5)
    INIT(28, 103) INIT(29, 107) INIT(30, 109) INIT(31, 113) INIT(32, 127) INIT(33, 131) INIT(34, 137) INIT(35, 139) INIT(36, 149) INIT(37, 151) INIT(38, 157) INIT(39, 163) INIT(40, 167) INIT(41, 7)
    INIT(42, 173) INIT(43, 179) INIT(44, 181) INIT(45, 191) INIT(46, 193) INIT(47, 197) INIT(48, 199) INIT(49, 211) INIT(50, 223) INIT(51, 227) INIT(52, 229) INIT(53, 233) INIT(54, 239) INIT(55, 9)
    for (int i = 0; i < iters; i++) {
        asm volatile(FP2_MUL_D ::: ALLV, ALLS);
        if (MIX == 1) asm volatile(MIX_MILLER ::: ALLV, ALLA, ALLS);
        if (MIX == 2) asm volatile(MIX_FINAL ::: ALLV, ALLA, ALLS);
        asm volatile("v_mov_b32 v0, v70\n\tv_mov_b32 v1, v71\n\tv_mov_b32 v2, v72\n\tv_mov_b32 v3, v73\n\tv_mov_b32 v4, v74\n\tv_mov_b32 v5, v75\n\tv_mov_b32 v6, v76\n\t"
                     "v_mov_b32 v7, v77\n\tv_mov_b32 v8, v78\n\tv_mov_b32 v9, v79\n\tv_mov_b32 v10, v80\n\tv_mov_b32 v11, v81\n\tv_mov_b32 v12, v82\n\tv_and_b32 v13, 0x0fffffff, v83" ::: ALLV);
    }
    uint32_t x;
    asm volatile("v_xor_b32 %0, v70, v84\n\tv_xor_b32 %0, %0, v75\n\tv_xor_b32 %0, %0, v112\n\tv_xor_b32 %0, %0, v127" : "=v"(x) :: ALLV);
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
template <int MIX> void run(const char* name, int per_iter, uint32_t* d, int cus) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double base = 0;
    for (int w = 1; w <= 3; w++) {           // waves per SIMD (140 registers: three fit)
        int blocks = cus * 4 * w, iters = 4000;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(kern<MIX>, dim3(blocks), dim3(64), 0, 0, d, 50);
            hipEventRecord(e0); hipLaunchKernelGGL(kern<MIX>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double ns_per_instr = best * 1e6 / iters / per_iter / w;          // SIMD time per wave-instruction
        if (w == 1) base = ns_per_instr;
        printf("%-34s %d wave(s) per SIMD: %8.3f ms  %.3f ns of SIMD time per wave-instruction  (%.1f %% of one wave)\n", name, w, best, ns_per_instr, 100 * ns_per_instr / base);
    }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    uint32_t* d; hipMalloc(&d, (size_t)p.multiProcessorCount * 4 * 3 * 64 * 4);
    run<0>("product routine alone (76.5/17/6.5)", 1281 + 14, d, p.multiProcessorCount);
    run<1>("k_miller mix (64.8/16.0/19.1)", 1281 + 230 + 14, d, p.multiProcessorCount);
    run<2>("k_final mix (64.3/12.9/22.8)", 1281 + 263 + 14, d, p.multiProcessorCount);
    return 0;
}
