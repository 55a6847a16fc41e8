// time two bodies of the digit-form Fp2 product in isolation, one wave per SIMD (dev tool): four plain column scans against
// Karatsuba on shared column sums (3 products, 64-bit combinations per column). python3 scripts/dbg/gen_fp2d_variants.py writes the
// include file. The Karatsuba body uses block 8 and v108..v111 as scratch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "fp2d_variants.inc"
#define ALLV "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139"
#define ALLS "s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","vcc","scc"
template <int V> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;
    asm volatile(LOAD_CONST ::: ALLS);
    // operands: digits below 2^28 in blocks 0..3
    for (int b = 0; b < 4; b++) {
        // (register-indexed initialisation through asm: one mov per digit)
    }
    asm volatile("v_mov_b32 v0, %0\n\tv_and_b32 v0, 0x0fffffff, v0" :: "v"(seed) : "v0");
#define INIT(r, k) asm volatile("v_mul_lo_u32 v" #r ", v0, %0\n\tv_and_b32 v" #r ", 0x0fffffff, v" #r :: "s"(k) : "v" #r);
    INIT(1, 3) INIT(2, 5) INIT(3, 7) INIT(4, 11) INIT(5, 13) INIT(6, 17) INIT(7, 19) INIT(8, 23) INIT(9, 29) INIT(10, 31) INIT(11, 37) INIT(12, 41) INIT(13, 3)
    INIT(14, 43) INIT(15, 47) INIT(16, 53) INIT(17, 59) INIT(18, 61) INIT(19, 67) INIT(20, 71) INIT(21, 73) INIT(22, 79) INIT(23, 83) INIT(24, 89) INIT(25, 97) INIT(26, 101) INIT(27, 5)
    INIT(28, 103) INIT(29, 107) INIT(30, 109) INIT(31, 113) INIT(32, 127) INIT(33, 131) INIT(34, 137) INIT(35, 139) INIT(36, 149) INIT(37, 151) INIT(38, 157) INIT(39, 163) INIT(40, 167) INIT(41, 7)
    INIT(42, 173) INIT(43, 179) INIT(44, 181) INIT(45, 191) INIT(46, 193) INIT(47, 197) INIT(48, 199) INIT(49, 211) INIT(50, 223) INIT(51, 227) INIT(52, 229) INIT(53, 233) INIT(54, 239) INIT(55, 9)
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(FP2_MUL_D_OLD ::: ALLV, ALLS);
        else asm volatile(FP2_MUL_D_KARA ::: ALLV, ALLS);
        // results (blocks 5, 6) become the next a operand (blocks 0, 1); digit 13 is cut back to 28 bits to stay inside the limits
        asm volatile("v_mov_b32 v0, v70\n\tv_mov_b32 v1, v71\n\tv_mov_b32 v2, v72\n\tv_mov_b32 v3, v73\n\tv_mov_b32 v4, v74\n\tv_mov_b32 v5, v75\n\tv_mov_b32 v6, v76\n\t"
                     "v_mov_b32 v7, v77\n\tv_mov_b32 v8, v78\n\tv_mov_b32 v9, v79\n\tv_mov_b32 v10, v80\n\tv_mov_b32 v11, v81\n\tv_mov_b32 v12, v82\n\tv_and_b32 v13, 0x0fffffff, v83\n\t"
                     "v_mov_b32 v14, v84\n\tv_mov_b32 v15, v85\n\tv_mov_b32 v16, v86\n\tv_mov_b32 v17, v87\n\tv_mov_b32 v18, v88\n\tv_mov_b32 v19, v89\n\tv_mov_b32 v20, v90\n\t"
                     "v_mov_b32 v21, v91\n\tv_mov_b32 v22, v92\n\tv_mov_b32 v23, v93\n\tv_mov_b32 v24, v94\n\tv_mov_b32 v25, v95\n\tv_mov_b32 v26, v96\n\tv_and_b32 v27, 0x0fffffff, v97" ::: ALLV);
    }
    uint32_t x;
    asm volatile("v_xor_b32 %0, v70, v84\n\tv_xor_b32 %0, %0, v75\n\tv_xor_b32 %0, %0, v90" : "=v"(x) :: ALLV);
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
template <int V> void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        int blocks = 1024, iters = 20000;
        hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, 10);
        hipEventRecord(e0); hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint32_t h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("%-10s %.3f ms  %.1f ns per product per wave  check %08x\n", name, ms, ms * 1e6 / iters, h[5]);
    }
}
int main() { uint32_t* d; hipMalloc(&d, 1024 * 64 * 4); run<0>("4 scans", d); run<1>("karatsuba", d); run<0>("4 scans", d); return 0; }
