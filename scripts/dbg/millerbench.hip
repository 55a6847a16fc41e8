// time the generated Miller doubling iteration in isolation (dev tool): full / without the multiplication calls / calls only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../milagro_bls_amd/csrc/mbls_fp_asm.inc"
#include "../../milagro_bls_amd/csrc/mbls_tower_asm.inc"
#include "miller_variants.inc"
#include "mac28.inc"
#include "call_variants.inc"
#define CALLASM(sym) "s_getpc_b64 s[40:41]\n\ts_add_u32 s40, s40, " sym "@rel32@lo+4\n\ts_addc_u32 s41, s41, " sym "@rel32@hi+12\n\ts_swappc_b64 s[30:31], s[40:41]"
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_mul_asm_fn() {
#ifdef USE28
 asm volatile(MUL28);
#else
 asm volatile(MBLS_FP2_MUL_ASM);
#endif
}
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_sqr_asm_fn() {
#ifdef USE28
 asm volatile(SQR28);
#else
 asm volatile(MBLS_FP2_SQR_ASM);
#endif
}
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_mulfp_asm_fn() {
#ifdef USE28
 asm volatile(MULFP28);
#else
 asm volatile(MBLS_FP2_MULFP_ASM);
#endif
}
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void v_full() { asm volatile(MBLS_MILLER_DBL_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void v_nocall() { asm volatile(MILLER_NOCALL); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void v_callonly() { asm volatile(MILLER_CALLONLY); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void v_mul() { asm volatile(ONLY_MUL); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void v_sqr() { asm volatile(ONLY_SQR); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void v_mulfp() { asm volatile(ONLY_MULFP); }
__shared__ uint32_t lds[160 * 64];
#define SMALL_CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","s30","s31","s36","s37","s39","s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","s62","s63","vcc","scc","memory"
__global__ void __launch_bounds__(64) kern_small(uint32_t* out, int iters) {
    uint32_t n = iters;
    asm volatile(CALLASM("v_mul") :: "{s38}"(n) : SMALL_CLOB, "s64", "s65", "v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117");
    out[blockIdx.x * 64 + threadIdx.x] = 0;
}
template <int V> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    for (int i = threadIdx.x; i < 160 * 64; i += 64) lds[i] = i * 2654435761u >> 4;
    __syncthreads();
    uint32_t addr = (uint32_t)(uintptr_t)(lds + threadIdx.x), flags = 0, n = iters;
    if (V == 0) asm volatile(CALLASM("v_full") :: "{v252}"(addr), "{v253}"(flags), "{s38}"(n) : MBLS_TOWER_ASM_CLOBBERS, "s64", "s65");
    if (V == 1) asm volatile(CALLASM("v_nocall") :: "{v252}"(addr), "{v253}"(flags), "{s38}"(n) : MBLS_TOWER_ASM_CLOBBERS, "s64", "s65");
    if (V == 2) asm volatile(CALLASM("v_callonly") :: "{v252}"(addr), "{v253}"(flags), "{s38}"(n) : MBLS_TOWER_ASM_CLOBBERS, "s64", "s65");
    if (V == 3) asm volatile(CALLASM("v_mul") :: "{v252}"(addr), "{v253}"(flags), "{s38}"(n) : MBLS_TOWER_ASM_CLOBBERS, "s64", "s65");
    if (V == 4) asm volatile(CALLASM("v_sqr") :: "{v252}"(addr), "{v253}"(flags), "{s38}"(n) : MBLS_TOWER_ASM_CLOBBERS, "s64", "s65");
    if (V == 5) asm volatile(CALLASM("v_mulfp") :: "{v252}"(addr), "{v253}"(flags), "{s38}"(n) : MBLS_TOWER_ASM_CLOBBERS, "s64", "s65");
    out[blockIdx.x * 64 + threadIdx.x] = lds[threadIdx.x];
}
template <int V> void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 1; wps *= 2) {
        int blocks = 1024 * wps, iters = 200;
        hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, 20);
        hipEventRecord(e0); hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s %.3f ms  %.0f ticks/iteration (2.33 GHz)\n", name, ms, ms * 1e-3 * 2.33e9 / iters);
    }
}
void run_small(uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 1024, iters = 200;
    hipLaunchKernelGGL(kern_small, dim3(blocks), dim3(64), 0, 0, d, 20);
    hipEventRecord(e0); hipLaunchKernelGGL(kern_small, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("60 mul, <128 regs  %.3f ms  %.0f ticks/iteration\n", ms, ms * 1e-3 * 2.33e9 / iters);
}
int main() { uint32_t* d; hipMalloc(&d, 1024 * 8 * 64 * 4); run<0>("full", d); run<1>("nocall", d); run<2>("callonly", d); run<3>("60 mul", d); run<4>("60 sqr", d); run<5>("60 mulfp", d); run_small(d); return 0; }
