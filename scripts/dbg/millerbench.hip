// time the generated Miller doubling iteration in isolation (dev tool): full / without the multiplication calls / calls only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../milagro_bls_amd/csrc/mbls_fp_asm.inc"
#include "../../milagro_bls_amd/csrc/mbls_tower_asm.inc"
#include "miller_variants.inc"
#define CALLASM(sym) "s_getpc_b64 s[40:41]\n\ts_add_u32 s40, s40, " sym "@rel32@lo+4\n\ts_addc_u32 s41, s41, " sym "@rel32@hi+12\n\ts_swappc_b64 s[30:31], s[40:41]"
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_mul_asm_fn() { asm volatile(MBLS_FP2_MUL_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_sqr_asm_fn() { asm volatile(MBLS_FP2_SQR_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_mulfp_asm_fn() { asm volatile(MBLS_FP2_MULFP_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void v_full() { asm volatile(MBLS_MILLER_DBL_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void v_nocall() { asm volatile(MILLER_NOCALL); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void v_callonly() { asm volatile(MILLER_CALLONLY); }
__shared__ uint32_t lds[160 * 64];
template <int V> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    for (int i = threadIdx.x; i < 160 * 64; i += 64) lds[i] = i * 2654435761u >> 4;
    __syncthreads();
    uint32_t addr = (uint32_t)(uintptr_t)(lds + threadIdx.x), flags = 0, n = iters;
    if (V == 0) asm volatile(CALLASM("v_full") :: "{v252}"(addr), "{v253}"(flags), "{s38}"(n) : MBLS_TOWER_ASM_CLOBBERS);
    if (V == 1) asm volatile(CALLASM("v_nocall") :: "{v252}"(addr), "{v253}"(flags), "{s38}"(n) : MBLS_TOWER_ASM_CLOBBERS);
    if (V == 2) asm volatile(CALLASM("v_callonly") :: "{v252}"(addr), "{v253}"(flags), "{s38}"(n) : MBLS_TOWER_ASM_CLOBBERS);
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
int main() { uint32_t* d; hipMalloc(&d, 1024 * 8 * 64 * 4); run<0>("full", d); run<1>("nocall", d); run<2>("callonly", d); return 0; }
