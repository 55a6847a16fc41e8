// time the fused Fp2 multiplication routines in isolation (dev tool): old sequential scans vs interleaved dual chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../milagro_bls_amd/csrc/mbls_fp_asm.inc"
#include "fp_asm_old.inc"
typedef uint32_t fp __attribute__((ext_vector_type(12), aligned(16)));
template <int V> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    fp a0, a1, b0, b1, c0, c1;
    for (int j = 0; j < 12; j++) { a0[j] = threadIdx.x * 77 + j; a1[j] = threadIdx.x * 31 + j * 5; b0[j] = blockIdx.x + j; b1[j] = j * 3 + 1; }
    a0[11] &= 0xffffff; a1[11] &= 0xffffff; b0[11] &= 0xffffff; b1[11] &= 0xffffff;
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(OLD_FP2_MUL_ASM : "={v[48:59]}"(c0), "={v[60:71]}"(c1) : "{v[0:11]}"(a0), "{v[12:23]}"(a1), "{v[24:35]}"(b0), "{v[36:47]}"(b1) : OLD_FP2_MUL_CLOBBERS);
        else asm volatile(MBLS_FP2_MUL_ASM : "={v[48:59]}"(c0), "={v[60:71]}"(c1) : "{v[0:11]}"(a0), "{v[12:23]}"(a1), "{v[24:35]}"(b0), "{v[36:47]}"(b1) : MBLS_FP2_MUL_CLOBBERS);
        a0 = c0; a1 = c1;
    }
    uint32_t x = 0; for (int j = 0; j < 12; j++) x ^= a0[j] ^ a1[j];
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
template <int V> void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 2000; it <= 200000; it *= 10) {
        int wps = 1; int blocks = 1024 * wps, iters = it;
        hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, 10);
        hipEventRecord(e0); hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint32_t h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("%-12s waves/SIMD=%d %.3f ms  %.1f clk/fp2_mul/wave  check %08x\n", name, wps, ms, ms * 1e-3 * 2.4e9 / iters / wps, h[5]);
    }
}
int main() { uint32_t* d; hipMalloc(&d, 1024 * 8 * 64 * 4); run<0>("old", d); run<1>("new", d); return 0; }
