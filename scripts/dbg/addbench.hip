// dependent vs interleaved carry chains for a lone wave (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "addbench_gen.inc"
template <int V> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(ADD_SEQ ::: ADD_CLOB);
        else if (V == 1) asm volatile(ADD_ILV ::: ADD_CLOB);
        else asm volatile(ADD_SEQ_NOP ::: ADD_CLOB);
    }
    out[blockIdx.x * 64 + threadIdx.x] = 0;
}
template <int V> void run(const char* name, uint32_t* d, int ninst) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; wps *= 2) {
        int blocks = 1024 * wps, iters = 20000;
        hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, 1000);
        hipEventRecord(e0); hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s waves/SIMD=%d %.3f ms  %.2f clk/inst (2.33 GHz)\n", name, wps, ms, ms * 1e-3 * 2.33e9 / iters / wps / ninst);
    }
}
int main() { uint32_t* d; hipMalloc(&d, 1024 * 8 * 64 * 4); run<0>("sequential", d, NINST); run<1>("interleaved", d, NINST); run<2>("seq+nops", d, NINST); return 0; }
