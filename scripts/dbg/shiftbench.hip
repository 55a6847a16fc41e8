// issue cost of v_lshrrev_b64 vs alignbit+lshr, v_mad_u64_u32 and v_lshl_add_u64 for a lone wave (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))
template <int V> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(R64("v_lshrrev_b64 v[10:11], 28, v[10:11]\n v_lshrrev_b64 v[12:13], 28, v[12:13]\n") ::: "v10","v11","v12","v13");
        if (V == 1) asm volatile(R64("v_alignbit_b32 v10, v11, v10, 28\n v_lshrrev_b32_e64 v11, 28, v11\n") ::: "v10","v11");
        if (V == 2) asm volatile(R64("v_mad_u64_u32 v[10:11], vcc, v14, v15, v[10:11]\n v_mad_u64_u32 v[12:13], s[62:63], v14, v15, v[12:13]\n") ::: "v10","v11","v12","v13","vcc","s62","s63");
        if (V == 3) asm volatile(R64("v_and_b32_e64 v10, v11, v12\n v_and_b32_e64 v13, v11, v12\n") ::: "v10","v13");
        if (V == 4) asm volatile(R64("v_mul_lo_u32 v10, v11, s20\n v_mul_lo_u32 v13, v11, s20\n") ::: "v10","v13");
        if (V == 5) asm volatile(R64("v_mad_u64_u32 v[10:11], vcc, s20, v15, v[10:11]\n v_mad_u64_u32 v[12:13], s[62:63], s20, v15, v[12:13]\n") ::: "v10","v11","v12","v13","vcc","s62","s63");
    }
    out[blockIdx.x * 64 + threadIdx.x] = 0;
}
template <int V> void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 1024, iters = 20000;
    hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, 1000);
    hipEventRecord(e0); hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.3f ms  %.2f clk/inst (2.33 GHz)\n", name, ms, ms * 1e-3 * 2.33e9 / iters / 128);
}
int main() { uint32_t* d; hipMalloc(&d, 1024 * 64 * 4); run<0>("v_lshrrev_b64 (2 chains)", d); run<1>("alignbit + lshr (dependent)", d); run<2>("v_mad_u64_u32 (2 chains, vgpr)", d); run<3>("v_and_b32_e64", d); run<4>("v_mul_lo_u32", d); run<5>("v_mad_u64_u32 (sgpr operand)", d); return 0; }
