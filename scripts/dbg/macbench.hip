// v_mad_i64_i32 with a VGPR multiplicand against one with an SGPR multiplicand (the Montgomery reduction's p digits live in SGPRs), dependent chain as in the product
// scans, one wave per SIMD (dev probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u;
    uint64_t c = a;
    uint32_t sa = __builtin_amdgcn_readfirstlane(b | 1);
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP64(asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");) }
        else if (MODE == 1) { REP64(asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c) : "s"(sa), "v"(b) : "vcc");) }
        else { REP64(asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %3, %2, %0" : "+v"(c) : "v"(a), "v"(b), "s"(sa) : "vcc");) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)c ^ (uint32_t)(c >> 32);
}
template <int MODE> void run(const char* name, uint32_t* d, int cus, int per) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 1; w <= 2; w++) {
        int blocks = cus * 4 * w, iters = 20000; float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(64), 0, 0, d, 100);
            (void)hipEventRecord(e0); hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %d wave(s)/SIMD: %.3f ns of SIMD time per instruction\n", name, w, best * 1e6 / iters / per / w);
    }
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    uint32_t* d; (void)hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 64 * 4);
    run<0>("v_mad_i64_i32 acc, v, v (dependent chain)", d, p.multiProcessorCount, 64);
    run<1>("v_mad_i64_i32 acc, s, v (dependent chain)", d, p.multiProcessorCount, 64);
    run<2>("alternating v,v / s,v (dependent chain)", d, p.multiProcessorCount, 128);
    return 0;
}
