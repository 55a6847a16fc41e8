// cost of an s_swappc call + return for a lone wave (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void stub_fn() { asm volatile("s_nop 0"); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void stub13_fn() {
    asm volatile("s_mov_b32 s40, 1\n s_mov_b32 s41, 1\n s_mov_b32 s42, 1\n s_mov_b32 s43, 1\n s_mov_b32 s44, 1\n s_mov_b32 s45, 1\n s_mov_b32 s46, 1\n s_mov_b32 s47, 1\n s_mov_b32 s56, 1\n s_mov_b32 s57, 1\n s_mov_b32 s58, 1\n s_mov_b32 s59, 1\n s_mov_b32 s60, 1" ::: "s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60");
}
#define CALL(sym) "s_getpc_b64 s[40:41]\n\ts_add_u32 s40, s40, " sym "@rel32@lo+4\n\ts_addc_u32 s41, s41, " sym "@rel32@hi+12\n\ts_swappc_b64 s[30:31], s[40:41]\n\t"
#define C4(s) CALL(s) CALL(s) CALL(s) CALL(s)
#define C16(s) C4(s) C4(s) C4(s) C4(s)
#define V8 "v_add_u32_e32 v10, v10, v11\n\tv_add_u32_e32 v11, v10, v11\n\tv_add_u32_e32 v10, v10, v11\n\tv_add_u32_e32 v11, v10, v11\n\tv_add_u32_e32 v10, v10, v11\n\tv_add_u32_e32 v11, v10, v11\n\tv_add_u32_e32 v10, v10, v11\n\tv_add_u32_e32 v11, v10, v11\n\t"
template <int V> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(C16("stub_fn") ::: "s30", "s31", "s40", "s41", "scc");
        if (V == 1) asm volatile(C16("stub13_fn") ::: "s30", "s31", "s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60", "scc");
        if (V == 2) asm volatile(V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn")
                                 V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") V8 CALL("stub_fn") ::: "s30", "s31", "s40", "s41", "scc", "v10", "v11");
    }
    out[blockIdx.x * 64 + threadIdx.x] = 0;
}
template <int V> void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 1024, iters = 20000;
    hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, 1000);
    hipEventRecord(e0); hipLaunchKernelGGL(kern<V>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %.3f ms  %.0f ticks per call (2.33 GHz)\n", name, ms, ms * 1e-3 * 2.33e9 / iters / 16);
}
int main() { uint32_t* d; hipMalloc(&d, 1024 * 64 * 4); run<0>("empty stub", d); run<1>("stub with 13 s_mov", d); run<2>("8 valu + empty stub", d); return 0; }
