// Instruction-footprint microbenchmark (dev tool): one wave per SIMD (40 KB of LDS per workgroup = 4 workgroups per CU) runs a loop whose body is
// KB kilobytes of straight-line VALU code; ns per instruction against the footprint shows what code that does not fit the instruction
// cache costs. Variants: 8-byte encodings (v_add_u32_e64: what the generated glue uses), 4-byte encodings (v_add_u32_e32), multiply-accumulates.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// body: INSTRS instructions via .rept of a group of 8
template <int GROUPS, int KIND> __global__ void __launch_bounds__(64) kern(uint32_t* out, int iters) {
    extern __shared__ uint32_t lds[];
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u;
    uint32_t h0 = a, h1 = b, h2 = a + 1, h3 = b + 1, h4 = a + 2, h5 = b + 2, h6 = a + 3, h7 = b + 3;
    uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1;
    if (KIND == 0)
        asm volatile("s_mov_b32 s22, %11\n .p2align 6\n 1:\n .rept %c10\n v_add_u32_e64 %0, %8, %0\n v_add_u32_e64 %1, %9, %1\n v_add_u32_e64 %2, %8, %2\n v_add_u32_e64 %3, %9, %3\n"
                     "v_add_u32_e64 %4, %8, %4\n v_add_u32_e64 %5, %9, %5\n v_add_u32_e64 %6, %8, %6\n v_add_u32_e64 %7, %9, %7\n .endr\n"
                     "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc0 2f\n s_getpc_b64 s[24:25]\n 3:\n s_sub_u32 s24, s24, 3b-1b\n s_subb_u32 s25, s25, 0\n s_setpc_b64 s[24:25]\n 2:\n"
                     : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a), "v"(b), "n"(GROUPS), "s"(iters) : "s22", "s24", "s25", "scc");
    else if (KIND == 1)
        asm volatile("s_mov_b32 s22, %11\n .p2align 6\n 1:\n .rept %c10\n v_add_u32_e32 %0, %8, %0\n v_add_u32_e32 %1, %9, %1\n v_add_u32_e32 %2, %8, %2\n v_add_u32_e32 %3, %9, %3\n"
                     "v_add_u32_e32 %4, %8, %4\n v_add_u32_e32 %5, %9, %5\n v_add_u32_e32 %6, %8, %6\n v_add_u32_e32 %7, %9, %7\n .endr\n"
                     "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc0 2f\n s_getpc_b64 s[24:25]\n 3:\n s_sub_u32 s24, s24, 3b-1b\n s_subb_u32 s25, s25, 0\n s_setpc_b64 s[24:25]\n 2:\n"
                     : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a), "v"(b), "n"(GROUPS), "s"(iters) : "s22", "s24", "s25", "scc");
    else
        asm volatile("s_mov_b32 s22, %7\n .p2align 6\n 1:\n .rept %c6\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, s[20:21], %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, s[20:21], %4, %5, %3\n"
                     "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, s[20:21], %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, s[20:21], %4, %5, %3\n .endr\n"
                     "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc0 2f\n s_getpc_b64 s[24:25]\n 3:\n s_sub_u32 s24, s24, 3b-1b\n s_subb_u32 s25, s25, 0\n s_setpc_b64 s[24:25]\n 2:\n"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "n"(GROUPS), "s"(iters) : "vcc", "s20", "s21", "s22", "s24", "s25", "scc");
    out[blockIdx.x * 64 + threadIdx.x] = h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5 ^ h6 ^ h7 ^ (uint32_t)(c0 ^ c1 ^ c2 ^ c3) ^ lds[threadIdx.x];
}
template <int GROUPS, int KIND> void run(const char* name, uint32_t* d, int blocks) {
    const long instr = 8L * GROUPS;
    int iters = (int)(40000000L / instr); if (iters < 4) iters = 4;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kern<GROUPS, KIND>), dim3(blocks), dim3(64), 40000, 0, d, 8);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((kern<GROUPS, KIND>), dim3(blocks), dim3(64), 40000, 0, d, iters);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const int bytes = KIND == 1 ? 4 : 8;
    printf("%-10s body %6ld instr %7.1f KB   %8.3f ms   %.3f ns/instr\n", name, instr, instr * bytes / 1024.0, best, best * 1e6 / ((double)instr * iters));
}
int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 4;
    uint32_t* d; CHK(hipMalloc(&d, (size_t)blocks * 64 * 4));
#define ROW(G) run<G, 0>("add_e64", d, blocks); run<G, 1>("add_e32", d, blocks); run<G, 2>("mad_u64", d, blocks);
    ROW(256) ROW(512) ROW(768) ROW(1024) ROW(1536) ROW(2048) ROW(4096) ROW(8192) ROW(16384)
    return 0;
}
