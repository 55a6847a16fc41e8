// mbls_fp.h -- Fp arithmetic for BLS12-381 on gfx950: 12 x 32-bit limbs in VGPRs, Montgomery form.
//
// One field element per lane. fp_mul is a product-scanning Montgomery multiplication whose inner step is one
// v_mad_u64_u32 (32x32+64 -> 64, carry-out in VCC) plus one v_addc_co_u32 into the third accumulator word; the Fp2-level
// routines (mbls_tower.h), the paired product and the fixed-exponent exponentiations below re-cut their operands into 14
// unsaturated 28-bit digits, where a product is a single v_mad_u64_u32 (tools/gen_fp_asm.py). No MFMA (integer carry work).
//
// Replaces amcl's `fp`/`big` modules that the reference reaches through BLSCurve::* (reference
// src/amcl_utils.rs:6-21). The same header compiles as plain C++ (MBLS_HOST_EMUL) so that the CPU-only
// test suite can run every lane function against the oracle without a GPU; that build is test
// infrastructure (tests/host_emul), never shipped in the product library.
#pragma once
#include <stdint.h>

#if defined(MBLS_HOST_EMUL)
#define MBLS_LDS
#define MBLS_FN static inline
#define MBLS_NOINLINE static __attribute__((noinline))
#define MBLS_CONST static const
#define MBLS_DEVICE_ASM 0
#else
#include <hip/hip_runtime.h>
#define MBLS_LDS __attribute__((address_space(3)))      // pointers into the workgroup's LDS (ds_* instructions, not flat)
#define MBLS_FN static __device__ __forceinline__
#define MBLS_NOINLINE static __device__ __noinline__
#define MBLS_CONST static __device__ __constant__ const
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MBLS_NO_ASM)
#define MBLS_DEVICE_ASM 1
#else
#define MBLS_DEVICE_ASM 0
#endif
#endif

// In the compiler-scheduled code (generic n-pair Miller loop, cold paths) Fp6/Fp12-level operations are inlined into the
// functions that own the loops: their Fp6 temporaries then live in VGPRs/AGPRs instead of lane-private memory.
// -DMBLS_OUTLINE_TOWER restores real functions on memory operands (smaller code, slower). The hot loops of the verification
// path do not go through this code at all: they are generated routines (mbls_towerd_asm.inc).
#if !defined(MBLS_OUTLINE_TOWER)
#define MBLS_INLINE_TOWER 1
#endif
#if defined(MBLS_INLINE_TOWER)
#define MBLS_TOWER_FN MBLS_FN
#else
#define MBLS_TOWER_FN MBLS_NOINLINE
#endif
// operations that sit on rarely-taken paths of the hot loops (5 of 63 iterations): kept out of line unless asked
#if defined(MBLS_INLINE_TOWER) && defined(MBLS_INLINE_COLD)
#define MBLS_TOWER_COLD_FN MBLS_FN
#else
#define MBLS_TOWER_COLD_FN MBLS_NOINLINE
#endif

#include "mbls_constants.inc"

// An Fp element is a 12-lane-private-dword vector so that it is passed to and returned from
// non-inlined device functions in VGPRs (a struct of 12 dwords would go through scratch).
// aligned(16): a 12-element vector is naturally 64-byte aligned, which makes every function with an Fp
// local realign its stack through a base-pointer register (s34). hipcc 7.2's inter-procedural register
// allocation does not treat s34 as preserved across calls (callees such as g2_psi use it as a plain
// temporary), so a realigning caller restores a garbage stack pointer on return -- observed as memory
// faults / silently overlapping frames on gfx950. With 16-byte alignment (the ABI stack alignment) no
// function realigns and no base pointer exists. tests/test_build.py checks the generated ISA for this.
typedef uint32_t fp __attribute__((ext_vector_type(12), aligned(16)));

// modulus limbs as literals (become s_mov_b32 immediates, no constant-memory round trip)
#define MBLS_P0 0xffffaaabu
#define MBLS_P1 0xb9feffffu
#define MBLS_P2 0xb153ffffu
#define MBLS_P3 0x1eabfffeu
#define MBLS_P4 0xf6b0f624u
#define MBLS_P5 0x6730d2a0u
#define MBLS_P6 0xf38512bfu
#define MBLS_P7 0x64774b84u
#define MBLS_P8 0x434bacd7u
#define MBLS_P9 0x4b1ba7b6u
#define MBLS_P10 0x397fe69au
#define MBLS_P11 0x1a0111eau

MBLS_FN uint32_t fp_plimb(int i) {
    switch (i) {
        case 0: return MBLS_P0; case 1: return MBLS_P1; case 2: return MBLS_P2; case 3: return MBLS_P3;
        case 4: return MBLS_P4; case 5: return MBLS_P5; case 6: return MBLS_P6; case 7: return MBLS_P7;
        case 8: return MBLS_P8; case 9: return MBLS_P9; case 10: return MBLS_P10; default: return MBLS_P11;
    }
}

MBLS_FN fp fp_zero() { fp r = 0; return r; }
MBLS_FN fp fp_load_const(const uint32_t* c) {
    fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r[i] = c[i];
    return r;
}
MBLS_FN fp fp_one() { return fp_load_const(MBLS_ONE); }

MBLS_FN bool fp_is_zero(fp a) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) t |= a[i];
    return t == 0;
}
MBLS_FN bool fp_eq(fp a, fp b) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) t |= a[i] ^ b[i];
    return t == 0;
}
MBLS_FN fp fp_select(bool c, fp a, fp b) {   // c ? a : b, branch-free
    fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r[i] = c ? a[i] : b[i];
    return r;
}

// Carry chains are written with __builtin_addc / __builtin_subc: they lower to v_add_co_u32 / v_addc_co_u32 /
// v_subb_co_u32 chains (36-48 VALU instructions per modular add). The obvious uint64_t formulation compiles to
// ~155 instructions per add on gfx950 (v_lshl_add_u64 with zero-extension moves), which made the additions of the
// Karatsuba tower cost as much as the multiplications.
// r = a - p if a >= p else a   (a < 2p; top = carry word above the 12 limbs)
MBLS_FN fp fp_reduce_once(fp a, uint32_t top) {
    fp d; unsigned br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned bo; d[i] = __builtin_subc(a[i], fp_plimb(i), br, &bo); br = bo; }
    bool ge = (top != 0) | (br == 0);
    return fp_select(ge, d, a);
}
MBLS_FN fp fp_add(fp a, fp b) {
    fp t; unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned co; t[i] = __builtin_addc(a[i], b[i], c, &co); c = co; }
    return fp_reduce_once(t, 0);      // a + b < 2p < 2^384: no carry out of the top limb
}
MBLS_FN fp fp_sub(fp a, fp b) {
    fp t; unsigned br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned bo; t[i] = __builtin_subc(a[i], b[i], br, &bo); br = bo; }
    uint32_t m = 0u - br; unsigned c = 0; fp r;      // add p back iff the subtraction borrowed
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned co; r[i] = __builtin_addc(t[i], fp_plimb(i) & m, c, &co); c = co; }
    return r;
}
// Unreduced variants for operands that feed a Montgomery multiplication directly: with R = 2^384 > 9.8 p the multiplier
// accepts factors below 2p (a b / R + p < 1.41 p, brought below p by its one conditional subtraction), so the sum of two
// reduced values, or a difference lifted by p, needs no reduction of its own (12 / 24 instructions instead of 36).
MBLS_FN fp fp_add_nr(fp a, fp b) {     // a + b in [0, 2p)
    fp t; unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned co; t[i] = __builtin_addc(a[i], b[i], c, &co); c = co; }
    return t;
}
MBLS_FN fp fp_sub_nr(fp a, fp b) {     // a - b + p in (0, 2p)
    fp t; unsigned c = 0, br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned co; t[i] = __builtin_addc(a[i], fp_plimb(i), c, &co); c = co; }
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned bo; t[i] = __builtin_subc(t[i], b[i], br, &bo); br = bo; }
    return t;
}
MBLS_FN fp fp_neg(fp a) {             // p - a, and 0 stays 0
    fp t; unsigned br = 0; uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned bo; t[i] = __builtin_subc(fp_plimb(i), a[i], br, &bo); br = bo; nz |= a[i]; }
    uint32_t m = nz ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int i = 0; i < 12; i++) t[i] &= m;
    return t;
}
MBLS_FN fp fp_dbl(fp a) { return fp_add(a, a); }
// a/2: (a + (a odd ? p : 0)) >> 1 -- valid on Montgomery representatives as well
MBLS_FN fp fp_half(fp a) {
    uint32_t m = 0u - (a[0] & 1u); fp t; unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned co; t[i] = __builtin_addc(a[i], fp_plimb(i) & m, c, &co); c = co; }
    fp r;
#pragma unroll
    for (int i = 0; i < 11; i++) r[i] = (t[i] >> 1) | (t[i + 1] << 31);
    r[11] = (t[11] >> 1) | ((uint32_t)c << 31);
    return r;
}

// ------------------------------------------------------------------------------------------------
// Montgomery multiplication r = a*b/2^384 mod p, product scanning over a 96-bit column accumulator.
#if defined(MBLS_HOST_EMUL)
static thread_local uint64_t mbls_cnt_mul = 0, mbls_cnt_sqr = 0;     // op census (host emulation only)
#define MBLS_COUNT_MUL() (mbls_cnt_mul++)
#define MBLS_COUNT_SQR() (mbls_cnt_sqr++)
#else
#define MBLS_COUNT_MUL()
#define MBLS_COUNT_SQR()
#endif
struct mbls_acc { uint64_t lo; uint32_t hi; };
MBLS_FN void mbls_mac(mbls_acc& s, uint32_t a, uint32_t b) {
#if MBLS_DEVICE_ASM
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "+v"(s.lo), "+v"(s.hi) : "v"(a), "v"(b) : "vcc");
#else
    uint64_t p = (uint64_t)a * b; uint64_t n = s.lo + p; s.hi += (n < p); s.lo = n;
#endif
}
MBLS_FN void mbls_mac_p(mbls_acc& s, uint32_t m, uint32_t pl) {   // pl = a modulus limb (scalar register)
#if MBLS_DEVICE_ASM
    asm("v_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "+v"(s.lo), "+v"(s.hi) : "v"(m), "s"(pl) : "vcc");
#else
    mbls_mac(s, m, pl);
#endif
}
MBLS_FN void mbls_acc_shift(mbls_acc& s) { s.lo = (s.lo >> 32) | ((uint64_t)s.hi << 32); s.hi = 0; }

#if MBLS_DEVICE_ASM
#include "mbls_fp_asm.inc"
// Hand-scheduled body (tools/gen_fp_asm.py): one asm statement, operands pinned to the registers the calling convention
// already uses (a: v[0:11], b: v[12:23], result: v[0:11]); 288 v_mad_u64_u32 + 288 v_addc_co_u32, no compiler padding.
// Private-convention routines (mbls_fp2_mul_asm_fn in mbls_tower.h) are ordinary device functions whose whole body is the asm
// text, reached through an s_swappc inside an asm statement: the statement's constraints and clobbers are the contract.
// (Entering 4 bytes past the symbol to skip hipcc's entry `s_waitcnt vmcnt(0)` was measured: no gain, not kept.)
#define MBLS_ASM_CALL(sym) "s_getpc_b64 s[40:41]\n\ts_add_u32 s40, s40, " sym "@rel32@lo+4\n\ts_addc_u32 s41, s41, " sym "@rel32@hi+12\n\ts_swappc_b64 s[30:31], s[40:41]"
// (fp_mul itself stays a regular function: routing its thousands of call sites through asm statements overwhelms hipcc.)
__attribute__((aligned(64))) MBLS_NOINLINE fp fp_mul(fp a, fp b) {
    fp r;
    asm volatile(MBLS_FP_MUL_ASM : "={v[0:11]}"(r), "+{v[12:23]}"(b) : "{v[0:11]}"(a) : MBLS_FP_MUL_CLOBBERS);
    return r;
}
MBLS_FN fp fp_sqr(fp a) { return fp_mul(a, a); }
// Two independent products at once (generated routine, two interleaved scans on the 28-bit core): ~4.4 k clocks for the
// pair against ~3.1 k for one fp_mul, whose single dependent multiply-accumulate chain cannot be overlapped with anything
// when the wave is alone on its SIMD.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp_mulpair_asm_fn() { asm volatile(MBLS_FP_MULPAIR_ASM); }
MBLS_FN void fp_mul_pair(fp* c0, fp* c1, fp a0, fp b0, fp a1, fp b1) {
    fp r0, r1;
    asm volatile(MBLS_ASM_CALL("mbls_fp_mulpair_asm_fn")
                 : "={v[48:59]}"(r0), "={v[60:71]}"(r1), "+{v[0:11]}"(a0), "+{v[12:23]}"(b0), "+{v[24:35]}"(a1), "+{v[36:47]}"(b1)
                 :
                 : MBLS_FP_MULPAIR_CLOBBERS, "s30", "s31");
    *c0 = r0; *c1 = r1;
}
#else
MBLS_NOINLINE fp fp_mul(fp a, fp b) {
    uint32_t m[12]; fp t; mbls_acc s = {0, 0};
    MBLS_COUNT_MUL();
#pragma unroll
    for (int k = 0; k < 12; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mbls_mac(s, a[i], b[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mbls_mac_p(s, m[i], fp_plimb(k - i));
        m[k] = (uint32_t)s.lo * MBLS_NP0;
        mbls_mac_p(s, m[k], MBLS_P0);
        mbls_acc_shift(s);
    }
#pragma unroll
    for (int k = 12; k < 24; k++) {
#pragma unroll
        for (int i = k - 11; i < 12; i++) mbls_mac(s, a[i], b[k - i]);
#pragma unroll
        for (int i = k - 11; i < 12; i++) mbls_mac_p(s, m[i], fp_plimb(k - i));
        t[k - 12] = (uint32_t)s.lo;
        mbls_acc_shift(s);
    }
    return fp_reduce_once(t, (uint32_t)s.lo);
}
// Squaring: off-diagonal products once, doubled, plus the diagonal.
MBLS_NOINLINE fp fp_sqr(fp a) {
    uint32_t m[12]; fp t; mbls_acc s = {0, 0};
    MBLS_COUNT_SQR();
#pragma unroll
    for (int k = 0; k < 24; k++) {
        mbls_acc d = {0, 0};
#pragma unroll
        for (int i = 0; i < 12; i++) {
            int j = k - i;
            if (j > i && j < 12) mbls_mac(d, a[i], a[j]);
        }
        // s += 2*d (+ a[k/2]^2 when k even)
        {
            uint32_t d2 = (d.hi << 1) | (uint32_t)(d.lo >> 63); uint64_t d01 = d.lo << 1;
            uint64_t n = s.lo + d01; s.hi += d2 + (n < d01); s.lo = n;
        }
        if ((k & 1) == 0) mbls_mac(s, a[k / 2], a[k / 2]);
        if (k < 12) {
#pragma unroll
            for (int i = 0; i < k; i++) mbls_mac_p(s, m[i], fp_plimb(k - i));
            m[k] = (uint32_t)s.lo * MBLS_NP0;
            mbls_mac_p(s, m[k], MBLS_P0);
        } else {
#pragma unroll
            for (int i = k - 11; i < 12; i++) mbls_mac_p(s, m[i], fp_plimb(k - i));
            t[k - 12] = (uint32_t)s.lo;
        }
        mbls_acc_shift(s);
    }
    return fp_reduce_once(t, (uint32_t)s.lo);
}
MBLS_FN void fp_mul_pair(fp* c0, fp* c1, fp a0, fp b0, fp a1, fp b1) { *c0 = fp_mul(a0, b0); *c1 = fp_mul(a1, b1); }
#endif

MBLS_FN fp fp_to_mont(fp raw) { return fp_mul(raw, fp_load_const(MBLS_R2)); }
MBLS_FN fp fp_from_mont(fp a) { fp one = 0; one[0] = 1; return fp_mul(a, one); }

// raw (non-Montgomery) comparisons
MBLS_FN bool fp_raw_geq_p(fp raw) {
    unsigned br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned bo; (void)__builtin_subc(raw[i], fp_plimb(i), br, &bo); br = bo; }
    return br == 0;
}
MBLS_FN bool fp_raw_gt_half(fp raw) {   // raw > (p-1)/2
    unsigned br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { unsigned bo; (void)__builtin_subc(MBLS_P_MINUS_1_DIV_2[i], raw[i], br, &bo); br = bo; }
    return br != 0;
}
MBLS_FN bool fp_lex_largest(fp a) { return fp_raw_gt_half(fp_from_mont(a)); }
MBLS_FN uint32_t fp_parity(fp a) { return fp_from_mont(a)[0] & 1u; }

// 48 big-endian bytes -> raw limbs / back
MBLS_FN fp fp_raw_from_be(const uint8_t* b) {
    fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint8_t* q = b + 44 - 4 * i;
        r[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    }
    return r;
}
MBLS_FN void fp_raw_to_be(uint8_t* b, fp r) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint8_t* q = b + 44 - 4 * i;
        q[0] = (uint8_t)(r[i] >> 24); q[1] = (uint8_t)(r[i] >> 16); q[2] = (uint8_t)(r[i] >> 8); q[3] = (uint8_t)r[i];
    }
}

// a^e, e = 12 little-endian 32-bit limbs in constant memory; fixed 4-bit windows (table in lane-private memory)
MBLS_NOINLINE fp fp_pow_const(fp a, const uint32_t* e) {
    fp tab[16];
    tab[0] = fp_one(); tab[1] = a;
    for (int i = 2; i < 16; i++) tab[i] = fp_mul(tab[i - 1], a);
    fp acc = fp_one();
    bool started = false;
    for (int w = 95; w >= 0; w--) {
        uint32_t nib = (e[w >> 3] >> ((w & 7) * 4)) & 0xFu;    // uniform across lanes: e is a constant
        if (started) { acc = fp_sqr(acc); acc = fp_sqr(acc); acc = fp_sqr(acc); acc = fp_sqr(acc); }
        if (nib) { acc = started ? fp_mul(acc, tab[nib]) : tab[nib]; started = true; }
    }
    return acc;
}
#if MBLS_DEVICE_ASM
// The two exponentiations of the hot path (square roots: (p-3)/4, inversion: p-2) as generated routines that work on 14
// unsaturated 28-bit digits from start to end (tools/gen_fp_asm.py, pow_body): no conversions or carries between the ~460
// dependent multiplications (sliding 5-bit windows: ~378 squarings + ~82 products), squarings with half the cross products; the
// table of the 16 odd powers lives in AGPRs and the running value ping-pongs between two register groups.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_pow_sqr_xy_asm_fn() { asm volatile(MBLS_POW_SQR_XY_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_pow_sqr_yx_asm_fn() { asm volatile(MBLS_POW_SQR_YX_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_pow_mul_xb_y_asm_fn() { asm volatile(MBLS_POW_MUL_XB_Y_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_pow_mul_yb_x_asm_fn() { asm volatile(MBLS_POW_MUL_YB_X_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp_pow_pm3d4_asm_fn() { asm volatile(MBLS_FP_POW_PM3D4_ASM); }
// The inversion is not an exponentiation: Bernstein-Yang safegcd divsteps on 13 signed 30-bit limbs (tools/gen_fp_asm.py,
// fp_inv_gcd_body), the same 900 divsteps for every operand, ~30 k instructions instead of ~190 k for a^(p-2).
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp_inv_gcd_asm_fn() { asm volatile(MBLS_FP_INV_GCD_ASM); }
MBLS_FN fp fp_inv(fp a) {                                                          // 0 -> 0
    asm volatile(MBLS_ASM_CALL("mbls_fp_inv_gcd_asm_fn") : "+{v[0:11]}"(a) : : MBLS_FP_INV_GCD_CLOBBERS, "s30", "s31");
    return a;
}
// the same with 4-bit windows (8 table entries, a0..a111): for kernels that fit 256 registers and run two waves per SIMD
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp_pow_pm3d4_w4_asm_fn() { asm volatile(MBLS_FP_POW_PM3D4_W4_ASM); }
// w = a^((p-3)/4): sqrt candidate = w*a, 1/a = chi * w^2 with chi = (w*a)^2 / a = +-1
MBLS_FN fp fp_pow_pm3d4(fp a) {
    asm volatile(MBLS_ASM_CALL("mbls_fp_pow_pm3d4_asm_fn") : "+{v[0:11]}"(a) : : MBLS_FP_POW_CLOBBERS);
    return a;
}
MBLS_FN fp fp_pow_pm3d4_w4(fp a) {
    asm volatile(MBLS_ASM_CALL("mbls_fp_pow_pm3d4_w4_asm_fn") : "+{v[0:11]}"(a) : : MBLS_FP_POW_W4_CLOBBERS);
    return a;
}
#else
MBLS_FN fp fp_inv(fp a) { return fp_pow_const(a, MBLS_EXP_P_MINUS_2); }            // 0 -> 0
// w = a^((p-3)/4): sqrt candidate = w*a, 1/a = chi * w^2 with chi = (w*a)^2 / a = +-1
MBLS_FN fp fp_pow_pm3d4(fp a) { return fp_pow_const(a, MBLS_EXP_P_MINUS_3_DIV_4); }
MBLS_FN fp fp_pow_pm3d4_w4(fp a) { return fp_pow_const(a, MBLS_EXP_P_MINUS_3_DIV_4); }
#endif
// returns true and a root in *r if a is a square (0 -> 0)
template <bool W4 = false>
MBLS_FN bool fp_sqrt(fp* r, fp a) {
    fp s = fp_mul(W4 ? fp_pow_pm3d4_w4(a) : fp_pow_pm3d4(a), a);
    *r = s;
    return fp_eq(fp_sqr(s), a);
}
