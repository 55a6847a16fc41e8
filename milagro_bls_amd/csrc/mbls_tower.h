// mbls_tower.h -- Fp2 / Fp6 / Fp12 tower for BLS12-381, one element per lane.
//   Fp2 = Fp[i]/(i^2+1), Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v), xi = 1 + i.
// Replaces amcl's fp2/fp4/fp12 modules (reference src/amcl_utils.rs:18-19 re-exports FP2/FP12); the
// tower shape is unobservable at the reference's API (only bools and serialized points leave it).
// Fp2-level operations are inlined around the two non-inlined Fp primitives; Fp6/Fp12-level
// operations are real functions on lane-private memory operands, which bounds code size and
// register pressure (an Fp12 is 144 dwords per lane).
#pragma once
#include "mbls_fp.h"

struct fp2 { fp c0, c1; };
struct fp6 { fp2 c0, c1, c2; };
struct fp12 { fp6 c0, c1; };

MBLS_FN fp2 fp2_zero() { fp2 r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
MBLS_FN fp2 fp2_one() { fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
MBLS_FN fp2 fp2_load_const(const uint32_t (*c)[12]) { fp2 r; r.c0 = fp_load_const(c[0]); r.c1 = fp_load_const(c[1]); return r; }
MBLS_FN bool fp2_is_zero(const fp2& a) { return fp_is_zero(a.c0) & fp_is_zero(a.c1); }
MBLS_FN bool fp2_eq(const fp2& a, const fp2& b) { return fp_eq(a.c0, b.c0) & fp_eq(a.c1, b.c1); }
MBLS_FN fp2 fp2_select(bool c, const fp2& a, const fp2& b) { fp2 r; r.c0 = fp_select(c, a.c0, b.c0); r.c1 = fp_select(c, a.c1, b.c1); return r; }
MBLS_FN fp2 fp2_add(const fp2& a, const fp2& b) { fp2 r; r.c0 = fp_add(a.c0, b.c0); r.c1 = fp_add(a.c1, b.c1); return r; }
MBLS_FN fp2 fp2_sub(const fp2& a, const fp2& b) { fp2 r; r.c0 = fp_sub(a.c0, b.c0); r.c1 = fp_sub(a.c1, b.c1); return r; }
MBLS_FN fp2 fp2_neg(const fp2& a) { fp2 r; r.c0 = fp_neg(a.c0); r.c1 = fp_neg(a.c1); return r; }
MBLS_FN fp2 fp2_dbl(const fp2& a) { return fp2_add(a, a); }
MBLS_FN fp2 fp2_conj(const fp2& a) { fp2 r; r.c0 = a.c0; r.c1 = fp_neg(a.c1); return r; }
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
// Fp2 multiplication as one hand-written routine with a private calling convention (tools/gen_fp_asm.py, fp2_mul_body):
// operands a0,a1,b0,b1 in v[0:47] (overwritten), results in v[48:71]. It computes c0 = a0 b0 + a1 (2p - b1) and
// c1 = a0 b1 + a1 b0 as two sum-of-two-products scans with one Montgomery reduction each and no modular additions, on 14
// digits of 28 bits so that a multiply-accumulate is a single v_mad_u64_u32 (1176 of them, against 900 x 2 instructions +
// five modular additions + three calls for Karatsuba on 32-bit limbs). The routine is reached by an s_swappc inside an asm
// statement, so the 32-VGPR argument limit of the regular calling convention does not apply; the statement's operand and
// clobber lists are the contract.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_mul_asm_fn() {
    asm volatile(MBLS_FP2_MUL_ASM);
}
MBLS_FN fp2 fp2_mul(const fp2& a, const fp2& b) {
    fp c0, c1, a0 = a.c0, a1 = a.c1, b0 = b.c0, b1 = b.c1;          // the routine overwrites its operand registers
    asm volatile(MBLS_ASM_CALL("mbls_fp2_mul_asm_fn")
                 : "={v[48:59]}"(c0), "={v[60:71]}"(c1), "+{v[0:11]}"(a0), "+{v[12:23]}"(a1), "+{v[24:35]}"(b0), "+{v[36:47]}"(b1)
                 :
                 : MBLS_FP2_MUL_CLOBBERS, "s30", "s31");
    fp2 r; r.c0 = c0; r.c1 = c1; return r;
}
#else
MBLS_FN fp2 fp2_mul(const fp2& a, const fp2& b) {
    fp t0 = fp_mul(a.c0, b.c0), t1 = fp_mul(a.c1, b.c1);
    fp t2 = fp_mul(fp_add_nr(a.c0, a.c1), fp_add_nr(b.c0, b.c1));     // factors < 2p: see fp_add_nr
    fp2 r; r.c0 = fp_sub(t0, t1); r.c1 = fp_sub(fp_sub(t2, t0), t1); return r;
}
#endif
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
// Fp2 squaring, same private-convention scheme: c0 = (a0+a1)(a0-a1+p), c1 = a0 (2 a1) on unreduced operands, two scans.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_sqr_asm_fn() {
    asm volatile(MBLS_FP2_SQR_ASM);
}
MBLS_FN fp2 fp2_sqr(const fp2& a) {
    fp c0, c1, a0 = a.c0, a1 = a.c1;
    asm volatile(MBLS_ASM_CALL("mbls_fp2_sqr_asm_fn")
                 : "={v[24:35]}"(c0), "={v[36:47]}"(c1), "+{v[0:11]}"(a0), "+{v[12:23]}"(a1)
                 :
                 : MBLS_FP2_SQR_CLOBBERS, "s30", "s31");
    fp2 r; r.c0 = c0; r.c1 = c1; return r;
}
#else
MBLS_FN fp2 fp2_sqr(const fp2& a) {
    fp m = fp_mul(a.c0, a.c1);
    fp2 r; r.c0 = fp_mul(fp_add_nr(a.c0, a.c1), fp_sub_nr(a.c0, a.c1)); r.c1 = fp_dbl(m); return r;
}
#endif
MBLS_FN fp2 fp2_mul_fp(const fp2& a, fp k) { fp2 r; r.c0 = fp_mul(a.c0, k); r.c1 = fp_mul(a.c1, k); return r; }
MBLS_FN fp2 fp2_mul_xi(const fp2& a) { fp2 r; r.c0 = fp_sub(a.c0, a.c1); r.c1 = fp_add(a.c0, a.c1); return r; }
MBLS_FN fp2 fp2_mul_i(const fp2& a) { fp2 r; r.c0 = fp_neg(a.c1); r.c1 = a.c0; return r; }
MBLS_FN fp fp2_norm(const fp2& a) { return fp_add(fp_sqr(a.c0), fp_sqr(a.c1)); }
MBLS_FN fp2 fp2_inv(const fp2& a) {
    fp n = fp_inv(fp2_norm(a));
    fp2 r; r.c0 = fp_mul(a.c0, n); r.c1 = fp_neg(fp_mul(a.c1, n)); return r;
}
// small-constant multiples by additions
MBLS_FN fp2 fp2_mul3(const fp2& a) { return fp2_add(fp2_dbl(a), a); }
MBLS_FN fp2 fp2_mul4(const fp2& a) { return fp2_dbl(fp2_dbl(a)); }
MBLS_FN fp2 fp2_mul8(const fp2& a) { return fp2_dbl(fp2_mul4(a)); }
MBLS_FN fp2 fp2_mul12(const fp2& a) { return fp2_add(fp2_mul8(a), fp2_mul4(a)); }
MBLS_FN bool fp2_lex_largest(const fp2& a) {         // ZCash rule: c1 first, then c0
    fp r1 = fp_from_mont(a.c1), r0 = fp_from_mont(a.c0);
    bool z1 = fp_is_zero(r1);
    return z1 ? fp_raw_gt_half(r0) : fp_raw_gt_half(r1);
}
MBLS_FN uint32_t fp2_sgn0(const fp2& a) {             // RFC 9380 section 4.1, m = 2
    fp r0 = fp_from_mont(a.c0), r1 = fp_from_mont(a.c1);
    uint32_t s0 = r0[0] & 1u, z0 = fp_is_zero(r0) ? 1u : 0u, s1 = r1[0] & 1u;
    return s0 | (z0 & s1);
}
// Square root in Fp2 by the complex method with two Fp exponentiations.
// Returns false if a is not a square. Any of the two roots may be returned.
MBLS_FN bool fp2_sqrt_inl(fp2* out, const fp2* ap) {
    fp2 a = *ap;
    fp n = fp2_norm(a);
    fp s;
    bool sq = fp_sqrt(&s, n);                         // s^2 = a0^2 + a1^2
    fp t = fp_half(fp_add(a.c0, s));
    fp t_alt = fp_half(fp_sub(a.c0, s));
    t = fp_select(fp_is_zero(t), t_alt, t);           // only when a1 = 0 and s = -a0
    fp w = fp_pow_pm3d4(t);                           // t^((p-3)/4)
    fp x0 = fp_mul(w, t);                             // x0^2 = chi * t
    bool chi = fp_eq(fp_sqr(x0), t);                  // t is a residue
    fp inv_x0 = fp_mul(x0, fp_sqr(w));                // 1/x0 = x0 w^2 (x0 != 0)
    fp other = fp_mul(fp_half(a.c1), inv_x0);         // a1 / (2 x0)
    fp2 r;
    r.c0 = fp_select(chi, x0, other);
    r.c1 = fp_select(chi, other, x0);
    bool zero = fp2_is_zero(a);
    r = fp2_select(zero, fp2_zero(), r);
    *out = r;
    // verify (also rejects non-squares whose norm happens to pass nothing: norm test is exact)
    return zero | (sq & fp2_eq(fp2_sqr(r), a));
}
MBLS_NOINLINE bool fp2_sqrt(fp2* out, const fp2* ap) { return fp2_sqrt_inl(out, ap); }

// ------------------------------------------------------------------------------------------------ Fp6
MBLS_FN void fp6_add(fp6* r, const fp6* a, const fp6* b) { r->c0 = fp2_add(a->c0, b->c0); r->c1 = fp2_add(a->c1, b->c1); r->c2 = fp2_add(a->c2, b->c2); }
MBLS_FN void fp6_sub(fp6* r, const fp6* a, const fp6* b) { r->c0 = fp2_sub(a->c0, b->c0); r->c1 = fp2_sub(a->c1, b->c1); r->c2 = fp2_sub(a->c2, b->c2); }
MBLS_FN void fp6_neg(fp6* r, const fp6* a) { r->c0 = fp2_neg(a->c0); r->c1 = fp2_neg(a->c1); r->c2 = fp2_neg(a->c2); }
MBLS_FN void fp6_mul_v(fp6* r, const fp6* a) { fp2 t = fp2_mul_xi(a->c2); r->c2 = a->c1; r->c1 = a->c0; r->c0 = t; }
MBLS_TOWER_FN void fp6_mul(fp6* r, const fp6* a, const fp6* b) {
    fp2 t0 = fp2_mul(a->c0, b->c0), t1 = fp2_mul(a->c1, b->c1), t2 = fp2_mul(a->c2, b->c2);
    fp2 c0 = fp2_mul(fp2_add(a->c1, a->c2), fp2_add(b->c1, b->c2));
    c0 = fp2_add(fp2_mul_xi(fp2_sub(fp2_sub(c0, t1), t2)), t0);
    fp2 c1 = fp2_mul(fp2_add(a->c0, a->c1), fp2_add(b->c0, b->c1));
    c1 = fp2_add(fp2_sub(fp2_sub(c1, t0), t1), fp2_mul_xi(t2));
    fp2 c2 = fp2_mul(fp2_add(a->c0, a->c2), fp2_add(b->c0, b->c2));
    c2 = fp2_add(fp2_sub(fp2_sub(c2, t0), t2), t1);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
// a * (x + y v)
MBLS_TOWER_FN void fp6_mul_01(fp6* r, const fp6* a, const fp2* x, const fp2* y) {
    fp2 t0 = fp2_mul(a->c0, *x), t1 = fp2_mul(a->c1, *y);
    fp2 c1 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a->c0, a->c1), fp2_add(*x, *y)), t0), t1);
    fp2 c0 = fp2_add(fp2_mul_xi(fp2_mul(a->c2, *y)), t0);
    fp2 c2 = fp2_add(fp2_mul(a->c2, *x), t1);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
// a * (y v)
MBLS_TOWER_FN void fp6_mul_1(fp6* r, const fp6* a, const fp2* y) {
    fp2 c0 = fp2_mul_xi(fp2_mul(a->c2, *y)), c1 = fp2_mul(a->c0, *y), c2 = fp2_mul(a->c1, *y);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
MBLS_NOINLINE void fp6_inv(fp6* r, const fp6* a) {
    fp2 A = fp2_sub(fp2_sqr(a->c0), fp2_mul_xi(fp2_mul(a->c1, a->c2)));
    fp2 B = fp2_sub(fp2_mul_xi(fp2_sqr(a->c2)), fp2_mul(a->c0, a->c1));
    fp2 C = fp2_sub(fp2_sqr(a->c1), fp2_mul(a->c0, a->c2));
    fp2 F = fp2_add(fp2_mul_xi(fp2_add(fp2_mul(a->c2, B), fp2_mul(a->c1, C))), fp2_mul(a->c0, A));
    F = fp2_inv(F);
    r->c0 = fp2_mul(A, F); r->c1 = fp2_mul(B, F); r->c2 = fp2_mul(C, F);
}

// ------------------------------------------------------------------------------------------------ Fp12
MBLS_FN void fp12_set_one(fp12* r) {
    r->c0.c0 = fp2_one(); r->c0.c1 = fp2_zero(); r->c0.c2 = fp2_zero();
    r->c1.c0 = fp2_zero(); r->c1.c1 = fp2_zero(); r->c1.c2 = fp2_zero();
}
MBLS_TOWER_COLD_FN void fp12_mul(fp12* r, const fp12* a, const fp12* b) {
    fp6 t0, t1, s, u, c1;
    fp6_mul(&t0, &a->c0, &b->c0); fp6_mul(&t1, &a->c1, &b->c1);
    fp6_add(&s, &a->c0, &a->c1); fp6_add(&u, &b->c0, &b->c1); fp6_mul(&c1, &s, &u);
    fp6_sub(&c1, &c1, &t0); fp6_sub(&c1, &c1, &t1);
    fp6_mul_v(&s, &t1); fp6_add(&r->c0, &t0, &s); r->c1 = c1;
}
MBLS_TOWER_FN void fp12_sqr(fp12* r, const fp12* a) {
    fp6 ab, s, t, va;
    fp6_mul(&ab, &a->c0, &a->c1);
    fp6_add(&s, &a->c0, &a->c1); fp6_mul_v(&va, &a->c1); fp6_add(&t, &a->c0, &va);
    fp6_mul(&s, &s, &t); fp6_sub(&s, &s, &ab); fp6_mul_v(&t, &ab); fp6_sub(&r->c0, &s, &t);
    fp6_add(&r->c1, &ab, &ab);
}
// f * (c0 + c2 w^2 + c3 w^3): tower positions c0.c0 = c0, c0.c1 = c2, c1.c1 = c3
MBLS_TOWER_FN void fp12_mul_line(fp12* r, const fp12* f, const fp2* c0, const fp2* c2, const fp2* c3) {
    fp6 t0, t1, s, c1; fp2 y;
    fp6_mul_01(&t0, &f->c0, c0, c2); fp6_mul_1(&t1, &f->c1, c3);
    fp6_add(&s, &f->c0, &f->c1); y = fp2_add(*c2, *c3); fp6_mul_01(&c1, &s, c0, &y);
    fp6_sub(&c1, &c1, &t0); fp6_sub(&c1, &c1, &t1);
    fp6_mul_v(&s, &t1); fp6_add(&r->c0, &t0, &s); r->c1 = c1;
}
MBLS_FN void fp12_conj(fp12* r, const fp12* a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }
MBLS_NOINLINE void fp12_inv(fp12* r, const fp12* a) {
    fp6 t0, t1;
    fp6_mul(&t0, &a->c0, &a->c0); fp6_mul(&t1, &a->c1, &a->c1); fp6_mul_v(&t1, &t1); fp6_sub(&t0, &t0, &t1);
    fp6_inv(&t0, &t0); fp6_mul(&r->c0, &a->c0, &t0); fp6_mul(&t1, &a->c1, &t0); fp6_neg(&r->c1, &t1);
}
// (c w^k)^p = conj(c) * FROB_W[k] w^k; tower coefficient of w^k: k even -> c0.c(k/2), k odd -> c1.c((k-1)/2)
MBLS_NOINLINE void fp12_frob(fp12* r, const fp12* a) {
    r->c0.c0 = fp2_conj(a->c0.c0);
    r->c1.c0 = fp2_mul(fp2_conj(a->c1.c0), fp2_load_const(MBLS_FROB_W[1]));
    r->c0.c1 = fp2_mul(fp2_conj(a->c0.c1), fp2_load_const(MBLS_FROB_W[2]));
    r->c1.c1 = fp2_mul(fp2_conj(a->c1.c1), fp2_load_const(MBLS_FROB_W[3]));
    r->c0.c2 = fp2_mul(fp2_conj(a->c0.c2), fp2_load_const(MBLS_FROB_W[4]));
    r->c1.c2 = fp2_mul(fp2_conj(a->c1.c2), fp2_load_const(MBLS_FROB_W[5]));
}
MBLS_FN bool fp12_is_one(const fp12* a) {
    bool z = fp2_is_zero(a->c0.c1) & fp2_is_zero(a->c0.c2) & fp2_is_zero(a->c1.c0) & fp2_is_zero(a->c1.c1) & fp2_is_zero(a->c1.c2);
    return z & fp2_eq(a->c0.c0, fp2_one());
}
// Granger-Scott squaring in the cyclotomic subgroup
MBLS_FN void fp4_sqr(fp2* c0, fp2* c1, const fp2& a, const fp2& b) {
    fp2 t0 = fp2_sqr(a), t1 = fp2_sqr(b);
    *c0 = fp2_add(fp2_mul_xi(t1), t0);
    *c1 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(a, b)), t0), t1);
}
MBLS_TOWER_FN void fp12_cyc_sqr(fp12* r, const fp12* f) {
    fp2 z0 = f->c0.c0, z4 = f->c0.c1, z3 = f->c0.c2, z2 = f->c1.c0, z1 = f->c1.c1, z5 = f->c1.c2;
    fp2 t0, t1, t2, t3;
    fp4_sqr(&t0, &t1, z0, z1);
    z0 = fp2_add(fp2_dbl(fp2_sub(t0, z0)), t0);
    z1 = fp2_add(fp2_dbl(fp2_add(t1, z1)), t1);
    fp4_sqr(&t0, &t1, z2, z3); fp4_sqr(&t2, &t3, z4, z5);
    z4 = fp2_add(fp2_dbl(fp2_sub(t0, z4)), t0);
    z5 = fp2_add(fp2_dbl(fp2_add(t1, z5)), t1);
    t0 = fp2_mul_xi(t3);
    z2 = fp2_add(fp2_dbl(fp2_add(t0, z2)), t0);
    z3 = fp2_add(fp2_dbl(fp2_sub(t2, z3)), t2);
    r->c0.c0 = z0; r->c0.c1 = z4; r->c0.c2 = z3; r->c1.c0 = z2; r->c1.c1 = z1; r->c1.c2 = z5;
}
// f^x, x = -0xd201000000010000, f in the cyclotomic subgroup
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
// The digit-form routines (tools/gen_fpd_asm.py, tools/gen_tower_d.py): every value is 14 signed 28-bit digits from the moment it enters a
// routine until it leaves -- bare product scans, carry-free additions, no conversions or conditional subtractions between
// multiplications; explicit VGPR/AGPR/LDS placement and no lane-private memory (the compiler-scheduled fp12_cyc_sqr spills and,
// with one wave per SIMD, waits a full memory round trip for every reload).
#include "mbls_fpd_asm.inc"
#include "mbls_towerd_asm.inc"
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_mul_d_asm_fn() { asm volatile(MBLS_FP2_MUL_D_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_sqr_d_asm_fn() { asm volatile(MBLS_FP2_SQR_D_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp2_mulfp_d_asm_fn() { asm volatile(MBLS_FP2_MULFP_D_ASM); }
// The whole final exponentiation as ONE routine (final_exp_d_routine): easy part (Fp12 inversion through the fixed-exponent Fp
// inversion, Frobenius^2), five powers by |x| with the running power in AGPRs, the Fp12 products between them; three Fp12 temporaries in
// the workspace. In: f in workspace slots 13..24 (the Miller kernel left it there). `spill`: 11 x 14 dwords per lane of LDS.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_final_exp_d_asm_fn() { asm volatile(MBLS_FINAL_EXP_D_ASM); }
MBLS_FN void final_exp_ws_d(fp12* r, uint32_t* ws_w, uint64_t ws_stride, uint64_t item, MBLS_LDS uint32_t* spill, uint32_t lane) {
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + lane);
    const uint64_t gb = (uint64_t)(uintptr_t)ws_w + 4ull * (item - lane) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws_stride * 4));
    fp f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
    asm volatile(MBLS_ASM_CALL("mbls_final_exp_d_asm_fn")
                 : MBLS_MILLER_D_OUT_REGS(f)
                 : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                 : MBLS_FINAL_EXP_D_ASM_CLOBBERS);
    fp* o = &r->c0.c0.c0;
    o[0] = f0; o[1] = f1; o[2] = f2; o[3] = f3; o[4] = f4; o[5] = f5; o[6] = f6; o[7] = f7; o[8] = f8; o[9] = f9; o[10] = f10; o[11] = f11;
}
// The same routine for lane PAIRS (tools/gen_tower_d.py, final_exp_d_routine(two_lane=True); kernel k_final2): lanes 2 j and 2 j + 1 of the wave
// work on ONE item -- both get the item's workspace words and ONE LDS column (everything outside the squaring chains is computed twice, on
// identical values, so the shared column and the doubled stores are benign) --, in the 315 compressed squarings each lane does one of
// the two Fp4 squarings, the roles swapping every iteration, the partner's coefficients through DPP, and everywhere else independent
// products of one kind share a call (pair_products: the even lane takes the first, the odd lane the second). `item` = the item of THIS lane.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_final_exp2_d_asm_fn() { asm volatile(MBLS_FINAL_EXP2_D_ASM); }
MBLS_FN void final_exp_ws_d2(fp12* r, uint32_t* ws_w, uint64_t ws_stride, uint64_t item, MBLS_LDS uint32_t* spill, uint32_t lane) {
    const uint32_t col = lane >> 1;
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + col);
    const uint64_t gb = (uint64_t)(uintptr_t)ws_w + 4ull * (item - col) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws_stride * 4));
    fp f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
    asm volatile(MBLS_ASM_CALL("mbls_final_exp2_d_asm_fn")
                 : MBLS_MILLER_D_OUT_REGS(f)
                 : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                 : MBLS_FINAL_EXP_D_ASM_CLOBBERS, MBLS_PAIR_EXEC_ASM_CLOBBERS);
    fp* o = &r->c0.c0.c0;
    o[0] = f0; o[1] = f1; o[2] = f2; o[3] = f3; o[4] = f4; o[5] = f5; o[6] = f6; o[7] = f7; o[8] = f8; o[9] = f9; o[10] = f10; o[11] = f11;
}
#endif
// f^x, x = -0xd201000000010000, f in the cyclotomic subgroup (the compiled version: host emulation and debug builds; the kernels run
// final_exp_ws_d)
MBLS_NOINLINE void fp12_cyc_exp_x(fp12* r, const fp12* f) {
    fp12 acc = *f;
    // acc never escapes (the 5 multiplications go through a short-lived copy), so the 63 squarings keep it in registers
    for (int i = 62; i >= 0; i--) {
        fp12_cyc_sqr(&acc, &acc);
        if ((MBLS_X_ABS >> i) & 1) { fp12 t = acc; fp12_mul(&t, &t, f); acc = t; }
    }
    fp12_conj(r, &acc);
}
