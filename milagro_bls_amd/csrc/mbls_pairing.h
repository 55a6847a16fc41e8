// mbls_pairing.h -- optimal-ate Miller loop (shared squaring over the pairs of one lane) and final
// exponentiation. Replaces amcl's pair::{ate2, initmp/another/miller, fexp} behind reference
// src/amcl_utils.rs:38-42 and src/aggregates.rs:142-169,270-315.
//
// Everything projective, no field inversion inside the loop:
//   * T and Q in homogeneous coordinates on the twist E': y^2 = x^3 + 4(1+i);
//   * the G1 argument P = (X:Y:Z) Jacobian enters only through PX = X*Z, PY = Y, PZ3 = Z^3: the line
//     (c0 + c2 xP w^2 + c3 yP w^3) is scaled by Z^3 in Fp, which the final exponentiation kills, like every
//     other factor from a proper subfield of Fp12 that these formulas drop.
// Lines: doubling  (Y^2 - 3b'Z^2) + (-3X^2 xP) w^2 + (2YZ yP) w^3
//        addition  (u X2 - v Y2) + (-u Z2 xP) w^2 + (v Z2 yP) w^3,  u = Y2 Z1 - Y1 Z2, v = X2 Z1 - X1 Z2
// The loop runs over |x| = 0xd201000000010000 (63 doublings, 5 additions; the bit pattern is a
// compile-time constant, so no lane ever diverges) and conjugates at the end because x < 0.
#pragma once
#include "mbls_curve.h"

struct g2h { fp2 x, y, z; };                 // homogeneous projective point on the twist
struct g1arg { fp px, py, pz3; };            // scaled G1 argument (see above)
struct mbls_pair { g2h q; g2h t; g1arg p; bool skip; };   // skip: a member is infinity -> contributes 1

MBLS_FN void g1arg_from_affine(g1arg* a, fp x, fp y) { a->px = x; a->py = y; a->pz3 = fp_one(); }
MBLS_FN void g1arg_from_jacobian(g1arg* a, const g1j* p) {
    a->px = fp_mul(p->x, p->z); a->py = p->y; a->pz3 = fp_mul(fp_sqr(p->z), p->z);
}
MBLS_FN void g2h_from_affine(g2h* h, const fp2& x, const fp2& y) { h->x = x; h->y = y; h->z = fp2_one(); }
MBLS_FN void g2h_from_jacobian(g2h* h, const g2j* p) {   // (X/Z^2, Y/Z^3) = (XZ : Y : Z^3)
    h->x = fp2_mul(p->x, p->z); h->y = p->y; h->z = fp2_mul(fp2_sqr(p->z), p->z);
}

// One doubling step for the pair (T, P): T <- 2T, f <- f * line. T and f are the caller's loop-carried values; the G1
// argument and the skip flag are read from the pair record in lane-private memory when needed.
// AFFINE: the G1 argument is affine (pz3 = 1), so the Z^3 scaling of the constant coefficient is skipped.
template <bool AFFINE>
MBLS_TOWER_FN void miller_dbl_step(fp12* f, g2h* T, const mbls_pair* pr) {
    fp2 B = fp2_sqr(T->y), C = fp2_sqr(T->z);
    fp2 E = fp2_mul12(fp2_mul_xi(C));                 // 3b' Z^2, b' = 4(1+i)
    fp2 F = fp2_mul3(E);
    fp2 X2 = fp2_sqr(T->x);
    fp2 YZ = fp2_mul(T->y, T->z);
    fp2 c0 = fp2_sub(B, E);
    if (!AFFINE) c0 = fp2_mul_fp(c0, pr->p.pz3);
    fp2 c2 = fp2_mul_fp(fp2_neg(fp2_mul3(X2)), pr->p.px);
    fp2 c3 = fp2_mul_fp(fp2_dbl(YZ), pr->p.py);
    fp2 x3 = fp2_dbl(fp2_mul(fp2_mul(T->x, T->y), fp2_sub(B, F)));
    fp2 y3 = fp2_sub(fp2_sqr(fp2_add(B, F)), fp2_mul12(fp2_sqr(E)));
    fp2 z3 = fp2_mul8(fp2_mul(B, YZ));
    T->x = x3; T->y = y3; T->z = z3;
    bool sk = pr->skip;
    c0 = fp2_select(sk, fp2_one(), c0); c2 = fp2_select(sk, fp2_zero(), c2); c3 = fp2_select(sk, fp2_zero(), c3);
    fp12_mul_line(f, f, &c0, &c2, &c3);
}
// Addition step T <- T + Q (5 of the 63 iterations): out of line, on memory operands.
MBLS_TOWER_COLD_FN void miller_add_step(fp12* f, g2h* T, const mbls_pair* pr) {
    const g2h* Q = &pr->q;
    fp2 y1z2 = fp2_mul(T->y, Q->z), x1z2 = fp2_mul(T->x, Q->z), z1z2 = fp2_mul(T->z, Q->z);
    fp2 u = fp2_sub(fp2_mul(Q->y, T->z), y1z2), v = fp2_sub(fp2_mul(Q->x, T->z), x1z2);
    fp2 c0 = fp2_mul_fp(fp2_sub(fp2_mul(u, Q->x), fp2_mul(v, Q->y)), pr->p.pz3);
    fp2 c2 = fp2_mul_fp(fp2_neg(fp2_mul(u, Q->z)), pr->p.px);
    fp2 c3 = fp2_mul_fp(fp2_mul(v, Q->z), pr->p.py);
    fp2 uu = fp2_sqr(u), vv = fp2_sqr(v), vvv = fp2_mul(v, vv), R = fp2_mul(vv, x1z2);
    fp2 A = fp2_sub(fp2_sub(fp2_mul(uu, z1z2), vvv), fp2_dbl(R));
    fp2 x3 = fp2_mul(v, A);
    fp2 y3 = fp2_sub(fp2_mul(u, fp2_sub(R, A)), fp2_mul(vvv, y1z2));
    fp2 z3 = fp2_mul(vvv, z1z2);
    T->x = x3; T->y = y3; T->z = z3;
    bool sk = pr->skip;
    c0 = fp2_select(sk, fp2_one(), c0); c2 = fp2_select(sk, fp2_zero(), c2); c3 = fp2_select(sk, fp2_zero(), c3);
    fp12_mul_line(f, f, &c0, &c2, &c3);
}
// f = prod_k f_{x,Q_k}(P_k) up to subfield factors; pairs[k].t must equal pairs[k].q on entry.
// The loop-carried state (f and the running points T_k) is held in locals whose address never leaves this function
// except through short-lived copies around the rare addition steps, so it can stay in VGPRs/AGPRs across iterations.
// Optional LDS home for the running points T_k: element e of point k of lane l at tstore[(k*72 + e)*64 + l] (conflict-free:
// consecutive lanes hit consecutive banks). Between doubling steps the points then occupy no registers, which leaves the
// register file to f and the line evaluation; tstore == nullptr keeps them in locals.
MBLS_FN void g2h_lds_load(g2h* T, const MBLS_LDS uint32_t* ts, int k, uint32_t lane) {
    const MBLS_LDS uint32_t* p = ts + (uint32_t)k * 72 * 64 + lane;
    fp* c = &T->x.c0;
    for (int e = 0; e < 6; e++) {
        fp v;
#pragma unroll
        for (int j = 0; j < 12; j++) v[j] = p[(e * 12 + j) * 64];
        c[e] = v;
    }
}
MBLS_FN void g2h_lds_store(MBLS_LDS uint32_t* ts, int k, uint32_t lane, const g2h* T) {
    MBLS_LDS uint32_t* p = ts + (uint32_t)k * 72 * 64 + lane;
    const fp* c = &T->x.c0;
    for (int e = 0; e < 6; e++) {
        fp v = c[e];
#pragma unroll
        for (int j = 0; j < 12; j++) p[(e * 12 + j) * 64] = v[j];
    }
}
template <int NP, bool AFFINE0>
MBLS_FN void miller_loop_n(fp12* f_out, mbls_pair* pairs, MBLS_LDS uint32_t* tstore, uint32_t lane, bool use_lds) {
    fp12 f; fp12_set_one(&f);
    g2h T0 = pairs[0].t, T1 = pairs[NP - 1].t;
    if (use_lds) { g2h_lds_store(tstore, 0, lane, &T0); if (NP > 1) g2h_lds_store(tstore, 1, lane, &T1); }
    for (int i = 62; i >= 0; i--) {
        if (i != 62) fp12_sqr(&f, &f);
        if (use_lds) g2h_lds_load(&T0, tstore, 0, lane);
        miller_dbl_step<AFFINE0>(&f, &T0, &pairs[0]);
        if (use_lds) g2h_lds_store(tstore, 0, lane, &T0);
        if (NP > 1) {
            if (use_lds) g2h_lds_load(&T1, tstore, 1, lane);
            miller_dbl_step<false>(&f, &T1, &pairs[NP - 1]);
            if (use_lds) g2h_lds_store(tstore, 1, lane, &T1);
        }
        if ((MBLS_X_ABS >> i) & 1) {
            fp12 ft = f; g2h tt;
            if (use_lds) g2h_lds_load(&tt, tstore, 0, lane); else tt = T0;
            miller_add_step(&ft, &tt, &pairs[0]);
            if (use_lds) g2h_lds_store(tstore, 0, lane, &tt); else T0 = tt;
            if (NP > 1) {
                if (use_lds) g2h_lds_load(&tt, tstore, 1, lane); else tt = T1;
                miller_add_step(&ft, &tt, &pairs[NP - 1]);
                if (use_lds) g2h_lds_store(tstore, 1, lane, &tt); else T1 = tt;
            }
            f = ft;
        }
    }
    fp12_conj(f_out, &f);
}
// npairs = 2 is the verification shape: pair 0 = (signature, -G1) with an affine G1 argument, pair 1 = (H(msg), apk)
MBLS_NOINLINE void miller_loop(fp12* f, mbls_pair* pairs, int npairs) {
    if (npairs == 2) miller_loop_n<2, true>(f, pairs, nullptr, 0, false); else miller_loop_n<1, false>(f, pairs, nullptr, 0, false);
}
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
// The verification shape (tools/gen_tower_d.py): the WHOLE loop -- 63 doubling iterations and the 5 addition steps of both pairs -- as one
// generated routine on 14 signed 28-bit digits per value (bare product scans, carry-free additions, bounds tracked at generation
// time). f lives in AGPRs; the running points, the fixed points Q_k and the second G1 argument live in the HBM workspace as packed
// words and are fetched (prefetched, where a register block is free) when a step needs them, which leaves the whole LDS allocation
// to the routine as spill space. No lane-private memory, no compiler-scheduled step inside the loop.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_miller_loop_d_asm_fn() {
    asm volatile(MBLS_MILLER_LOOP_D_ASM);
}
#define MBLS_MILLER_D_LDS_DWORDS (11 * 14)        // spill slots per lane
#define MBLS_SLOT_QARG 0                          // workspace slots the routine reads: (-px, py, pz^3) of pair 1 over the aggregate key,
#define MBLS_SLOT_Q1 7                            // Q1 in homogeneous form over the Jacobian H(m); slots 3..6 already hold Q0 = the signature
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_miller_loop_1p_d_asm_fn() {
    asm volatile(MBLS_MILLER_LOOP_1P_D_ASM);
}
// pair 1's operands into the workspace slots the routines read, then the call. SINGLE: the one-pair routine (n-pairing paths).
template <bool SINGLE>
MBLS_FN void miller_loop_d_call(fp12* f_out, const mbls_pair* pr1, uint32_t flags, uint32_t* ws_w, uint64_t ws_stride, uint64_t item,
                                MBLS_LDS uint32_t* spill, uint32_t lane) {
    uint32_t* w0 = ws_w + item;
    const fp npx = fp_neg(pr1->p.px);
    const fp* src[9] = {&npx, &pr1->p.py, &pr1->p.pz3, &pr1->q.x.c0, &pr1->q.x.c1, &pr1->q.y.c0, &pr1->q.y.c1, &pr1->q.z.c0, &pr1->q.z.c1};
    const int slot[9] = {MBLS_SLOT_QARG, MBLS_SLOT_QARG + 1, MBLS_SLOT_QARG + 2, MBLS_SLOT_Q1, MBLS_SLOT_Q1 + 1, MBLS_SLOT_Q1 + 2, MBLS_SLOT_Q1 + 3, MBLS_SLOT_Q1 + 4, MBLS_SLOT_Q1 + 5};
#pragma unroll
    for (int t = 0; t < 9; t++) {
        fp v = *src[t];
#pragma unroll
        for (int j = 0; j < 12; j++) w0[((uint64_t)slot[t] * 12 + j) * ws_stride] = v[j];
    }
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + lane);
    // the routine addresses word j of slot s as base + (12 s + j) * stride4 + v252: fold the item offset and the LDS address into the base
    const uint64_t gb = (uint64_t)(uintptr_t)ws_w + 4ull * (item - lane) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws_stride * 4));
    fp f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
    if (SINGLE)
        asm volatile(MBLS_ASM_CALL("mbls_miller_loop_1p_d_asm_fn")
                     : MBLS_MILLER_D_OUT_REGS(f), "+{v253}"(flags)
                     : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                     : MBLS_MILLER_D_ASM_CLOBBERS);
    else
        asm volatile(MBLS_ASM_CALL("mbls_miller_loop_d_asm_fn")
                     : MBLS_MILLER_D_OUT_REGS(f), "+{v253}"(flags)
                     : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                     : MBLS_MILLER_D_ASM_CLOBBERS);
    fp12 f; fp* c = &f.c0.c0.c0;
    c[0] = f0; c[1] = f1; c[2] = f2; c[3] = f3; c[4] = f4; c[5] = f5; c[6] = f6; c[7] = f7; c[8] = f8; c[9] = f9; c[10] = f10; c[11] = f11;
    fp12_conj(f_out, &f);
}
// The one-pair loop on TWO lanes (tools/gen_tower_d.py, pair_products; kernels k_miller_split / k_miller_single with two lanes per pair): lanes
// 2 j and 2 j + 1 of the wave walk the SAME loop on the same values -- same workspace item, one LDS column -- and share its products: two
// independent Fp2 products of one kind are one call, the even lane taking the first, the odd lane the second (20 calls per doubling
// iteration instead of 37). `item` = the workspace item of THIS lane's pair; both lanes come back with the same f.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_miller_loop_1p_pair_d_asm_fn() {
    asm volatile(MBLS_MILLER_LOOP_1P_PAIR_D_ASM);
}
MBLS_FN void miller_loop_single_pair_d(fp12* f_out, const mbls_pair* pr, uint32_t* ws_w, uint64_t ws_stride, uint64_t item,
                                       MBLS_LDS uint32_t* spill, uint32_t lane) {
    uint32_t* w0 = ws_w + item;
    const fp npx = fp_neg(pr->p.px);
    const fp* src[9] = {&npx, &pr->p.py, &pr->p.pz3, &pr->q.x.c0, &pr->q.x.c1, &pr->q.y.c0, &pr->q.y.c1, &pr->q.z.c0, &pr->q.z.c1};
    const int slot[9] = {MBLS_SLOT_QARG, MBLS_SLOT_QARG + 1, MBLS_SLOT_QARG + 2, MBLS_SLOT_Q1, MBLS_SLOT_Q1 + 1, MBLS_SLOT_Q1 + 2, MBLS_SLOT_Q1 + 3, MBLS_SLOT_Q1 + 4, MBLS_SLOT_Q1 + 5};
#pragma unroll
    for (int t = 0; t < 9; t++) {                   // (both lanes of the pair store the same words)
        fp v = *src[t];
#pragma unroll
        for (int j = 0; j < 12; j++) w0[((uint64_t)slot[t] * 12 + j) * ws_stride] = v[j];
    }
    const uint32_t col = lane >> 1;
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + col);
    const uint64_t gb = (uint64_t)(uintptr_t)ws_w + 4ull * (item - col) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws_stride * 4));
    uint32_t flags = pr->skip ? 2u : 0u;
    fp f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
    asm volatile(MBLS_ASM_CALL("mbls_miller_loop_1p_pair_d_asm_fn")
                 : MBLS_MILLER_D_OUT_REGS(f), "+{v253}"(flags)
                 : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                 : MBLS_MILLER_D_ASM_CLOBBERS, MBLS_PAIR_D_ASM_CLOBBERS);
    fp12 f; fp* c = &f.c0.c0.c0;
    c[0] = f0; c[1] = f1; c[2] = f2; c[3] = f3; c[4] = f4; c[5] = f5; c[6] = f6; c[7] = f7; c[8] = f8; c[9] = f9; c[10] = f10; c[11] = f11;
    fp12_conj(f_out, &f);
}
// the verification shape: pair 0 = (signature, -G1) -- the signature already sits in workspace slots 3..6 --, pair 1 = (H(m), apk)
MBLS_FN void miller_loop_verify_d(fp12* f_out, const mbls_pair* pairs, uint32_t* ws_w, uint64_t ws_stride, uint64_t item,
                                  MBLS_LDS uint32_t* spill, uint32_t lane) {
    miller_loop_d_call<false>(f_out, &pairs[1], (pairs[0].skip ? 1u : 0u) | (pairs[1].skip ? 2u : 0u), ws_w, ws_stride, item, spill, lane);
}
// one general pair (Q, P): f_{|x|,Q}(P)
MBLS_FN void miller_loop_single_d(fp12* f_out, const mbls_pair* pr, uint32_t* ws_w, uint64_t ws_stride, uint64_t item,
                                  MBLS_LDS uint32_t* spill, uint32_t lane) {
    miller_loop_d_call<true>(f_out, pr, pr->skip ? 2u : 0u, ws_w, ws_stride, item, spill, lane);
}
#endif
// f^(3 (p^12-1)/r). Hard part: 3 (p^4-p^2+1)/r = (x-1)^2 (x+p) (x^2+p^2-1) + 3 (Hayashida-Hayasaka-Teruya);
// gcd(3, r) = 1, so comparing with 1 gives the same boolean as amcl's fexp (reference src/amcl_utils.rs:40-41).
// This compiled version serves host emulation and debug builds; the kernels run the same sequence as one generated routine
// (final_exp_ws_d in mbls_tower.h, tools/gen_tower_d.py final_exp_d_routine).
MBLS_NOINLINE void final_exp(fp12* r, const fp12* f) {
    fp12 t, u, a, b, c, m;
    fp12_conj(&t, f); fp12_inv(&u, f); fp12_mul(&t, &t, &u);                 // f^(p^6-1)
    fp12_frob(&u, &t); fp12_frob(&u, &u); fp12_mul(&m, &u, &t);               // ^(p^2+1): now cyclotomic
    fp12_cyc_exp_x(&a, &m); fp12_conj(&u, &m); fp12_mul(&a, &a, &u);          // m^(x-1)
    fp12_cyc_exp_x(&t, &a); fp12_conj(&u, &a); fp12_mul(&a, &t, &u);          // m^((x-1)^2)
    fp12_cyc_exp_x(&b, &a); fp12_frob(&u, &a); fp12_mul(&b, &b, &u);          // a^(x+p)
    fp12_cyc_exp_x(&c, &b); fp12_cyc_exp_x(&c, &c);                           // b^(x^2)
    fp12_frob(&u, &b); fp12_frob(&u, &u); fp12_mul(&c, &c, &u);               // * b^(p^2)
    fp12_conj(&u, &b); fp12_mul(&c, &c, &u);                                  // * b^-1
    fp12_cyc_sqr(&u, &m); fp12_mul(&u, &u, &m); fp12_mul(r, &c, &u);          // * m^3
}
