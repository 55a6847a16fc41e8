/*
 * bls_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the verification path of sigp/milagro_bls. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * path (milagro_bls_amd/csrc) never links or calls it.
 *
 * Pinning status: the reference's arithmetic lives in the `amcl` crate (git submodule
 * sigp/incubator-milagro-crypto-rust, branch master, commit unrecoverable: the submodule
 * directory is empty in /root/reference and Cargo.lock is git-ignored), so it cannot be built
 * here. This restatement follows the public algorithms amcl implements (RFC 9380 hash-to-curve
 * suite BLS12381G2_XMD:SHA-256_SSWU_RO_, IETF BLS signature draft-04 POP ciphersuite, the ZCash
 * point serialization) and is pinned by tests/test_oracle_cpu.py against
 *   - every known-answer vector the reference's own tests hold for the path
 *     (reference src/amcl_utils.rs:83-139 codec strings, src/keys.rs:238-350 structural cases,
 *      src/aggregates.rs:384-410 edge cases, the fixed-key property tests :555-609),
 *   - RFC 9380 appendix J.10.1 hash_to_curve vectors and an Eth2 BLS sign vector (external pins),
 *   - the independent big-integer Python model oracle/pymodel/bls12_381.py via tests/golden/.
 *
 * Semantics mirrored (reference file:line given at each function):
 *   decompress/compress        src/amcl_utils.rs:46-74, src/keys.rs:140-175
 *   key_validate               src/keys.rs:181-186
 *   hash_to_curve_g2           src/amcl_utils.rs:33-35
 *   ate2_evaluation            src/amcl_utils.rs:38-42
 *   Signature::new/verify      src/signature.rs:17-40
 *   aggregate / verify family  src/aggregates.rs:29-56,100-124,130-316
 *
 * Representation: Fp = 6 x 64-bit limbs in Montgomery form (R = 2^384); Fp2 = Fp[i]/(i^2+1);
 * Fp6 = Fp2[v]/(v^3-(1+i)); Fp12 = Fp6[w]/(w^2-v); points in homogeneous projective coordinates
 * with the complete formulas of Renes-Costello-Batina (a = 0), like amcl's ECP/ECP2, so `add`
 * handles infinity and doubling (reference src/aggregates.rs:34-37 relies on that).
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include "orc_constants.h"
#include "bls_oracle.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 c0, c1, c2; } fp6;
typedef struct { fp6 c0, c1; } fp12;
typedef struct { fp x, y, z; } g1p;     /* homogeneous projective, infinity = (0,1,0) */
typedef struct { fp2 x, y, z; } g2p;

/* ------------------------------------------------------------------ counters (op census) */
static __thread uint64_t cnt_mul, cnt_sqr;

/* ------------------------------------------------------------------ Fp */
static fp FP_ZERO, FP_ONE, FP_R2;
static const fp *PMOD = (const fp *)ORC_P;

static inline int fp_is_zero(const fp *a) {
    uint64_t t = 0; for (int i = 0; i < 6; i++) t |= a->l[i]; return t == 0;
}
static inline int fp_eq(const fp *a, const fp *b) {
    uint64_t t = 0; for (int i = 0; i < 6; i++) t |= a->l[i] ^ b->l[i]; return t == 0;
}
/* a >= b on raw limbs */
static inline int limbs_geq(const uint64_t *a, const uint64_t *b, int n) {
    for (int i = n - 1; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static inline void fp_add(fp *r, const fp *a, const fp *b) {
    u128 c = 0; uint64_t t[6];
    for (int i = 0; i < 6; i++) { c += (u128)a->l[i] + b->l[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (c || limbs_geq(t, ORC_P, 6)) {
        u128 br = 0;
        for (int i = 0; i < 6; i++) { u128 d = (u128)t[i] - ORC_P[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
    }
    memcpy(r->l, t, sizeof t);
}
static inline void fp_sub(fp *r, const fp *a, const fp *b) {
    u128 br = 0; uint64_t t[6];
    for (int i = 0; i < 6; i++) { u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 6; i++) { c += (u128)t[i] + ORC_P[i]; t[i] = (uint64_t)c; c >>= 64; } }
    memcpy(r->l, t, sizeof t);
}
static inline void fp_neg(fp *r, const fp *a) { if (fp_is_zero(a)) *r = *a; else fp_sub(r, PMOD, a); }
static inline void fp_dbl(fp *r, const fp *a) { fp_add(r, a, a); }

/* Montgomery multiplication, CIOS */
static void fp_mul(fp *r, const fp *a, const fp *b) {
    uint64_t t[8] = {0};
    cnt_mul++;
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * ORC_NP0;
        c = (u128)m * ORC_P[0] + t[0]; c >>= 64;
        for (int j = 1; j < 6; j++) { c += (u128)m * ORC_P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
    }
    if (t[6] || limbs_geq(t, ORC_P, 6)) {
        u128 br = 0;
        for (int i = 0; i < 6; i++) { u128 d = (u128)t[i] - ORC_P[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
    }
    memcpy(r->l, t, 48);
}
static inline void fp_sqr(fp *r, const fp *a) { cnt_sqr++; cnt_mul--; fp_mul(r, a, a); }

static void fp_from_raw(fp *r, const uint64_t raw[6]) { fp t; memcpy(t.l, raw, 48); fp_mul(r, &t, &FP_R2); }
static void fp_to_raw(uint64_t raw[6], const fp *a) { fp one = {{1, 0, 0, 0, 0, 0}}, t; fp_mul(&t, a, &one); memcpy(raw, t.l, 48); }

/* exponent = n little-endian limbs */
static void fp_pow(fp *r, const fp *a, const uint64_t *e, int n) {
    fp acc = FP_ONE, base = *a;
    int top = n * 64 - 1;
    while (top >= 0 && !((e[top / 64] >> (top % 64)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        fp_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, &base);
    }
    *r = acc;
}
static void fp_inv(fp *r, const fp *a) { fp_pow(r, a, ORC_P_MINUS_2, 6); }
/* returns 1 and a root if a is a square */
static int fp_sqrt(fp *r, const fp *a) {
    fp s, t; fp_pow(&s, a, ORC_P_PLUS_1_DIV_4, 6); fp_sqr(&t, &s);
    if (!fp_eq(&t, a)) return 0;
    *r = s; return 1;
}
static int fp_is_square(const fp *a) {
    fp t; if (fp_is_zero(a)) return 1; fp_pow(&t, a, ORC_P_MINUS_1_DIV_2, 6); return fp_eq(&t, &FP_ONE);
}
/* 48-byte big-endian <-> Fp; returns 0 if value >= p */
static int fp_from_be(fp *r, const uint8_t *b) {
    uint64_t raw[6];
    for (int i = 0; i < 6; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v = (v << 8) | b[(5 - i) * 8 + j]; raw[i] = v; }
    if (limbs_geq(raw, ORC_P, 6)) return 0;
    fp_from_raw(r, raw); return 1;
}
static void fp_to_be(uint8_t *b, const fp *a) {
    uint64_t raw[6]; fp_to_raw(raw, a);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 8; j++) b[(5 - i) * 8 + j] = (uint8_t)(raw[i] >> (8 * (7 - j)));
}
static int fp_lex_largest(const fp *a) {     /* a > (p-1)/2 */
    uint64_t raw[6]; fp_to_raw(raw, a);
    return !limbs_geq(ORC_P_MINUS_1_DIV_2, raw, 6);
}
static int fp_is_odd(const fp *a) { uint64_t raw[6]; fp_to_raw(raw, a); return raw[0] & 1; }

/* ------------------------------------------------------------------ Fp2 */
static fp2 F2_ZERO, F2_ONE;
static inline int fp2_is_zero(const fp2 *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline int fp2_eq(const fp2 *a, const fp2 *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static inline void fp2_add(fp2 *r, const fp2 *a, const fp2 *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static inline void fp2_sub(fp2 *r, const fp2 *a, const fp2 *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static inline void fp2_neg(fp2 *r, const fp2 *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static inline void fp2_dbl(fp2 *r, const fp2 *a) { fp2_add(r, a, a); }
static inline void fp2_conj(fp2 *r, const fp2 *a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
static void fp2_mul(fp2 *r, const fp2 *a, const fp2 *b) {
    fp t0, t1, s0, s1, t2;
    fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&s0, &a->c0, &a->c1); fp_add(&s1, &b->c0, &b->c1); fp_mul(&t2, &s0, &s1);
    fp_sub(&r->c0, &t0, &t1);
    fp_sub(&t2, &t2, &t0); fp_sub(&r->c1, &t2, &t1);
}
static void fp2_sqr(fp2 *r, const fp2 *a) {
    fp s, d, m;
    fp_add(&s, &a->c0, &a->c1); fp_sub(&d, &a->c0, &a->c1); fp_mul(&m, &a->c0, &a->c1);
    fp_mul(&r->c0, &s, &d); fp_dbl(&r->c1, &m);
}
static void fp2_mul_fp(fp2 *r, const fp2 *a, const fp *k) { fp_mul(&r->c0, &a->c0, k); fp_mul(&r->c1, &a->c1, k); }
/* multiply by xi = 1+i */
static void fp2_mul_xi(fp2 *r, const fp2 *a) { fp t; fp_sub(&t, &a->c0, &a->c1); fp_add(&r->c1, &a->c0, &a->c1); r->c0 = t; }
static void fp2_inv(fp2 *r, const fp2 *a) {
    fp n, t; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); fp_inv(&n, &n);
    fp_mul(&r->c0, &a->c0, &n); fp_mul(&t, &a->c1, &n); fp_neg(&r->c1, &t);
}
static int fp2_is_square(const fp2 *a) {
    fp n, t; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); return fp_is_square(&n);
}
/* complex-method square root; returns 0 if none */
static int fp2_sqrt(fp2 *r, const fp2 *a) {
    if (fp2_is_zero(a)) { *r = F2_ZERO; return 1; }
    if (fp_is_zero(&a->c1)) {
        fp s;
        if (fp_sqrt(&s, &a->c0)) { r->c0 = s; r->c1 = FP_ZERO; return 1; }
        fp na; fp_neg(&na, &a->c0);
        if (!fp_sqrt(&s, &na)) return 0;
        r->c0 = FP_ZERO; r->c1 = s; return 1;
    }
    fp n, t, inv2, two;
    fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t);
    if (!fp_sqrt(&n, &n)) return 0;
    fp_add(&two, &FP_ONE, &FP_ONE); fp_inv(&inv2, &two);
    for (int k = 0; k < 2; k++) {
        fp x0, x1, d;
        if (k == 0) fp_add(&t, &a->c0, &n); else fp_sub(&t, &a->c0, &n);
        fp_mul(&t, &t, &inv2);
        if (!fp_sqrt(&x0, &t) || fp_is_zero(&x0)) continue;
        fp_dbl(&d, &x0); fp_inv(&d, &d); fp_mul(&x1, &a->c1, &d);
        fp2 cand = {x0, x1}, sq; fp2_sqr(&sq, &cand);
        if (fp2_eq(&sq, a)) { *r = cand; return 1; }
    }
    return 0;
}
static int fp2_sgn0(const fp2 *a) {
    int s0 = fp_is_odd(&a->c0), z0 = fp_is_zero(&a->c0), s1 = fp_is_odd(&a->c1);
    return s0 | (z0 & s1);
}
static int fp2_lex_largest(const fp2 *a) {
    if (!fp_is_zero(&a->c1)) return fp_lex_largest(&a->c1);
    return fp_lex_largest(&a->c0);
}
static void fp2_from_raw(fp2 *r, const uint64_t raw[2][6]) { fp_from_raw(&r->c0, raw[0]); fp_from_raw(&r->c1, raw[1]); }

/* ------------------------------------------------------------------ Fp6, Fp12 */
static void fp6_add(fp6 *r, const fp6 *a, const fp6 *b) { fp2_add(&r->c0, &a->c0, &b->c0); fp2_add(&r->c1, &a->c1, &b->c1); fp2_add(&r->c2, &a->c2, &b->c2); }
static void fp6_sub(fp6 *r, const fp6 *a, const fp6 *b) { fp2_sub(&r->c0, &a->c0, &b->c0); fp2_sub(&r->c1, &a->c1, &b->c1); fp2_sub(&r->c2, &a->c2, &b->c2); }
static void fp6_neg(fp6 *r, const fp6 *a) { fp2_neg(&r->c0, &a->c0); fp2_neg(&r->c1, &a->c1); fp2_neg(&r->c2, &a->c2); }
static void fp6_mul(fp6 *r, const fp6 *a, const fp6 *b) {
    fp2 t0, t1, t2, s, u, c0, c1, c2;
    fp2_mul(&t0, &a->c0, &b->c0); fp2_mul(&t1, &a->c1, &b->c1); fp2_mul(&t2, &a->c2, &b->c2);
    fp2_add(&s, &a->c1, &a->c2); fp2_add(&u, &b->c1, &b->c2); fp2_mul(&c0, &s, &u);
    fp2_sub(&c0, &c0, &t1); fp2_sub(&c0, &c0, &t2); fp2_mul_xi(&c0, &c0); fp2_add(&c0, &c0, &t0);
    fp2_add(&s, &a->c0, &a->c1); fp2_add(&u, &b->c0, &b->c1); fp2_mul(&c1, &s, &u);
    fp2_sub(&c1, &c1, &t0); fp2_sub(&c1, &c1, &t1); fp2_mul_xi(&s, &t2); fp2_add(&c1, &c1, &s);
    fp2_add(&s, &a->c0, &a->c2); fp2_add(&u, &b->c0, &b->c2); fp2_mul(&c2, &s, &u);
    fp2_sub(&c2, &c2, &t0); fp2_sub(&c2, &c2, &t2); fp2_add(&c2, &c2, &t1);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
/* multiply by v: (c0,c1,c2) -> (xi c2, c0, c1) */
static void fp6_mul_v(fp6 *r, const fp6 *a) { fp2 t; fp2_mul_xi(&t, &a->c2); r->c2 = a->c1; r->c1 = a->c0; r->c0 = t; }
/* multiply by sparse (x, y, 0) */
static void fp6_mul_01(fp6 *r, const fp6 *a, const fp2 *x, const fp2 *y) {
    fp2 t0, t1, s, u, c0, c1, c2;
    fp2_mul(&t0, &a->c0, x); fp2_mul(&t1, &a->c1, y);
    fp2_add(&s, &a->c0, &a->c1); fp2_add(&u, x, y); fp2_mul(&c1, &s, &u); fp2_sub(&c1, &c1, &t0); fp2_sub(&c1, &c1, &t1);
    fp2_mul(&c0, &a->c2, y); fp2_mul_xi(&c0, &c0); fp2_add(&c0, &c0, &t0);
    fp2_mul(&c2, &a->c2, x); fp2_add(&c2, &c2, &t1);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
/* multiply by sparse (0, y, 0) */
static void fp6_mul_1(fp6 *r, const fp6 *a, const fp2 *y) {
    fp2 c0, c1, c2;
    fp2_mul(&c0, &a->c2, y); fp2_mul_xi(&c0, &c0); fp2_mul(&c1, &a->c0, y); fp2_mul(&c2, &a->c1, y);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
static void fp6_inv(fp6 *r, const fp6 *a) {
    fp2 A, B, C, t, F;
    fp2_sqr(&A, &a->c0); fp2_mul(&t, &a->c1, &a->c2); fp2_mul_xi(&t, &t); fp2_sub(&A, &A, &t);
    fp2_sqr(&B, &a->c2); fp2_mul_xi(&B, &B); fp2_mul(&t, &a->c0, &a->c1); fp2_sub(&B, &B, &t);
    fp2_sqr(&C, &a->c1); fp2_mul(&t, &a->c0, &a->c2); fp2_sub(&C, &C, &t);
    fp2_mul(&F, &a->c2, &B); fp2_mul(&t, &a->c1, &C); fp2_add(&F, &F, &t); fp2_mul_xi(&F, &F);
    fp2_mul(&t, &a->c0, &A); fp2_add(&F, &F, &t); fp2_inv(&F, &F);
    fp2_mul(&r->c0, &A, &F); fp2_mul(&r->c1, &B, &F); fp2_mul(&r->c2, &C, &F);
}
static fp12 F12_ONE;
static fp2 FROB_W[6], PSI_CX, PSI_CY;
static void fp12_mul(fp12 *r, const fp12 *a, const fp12 *b) {
    fp6 t0, t1, s, u, c1;
    fp6_mul(&t0, &a->c0, &b->c0); fp6_mul(&t1, &a->c1, &b->c1);
    fp6_add(&s, &a->c0, &a->c1); fp6_add(&u, &b->c0, &b->c1); fp6_mul(&c1, &s, &u);
    fp6_sub(&c1, &c1, &t0); fp6_sub(&c1, &c1, &t1);
    fp6_mul_v(&s, &t1); fp6_add(&r->c0, &t0, &s); r->c1 = c1;
}
static void fp12_sqr(fp12 *r, const fp12 *a) {
    fp6 ab, s, t, va;
    fp6_mul(&ab, &a->c0, &a->c1);
    fp6_add(&s, &a->c0, &a->c1); fp6_mul_v(&va, &a->c1); fp6_add(&t, &a->c0, &va);
    fp6_mul(&s, &s, &t); fp6_sub(&s, &s, &ab); fp6_mul_v(&t, &ab); fp6_sub(&r->c0, &s, &t);
    fp6_add(&r->c1, &ab, &ab);
}
/* f * (c0 + c2 w^2 + c3 w^3): tower positions a0 = c0, a1 = c2, b1 = c3 */
static void fp12_mul_line(fp12 *r, const fp12 *f, const fp2 *c0, const fp2 *c2, const fp2 *c3) {
    fp6 t0, t1, s, c1; fp2 y;
    fp6_mul_01(&t0, &f->c0, c0, c2); fp6_mul_1(&t1, &f->c1, c3);
    fp6_add(&s, &f->c0, &f->c1); fp2_add(&y, c2, c3); fp6_mul_01(&c1, &s, c0, &y);
    fp6_sub(&c1, &c1, &t0); fp6_sub(&c1, &c1, &t1);
    fp6_mul_v(&s, &t1); fp6_add(&r->c0, &t0, &s); r->c1 = c1;
}
static void fp12_conj(fp12 *r, const fp12 *a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }
static void fp12_inv(fp12 *r, const fp12 *a) {
    fp6 t0, t1; fp6_mul(&t0, &a->c0, &a->c0); fp6_mul(&t1, &a->c1, &a->c1); fp6_mul_v(&t1, &t1); fp6_sub(&t0, &t0, &t1);
    fp6_inv(&t0, &t0); fp6_mul(&r->c0, &a->c0, &t0); fp6_mul(&t1, &a->c1, &t0); fp6_neg(&r->c1, &t1);
}
/* tower coefficient of w^k: k even -> c0.c(k/2), k odd -> c1.c((k-1)/2) */
static fp2 *fp12_wcoef(fp12 *a, int k) { fp6 *h = (k & 1) ? &a->c1 : &a->c0; int j = k >> 1; return j == 0 ? &h->c0 : (j == 1 ? &h->c1 : &h->c2); }
static void fp12_frob(fp12 *r, const fp12 *a) {
    fp12 t = *a;
    for (int k = 0; k < 6; k++) { fp2 *c = fp12_wcoef(&t, k); fp2 cj; fp2_conj(&cj, c); fp2_mul(c, &cj, &FROB_W[k]); }
    *r = t;
}
static int fp12_is_one(const fp12 *a) { return memcmp(a, &F12_ONE, sizeof(fp12)) == 0; }

/* Granger-Scott squaring in the cyclotomic subgroup */
static void fp4_sqr(fp2 *c0, fp2 *c1, const fp2 *a, const fp2 *b) {
    fp2 t0, t1, t2;
    fp2_sqr(&t0, a); fp2_sqr(&t1, b); fp2_mul_xi(&t2, &t1); fp2_add(c0, &t2, &t0);
    fp2_add(&t2, a, b); fp2_sqr(&t2, &t2); fp2_sub(&t2, &t2, &t0); fp2_sub(c1, &t2, &t1);
}
static void fp12_cyc_sqr(fp12 *r, const fp12 *f) {
    fp2 z0 = f->c0.c0, z4 = f->c0.c1, z3 = f->c0.c2, z2 = f->c1.c0, z1 = f->c1.c1, z5 = f->c1.c2;
    fp2 t0, t1, t2, t3;
    fp4_sqr(&t0, &t1, &z0, &z1);
    fp2_sub(&z0, &t0, &z0); fp2_dbl(&z0, &z0); fp2_add(&z0, &z0, &t0);
    fp2_add(&z1, &t1, &z1); fp2_dbl(&z1, &z1); fp2_add(&z1, &z1, &t1);
    fp4_sqr(&t0, &t1, &z2, &z3); fp4_sqr(&t2, &t3, &z4, &z5);
    fp2_sub(&z4, &t0, &z4); fp2_dbl(&z4, &z4); fp2_add(&z4, &z4, &t0);
    fp2_add(&z5, &t1, &z5); fp2_dbl(&z5, &z5); fp2_add(&z5, &z5, &t1);
    fp2_mul_xi(&t0, &t3);
    fp2_add(&z2, &t0, &z2); fp2_dbl(&z2, &z2); fp2_add(&z2, &z2, &t0);
    fp2_sub(&z3, &t2, &z3); fp2_dbl(&z3, &z3); fp2_add(&z3, &z3, &t2);
    r->c0.c0 = z0; r->c0.c1 = z4; r->c0.c2 = z3; r->c1.c0 = z2; r->c1.c1 = z1; r->c1.c2 = z5;
}
/* f^x for f in the cyclotomic subgroup, x = -X_ABS */
static void fp12_cyc_exp_x(fp12 *r, const fp12 *f) {
    fp12 acc = *f;
    for (int i = 62; i >= 0; i--) {
        fp12_cyc_sqr(&acc, &acc);
        if ((ORC_X_ABS >> i) & 1) fp12_mul(&acc, &acc, f);
    }
    fp12_conj(r, &acc);
}

/* ------------------------------------------------------------------ G1 (complete formulas, a = 0, b = 4) */
static fp B3_G1; static fp2 B3_G2; static fp FP_B1; static fp2 FP2_B2;
static g1p G1_GEN, G1_NEG_GEN; static g2p G2_GEN;
static void g1_set_inf(g1p *p) { p->x = FP_ZERO; p->y = FP_ONE; p->z = FP_ZERO; }
static int g1_is_inf(const g1p *p) { return fp_is_zero(&p->z); }
static void g1_add(g1p *r, const g1p *p, const g1p *q) {
    fp t0, t1, t2, t3, t4, x3, y3, z3;
    fp_mul(&t0, &p->x, &q->x); fp_mul(&t1, &p->y, &q->y); fp_mul(&t2, &p->z, &q->z);
    fp_add(&t3, &p->x, &p->y); fp_add(&t4, &q->x, &q->y); fp_mul(&t3, &t3, &t4);
    fp_add(&t4, &t0, &t1); fp_sub(&t3, &t3, &t4); fp_add(&t4, &p->y, &p->z);
    fp_add(&x3, &q->y, &q->z); fp_mul(&t4, &t4, &x3); fp_add(&x3, &t1, &t2);
    fp_sub(&t4, &t4, &x3); fp_add(&x3, &p->x, &p->z); fp_add(&y3, &q->x, &q->z);
    fp_mul(&x3, &x3, &y3); fp_add(&y3, &t0, &t2); fp_sub(&y3, &x3, &y3);
    fp_add(&x3, &t0, &t0); fp_add(&t0, &x3, &t0); fp_mul(&t2, &B3_G1, &t2);
    fp_add(&z3, &t1, &t2); fp_sub(&t1, &t1, &t2); fp_mul(&y3, &B3_G1, &y3);
    fp_mul(&x3, &t4, &y3); fp_mul(&t2, &t3, &t1); fp_sub(&x3, &t2, &x3);
    fp_mul(&y3, &y3, &t0); fp_mul(&t1, &t1, &z3); fp_add(&y3, &t1, &y3);
    fp_mul(&t0, &t0, &t3); fp_mul(&z3, &z3, &t4); fp_add(&z3, &z3, &t0);
    r->x = x3; r->y = y3; r->z = z3;
}
static void g1_dbl(g1p *r, const g1p *p) {
    fp t0, t1, t2, x3, y3, z3;
    fp_sqr(&t0, &p->y); fp_add(&z3, &t0, &t0); fp_add(&z3, &z3, &z3); fp_add(&z3, &z3, &z3);
    fp_mul(&t1, &p->y, &p->z); fp_sqr(&t2, &p->z); fp_mul(&t2, &B3_G1, &t2);
    fp_mul(&x3, &t2, &z3); fp_add(&y3, &t0, &t2); fp_mul(&z3, &t1, &z3);
    fp_add(&t1, &t2, &t2); fp_add(&t2, &t1, &t2); fp_sub(&t0, &t0, &t2);
    fp_mul(&y3, &t0, &y3); fp_add(&y3, &x3, &y3); fp_mul(&t1, &p->x, &p->y);
    fp_mul(&x3, &t0, &t1); fp_add(&x3, &x3, &x3);
    r->x = x3; r->y = y3; r->z = z3;
}
static void g1_neg(g1p *r, const g1p *p) { r->x = p->x; fp_neg(&r->y, &p->y); r->z = p->z; }
static void g1_affine(g1p *p) {
    if (g1_is_inf(p)) { g1_set_inf(p); return; }
    fp zi; fp_inv(&zi, &p->z); fp_mul(&p->x, &p->x, &zi); fp_mul(&p->y, &p->y, &zi); p->z = FP_ONE;
}
/* scalar = n little-endian 64-bit limbs */
static void g1_mul(g1p *r, const g1p *p, const uint64_t *k, int n) {
    g1p acc; g1_set_inf(&acc);
    for (int i = n * 64 - 1; i >= 0; i--) { g1_dbl(&acc, &acc); if ((k[i / 64] >> (i % 64)) & 1) g1_add(&acc, &acc, p); }
    *r = acc;
}
static int g1_on_curve_affine(const fp *x, const fp *y) {
    fp l, r; fp_sqr(&l, y); fp_sqr(&r, x); fp_mul(&r, &r, x); fp_add(&r, &r, &FP_B1); return fp_eq(&l, &r);
}

/* ------------------------------------------------------------------ G2 */
static void g2_set_inf(g2p *p) { p->x = F2_ZERO; p->y = F2_ONE; p->z = F2_ZERO; }
static int g2_is_inf(const g2p *p) { return fp2_is_zero(&p->z); }
static void g2_add(g2p *r, const g2p *p, const g2p *q) {
    fp2 t0, t1, t2, t3, t4, x3, y3, z3;
    fp2_mul(&t0, &p->x, &q->x); fp2_mul(&t1, &p->y, &q->y); fp2_mul(&t2, &p->z, &q->z);
    fp2_add(&t3, &p->x, &p->y); fp2_add(&t4, &q->x, &q->y); fp2_mul(&t3, &t3, &t4);
    fp2_add(&t4, &t0, &t1); fp2_sub(&t3, &t3, &t4); fp2_add(&t4, &p->y, &p->z);
    fp2_add(&x3, &q->y, &q->z); fp2_mul(&t4, &t4, &x3); fp2_add(&x3, &t1, &t2);
    fp2_sub(&t4, &t4, &x3); fp2_add(&x3, &p->x, &p->z); fp2_add(&y3, &q->x, &q->z);
    fp2_mul(&x3, &x3, &y3); fp2_add(&y3, &t0, &t2); fp2_sub(&y3, &x3, &y3);
    fp2_add(&x3, &t0, &t0); fp2_add(&t0, &x3, &t0); fp2_mul(&t2, &B3_G2, &t2);
    fp2_add(&z3, &t1, &t2); fp2_sub(&t1, &t1, &t2); fp2_mul(&y3, &B3_G2, &y3);
    fp2_mul(&x3, &t4, &y3); fp2_mul(&t2, &t3, &t1); fp2_sub(&x3, &t2, &x3);
    fp2_mul(&y3, &y3, &t0); fp2_mul(&t1, &t1, &z3); fp2_add(&y3, &t1, &y3);
    fp2_mul(&t0, &t0, &t3); fp2_mul(&z3, &z3, &t4); fp2_add(&z3, &z3, &t0);
    r->x = x3; r->y = y3; r->z = z3;
}
static void g2_dbl(g2p *r, const g2p *p) {
    fp2 t0, t1, t2, x3, y3, z3;
    fp2_sqr(&t0, &p->y); fp2_add(&z3, &t0, &t0); fp2_add(&z3, &z3, &z3); fp2_add(&z3, &z3, &z3);
    fp2_mul(&t1, &p->y, &p->z); fp2_sqr(&t2, &p->z); fp2_mul(&t2, &B3_G2, &t2);
    fp2_mul(&x3, &t2, &z3); fp2_add(&y3, &t0, &t2); fp2_mul(&z3, &t1, &z3);
    fp2_add(&t1, &t2, &t2); fp2_add(&t2, &t1, &t2); fp2_sub(&t0, &t0, &t2);
    fp2_mul(&y3, &t0, &y3); fp2_add(&y3, &x3, &y3); fp2_mul(&t1, &p->x, &p->y);
    fp2_mul(&x3, &t0, &t1); fp2_add(&x3, &x3, &x3);
    r->x = x3; r->y = y3; r->z = z3;
}
static void g2_neg(g2p *r, const g2p *p) { r->x = p->x; fp2_neg(&r->y, &p->y); r->z = p->z; }
static void g2_affine(g2p *p) {
    if (g2_is_inf(p)) { g2_set_inf(p); return; }
    fp2 zi; fp2_inv(&zi, &p->z); fp2_mul(&p->x, &p->x, &zi); fp2_mul(&p->y, &p->y, &zi); p->z = F2_ONE;
}
static void g2_mul(g2p *r, const g2p *p, const uint64_t *k, int n) {
    g2p acc; g2_set_inf(&acc);
    for (int i = n * 64 - 1; i >= 0; i--) { g2_dbl(&acc, &acc); if ((k[i / 64] >> (i % 64)) & 1) g2_add(&acc, &acc, p); }
    *r = acc;
}
static int g2_on_curve_affine(const fp2 *x, const fp2 *y) {
    fp2 l, r; fp2_sqr(&l, y); fp2_sqr(&r, x); fp2_mul(&r, &r, x); fp2_add(&r, &r, &FP2_B2); return fp2_eq(&l, &r);
}
static void g2_psi(g2p *r, const g2p *p) {   /* homogeneous: psi acts coordinate-wise with z -> conj(z) */
    fp2 t; fp2_conj(&t, &p->x); fp2_mul(&r->x, &t, &PSI_CX);
    fp2_conj(&t, &p->y); fp2_mul(&r->y, &t, &PSI_CY);
    fp2_conj(&r->z, &p->z);
}
/* amcl subgroup checks: [r]P == O (reference src/keys.rs:182, src/signature.rs:29). Infinity passes. */
static int g1_in_subgroup(const g1p *p) { g1p t; g1_mul(&t, p, ORC_ORDER, 4); return g1_is_inf(&t); }
static int g2_in_subgroup(const g2p *p) { g2p t; g2_mul(&t, p, ORC_ORDER, 4); return g2_is_inf(&t); }

/* ------------------------------------------------------------------ serialization (ZCash format) */
/* "decoded point" at this library's boundary = canonical uncompressed bytes:
   G1: x||y (96 B), infinity = 0x40||0...;  G2: x.c1||x.c0||y.c1||y.c0 (192 B), infinity = 0x40||0... */
static int all_zero(const uint8_t *b, size_t n) { uint8_t t = 0; for (size_t i = 0; i < n; i++) t |= b[i]; return t == 0; }

static int g1_from_unc(g1p *p, const uint8_t b[96]) {     /* reference src/keys.rs:170-175 */
    if (b[0] & 0x80) return ORC_ERR_G1_SIZE;               /* compressed flag on a 96-byte string */
    if (b[0] & 0x40) { if ((b[0] & 0x3F) || !all_zero(b + 1, 95)) return ORC_ERR_POINT; g1_set_inf(p); return ORC_OK; }
    if (b[0] & 0x20) return ORC_ERR_POINT;
    if (!fp_from_be(&p->x, b) || !fp_from_be(&p->y, b + 48)) return ORC_ERR_POINT;
    if (!g1_on_curve_affine(&p->x, &p->y)) return ORC_ERR_POINT;
    p->z = FP_ONE; return ORC_OK;
}
static void g1_to_unc(uint8_t b[96], const g1p *p_) {
    g1p p = *p_; g1_affine(&p);
    if (g1_is_inf(&p)) { memset(b, 0, 96); b[0] = 0x40; return; }
    fp_to_be(b, &p.x); fp_to_be(b + 48, &p.y);
}
static int g1_from_comp(g1p *p, const uint8_t b[48]) {    /* reference src/amcl_utils.rs:52-58 */
    if (!(b[0] & 0x80)) return ORC_ERR_G1_SIZE;
    if (b[0] & 0x40) { if ((b[0] & 0x3F) || !all_zero(b + 1, 47)) return ORC_ERR_POINT; g1_set_inf(p); return ORC_OK; }
    uint8_t t[48]; memcpy(t, b, 48); t[0] &= 0x1F;
    if (!fp_from_be(&p->x, t)) return ORC_ERR_POINT;
    fp y2; fp_sqr(&y2, &p->x); fp_mul(&y2, &y2, &p->x); fp_add(&y2, &y2, &FP_B1);
    if (!fp_sqrt(&p->y, &y2)) return ORC_ERR_POINT;
    if (fp_lex_largest(&p->y) != !!(b[0] & 0x20)) fp_neg(&p->y, &p->y);
    p->z = FP_ONE; return ORC_OK;
}
static void g1_to_comp(uint8_t b[48], const g1p *p_) {    /* reference src/amcl_utils.rs:46-48 */
    g1p p = *p_; g1_affine(&p);
    if (g1_is_inf(&p)) { memset(b, 0, 48); b[0] = 0xC0; return; }
    fp_to_be(b, &p.x); b[0] |= 0x80; if (fp_lex_largest(&p.y)) b[0] |= 0x20;
}
static int g2_from_unc(g2p *p, const uint8_t b[192]) {
    if (b[0] & 0x80) return ORC_ERR_G2_SIZE;
    if (b[0] & 0x40) { if ((b[0] & 0x3F) || !all_zero(b + 1, 191)) return ORC_ERR_POINT; g2_set_inf(p); return ORC_OK; }
    if (b[0] & 0x20) return ORC_ERR_POINT;
    if (!fp_from_be(&p->x.c1, b) || !fp_from_be(&p->x.c0, b + 48) || !fp_from_be(&p->y.c1, b + 96) || !fp_from_be(&p->y.c0, b + 144)) return ORC_ERR_POINT;
    if (!g2_on_curve_affine(&p->x, &p->y)) return ORC_ERR_POINT;
    p->z = F2_ONE; return ORC_OK;
}
static void g2_to_unc(uint8_t b[192], const g2p *p_) {
    g2p p = *p_; g2_affine(&p);
    if (g2_is_inf(&p)) { memset(b, 0, 192); b[0] = 0x40; return; }
    fp_to_be(b, &p.x.c1); fp_to_be(b + 48, &p.x.c0); fp_to_be(b + 96, &p.y.c1); fp_to_be(b + 144, &p.y.c0);
}
static int g2_from_comp(g2p *p, const uint8_t b[96]) {    /* reference src/amcl_utils.rs:68-74 */
    if (!(b[0] & 0x80)) return ORC_ERR_G2_SIZE;
    if (b[0] & 0x40) { if ((b[0] & 0x3F) || !all_zero(b + 1, 95)) return ORC_ERR_POINT; g2_set_inf(p); return ORC_OK; }
    uint8_t t[48]; memcpy(t, b, 48); t[0] &= 0x1F;
    if (!fp_from_be(&p->x.c1, t) || !fp_from_be(&p->x.c0, b + 48)) return ORC_ERR_POINT;
    fp2 y2; fp2_sqr(&y2, &p->x); fp2_mul(&y2, &y2, &p->x); fp2_add(&y2, &y2, &FP2_B2);
    if (!fp2_sqrt(&p->y, &y2)) return ORC_ERR_POINT;
    if (fp2_lex_largest(&p->y) != !!(b[0] & 0x20)) fp2_neg(&p->y, &p->y);
    p->z = F2_ONE; return ORC_OK;
}
static void g2_to_comp(uint8_t b[96], const g2p *p_) {    /* reference src/amcl_utils.rs:62-64 */
    g2p p = *p_; g2_affine(&p);
    if (g2_is_inf(&p)) { memset(b, 0, 96); b[0] = 0xC0; return; }
    fp_to_be(b, &p.x.c1); fp_to_be(b + 48, &p.x.c0); b[0] |= 0x80; if (fp2_lex_largest(&p.y)) b[0] |= 0x20;
}

/* ------------------------------------------------------------------ SHA-256 */
static const uint32_t SHA_K[64] = {
0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; } sha256_ctx;
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t h[8], const uint8_t *p) {
    uint32_t w[64], a, b, c, d, e, f, g, hh;
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    a = h[0]; b = h[1]; c = h[2]; d = h[3]; e = h[4]; f = h[5]; g = h[6]; hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + SHA_K[i] + w[i];
        uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
static void sha256_init(sha256_ctx *c) {
    static const uint32_t iv[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
    memcpy(c->h, iv, 32); c->len = 0;
}
static void sha256_update(sha256_ctx *c, const uint8_t *p, size_t n) {
    while (n) {
        size_t off = c->len % 64, take = 64 - off; if (take > n) take = n;
        memcpy(c->buf + off, p, take); c->len += take; p += take; n -= take;
        if (c->len % 64 == 0) sha256_block(c->h, c->buf);
    }
}
static void sha256_final(sha256_ctx *c, uint8_t out[32]) {
    uint64_t bits = c->len * 8; uint8_t pad = 0x80; sha256_update(c, &pad, 1);
    uint8_t z = 0; while (c->len % 64 != 56) sha256_update(c, &z, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (8 * (7 - i)));
    sha256_update(c, lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = c->h[i] >> 24; out[4 * i + 1] = c->h[i] >> 16; out[4 * i + 2] = c->h[i] >> 8; out[4 * i + 3] = c->h[i]; }
}

/* ------------------------------------------------------------------ hash to G2 (RFC 9380) */
static fp2 SSWU_A, SSWU_B, SSWU_Z, SSWU_MBA, SSWU_BZA;
static fp2 ISO_XNUM[4], ISO_XDEN[3], ISO_YNUM[4], ISO_YDEN[4];

static void expand_message_xmd(uint8_t *out, size_t n, const uint8_t *msg, size_t mlen, const uint8_t *dst, size_t dlen) {
    uint8_t b0[32], bi[32], zpad[64] = {0}, tmp[32];
    size_t ell = (n + 31) / 32;
    uint8_t dl = (uint8_t)dlen, lib[3] = {(uint8_t)(n >> 8), (uint8_t)n, 0};
    sha256_ctx c;
    sha256_init(&c); sha256_update(&c, zpad, 64); sha256_update(&c, msg, mlen); sha256_update(&c, lib, 3);
    sha256_update(&c, dst, dlen); sha256_update(&c, &dl, 1); sha256_final(&c, b0);
    uint8_t one = 1;
    sha256_init(&c); sha256_update(&c, b0, 32); sha256_update(&c, &one, 1); sha256_update(&c, dst, dlen); sha256_update(&c, &dl, 1); sha256_final(&c, bi);
    size_t off = 0;
    for (size_t i = 1; i <= ell; i++) {
        size_t take = n - off < 32 ? n - off : 32; memcpy(out + off, bi, take); off += take;
        if (i == ell) break;
        for (int k = 0; k < 32; k++) tmp[k] = b0[k] ^ bi[k];
        uint8_t idx = (uint8_t)(i + 1);
        sha256_init(&c); sha256_update(&c, tmp, 32); sha256_update(&c, &idx, 1); sha256_update(&c, dst, dlen); sha256_update(&c, &dl, 1); sha256_final(&c, bi);
    }
}
/* 64 big-endian bytes mod p -> Fp (Montgomery) */
static void fp_from_be64(fp *r, const uint8_t *b) {
    /* value = hi * 2^256 + lo, hi = first 32 bytes, lo = last 32 bytes; both < p */
    uint8_t t[48]; fp hi, lo, two256; uint64_t raw[6] = {0, 0, 0, 0, 1, 0};
    memset(t, 0, 16); memcpy(t + 16, b, 32); fp_from_be(&hi, t);
    memcpy(t + 16, b + 32, 32); fp_from_be(&lo, t);
    fp_from_raw(&two256, raw); fp_mul(&hi, &hi, &two256); fp_add(r, &hi, &lo);
}
static void sswu_g2(fp2 *x, fp2 *y, const fp2 *u) {
    fp2 u2, zu2, den, x1, gx1, x2, gx2, t;
    fp2_sqr(&u2, u); fp2_mul(&zu2, &SSWU_Z, &u2);
    fp2_sqr(&den, &zu2); fp2_add(&den, &den, &zu2);
    if (fp2_is_zero(&den)) x1 = SSWU_BZA;
    else { fp2_inv(&t, &den); fp2_add(&t, &t, &F2_ONE); fp2_mul(&x1, &SSWU_MBA, &t); }
    fp2_sqr(&gx1, &x1); fp2_add(&gx1, &gx1, &SSWU_A); fp2_mul(&gx1, &gx1, &x1); fp2_add(&gx1, &gx1, &SSWU_B);
    fp2_mul(&x2, &zu2, &x1);
    fp2_sqr(&gx2, &x2); fp2_add(&gx2, &gx2, &SSWU_A); fp2_mul(&gx2, &gx2, &x2); fp2_add(&gx2, &gx2, &SSWU_B);
    if (fp2_is_square(&gx1)) { *x = x1; fp2_sqrt(y, &gx1); } else { *x = x2; fp2_sqrt(y, &gx2); }
    if (fp2_sgn0(u) != fp2_sgn0(y)) fp2_neg(y, y);
}
static void poly_eval(fp2 *r, const fp2 *c, int n, const fp2 *x) {
    fp2 acc = c[n - 1];
    for (int i = n - 2; i >= 0; i--) { fp2_mul(&acc, &acc, x); fp2_add(&acc, &acc, &c[i]); }
    *r = acc;
}
static void iso3_g2(g2p *r, const fp2 *x, const fp2 *y) {
    fp2 xn, xd, yn, yd;
    poly_eval(&xn, ISO_XNUM, 4, x); poly_eval(&xd, ISO_XDEN, 3, x); poly_eval(&yn, ISO_YNUM, 4, x); poly_eval(&yd, ISO_YDEN, 4, x);
    if (fp2_is_zero(&xd) || fp2_is_zero(&yd)) { g2_set_inf(r); return; }
    /* homogeneous: X = xn*yd, Y = y*yn*xd, Z = xd*yd */
    fp2_mul(&r->x, &xn, &yd); fp2_mul(&r->y, y, &yn); fp2_mul(&r->y, &r->y, &xd); fp2_mul(&r->z, &xd, &yd);
}
static void g2_mul_x(g2p *r, const g2p *p) {      /* [x]P with x = -X_ABS */
    uint64_t k = ORC_X_ABS; g2p t; g2_mul(&t, p, &k, 1); g2_neg(r, &t);
}
static void clear_cofactor_g2(g2p *r, const g2p *p) {   /* Budroni-Pintore, RFC 9380 appendix G.3 */
    g2p t1, t2, t3, n;
    g2_mul_x(&t1, p); g2_psi(&t2, p);
    g2_dbl(&t3, p); g2_psi(&t3, &t3); g2_psi(&t3, &t3);
    g2_neg(&n, &t2); g2_add(&t3, &t3, &n);
    g2_add(&t2, &t1, &t2); g2_mul_x(&t2, &t2);
    g2_add(&t3, &t3, &t2); g2_neg(&n, &t1); g2_add(&t3, &t3, &n);
    g2_neg(&n, p); g2_add(r, &t3, &n);
}
/* reference src/amcl_utils.rs:33-35 */
static void hash_to_g2(g2p *r, const uint8_t *msg, size_t mlen, const uint8_t *dst, size_t dlen) {
    uint8_t ub[256]; fp2 u0, u1, x, y; g2p q0, q1;
    expand_message_xmd(ub, 256, msg, mlen, dst, dlen);
    fp_from_be64(&u0.c0, ub); fp_from_be64(&u0.c1, ub + 64); fp_from_be64(&u1.c0, ub + 128); fp_from_be64(&u1.c1, ub + 192);
    sswu_g2(&x, &y, &u0); iso3_g2(&q0, &x, &y);
    sswu_g2(&x, &y, &u1); iso3_g2(&q1, &x, &y);
    g2_add(&q0, &q0, &q1); clear_cofactor_g2(r, &q0);
}
static const uint8_t DST_POP[] = "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_";
#define DST_POP_LEN 43

/* ------------------------------------------------------------------ pairing */
/* One Miller accumulator step for the pair (Q on twist affine, P in G1 affine):
   lines scaled by subfield factors (killed by the final exponentiation):
     doubling  T=(X,Y,Z): (Y^2 - 3b'Z^2) + (-3X^2 xP) w^2 + (2YZ yP) w^3
     addition  T+Q, th = Y - yQ Z, mu = X - xQ Z: (th xQ - mu yQ) + (-th xP) w^2 + (mu yP) w^3 */
static void line_dbl(fp12 *f, g2p *T, const fp *xP, const fp *yP) {
    fp2 c0, c2, c3, t;
    fp2_sqr(&c0, &T->y); fp2_sqr(&t, &T->z); fp2_mul(&t, &t, &B3_G2); fp2_sub(&c0, &c0, &t);
    fp2_sqr(&c2, &T->x); fp2_dbl(&t, &c2); fp2_add(&c2, &c2, &t); fp2_neg(&c2, &c2); fp2_mul_fp(&c2, &c2, xP);
    fp2_mul(&c3, &T->y, &T->z); fp2_dbl(&c3, &c3); fp2_mul_fp(&c3, &c3, yP);
    fp12_mul_line(f, f, &c0, &c2, &c3);
    g2_dbl(T, T);
}
static void line_add(fp12 *f, g2p *T, const g2p *Q, const fp *xP, const fp *yP) {
    fp2 th, mu, c0, c2, c3, t;
    fp2_mul(&t, &Q->y, &T->z); fp2_sub(&th, &T->y, &t);
    fp2_mul(&t, &Q->x, &T->z); fp2_sub(&mu, &T->x, &t);
    fp2_mul(&c0, &th, &Q->x); fp2_mul(&t, &mu, &Q->y); fp2_sub(&c0, &c0, &t);
    fp2_neg(&c2, &th); fp2_mul_fp(&c2, &c2, xP);
    fp2_mul_fp(&c3, &mu, yP);
    fp12_mul_line(f, f, &c0, &c2, &c3);
    g2_add(T, T, Q);
}
/* prod_i f_{|x|,Q_i}(P_i), conjugated (x < 0). Pairs with an infinite member contribute 1 (mathematical
   convention; amcl's behaviour there is not pinned by any reference test -- see DESIGN.md). */
static void miller_loop_n(fp12 *f, const g2p *Qs, const g1p *Ps, int n) {
    g2p *T = malloc(sizeof(g2p) * (n ? n : 1)), *Q = malloc(sizeof(g2p) * (n ? n : 1));
    g1p *Pa = malloc(sizeof(g1p) * (n ? n : 1)); int m = 0;
    for (int i = 0; i < n; i++) {
        if (g2_is_inf(&Qs[i]) || g1_is_inf(&Ps[i])) continue;
        Q[m] = Qs[i]; g2_affine(&Q[m]); T[m] = Q[m]; Pa[m] = Ps[i]; g1_affine(&Pa[m]); m++;
    }
    *f = F12_ONE;
    for (int i = 62; i >= 0; i--) {
        fp12_sqr(f, f);
        for (int k = 0; k < m; k++) line_dbl(f, &T[k], &Pa[k].x, &Pa[k].y);
        if ((ORC_X_ABS >> i) & 1) for (int k = 0; k < m; k++) line_add(f, &T[k], &Q[k], &Pa[k].x, &Pa[k].y);
    }
    fp12_conj(f, f);
    free(T); free(Q); free(Pa);
}
/* f^(3 (p^12-1)/r): easy part, then 3*hard = (x-1)^2 (x+p) (x^2+p^2-1) + 3 (Hayashida-Hayasaka-Teruya).
   3 is coprime to r, so "== 1" is unchanged w.r.t. amcl's fexp (reference src/amcl_utils.rs:40-41). */
static void final_exp(fp12 *r, const fp12 *f) {
    fp12 t, u, a, b, c, m;
    fp12_conj(&t, f); fp12_inv(&u, f); fp12_mul(&t, &t, &u);            /* f^(p^6-1) */
    fp12_frob(&u, &t); fp12_frob(&u, &u); fp12_mul(&m, &u, &t);          /* ^(p^2+1) */
    /* a = m^((x-1)^2) */
    fp12_cyc_exp_x(&a, &m); fp12_conj(&u, &m); fp12_mul(&a, &a, &u);
    fp12_cyc_exp_x(&t, &a); fp12_conj(&u, &a); fp12_mul(&a, &t, &u);
    /* b = a^(x+p) */
    fp12_cyc_exp_x(&b, &a); fp12_frob(&u, &a); fp12_mul(&b, &b, &u);
    /* c = b^(x^2+p^2-1) */
    fp12_cyc_exp_x(&c, &b); fp12_cyc_exp_x(&c, &c); fp12_frob(&u, &b); fp12_frob(&u, &u); fp12_mul(&c, &c, &u);
    fp12_conj(&u, &b); fp12_mul(&c, &c, &u);
    /* * m^3 */
    fp12_cyc_sqr(&u, &m); fp12_mul(&u, &u, &m); fp12_mul(r, &c, &u);
}
static int pairing_product_is_one(const g2p *Qs, const g1p *Ps, int n) {
    fp12 f; miller_loop_n(&f, Qs, Ps, n); final_exp(&f, &f); return fp12_is_one(&f);
}

/* ------------------------------------------------------------------ init */
static pthread_once_t init_once = PTHREAD_ONCE_INIT;
static void do_init(void) {
    memset(&FP_ZERO, 0, sizeof FP_ZERO);
    memcpy(FP_R2.l, ORC_R2, 48);
    { fp one_raw = {{1, 0, 0, 0, 0, 0}}; fp_mul(&FP_ONE, &one_raw, &FP_R2); }
    F2_ZERO.c0 = FP_ZERO; F2_ZERO.c1 = FP_ZERO; F2_ONE.c0 = FP_ONE; F2_ONE.c1 = FP_ZERO;
    memset(&F12_ONE, 0, sizeof F12_ONE); F12_ONE.c0.c0 = F2_ONE;
    { uint64_t four[6] = {4, 0, 0, 0, 0, 0}, twelve[6] = {12, 0, 0, 0, 0, 0};
      fp_from_raw(&FP_B1, four); fp_from_raw(&B3_G1, twelve);
      FP2_B2.c0 = FP_B1; FP2_B2.c1 = FP_B1; B3_G2.c0 = B3_G1; B3_G2.c1 = B3_G1; }
    fp_from_raw(&G1_GEN.x, ORC_G1_X); fp_from_raw(&G1_GEN.y, ORC_G1_Y); G1_GEN.z = FP_ONE; g1_neg(&G1_NEG_GEN, &G1_GEN);
    fp2_from_raw(&G2_GEN.x, ORC_G2_X); fp2_from_raw(&G2_GEN.y, ORC_G2_Y); G2_GEN.z = F2_ONE;
    for (int k = 0; k < 6; k++) fp2_from_raw(&FROB_W[k], ORC_FROB_W[k]);
    fp2_from_raw(&PSI_CX, ORC_PSI_CX); fp2_from_raw(&PSI_CY, ORC_PSI_CY);
    for (int k = 0; k < 4; k++) { fp2_from_raw(&ISO_XNUM[k], ORC_ISO3_XNUM[k]); fp2_from_raw(&ISO_YNUM[k], ORC_ISO3_YNUM[k]); fp2_from_raw(&ISO_YDEN[k], ORC_ISO3_YDEN[k]); }
    for (int k = 0; k < 3; k++) fp2_from_raw(&ISO_XDEN[k], ORC_ISO3_XDEN[k]);
    fp2_from_raw(&SSWU_A, ORC_SSWU_A); fp2_from_raw(&SSWU_B, ORC_SSWU_B); fp2_from_raw(&SSWU_Z, ORC_SSWU_Z);
    { fp2 t; fp2_inv(&t, &SSWU_A); fp2_mul(&SSWU_MBA, &SSWU_B, &t); fp2_neg(&SSWU_MBA, &SSWU_MBA);
      fp2_mul(&t, &SSWU_Z, &SSWU_A); fp2_inv(&t, &t); fp2_mul(&SSWU_BZA, &SSWU_B, &t); }
}
void orc_init(void) { pthread_once(&init_once, do_init); }

/* ------------------------------------------------------------------ exported: field-level probes */
void orc_fp_mul(const uint8_t a[48], const uint8_t b[48], uint8_t out[48]) {
    orc_init(); fp x, y; fp_from_be(&x, a); fp_from_be(&y, b); fp_mul(&x, &x, &y); fp_to_be(out, &x);
}
void orc_fp_inv(const uint8_t a[48], uint8_t out[48]) { orc_init(); fp x; fp_from_be(&x, a); fp_inv(&x, &x); fp_to_be(out, &x); }
int orc_fp_sqrt(const uint8_t a[48], uint8_t out[48]) {
    orc_init(); fp x, s; fp_from_be(&x, a); if (!fp_sqrt(&s, &x)) return 0; fp_to_be(out, &s); return 1;
}
void orc_op_counts(uint64_t *mul, uint64_t *sqr, int reset) { *mul = cnt_mul; *sqr = cnt_sqr; if (reset) cnt_mul = cnt_sqr = 0; }

/* ------------------------------------------------------------------ exported: codec (decoded form = uncompressed bytes) */
int orc_g1_from_compressed(const uint8_t *in, size_t len, uint8_t out[96]) {      /* PublicKey::from_bytes_unchecked, src/keys.rs:150-155 */
    orc_init(); if (len != 48) return ORC_ERR_G1_SIZE;
    g1p p; int e = g1_from_comp(&p, in); if (e) return e; g1_to_unc(out, &p); return ORC_OK;
}
int orc_g1_from_uncompressed(const uint8_t *in, size_t len, uint8_t out[96]) {    /* PublicKey::from_uncompressed_bytes, src/keys.rs:170-175 */
    orc_init(); if (len != 96) return ORC_ERR_G1_SIZE;
    g1p p; int e = g1_from_unc(&p, in); if (e) return e; g1_to_unc(out, &p); return ORC_OK;
}
int orc_g1_key_validate(const uint8_t pk[96]) {                                    /* src/keys.rs:181-186 */
    orc_init(); g1p p; if (g1_from_unc(&p, pk)) return 0;
    if (g1_is_inf(&p) || !g1_in_subgroup(&p)) return 0;
    return 1;
}
int orc_pk_from_bytes(const uint8_t *in, size_t len, uint8_t out[96]) {            /* PublicKey::from_bytes, src/keys.rs:140-147 */
    int e = orc_g1_from_compressed(in, len, out); if (e) return e;
    return orc_g1_key_validate(out) ? ORC_OK : ORC_ERR_POINT;
}
int orc_g1_compress(const uint8_t pk[96], uint8_t out[48]) { orc_init(); g1p p; int e = g1_from_unc(&p, pk); if (e) return e; g1_to_comp(out, &p); return ORC_OK; }
int orc_g2_from_compressed(const uint8_t *in, size_t len, uint8_t out[192]) {     /* Signature::from_bytes, src/signature.rs:43-46 */
    orc_init(); if (len != 96) return ORC_ERR_G2_SIZE;
    g2p p; int e = g2_from_comp(&p, in); if (e) return e; g2_to_unc(out, &p); return ORC_OK;
}
int orc_g2_compress(const uint8_t sig[192], uint8_t out[96]) { orc_init(); g2p p; int e = g2_from_unc(&p, sig); if (e) return e; g2_to_comp(out, &p); return ORC_OK; }
int orc_g2_subgroup_check(const uint8_t sig[192]) { orc_init(); g2p p; if (g2_from_unc(&p, sig)) return 0; return g2_in_subgroup(&p); }

/* ------------------------------------------------------------------ exported: group ops */
static void scalar_from_be32(uint64_t k[4], const uint8_t b[32]) {
    for (int i = 0; i < 4; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v = (v << 8) | b[(3 - i) * 8 + j]; k[i] = v; }
}
int orc_g1_add(const uint8_t a[96], const uint8_t b[96], uint8_t out[96]) {        /* AggregatePublicKey::add, src/aggregates.rs:68-70 */
    orc_init(); g1p p, q; int e; if ((e = g1_from_unc(&p, a)) || (e = g1_from_unc(&q, b))) return e;
    g1_add(&p, &p, &q); g1_to_unc(out, &p); return ORC_OK;
}
int orc_g2_add(const uint8_t a[192], const uint8_t b[192], uint8_t out[192]) {     /* AggregateSignature::add, src/aggregates.rs:114-116 */
    orc_init(); g2p p, q; int e; if ((e = g2_from_unc(&p, a)) || (e = g2_from_unc(&q, b))) return e;
    g2_add(&p, &p, &q); g2_to_unc(out, &p); return ORC_OK;
}
int orc_g1_mul(const uint8_t a[96], const uint8_t k32[32], uint8_t out[96]) {
    orc_init(); g1p p; int e; if ((e = g1_from_unc(&p, a))) return e; uint64_t k[4]; scalar_from_be32(k, k32);
    g1_mul(&p, &p, k, 4); g1_to_unc(out, &p); return ORC_OK;
}
int orc_g2_mul(const uint8_t a[192], const uint8_t k32[32], uint8_t out[192]) {
    orc_init(); g2p p; int e; if ((e = g2_from_unc(&p, a))) return e; uint64_t k[4]; scalar_from_be32(k, k32);
    g2_mul(&p, &p, k, 4); g2_to_unc(out, &p); return ORC_OK;
}
/* AggregatePublicKey::aggregate, src/aggregates.rs:29-39 */
int orc_aggregate_pks(const uint8_t *pks, size_t n, uint8_t out[96]) {
    orc_init(); if (n == 0) return ORC_ERR_EMPTY;
    g1p acc, p; g1_set_inf(&acc);
    for (size_t i = 0; i < n; i++) { int e = g1_from_unc(&p, pks + 96 * i); if (e) return e; g1_add(&acc, &acc, &p); }
    g1_to_unc(out, &acc); return ORC_OK;
}
void orc_sk_to_pk(const uint8_t sk[32], uint8_t out[96]) {                          /* PublicKey::from_secret_key, src/keys.rs:124-137 */
    orc_init(); uint64_t k[4]; scalar_from_be32(k, sk); g1p p; g1_mul(&p, &G1_GEN, k, 4); g1_to_unc(out, &p);
}
void orc_hash_to_g2(const uint8_t *msg, size_t mlen, const uint8_t *dst, size_t dlen, uint8_t out[192]) {
    orc_init(); g2p h; if (!dst) { dst = DST_POP; dlen = DST_POP_LEN; } hash_to_g2(&h, msg, mlen, dst, dlen); g2_to_unc(out, &h);
}
void orc_sign(const uint8_t *msg, size_t mlen, const uint8_t sk[32], uint8_t out[192]) {   /* Signature::new, src/signature.rs:17-21 */
    orc_init(); g2p h; hash_to_g2(&h, msg, mlen, DST_POP, DST_POP_LEN); uint64_t k[4]; scalar_from_be32(k, sk);
    g2_mul(&h, &h, k, 4); g2_to_unc(out, &h);
}

/* ------------------------------------------------------------------ exported: verification family */
static int core_pair_check(const g2p *sig, const g2p *h, const g1p *pk) {
    g2p Q[2] = {*sig, *h}; g1p Pp[2] = {G1_NEG_GEN, *pk};
    return pairing_product_is_one(Q, Pp, 2);       /* ate2_evaluation, src/amcl_utils.rs:38-42 */
}
int orc_verify(const uint8_t sig[192], const uint8_t *msg, size_t mlen, const uint8_t pk[96]) {   /* src/signature.rs:27-40 */
    orc_init(); g2p s, h; g1p p;
    if (g2_from_unc(&s, sig) || g1_from_unc(&p, pk)) return 0;
    if (!g2_in_subgroup(&s)) return 0;
    hash_to_g2(&h, msg, mlen, DST_POP, DST_POP_LEN);
    return core_pair_check(&s, &h, &p);
}
int orc_fast_aggregate_verify_pre_aggregated(const uint8_t sig[192], const uint8_t *msg, size_t mlen, const uint8_t apk[96]) {   /* src/aggregates.rs:223-253 */
    orc_init(); g2p s, h; g1p p;
    if (g2_from_unc(&s, sig) || g1_from_unc(&p, apk)) return 0;
    if (!g2_in_subgroup(&s)) return 0;
    if (g1_is_inf(&p)) return 0;
    hash_to_g2(&h, msg, mlen, DST_POP, DST_POP_LEN);
    return core_pair_check(&s, &h, &p);
}
static int fav_points(const g2p *s, const uint8_t *msg, size_t mlen, const g1p *pks, size_t n) {   /* src/aggregates.rs:177-215 */
    if (n == 0) return 0;
    if (!g2_in_subgroup(s)) return 0;
    g1p acc; g1_set_inf(&acc);
    for (size_t i = 0; i < n; i++) g1_add(&acc, &acc, &pks[i]);
    if (g1_is_inf(&acc)) return 0;
    g2p h; hash_to_g2(&h, msg, mlen, DST_POP, DST_POP_LEN);
    return core_pair_check(s, &h, &acc);
}
int orc_fast_aggregate_verify(const uint8_t sig[192], const uint8_t *msg, size_t mlen, const uint8_t *pks, size_t n) {
    orc_init(); g2p s; if (g2_from_unc(&s, sig)) return 0;
    g1p *P_ = malloc(sizeof(g1p) * (n ? n : 1)); int ok = 1;
    for (size_t i = 0; i < n && ok; i++) if (g1_from_unc(&P_[i], pks + 96 * i)) ok = 0;
    if (ok) ok = fav_points(&s, msg, mlen, P_, n);
    free(P_); return ok;
}
/* src/aggregates.rs:130-170; msgs concatenated, lens[i] each */
int orc_aggregate_verify(const uint8_t sig[192], const uint8_t *msgs, const size_t *lens, size_t n_msgs, const uint8_t *pks, size_t n_pks) {
    orc_init(); if (n_msgs != n_pks || n_pks == 0) return 0;
    g2p s; if (g2_from_unc(&s, sig)) return 0;
    if (!g2_in_subgroup(&s)) return 0;
    g2p *Q = malloc(sizeof(g2p) * (n_pks + 1)); g1p *Pp = malloc(sizeof(g1p) * (n_pks + 1)); int ok = 1; size_t off = 0;
    for (size_t i = 0; i < n_pks && ok; i++) {
        if (g1_from_unc(&Pp[i], pks + 96 * i)) { ok = 0; break; }
        hash_to_g2(&Q[i], msgs + off, lens[i], DST_POP, DST_POP_LEN); off += lens[i];
    }
    if (ok) { Q[n_pks] = s; Pp[n_pks] = G1_NEG_GEN; ok = pairing_product_is_one(Q, Pp, (int)n_pks + 1); }
    free(Q); free(Pp); return ok;
}
/* src/aggregates.rs:261-316; rands[i] = the nonzero 63-bit blinding scalars the caller's RNG produced */
int orc_verify_multiple(const uint8_t *sigs, const uint8_t *apks, const uint8_t *msgs, const size_t *lens, const uint64_t *rands, size_t n) {
    orc_init();
    g2p *Q = malloc(sizeof(g2p) * (n + 1)); g1p *Pp = malloc(sizeof(g1p) * (n + 1)); g2p acc, s, t; g2_set_inf(&acc);
    int ok = 1; size_t off = 0;
    for (size_t i = 0; i < n && ok; i++) {
        g1p a;
        if (g2_from_unc(&s, sigs + 192 * i) || g1_from_unc(&a, apks + 96 * i)) { ok = 0; break; }
        if (!g2_in_subgroup(&s)) { ok = 0; break; }
        hash_to_g2(&Q[i], msgs + off, lens[i], DST_POP, DST_POP_LEN); off += lens[i];
        g1_mul(&Pp[i], &a, &rands[i], 1);
        g2_mul(&t, &s, &rands[i], 1); g2_add(&acc, &acc, &t);
    }
    if (ok) { Q[n] = acc; Pp[n] = G1_NEG_GEN; ok = pairing_product_is_one(Q, Pp, (int)n + 1); }
    free(Q); free(Pp); return ok;
}

/* ------------------------------------------------------------------ exported: batch (serialized wire formats) */
typedef struct {
    const uint8_t *sigs, *msgs, *pks; size_t n, k, msg_len; int pk_fmt; uint8_t *out; size_t lo, hi;
} fav_job;
/* one tuple, from wire bytes: sig 96 B compressed, pks 48 B compressed or 96 B uncompressed */
static int fav_wire(const uint8_t *sig, const uint8_t *msg, size_t mlen, const uint8_t *pks, size_t k, int fmt) {
    g2p s; if (g2_from_comp(&s, sig)) return 0;
    g1p *P_ = malloc(sizeof(g1p) * (k ? k : 1)); int ok = 1;
    for (size_t i = 0; i < k && ok; i++) {
        int e = fmt == ORC_PK_COMPRESSED ? g1_from_comp(&P_[i], pks + 48 * i) : g1_from_unc(&P_[i], pks + 96 * i);
        if (e) ok = 0;
    }
    if (ok) ok = fav_points(&s, msg, mlen, P_, k);
    free(P_); return ok;
}
static void *fav_worker(void *arg) {
    fav_job *j = arg; size_t pkb = j->pk_fmt == ORC_PK_COMPRESSED ? 48 : 96;
    for (size_t i = j->lo; i < j->hi; i++)
        j->out[i] = (uint8_t)fav_wire(j->sigs + 96 * i, j->msgs + j->msg_len * i, j->msg_len, j->pks + pkb * j->k * i, j->k, j->pk_fmt);
    return NULL;
}
/* out[i] = 0/1 per tuple */
void orc_batch_fast_aggregate_verify(const uint8_t *sigs, const uint8_t *msgs, size_t msg_len, const uint8_t *pks, int pk_fmt,
                                     size_t n, size_t k, uint8_t *out, int nthreads) {
    orc_init(); if (nthreads < 1) nthreads = 1; if ((size_t)nthreads > n && n) nthreads = (int)n;
    pthread_t th[256]; fav_job jobs[256]; if (nthreads > 256) nthreads = 256;
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (fav_job){sigs, msgs, pks, n, k, msg_len, pk_fmt, out, n * t / nthreads, n * (t + 1) / nthreads};
        if (nthreads == 1) fav_worker(&jobs[t]); else pthread_create(&th[t], NULL, fav_worker, &jobs[t]);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}
typedef struct { const uint8_t *sks, *msgs; size_t msg_len; uint8_t *out; size_t lo, hi; int what; } gen_job;
static void *gen_worker(void *arg) {
    gen_job *j = arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        uint64_t k[4]; scalar_from_be32(k, j->sks + 32 * i);
        if (j->what == 0) { g2p h; hash_to_g2(&h, j->msgs + j->msg_len * i, j->msg_len, DST_POP, DST_POP_LEN); g2_mul(&h, &h, k, 4); g2_to_comp(j->out + 96 * i, &h); }
        else if (j->what == 1) { g1p p; g1_mul(&p, &G1_GEN, k, 4); g1_to_comp(j->out + 48 * i, &p); }
        else { g1p p; g1_mul(&p, &G1_GEN, k, 4); g1_to_unc(j->out + 96 * i, &p); }
    }
    return NULL;
}
static void run_gen(gen_job proto, size_t n, int nthreads) {
    orc_init(); if (nthreads < 1) nthreads = 1; if ((size_t)nthreads > n && n) nthreads = (int)n; if (nthreads > 256) nthreads = 256;
    pthread_t th[256]; gen_job jobs[256];
    for (int t = 0; t < nthreads; t++) { jobs[t] = proto; jobs[t].lo = n * t / nthreads; jobs[t].hi = n * (t + 1) / nthreads;
        if (nthreads == 1) gen_worker(&jobs[t]); else pthread_create(&th[t], NULL, gen_worker, &jobs[t]); }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}
/* sigs_out[i] = compress([sk_i] H(msg_i)) */
void orc_batch_sign(const uint8_t *sks, const uint8_t *msgs, size_t msg_len, size_t n, uint8_t *sigs_out, int nthreads) {
    run_gen((gen_job){sks, msgs, msg_len, sigs_out, 0, 0, 0}, n, nthreads);
}
void orc_batch_sk_to_pk(const uint8_t *sks, size_t n, int pk_fmt, uint8_t *pks_out, int nthreads) {
    run_gen((gen_job){sks, NULL, 0, pks_out, 0, 0, pk_fmt == ORC_PK_COMPRESSED ? 1 : 2}, n, nthreads);
}
/* Signature::verify over wire bytes (sig 96 B, pk 48 B compressed, msg msg_len), out[i] = 0/1.
   A pk that fails to decode yields 0 (the reference caller could not have built a PublicKey from it). */
typedef struct { const uint8_t *sigs, *msgs, *pks; size_t msg_len; uint8_t *out; size_t lo, hi; } ver_job;
static void *ver_worker(void *arg) {
    ver_job *j = arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        g2p s, h; g1p p; int ok = !g2_from_comp(&s, j->sigs + 96 * i) && !g1_from_comp(&p, j->pks + 48 * i);
        if (ok) ok = g2_in_subgroup(&s);
        if (ok) { hash_to_g2(&h, j->msgs + j->msg_len * i, j->msg_len, DST_POP, DST_POP_LEN); ok = core_pair_check(&s, &h, &p); }
        j->out[i] = (uint8_t)ok;
    }
    return NULL;
}
void orc_batch_verify(const uint8_t *sigs, const uint8_t *msgs, size_t msg_len, const uint8_t *pks, size_t n, uint8_t *out, int nthreads) {
    orc_init(); if (nthreads < 1) nthreads = 1; if ((size_t)nthreads > n && n) nthreads = (int)n; if (nthreads > 256) nthreads = 256;
    pthread_t th[256]; ver_job jobs[256];
    for (int t = 0; t < nthreads; t++) { jobs[t] = (ver_job){sigs, msgs, pks, msg_len, out, n * t / nthreads, n * (t + 1) / nthreads};
        if (nthreads == 1) ver_worker(&jobs[t]); else pthread_create(&th[t], NULL, ver_worker, &jobs[t]); }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}
/* hash_to_curve_g2 over a batch of fixed-length messages, compressed output */
void orc_batch_hash_to_g2(const uint8_t *msgs, size_t msg_len, size_t n, uint8_t *out96) {
    orc_init(); for (size_t i = 0; i < n; i++) { g2p h; hash_to_g2(&h, msgs + msg_len * i, msg_len, DST_POP, DST_POP_LEN); g2_to_comp(out96 + 96 * i, &h); }
}
