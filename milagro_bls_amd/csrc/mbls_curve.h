// mbls_curve.h -- G1 / G2 group law, endomorphisms, subgroup checks, ZCash point codec. One point per lane.
// Replaces amcl's ECP / ECP2 (GroupG1 / GroupG2, reference src/amcl_utils.rs:23-24), g1mul/g2mul
// (:20), subgroup_check_g1/g2 (:15-16) and (de)serialize_g1/g2 (:13-17).
// Jacobian coordinates (x = X/Z^2, y = Y/Z^3, infinity <=> Z = 0); `add` handles infinity, doubling and
// inverse points like amcl's complete `add` (reference src/aggregates.rs:34-37 starts from infinity and
// may add equal points).
#pragma once
#include "mbls_tower.h"

// per-item status bits carried between kernels instead of divergent early exits
#define MBLS_ST_BAD_SIG_ENCODING 0x01u   // Signature::from_bytes would have failed (reference src/signature.rs:43-46)
#define MBLS_ST_SIG_NOT_IN_G2    0x02u   // subgroup_check_g2 failed (reference src/aggregates.rs:184-186)
#define MBLS_ST_BAD_PK_ENCODING  0x04u   // PublicKey::from_* would have failed (reference src/keys.rs:140-175)
#define MBLS_ST_APK_INFINITY     0x08u   // aggregate key is infinity (reference src/aggregates.rs:196-198)
#define MBLS_ST_NO_KEYS          0x10u   // empty key list (reference src/aggregates.rs:179-181)
#define MBLS_ST_PK_INFINITY      0x20u   // informational: a decoded key is the point at infinity
#define MBLS_ST_PAIRING_FAILED   0x40u   // pairing product != 1
#define MBLS_ST_BAD_SCALAR       0x80u   // verify_multiple: a blinding scalar is zero (reference src/aggregates.rs:280-287 never draws one)
#define MBLS_ST_BAD_MSG_RANGE    0x100u  // the item's entry of a message offset table runs backwards (no &[u8] a reference caller could pass)

struct g1j { fp x, y, z; };
struct g2j { fp2 x, y, z; };

// ------------------------------------------------------------------------------------------------ G1
MBLS_FN void g1_set_inf(g1j* p) { p->x = fp_zero(); p->y = fp_one(); p->z = fp_zero(); }
MBLS_FN bool g1_is_inf(const g1j* p) { return fp_is_zero(p->z); }
MBLS_NOINLINE void g1_dbl(g1j* r, const g1j* p) {
    fp A = fp_sqr(p->x), B = fp_sqr(p->y), C = fp_sqr(B);
    fp D = fp_dbl(fp_sub(fp_sub(fp_sqr(fp_add(p->x, B)), A), C));
    fp E = fp_add(fp_dbl(A), A), F = fp_sqr(E);
    fp z3 = fp_dbl(fp_mul(p->y, p->z));
    fp x3 = fp_sub(F, fp_dbl(D));
    fp c8 = fp_dbl(fp_dbl(fp_dbl(C)));
    r->y = fp_sub(fp_mul(E, fp_sub(D, x3)), c8); r->x = x3; r->z = z3;
}
// r = p + (x2, y2) affine; inf2 marks the affine operand as the point at infinity
MBLS_FN void g1_madd_inl(g1j* r, const g1j* p, fp x2, fp y2, bool inf2) {
    bool inf1 = g1_is_inf(p);
    // the 11 multiplications as five independent pairs (fp_mul_pair) and one single
    fp z1z1, t, u2, s2, hh, rr2, j, v, m0, m1;
    fp_mul_pair(&z1z1, &t, p->z, p->z, y2, p->z);
    fp_mul_pair(&u2, &s2, x2, z1z1, t, z1z1);
    fp h = fp_sub(u2, p->x), rr = fp_dbl(fp_sub(s2, p->y));
    bool h0 = fp_is_zero(h), r0 = fp_is_zero(rr);
    fp_mul_pair(&hh, &rr2, h, h, rr, rr);
    fp i4 = fp_dbl(fp_dbl(hh));
    fp_mul_pair(&j, &v, h, i4, p->x, i4);
    fp x3 = fp_sub(fp_sub(rr2, j), fp_dbl(v));
    fp_mul_pair(&m0, &m1, rr, fp_sub(v, x3), p->y, j);
    fp y3 = fp_sub(m0, fp_dbl(m1));
    fp z3 = fp_sub(fp_sub(fp_sqr(fp_add(p->z, h)), z1z1), hh);
    g1j out; out.x = x3; out.y = y3; out.z = z3;
    if (h0 & !inf1 & !inf2) {                 // same x: doubling or inverse points (rare, divergent)
        if (r0) { g1j q; q.x = x2; q.y = y2; q.z = fp_one(); g1_dbl(&out, &q); }
        else g1_set_inf(&out);
    }
    // p infinite -> the affine operand; affine operand infinite -> p
    out.x = fp_select(inf1, x2, out.x); out.y = fp_select(inf1, y2, out.y); out.z = fp_select(inf1, fp_one(), out.z);
    out.x = fp_select(inf2, p->x, out.x); out.y = fp_select(inf2, p->y, out.y); out.z = fp_select(inf2, p->z, out.z);
    *r = out;
}
MBLS_NOINLINE void g1_madd(g1j* r, const g1j* p, fp x2, fp y2, bool inf2) { g1_madd_inl(r, p, x2, y2, inf2); }
MBLS_NOINLINE void g1_add(g1j* r, const g1j* p, const g1j* q) {
    bool inf1 = g1_is_inf(p), inf2 = g1_is_inf(q);
    fp z1z1 = fp_sqr(p->z), z2z2 = fp_sqr(q->z);
    fp u1 = fp_mul(p->x, z2z2), u2 = fp_mul(q->x, z1z1);
    fp s1 = fp_mul(fp_mul(p->y, q->z), z2z2), s2 = fp_mul(fp_mul(q->y, p->z), z1z1);
    fp h = fp_sub(u2, u1), rr = fp_dbl(fp_sub(s2, s1));
    bool h0 = fp_is_zero(h), r0 = fp_is_zero(rr);
    fp i4 = fp_sqr(fp_dbl(h)), j = fp_mul(h, i4), v = fp_mul(u1, i4);
    fp x3 = fp_sub(fp_sub(fp_sqr(rr), j), fp_dbl(v));
    fp y3 = fp_sub(fp_mul(rr, fp_sub(v, x3)), fp_dbl(fp_mul(s1, j)));
    fp z3 = fp_mul(fp_sub(fp_sub(fp_sqr(fp_add(p->z, q->z)), z1z1), z2z2), h);
    g1j out; out.x = x3; out.y = y3; out.z = z3;
    if (h0 & !inf1 & !inf2) { if (r0) g1_dbl(&out, p); else g1_set_inf(&out); }
    out.x = fp_select(inf1, q->x, out.x); out.y = fp_select(inf1, q->y, out.y); out.z = fp_select(inf1, q->z, out.z);
    out.x = fp_select(inf2, p->x, out.x); out.y = fp_select(inf2, p->y, out.y); out.z = fp_select(inf2, p->z, out.z);
    *r = out;
}
MBLS_FN void g1_neg(g1j* r, const g1j* p) { r->x = p->x; r->y = fp_neg(p->y); r->z = p->z; }
// [k]P, k = nbits-bit scalar given as little-endian 32-bit words (per-lane value)
MBLS_NOINLINE void g1_mul(g1j* r, const g1j* p, const uint32_t* k, int nbits) {
    g1j acc; g1_set_inf(&acc);
    for (int i = nbits - 1; i >= 0; i--) {
        g1_dbl(&acc, &acc);
        g1j t; g1_add(&t, &acc, p);
        bool bit = (k[i >> 5] >> (i & 31)) & 1u;
        acc.x = fp_select(bit, t.x, acc.x); acc.y = fp_select(bit, t.y, acc.y); acc.z = fp_select(bit, t.z, acc.z);
    }
    *r = acc;
}
MBLS_FN bool g1_on_curve(fp x, fp y) {       // y^2 = x^3 + 4
    fp four = fp_dbl(fp_dbl(fp_one()));
    return fp_eq(fp_sqr(y), fp_add(fp_mul(fp_sqr(x), x), four));
}
MBLS_FN void g1_to_affine(fp* x, fp* y, bool* inf, const g1j* p) {
    *inf = g1_is_inf(p);
    fp zi = fp_inv(p->z), zi2 = fp_sqr(zi);
    *x = fp_mul(p->x, zi2); *y = fp_mul(fp_mul(p->y, zi2), zi);
}
// [|x|]P for the curve parameter (64 bits, weight 6): the bits are compile-time constants, so no selects
MBLS_NOINLINE void g1_mul_x_abs(g1j* r, const g1j* p) {
    g1j acc = *p;
    for (int i = 62; i >= 0; i--) {
        g1_dbl(&acc, &acc);
        if ((MBLS_X_ABS >> i) & 1ull) { g1j t; g1_add(&t, &acc, p); acc = t; }
    }
    *r = acc;
}
// subgroup_check_g1: [r]P == O (reference src/keys.rs:182), decided as phi(P) == [-x^2]P with phi(x, y) = (beta x, y):
// phi^2 + phi + 1 = 0 on the curve, so phi(P) = [-x^2]P gives [x^4 - x^2 + 1]P = [r]P = O, and on the order-r subgroup phi IS
// multiplication by -x^2 (oracle/gen_constants.py picks that beta) -- the same set of points as the 255-bit multiplication, for
// 126 doublings + 10 additions. Infinity passes.
MBLS_NOINLINE bool g1_in_subgroup(const g1j* p) {
    if (g1_is_inf(p)) return true;
    g1j t, q;
    g1_mul_x_abs(&t, p); g1_mul_x_abs(&q, &t);                       // q = [x^2]P; must equal -phi(P) = (beta x, -y)
    fp zp2 = fp_sqr(p->z), zq2 = fp_sqr(q.z);
    fp bx = fp_mul(fp_mul(fp_load_const(MBLS_G1_BETA), p->x), zq2);
    fp ny = fp_mul(fp_mul(fp_neg(p->y), zq2), q.z);
    bool ex = fp_eq(fp_mul(q.x, zp2), bx);
    bool ey = fp_eq(fp_mul(fp_mul(q.y, zp2), p->z), ny);
    return !g1_is_inf(&q) && ex && ey;
}

// ZCash codec. Decoders return 0 or a status/err and produce affine Montgomery coordinates.
#define MBLS_DEC_OK 0
#define MBLS_DEC_SIZE 1       // flag/length mismatch -> AmclError::InvalidG1Size / InvalidG2Size
#define MBLS_DEC_POINT 3      // AmclError::InvalidPoint
MBLS_FN bool mbls_all_zero(const uint8_t* b, int n) { uint32_t t = 0; for (int i = 0; i < n; i++) t |= b[i]; return t == 0; }

// PublicKey::from_bytes_unchecked (reference src/keys.rs:150-155): 48 compressed bytes
template <bool W4 = false>
MBLS_NOINLINE int g1_decode_compressed(fp* x, fp* y, bool* inf, const uint8_t* b) {
    uint8_t b0 = b[0];
    *inf = false; *x = fp_zero(); *y = fp_zero();
    if (!(b0 & 0x80)) return MBLS_DEC_SIZE;
    if (b0 & 0x40) { if ((b0 & 0x3F) || !mbls_all_zero(b + 1, 47)) return MBLS_DEC_POINT; *inf = true; return MBLS_DEC_OK; }
    fp raw = fp_raw_from_be(b); raw[11] &= 0x1FFFFFFFu;
    if (fp_raw_geq_p(raw)) return MBLS_DEC_POINT;
    fp xm = fp_to_mont(raw);
    fp four = fp_dbl(fp_dbl(fp_one()));
    fp ym;
    if (!fp_sqrt<W4>(&ym, fp_add(fp_mul(fp_sqr(xm), xm), four))) return MBLS_DEC_POINT;
    bool want = (b0 & 0x20) != 0;
    ym = fp_select(fp_lex_largest(ym) != want, fp_neg(ym), ym);
    *x = xm; *y = ym; return MBLS_DEC_OK;
}
// PublicKey::from_uncompressed_bytes (reference src/keys.rs:170-175): 96 bytes x || y, on-curve check.
// The 96 bytes travel as 24 dwords in memory order (wx = bytes 0..47, wy = bytes 48..95): a 16-byte aligned key is fetched
// with 6 dwordx4 loads instead of 96 byte loads, and the caller can issue the loads of the next key before this one is used.
MBLS_FN void g1_load_words96(fp* wx, fp* wy, const uint8_t* b, bool aligned16) {
    fp x, y;
    if (aligned16) {
        const uint32_t* q = (const uint32_t*)__builtin_assume_aligned(b, 16);
#pragma unroll
        for (int i = 0; i < 12; i++) { x[i] = q[i]; y[i] = q[12 + i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const uint8_t* q = b + 4 * i;
            x[i] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
            y[i] = (uint32_t)q[48] | ((uint32_t)q[49] << 8) | ((uint32_t)q[50] << 16) | ((uint32_t)q[51] << 24);
        }
    }
    *wx = x; *wy = y;
}
MBLS_FN uint32_t mbls_bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }
MBLS_FN fp fp_raw_from_be_words(fp w) {          // big-endian byte string held as memory-order dwords -> little-endian limbs
    fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r[i] = mbls_bswap32(w[11 - i]);
    return r;
}
MBLS_FN int g1_decode_uncompressed_w(fp* x, fp* y, bool* inf, fp wx, fp wy) {
    uint32_t b0 = wx[0] & 0xFFu;
    *inf = false; *x = fp_zero(); *y = fp_zero();
    if (b0 & 0x80) return MBLS_DEC_SIZE;
    if (b0 & 0x40) {
        uint32_t rest = wx[0] >> 8;
#pragma unroll
        for (int i = 1; i < 12; i++) rest |= wx[i];
#pragma unroll
        for (int i = 0; i < 12; i++) rest |= wy[i];
        if ((b0 & 0x3F) || rest) return MBLS_DEC_POINT;
        *inf = true; return MBLS_DEC_OK;
    }
    if (b0 & 0x20) return MBLS_DEC_POINT;
    fp rx = fp_raw_from_be_words(wx), ry = fp_raw_from_be_words(wy);
    if (fp_raw_geq_p(rx) | fp_raw_geq_p(ry)) return MBLS_DEC_POINT;
    fp xm, ym, xx, yy;
    fp_mul_pair(&xm, &ym, rx, fp_load_const(MBLS_R2), ry, fp_load_const(MBLS_R2));     // to Montgomery form
    fp_mul_pair(&xx, &yy, xm, xm, ym, ym);
    fp four = fp_dbl(fp_dbl(fp_one()));
    if (!fp_eq(yy, fp_add(fp_mul(xx, xm), four))) return MBLS_DEC_POINT;                 // y^2 = x^3 + 4
    *x = xm; *y = ym; return MBLS_DEC_OK;
}
MBLS_NOINLINE int g1_decode_uncompressed(fp* x, fp* y, bool* inf, const uint8_t* b) {
    fp wx, wy; g1_load_words96(&wx, &wy, b, false);
    return g1_decode_uncompressed_w(x, y, inf, wx, wy);
}
MBLS_FN void g1_encode_compressed(uint8_t* b, fp x, fp y, bool inf) {          // reference src/amcl_utils.rs:46-48
    if (inf) { for (int i = 0; i < 48; i++) b[i] = 0; b[0] = 0xC0; return; }
    fp_raw_to_be(b, fp_from_mont(x)); b[0] |= 0x80; if (fp_lex_largest(y)) b[0] |= 0x20;
}
MBLS_FN void g1_encode_uncompressed(uint8_t* b, fp x, fp y, bool inf) {        // reference src/keys.rs:163-165
    if (inf) { for (int i = 0; i < 96; i++) b[i] = 0; b[0] = 0x40; return; }
    fp_raw_to_be(b, fp_from_mont(x)); fp_raw_to_be(b + 48, fp_from_mont(y));
}

// ------------------------------------------------------------------------------------------------ G2
MBLS_FN void g2_set_inf(g2j* p) { p->x = fp2_zero(); p->y = fp2_one(); p->z = fp2_zero(); }
MBLS_FN bool g2_is_inf(const g2j* p) { return fp2_is_zero(p->z); }
MBLS_FN void g2_select(g2j* r, bool c, const g2j* a, const g2j* b) {
    r->x = fp2_select(c, a->x, b->x); r->y = fp2_select(c, a->y, b->y); r->z = fp2_select(c, a->z, b->z);
}
MBLS_NOINLINE void g2_dbl(g2j* r, const g2j* p) {
    fp2 A = fp2_sqr(p->x), B = fp2_sqr(p->y), C = fp2_sqr(B);
    fp2 D = fp2_dbl(fp2_sub(fp2_sub(fp2_sqr(fp2_add(p->x, B)), A), C));
    fp2 E = fp2_mul3(A), F = fp2_sqr(E);
    fp2 z3 = fp2_dbl(fp2_mul(p->y, p->z));
    fp2 x3 = fp2_sub(F, fp2_dbl(D));
    r->y = fp2_sub(fp2_mul(E, fp2_sub(D, x3)), fp2_mul8(C)); r->x = x3; r->z = z3;
}
MBLS_NOINLINE void g2_add(g2j* r, const g2j* p, const g2j* q) {
    bool inf1 = g2_is_inf(p), inf2 = g2_is_inf(q);
    fp2 z1z1 = fp2_sqr(p->z), z2z2 = fp2_sqr(q->z);
    fp2 u1 = fp2_mul(p->x, z2z2), u2 = fp2_mul(q->x, z1z1);
    fp2 s1 = fp2_mul(fp2_mul(p->y, q->z), z2z2), s2 = fp2_mul(fp2_mul(q->y, p->z), z1z1);
    fp2 h = fp2_sub(u2, u1), rr = fp2_dbl(fp2_sub(s2, s1));
    bool h0 = fp2_is_zero(h), r0 = fp2_is_zero(rr);
    fp2 i4 = fp2_sqr(fp2_dbl(h)), j = fp2_mul(h, i4), v = fp2_mul(u1, i4);
    g2j out;
    out.x = fp2_sub(fp2_sub(fp2_sqr(rr), j), fp2_dbl(v));
    out.y = fp2_sub(fp2_mul(rr, fp2_sub(v, out.x)), fp2_dbl(fp2_mul(s1, j)));
    out.z = fp2_mul(fp2_sub(fp2_sub(fp2_sqr(fp2_add(p->z, q->z)), z1z1), z2z2), h);
    if (h0 & !inf1 & !inf2) { if (r0) g2_dbl(&out, p); else g2_set_inf(&out); }
    g2_select(&out, inf1, q, &out);
    g2_select(&out, inf2, p, &out);
    *r = out;
}
MBLS_FN void g2_neg(g2j* r, const g2j* p) { r->x = p->x; r->y = fp2_neg(p->y); r->z = p->z; }
MBLS_NOINLINE void g2_psi(g2j* r, const g2j* p) {
    r->x = fp2_mul(fp2_conj(p->x), fp2_load_const(MBLS_PSI_CX));
    r->y = fp2_mul(fp2_conj(p->y), fp2_load_const(MBLS_PSI_CY));
    r->z = fp2_conj(p->z);
}
MBLS_FN void g2_psi2(g2j* r, const g2j* p) {
    r->x = fp2_mul_fp(p->x, fp_load_const(MBLS_PSI2_CX)); r->y = fp2_neg(p->y); r->z = p->z;
}
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
// n >= 1 doublings in place as one generated straight-line routine (tools/gen_tower_d.py, prog_g2_dbl_d): the run of doublings on
// 14 signed 28-bit digits per coordinate; the point is cut into digits on entry and brought back to canonical words on exit
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g2_dbl_d_asm_fn() {
    asm volatile(MBLS_G2_DBL_D_ASM);
}
MBLS_FN void g2_dbl_n(g2j* p, uint32_t n) {
    fp p0 = p->x.c0, p1 = p->x.c1, p2 = p->y.c0, p3 = p->y.c1, p4 = p->z.c0, p5 = p->z.c1;
    asm volatile(MBLS_ASM_CALL("mbls_g2_dbl_d_asm_fn") : MBLS_G2D_ARG_REGS(p), "+{s38}"(n) : : MBLS_G2D_ASM_CLOBBERS);
    p->x.c0 = p0; p->x.c1 = p1; p->y.c0 = p2; p->y.c1 = p3; p->z.c0 = p4; p->z.c1 = p5;
}
#else
MBLS_FN void g2_dbl_n(g2j* p, uint32_t n) { for (uint32_t i = 0; i < n; i++) g2_dbl(p, p); }
#endif
// [x]P, x = -0xd201000000010000 (uniform bit pattern: no lane divergence): runs of doublings between the set bits
MBLS_NOINLINE void g2_mul_x(g2j* r, const g2j* p) {
    g2j acc = *p;
    int i = 62;
    while (i >= 0) {
        int j = i;
        while (j > 0 && !((MBLS_X_ABS >> j) & 1)) j--;
        g2_dbl_n(&acc, (uint32_t)(i - j + 1));
        if ((MBLS_X_ABS >> j) & 1) g2_add(&acc, &acc, p);
        i = j - 1;
    }
    g2_neg(r, &acc);
}
// per-lane scalar, constant time in the scalar (used for blinding scalars and signing)
MBLS_NOINLINE void g2_mul(g2j* r, const g2j* p, const uint32_t* k, int nbits) {
    g2j acc; g2_set_inf(&acc);
    for (int i = nbits - 1; i >= 0; i--) {
        g2_dbl(&acc, &acc);
        g2j t; g2_add(&t, &acc, p);
        bool bit = (k[i >> 5] >> (i & 31)) & 1u;
        g2_select(&acc, bit, &t, &acc);
    }
    *r = acc;
}
MBLS_FN bool g2_eq(const g2j* a, const g2j* b) {   // projective equality
    bool ia = g2_is_inf(a), ib = g2_is_inf(b);
    fp2 za2 = fp2_sqr(a->z), zb2 = fp2_sqr(b->z);
    bool ex = fp2_eq(fp2_mul(a->x, zb2), fp2_mul(b->x, za2));
    bool ey = fp2_eq(fp2_mul(fp2_mul(a->y, zb2), b->z), fp2_mul(fp2_mul(b->y, za2), a->z));
    return (ia & ib) | (!ia & !ib & ex & ey);
}
// subgroup_check_g2 (reference src/signature.rs:29, src/aggregates.rs:184): amcl tests [r]P == O; the
// equivalent endomorphism test psi(P) == [x]P (Scott, "A note on group membership tests", 2021) gives
// the same boolean for every point of E'(Fp2) at ~1/4 of the cost. Infinity passes.
MBLS_NOINLINE bool g2_in_subgroup(const g2j* p) {
    g2j a, b; g2_psi(&a, p); g2_mul_x(&b, p);
    return g2_eq(&a, &b);
}
MBLS_FN bool g2_on_curve(const fp2& x, const fp2& y) {   // y^2 = x^3 + 4(1+i)
    fp four = fp_dbl(fp_dbl(fp_one())); fp2 b; b.c0 = four; b.c1 = four;
    return fp2_eq(fp2_sqr(y), fp2_add(fp2_mul(fp2_sqr(x), x), b));
}
// Budroni-Pintore cofactor clearing (RFC 9380 appendix G.3): [x^2-x-1]P + [x-1]psi(P) + psi^2(2P)
MBLS_NOINLINE void g2_clear_cofactor(g2j* r, const g2j* p) {
    g2j t1, t2, t3, n;
    g2_mul_x(&t1, p); g2_psi(&t2, p);
    g2_dbl(&t3, p); g2_psi2(&t3, &t3);
    g2_neg(&n, &t2); g2_add(&t3, &t3, &n);
    g2_add(&t2, &t1, &t2); g2_mul_x(&t2, &t2);
    g2_add(&t3, &t3, &t2); g2_neg(&n, &t1); g2_add(&t3, &t3, &n);
    g2_neg(&n, p); g2_add(r, &t3, &n);
}
MBLS_NOINLINE void g2_to_affine(fp2* x, fp2* y, bool* inf, const g2j* p) {
    *inf = g2_is_inf(p);
    fp2 zi = fp2_inv(p->z), zi2 = fp2_sqr(zi);
    *x = fp2_mul(p->x, zi2); *y = fp2_mul(fp2_mul(p->y, zi2), zi);
}
// Signature::from_bytes (reference src/signature.rs:43-46): 96 compressed bytes x.c1 || x.c0. No subgroup check.
template <bool INL>
MBLS_FN int g2_decode_compressed_t(fp2* x, fp2* y, bool* inf, const uint8_t* b) {
    uint8_t b0 = b[0];
    *inf = false; *x = fp2_zero(); *y = fp2_zero();
    if (!(b0 & 0x80)) return MBLS_DEC_SIZE;
    if (b0 & 0x40) { if ((b0 & 0x3F) || !mbls_all_zero(b + 1, 95)) return MBLS_DEC_POINT; *inf = true; return MBLS_DEC_OK; }
    fp r1 = fp_raw_from_be(b); r1[11] &= 0x1FFFFFFFu;
    fp r0 = fp_raw_from_be(b + 48);
    if (fp_raw_geq_p(r1) | fp_raw_geq_p(r0)) return MBLS_DEC_POINT;
    fp2 xm; xm.c0 = fp_to_mont(r0); xm.c1 = fp_to_mont(r1);
    fp four = fp_dbl(fp_dbl(fp_one())); fp2 bb; bb.c0 = four; bb.c1 = four;
    fp2 y2 = fp2_add(fp2_mul(fp2_sqr(xm), xm), bb), ym;
    if (!(INL ? fp2_sqrt_inl(&ym, &y2) : fp2_sqrt(&ym, &y2))) return MBLS_DEC_POINT;
    bool want = (b0 & 0x20) != 0;
    ym = fp2_select(fp2_lex_largest(ym) != want, fp2_neg(ym), ym);
    *x = xm; *y = ym; return MBLS_DEC_OK;
}
MBLS_NOINLINE int g2_decode_compressed(fp2* x, fp2* y, bool* inf, const uint8_t* b) { return g2_decode_compressed_t<false>(x, y, inf, b); }
MBLS_FN void g2_encode_compressed(uint8_t* b, const fp2& x, const fp2& y, bool inf) {   // reference src/amcl_utils.rs:62-64
    if (inf) { for (int i = 0; i < 96; i++) b[i] = 0; b[0] = 0xC0; return; }
    fp_raw_to_be(b, fp_from_mont(x.c1)); fp_raw_to_be(b + 48, fp_from_mont(x.c0));
    b[0] |= 0x80; if (fp2_lex_largest(y)) b[0] |= 0x20;
}
