// mbls_ops.h -- per-item bodies of the auxiliary kernels (codec, key validation, signing, aggregation
// outputs, field probes). Each op_* processes item i; mbls_kernels.hip wraps them in __global__ kernels.
#pragma once
#include "mbls_lanes.h"

// PublicKey::from_bytes / from_bytes_unchecked / from_uncompressed_bytes (reference src/keys.rs:140-175):
// in: wire bytes (48 or 96 per item), out: 96-byte uncompressed form + AmclError-style code per item.
// validate != 0 adds KeyValidate (reject infinity and points outside G1, reference src/keys.rs:181-186).
MBLS_FN void op_g1_decode(uint64_t i, const uint8_t* in, int fmt, int validate, uint8_t* out96, uint8_t* err) {
    fp x, y; bool inf;
    int e = (fmt == MBLS_PK_COMPRESSED) ? g1_decode_compressed(&x, &y, &inf, in + 48 * i) : g1_decode_uncompressed(&x, &y, &inf, in + 96 * i);
    if (!e && validate) {
        g1j p; p.x = x; p.y = y; p.z = fp_one();
        if (inf) e = MBLS_DEC_POINT;
        else if (!g1_in_subgroup(&p)) e = MBLS_DEC_POINT;
    }
    if (e) { for (int j = 0; j < 96; j++) out96[96 * i + j] = 0; }
    else g1_encode_uncompressed(out96 + 96 * i, x, y, inf);
    err[i] = (uint8_t)e;
}
// PublicKey::key_validate on decoded (uncompressed) keys -> 1/0
MBLS_FN void op_g1_key_validate(uint64_t i, const uint8_t* in96, uint8_t* ok) {
    fp x, y; bool inf; int e = g1_decode_uncompressed(&x, &y, &inf, in96 + 96 * i);
    g1j p; p.x = x; p.y = y; p.z = fp_one();
    ok[i] = (!e && !inf && g1_in_subgroup(&p)) ? 1 : 0;
}
// PublicKey::as_bytes (reference src/keys.rs:158-160): uncompressed -> compressed
MBLS_FN void op_g1_compress(uint64_t i, const uint8_t* in96, uint8_t* out48, uint8_t* err) {
    fp x, y; bool inf; int e = g1_decode_uncompressed(&x, &y, &inf, in96 + 96 * i);
    if (e) { for (int j = 0; j < 48; j++) out48[48 * i + j] = 0; } else g1_encode_compressed(out48 + 48 * i, x, y, inf);
    err[i] = (uint8_t)e;
}
// Signature::from_bytes validity (reference src/signature.rs:43-46): err code per item; optional subgroup flag
MBLS_FN void op_g2_check(uint64_t i, const uint8_t* in96, uint8_t* err, uint8_t* in_g2) {
    fp2 x, y; bool inf; int e = g2_decode_compressed(&x, &y, &inf, in96 + 96 * i);
    err[i] = (uint8_t)e;
    if (in_g2) {
        g2j p; p.x = x; p.y = y; p.z = fp2_one(); if (inf || e) g2_set_inf(&p);
        in_g2[i] = (!e && g2_in_subgroup(&p)) ? 1 : 0;
    }
}
// key table record from wire bytes: the decode of op_g1_decode, stored as affine Montgomery limbs (mbls_lanes.h, keyrec_load)
MBLS_FN void op_keytable_append(uint64_t i, const uint8_t* in, int fmt, int validate, uint32_t* recs, uint8_t* errs) {
    fp x, y; bool inf;
    int e = (fmt == MBLS_PK_COMPRESSED) ? g1_decode_compressed(&x, &y, &inf, in + 48 * i) : g1_decode_uncompressed(&x, &y, &inf, in + 96 * i);
    if (!e && validate) {
        g1j p; p.x = x; p.y = y; p.z = fp_one();
        if (inf) e = MBLS_DEC_POINT;
        else if (!g1_in_subgroup(&p)) e = MBLS_DEC_POINT;
    }
    uint32_t* o = recs + MBLS_KEYREC_DWORDS * i;
    if (e) { x = fp_zero(); y = fp_zero(); }
#pragma unroll
    for (int t = 0; t < 12; t++) { o[t] = x[t]; o[12 + t] = y[t]; }
    o[24] = (e ? (MBLS_KEYFLAG_BAD | MBLS_KEYFLAG_INF) : 0u) | (inf ? MBLS_KEYFLAG_INF : 0u);
    for (int t = 25; t < MBLS_KEYREC_DWORDS; t++) o[t] = 0;
    errs[i] = (uint8_t)e;
}
// PublicKey::as_uncompressed_bytes of table entries (src/keys.rs:163-165); err = InvalidPoint for an entry that failed to decode
MBLS_FN void op_keytable_export(uint64_t i, const uint32_t* recs, uint8_t* out96, uint8_t* errs) {
    fp x, y; uint32_t f; keyrec_load(&x, &y, &f, recs, ~(uint64_t)0, (uint32_t)i);
    if (f & MBLS_KEYFLAG_BAD) { for (int j = 0; j < 96; j++) out96[96 * i + j] = 0; errs[i] = MBLS_DEC_POINT; return; }
    g1_encode_uncompressed(out96 + 96 * i, x, y, (f & MBLS_KEYFLAG_INF) != 0);
    errs[i] = MBLS_DEC_OK;
}
// Signature::from_bytes into affine Montgomery limbs (48 dwords: x.c0, x.c1, y.c0, y.c1) + flag byte (decode error code | 0x80 = infinity)
MBLS_FN void op_g2_decode_affine(uint64_t i, const uint8_t* sigs96, uint32_t* xy, uint8_t* flags) {
    fp2 x, y; bool inf; int e = g2_decode_compressed(&x, &y, &inf, sigs96 + 96 * i);
    uint32_t* o = xy + 48 * i;
#pragma unroll
    for (int t = 0; t < 12; t++) { o[t] = x.c0[t]; o[12 + t] = x.c1[t]; o[24 + t] = y.c0[t]; o[36 + t] = y.c1[t]; }
    flags[i] = (uint8_t)(e | ((inf || e) ? 0x80 : 0));
}
MBLS_FN void g2_encode_jacobian(uint8_t* out96, const g2j* p);
// AggregateSignature::aggregate (src/aggregates.rs:100-106): sum of k decoded signatures starting from infinity
MBLS_FN void op_g2_sum(uint64_t i, const uint32_t* xy, const uint8_t* flags, uint32_t k, uint8_t* out96, uint8_t* errs) {
    g2j acc; g2_set_inf(&acc); int e = 0;
    for (uint32_t j = 0; j < k; j++) {
        const uint32_t* o = xy + 48 * (uint64_t)j;
        g2j q;
#pragma unroll
        for (int t = 0; t < 12; t++) { q.x.c0[t] = o[t]; q.x.c1[t] = o[12 + t]; q.y.c0[t] = o[24 + t]; q.y.c1[t] = o[36 + t]; }
        q.z = fp2_one();
        uint32_t f = flags[j];
        if ((f & 0x7F) && !e) e = (int)(f & 0x7F);
        if (f & 0x80) g2_set_inf(&q);
        g2_add(&acc, &acc, &q);
    }
    if (e) { for (int j = 0; j < 96; j++) out96[96 * i + j] = 0; }
    else g2_encode_jacobian(out96 + 96 * i, &acc);
    errs[i] = (uint8_t)e;
}
MBLS_FN void g2_encode_jacobian(uint8_t* out96, const g2j* p) {
    fp2 x, y; bool inf; g2_to_affine(&x, &y, &inf, p); g2_encode_compressed(out96, x, y, inf);
}
// AggregateSignature::add (reference src/aggregates.rs:114-116): out = a + b on compressed signatures
MBLS_FN void op_g2_add(uint64_t i, const uint8_t* a96, const uint8_t* b96, uint8_t* out96, uint8_t* err) {
    fp2 x, y; bool inf; g2j p, q;
    int e = g2_decode_compressed(&x, &y, &inf, a96 + 96 * i);
    p.x = x; p.y = y; p.z = fp2_one(); if (inf) g2_set_inf(&p);
    int e2 = g2_decode_compressed(&x, &y, &inf, b96 + 96 * i);
    q.x = x; q.y = y; q.z = fp2_one(); if (inf) g2_set_inf(&q);
    if (!e) e = e2;
    if (e) { for (int j = 0; j < 96; j++) out96[96 * i + j] = 0; }
    else { g2_add(&p, &p, &q); g2_encode_jacobian(out96 + 96 * i, &p); }
    err[i] = (uint8_t)e;
}
// AggregatePublicKey::add (reference src/aggregates.rs:68-77) on decoded keys
MBLS_FN void op_g1_add(uint64_t i, const uint8_t* a96, const uint8_t* b96, uint8_t* out96, uint8_t* err) {
    fp x, y, x2, y2; bool inf, inf2;
    int e = g1_decode_uncompressed(&x, &y, &inf, a96 + 96 * i), e2 = g1_decode_uncompressed(&x2, &y2, &inf2, b96 + 96 * i);
    if (!e) e = e2;
    if (e) { for (int j = 0; j < 96; j++) out96[96 * i + j] = 0; }
    else {
        g1j p; p.x = x; p.y = y; p.z = fp_one(); if (inf) g1_set_inf(&p);
        g1_madd(&p, &p, x2, y2, inf2);
        g1_to_affine(&x, &y, &inf, &p); g1_encode_uncompressed(out96 + 96 * i, x, y, inf);
    }
    err[i] = (uint8_t)e;
}
// the aggregate key of item i (left in the workspace by lane_aggregate) as 96 uncompressed bytes
MBLS_FN void op_apk_export(const mbls_ws& ws, uint64_t i, uint8_t* out96) {
    g1j a; a.x = ws_ld(ws, MBLS_SLOT_APK, i); a.y = ws_ld(ws, MBLS_SLOT_APK + 1, i); a.z = ws_ld(ws, MBLS_SLOT_APK + 2, i);
    fp x, y; bool inf; g1_to_affine(&x, &y, &inf, &a); g1_encode_uncompressed(out96 + 96 * i, x, y, inf);
}
MBLS_FN void scalar_from_be32(uint32_t* k, const uint8_t* b) {
    for (int j = 0; j < 8; j++) { const uint8_t* q = b + 28 - 4 * j; k[j] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; }
}
// Signature::new (reference src/signature.rs:17-21): sig = [sk] H(msg), compressed. The plain ladder: the host emulation's statement of the
// operation (tests/host_emul); the library signs with k_sign_blind (mbls_kernels.hip)
MBLS_FN void op_sign(uint64_t i, const uint8_t* sks32, const uint8_t* msgs, uint32_t mlen, uint8_t* out96) {
    uint32_t k[8]; scalar_from_be32(k, sks32 + 32 * i);
    g2j h; hash_to_g2(&h, msgs + (uint64_t)mlen * i, mlen, MBLS_DST_POP, MBLS_DST_POP_LEN);
    g2_mul(&h, &h, k, 256);
    g2_encode_jacobian(out96 + 96 * i, &h);
}
// PublicKey::from_secret_key (reference src/keys.rs:124-137): pk = [sk] G1
MBLS_FN void op_sk_to_pk(uint64_t i, const uint8_t* sks32, int fmt, uint8_t* out) {
    uint32_t k[8]; scalar_from_be32(k, sks32 + 32 * i);
    g1j g; g.x = fp_load_const(MBLS_G1_X); g.y = fp_load_const(MBLS_G1_Y); g.z = fp_one();
    g1_mul(&g, &g, k, 256);
    fp x, y; bool inf; g1_to_affine(&x, &y, &inf, &g);
    if (fmt == MBLS_PK_COMPRESSED) g1_encode_compressed(out + 48 * i, x, y, inf); else g1_encode_uncompressed(out + 96 * i, x, y, inf);
}
// hash_to_curve_g2 (reference src/amcl_utils.rs:33-35), compressed output
MBLS_FN void op_hash_to_g2(uint64_t i, const uint8_t* msgs, uint32_t mlen, uint8_t* out96) {
    g2j h; hash_to_g2(&h, msgs + (uint64_t)mlen * i, mlen, MBLS_DST_POP, MBLS_DST_POP_LEN);
    g2_encode_jacobian(out96 + 96 * i, &h);
}
// field probe on canonical 48-byte big-endian values (parity tests of the hand-written routines). op: 0 a*b, 1 a^2,
// 2 Fp2 product and 3 Fp2 square over element pairs (2i, 2i+1) = (real, imaginary), 4 a^-1 (0 -> 0), 5 a^((p-3)/4),
// 6 the paired-product routine on elements 2i and 2i+1
MBLS_FN void op_fp_mul(uint64_t i, uint64_t n, const uint8_t* a48, const uint8_t* b48, uint8_t* out48, int op) {
    if (op == 2 || op == 3 || op == 6) {
        if ((i & 1) || i + 1 >= n) return;
        fp a0 = fp_to_mont(fp_raw_from_be(a48 + 48 * i)), a1 = fp_to_mont(fp_raw_from_be(a48 + 48 * (i + 1)));
        fp b0 = fp_to_mont(fp_raw_from_be(b48 + 48 * i)), b1 = fp_to_mont(fp_raw_from_be(b48 + 48 * (i + 1)));
        fp r0, r1;
        if (op == 6) fp_mul_pair(&r0, &r1, a0, b0, a1, b1);
        else {
            fp2 a, b; a.c0 = a0; a.c1 = a1; b.c0 = b0; b.c1 = b1;
            fp2 r = op == 2 ? fp2_mul(a, b) : fp2_sqr(a);
            r0 = r.c0; r1 = r.c1;
        }
        fp_raw_to_be(out48 + 48 * i, fp_from_mont(r0)); fp_raw_to_be(out48 + 48 * (i + 1), fp_from_mont(r1));
        return;
    }
    fp a = fp_to_mont(fp_raw_from_be(a48 + 48 * i)), b = fp_to_mont(fp_raw_from_be(b48 + 48 * i));
    fp r = op == 1 ? fp_sqr(a) : op == 4 ? fp_inv(a) : op == 5 ? fp_pow_pm3d4(a) : fp_mul(a, b);
    fp_raw_to_be(out48 + 48 * i, fp_from_mont(r));
}
