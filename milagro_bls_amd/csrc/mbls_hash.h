// mbls_hash.h -- hash_to_curve_g2 for one message per lane (RFC 9380 suite BLS12381G2_XMD:SHA-256_SSWU_RO_).
// Replaces amcl's bls381::utils::hash_to_curve_g2(msg, DST_G2) behind reference src/amcl_utils.rs:33-35:
//   expand_message_xmd(SHA-256, 256 B) -> 2 x Fp2 -> simplified SWU on E' -> 3-isogeny -> add -> clear cofactor.
// The SWU map uses two Fp exponentiations per field element (complex-method square root in Fp2 whose
// first exponentiation also decides is_square and yields the inverse needed for the affine y), and the
// isogeny lands directly in Jacobian coordinates because its denominators are (x+k)^2 and (x+k)^3.
#pragma once
#include "mbls_curve.h"

// ------------------------------------------------------------------------------------------------ SHA-256
MBLS_CONST uint32_t MBLS_SHA_K[64] = {
0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
// the POP ciphersuite tag (amcl proof_of_possession::DST_G2, imported at reference src/amcl_utils.rs:6)
MBLS_CONST uint8_t MBLS_DST_POP[43] = {'B','L','S','_','S','I','G','_','B','L','S','1','2','3','8','1','G','2','_','X','M','D',':','S','H','A','-','2','5','6','_','S','S','W','U','_','R','O','_','P','O','P','_'};
#define MBLS_DST_POP_LEN 43

MBLS_FN uint32_t mbls_ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
// state and message block travel by value (vector types: registers), so that no caller needs an addressable array for them
typedef uint32_t mbls_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t mbls_u32x16 __attribute__((ext_vector_type(16)));
MBLS_NOINLINE mbls_u32x8 sha256_compress(mbls_u32x8 h, mbls_u32x16 blk) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = blk[i];
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t wi;
        if (i < 16) wi = w[i];
        else {
            uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            uint32_t s0 = mbls_ror(w15, 7) ^ mbls_ror(w15, 18) ^ (w15 >> 3), s1 = mbls_ror(w2, 17) ^ mbls_ror(w2, 19) ^ (w2 >> 10);
            wi = w[i & 15] + s0 + w[(i + 9) & 15] + s1; w[i & 15] = wi;
        }
        uint32_t S1 = mbls_ror(e, 6) ^ mbls_ror(e, 11) ^ mbls_ror(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + MBLS_SHA_K[i] + wi;
        uint32_t S0 = mbls_ror(a, 2) ^ mbls_ror(a, 13) ^ mbls_ror(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    mbls_u32x8 r = h;
    r[0] += a; r[1] += b; r[2] += c; r[3] += d; r[4] += e; r[5] += f; r[6] += g; r[7] += hh;
    return r;
}
// SHA-256 of a "virtual" byte string described by byte_at(pos), pos in [0, len)
template <typename F>
MBLS_FN mbls_u32x8 sha256_virtual_v(uint32_t len, F byte_at) {
    mbls_u32x8 h = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint32_t padded = ((len + 9 + 63) / 64) * 64;
    uint64_t bits = (uint64_t)len * 8;
    for (uint32_t off = 0; off < padded; off += 64) {
        mbls_u32x16 blk;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            uint32_t wv = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                uint32_t pos = off + 4 * j + k; uint32_t by;
                if (pos < len) by = byte_at(pos);
                else if (pos == len) by = 0x80;
                else if (pos >= padded - 8) by = (uint32_t)(bits >> (8 * (padded - 1 - pos))) & 0xFF;
                else by = 0;
                wv = (wv << 8) | by;
            }
            blk[j] = wv;
        }
        h = sha256_compress(h, blk);
    }
    return h;
}
template <typename F>
MBLS_FN void sha256_virtual(uint32_t* digest, uint32_t len, F byte_at) {
    mbls_u32x8 h = sha256_virtual_v(len, byte_at);
#pragma unroll
    for (int i = 0; i < 8; i++) digest[i] = h[i];
}
MBLS_FN uint32_t mbls_digest_byte_v(mbls_u32x8 d, uint32_t i) {          // byte i of a digest held as eight big-endian words, without indexing memory
    uint32_t w = d[0];
#pragma unroll
    for (uint32_t q = 1; q < 8; q++) w = (i >> 2) == q ? d[q] : w;
    return (w >> (8 * (3 - (i & 3)))) & 0xFF;
}
MBLS_FN uint32_t mbls_digest_byte(const uint32_t* d, uint32_t i) { return (d[i >> 2] >> (8 * (3 - (i & 3)))) & 0xFF; }

// expand_message_xmd to 256 bytes = b_1 .. b_8, each as 8 big-endian words: out[8*(i-1) + j]
MBLS_NOINLINE void expand_message_xmd_256(uint32_t* out, const uint8_t* msg, uint32_t mlen, const uint8_t* dst, uint32_t dlen) {
    uint32_t b0[8], bi[8];
    // b_0 = H(Z_pad(64) || msg || I2OSP(256,2) || I2OSP(0,1) || DST || I2OSP(len(DST),1))
    sha256_virtual(b0, 64 + mlen + 3 + dlen + 1, [&](uint32_t pos) -> uint32_t {
        if (pos < 64) return 0;
        pos -= 64; if (pos < mlen) return msg[pos];
        pos -= mlen; if (pos < 3) return pos == 0 ? 1u : 0u;
        pos -= 3; if (pos < dlen) return dst[pos];
        return dlen;
    });
    for (uint32_t i = 1; i <= 8; i++) {
        // b_i = H((b_0 xor b_(i-1)) || I2OSP(i,1) || DST'), b_(0) taken as zero for i = 1
        uint32_t x[8];
        for (int j = 0; j < 8; j++) x[j] = (i == 1) ? b0[j] : (b0[j] ^ bi[j]);
        sha256_virtual(bi, 32 + 1 + dlen + 1, [&](uint32_t pos) -> uint32_t {
            if (pos < 32) return mbls_digest_byte(x, pos);
            pos -= 32; if (pos == 0) return i;
            pos -= 1; if (pos < dlen) return dst[pos];
            return dlen;
        });
        for (int j = 0; j < 8; j++) out[8 * (i - 1) + j] = bi[j];
    }
}
// the same from digests held in registers
MBLS_FN fp fp_from_two_digests_v(mbls_u32x8 hi, mbls_u32x8 lo) {
    fp rh = 0, rl = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { rh[i] = hi[7 - i]; rl[i] = lo[7 - i]; }
    return fp_add(fp_mul(fp_to_mont(rh), fp_load_const(MBLS_TWO_256)), fp_to_mont(rl));
}
// one block b_i = H((b_0 xor b_(i-1)) || I2OSP(i,1) || DST || I2OSP(len(DST),1)) of expand_message_xmd
MBLS_FN mbls_u32x8 expand_xmd_block(mbls_u32x8 b0, mbls_u32x8 prev, uint32_t i, const uint8_t* dst, uint32_t dlen) {
    mbls_u32x8 x = b0 ^ prev;
    return sha256_virtual_v(32 + 1 + dlen + 1, [&](uint32_t pos) -> uint32_t {
        if (pos < 32) return mbls_digest_byte_v(x, pos);
        pos -= 32; if (pos == 0) return i;
        pos -= 1; if (pos < dlen) return dst[pos];
        return dlen;
    });
}
MBLS_FN mbls_u32x8 expand_xmd_b0(const uint8_t* msg, uint32_t mlen, const uint8_t* dst, uint32_t dlen) {
    return sha256_virtual_v(64 + mlen + 3 + dlen + 1, [&](uint32_t pos) -> uint32_t {
        if (pos < 64) return 0;
        pos -= 64; if (pos < mlen) return msg[pos];
        pos -= mlen; if (pos < 3) return pos == 0 ? 1u : 0u;
        pos -= 3; if (pos < dlen) return dst[pos];
        return dlen;
    });
}
// OS2IP(64 bytes) mod p in Montgomery form; the 64 bytes are the two digests hi, lo (big-endian words)
MBLS_FN fp fp_from_two_digests(const uint32_t* hi, const uint32_t* lo) {
    fp rh = 0, rl = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { rh[i] = hi[7 - i]; rl[i] = lo[7 - i]; }
    return fp_add(fp_mul(fp_to_mont(rh), fp_load_const(MBLS_TWO_256)), fp_to_mont(rl));
}

// ------------------------------------------------------------------------------------------------ SSWU + isogeny
// Simplified SWU for E': y^2 = x^3 + A'x + B' followed by the 3-isogeny to E, result in Jacobian coordinates.
MBLS_FN void map_to_curve_g2_inl(g2j* out, const fp2* up) {
    const fp2 u = *up;
    const fp2 A = fp2_load_const(MBLS_SSWU_A), B = fp2_load_const(MBLS_SSWU_B), Z = fp2_load_const(MBLS_SSWU_Z);
    fp2 tv1 = fp2_mul(Z, fp2_sqr(u));                        // Z u^2
    fp2 tv2 = fp2_add(fp2_sqr(tv1), tv1);                    // Z^2 u^4 + Z u^2
    fp2 xn = fp2_mul(B, fp2_add(tv2, fp2_one()));            // x1 = xn / xd
    fp2 xd = fp2_mul(A, fp2_select(fp2_is_zero(tv2), Z, fp2_neg(tv2)));
    fp2 xd2 = fp2_sqr(xd), D = fp2_mul(xd2, xd);             // D = xd^3
    fp2 N = fp2_add(fp2_mul(fp2_add(fp2_sqr(xn), fp2_mul(A, xd2)), xn), fp2_mul(B, D));   // gx1 = N / D
    fp2 g = fp2_mul(N, D);                                   // gx1 = g / D^2
    // first exponentiation: w1 = norm(g)^((p-3)/4): sqrt candidate, quadratic character and inverse of norm(g)
    fp nd = fp2_norm(xd), nN = fp2_norm(N);
    fp ng = fp_mul(nN, fp_mul(fp_sqr(nd), nd));              // norm(g) = norm(N) norm(xd)^3
    fp w1 = fp_pow_pm3d4(ng);
    fp s = fp_mul(w1, ng);
    bool sq1 = fp_eq(fp_sqr(s), ng);                         // gx1 is a square in Fp2
    fp w1sq = fp_sqr(w1);
    fp inv_ng = fp_select(sq1, w1sq, fp_neg(w1sq));          // 1/norm(g) = chi * w1^2
    fp inv_nd = fp_mul(fp_mul(nN, fp_sqr(nd)), inv_ng);      // 1/norm(xd)
    fp2 inv_xd; inv_xd.c0 = fp_mul(xd.c0, inv_nd); inv_xd.c1 = fp_neg(fp_mul(xd.c1, inv_nd));
    // if gx1 is not a square, gx2 = (Z u^2)^3 gx1 is: take G = tv1 * g, whose norm has the root norm(u) C5 s
    fp2 G = fp2_select(sq1, g, fp2_mul(tv1, g));
    fp s_alt = fp_mul(fp_mul(fp2_norm(u), fp_load_const(MBLS_SQRT_C5)), s);
    fp S = fp_select(sq1, s, s_alt);
    // second exponentiation: complex-method root r of G
    fp t = fp_half(fp_add(G.c0, S)), t_alt = fp_half(fp_sub(G.c0, S));
    t = fp_select(fp_is_zero(t), t_alt, t);
    fp w2 = fp_pow_pm3d4(t);
    fp x0 = fp_mul(w2, t);
    bool chi = fp_eq(fp_sqr(x0), t);
    fp other = fp_mul(fp_half(G.c1), fp_mul(x0, fp_sqr(w2)));
    fp2 r; r.c0 = fp_select(chi, x0, other); r.c1 = fp_select(chi, other, x0);
    // affine point on E': square case (xn/xd, r/D); otherwise (tv1 xn/xd, tv1 r/D) since r^2 = tv1 g
    fp2 inv_D = fp2_mul(fp2_sqr(inv_xd), inv_xd);
    fp2 x = fp2_mul(xn, inv_xd), y = fp2_mul(r, inv_D);
    x = fp2_select(sq1, x, fp2_mul(tv1, x));
    y = fp2_select(sq1, y, fp2_mul(tv1, y));
    y = fp2_select(fp2_sgn0(u) != fp2_sgn0(y), fp2_neg(y), y);
    // 3-isogeny (RFC 9380 appendix E.3): x_den = (x+k)^2, y_den = (x+k)^3 => Jacobian (x_num, y y_num, x + k)
    fp2 xnum = fp2_load_const(MBLS_ISO3_XNUM[3]), ynum = fp2_load_const(MBLS_ISO3_YNUM[3]);
    for (int i = 2; i >= 0; i--) {
        xnum = fp2_add(fp2_mul(xnum, x), fp2_load_const(MBLS_ISO3_XNUM[i]));
        ynum = fp2_add(fp2_mul(ynum, x), fp2_load_const(MBLS_ISO3_YNUM[i]));
    }
    out->x = xnum; out->y = fp2_mul(y, ynum); out->z = fp2_add(x, fp2_load_const(MBLS_ISO3_K));
}
MBLS_NOINLINE void map_to_curve_g2(g2j* out, const fp2* up) { map_to_curve_g2_inl(out, up); }
// hash_to_field + the two map_to_curve evaluations of hash_to_curve_g2: points of E'(Fp2) in Jacobian coordinates
template <bool INL>
MBLS_FN void hash_to_g2_maps_t(g2j* q0, g2j* q1, const uint8_t* msg, uint32_t mlen, const uint8_t* dst, uint32_t dlen) {
    uint32_t ub[64];
    expand_message_xmd_256(ub, msg, mlen, dst, dlen);
    fp2 u0, u1;
    u0.c0 = fp_from_two_digests(ub, ub + 8); u0.c1 = fp_from_two_digests(ub + 16, ub + 24);
    u1.c0 = fp_from_two_digests(ub + 32, ub + 40); u1.c1 = fp_from_two_digests(ub + 48, ub + 56);
    if (INL) { map_to_curve_g2_inl(q0, &u0); map_to_curve_g2_inl(q1, &u1); }
    else { map_to_curve_g2(q0, &u0); map_to_curve_g2(q1, &u1); }
}
MBLS_NOINLINE void hash_to_g2_maps(g2j* q0, g2j* q1, const uint8_t* msg, uint32_t mlen, const uint8_t* dst, uint32_t dlen) {
    hash_to_g2_maps_t<false>(q0, q1, msg, mlen, dst, dlen);
}
// hash_to_curve_g2 (reference src/amcl_utils.rs:33-35); result in Jacobian coordinates, in G2
MBLS_NOINLINE void hash_to_g2(g2j* out, const uint8_t* msg, uint32_t mlen, const uint8_t* dst, uint32_t dlen) {
    g2j q0, q1;
    hash_to_g2_maps(&q0, &q1, msg, mlen, dst, dlen);
    g2_add(&q0, &q0, &q1);
    g2_clear_cofactor(out, &q0);
}
