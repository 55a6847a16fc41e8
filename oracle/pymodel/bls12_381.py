"""Big-integer Python model of BLS12-381 signature verification (TEST INFRASTRUCTURE ONLY).

This file is part of the oracle: it is imported only by tests/, by the golden-vector
generator tests/golden/gen_golden.py and by oracle/gen_constants.py. The product path
(milagro_bls_amd/) never imports it.

It restates, with plain Python integers and the slowest/most obvious formulas (affine
group law, generic Fp12 polynomial arithmetic, final exponentiation by the literal
exponent), the algorithms that sigp/milagro_bls reaches through the out-of-tree `amcl`
crate (reference src/amcl_utils.rs:6-21 imports; amcl source absent from /root/reference):

  * hash_to_curve_g2        reference src/amcl_utils.rs:33-35  (RFC 9380 suite
                            BLS12381G2_XMD:SHA-256_SSWU_RO_, DST = POP ciphersuite tag)
  * ate2_evaluation         reference src/amcl_utils.rs:38-42
  * (de)compress_g1/g2      reference src/amcl_utils.rs:46-74  (ZCash format)
  * subgroup_check_g1/g2    reference src/keys.rs:182, src/signature.rs:29
  * Signature::{new,verify} reference src/signature.rs:17-40
  * AggregateSignature::fast_aggregate_verify etc.  reference src/aggregates.rs:130-316

Its job is to (1) derive every curve constant the C oracle and the HIP kernels need from
first principles (so a mistyped constant cannot hide), (2) be pinned against the golden
vectors the reference holds and the published RFC 9380 / Eth2 vectors, and (3) generate
the fixtures under tests/golden/.
"""
import hashlib

# ----------------------------------------------------------------------------- parameters
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
X_ABS = 0xd201000000010000          # |x|, the BLS parameter is x = -X_ABS
X = -X_ABS
assert R == X**4 - X**2 + 1
assert P == ((X - 1)**2 * R) // 3 + X
H1 = 0x396c8c005555e1568c00aaab0000aaab   # G1 cofactor = (x-1)^2/3
assert H1 == (X - 1)**2 // 3

DST_POP = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"

G1_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
G2_X = (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
        0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e)
G2_Y = (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
        0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)

# ----------------------------------------------------------------------------- Fp
def fp_inv(a):
    return pow(a, P - 2, P)

def fp_sqrt(a):
    """p = 3 mod 4: candidate a^((p+1)/4); None if a is a non-residue."""
    a %= P
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a else None

# ----------------------------------------------------------------------------- Fp2 = Fp[i]/(i^2+1)
def f2(a, b=0):
    return (a % P, b % P)

F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (1, 1)                      # the sextic non-residue 1+i

def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_sqr(a): return f2_mul(a, a)
def f2_muls(a, k): return (a[0] * k % P, a[1] * k % P)
def f2_conj(a): return (a[0], (-a[1]) % P)
def f2_inv(a):
    n = fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * n % P, (-a[1]) * n % P)
def f2_is_zero(a): return a[0] % P == 0 and a[1] % P == 0
def f2_eq(a, b): return (a[0] - b[0]) % P == 0 and (a[1] - b[1]) % P == 0
def f2_pow(a, e):
    r = F2_ONE
    for bit in bin(e)[2:]:
        r = f2_sqr(r)
        if bit == '1':
            r = f2_mul(r, a)
    return r

def f2_is_square(a):
    # a is a square in Fp2 iff its norm is a square in Fp
    n = (a[0] * a[0] + a[1] * a[1]) % P
    return n == 0 or pow(n, (P - 1) // 2, P) == 1

def f2_sqrt(a):
    """Some square root of a in Fp2, or None. Complex method."""
    a = f2(*a)
    if f2_is_zero(a):
        return F2_ZERO
    if a[1] == 0:
        s = fp_sqrt(a[0])
        if s is not None:
            return (s, 0)
        s = fp_sqrt((-a[0]) % P)      # sqrt(-1) = i
        return (0, s)
    n = fp_sqrt((a[0] * a[0] + a[1] * a[1]) % P)
    if n is None:
        return None
    inv2 = fp_inv(2)
    for nn in (n, (-n) % P):
        t = (a[0] + nn) * inv2 % P
        x0 = fp_sqrt(t)
        if x0 is None or x0 == 0:
            continue
        x1 = a[1] * fp_inv(2 * x0 % P) % P
        if f2_eq(f2_sqr((x0, x1)), a):
            return (x0, x1)
    return None

def f2_sgn0(a):
    """RFC 9380 section 4.1 sgn0 for m = 2."""
    sign_0 = a[0] % 2
    zero_0 = 1 if a[0] % P == 0 else 0
    sign_1 = a[1] % 2
    return sign_0 | (zero_0 & sign_1)

def fp_lex_largest(y):
    return y % P > (P - 1) // 2

def f2_lex_largest(y):
    """ZCash rule: compare c1 first, then c0."""
    if y[1] % P != 0:
        return y[1] % P > (P - 1) // 2
    return y[0] % P > (P - 1) // 2

# ----------------------------------------------------------------------------- Fp12 = Fp2[w]/(w^6 - xi)
# element = list of 6 Fp2 coefficients of w^0..w^5.  Tower view used by the C oracle and
# the kernels: Fp6 = Fp2[v]/(v^3-xi), Fp12 = Fp6[w]/(w^2-v), v = w^2:
#   (a0 + a1 v + a2 v^2) + (b0 + b1 v + b2 v^2) w  <->  [a0, b0, a1, b1, a2, b2]
F12_ONE = [F2_ONE] + [F2_ZERO] * 5

def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if f2_is_zero(a[i]):
            continue
        for j in range(6):
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    r = t[:6]
    for k in range(6, 11):
        r[k - 6] = f2_add(r[k - 6], f2_mul(t[k], XI))
    return r

def f12_sqr(a): return f12_mul(a, a)
def f12_eq(a, b): return all(f2_eq(x, y) for x, y in zip(a, b))
def f12_is_one(a): return f12_eq(a, F12_ONE)

def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == '1':
            r = f12_mul(r, a)
    return r

def f12_conj(a):
    """a^(p^6): w -> -w."""
    return [a[k] if k % 2 == 0 else f2_neg(a[k]) for k in range(6)]

def _frob_coeffs():
    # (c w^k)^p = conj(c) * w^(kp) = conj(c) * xi^(k(p-1)/6) * w^k
    g = f2_pow(XI, (P - 1) // 6)
    out = [F2_ONE]
    for _ in range(5):
        out.append(f2_mul(out[-1], g))
    return out
FROB_W = _frob_coeffs()          # FROB_W[k] = xi^(k(p-1)/6)

def f12_frob(a):
    return [f2_mul(f2_conj(a[k]), FROB_W[k]) for k in range(6)]

def f12_inv(a):
    # generic: a^(p^12-2) would be far too slow; use norm descent via conjugates:
    # a^-1 = (prod_{j=1..11} frob^j(a)) / N(a),  N(a) in Fp.  Cheap enough for a model.
    acc = F12_ONE
    t = a
    for _ in range(11):
        t = f12_frob(t)
        acc = f12_mul(acc, t)
    n = f12_mul(acc, a)           # the norm, lies in Fp
    assert all(f2_is_zero(c) for c in n[1:]) and n[0][1] == 0
    ninv = fp_inv(n[0][0])
    return [f2_muls(c, ninv) for c in acc]

# ----------------------------------------------------------------------------- curves (affine; None = infinity)
B1 = 4
B2 = (4, 4)                       # 4*(1+i)

def g1_on_curve(pt):
    if pt is None: return True
    x, y = pt
    return (y * y - x * x * x - B1) % P == 0

def g1_add(a, b):
    if a is None: return b
    if b is None: return a
    x1, y1 = a; x2, y2 = b
    if (x1 - x2) % P == 0:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * fp_inv(2 * y1) % P
    else:
        lam = (y2 - y1) * fp_inv(x2 - x1) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)

def g1_neg(a):
    return None if a is None else (a[0], (-a[1]) % P)

def g1_mul(a, k):
    if k < 0:
        return g1_mul(g1_neg(a), -k)
    r = None
    for bit in bin(k)[2:] if k else '':
        r = g1_add(r, r)
        if bit == '1':
            r = g1_add(r, a)
    return r

def g2_on_curve(pt):
    if pt is None: return True
    x, y = pt
    return f2_eq(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B2))

def g2_add(a, b):
    if a is None: return b
    if b is None: return a
    x1, y1 = a; x2, y2 = b
    if f2_eq(x1, x2):
        if f2_is_zero(f2_add(y1, y2)):
            return None
        lam = f2_mul(f2_muls(f2_sqr(x1), 3), f2_inv(f2_muls(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))

def g2_neg(a):
    return None if a is None else (a[0], f2_neg(a[1]))

def g2_mul(a, k):
    if k < 0:
        return g2_mul(g2_neg(a), -k)
    r = None
    for bit in bin(k)[2:] if k else '':
        r = g2_add(r, r)
        if bit == '1':
            r = g2_add(r, a)
    return r

def g2_eq(a, b):
    if a is None or b is None:
        return a is None and b is None
    return f2_eq(a[0], b[0]) and f2_eq(a[1], b[1])

G1 = (G1_X, G1_Y)
G2 = (G2_X, G2_Y)

# psi = twist o frobenius o untwist on E'(Fp2)
PSI_CX = f2_inv(f2_pow(XI, (P - 1) // 3))
PSI_CY = f2_inv(f2_pow(XI, (P - 1) // 2))

def g2_psi(a):
    if a is None: return None
    return (f2_mul(f2_conj(a[0]), PSI_CX), f2_mul(f2_conj(a[1]), PSI_CY))

def subgroup_check_g1(a):
    """amcl: [r]P == O (reference src/keys.rs:182). Infinity passes."""
    return g1_on_curve(a) and g1_mul(a, R) is None

def subgroup_check_g2(a):
    """amcl: [r]P == O (reference src/signature.rs:29). Infinity passes."""
    return g2_on_curve(a) and g2_mul(a, R) is None

def subgroup_check_g2_psi(a):
    """Equivalent endomorphism test psi(P) == [x]P used by the kernels (Scott 2021)."""
    return g2_eq(g2_psi(a), g2_mul(a, X))

# ----------------------------------------------------------------------------- ZCash serialization
ERR_OK, ERR_SIZE, ERR_POINT = 0, 1, 2

def g1_compress(pt):
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    b = bytearray(x.to_bytes(48, 'big'))
    b[0] |= 0x80
    if fp_lex_largest(y):
        b[0] |= 0x20
    return bytes(b)

def g1_serialize_uncompressed(pt):
    if pt is None:
        return bytes([0x40]) + bytes(95)
    return pt[0].to_bytes(48, 'big') + pt[1].to_bytes(48, 'big')

def g1_decompress(b):
    """-> (err, point). Canonical encodings only (reference fuzz/fuzz_targets/fuzz_serde_public_key.rs:5-10)."""
    if len(b) != 48:
        return ERR_SIZE, None
    if not b[0] & 0x80:
        return ERR_SIZE, None              # uncompressed flag with a 48-byte string
    if b[0] & 0x40:
        if b[0] & 0x3F or any(b[1:]):
            return ERR_POINT, None
        return ERR_OK, None
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], 'big')
    if x >= P:
        return ERR_POINT, None
    y = fp_sqrt((x * x * x + B1) % P)
    if y is None:
        return ERR_POINT, None
    if fp_lex_largest(y) != bool(b[0] & 0x20):
        y = (-y) % P
    return ERR_OK, (x, y)

def g1_deserialize_uncompressed(b):
    if len(b) != 96:
        return ERR_SIZE, None
    if b[0] & 0x80:
        return ERR_SIZE, None              # compressed flag with a 96-byte string
    if b[0] & 0x40:
        if b[0] & 0x3F or any(b[1:]):
            return ERR_POINT, None
        return ERR_OK, None
    if b[0] & 0x20:
        return ERR_POINT, None
    x = int.from_bytes(b[:48], 'big')
    y = int.from_bytes(b[48:], 'big')
    if x >= P or y >= P or not g1_on_curve((x, y)):
        return ERR_POINT, None
    return ERR_OK, (x, y)

def g2_compress(pt):
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    x, y = pt
    b = bytearray(x[1].to_bytes(48, 'big') + x[0].to_bytes(48, 'big'))
    b[0] |= 0x80
    if f2_lex_largest(y):
        b[0] |= 0x20
    return bytes(b)

def g2_decompress(b):
    if len(b) != 96:
        return ERR_SIZE, None
    if not b[0] & 0x80:
        return ERR_SIZE, None
    if b[0] & 0x40:
        if b[0] & 0x3F or any(b[1:]):
            return ERR_POINT, None
        return ERR_OK, None
    x1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], 'big')
    x0 = int.from_bytes(b[48:], 'big')
    if x0 >= P or x1 >= P:
        return ERR_POINT, None
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), B2))
    if y is None:
        return ERR_POINT, None
    if f2_lex_largest(y) != bool(b[0] & 0x20):
        y = f2_neg(y)
    return ERR_OK, (x, y)

# ----------------------------------------------------------------------------- hash to G2 (RFC 9380)
def expand_message_xmd(msg, dst, n):
    assert len(dst) <= 255
    ell = (n + 31) // 32
    assert ell <= 255
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + n.to_bytes(2, 'big') + b'\x00' + dst_prime).digest()
    bi = hashlib.sha256(b0 + b'\x01' + dst_prime).digest()
    out = bi
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:n]

def hash_to_field_fp2(msg, dst, count=2):
    u = expand_message_xmd(msg, dst, count * 2 * 64)
    out = []
    for i in range(count):
        e = []
        for j in range(2):
            off = 64 * (j + i * 2)
            e.append(int.from_bytes(u[off:off + 64], 'big') % P)
        out.append((e[0], e[1]))
    return out

SSWU_A = (0, 240)
SSWU_B = (1012, 1012)
SSWU_Z = f2(-2, -1)

def sswu_g2(u):
    """Simplified SWU to E': y^2 = x^3 + A'x + B' (RFC 9380 section 6.6.2, straight-line version)."""
    A, B, Z = SSWU_A, SSWU_B, SSWU_Z
    u2 = f2_sqr(u)
    zu2 = f2_mul(Z, u2)
    den = f2_add(f2_sqr(zu2), zu2)
    if f2_is_zero(den):
        x1 = f2_mul(B, f2_inv(f2_mul(Z, A)))
    else:
        x1 = f2_mul(f2_mul(f2_neg(B), f2_inv(A)), f2_add(F2_ONE, f2_inv(den)))
    gx1 = f2_add(f2_add(f2_mul(f2_sqr(x1), x1), f2_mul(A, x1)), B)
    x2 = f2_mul(zu2, x1)
    gx2 = f2_add(f2_add(f2_mul(f2_sqr(x2), x2), f2_mul(A, x2)), B)
    if f2_is_square(gx1):
        x, y = x1, f2_sqrt(gx1)
    else:
        x, y = x2, f2_sqrt(gx2)
    assert y is not None
    if f2_sgn0(u) != f2_sgn0(y):
        y = f2_neg(y)
    return (x, y)

def _derive_iso3():
    """Derive the 3-isogeny E' -> E from its kernel with Velu's formulas, then fix the
    isomorphism onto y^2 = x^3 + 4(1+i).  Returns (xnum, xden, ynum, yden) coefficient lists
    (ascending powers), normalised like RFC 9380 appendix E.3 (monic denominators)."""
    A, B = SSWU_A, SSWU_B
    # kernel x-coordinate: root of the 3-division polynomial 3x^4 + 6Ax^2 + 12Bx - A^2 in Fp2
    x0 = f2(-6, 6)
    psi3 = f2_sub(f2_add(f2_add(f2_muls(f2_sqr(f2_sqr(x0)), 3), f2_muls(f2_mul(A, f2_sqr(x0)), 6)),
                         f2_muls(f2_mul(B, x0), 12)), f2_sqr(A))
    assert f2_is_zero(psi3)
    # Velu for a kernel {O, Q, -Q}, Q = (x0, y0): t = 2(3x0^2 + A), u = 4 y0^2, w = u + x0 t
    y0sq = f2_add(f2_add(f2_mul(f2_sqr(x0), x0), f2_mul(A, x0)), B)
    t = f2_muls(f2_add(f2_muls(f2_sqr(x0), 3), A), 2)
    u = f2_muls(y0sq, 4)
    # X = x + t/(x-x0) + u/(x-x0)^2 ; Y = y (1 - t/(x-x0)^2 - 2u/(x-x0)^3)
    # image curve: A2 = A - 5t, B2 = B - 7(u + x0 t)
    A2 = f2_sub(A, f2_muls(t, 5))
    Bv = f2_sub(B, f2_muls(f2_add(u, f2_mul(x0, t)), 7))
    assert f2_is_zero(A2)
    # isomorphism (X, Y) -> (X/c^2, Y/c^3) maps y^2=x^3+Bv to y^2 = x^3 + Bv/c^6; need Bv/c^6 = 4(1+i)
    return x0, t, u, Bv

def _poly_mul(a, b):
    r = [F2_ZERO] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            r[i + j] = f2_add(r[i + j], f2_mul(x, y))
    return r

def _poly_add(a, b):
    n = max(len(a), len(b))
    a = a + [F2_ZERO] * (n - len(a)); b = b + [F2_ZERO] * (n - len(b))
    return [f2_add(x, y) for x, y in zip(a, b)]

def _poly_eval(c, x):
    r = F2_ZERO
    for k in reversed(c):
        r = f2_add(f2_mul(r, x), k)
    return r

def _h(s): return int(s, 16)
# RFC 9380 appendix E.3 constants (transcribed; verified below against Velu's formulas).
ISO3_XNUM = [
    (_h("05c759507e8e333ebb5b7a9a47d7ed8532c52d39fd3a042a88b58423c50ae15d5c2638e343d9c71c6238aaaaaaaa97d6"),
     _h("05c759507e8e333ebb5b7a9a47d7ed8532c52d39fd3a042a88b58423c50ae15d5c2638e343d9c71c6238aaaaaaaa97d6")),
    (0, _h("11560bf17baa99bc32126fced787c88f984f87adf7ae0c7f9a208c6b4f20a4181472aaa9cb8d555526a9ffffffffc71a")),
    (_h("11560bf17baa99bc32126fced787c88f984f87adf7ae0c7f9a208c6b4f20a4181472aaa9cb8d555526a9ffffffffc71e"),
     _h("08ab05f8bdd54cde190937e76bc3e447cc27c3d6fbd7063fcd104635a790520c0a395554e5c6aaaa9354ffffffffe38d")),
    (_h("171d6541fa38ccfaed6dea691f5fb614cb14b4e7f4e810aa22d6108f142b85757098e38d0f671c7188e2aaaaaaaa5ed1"), 0),
]
ISO3_XDEN = [
    (0, _h("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaa63")),
    (0xc, _h("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaa9f")),
    F2_ONE,
]
ISO3_YNUM = [
    (_h("1530477c7ab4113b59a4c18b076d11930f7da5d4a07f649bf54439d87d27e500fc8c25ebf8c92f6812cfc71c71c6d706"),
     _h("1530477c7ab4113b59a4c18b076d11930f7da5d4a07f649bf54439d87d27e500fc8c25ebf8c92f6812cfc71c71c6d706")),
    (0, _h("05c759507e8e333ebb5b7a9a47d7ed8532c52d39fd3a042a88b58423c50ae15d5c2638e343d9c71c6238aaaaaaaa97be")),
    (_h("11560bf17baa99bc32126fced787c88f984f87adf7ae0c7f9a208c6b4f20a4181472aaa9cb8d555526a9ffffffffc71c"),
     _h("08ab05f8bdd54cde190937e76bc3e447cc27c3d6fbd7063fcd104635a790520c0a395554e5c6aaaa9354ffffffffe38f")),
    (_h("124c9ad43b6cf79bfbf7043de3811ad0761b0f37a1e26286b0e977c69aa274524e79097a56dc4bd9e1b371c71c718b10"), 0),
]
ISO3_YDEN = [
    (_h("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffa8fb"),
     _h("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffa8fb")),
    (0, _h("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffa9d3")),
    (0x12, _h("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaa99")),
    F2_ONE,
]

def iso3_g2(pt):
    """3-isogeny E' -> E (RFC 9380 appendix E.3)."""
    x, y = pt
    xn = _poly_eval(ISO3_XNUM, x); xd = _poly_eval(ISO3_XDEN, x)
    yn = _poly_eval(ISO3_YNUM, x); yd = _poly_eval(ISO3_YDEN, x)
    if f2_is_zero(xd) or f2_is_zero(yd):
        return None
    return (f2_mul(xn, f2_inv(xd)), f2_mul(y, f2_mul(yn, f2_inv(yd))))

H_EFF_G2 = 0xbc69f08f2ee75b3584c6a0ea91b352888e2a8e9145ad7689986ff031508ffe1329c2f178731db956d82bf015d1212b02ec0ec69d7477c1ae954cbc06689f6a359894c0adebbf6b4e8020005aaa95551

def clear_cofactor_g2(pt):
    """Budroni-Pintore (RFC 9380 appendix G.3): [x^2-x-1]P + [x-1]psi(P) + psi^2(2P)."""
    t1 = g2_mul(pt, X)                    # [x]P
    t2 = g2_psi(pt)                       # psi(P)
    t3 = g2_psi(g2_psi(g2_add(pt, pt)))   # psi^2(2P)
    t3 = g2_add(t3, g2_neg(t2))           # psi^2(2P) - psi(P)
    t2 = g2_add(t1, t2)                   # [x]P + psi(P)
    t2 = g2_mul(t2, X)                    # [x^2]P + [x]psi(P)
    t3 = g2_add(t3, t2)
    t3 = g2_add(t3, g2_neg(t1))
    return g2_add(t3, g2_neg(pt))

def hash_to_curve_g2(msg, dst=DST_POP):
    """reference src/amcl_utils.rs:33-35."""
    u0, u1 = hash_to_field_fp2(msg, dst, 2)
    q0 = iso3_g2(sswu_g2(u0))
    q1 = iso3_g2(sswu_g2(u1))
    return clear_cofactor_g2(g2_add(q0, q1))

# ----------------------------------------------------------------------------- pairing
def _line(T, Q2, Pt):
    """Line through twist points T and Q2 (tangent if equal) evaluated at the G1 point Pt,
    scaled by w^3 (a factor in a proper subfield, killed by the final exponentiation):
       l = (lam*xT - yT) + (-lam*xP) w^2 + yP w^3."""
    xT, yT = T
    if g2_eq(T, Q2):
        lam = f2_mul(f2_muls(f2_sqr(xT), 3), f2_inv(f2_muls(yT, 2)))
    else:
        lam = f2_mul(f2_sub(Q2[1], yT), f2_inv(f2_sub(Q2[0], xT)))
    xP, yP = Pt
    c = [F2_ZERO] * 6
    c[0] = f2_sub(f2_mul(lam, xT), yT)
    c[2] = f2_muls(f2_neg(lam), xP)
    c[3] = (yP % P, 0)
    return c

def miller_loop(pairs):
    """prod_i f_{|x|,Q_i}(P_i), conjugated because x < 0. pairs = [(Q in G2 affine, P in G1 affine)].
    Pairs with an infinite member contribute 1."""
    pairs = [(q, p) for q, p in pairs if q is not None and p is not None]
    f = F12_ONE
    Ts = [q for q, _ in pairs]
    bits = bin(X_ABS)[3:]
    for bit in bits:
        f = f12_sqr(f)
        for k, (q, p) in enumerate(pairs):
            f = f12_mul(f, _line(Ts[k], Ts[k], p))
            Ts[k] = g2_add(Ts[k], Ts[k])
        if bit == '1':
            for k, (q, p) in enumerate(pairs):
                f = f12_mul(f, _line(Ts[k], q, p))
                Ts[k] = g2_add(Ts[k], q)
    return f12_conj(f)

FINAL_EXP = (P**12 - 1) // R

def final_exp(f):
    # easy part by frobenius/inversion, hard part by the literal exponent
    t = f12_mul(f12_conj(f), f12_inv(f))            # f^(p^6-1)
    t = f12_mul(f12_frob(f12_frob(t)), t)           # ^(p^2+1)
    return f12_pow(t, (P**4 - P**2 + 1) // R)

def pairing_product_is_one(pairs):
    return f12_is_one(final_exp(miller_loop(pairs)))

# ----------------------------------------------------------------------------- scheme layer
def sk_to_pk(sk):
    return g1_mul(G1, sk)

def sign(msg, sk):
    """reference src/signature.rs:17-21."""
    return g2_mul(hash_to_curve_g2(msg), sk)

def verify(sig, msg, pk):
    """reference src/signature.rs:27-40 (no pk infinity / subgroup check)."""
    if not subgroup_check_g2(sig):
        return False
    h = hash_to_curve_g2(msg)
    return pairing_product_is_one([(sig, g1_neg(G1)), (h, pk)])

def aggregate_pks(pks):
    acc = None
    for pk in pks:
        acc = g1_add(acc, pk)
    return acc

def fast_aggregate_verify(sig, msg, pks):
    """reference src/aggregates.rs:177-215."""
    if len(pks) == 0:
        return False
    if not subgroup_check_g2(sig):
        return False
    apk = aggregate_pks(pks)
    if apk is None:
        return False
    h = hash_to_curve_g2(msg)
    return pairing_product_is_one([(sig, g1_neg(G1)), (h, apk)])

def aggregate_verify(sig, msgs, pks):
    """reference src/aggregates.rs:130-170."""
    if len(msgs) != len(pks) or len(pks) == 0:
        return False
    if not subgroup_check_g2(sig):
        return False
    pairs = [(hash_to_curve_g2(m), pk) for m, pk in zip(msgs, pks)]
    pairs.append((sig, g1_neg(G1)))
    return pairing_product_is_one(pairs)

def verify_multiple(sets, rands):
    """reference src/aggregates.rs:261-316. sets = [(sig, apk, msg)], rands = nonzero 63-bit ints."""
    acc_sig = None
    pairs = []
    for (sig, apk, msg), rnd in zip(sets, rands):
        if not subgroup_check_g2(sig):
            return False
        h = hash_to_curve_g2(msg)
        pairs.append((h, g1_mul(apk, rnd)))
        acc_sig = g2_add(acc_sig, g2_mul(sig, rnd))
    pairs.append((acc_sig, g1_neg(G1)))
    return pairing_product_is_one(pairs)
