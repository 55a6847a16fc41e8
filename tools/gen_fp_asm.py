#!/usr/bin/env python3
"""Generate milagro_bls_amd/csrc/mbls_fp_asm.inc: the hand-scheduled gfx950 multiplication routines.

* fp_mul_body      -- one Montgomery product in the regular calling convention (a v[0:11], b v[12:23] -> v[0:11]), one asm
                      statement: a call moves no arguments and hipcc cannot pad the stream (it pads every inline-asm
                      boundary). Product scanning over 12 x 32-bit limbs, v_mad_u64_u32 + v_addc_co_u32 per product (Chain).
* fp2_mul_body, fp2_sqr_body, fp2_mulfp_body -- Fp2 product / square / Fp2 x Fp with a private calling convention (operands in
                      fixed blocks of v0..v47, reached by s_swappc). Sum-of-products scans with one Montgomery reduction per
                      output coefficient, on 14 digits of 28 bits so that a multiply-accumulate is a single v_mad_u64_u32
                      (Chain28), two independent scans interleaved instruction by instruction.
Run:  python3 tools/gen_fp_asm.py   (output is committed; tests/test_asm_sim_cpu.py interprets every routine on the CPU and
checks that the committed file is up to date)
"""
import os
import re
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
NP0 = (-pow(P, -1, 1 << 32)) % (1 << 32)
PL = [(P >> (32 * i)) & 0xFFFFFFFF for i in range(12)]
# scalar registers that the calling convention leaves to the callee (s40-s47, s56-s63): no save/restore code is generated
SP = lambda i: "s%d" % (40 + i if i < 8 else 56 + i - 8)    # modulus limbs
SNP = "s60"


class Chain:
    """One product-scanning Montgomery chain: sum of the listed operand products, one reduction, result (< 2p before the final
    conditional subtraction) left in `out`. The Montgomery quotients live in the `out` registers too (m[j] is dead after column
    j+11, out[j] is written in column j+12). The 96-bit column accumulator alternates between two aligned register pairs
    r[0:1] / r[2:3] with the third word in the other pair's high register, so a column boundary costs one move (two when a
    result limb is emitted) and the third word is initialised by the first carry of the column instead of a move."""

    def __init__(self, pairs, out, r, carry):
        self.pairs, self.out, self.r, self.carry = pairs, out, r, carry

    def stream(self):
        r, cy, out = self.r, self.carry, self.out
        S = []
        for k in range(23):
            lo, mid = (r, r + 1) if k % 2 == 0 else (r + 2, r + 3)
            hi = r + 3 if k % 2 == 0 else r + 1
            nlo = r + 2 if k % 2 == 0 else r
            acc = "v[%d:%d]" % (lo, mid)
            macs = []
            for i in range(max(0, k - 11), min(k, 11) + 1):
                for (X, Y) in self.pairs:
                    macs.append((X(i), Y(k - i)))
            if k < 12:
                for i in range(0, k):
                    macs.append((SP(k - i), out(i)))
            else:
                for i in range(k - 11, 12):
                    macs.append((SP(k - i), out(i)))
            first = True
            for (x, y) in macs:
                src2 = "0" if (k == 0 and first) else acc
                S.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, cy, x, y, src2))
                S.append("v_addc_co_u32_e64 v%d, %s, 0, %s, %s" % (hi, cy, "0" if first else "v%d" % hi, cy))
                first = False
            if k < 12:
                S.append("v_mul_lo_u32 %s, %s, v%d" % (out(k), SNP, lo))
                S.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (acc, cy, SP(0), out(k), acc))
                S.append("v_addc_co_u32_e64 v%d, %s, 0, v%d, %s" % (hi, cy, hi, cy))
                S.append("v_mov_b32_e64 v%d, v%d" % (nlo, mid))
            elif k < 22:
                S.append("v_mov_b32_e64 %s, v%d" % (out(k - 12), lo))
                S.append("v_mov_b32_e64 v%d, v%d" % (nlo, mid))
            else:
                # the result is < 2p < 2^384: the third word is zero, column 23 is the middle word
                S.append("v_mov_b32_e64 %s, v%d" % (out(10), lo))
                S.append("v_mov_b32_e64 %s, v%d" % (out(11), mid))
        return S


def cond_sub(out, diff, tmp):
    """out = out - p if that does not borrow (one conditional subtraction; p goes through a VGPR because an SGPR source next to
    the VCC carry-in would be two constant-bus reads)"""
    S = ["v_mov_b32_e32 %s, %s" % (tmp, SP(0)), "v_sub_co_u32_e32 %s, vcc, %s, %s" % (diff(0), out(0), tmp)]
    for i in range(1, 12):
        S.append("v_mov_b32_e32 %s, %s" % (tmp, SP(i)))
        S.append("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (diff(i), out(i), tmp))
    for i in range(12):
        S.append("v_cndmask_b32_e32 %s, %s, %s, vcc" % (out(i), diff(i), out(i)))     # borrow ? t : t - p
    return S


def zip2(sa, sb):
    assert len(sa) == len(sb)
    # every instruction of the interleaved scans is 8 bytes long; a lone wave fetches them ~20 % slower when they sit at
    # 4 mod 8 (measured: 9318 vs 7785 clocks per Fp2 multiplication), so the stream starts on a cache-line boundary
    L = [".p2align 6"]
    for x, y in zip(sa, sb):
        L += [x, y]
    return L


def load_modulus():
    L = ["s_mov_b32 %s, 0x%08x" % (SP(i), PL[i]) for i in range(12)]
    L.append("s_mov_b32 %s, 0x%08x" % (SNP, NP0))
    return L


CARRY_B = "s[62:63]"
VR = lambda base: (lambda i: "v%d" % (base + i))


def fp_mul_body():
    """Single Montgomery product in the regular calling convention (a v[0:11], b v[12:23], result v[0:11]; v24-v39 scratch):
    one product-scanning chain, every instruction 8 bytes long and 8-byte aligned."""
    A, B, M = VR(0), VR(12), VR(24)
    L = load_modulus() + [".p2align 6"] + Chain([(A, B)], M, 36, "vcc").stream()
    # result = M - p if that does not borrow, into the (dead) a registers
    L.append("v_mov_b32_e32 v12, %s" % SP(0))
    L.append("v_sub_co_u32_e32 v0, vcc, v24, v12")
    for i in range(1, 12):
        L.append("v_mov_b32_e32 v12, %s" % SP(i))
        L.append("v_subb_co_u32_e32 v%d, vcc, v%d, v12, vcc" % (i, 24 + i))
    for i in range(12):
        L.append("v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (i, i, 24 + i))     # borrow ? t : t - p
    return L


# ----------------------------------------------------------------------------------------------------------------------
# 28-bit-digit multiplication core. Inside a routine the operands are re-cut into 14 digits of 28 bits, so that a whole column
# of the product scan (up to 28 products of 56/57 bits plus 14 reduction products) accumulates in ONE 64-bit register pair
# without carries: a multiply-accumulate is a single v_mad_u64_u32 instead of v_mad_u64_u32 + v_addc_co_u32. The Montgomery
# radix of 14 digits is 2^392; the first factor of every product is converted as a * 2^8 (digit j = bits [28j-8, 28j+20)),
# so the routine still returns a b 2^-384 and the Montgomery domain of the rest of the library (R = 2^384, 12 x 32-bit
# limbs between routines) is unchanged.
M28 = (1 << 28) - 1
P28 = [(P >> (28 * i)) & M28 for i in range(14)]
NP28 = (-pow(P, -1, 1 << 28)) % (1 << 28)
SP28 = lambda i: "s%d" % (40 + i if i < 8 else 56 + i - 8)      # 14 digits of p: s40-s47, s56-s61
SNP28, SMASK28 = "s64", "s65"
# digits of 2p, each raised above 2^28 by borrowing from the next one, so that digit-wise subtraction of a value < 2p never
# underflows: NEG2P[j] - b[j] are the (29-bit) digits of 2p - b
_d2p = [((2 * P) >> (28 * i)) & M28 for i in range(14)]
NEG2P = [_d2p[0] + (1 << 28)] + [_d2p[i] + (1 << 28) - 1 for i in range(1, 13)] + [_d2p[13] - 1]
assert sum(d << (28 * i) for i, d in enumerate(NEG2P)) == 2 * P and NEG2P[13] > (P >> 364)     # fine for any b <= p


def load_modulus28():
    L = ["s_mov_b32 %s, 0x%08x" % (SP28(i), P28[i]) for i in range(14)]
    L += ["s_mov_b32 %s, 0x%08x" % (SNP28, NP28), "s_mov_b32 %s, 0x%08x" % (SMASK28, M28)]
    return L


def conv28(dst, src, aform):
    """14 digits of 28 bits from 12 x 32-bit words; aform: digits of src * 2^8"""
    L = []
    for j in range(14):
        o = 28 * j - (8 if aform else 0)
        d = dst(j)
        if o < 0:
            L += ["v_lshlrev_b32_e64 %s, 8, %s" % (d, src(0)), "v_and_b32_e64 %s, %s, %s" % (d, d, SMASK28)]
            continue
        q, r = o >> 5, o & 31
        if o + 28 > 384:
            assert q == 11
            L.append("v_lshrrev_b32_e64 %s, %d, %s" % (d, r, src(11)))
        elif r == 0:
            L.append("v_and_b32_e64 %s, %s, %s" % (d, src(q), SMASK28))
        elif r + 28 == 32:
            L.append("v_lshrrev_b32_e64 %s, %d, %s" % (d, r, src(q)))
        elif r + 28 < 32:
            L.append("v_bfe_u32 %s, %s, %d, 28" % (d, src(q), r))
        else:
            L += ["v_alignbit_b32 %s, %s, %s, %d" % (d, src(q + 1), src(q), r), "v_and_b32_e64 %s, %s, %s" % (d, d, SMASK28)]
    return L


def to32(dst, t):
    """12 x 32-bit words from 14 digits of exactly 28 bits"""
    L = []
    for q in range(12):
        j, off = (32 * q) // 28, (32 * q) % 28
        if off == 0:
            L.append("v_lshl_or_b32 %s, %s, 28, %s" % (dst(q), t(j + 1), t(j)))
        else:
            L.append("v_lshrrev_b32_e64 %s, %d, %s" % (dst(q), off, t(j)))
            L.append("v_lshl_or_b32 %s, %s, %d, %s" % (dst(q), t(j + 1), 28 - off, dst(q)))
            assert 56 - off >= 32
    return L


def digit_column_scan(terms, out, acc, acc2, cy="vcc", last_masked=False):
    """product scan with explicit per-column product lists; products alternate between two 64-bit accumulators (a single
    dependent v_mad_u64_u32 chain stalls); digits of the Montgomery quotient / result in `out`"""
    A, lo = "v[%d:%d]" % (acc, acc + 1), "v%d" % acc
    A2 = "v[%d:%d]" % (acc2, acc2 + 1) if acc2 is not None else None
    S, first = [], True
    for k in range(28):
        macs = list(terms(k))
        if k < 14:
            macs += [(SP28(k - i), out(i)) for i in range(k)]
        else:
            macs += [(SP28(k - i), out(i)) for i in range(k - 13, 14)]
        used2 = False
        for n, (x, y) in enumerate(macs):
            if A2 is not None and n % 2 == 1:
                S.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (A2, cy, x, y, A2 if used2 else "0")); used2 = True
            else:
                S.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (A, cy, x, y, "0" if first else A)); first = False
        if used2:
            S.append("v_lshl_add_u64 %s, %s, 0, %s" % (A, A2, A))
        if k < 14:
            S += ["v_mul_lo_u32 %s, %s, %s" % (out(k), lo, SNP28), "v_and_b32_e64 %s, %s, %s" % (out(k), out(k), SMASK28),
                  "v_mad_u64_u32 %s, %s, %s, %s, %s" % (A, cy, SP28(0), out(k), A), "v_lshrrev_b64 %s, 28, %s" % (A, A)]
        elif k < 27:
            S += ["v_and_b32_e64 %s, %s, %s" % (out(k - 14), lo, SMASK28), "v_lshrrev_b64 %s, 28, %s" % (A, A)]
        elif last_masked:
            S.append("v_and_b32_e64 %s, %s, %s" % (out(13), lo, SMASK28))
        else:
            S.append("v_mov_b32_e64 %s, %s" % (out(13), lo))          # top digit: whatever is left (the value is < 2^392 + 2p)
    return S


class Chain28:
    """product scan over 28-bit digits: sum of the listed operand products (first factors in a*2^8 form), Montgomery-reduced by
    2^392; the quotient digits and then the result digits live in `out` (14 registers); acc: an aligned register pair;
    acc2: optional second pair, the products of a column then alternate between two accumulators"""

    def __init__(self, pairs, out, acc, carry, acc2=None):
        self.pairs, self.out, self.acc, self.carry, self.acc2 = pairs, out, acc, carry, acc2

    def stream(self):
        def terms(k):
            for i in range(max(0, k - 13), min(k, 13) + 1):
                for (X, Y) in self.pairs:
                    yield (X(i), Y(k - i))
        return digit_column_scan(terms, self.out, self.acc, self.acc2, self.carry, last_masked=True)


def cond_sub32(out, diff, tmp):
    """out = out - p if that does not borrow; p as literals (the SGPRs hold the 28-bit digits)"""
    S = ["v_mov_b32_e32 %s, 0x%08x" % (tmp, PL[0]), "v_sub_co_u32_e64 %s, vcc, %s, %s" % (diff(0), out(0), tmp)]
    for i in range(1, 12):
        S.append("v_mov_b32_e32 %s, 0x%08x" % (tmp, PL[i]))
        S.append("v_subb_co_u32_e64 %s, vcc, %s, %s, vcc" % (diff(i), out(i), tmp))
    for i in range(12):
        S.append("v_cndmask_b32_e64 %s, %s, %s, vcc" % (out(i), diff(i), out(i)))     # borrow ? t : t - p
    return S


ACC2A, ACC2B = None, None      # a second accumulator per scan was measured slower (6802 vs 6604 clocks per Fp2 multiplication)


def fp2_mul_body():
    """Fp2 product on the 28-bit core. a0 v[0:11], a1 v[12:23], b0 v[24:35], b1 v[36:47] (all OVERWRITTEN);
    c0 -> v[48:59], c1 -> v[60:71]; scratch up to v117."""
    A0w, A1w, B0w, B1w = VR(0), VR(12), VR(24), VR(36)
    A0, A1, B0, B1, NB = VR(48), VR(62), VR(76), VR(90), VR(104)
    TA, TB = VR(0), VR(14)
    C0, C1, DF = VR(48), VR(60), VR(72)
    L = load_modulus28()
    L += conv28(A0, A0w, True) + conv28(A1, A1w, True) + conv28(B0, B0w, False) + conv28(B1, B1w, False)
    L += ["v_sub_u32_e32 %s, 0x%08x, %s" % (NB(j), NEG2P[j], B1(j)) for j in range(14)]        # digits of 2p - b1
    L += zip2(Chain28([(A0, B0), (A1, NB)], TA, 28, "vcc", ACC2A).stream(), Chain28([(A0, B1), (A1, B0)], TB, 30, CARRY_B, ACC2B).stream())
    L += to32(C0, TA) + to32(C1, TB)
    L += cond_sub32(C0, DF, "v84") + cond_sub32(C1, DF, "v84")
    return L


def fp2_sqr_body():
    """Fp2 square on the 28-bit core: c0 = (a0 + a1)(a0 - a1 + p), c1 = a0 (2 a1). a0 v[0:11], a1 v[12:23] (OVERWRITTEN);
    c0 -> v[24:35], c1 -> v[36:47]; scratch up to v103."""
    A0w, A1w, Sw, Dw = VR(0), VR(12), VR(24), VR(36)
    SA, DB, A0A, A1D = VR(48), VR(62), VR(76), VR(90)
    TA, TB = VR(0), VR(14)
    C0, C1, DF = VR(24), VR(36), VR(48)
    tmp = "v104"
    L = load_modulus28()
    L.append("v_add_co_u32_e64 %s, vcc, %s, %s" % (Sw(0), A0w(0), A1w(0)))
    for i in range(1, 12):
        L.append("v_addc_co_u32_e64 %s, vcc, %s, %s, vcc" % (Sw(i), A0w(i), A1w(i)))
    L.append("v_mov_b32_e32 %s, 0x%08x" % (tmp, PL[0]))
    L.append("v_add_co_u32_e64 %s, vcc, %s, %s" % (Dw(0), A0w(0), tmp))
    for i in range(1, 12):
        L.append("v_mov_b32_e32 %s, 0x%08x" % (tmp, PL[i]))
        L.append("v_addc_co_u32_e64 %s, vcc, %s, %s, vcc" % (Dw(i), A0w(i), tmp))
    L.append("v_sub_co_u32_e64 %s, vcc, %s, %s" % (Dw(0), Dw(0), A1w(0)))
    for i in range(1, 12):
        L.append("v_subb_co_u32_e64 %s, vcc, %s, %s, vcc" % (Dw(i), Dw(i), A1w(i)))
    L += conv28(SA, Sw, True) + conv28(DB, Dw, False) + conv28(A0A, A0w, True) + conv28(A1D, A1w, False)
    L += ["v_lshlrev_b32_e64 %s, 1, %s" % (A1D(j), A1D(j)) for j in range(14)]                   # digits of 2 a1 (29 bits)
    L += zip2(Chain28([(SA, DB)], TA, 28, "vcc", ACC2A).stream(), Chain28([(A0A, A1D)], TB, 30, CARRY_B, ACC2B).stream())
    L += to32(C1, TB) + to32(C0, TA)          # in this order: c0's registers overlap the digits of the second chain
    L += cond_sub32(C0, DF, tmp) + cond_sub32(C1, DF, tmp)
    return L


def fp2_mulfp_body():
    """c0 = a0 s, c1 = a1 s on the 28-bit core. a0 v[0:11], a1 v[12:23], s v[24:35] (OVERWRITTEN); c0 -> v[36:47],
    c1 -> v[48:59]; scratch up to v89."""
    A0w, A1w, Sw = VR(0), VR(12), VR(24)
    A0, A1, SB = VR(48), VR(62), VR(76)
    TA, TB = VR(0), VR(14)
    C0, C1, DF = VR(36), VR(48), VR(60)
    L = load_modulus28()
    L += conv28(A0, A0w, True) + conv28(A1, A1w, True) + conv28(SB, Sw, False)
    L += zip2(Chain28([(A0, SB)], TA, 28, "vcc", ACC2A).stream(), Chain28([(A1, SB)], TB, 30, CARRY_B, ACC2B).stream())
    L += to32(C0, TA) + to32(C1, TB)
    L += cond_sub32(C0, DF, "v72") + cond_sub32(C1, DF, "v72")
    return L


# ----------------------------------------------------------------------------------------------------------------------
# Fixed-exponent Fp exponentiation entirely on 28-bit digits (domain 2^392 inside the routine): the square roots and the
# inversion are ~480 dependent Fp multiplications each, 6-7 per verified item. Inside, a value is 14 unsaturated digits and
# never converted: a squaring uses the symmetry (cross products once, with a doubled digit vector -- exact because digits
# carry no carries), 105 + 196 multiply-accumulates instead of 2 x 300 on saturated limbs.
def sqr_digits(X, D, out, acc, acc2):
    """out = X^2 / 2^392 on digit vectors (D: scratch for the doubled digits)"""
    L = ["v_lshlrev_b32_e64 %s, 1, %s" % (D(j), X(j)) for j in range(14)]

    def terms(k):
        for i in range(max(0, k - 13), min(k, 13) + 1):
            j = k - i
            if i < j:
                yield (X(i), D(j))
            elif i == j:
                yield (X(i), X(i))
    return L + digit_column_scan(terms, out, acc, acc2)


def mul_digits(X, Y, out, acc, acc2):
    def terms(k):
        for i in range(max(0, k - 13), min(k, 13) + 1):
            yield (X(i), Y(k - i))
    return digit_column_scan(terms, out, acc, acc2)


POW_X, POW_Y, POW_D, POW_B = VR(0), VR(14), VR(28), VR(42)      # value (ping), value (pong), doubled digits, table entry
POW_ACC, POW_ACC2 = 56, 58
R384_DIGITS = [(((1 << 384) % P) >> (28 * i)) & M28 for i in range(14)]


def pow_subroutines():
    """the four leaf routines of an exponentiation: the value ping-pongs between X and Y (a squaring or a product by the table
    entry B reads one and writes the other, so nothing is ever moved)"""
    return {
        "mbls_pow_sqr_xy_asm_fn": [".p2align 6"] + sqr_digits(POW_X, POW_D, POW_Y, POW_ACC, POW_ACC2),
        "mbls_pow_sqr_yx_asm_fn": [".p2align 6"] + sqr_digits(POW_Y, POW_D, POW_X, POW_ACC, POW_ACC2),
        "mbls_pow_mul_xb_y_asm_fn": [".p2align 6"] + mul_digits(POW_X, POW_B, POW_Y, POW_ACC, POW_ACC2),
        "mbls_pow_mul_yb_x_asm_fn": [".p2align 6"] + mul_digits(POW_Y, POW_B, POW_X, POW_ACC, POW_ACC2),
    }


POW_WINDOW = 5                     # sliding window: the odd powers a^1 .. a^31 in a0..a223


def pow_schedule(e, w=POW_WINDOW):
    """left-to-right sliding-window schedule of a^e: a list of ('sqr', n) and ('mul', odd value); the first entry is ('load', v)"""
    bits = bin(e)[2:]
    ops, i = [], 0
    while i < len(bits):
        if bits[i] == "0":
            ops.append(("sqr", 1)); i += 1
            continue
        j = min(i + w, len(bits))
        while bits[j - 1] == "0":
            j -= 1
        v = int(bits[i:j], 2)
        ops.append(("load", v) if not ops else ("win", j - i, v))
        i = j
    out = []
    for op in ops:                                     # merge runs of squarings
        if op[0] == "win":
            out += [("sqr", op[1]), ("mul", op[2])]
        else:
            out.append(op)
    merged = []
    for op in out:
        if op[0] == "sqr" and merged and merged[-1][0] == "sqr":
            merged[-1] = ("sqr", merged[-1][1] + op[1])
        else:
            merged.append(op)
    return merged


def pow_body(e, window=POW_WINDOW):
    """a^e for a in v[0:11] (Montgomery form, R = 2^384), result in v[0:11]; sliding `window`-bit windows over the odd powers a, a^3,
    .. kept in AGPRs as digit vectors (window 5: a0..a223; window 4: a0..a111, for kernels that run two waves per SIMD).
    Pseudo-instruction CALL name = s_getpc/s_add/s_addc/s_swappc."""
    L = ["s_mov_b64 s[36:37], s[30:31]"] + load_modulus28()
    L += conv28(VR(60), VR(0), True)                              # digits of a * 2^8: the value in the 2^392 domain
    L += ["v_mov_b32_e64 %s, %s" % (POW_X(j), "v%d" % (60 + j)) for j in range(14)]
    tab = lambda n, j: "a%d" % (14 * ((n - 1) // 2) + j)
    REG = {"X": POW_X, "Y": POW_Y}
    other = {"X": "Y", "Y": "X"}
    sqr = {"X": "CALL mbls_pow_sqr_xy_asm_fn", "Y": "CALL mbls_pow_sqr_yx_asm_fn"}
    mul = {"X": "CALL mbls_pow_mul_xb_y_asm_fn", "Y": "CALL mbls_pow_mul_yb_x_asm_fn"}
    L += ["v_accvgpr_write_b32 %s, %s" % (tab(1, j), POW_X(j)) for j in range(14)]
    L.append(sqr["X"])                                            # a^2 -> Y -> B; the odd powers by repeated products with it
    L += ["v_mov_b32_e64 %s, %s" % (POW_B(j), POW_Y(j)) for j in range(14)]
    loc = "X"
    for n in range(3, 1 << window, 2):
        L.append(mul[loc]); loc = other[loc]
        L += ["v_accvgpr_write_b32 %s, %s" % (tab(n, j), REG[loc](j)) for j in range(14)]
    for op in pow_schedule(e, window):
        if op[0] == "load":
            loc = "X"
            L += ["v_accvgpr_read_b32 %s, %s" % (POW_X(j), tab(op[1], j)) for j in range(14)]
        elif op[0] == "sqr":
            for _ in range(op[1]):
                L.append(sqr[loc]); loc = other[loc]
        else:
            L += ["v_accvgpr_read_b32 %s, %s" % (POW_B(j), tab(op[1], j)) for j in range(14)]
            L.append(mul[loc]); loc = other[loc]
    # leave the 2^392 domain: value * (2^384 mod p) / 2^392, then 12 x 32-bit words and the final conditional subtraction
    L += ["v_mov_b32_e32 %s, 0x%08x" % (POW_B(j), R384_DIGITS[j]) for j in range(14)]
    L.append(mul[loc]); loc = other[loc]
    L += to32(VR(60), REG[loc])
    L += cond_sub32(VR(60), VR(72), "v84")
    L += ["v_mov_b32_e64 v%d, v%d" % (j, 60 + j) for j in range(12)]
    L.append("s_mov_b64 s[30:31], s[36:37]")
    return L


def expand_pow_calls(lines):
    """the leaf routines use s40.. (modulus digits) themselves, so the call address goes through s[66:67]"""
    out = []
    for l in lines:
        if l.startswith("CALL "):
            sym = l.split()[1]
            out += ["s_getpc_b64 s[66:67]", "s_add_u32 s66, s66, %s@rel32@lo+4" % sym, "s_addc_u32 s67, s67, %s@rel32@hi+12" % sym,
                    "s_swappc_b64 s[30:31], s[66:67]"]
        else:
            out.append(l)
    return out


EXP_PM3D4 = (P - 3) // 4
EXP_PM2 = P - 2


def fp_mulpair_body():
    """two independent Fp products, interleaved: c0 = a0 b0, c1 = a1 b1. a0 v[0:11], b0 v[12:23], a1 v[24:35], b1 v[36:47]
    (OVERWRITTEN); c0 -> v[48:59], c1 -> v[60:71]; scratch up to v103. For code that has independent multiplications in
    pairs (G1 point addition): a lone Fp product is one dependent chain and issues ~15 % slower per multiply-accumulate."""
    A0w, B0w, A1w, B1w = VR(0), VR(12), VR(24), VR(36)
    A0, B0, A1, B1 = VR(48), VR(62), VR(76), VR(90)
    TA, TB = VR(0), VR(14)
    C0, C1, DF = VR(48), VR(60), VR(72)
    L = load_modulus28()
    L += conv28(A0, A0w, True) + conv28(B0, B0w, False) + conv28(A1, A1w, True) + conv28(B1, B1w, False)
    L += zip2(Chain28([(A0, B0)], TA, 28, "vcc").stream(), Chain28([(A1, B1)], TB, 30, CARRY_B).stream())
    L += to32(C0, TA) + to32(C1, TB)
    L += cond_sub32(C0, DF, "v84") + cond_sub32(C1, DF, "v84")
    return L


# ----------------------------------------------------------------------------------------------------------------------
# Fp inversion by the Bernstein-Yang "safegcd" divsteps (the constant-time variant with delta starting at 1/2, as published for
# secp256k1's modinv32): 30 rounds of 30 divsteps on the low words of f and g give a 2x2 transition matrix each, which is then
# applied to the full f, g (13 signed 30-bit limbs) and, modulo p, to d, e. 900 >= floor((45907 * 381 + 26313) / 19929) = 878
# divsteps end with g = 0, f = +-gcd and d = +-(e0 / x). Starting e at 2^768 mod p turns x R into x^-1 R without a final product.
# About 30 k instructions against 190 k for a^(p-2); every lane does the same work whatever its operand (0 -> 0).
GCD_N, GCD_ROUNDS = 13, 30
M30 = (1 << 30) - 1
GF, GG, GD, GE, GM = VR(12), VR(25), VR(38), VR(51), VR(64)          # f, g, d, e, modulus: 13 limbs each
G_U, G_V, G_Q, G_R, G_ZETA, G_FW, G_GW, G_C1, G_C2, G_X, G_Y, G_Z = ["v%d" % i for i in range(12)]
G_MD, G_ME, G_T, G_T2 = "v77", "v78", "v79", "v84"
G_CD, G_CE = "v[80:81]", "v[82:83]"
G_INV30 = "s66"


def limbs30(x):
    return [(x >> (30 * i)) & M30 for i in range(GCD_N - 1)] + [x >> (30 * (GCD_N - 1))]


def gcd_prologue():
    """x (12 words, v0..v11) -> g; f = p; d = 0; e = 2^768 mod p; zeta = -1"""
    L = ["s_mov_b32 %s, 0x%08x" % (G_INV30, pow(P, -1, 1 << 30))]
    for i in range(GCD_N):                                          # limb i = bits [30 i, 30 i + 30)
        o = 30 * i
        q, r = o >> 5, o & 31
        if q >= 12:
            L.append("v_mov_b32_e32 %s, 0" % GG(i))
        elif q == 11 or r + 30 <= 32:
            L.append("v_bfe_u32 %s, v%d, %d, 30" % (GG(i), q, r) if r + 30 <= 32 else "v_lshrrev_b32_e64 %s, %d, v%d" % (GG(i), r, q))
        else:
            L += ["v_alignbit_b32 %s, v%d, v%d, %d" % (GG(i), q + 1, q, r), "v_and_b32_e32 %s, 0x%08x, %s" % (GG(i), M30, GG(i))]
    for i, (m, e) in enumerate(zip(limbs30(P), limbs30(pow(2, 768, P)))):
        L += ["v_mov_b32_e32 %s, 0x%08x" % (GM(i), m), "v_mov_b32_e32 %s, 0x%08x" % (GF(i), m),
              "v_mov_b32_e32 %s, 0" % GD(i), "v_mov_b32_e32 %s, 0x%08x" % (GE(i), e)]
    L.append("v_mov_b32_e32 %s, -1" % G_ZETA)
    return L


def gcd_round_head():
    return ["v_lshl_or_b32 %s, %s, 30, %s" % (G_FW, GF(1), GF(0)), "v_lshl_or_b32 %s, %s, 30, %s" % (G_GW, GG(1), GG(0)),
            "v_mov_b32_e32 %s, 1" % G_U, "v_mov_b32_e32 %s, 0" % G_V, "v_mov_b32_e32 %s, 0" % G_Q, "v_mov_b32_e32 %s, 1" % G_R]


def gcd_divstep():
    L = ["v_ashrrev_i32_e64 %s, 31, %s" % (G_C1, G_ZETA), "v_bfe_i32 %s, %s, 0, 1" % (G_C2, G_GW)]       # zeta < 0; g odd (all-ones masks)
    for dst, src in ((G_X, G_FW), (G_Y, G_U), (G_Z, G_V)):                                               # conditionally negated f, u, v
        L += ["v_xor_b32_e64 %s, %s, %s" % (dst, src, G_C1), "v_sub_u32_e64 %s, %s, %s" % (dst, dst, G_C1)]
    for acc, src in ((G_GW, G_X), (G_Q, G_Y), (G_R, G_Z)):                                               # g, q, r += those, if g is odd
        L += ["v_and_b32_e64 %s, %s, %s" % (G_T, src, G_C2), "v_add_u32_e64 %s, %s, %s" % (acc, acc, G_T)]
    L += ["v_and_b32_e64 %s, %s, %s" % (G_C1, G_C1, G_C2),
          "v_xor_b32_e64 %s, %s, %s" % (G_ZETA, G_ZETA, G_C1), "v_add_u32_e64 %s, -1, %s" % (G_ZETA, G_ZETA)]
    for acc, src in ((G_FW, G_GW), (G_U, G_Q), (G_V, G_R)):                                              # f, u, v += g, q, r on a swap
        L += ["v_and_b32_e64 %s, %s, %s" % (G_T, src, G_C1), "v_add_u32_e64 %s, %s, %s" % (acc, acc, G_T)]
    L += ["v_lshrrev_b32_e64 %s, 1, %s" % (G_GW, G_GW), "v_lshlrev_b32_e64 %s, 1, %s" % (G_U, G_U), "v_lshlrev_b32_e64 %s, 1, %s" % (G_V, G_V)]
    return L


def gcd_round_tail():
    """apply the matrix (u v; q r) / 2^30 to (d, e) modulo p and to (f, g) exactly"""
    mad = lambda acc, a, b, first=False: "v_mad_i64_i32 %s, vcc, %s, %s, %s" % (acc, a, b, "0" if first else acc)
    lo = lambda pair: "v%s" % pair[2:pair.index(":")]
    L = ["v_ashrrev_i32_e64 %s, 31, %s" % (G_T, GD(12)), "v_ashrrev_i32_e64 %s, 31, %s" % (G_T2, GE(12))]
    L += ["v_and_b32_e64 %s, %s, %s" % (G_MD, G_U, G_T), "v_and_b32_e64 %s, %s, %s" % (G_C1, G_V, G_T2), "v_add_u32_e64 %s, %s, %s" % (G_MD, G_MD, G_C1),
          "v_and_b32_e64 %s, %s, %s" % (G_ME, G_Q, G_T), "v_and_b32_e64 %s, %s, %s" % (G_C1, G_R, G_T2), "v_add_u32_e64 %s, %s, %s" % (G_ME, G_ME, G_C1)]
    L += [mad(G_CD, G_U, GD(0), True), mad(G_CD, G_V, GE(0)), mad(G_CE, G_Q, GD(0), True), mad(G_CE, G_R, GE(0))]
    for m, c in ((G_MD, G_CD), (G_ME, G_CE)):                       # the multiple of p that makes the low 30 bits vanish
        L += ["v_mul_lo_u32 %s, %s, %s" % (G_T, lo(c), G_INV30), "v_add_u32_e64 %s, %s, %s" % (G_T, G_T, m),
              "v_and_b32_e32 %s, 0x%08x, %s" % (G_T, M30, G_T), "v_sub_u32_e64 %s, %s, %s" % (m, m, G_T)]
    L += [mad(G_CD, GM(0), G_MD), mad(G_CE, GM(0), G_ME), "v_ashrrev_i64 %s, 30, %s" % (G_CD, G_CD), "v_ashrrev_i64 %s, 30, %s" % (G_CE, G_CE)]
    for i in range(1, GCD_N):
        L += [mad(G_CD, G_U, GD(i)), mad(G_CE, G_Q, GD(i)), mad(G_CD, G_V, GE(i)), mad(G_CE, G_R, GE(i)), mad(G_CD, GM(i), G_MD), mad(G_CE, GM(i), G_ME)]
        L += ["v_and_b32_e32 %s, 0x%08x, %s" % (GD(i - 1), M30, lo(G_CD)), "v_and_b32_e32 %s, 0x%08x, %s" % (GE(i - 1), M30, lo(G_CE)),
              "v_ashrrev_i64 %s, 30, %s" % (G_CD, G_CD), "v_ashrrev_i64 %s, 30, %s" % (G_CE, G_CE)]
    L += ["v_mov_b32_e32 %s, %s" % (GD(12), lo(G_CD)), "v_mov_b32_e32 %s, %s" % (GE(12), lo(G_CE))]
    L += [mad(G_CD, G_U, GF(0), True), mad(G_CE, G_Q, GF(0), True), mad(G_CD, G_V, GG(0)), mad(G_CE, G_R, GG(0)),
          "v_ashrrev_i64 %s, 30, %s" % (G_CD, G_CD), "v_ashrrev_i64 %s, 30, %s" % (G_CE, G_CE)]
    for i in range(1, GCD_N):
        L += [mad(G_CD, G_U, GF(i)), mad(G_CE, G_Q, GF(i)), mad(G_CD, G_V, GG(i)), mad(G_CE, G_R, GG(i))]
        L += ["v_and_b32_e32 %s, 0x%08x, %s" % (GF(i - 1), M30, lo(G_CD)), "v_and_b32_e32 %s, 0x%08x, %s" % (GG(i - 1), M30, lo(G_CE)),
              "v_ashrrev_i64 %s, 30, %s" % (G_CD, G_CD), "v_ashrrev_i64 %s, 30, %s" % (G_CE, G_CE)]
    L += ["v_mov_b32_e32 %s, %s" % (GF(12), lo(G_CD)), "v_mov_b32_e32 %s, %s" % (GG(12), lo(G_CE))]
    return L


def gcd_epilogue():
    """d, with the sign of f, into [0, p) -> 12 words in v0..v11"""
    L = []

    def cond_add_p():
        S = ["v_ashrrev_i32_e64 %s, 31, %s" % (G_T, GD(12))]
        for i in range(GCD_N):
            S += ["v_and_b32_e64 %s, %s, %s" % (G_T2, GM(i), G_T), "v_add_u32_e64 %s, %s, %s" % (GD(i), GD(i), G_T2)]
        return S

    def carries():
        S = []
        for i in range(GCD_N - 1):
            S += ["v_ashrrev_i32_e64 %s, 30, %s" % (G_T2, GD(i)), "v_add_u32_e64 %s, %s, %s" % (GD(i + 1), GD(i + 1), G_T2),
                  "v_and_b32_e32 %s, 0x%08x, %s" % (GD(i), M30, GD(i))]
        return S
    L += cond_add_p()
    L.append("v_ashrrev_i32_e64 %s, 31, %s" % (G_T, GF(12)))
    for i in range(GCD_N):
        L += ["v_xor_b32_e64 %s, %s, %s" % (GD(i), GD(i), G_T), "v_sub_u32_e64 %s, %s, %s" % (GD(i), GD(i), G_T)]
    L += carries() + cond_add_p() + carries()
    for q in range(12):                                             # word q = bits [32 q, 32 q + 32)
        j, off = (32 * q) // 30, (32 * q) % 30
        if off == 0:
            L.append("v_lshl_or_b32 v%d, %s, 30, %s" % (q, GD(j + 1), GD(j)))
        else:
            L.append("v_lshrrev_b32_e64 v%d, %d, %s" % (q, off, GD(j)))
            L.append("v_lshl_or_b32 v%d, %s, %d, v%d" % (q, GD(j + 1), 30 - off, q))
            if 60 - off < 32:
                L.append("v_lshl_or_b32 v%d, %s, %d, v%d" % (q, GD(j + 2), 60 - off, q))
    return L


def fp_inv_gcd_pieces():
    return dict(pro=gcd_prologue(), head=gcd_round_head(), step=gcd_divstep(), tail=gcd_round_tail(), epi=gcd_epilogue())


def fp_inv_gcd_body(unrolled=False):
    """1 / a for a in v[0:11] (Montgomery form, R = 2^384, canonical; 0 -> 0), result in v[0:11]. Overwrites v12..v84, vcc, s66, s76,
    s77. unrolled: the flat instruction list (for the simulator); otherwise two counted loops."""
    pc = fp_inv_gcd_pieces()
    if unrolled:
        L = list(pc["pro"])
        for _ in range(GCD_ROUNDS):
            L += pc["head"] + pc["step"] * 30 + pc["tail"]
        return L + pc["epi"]
    L = pc["pro"] + ["s_mov_b32 s76, %d" % GCD_ROUNDS, ".p2align 6", "1:"] + pc["head"] + ["s_mov_b32 s77, 30", "2:"] + pc["step"]
    L += ["s_sub_u32 s77, s77, 1", "s_cmp_lg_u32 s77, 0", "s_cbranch_scc1 2b"] + pc["tail"]
    L += ["s_sub_u32 s76, s76, 1", "s_cmp_lg_u32 s76, 0", "s_cbranch_scc1 1b"] + pc["epi"]
    return L


# MBLS_GEN_E32=1 (experiment, scripts/dbg/ab_gen.sh): plain operations in their 4-byte VOP1 / VOP2 encodings where the operands allow it instead of the 8-byte VOP3
# forms the generators write -- same arithmetic, smaller code
PREFER_E32 = os.environ.get("MBLS_GEN_E32", "0") == "1"


def _is_v(x):
    return re.fullmatch(r"v\d+", x) is not None


def to_e32(line):
    m = re.match(r"^(v_and_b32|v_add_u32|v_sub_u32|v_or_b32|v_xor_b32)_e64 (\S+), (\S+), (\S+)$", line)
    if m:
        op, d, a, b = m.groups(); a = a; d = d.rstrip(","); a = a.rstrip(",")
        if _is_v(b):
            return "%s_e32 %s, %s, %s" % (op, d, a, b)
        if _is_v(a):
            if op == "v_sub_u32":
                return "v_subrev_u32_e32 %s, %s, %s" % (d, b, a)
            return "%s_e32 %s, %s, %s" % (op, d, b, a)
        return line
    m = re.match(r"^(v_ashrrev_i32|v_lshlrev_b32|v_lshrrev_b32)_e64 (\S+), (\S+), (\S+)$", line)
    if m:
        op, d, a, b = m.groups(); d = d.rstrip(","); a = a.rstrip(",")
        return "%s_e32 %s, %s, %s" % (op, d, a, b) if _is_v(b) else line
    m = re.match(r"^(v_mov_b32|v_mov_b64)_e64 (.*)$", line)
    if m:
        return "%s_e32 %s" % (m.group(1), m.group(2))
    return line


# ---- instruction alignment (round 5). A lone wave fetches an 8-byte instruction that sits at an address = 4 mod 8 measurably slower than an aligned one (the
# product routines are all 8-byte encodings behind a .p2align 6 and never misaligned; the bodies around them mix in 4-byte scalar instructions, and after an odd
# number of those everything that follows is misaligned until the next one). Default (MBLS_GEN_ALIGN8=0 switches it off for an A/B): a post-pass over every emitted routine that keeps 8-byte
# instructions on 8-byte boundaries by putting an `s_nop 0` in front of the first misaligned one of a run. Sizes: SOP* and VOP1 / VOP2 (_e32) encodings are 4
# bytes, + 4 with a 32-bit literal or a relocation; everything else (VOP3, VOP3P, DS, FLAT) is 8.
ALIGN8 = os.environ.get("MBLS_GEN_ALIGN8", "1") == "1"
_INLINE_F = ("0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0")


def _is_literal(tok):
    tok = tok.strip()
    if "@rel32" in tok or re.search(r"\d+[fb]-\d+[fb]", tok) or re.fullmatch(r"\d+[fb]", tok):
        return True                                  # relocations and label differences are assembled as 32-bit literals
    if re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", tok):
        v = int(tok, 0)
        return not (-16 <= v <= 64)
    return False


def instr_size(line):
    """bytes of one instruction line (0 for labels and directives)"""
    l = line.strip()
    if not l or l.endswith(":") or l.startswith("."):
        return 0
    op = l.split()[0]
    args = l[len(op):]
    toks = [t for t in re.split(r"[,\s]+", args) if t]
    if op.startswith("s_"):
        if op in ("s_waitcnt", "s_nop", "s_barrier", "s_endpgm") or op.startswith("s_cbranch") or op in ("s_setpc_b64", "s_getpc_b64", "s_swappc_b64"):
            return 4
        return 8 if any(_is_literal(t) for t in toks[1:]) else 4
    if op.endswith("_e32"):
        return 8 if any(_is_literal(t) for t in toks[1:]) else 4
    return 8


def align8(lines):
    # (the compiler puts an s_waitcnt in front of an asm body that is a function of its own: the body is put on an 8-byte boundary first)
    out, off = [".p2align 3"], 0
    i = 0
    while i < len(lines):
        l = lines[i]
        s = l.strip()
        if s.startswith(".p2align"):
            off = 0; out.append(l); i += 1; continue
        sz = instr_size(l)
        if sz == 0:
            out.append(l); i += 1; continue
        # a call through a relocation: s_getpc / s_add @rel32@lo+4 / s_addc @rel32@hi+12 / s_swappc -- the offsets in the relocations fix the distances, nothing may
        # go between them: the group starts on an 8-byte boundary (its two 8-byte members then sit at 4 mod 8, the return address is aligned again)
        if s.startswith("s_getpc_b64") and i + 1 < len(lines) and "@rel32" in lines[i + 1]:
            if off % 8:
                out.append("s_nop 0"); off += 4
            for k in range(4):
                out.append(lines[i + k]); off += instr_size(lines[i + k])
            i += 4; continue
        if sz == 8 and off % 8:
            out.append("s_nop 0"); off += 4
        out.append(l); off += sz; i += 1
    return out


# MBLS_GEN_TIMING_NO_VMWAIT=1 / MBLS_GEN_TIMING_NO_LGKMWAIT=1: THROW-AWAY builds for timing only -- the three big routines without their waits for workspace
# loads / for LDS reads (wrong results, same instruction stream otherwise): what the exposed memory latency of a lone wave costs (scripts/dbg/ab_gen.sh). Never
# committed: the freshness tests compare the tracked files with the default generation.
TIMING_NO_VMWAIT = os.environ.get("MBLS_GEN_TIMING_NO_VMWAIT", "0") == "1"
TIMING_NO_LGKMWAIT = os.environ.get("MBLS_GEN_TIMING_NO_LGKMWAIT", "0") == "1"
TIMING_NO_LDS = os.environ.get("MBLS_GEN_TIMING_NO_LDS", "0") == "1"                # ... without their LDS instructions / their workspace loads: what ISSUING them costs
TIMING_NO_GLOADS = os.environ.get("MBLS_GEN_TIMING_NO_GLOADS", "0") == "1"
TIMING_X4 = os.environ.get("MBLS_GEN_TIMING_X4", "0") == "1"                        # ... with every FOUR workspace words of a value moved by ONE 16-byte instruction
TIMING_ROUTINES = ("MBLS_MILLER_LOOP_D_ASM", "MBLS_FINAL_EXP_D_ASM", "MBLS_G2_HASH_TAIL_D_ASM")


def timing_x4(lines):
    """timing only (scripts/dbg/ab_x4.sh): what a workspace of 16-byte vectors per lane would cost to access -- of the twelve dword loads / stores of a value the
    1st, 5th and 9th become dwordx4 accesses at 16 bytes per lane (lane offset 4 x v252 in v102, a register no routine writes), the other nine go away. The
    addresses stay inside the value's own rows; the data is garbage."""
    out = []
    for n, l in enumerate(lines):
        m = re.match(r"global_load_dword v(\d+), v252, s\[74:75\]$", l)
        if m:
            r = int(m.group(1)); j = r % 14 - 2
            if j % 4 == 0 and 0 <= j <= 8:
                out.append("global_load_dwordx4 v[%d:%d], v102, s[74:75]" % (r, r + 3))
            continue
        m = re.match(r"global_store_dword v252, v(\d+), s\[74:75\]$", l)
        if m:
            r = int(m.group(1)); j = r % 14
            if j % 4 == 0 and j <= 8:
                out.append("global_store_dwordx4 v102, v[%d:%d], s[74:75]" % (r, r + 3))
            continue
        out.append(l)
        if n == 0:
            out.append("v_lshlrev_b32_e64 v102, 2, v252")
    return out


def emit(name, lines):
    out = ["#define %s \\" % name]
    if name in TIMING_ROUTINES:
        if TIMING_NO_VMWAIT:
            lines = [l for l in lines if not l.startswith("s_waitcnt vmcnt")]
        if TIMING_NO_LGKMWAIT:
            lines = [l for l in lines if not l.startswith("s_waitcnt lgkmcnt")]
        if TIMING_X4:
            lines = timing_x4(lines)
        if TIMING_NO_LDS:
            lines = [l for l in lines if not l.startswith("ds_")]
        if TIMING_NO_GLOADS:
            lines = [l for l in lines if not l.startswith("global_load")]
    if PREFER_E32:
        lines = [to_e32(l) for l in lines]
    if ALIGN8:
        lines = align8(lines)
    for l in lines:
        out.append('    "%s\\n\\t" \\' % l)
    out.append('    ""')
    return "\n".join(out)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    # MBLS_GEN_OUT_DIR: write there instead of over the tracked file (the freshness tests generate into a temporary directory and compare)
    path = os.path.join(os.environ.get("MBLS_GEN_OUT_DIR") or os.path.join(os.path.dirname(here), "milagro_bls_amd", "csrc"), "mbls_fp_asm.inc")
    txt = "// GENERATED by tools/gen_fp_asm.py -- do not edit.\n// gfx950 Montgomery multiplication, one asm statement, registers per the AMDGPU calling convention.\n"
    txt += emit("MBLS_FP_MUL_ASM", fp_mul_body()) + "\n"
    txt += '#define MBLS_FP_MUL_CLOBBERS "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \\\n'
    txt += '    "s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","s62","s63","vcc","scc"\n'
    txt += emit("MBLS_FP2_MUL_ASM", fp2_mul_body()) + "\n"
    vl = lambda a, b: ",".join('"v%d"' % i for i in range(a, b + 1))
    sg = '"s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","vcc","scc"'
    txt += "// the Fp2 routines overwrite their operand registers (v0..v47 / v0..v23): call sites pass them as read-write operands\n"
    txt += "#define MBLS_FP2_MUL_CLOBBERS %s, %s\n" % (vl(72, 117), sg)
    txt += emit("MBLS_FP2_SQR_ASM", fp2_sqr_body()) + "\n"
    txt += "#define MBLS_FP2_SQR_CLOBBERS %s, %s\n" % (vl(48, 104), sg)
    txt += emit("MBLS_FP2_MULFP_ASM", fp2_mulfp_body()) + "\n"
    txt += "#define MBLS_FP2_MULFP_CLOBBERS %s, %s\n" % (vl(60, 89), sg)
    txt += emit("MBLS_FP_MULPAIR_ASM", fp_mulpair_body()) + "\n"
    txt += "#define MBLS_FP_MULPAIR_CLOBBERS %s, %s\n" % (vl(72, 103), sg)
    for sym, body in pow_subroutines().items():
        txt += emit("MBLS_" + sym.upper()[5:-7] + "_ASM", body) + "\n"
    txt += emit("MBLS_FP_POW_PM3D4_ASM", expand_pow_calls(pow_body(EXP_PM3D4))) + "\n"
    txt += emit("MBLS_FP_POW_PM3D4_W4_ASM", expand_pow_calls(pow_body(EXP_PM3D4, 4))) + "\n"
    txt += emit("MBLS_FP_INV_GCD_ASM", fp_inv_gcd_body()) + "\n"
    txt += "#define MBLS_FP_INV_GCD_CLOBBERS %s, \"s66\",\"s76\",\"s77\",\"vcc\",\"scc\"\n" % vl(12, 84)
    txt += "#define MBLS_FP_POW_CLOBBERS %s,%s, \\\n" % (vl(12, 84), ",".join('"a%d"' % i for i in range(224)))
    txt += '    "s30","s31","s36","s37","s66","s67", %s\n' % sg
    txt += "#define MBLS_FP_POW_W4_CLOBBERS %s,%s, \\\n" % (vl(12, 84), ",".join('"a%d"' % i for i in range(112)))
    txt += '    "s30","s31","s36","s37","s66","s67", %s\n' % sg
    with open(path, "w") as f:
        f.write(txt)
    print("wrote", path, "(%d + %d + %d + %d instructions)" % (len(fp_mul_body()), len(fp2_mul_body()), len(fp2_sqr_body()), len(fp2_mulfp_body())))


if __name__ == "__main__":
    main()
