#!/usr/bin/env python3
"""Generate milagro_bls_amd/csrc/mbls_fp_asm.inc: the hand-scheduled gfx950 body of the Montgomery multiplication.

One asm statement per multiplication, physical registers chosen to coincide with the AMDGPU calling convention
(a in v[0:11], b in v[12:23], result in v[0:11]), so a call costs no argument shuffling and hipcc cannot interleave
its own s_nop padding (it pads every inline-asm boundary, which costs an issue slot per multiply-accumulate when only
one wave lives on the SIMD). Product scanning over a 96-bit column accumulator v[36:37], v38:
    v_mad_u64_u32 v[36:37], vcc, x, y, v[36:37] ; v_addc_co_u32 v38, vcc, 0, v38, vcc
Run:  python3 tools/gen_fp_asm.py   (output is committed; tests/test_build_cpu.py checks it is up to date)
"""
import os
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
NP0 = (-pow(P, -1, 1 << 32)) % (1 << 32)
PL = [(P >> (32 * i)) & 0xFFFFFFFF for i in range(12)]
A = lambda i: "v%d" % i            # a[i], later t[i]
B = lambda i: "v%d" % (12 + i)
M = lambda i: "v%d" % (24 + i)     # m[i], later d[i]
# scalar registers that the calling convention leaves to the callee (s40-s47, s56-s63): no save/restore code is generated
SP = lambda i: "s%d" % (40 + i if i < 8 else 56 + i - 8)    # modulus limbs
SNP = "s60"
ACC, ACC_HI, LO, MID = "v[36:37]", "v38", "v36", "v37"


def mac(x, y):
    return ["v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, x, y, ACC), "v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (ACC_HI, ACC_HI)]


def body(square=False):
    L = []
    for i in range(12):
        L.append("s_mov_b32 %s, 0x%08x" % (SP(i), PL[i]))
    L.append("s_mov_b32 %s, 0x%08x" % (SNP, NP0))
    L += ["v_mov_b32_e32 %s, 0" % LO, "v_mov_b32_e32 %s, 0" % MID, "v_mov_b32_e32 %s, 0" % ACC_HI]
    bb = (lambda i: A(i)) if square else B
    for k in range(24):
        lo, hi = max(0, k - 11), min(k, 11)
        for i in range(lo, hi + 1):
            L += mac(A(i), bb(k - i))
        if k < 12:
            for i in range(0, k):
                L += mac(SP(k - i), M(i))
            L.append("v_mul_lo_u32 %s, %s, %s" % (M(k), SNP, LO))
            L += mac(SP(0), M(k))
            # shift the accumulator by one limb (low word is now zero)
            L += ["v_mov_b32_e32 %s, %s" % (LO, MID), "v_mov_b32_e32 %s, %s" % (MID, ACC_HI), "v_mov_b32_e32 %s, 0" % ACC_HI]
        else:
            for i in range(k - 11, 12):
                L += mac(SP(k - i), M(i))
            # a[k-12] is dead from column k-1 on (its last use is column k-1): the result limb goes there
            L += ["v_mov_b32_e32 %s, %s" % (A(k - 12), LO), "v_mov_b32_e32 %s, %s" % (LO, MID), "v_mov_b32_e32 %s, %s" % (MID, ACC_HI),
                  "v_mov_b32_e32 %s, 0" % ACC_HI]
    # conditional subtraction of p: d = t - p in the m registers
    # (an SGPR source plus the VCC carry-in would be two constant-bus reads: the modulus goes through the dead b registers)
    for i in range(12):
        L.append("v_mov_b32_e32 %s, %s" % (B(i), SP(i)))
    L.append("v_sub_co_u32_e32 %s, vcc, %s, %s" % (M(0), A(0), B(0)))
    for i in range(1, 12):
        L.append("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (M(i), A(i), B(i)))
    # take d when there was no borrow or the 13th limb is set
    L += ["v_cmp_ne_u32_e64 s[62:63], 0, %s" % LO, "s_not_b64 vcc, vcc", "s_nop 1", "s_or_b64 vcc, vcc, s[62:63]", "s_nop 1"]
    for i in range(12):
        L.append("v_cndmask_b32_e32 %s, %s, %s, vcc" % (A(i), A(i), M(i)))
    return L


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


def fp2_mul_body():
    """c0 = a0 b0 + a1 (p - b1), c1 = a0 b1 + a1 b0 as two sum-of-two-products scans with ONE Montgomery reduction each
    (864 multiply-accumulates instead of 3 x 300 for Karatsuba, no modular additions besides the negation of b1). The two
    scans are independent and are interleaved instruction by instruction (carries in vcc and s[62:63]) so that a wave that is
    alone on its SIMD always has an independent instruction to issue.
    Private calling convention (see fp2_mul in mbls_tower.h): a0 v[0:11], a1 v[12:23], b0 v[24:35], b1 v[36:47] (all preserved);
    c0 -> v[48:59], c1 -> v[60:71]; scratch v72-v92."""
    A0, A1, B0, B1, C0, C1, NB = VR(0), VR(12), VR(24), VR(36), VR(48), VR(60), VR(72)
    tmp = "v92"
    L = load_modulus()
    # nb1 = p - b1 (b1 < p; b1 = 0 gives p, which is fine as a factor)
    L.append("v_mov_b32_e32 %s, %s" % (tmp, SP(0)))
    L.append("v_sub_co_u32_e32 %s, vcc, %s, %s" % (NB(0), tmp, B1(0)))
    for i in range(1, 12):
        L.append("v_mov_b32_e32 %s, %s" % (tmp, SP(i)))
        L.append("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (NB(i), tmp, B1(i)))
    L += zip2(Chain([(A0, B0), (A1, NB)], C0, 84, "vcc").stream(), Chain([(A0, B1), (A1, B0)], C1, 88, CARRY_B).stream())
    L += cond_sub(C0, NB, tmp)
    L += cond_sub(C1, NB, tmp)
    return L


def fp2_sqr_body():
    """c0 = (a0 + a1)(a0 - a1 + p), c1 = a0 (2 a1): two single-product scans on unreduced operands (all < 2p, so the product
    is < 1.41 p after reduction and one conditional subtraction finishes it); 600 multiply-accumulates, no modular additions;
    the two scans interleaved like in fp2_mul.
    Private convention: a0 v[0:11], a1 v[12:23] (preserved); c0 -> v[24:35], c1 -> v[36:47]; scratch v48-v92."""
    A0, A1, C0, C1, S, D, A1D = VR(0), VR(12), VR(24), VR(36), VR(48), VR(60), VR(72)
    tmp = "v92"
    L = load_modulus()
    L.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (S(0), A0(0), A1(0)))
    for i in range(1, 12):
        L.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (S(i), A0(i), A1(i)))
    L.append("v_mov_b32_e32 %s, %s" % (tmp, SP(0)))
    L.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (D(0), A0(0), tmp))
    for i in range(1, 12):
        L.append("v_mov_b32_e32 %s, %s" % (tmp, SP(i)))
        L.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (D(i), A0(i), tmp))
    L.append("v_sub_co_u32_e32 %s, vcc, %s, %s" % (D(0), D(0), A1(0)))
    for i in range(1, 12):
        L.append("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (D(i), D(i), A1(i)))
    L.append("v_lshlrev_b32_e32 %s, 1, %s" % (A1D(0), A1(0)))
    for i in range(1, 12):
        L.append("v_alignbit_b32 %s, %s, %s, 31" % (A1D(i), A1(i), A1(i - 1)))
    L += zip2(Chain([(S, D)], C0, 84, "vcc").stream(), Chain([(A0, A1D)], C1, 88, CARRY_B).stream())
    L += cond_sub(C0, S, tmp)
    L += cond_sub(C1, S, tmp)
    return L


def fp2_mulfp_body():
    """c0 = a0 s, c1 = a1 s (an Fp2 element times an Fp element) as two interleaved single-product scans.
    Private convention: a0 v[0:11], a1 v[12:23], s v[24:35] (preserved); c0 -> v[36:47], c1 -> v[48:59]; scratch v60-v92."""
    A0, A1, S, C0, C1, DF = VR(0), VR(12), VR(24), VR(36), VR(48), VR(60)
    tmp = "v92"
    L = load_modulus()
    L += zip2(Chain([(A0, S)], C0, 84, "vcc").stream(), Chain([(A1, S)], C1, 88, CARRY_B).stream())
    L += cond_sub(C0, DF, tmp)
    L += cond_sub(C1, DF, tmp)
    return L


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


def emit(name, lines):
    out = ["#define %s \\" % name]
    for l in lines:
        out.append('    "%s\\n\\t" \\' % l)
    out.append('    ""')
    return "\n".join(out)


def square_body():
    # squaring in a square-only call: b is not an argument, so a[k-12] may not be overwritten while a[j] (j = k - i) is still
    # needed as the second factor: a[j] with j up to 11 is needed until column 22, so the result goes to v[12:23] instead.
    L = body(square=True)
    return L


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.path.dirname(here), "milagro_bls_amd", "csrc", "mbls_fp_asm.inc")
    txt = "// GENERATED by tools/gen_fp_asm.py -- do not edit.\n// gfx950 Montgomery multiplication, one asm statement, registers per the AMDGPU calling convention.\n"
    txt += emit("MBLS_FP_MUL_ASM", fp_mul_body()) + "\n"
    txt += '#define MBLS_FP_MUL_CLOBBERS "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \\\n'
    txt += '    "s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","s62","s63","vcc","scc"\n'
    txt += emit("MBLS_FP2_MUL_ASM", fp2_mul_body()) + "\n"
    vl = lambda a, b: ",".join('"v%d"' % i for i in range(a, b + 1))
    sg = '"s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","s62","s63","vcc","scc"'
    txt += "#define MBLS_FP2_MUL_CLOBBERS %s, %s\n" % (vl(72, 92), sg)
    txt += emit("MBLS_FP2_SQR_ASM", fp2_sqr_body()) + "\n"
    txt += "#define MBLS_FP2_SQR_CLOBBERS %s, %s\n" % (vl(48, 92), sg)
    txt += emit("MBLS_FP2_MULFP_ASM", fp2_mulfp_body()) + "\n"
    txt += "#define MBLS_FP2_MULFP_CLOBBERS %s, %s\n" % (vl(60, 92), sg)
    with open(path, "w") as f:
        f.write(txt)
    print("wrote", path, "(%d + %d + %d + %d instructions)" % (len(fp_mul_body()), len(fp2_mul_body()), len(fp2_sqr_body()), len(fp2_mulfp_body())))


if __name__ == "__main__":
    main()
