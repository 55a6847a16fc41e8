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


def fp2_mul_body():
    """c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0 as two sum-of-two-products scans with ONE Montgomery reduction each
    (864 multiply-accumulates instead of 3 x 300 for Karatsuba, no modular additions besides the negation of b1).
    Private calling convention (see fp2_mul in mbls_tower.h): a0 v[0:11], a1 v[12:23], b0 v[24:35], b1 v[36:47] (all preserved);
    c0 -> v[48:59], c1 -> v[60:71]; scratch v72-v99."""
    A0 = lambda i: "v%d" % i
    A1 = lambda i: "v%d" % (12 + i)
    B0 = lambda i: "v%d" % (24 + i)
    B1 = lambda i: "v%d" % (36 + i)
    C0 = lambda i: "v%d" % (48 + i)
    C1 = lambda i: "v%d" % (60 + i)
    NB = lambda i: "v%d" % (72 + i)       # p - b1
    MM = lambda i: "v%d" % (84 + i)       # Montgomery quotients, then t - p
    acc, lo, mid, hi = "v[96:97]", "v96", "v97", "v98"

    def mac2(x, y):
        return ["v_mad_u64_u32 %s, vcc, %s, %s, %s" % (acc, x, y, acc), "v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (hi, hi)]
    L = []
    for i in range(12):
        L.append("s_mov_b32 %s, 0x%08x" % (SP(i), PL[i]))
    L.append("s_mov_b32 %s, 0x%08x" % (SNP, NP0))
    # nb1 = p - b1 (b1 < p; b1 = 0 gives p, which is fine as a factor)
    for i in range(12):
        L.append("v_mov_b32_e32 %s, %s" % (MM(i), SP(i)))
    L.append("v_sub_co_u32_e32 %s, vcc, %s, %s" % (NB(0), MM(0), B1(0)))
    for i in range(1, 12):
        L.append("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (NB(i), MM(i), B1(i)))

    def scan(X0, Y0, X1, Y1, OUT):
        S = ["v_mov_b32_e32 %s, 0" % lo, "v_mov_b32_e32 %s, 0" % mid, "v_mov_b32_e32 %s, 0" % hi]
        for k in range(24):
            for i in range(max(0, k - 11), min(k, 11) + 1):
                S += mac2(X0(i), Y0(k - i))
                S += mac2(X1(i), Y1(k - i))
            if k < 12:
                for i in range(0, k):
                    S += mac2(SP(k - i), MM(i))
                S.append("v_mul_lo_u32 %s, %s, %s" % (MM(k), SNP, lo))
                S += mac2(SP(0), MM(k))
                S += ["v_mov_b32_e32 %s, %s" % (lo, mid), "v_mov_b32_e32 %s, %s" % (mid, hi), "v_mov_b32_e32 %s, 0" % hi]
            else:
                for i in range(k - 11, 12):
                    S += mac2(SP(k - i), MM(i))
                S += ["v_mov_b32_e32 %s, %s" % (OUT(k - 12), lo), "v_mov_b32_e32 %s, %s" % (lo, mid), "v_mov_b32_e32 %s, %s" % (mid, hi),
                      "v_mov_b32_e32 %s, 0" % hi]
        # t < 1.2 p: one conditional subtraction (the 13th limb is always zero here); p goes through v99 limb by limb
        S.append("v_mov_b32_e32 v99, %s" % SP(0))
        S.append("v_sub_co_u32_e32 %s, vcc, %s, v99" % (MM(0), OUT(0)))
        for i in range(1, 12):
            S.append("v_mov_b32_e32 v99, %s" % SP(i))
            S.append("v_subb_co_u32_e32 %s, vcc, %s, v99, vcc" % (MM(i), OUT(i)))
        for i in range(12):
            S.append("v_cndmask_b32_e32 %s, %s, %s, vcc" % (OUT(i), MM(i), OUT(i)))     # borrow ? t : t - p
        return S
    L += scan(A0, B0, A1, NB, C0)
    L += scan(A0, B1, A1, B0, C1)
    return L


def fp2_sqr_body():
    """c0 = (a0 + a1)(a0 - a1 + p), c1 = a0 (2 a1): two single-product scans on unreduced operands (all < 2p, so the product
    is < 1.41 p after reduction and one conditional subtraction finishes it); 600 multiply-accumulates, no modular additions.
    Private convention: a0 v[0:11], a1 v[12:23] (preserved); c0 -> v[24:35], c1 -> v[36:47]; scratch v48-v99."""
    A0 = lambda i: "v%d" % i
    A1 = lambda i: "v%d" % (12 + i)
    C0 = lambda i: "v%d" % (24 + i)
    C1 = lambda i: "v%d" % (36 + i)
    S = lambda i: "v%d" % (48 + i)
    D = lambda i: "v%d" % (60 + i)
    A1D = lambda i: "v%d" % (72 + i)
    MM = lambda i: "v%d" % (84 + i)
    acc, lo, mid, hi = "v[96:97]", "v96", "v97", "v98"

    def mac2(x, y):
        return ["v_mad_u64_u32 %s, vcc, %s, %s, %s" % (acc, x, y, acc), "v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (hi, hi)]
    L = []
    for i in range(12):
        L.append("s_mov_b32 %s, 0x%08x" % (SP(i), PL[i]))
    L.append("s_mov_b32 %s, 0x%08x" % (SNP, NP0))
    L.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (S(0), A0(0), A1(0)))
    for i in range(1, 12):
        L.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (S(i), A0(i), A1(i)))
    for i in range(12):
        L.append("v_mov_b32_e32 %s, %s" % (MM(i), SP(i)))
    L.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (D(0), A0(0), MM(0)))
    for i in range(1, 12):
        L.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (D(i), A0(i), MM(i)))
    L.append("v_sub_co_u32_e32 %s, vcc, %s, %s" % (D(0), D(0), A1(0)))
    for i in range(1, 12):
        L.append("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (D(i), D(i), A1(i)))
    L.append("v_lshlrev_b32_e32 %s, 1, %s" % (A1D(0), A1(0)))
    for i in range(1, 12):
        L.append("v_alignbit_b32 %s, %s, %s, 31" % (A1D(i), A1(i), A1(i - 1)))

    def scan(X, Y, OUT):
        Sx = ["v_mov_b32_e32 %s, 0" % lo, "v_mov_b32_e32 %s, 0" % mid, "v_mov_b32_e32 %s, 0" % hi]
        for k in range(24):
            for i in range(max(0, k - 11), min(k, 11) + 1):
                Sx += mac2(X(i), Y(k - i))
            if k < 12:
                for i in range(0, k):
                    Sx += mac2(SP(k - i), MM(i))
                Sx.append("v_mul_lo_u32 %s, %s, %s" % (MM(k), SNP, lo))
                Sx += mac2(SP(0), MM(k))
                Sx += ["v_mov_b32_e32 %s, %s" % (lo, mid), "v_mov_b32_e32 %s, %s" % (mid, hi), "v_mov_b32_e32 %s, 0" % hi]
            else:
                for i in range(k - 11, 12):
                    Sx += mac2(SP(k - i), MM(i))
                Sx += ["v_mov_b32_e32 %s, %s" % (OUT(k - 12), lo), "v_mov_b32_e32 %s, %s" % (lo, mid), "v_mov_b32_e32 %s, %s" % (mid, hi),
                       "v_mov_b32_e32 %s, 0" % hi]
        Sx.append("v_mov_b32_e32 v99, %s" % SP(0))
        Sx.append("v_sub_co_u32_e32 %s, vcc, %s, v99" % (MM(0), OUT(0)))
        for i in range(1, 12):
            Sx.append("v_mov_b32_e32 v99, %s" % SP(i))
            Sx.append("v_subb_co_u32_e32 %s, vcc, %s, v99, vcc" % (MM(i), OUT(i)))
        for i in range(12):
            Sx.append("v_cndmask_b32_e32 %s, %s, %s, vcc" % (OUT(i), MM(i), OUT(i)))
        return Sx
    L += scan(S, D, C0)
    L += scan(A0, A1D, C1)
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
    txt += emit("MBLS_FP_MUL_ASM", body()) + "\n"
    txt += '#define MBLS_FP_MUL_CLOBBERS "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38", \\\n'
    txt += '    "s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","s62","s63","vcc","scc"\n'
    txt += emit("MBLS_FP2_MUL_ASM", fp2_mul_body()) + "\n"
    txt += '#define MBLS_FP2_MUL_CLOBBERS "v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90", \\\n'
    txt += '    "v91","v92","v93","v94","v95","v96","v97","v98","v99","s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","vcc","scc"\n'
    txt += emit("MBLS_FP2_SQR_ASM", fp2_sqr_body()) + "\n"
    txt += '#define MBLS_FP2_SQR_CLOBBERS "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67", \\\n'
    txt += '    "v68","v69","v70","v71", MBLS_FP2_MUL_CLOBBERS\n'
    with open(path, "w") as f:
        f.write(txt)
    print("wrote", path, "(%d + %d + %d instructions)" % (len(body()), len(fp2_mul_body()), len(fp2_sqr_body())))


if __name__ == "__main__":
    main()
