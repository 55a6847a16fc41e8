#!/usr/bin/env python3
"""Generate milagro_bls_amd/csrc/mbls_fpd_asm.inc: the Fp2 multiplication routines of the digit-form ("D-form") tower routines.

D-form: an Fp value is 14 SIGNED 32-bit digits d_0..d_13, value = sum d_j 2^(28 j), a representative of a * 2^392 mod p (Montgomery
radix 2^392 = the radix of 14 digits). Digits are unsaturated: additions and subtractions are 14 independent v_add_u32 / v_sub_u32
with no carries, no modular correction and no constant offsets (signed digits absorb the subtractions); a product column is a
signed 64-bit sum accumulated by v_mad_i64_i32. The generator of the callers (tools/gen_tower_d.py) tracks exact bounds on the digits
and on the value of everything it computes and renormalises (one carry pass) only where a routine's input limit would be exceeded.

Compared with the 12 x 32-bit-limb routines of tools/gen_fp_asm.py (operands re-cut into digits on entry, results re-packed and
conditionally reduced on exit) a routine here is only the two interleaved product scans: no conversions, no final subtraction, and
its operand registers survive the call.

Register contract (blocks of 14 VGPRs, block i = v[14 i .. 14 i + 13]):
    mbls_fp2_mul_d_asm_fn    a0 blk0, a1 blk1, b0 blk2, b1 blk3 (preserved)  ->  c0 blk5, c1 blk6   (blk4 = a0-a1, blk8 = b1-b0, v98..v101 + v108..v111 accumulators)
    mbls_fp2_sqr_d_asm_fn    a0 blk0, a1 blk1 (preserved)                    ->  c0 blk5, c1 blk6   (blk2..4 = a0+a1, a0-a1, 2 a1)
    mbls_fp2_mulfp_d_asm_fn  a0 blk0, a1 blk1, s blk2 (preserved)            ->  a0 s blk5, a1 s blk6
    mbls_fp_mulpair_d_asm_fn a0 blk0, a1 blk1, b0 blk2, b1 blk3 (preserved)  ->  a0 b0 blk5, a1 b1 blk6   (two independent Fp products)
    mbls_fp_mul1_d_asm_fn    a0 blk0, b0 blk2 (preserved)                    ->  a0 b0 blk5
    mbls_fp_sqrpair_d_asm_fn a0 blk0, a1 blk1 (preserved)                    ->  a0^2 blk5, a1^2 blk6   (blk2, blk3 = doubled digits)
Resident constants (loaded once by the calling routine's shell, load_constants()): digits of p in s40-s47, s56-s61, -p^-1 mod 2^28
in s64, the digit mask in s65. Carries: vcc and s[62:63]. Results: digits 0..12 in [0, 2^28), digit 13 signed (the value lies in
(-X, p + X) with X = sum |a||b| / 2^392, a tiny multiple of p for every operand the callers produce).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_fp_asm import P, P28, NP28, M28, SP28, SNP28, SMASK28, emit, zip2  # noqa: E402

BLK = lambda i: (lambda j: "v%d" % (14 * i + j))
ACC_A, ACC_B = 98, 100
CARRY_A, CARRY_B = "vcc", "s[62:63]"
R392 = 1 << 392


def load_constants():
    L = ["s_mov_b32 %s, 0x%08x" % (SP28(i), P28[i]) for i in range(14)]
    L += ["s_mov_b32 %s, 0x%08x" % (SNP28, NP28), "s_mov_b32 %s, 0x%08x" % (SMASK28, M28)]
    return L


def signed_scan(pairs, out, acc, cy):
    """sum of the listed digit-vector products, Montgomery-reduced by 2^392, on signed digits; quotient digits and then result
    digits in `out` (14 registers, distinct from every operand: the operands are read until the last column)"""
    A, lo = "v[%d:%d]" % (acc, acc + 1), "v%d" % acc
    S, first = [], True
    for k in range(28):
        macs = []
        for i in range(max(0, k - 13), min(k, 13) + 1):
            for (X, Y) in pairs:
                macs.append((X(i), Y(k - i)))
        if k < 14:
            macs += [(SP28(k - i), out(i)) for i in range(k)]
        else:
            macs += [(SP28(k - i), out(i)) for i in range(k - 13, 14)]
        for (x, y) in macs:
            S.append("v_mad_i64_i32 %s, %s, %s, %s, %s" % (A, cy, x, y, "0" if first else A)); first = False
        if k < 14:
            S += ["v_mul_lo_u32 %s, %s, %s" % (out(k), lo, SNP28), "v_and_b32_e64 %s, %s, %s" % (out(k), out(k), SMASK28),
                  "v_mad_i64_i32 %s, %s, %s, %s, %s" % (A, cy, SP28(0), out(k), A), "v_ashrrev_i64 %s, 28, %s" % (A, A)]
        elif k < 27:
            S += ["v_and_b32_e64 %s, %s, %s" % (out(k - 14), lo, SMASK28), "v_ashrrev_i64 %s, 28, %s" % (A, A)]
        else:
            S.append("v_mov_b32_e64 %s, %s" % (out(13), lo))          # top digit: signed, whatever is left
    return S


def fp2_mul_d_body(blocks=None):
    """blocks: the register blocks (A0, A1, B0, B1, DA, C0, C1, DB) as functions digit index -> register name; default: the blocks of the called
    routine (0, 1, 2, 3, 4, 5, 6, 8). tools/gen_tower_d.py inlines the scan at a call site with the blocks the allocator chose: the operands
    are read where they are and the results land where they are wanted. The accumulators stay v98..v101 / v108..v111 (block 7).
    c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0 with THREE products: Karatsuba on the 64-bit column sums, in the subtractive form
    c1 = (a0 - a1)(b1 - b0) + a0 b0 + a1 b1 (differences of non-negative digits are no larger than the digits, so the input limits stay
    those of four plain scans). Both scans walk the columns in lockstep; X_k = sum a0_i b0_(k-i) and Y_k = sum a1_i b1_(k-i) are formed
    once per column (fresh accumulators), the third product accumulates straight into c1's scan, and c0's scan receives X_k - Y_k, c1's
    X_k + Y_k: 3 n + 5 instructions per column instead of 4 n. Measured in isolation (scripts/dbg/fp2dbench.hip): 5 % faster."""
    A0, A1, B0, B1, DA, C0, C1, DB = blocks or (BLK(0), BLK(1), BLK(2), BLK(3), BLK(4), BLK(5), BLK(6), BLK(8))
    ACC0, ACC1, X, Y = "v[98:99]", "v[100:101]", "v[108:109]", "v[110:111]"          # v102..v107 stay untouched (see gen_tower_d.py)
    L = []
    for j in range(14):
        L += ["v_sub_u32_e64 %s, %s, %s" % (DA(j), A0(j), A1(j)), "v_sub_u32_e64 %s, %s, %s" % (DB(j), B1(j), B0(j))]
    first0 = first1 = True
    for k in range(28):
        idx = list(range(max(0, k - 13), min(k, 13) + 1))
        fx = True
        for i in idx:                                   # three independent chains, interleaved
            L.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (X, A0(i), B0(k - i), "0" if fx else X))
            L.append("v_mad_i64_i32 %s, s[62:63], %s, %s, %s" % (Y, A1(i), B1(k - i), "0" if fx else Y)); fx = False
            L.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (ACC1, DA(i), DB(k - i), "0" if first1 else ACC1)); first1 = False
        for i in (range(k) if k < 14 else range(k - 13, 14)):                       # the Montgomery quotient digits times p, both scans
            L.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (ACC0, SP28(k - i), C0(i), "0" if first0 else ACC0)); first0 = False
            L.append("v_mad_i64_i32 %s, s[62:63], %s, %s, %s" % (ACC1, SP28(k - i), C1(i), ACC1))
        if idx:
            L += ["v_lshl_add_u64 %s, %s, 0, %s" % (ACC1, X, ACC1), "v_lshl_add_u64 %s, %s, 0, %s" % (ACC1, Y, ACC1)]                   # + X + Y
            L += ["v_sub_co_u32_e64 v108, vcc, v108, v110", "v_subb_co_u32_e64 v109, vcc, v109, v111, vcc"]                            # X - Y
            L.append("v_mov_b64_e64 %s, %s" % (ACC0, X) if first0 else "v_lshl_add_u64 %s, %s, 0, %s" % (ACC0, X, ACC0)); first0 = False
        for (out, A, lo, cy) in ((C0, ACC0, "v98", "vcc"), (C1, ACC1, "v100", "s[62:63]")):
            if k < 14:
                L += ["v_mul_lo_u32 %s, %s, %s" % (out(k), lo, SNP28), "v_and_b32_e64 %s, %s, %s" % (out(k), out(k), SMASK28),
                      "v_mad_i64_i32 %s, %s, %s, %s, %s" % (A, cy, SP28(0), out(k), A), "v_ashrrev_i64 %s, 28, %s" % (A, A)]
            elif k < 27:
                L += ["v_and_b32_e64 %s, %s, %s" % (out(k - 14), lo, SMASK28), "v_ashrrev_i64 %s, 28, %s" % (A, A)]
            else:
                L.append("v_mov_b32_e64 %s, %s" % (out(13), lo))
    return L


def fp2_mul_d4_body():
    """the same product as four plain scans (two merged pairs): the form fp2_mul_d_body replaced; kept as the reference of
    scripts/dbg/fp2dbench.hip and of the tests (both bodies must give identical digits)"""
    A0, A1, B0, B1, NB, C0, C1 = BLK(0), BLK(1), BLK(2), BLK(3), BLK(4), BLK(5), BLK(6)
    L = ["v_sub_u32_e64 %s, 0, %s" % (NB(j), B1(j)) for j in range(14)]                      # -b1: c0 = a0 b0 - a1 b1
    L += zip2(signed_scan([(A0, B0), (A1, NB)], C0, ACC_A, CARRY_A), signed_scan([(A0, B1), (A1, B0)], C1, ACC_B, CARRY_B))
    return L


def fp2_sqr_d_body(blocks=None):
    """blocks: (A0, A1, S, D, A1D, C0, C1), default 0..6"""
    A0, A1, S, D, A1D, C0, C1 = blocks or (BLK(0), BLK(1), BLK(2), BLK(3), BLK(4), BLK(5), BLK(6))
    L = []
    for j in range(14):                                                                          # c0 = (a0 + a1)(a0 - a1), c1 = a0 (2 a1)
        L += ["v_add_u32_e64 %s, %s, %s" % (S(j), A0(j), A1(j)), "v_sub_u32_e64 %s, %s, %s" % (D(j), A0(j), A1(j)),
              "v_lshlrev_b32_e64 %s, 1, %s" % (A1D(j), A1(j))]
    L += zip2(signed_scan([(S, D)], C0, ACC_A, CARRY_A), signed_scan([(A0, A1D)], C1, ACC_B, CARRY_B))
    return L


def fp2_mulfp_d_body(blocks=None):
    """blocks: (A0, A1, SB, C0, C1), default 0, 1, 2, 5, 6"""
    A0, A1, SB, C0, C1 = blocks or (BLK(0), BLK(1), BLK(2), BLK(5), BLK(6))
    return zip2(signed_scan([(A0, SB)], C0, ACC_A, CARRY_A), signed_scan([(A1, SB)], C1, ACC_B, CARRY_B))


def fp_mulpair_d_body():
    """two independent Fp products (the G1 formulas have their multiplications in pairs): a0 b0, a1 b1"""
    A0, A1, B0, B1, C0, C1 = BLK(0), BLK(1), BLK(2), BLK(3), BLK(5), BLK(6)
    return zip2(signed_scan([(A0, B0)], C0, ACC_A, CARRY_A), signed_scan([(A1, B1)], C1, ACC_B, CARRY_B))


def signed_sqr_scan(X, D, out, acc, cy):
    """X^2, Montgomery-reduced: the cross products once, against the doubled digit vector D (exact: digits carry no carries)"""
    A, lo = "v[%d:%d]" % (acc, acc + 1), "v%d" % acc
    S, first = [], True
    for k in range(28):
        macs = []
        for i in range(max(0, k - 13), min(k, 13) + 1):
            j = k - i
            if i < j:
                macs.append((X(i), D(j)))
            elif i == j:
                macs.append((X(i), X(i)))
        if k < 14:
            macs += [(SP28(k - i), out(i)) for i in range(k)]
        else:
            macs += [(SP28(k - i), out(i)) for i in range(k - 13, 14)]
        for (x, y) in macs:
            S.append("v_mad_i64_i32 %s, %s, %s, %s, %s" % (A, cy, x, y, "0" if first else A)); first = False
        if k < 14:
            S += ["v_mul_lo_u32 %s, %s, %s" % (out(k), lo, SNP28), "v_and_b32_e64 %s, %s, %s" % (out(k), out(k), SMASK28),
                  "v_mad_i64_i32 %s, %s, %s, %s, %s" % (A, cy, SP28(0), out(k), A), "v_ashrrev_i64 %s, 28, %s" % (A, A)]
        elif k < 27:
            S += ["v_and_b32_e64 %s, %s, %s" % (out(k - 14), lo, SMASK28), "v_ashrrev_i64 %s, 28, %s" % (A, A)]
        else:
            S.append("v_mov_b32_e64 %s, %s" % (out(13), lo))
    return S


def fp_sqrpair_d_body():
    """two independent Fp squarings a0^2, a1^2 (blocks 2, 3 = the doubled digits)"""
    A0, A1, D0, D1, C0, C1 = BLK(0), BLK(1), BLK(2), BLK(3), BLK(5), BLK(6)
    L = []
    for j in range(14):
        L += ["v_lshlrev_b32_e64 %s, 1, %s" % (D0(j), A0(j)), "v_lshlrev_b32_e64 %s, 1, %s" % (D1(j), A1(j))]
    return L + zip2(signed_sqr_scan(A0, D0, C0, ACC_A, CARRY_A), signed_sqr_scan(A1, D1, C1, ACC_B, CARRY_B))


def fp_mul1_d_body():
    """one Fp product: a0 b0"""
    return signed_scan([(BLK(0), BLK(2))], BLK(5), ACC_A, CARRY_A)


ROUTINE_BODIES = {"mbls_fp2_mul_d_asm_fn": fp2_mul_d_body, "mbls_fp2_sqr_d_asm_fn": fp2_sqr_d_body, "mbls_fp2_mulfp_d_asm_fn": fp2_mulfp_d_body,
                  "mbls_fp_mulpair_d_asm_fn": fp_mulpair_d_body, "mbls_fp_mul1_d_asm_fn": fp_mul1_d_body,
                  "mbls_fp_sqrpair_d_asm_fn": fp_sqrpair_d_body}


# ---- input limits: the worst column of a scan must stay inside a signed 64-bit accumulator
def column_ok(products):
    """products: list of (Da, Db) digit-magnitude bounds of the vector products summed in one scan"""
    worst = 14 * sum(a * b for a, b in products) + 14 * (1 << 56)
    worst += (worst >> 28) + (1 << 32)           # carry from the previous column, slack
    return worst < (1 << 63)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    # MBLS_GEN_OUT_DIR: write there instead of over the tracked file (the freshness tests generate into a temporary directory and compare)
    path = os.path.join(os.environ.get("MBLS_GEN_OUT_DIR") or os.path.join(os.path.dirname(here), "milagro_bls_amd", "csrc"), "mbls_fpd_asm.inc")
    txt = "// GENERATED by tools/gen_fpd_asm.py -- do not edit.\n// gfx950 Fp2 multiplication routines on 14 signed 28-bit digits (D-form), private calling convention.\n"
    for sym, fn in ROUTINE_BODIES.items():
        body = fn()
        txt += emit("MBLS_" + sym.upper()[5:-7] + "_ASM", [".p2align 6"] + body) + "\n"
        print(sym, len(body), "instructions")
    with open(path, "w") as f:
        f.write(txt)
    print("wrote", path)


if __name__ == "__main__":
    main()
