#!/usr/bin/env python3
"""Generate milagro_bls_amd/csrc/mbls_towerd_asm.inc: the hot paths of the pairing as straight-line gfx950 routines in DIGIT FORM --
the whole Miller loop (two pairs / one pair), the whole final exponentiation, and runs of G2 doublings.

With 12 saturated 32-bit limbs between multiplications (tools/gen_fp_asm.py's routines, still used by the compiled code around the hot
paths) every multiplication re-cuts its operands into 28-bit digits and packs / conditionally reduces its results (15 % of its
instructions), and every addition is a carry chain with a conditional correction (36 instructions).
Here a value STAYS 14 signed 28-bit digits (tools/gen_fpd_asm.py, "D-form", Montgomery radix 2^392) from the moment it enters a
routine until it leaves: multiplications are bare product scans, additions / subtractions / doublings are 14 independent
instructions without carries or corrections. What replaces modular reduction is bookkeeping done HERE, at generation time: every value
carries exact interval bounds on its digits, its top digit and its integer value (class Bound); an operation whose result could
leave the 32-bit digit range, or an operand that could overflow a 64-bit product column of a multiplication routine, gets one
carry pass ("norm", 39 instructions) or, when it is the value itself that has grown too wide, a fused quotient-estimate / subtract /
carry pass ("reduce", 60 instructions) inserted in front. A Montgomery product brings any operands back to (-eps, p + eps), so
products never accumulate growth.

A computation is written once as a program (class Prog: a list of operations on value handles, same formulas as mbls_tower.h /
mbls_pairing.h) and walked by the allocator (class AllocD). Storage: 14-register blocks -- 18 VGPR blocks (blocks 0..7 are the window
of the multiplication routines: operands in 0..3, which the routines preserve, results in 5 and 6, block 4 / 7 scratch), 18 AGPR
blocks, 11 LDS digit slots per lane, and workspace (HBM) slots that serve as homes: a value with a home can be dropped from the
registers and fetched again (prefetched a few calls ahead, with exact s_waitcnt vmcnt counts); Belady eviction with exact next-use
knowledge. No lane-private memory anywhere.

Everything emitted is executed on the CPU by tools/asm_sim.py: the instruction stream of every program against the same program run
on field values, and the complete routines against independent big-integer models (tests/test_asm_sim_d_cpu.py).
Run:  python3 tools/gen_tower_d.py    (output committed; tests check it is up to date)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_fp_asm import P, P28, M28, SP28, SMASK28, emit  # noqa: E402
from gen_fpd_asm import load_constants, column_ok, fp2_mul_d_body, fp2_sqr_d_body, fp2_mulfp_d_body  # noqa: E402


class Prog:
    """Records a tower computation as a list of operations on Fp value handles: (kind, outputs, inputs, aux). The allocator below walks
    the list and emits digit-form code; tests/test_asm_sim_d_cpu.py runs the same programs on field values. On carry-free digits the
    small-constant multiple k * a is one instruction per digit and 2^s a + b is one v_lshl_add_u32: both are single operations."""

    def __init__(self):
        self.ops = []
        self.nval = 0
        self.init_loc = {}

    def new(self):
        self.nval += 1
        return self.nval - 1

    def live_in(self, loc):
        v = self.new()
        self.init_loc[v] = loc
        return v

    def add(self, a, b):
        d = self.new(); self.ops.append(("add", [d], [a, b], None)); return d

    def sub(self, a, b):
        d = self.new(); self.ops.append(("sub", [d], [a, b], None)); return d

    def const(self, value):
        d = self.new(); self.ops.append(("const", [d], [], value)); return d

    def sel(self, mask, a, b):                 # mask (an SGPR pair, one bit per lane) ? b : a
        d = self.new(); self.ops.append(("sel", [d], [a, b], mask)); return d

    def call(self, kind, ins):
        outs = [self.new() for _ in ROUTINES[kind]["outs"]]
        self.ops.append((kind, outs, list(ins), None))
        return tuple(outs)

    def store(self, a, loc):
        self.ops.append(("store", [], [a], loc))

    def keep(self, vals):
        """values that stay in their live-in homes for the next round of the loop (never moved, only copied)"""
        self.ops.append(("keep", [], list(vals), None))

    # ---- Fp2 layer (an Fp2 is a pair of values)
    def pair(self, k0, a0, b0, k1, a1, b1):
        """two independent Fp additions / subtractions (the two halves of an Fp2 operation)"""
        d0, d1 = self.new(), self.new()
        self.ops.append(("pair", [d0, d1], [a0, b0, a1, b1], (k0, k1)))
        return (d0, d1)

    def add2(self, a, b): return self.pair("add", a[0], b[0], "add", a[1], b[1])
    def sub2(self, a, b): return self.pair("sub", a[0], b[0], "sub", a[1], b[1])
    def sqr2(self, a): return self.call("sqr", [a[0], a[1]])
    def mulfp2(self, a, s): return self.call("mulfp", [a[0], a[1], s])
    def mul_xi2(self, a): return self.pair("sub", a[0], a[1], "add", a[0], a[1])    # (1 + i) a
    def sel2(self, mask, a, b): return (self.sel(mask, a[0], b[0]), self.sel(mask, a[1], b[1]))
    def store2(self, a, slot2): self.store(a[0], ("l", 2 * slot2)); self.store(a[1], ("l", 2 * slot2 + 1))

    # ---- Fp6 layer (lists of three Fp2), same formulas as mbls_tower.h
    def add6(self, a, b): return [self.add2(a[i], b[i]) for i in range(3)]
    def sub6(self, a, b): return [self.sub2(a[i], b[i]) for i in range(3)]
    def mul_v6(self, a): return [self.mul_xi2(a[2]), a[0], a[1]]

    def mul6(self, a, b):
        t0, t1, t2 = self.mul2(a[0], b[0]), self.mul2(a[1], b[1]), self.mul2(a[2], b[2])
        c0 = self.mul2(self.add2(a[1], a[2]), self.add2(b[1], b[2]))
        c0 = self.add2(self.mul_xi2(self.sub2(self.sub2(c0, t1), t2)), t0)
        c1 = self.mul2(self.add2(a[0], a[1]), self.add2(b[0], b[1]))
        c1 = self.add2(self.sub2(self.sub2(c1, t0), t1), self.mul_xi2(t2))
        c2 = self.mul2(self.add2(a[0], a[2]), self.add2(b[0], b[2]))
        c2 = self.add2(self.sub2(self.sub2(c2, t0), t2), t1)
        return [c0, c1, c2]

    def mul6_01(self, a, x, y):            # a (x + y v)
        t0, t1 = self.mul2(a[0], x), self.mul2(a[1], y)
        c1 = self.sub2(self.sub2(self.mul2(self.add2(a[0], a[1]), self.add2(x, y)), t0), t1)
        c0 = self.add2(self.mul_xi2(self.mul2(a[2], y)), t0)
        c2 = self.add2(self.mul2(a[2], x), t1)
        return [c0, c1, c2]

    def mul6_1(self, a, y):                # a (y v)
        return [self.mul_xi2(self.mul2(a[2], y)), self.mul2(a[0], y), self.mul2(a[1], y)]

    # ---- Fp12 layer: (c0, c1) of Fp6
    def sqr12(self, f):
        a, b = f
        ab = self.mul6(a, b)
        s = self.add6(a, b)
        t = self.add6(a, self.mul_v6(b))
        st = self.sub6(self.mul6(s, t), ab)
        return (self.sub6(st, self.mul_v6(ab)), self.add6(ab, ab))

    def mul12_line(self, f, c0, c2, c3):   # f (c0 + c2 w^2 + c3 w^3)
        a, b = f
        t0 = self.mul6_01(a, c0, c2)
        t1 = self.mul6_1(b, c3)
        c1 = self.mul6_01(self.add6(a, b), c0, self.add2(c2, c3))
        c1 = self.sub6(self.sub6(c1, t0), t1)
        return (self.add6(t0, self.mul_v6(t1)), c1)

    # ---- Fp products in pairs (the G1 formulas) and masks
    def mulpair(self, a0, b0, a1, b1):
        return self.call("sqrpair", [a0, a1]) if (a0 == b0 and a1 == b1 and a0 != a1) else self.call("mulpair", [a0, a1, b0, b1])
    def mul1(self, a, b): return self.call("mul1", [a, b])[0]

    def iszero(self, a, mask):
        """mask (an SGPR pair) <- lanes in which a = 0 mod p"""
        self.ops.append(("iszero", [], [a], mask))

    def mask_and(self, dst, a, b):
        self.ops.append(("mask_and", [], [], (dst, a, b)))

    def iszero2(self, a, mask, tmp):
        """mask <- lanes in which the Fp2 value a is zero (tmp: a second SGPR pair)"""
        self.iszero(a[0], mask); self.iszero(a[1], tmp); self.mask_and(mask, mask, tmp)

    def mask_orn2(self, dst, a, b):
        """dst <- a | ~b on lane masks"""
        self.ops.append(("mask_orn2", [], [], (dst, a, b)))

    # ---- digit-form operations
    def scale(self, a, k):
        d = self.new(); self.ops.append(("scale", [d], [a], k)); return d

    def shadd(self, a, s, b):
        d = self.new(); self.ops.append(("shadd", [d], [a, b], s)); return d

    def neg(self, a):
        d = self.new(); self.ops.append(("neg", [d], [a], None)); return d

    def inv(self, a):
        """1 / a by the inversion routine of tools/gen_fp_asm.py (safegcd divsteps; 0 -> 0). That routine works on 12 canonical words of
        the 2^384 domain: a * (2^384 mod p) is such a value in digit form, and the words it returns, cut as digits of words * 2^8, are
        the inverse in the 2^392 domain again."""
        w = self.mulfp2((a, self.const(0)), self.const(K384))[0]
        r = self.new(); self.ops.append(("reduce", [r], [w], None))
        d = self.new(); self.ops.append(("inv", [d], [r], None)); return d

    def mul2(self, a, b): return self.sqr2(a) if tuple(a) == tuple(b) else self.call("mul", [a[0], a[1], b[0], b[1]])
    def pow34(self, a):
        """a^((p-3)/4) by the fixed-exponent routine of tools/gen_fp_asm.py (same word interface as inv)"""
        w = self.mulfp2((a, self.const(0)), self.const(K384))[0]
        r = self.new(); self.ops.append(("reduce", [r], [w], None))
        d = self.new(); self.ops.append(("pow34", [d], [r], None)); return d

    def sgn0_2(self, a, mask, tmp):
        """mask <- lanes in which sgn0(a) = 1 for the Fp2 value a (RFC 9380 section 4.1: parity of the plain representative of c0,
        or of c1 when c0 = 0). The plain integers come out of a Montgomery product with the constant whose representation is 1."""
        one = self.const(1)
        x = self.mulpair(a[0], one, a[1], one)
        self.ops.append(("sgn0", [], [x[0], x[1]], (mask, tmp)))

    def mask_xor(self, dst, a, b):
        self.ops.append(("mask_xor", [], [], (dst, a, b)))

    def mask_or(self, dst, a, b):
        self.ops.append(("mask_or", [], [], (dst, a, b)))

    def neg2(self, a): return (self.neg(a[0]), self.neg(a[1]))
    def conj2(self, a): return (a[0], self.neg(a[1]))
    def neg6(self, a): return [self.neg2(c) for c in a]
    def scale2(self, a, k): return (self.scale(a[0], k), self.scale(a[1], k))
    def shadd2(self, a, s, b): return (self.shadd(a[0], s, b[0]), self.shadd(a[1], s, b[1]))
    def dbl2(self, a): return self.scale2(a, 2)
    def mul3_2(self, a): return self.scale2(a, 3)
    def mul4_2(self, a): return self.scale2(a, 4)
    def mul8_2(self, a): return self.scale2(a, 8)
    def mul12_2(self, a): return self.scale2(a, 12)


# ---------------------------------------------------------------------------------------------- products in pairs (two lanes per item)
# Batches that leave at least half of the SIMDs idle give every item TWO lanes (2 j and 2 j + 1 of a wave). Both lanes hold the same values and run
# the same stream; what the second lane buys is the products: two independent Fp2 products of the same kind become ONE call -- the even
# lane multiplies the first pair of operands, the odd lane the second (its operand slots are overwritten under an exec mask), the results
# cross through DPP and a masked v_swap_b32 sorts them, so that afterwards both lanes hold both products again. pair_products() finds the
# pairs in a recorded program: a later product of the same kind whose operands are ready, or become ready by hoisting carry-free
# operations (the Karatsuba sums) in front of the earlier one.
PAIR_ROLE, PAIR_EXEC = "s[94:95]", "s[82:83]"        # the odd lanes (of those that entered the routine) / every lane that entered the routine
PAIRABLE = ("mul", "sqr", "mulfp")
PURE_OPS = {"add", "sub", "pair", "scale", "shadd", "neg", "const", "norm", "reduce"}


def pair_products(p, window=160):
    ops = list(p.ops)
    if not ops:                                      # (a model program computes as it records: nothing to pair)
        return p
    produced_at = {}
    for i, op in enumerate(ops):
        for o in op[1]:
            produced_at[o] = i
    emitted, out = set(), []

    def chain(j, i):
        """the not yet emitted operations in (i, j) that op j needs, in order -- or None if one of them is no carry-free operation"""
        need, stack = set(), list(ops[j][2])
        while stack:
            k = produced_at.get(stack.pop(), -1)
            if k < 0 or k in emitted or k in need:
                continue
            if k >= i and (k == i or ops[k][0] not in PURE_OPS):
                return None
            need.add(k); stack += list(ops[k][2])
        return sorted(need)
    for i in range(len(ops)):
        if i in emitted:
            continue
        kind = ops[i][0]
        if kind in PAIRABLE:
            for j in range(i + 1, min(len(ops), i + window)):
                if j in emitted or ops[j][0] != kind:
                    continue
                ch = chain(j, i)
                if ch is None:
                    continue
                for k in ch:
                    out.append(ops[k]); emitted.add(k)
                out.append((kind + "x2", list(ops[i][1]) + list(ops[j][1]), list(ops[i][2]) + list(ops[j][2]), None))
                emitted.add(i); emitted.add(j)
                break
            if i in emitted:
                continue
        out.append(ops[i]); emitted.add(i)
    p.ops = out
    return p


def is_product(kind):
    return kind in ROUTINES or (kind.endswith("x2") and kind[:-2] in ROUTINES)


R392 = 1 << 392
K384 = (1 << 384) % P                # multiplying by it in the 2^392 domain leaves a 2^384-domain value
PTOP = P >> 364                      # top digit of p
INF = 1 << 60
NV, NA = 18, 18                      # VGPR / AGPR blocks of 14
WIN_IN = [0, 1, 2, 3]
FREE_V = list(range(9, 18))                     # blocks no routine overwrites (block 8 is scratch of the Fp2 product)
ALL_V = FREE_V + [8, 3, 2, 1, 0, 4, 6, 5]       # block 7 holds the routines' accumulators: never a home
LADDR, TMP = "v252", "v254"             # v253: -q of reduce, v[254:255]: its 64-bit running sum; v254 also the carry of norm


UNTOUCHED_V = range(102, 108)        # block 7 only holds the product scans' accumulators (v98..v101); the shells' register groups start at
# v108: these six registers are never written by any routine here and stay available to the calling code across a call


def vb(b):
    return 14 * b


ROUTINES = {
    "mul": dict(name="mbls_fp2_mul_d_asm_fn", ins=[0, 1, 2, 3], outs=[5, 6], clob=[4, 5, 6, 7, 8]),
    "sqr": dict(name="mbls_fp2_sqr_d_asm_fn", ins=[0, 1], outs=[5, 6], clob=[2, 3, 4, 5, 6, 7]),
    "mulfp": dict(name="mbls_fp2_mulfp_d_asm_fn", ins=[0, 1, 2], outs=[5, 6], clob=[5, 6, 7]),
    "mulpair": dict(name="mbls_fp_mulpair_d_asm_fn", ins=[0, 1, 2, 3], outs=[5, 6], clob=[5, 6, 7]),     # (a0 b0, a1 b1)
    "mul1": dict(name="mbls_fp_mul1_d_asm_fn", ins=[0, 2], outs=[5], clob=[5, 7]),
    "sqrpair": dict(name="mbls_fp_sqrpair_d_asm_fn", ins=[0, 1], outs=[5, 6], clob=[2, 3, 5, 6, 7]),     # (a0^2, a1^2)
}


class Bound:
    """exact interval bounds of a D-form value: digits 0..12 in [dlo, dhi], top digit in [tlo, thi], integer value in [vlo, vhi]"""
    __slots__ = ("dlo", "dhi", "tlo", "thi", "vlo", "vhi")

    def __init__(self, dlo, dhi, tlo, thi, vlo, vhi):
        self.dlo, self.dhi, self.tlo, self.thi, self.vlo, self.vhi = dlo, dhi, tlo, thi, vlo, vhi

    @staticmethod
    def normalised(vlo, vhi):
        return Bound(0, M28, (vlo >> 364) - 1, (vhi >> 364) + 1, vlo, vhi)

    def mag(self):
        return max(abs(self.dlo), abs(self.dhi), abs(self.tlo), abs(self.thi))

    def fits(self):
        return self.mag() < (1 << 31)

    def __add__(self, o):
        return Bound(self.dlo + o.dlo, self.dhi + o.dhi, self.tlo + o.tlo, self.thi + o.thi, self.vlo + o.vlo, self.vhi + o.vhi)

    def __sub__(self, o):
        return Bound(self.dlo - o.dhi, self.dhi - o.dlo, self.tlo - o.thi, self.thi - o.tlo, self.vlo - o.vhi, self.vhi - o.vlo)

    def neg(self):
        return Bound(-self.dhi, -self.dlo, -self.thi, -self.tlo, -self.vhi, -self.vlo)

    def scaled(self, k):
        assert k > 0
        return Bound(self.dlo * k, self.dhi * k, self.tlo * k, self.thi * k, self.vlo * k, self.vhi * k)

    def union(self, o):
        return Bound(min(self.dlo, o.dlo), max(self.dhi, o.dhi), min(self.tlo, o.tlo), max(self.thi, o.thi), min(self.vlo, o.vlo), max(self.vhi, o.vhi))

    def vabs(self):
        return max(abs(self.vlo), abs(self.vhi))

    def __repr__(self):
        return "B(d[%.2f,%.2f] v[%.2fp,%.2fp])" % (self.dlo / (1 << 28), self.dhi / (1 << 28), self.vlo / P, self.vhi / P)


def product_bound(pairs):
    """result of a Montgomery scan over the listed (a, b) operand bounds: digits normalised, value in (-X, p + X)"""
    X = sum(a.vabs() * b.vabs() for a, b in pairs) // R392 + 2
    return Bound.normalised(-X, P + X)


STATE_IN = None
REDUCED = Bound.normalised(-(P // 2) - (P >> 10), (P // 2) + (P >> 10))      # after `reduce`: the representative nearest to zero


STATE_IN = Bound.normalised(REDUCED.vlo, P)         # a loop-carried value: canonical on entry of a routine, reduced afterwards
PACKED = Bound.normalised(P // 2 - (P >> 10), P + P // 2 + (P >> 10))      # a value parked in LDS by seq_pack_pass
G_IN = Bound(0, M28, 0, M28, 0, (1 << 392) - 1)     # 12 words (2^384 domain) cut as digits of words * 2^8: the 2^392 domain, unreduced
GBASE, GSTRIDE, GADDR, GT0, GT1 = "s[68:69]", "s70", "s[74:75]", "s76", "s77"   # HBM workspace: base (adjusted by the caller so that
# LADDR is the lane offset), bytes between consecutive words of a value, running address, temporaries


GKOFF = "s71"                       # run-time byte offset added to the address of the slots of kind 'gk' (one record of several)
VOFF = "v250"                       # slots of kind 'gv': the record is chosen PER LANE -- this register replaces LADDR as the lane offset


# MBLS_GEN_TIMING_NO_STORES=1: a THROW-AWAY build for timing only -- every workspace store of the generated routines is left out (wrong results, same instruction
# stream otherwise): what the write half of the workspace traffic costs (scripts/dbg/ab_gen.sh). Never committed: the freshness tests compare
# the tracked file with the default generation.
TIMING_NO_STORES = os.environ.get("MBLS_GEN_TIMING_NO_STORES", "0") == "1"


def seq_gaddr(slot, koff=False):
    L = ["s_mul_i32 %s, %s, %d" % (GT0, GSTRIDE, 12 * slot), "s_mul_hi_u32 %s, %s, %d" % (GT1, GSTRIDE, 12 * slot),
         "s_add_u32 s74, s68, %s" % GT0, "s_addc_u32 s75, s69, %s" % GT1]
    if koff:
        L += ["s_add_u32 s74, s74, %s" % GKOFF, "s_addc_u32 s75, s75, 0"]
    return L


def seq_gstore(reg, slot, koff=False, lane=None):
    """12 packed words in reg(0)..reg(11) -> workspace slot (lane: a register that replaces LADDR as the lane's byte offset)"""
    L = seq_gaddr(slot, koff)
    for j in range(12):
        if not TIMING_NO_STORES:
            L.append("global_store_dword %s, %s, %s" % (lane or LADDR, reg(j), GADDR))
        if j < 11:
            L += ["s_add_u32 s74, s74, %s" % GSTRIDE, "s_addc_u32 s75, s75, 0"]
    return L


def seq_gload(reg, slot, aform=True, koff=False, lane=None):
    """12 words of workspace slot `slot` (limb-major: word j of slot s at base + (12 s + j) * stride) into reg(2)..reg(13), then 14
    digits into reg(0)..reg(13): of words * 2^8 (aform: a 2^384-domain value enters the 2^392 domain) or of the words themselves"""
    L = seq_gaddr(slot, koff)
    for j in range(12):
        L.append("global_load_dword %s, %s, %s" % (reg(j + 2), lane or LADDR, GADDR))
        if j < 11:
            L += ["s_add_u32 s74, s74, %s" % GSTRIDE, "s_addc_u32 s75, s75, 0"]
    if aform is None:                                # issue only: the caller waits and converts later (prefetch)
        return L
    L += ["s_waitcnt vmcnt(0)", "s_nop 0"]
    return L + seq_conv(reg, [reg(j + 2) for j in range(12)], aform)


# ---------------------------------------------------------------------------------------------- instruction sequences
def seq_conv(dst, src_regs, aform):
    """14 digits from 12 x 32-bit words held in the registers src_regs (a list of 12 names); aform: digits of src * 2^8 (moves a
    Montgomery-2^384 operand into the 2^392 domain). dst(j) may overlap src when the words sit in dst(2)..dst(13)."""
    L = []
    src = lambda q: src_regs[q]
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


def seq_to32(reg):
    """12 x 32-bit words (in reg(0)..reg(11)) from 14 digits of exactly 28 bits in reg(0)..reg(13), in place"""
    L = []
    for q in range(12):
        j, off = (32 * q) // 28, (32 * q) % 28
        if off == 0:
            L.append("v_lshl_or_b32 %s, %s, 28, %s" % (reg(q), reg(j + 1), reg(j)))
        else:
            L.append("v_lshrrev_b32_e64 %s, %d, %s" % (reg(q), off, reg(j)))
            L.append("v_lshl_or_b32 %s, %s, %d, %s" % (reg(q), reg(j + 1), 28 - off, reg(q)))
    return L


def seq_norm(reg):
    """one sequential carry pass: digits 0..12 into [0, 2^28), the top digit absorbs the rest (signed)"""
    L = []
    for j in range(13):
        L += ["v_ashrrev_i32_e64 %s, 28, %s" % (TMP, reg(j)), "v_and_b32_e64 %s, %s, %s" % (reg(j), reg(j), SMASK28),
              "v_add_u32_e64 %s, %s, %s" % (reg(j + 1), reg(j + 1), TMP)]
    return L


RECIP_PTOP = "0x%08x" % int.from_bytes(__import__("struct").pack(">f", 1.0 / PTOP), "big")


def seq_qpass(reg, setup):
    """ONE carry pass over a signed 64-bit running sum: acc = carry - q p_j + d_j (two v_mad_i64_i32, the second one by the inline
    constant 1), digit = acc mod 2^28, carry = acc >> 28; `setup` leaves -q in NQ. Digits 0..12 come out in [0, 2^28); the 64-bit sum
    makes the pass indifferent to the magnitudes of q and of the incoming digits."""
    A, lo = "v[254:255]", "v254"
    L = list(setup)
    for j in range(14):
        L += ["v_mad_i64_i32 %s, vcc, %s, %s, %s" % (A, NQ, SP28(j), "0" if j == 0 else A), "v_mad_i64_i32 %s, vcc, %s, 1, %s" % (A, reg(j), A)]
        if j < 13:
            L += ["v_and_b32_e64 %s, %s, %s" % (reg(j), lo, SMASK28), "v_ashrrev_i64 %s, 28, %s" % (A, A)]
        else:
            L.append("v_mov_b32_e64 %s, %s" % (reg(13), lo))
    return L


def seq_reduce(reg):
    """value -> the representative nearest to zero, digits normalised: q = rndne(top digit / top digit of p) in floating point (the
    unnormalised lower digits and the lower digits of p move the estimate by < 2e-4)"""
    return seq_qpass(reg, ["v_cvt_f32_i32_e64 %s, %s" % (NQ, reg(13)), "v_mul_f32_e64 %s, %s, %s" % (NQ, RECIP_PTOP_S, NQ), "v_rndne_f32_e64 %s, %s" % (NQ, NQ),
                           "v_cvt_i32_f32_e64 %s, %s" % (NQ, NQ), "v_sub_u32_e64 %s, 0, %s" % (NQ, NQ)])


def seq_pack_pass(reg):
    """any value -> the representative in (0.5 p, 1.5 p), digits normalised (q = rndne(estimate - 1)): positive and below 2^384, which is
    what packing into 12 words needs; one pass, no sign test"""
    return seq_qpass(reg, ["v_cvt_f32_i32_e64 %s, %s" % (NQ, reg(13)), "v_mul_f32_e64 %s, %s, %s" % (NQ, RECIP_PTOP_S, NQ), "v_add_f32_e64 %s, -1.0, %s" % (NQ, NQ),
                           "v_rndne_f32_e64 %s, %s" % (NQ, NQ), "v_cvt_i32_f32_e64 %s, %s" % (NQ, NQ), "v_sub_u32_e64 %s, 0, %s" % (NQ, NQ)])


def seq_canonical(reg):
    """a normalised value in (-p, p) -> [0, p): add p iff the top digit is negative (-q = top digit >> 31 = the sign mask ... as +1)"""
    return seq_qpass(reg, ["v_lshrrev_b32_e64 %s, 31, %s" % (NQ, reg(13))])


NQ = "v253"
RECIP_PTOP_S = "s73"                 # 1 / (top digit of p) as f32, loaded by shell_constants()


def shell_constants():
    return load_constants() + ["s_mov_b32 %s, %s" % (RECIP_PTOP_S, RECIP_PTOP)]


def digits_of(c):
    assert 0 <= c < (1 << 392)
    return [(c >> (28 * j)) & M28 for j in range(13)] + [c >> 364]


def seq_add_const(reg, c):
    """reg += the digits of the non-negative constant c"""
    return ["v_add_u32_e32 %s, 0x%08x, %s" % (reg(j), d, reg(j)) for j, d in enumerate(digits_of(c)) if d]


def lds_read_words(dst_regs, slot_dword):
    """12 packed words of an LDS slot (dword offset slot_dword within the lane's column) into the listed registers"""
    L = []
    for j in range(0, 12, 2):
        o = slot_dword + j
        assert dst_regs[j + 1] == "v%d" % (int(dst_regs[j][1:]) + 1)
        L.append("ds_read2st64_b32 v[%s:%s], %s offset0:%d offset1:%d" % (dst_regs[j][1:], dst_regs[j + 1][1:], LADDR, o, o + 1))
    return L


def lds_write_words(src_regs, slot_dword):
    L = []
    for j in range(0, 12, 2):
        o = slot_dword + j
        L.append("ds_write2st64_b32 %s, %s, %s offset0:%d offset1:%d" % (LADDR, src_regs[j], src_regs[j + 1], o, o + 1))
    return L


def lds_rw_digits(read, blk_or_regs, dword):
    L = []
    for j in range(0, 14, 2):
        o = dword + j
        if read:
            L.append("ds_read2st64_b32 v[%d:%d], %s offset0:%d offset1:%d" % (blk_or_regs + j, blk_or_regs + j + 1, LADDR, o, o + 1))
        else:
            L.append("ds_write2st64_b32 %s, v%d, v%d offset0:%d offset1:%d" % (LADDR, blk_or_regs + j, blk_or_regs + j + 1, o, o + 1))
    return L


WAIT_LDS = ["s_waitcnt lgkmcnt(0)", "s_nop 0"]


# ---------------------------------------------------------------------------------------------- the allocator
class AllocD:
    """Walks a Prog and emits D-form code. Locations: ('v', blk), ('a', blk), ('l', k) = LDS digit slot k."""

    def __init__(self, prog, in_bounds, n_lds=0, lds_base=0, a_pool=None, free_v=None, inline=False):
        self.p = prog
        self.inline = inline                                 # Fp2 products as inlined scans on the blocks the values already live in (do_call_inline)
        self.free_v = list(FREE_V) if free_v is None else list(free_v)     # VGPR blocks outside the routines' window
        self.all_v = self.free_v + ([8] if 8 not in self.free_v else []) + [3, 2, 1, 0, 4, 6, 5]
        self.uses = {}
        for k, (kind, outs, ins, aux) in enumerate(prog.ops):
            for v in ins:
                self.uses.setdefault(v, []).append(k)
        self.loc = dict(prog.init_loc)
        self.at = {l: v for v, l in self.loc.items()}
        self.bound = dict(in_bounds)
        self.out = []
        self.pending_lds = False
        self.n_lds, self.lds_base = n_lds, lds_base
        self.a_pool = list(range(NA)) if a_pool is None else a_pool
        self.stats = dict(vmov=0, acc=0, lds=0, arith=0, norm=0, reduce=0, calls=0, unpack=0)
        # values whose live-in location is a packed LDS slot ('lp', s) keep it as a read-only home: evicting a register copy of
        # such a value costs nothing, fetching it again is 6 LDS reads + the conversion into digits
        self.home = {v: l for v, l in prog.init_loc.items() if l[0] in ("lp", "g", "gd", "gk", "gka", "gv")}
        self.home_bound = {v: in_bounds[v] for v in self.home}
        self.vm = 0                                          # vector-memory operations issued so far (they retire in issue order)
        self.vm_mark = {}                                    # value -> count after the last load of its prefetch

    def next_use(self, v, k):
        for u in self.uses.get(v, ()):
            if u >= k:
                return u
        return INF

    def place(self, v, l):
        old = self.loc.get(v)
        if old is not None and self.at.get(old) == v:
            del self.at[old]
        self.loc[v] = l
        self.at[l] = v

    def release(self, v):
        l = self.loc.pop(v, None)
        if l is not None and self.at.get(l) == v:
            del self.at[l]

    def e(self, s):
        if s.startswith("global_"):
            self.vm += 1
        self.out.append(s)

    def wait_lds(self):
        if self.pending_lds:
            for l in WAIT_LDS:
                self.e(l)
            self.pending_lds = False

    def copy(self, src, dst):
        (sk, sb), (dk, db) = src, dst
        if sk == "v" and dk == "v":
            for j in range(0, 14, 2):
                self.e("v_mov_b64_e64 v[%d:%d], v[%d:%d]" % (vb(db) + j, vb(db) + j + 1, vb(sb) + j, vb(sb) + j + 1))
            self.stats["vmov"] += 7
        elif sk == "v" and dk == "a":
            for j in range(14):
                self.e("v_accvgpr_write_b32 a%d, v%d" % (vb(db) + j, vb(sb) + j))
            self.stats["acc"] += 14
        elif sk == "a" and dk == "v":
            for j in range(14):
                self.e("v_accvgpr_read_b32 v%d, a%d" % (vb(db) + j, vb(sb) + j))
            self.stats["acc"] += 14
        elif sk == "lp" and dk == "v":
            reg = lambda j: "v%d" % (vb(db) + j)
            self.wait_lds()
            for l in lds_read_words([reg(j + 2) for j in range(12)], 12 * sb) + WAIT_LDS + seq_conv(reg, [reg(j + 2) for j in range(12)], False):
                self.e(l)
            self.stats["unpack"] += 30
        elif sk in ("g", "gd", "gk", "gka", "gv") and dk == "v":           # gk / gka: gd / g in the record the run-time offset selects; gv: per lane
            reg = lambda j: "v%d" % (vb(db) + j)
            self.wait_lds()
            for l in seq_gload(reg, sb, aform=(sk in ("g", "gka")), koff=(sk in ("gk", "gka")), lane=(VOFF if sk == "gv" else None)):
                self.e(l)
            self.stats["unpack"] += 60
        elif sk == "l" and dk == "v":
            for l in lds_rw_digits(True, vb(db), self.lds_base + 14 * sb):
                self.e(l)
            self.pending_lds = True
            self.stats["lds"] += 7
        elif sk == "v" and dk == "l":
            for l in lds_rw_digits(False, vb(sb), self.lds_base + 14 * db):
                self.e(l)
            self.stats["lds"] += 7
        else:
            raise ValueError((src, dst))

    def free_block(self, kind, pool, avoid=()):
        for b in pool:
            if (kind, b) not in self.at and b not in avoid and not (kind == "v" and ("vw", b) in self.at):
                return b
        return None

    def drop_prefetch(self, b):
        """a block that only holds prefetched words of a value with a home can be taken at any time"""
        w = self.at.pop(("vw", b), None)
        if w is not None:
            self.loc[w] = self.home[w]

    def alloc_v(self, k, avoid=(), hint=None):
        if hint is not None and ("v", hint) not in self.at and ("vw", hint) not in self.at and hint not in avoid:
            return hint
        b = self.free_block("v", self.all_v, avoid)
        if b is not None:
            return b
        best, bu, pre = None, -1, False                      # Belady: the block whose content is needed last (prefetched words included)
        for blk in self.all_v:
            if blk in avoid:
                continue
            w = self.at.get(("v", blk))
            is_pre = w is None
            if is_pre:
                w = self.at[("vw", blk)]
            u = self.next_use(w, k)
            if u > bu:
                best, bu, pre = blk, u, is_pre
        assert best is not None, "no evictable block"
        if pre:
            self.drop_prefetch(best)
            return best
        w = self.at[("v", best)]
        if bu == INF:
            self.release(w)
            return best
        self.spill(w)
        return best

    def spill(self, w):
        src = self.loc[w]
        ab = self.free_block("a", self.a_pool)
        if ab is not None:                                   # an AGPR round trip (28 moves) beats a second fetch from the home
            self.copy(src, ("a", ab)); self.place(w, ("a", ab)); return
        if w in self.home:                                   # rematerialisable: drop the copy
            self.place(w, self.home[w]); self.bound[w] = self.home_bound[w]; return
        ls = self.free_block("l", range(self.n_lds))
        if ls is not None:
            self.copy(src, ("l", ls)); self.place(w, ("l", ls)); return
        # everything is full: take the AGPR block of the rematerialisable value needed last (it falls back to its workspace home)
        best, bu = None, -1
        for blk in self.a_pool:
            x = self.at.get(("a", blk))
            if x is None or isinstance(x, tuple) or x not in self.home:
                continue
            u = min((q for q in self.uses.get(x, ()) if q >= self.k_now), default=INF)
            if u > bu:
                best, bu = blk, u
        if best is None:
            raise RuntimeError("out of storage")
        x = self.at[("a", best)]
        self.place(x, self.home[x]); self.bound[x] = self.home_bound[x]
        self.copy(src, ("a", best)); self.place(w, ("a", best))

    def to_vgpr(self, v, k, avoid=()):
        l = self.loc[v]
        if l[0] == "v":
            return l[1]
        if l[0] == "vw":                                    # words prefetched into this block: wait for them, cut them into digits
            reg = lambda j: "v%d" % (vb(l[1]) + j)
            younger = min(self.vm - self.vm_mark[v], 63)     # operations issued after its loads may still be in flight
            for x in ["s_waitcnt vmcnt(%d)" % younger, "s_nop 0"] + seq_conv(reg, [reg(j + 2) for j in range(12)], self.home[v][0] in ("g", "gka")):
                self.e(x)
            del self.at[l]
            self.loc[v] = ("v", l[1]); self.at[("v", l[1])] = v
            self.bound[v] = self.home_bound[v]
            return l[1]
        b = self.alloc_v(k, avoid)
        self.copy(l, ("v", b))
        self.place(v, ("v", b))
        return b

    def hint_for(self, d, k):
        if self.inline:
            return None
        u = self.next_use(d, k + 1)
        if u == INF:
            return None
        for j in range(k + 1, u):
            if is_product(self.p.ops[j][0]):
                return None
        kind, outs, ins, aux = self.p.ops[u]
        if kind in ROUTINES:
            return ROUTINES[kind]["ins"][ins.index(d)]
        if is_product(kind):                                 # the first product of a pair sits in the operand slots like any call's
            slots = ROUTINES[kind[:-2]]["ins"]
            return slots[ins.index(d)] if ins.index(d) < len(slots) else None
        return None

    def prefetch(self, k, horizon=int(os.environ.get("MBLS_GEN_PREFETCH_HORIZON", "3")), avoid=()):
        """before a multiplication call: issue the HBM loads of values that the next few operations need and that only live in their
        workspace home, into free blocks outside the routines' window (no eviction: a prefetch must not cost a spill)"""
        calls, j = 0, k + 1
        ins_now = set(self.p.ops[k][2])
        while j < len(self.p.ops) and calls < horizon:
            kind, outs, ins, aux = self.p.ops[j]
            for v in ins:
                if LDS_PREFETCH and self.loc.get(v, ("", 0))[0] == "l" and v not in ins_now:
                    # a value parked in LDS that one of the next operations needs: read it back now, into a free block, so that the read's
                    # latency passes under the product scan instead of in front of the operation (no eviction, like the HBM prefetches)
                    b = self.free_block("v", self.free_v, avoid)
                    if b is not None:
                        self.copy(self.loc[v], ("v", b)); self.place(v, ("v", b))
                    continue
                if self.loc.get(v, ("", 0))[0] in ("g", "gd", "gk", "gka", "gv") and v in self.home:
                    b = self.free_block("v", self.free_v, avoid)
                    if b is None:                            # take the block whose value is needed last, if that is later than this use
                        best, bu = None, j
                        for blk in self.free_v:
                            w = self.at.get(("v", blk))
                            if w is None or w in ins_now or blk in avoid:
                                continue
                            u = self.next_use(w, k)
                            if u > bu:
                                best, bu = blk, u
                        if best is None:
                            return
                        w = self.at[("v", best)]
                        if bu == INF:
                            self.release(w)
                        else:
                            self.spill(w)
                        b = best
                    reg = lambda q, b=b: "v%d" % (vb(b) + q)
                    for x in seq_gload(reg, self.home[v][1], aform=None, koff=(self.home[v][0] in ("gk", "gka")), lane=(VOFF if self.home[v][0] == "gv" else None)):
                        self.e(x)
                    self.vm_mark[v] = self.vm
                    self.loc[v] = ("vw", b); self.at[("vw", b)] = v
                    self.stats["unpack"] += 40
            if is_product(kind):
                calls += 1
            j += 1

    # ---- bound maintenance
    def ensure(self, v, k, ok=None):
        """carry pass on v (in place, in a VGPR block) if ok(bound) does not hold (always when ok is None); returns its block"""
        b = self.to_vgpr(v, k)
        if ok is None or not ok(self.bound[v]):
            self.wait_lds()
            for l in seq_norm(lambda j: "v%d" % (vb(b) + j)):
                self.e(l)
            self.stats["norm"] += 39
            self.bound[v] = Bound.normalised(self.bound[v].vlo, self.bound[v].vhi)
            assert ok is None or ok(self.bound[v]), ("operand cannot be brought inside the limit", self.bound[v])
        return b

    def narrow(self, v, k):
        """bring v's digits down, in place: a carry pass, or -- when the digits are normalised already and it is the value (the top
        digit) that is too wide -- a reduction"""
        if self.bound[v].dlo >= 0 and self.bound[v].dhi <= M28:
            b = self.to_vgpr(v, k)
            self.wait_lds()
            assert self.bound[v].vabs() < (P << 16)
            for l in seq_reduce(lambda j: "v%d" % (vb(b) + j)):
                self.e(l)
            self.stats["reduce"] += 60
            self.bound[v] = REDUCED
        else:
            self.ensure(v, k)

    def run(self):
        for k, (kind, outs, ins, aux) in enumerate(self.p.ops):
            self.k_now = k
            if kind in ("add", "sub", "sel"):
                self.do_arith(k, kind, outs[0], ins[0], ins[1], aux)
            elif kind == "pair":
                a0, b0, a1, b1 = ins
                self.do_arith(k, aux[0], outs[0], a0, b0, None, keep=(a1, b1))
                self.do_arith(k, aux[1], outs[1], a1, b1, None)
            elif kind == "const":
                b = self.alloc_v(k, hint=self.hint_for(outs[0], k))
                for j, d in enumerate(digits_of(aux)):
                    self.e("v_mov_b32_e32 v%d, 0x%08x" % (vb(b) + j, d))
                self.place(outs[0], ("v", b))
                self.bound[outs[0]] = Bound(0, M28, aux >> 364, aux >> 364, aux, aux)
            elif kind in ROUTINES:
                self.do_call(k, kind, outs, ins)
            elif is_product(kind):
                self.do_callx2(k, kind, outs, ins)
            elif kind == "store":
                self.do_store(k, ins[0], aux)
            elif kind == "reduce":
                self.do_reduce(k, outs[0], ins[0])
            elif kind == "norm":
                self.do_norm(k, outs[0], ins[0])
            elif kind == "neg":
                self.do_neg(k, outs[0], ins[0])
            elif kind == "iszero":
                self.do_iszero(k, ins[0], aux)
            elif kind == "mask_orn2":
                self.e("s_orn2_b64 %s, %s, %s" % aux)
            elif kind == "mask_and":
                self.e("s_and_b64 %s, %s, %s" % aux)
            elif kind in ("inv", "pow34"):
                self.do_inv(k, outs[0], ins[0], kind)
            elif kind == "sgn0":
                self.do_sgn0(k, ins[0], ins[1], aux)
            elif kind == "mask_xor":
                self.e("s_xor_b64 %s, %s, %s" % aux)
            elif kind == "mask_or":
                self.e("s_or_b64 %s, %s, %s" % aux)
            elif kind == "scale":
                self.do_scale(k, outs[0], ins[0], aux)
            elif kind == "shadd":
                self.do_arith(k, "shadd", outs[0], ins[0], ins[1], aux)
            elif kind == "storep":
                self.do_storep(k, ins[0], aux)
            elif kind == "keep":
                for v in ins:
                    assert self.loc[v] == self.p.init_loc[v], "pinned value moved"
                continue
            for v in set(ins):
                if self.next_use(v, k + 1) == INF:
                    self.release(v)
        self.wait_lds()
        return self.out

    def do_arith(self, k, kind, d, a, b, aux, keep=()):
        """keep: values that must survive this operation although their last use is the same program op (second half of a pair)"""
        self.to_vgpr(a, k)
        if b != a:
            self.to_vgpr(b, k, avoid=(self.loc[a][1],))
            if self.loc[a][0] != "v":                      # fetching b evicted a
                self.to_vgpr(a, k, avoid=(self.loc[b][1],))
        ba, bb = self.loc[a][1], self.loc[b][1]
        Ba, Bb = self.bound[a], self.bound[b]
        res = {"add": lambda: Ba + Bb, "sub": lambda: Ba - Bb, "sel": lambda: Ba.union(Bb), "shadd": lambda: Ba.scaled(1 << aux) + Bb}[kind]
        for attempt in range(2):                            # renormalise the larger operand(s) first: carry pass, then reduction
            if res().fits():
                break
            for v in sorted({a, b}, key=lambda x: -self.bound[x].mag()):
                self.narrow(v, k)
                Ba, Bb = self.bound[a], self.bound[b]
                if res().fits():
                    break
        assert res().fits(), ("digit overflow", kind, Ba, Bb)
        self.wait_lds()
        avoid = tuple(self.loc[v][1] for v in keep if self.loc[v][0] == "v")
        hint = self.hint_for(d, k)
        bd = None
        if hint is not None and hint not in (ba, bb) and hint not in avoid and ("v", hint) not in self.at:
            bd = hint
        if bd is None:
            for v, blk in ((a, ba), (b, bb)):
                if self.next_use(v, k + 1) == INF and v not in keep and blk not in avoid:
                    bd = blk
                    break
        if bd is None:
            bd = self.alloc_v(k, avoid=(ba, bb) + avoid)
        D, A, B_ = vb(bd), vb(ba), vb(bb)
        if kind == "add":
            for j in range(14):
                self.e("v_add_u32_e64 v%d, v%d, v%d" % (D + j, A + j, B_ + j))
        elif kind == "sub":
            for j in range(14):
                self.e("v_sub_u32_e64 v%d, v%d, v%d" % (D + j, A + j, B_ + j))
        elif kind == "shadd":                               # 2^aux a + b
            for j in range(14):
                self.e("v_lshl_add_u32 v%d, v%d, %d, v%d" % (D + j, A + j, aux, B_ + j))
        else:                                               # sel: mask ? b : a
            for j in range(14):
                self.e("v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (D + j, A + j, B_ + j, aux))
        self.stats["arith"] += 14
        for v in (a, b):
            if self.loc.get(v) == ("v", bd):
                self.release(v)
        self.place(d, ("v", bd))
        self.bound[d] = res()

    def do_reduce(self, k, d, a):
        b = self.to_vgpr(a, k)
        self.wait_lds()
        assert self.bound[a].vabs() < (P << 16), ("reduce: quotient estimate out of range", self.bound[a])
        if self.next_use(a, k + 1) != INF:                  # the operand lives on: work on a copy
            nb = self.alloc_v(k, avoid=(b,))
            self.copy(("v", b), ("v", nb)); b = nb
        else:
            self.release(a)
        for l in seq_reduce(lambda j: "v%d" % (vb(b) + j)):
            self.e(l)
        self.stats["reduce"] += 60
        self.place(d, ("v", b))
        self.bound[d] = REDUCED

    def do_scale(self, k, d, a, c):
        """d = c a for a small positive constant: a shift or one multiplication by an inline constant per digit; where c a would
        leave the 32-bit digit range the constant is split into factors with a carry pass in between (12 = 4 * 3)"""
        b = self.to_vgpr(a, k)
        self.wait_lds()
        hint = self.hint_for(d, k)
        if hint is not None and hint != b and ("v", hint) not in self.at and ("vw", hint) not in self.at:
            bd = hint
        elif self.next_use(a, k + 1) == INF:
            bd = b
        else:
            bd = self.alloc_v(k, avoid=(b,))
        B, src, rest = self.bound[a], b, c
        while rest > 1:
            f = max((g for g in range(2, rest + 1) if rest % g == 0 and B.scaled(g).fits()), default=None)
            if f is None:                                   # no factor fits: carry pass first (in the destination block)
                assert B.mag() > (1 << 28) + 64, ("scale: digit overflow", B, c)
                if src != bd:
                    self.copy(("v", src), ("v", bd)); src = bd
                for l in seq_norm(lambda j: "v%d" % (vb(bd) + j)):
                    self.e(l)
                self.stats["norm"] += 39
                B = Bound.normalised(B.vlo, B.vhi)
                continue
            for j in range(14):
                if f & (f - 1) == 0:
                    self.e("v_lshlrev_b32_e64 v%d, %d, v%d" % (vb(bd) + j, f.bit_length() - 1, vb(src) + j))
                else:
                    self.e("v_mul_lo_u32 v%d, v%d, %d" % (vb(bd) + j, vb(src) + j, f))
            self.stats["arith"] += 14
            B, src, rest = B.scaled(f), bd, rest // f
        if self.loc.get(a) == ("v", bd):
            self.release(a)
        self.place(d, ("v", bd))
        self.bound[d] = B

    def do_norm(self, k, d, a):
        """explicit carry pass (a loop-carried value must meet its live-in bound)"""
        b = self.to_vgpr(a, k)
        self.wait_lds()
        if self.next_use(a, k + 1) != INF:
            nb = self.alloc_v(k, avoid=(b,))
            self.copy(("v", b), ("v", nb)); b = nb
        else:
            self.release(a)
        for l in seq_norm(lambda j: "v%d" % (vb(b) + j)):
            self.e(l)
        self.stats["norm"] += 39
        self.place(d, ("v", b))
        self.bound[d] = Bound.normalised(self.bound[a].vlo, self.bound[a].vhi)

    def do_neg(self, k, d, a):
        b = self.to_vgpr(a, k)
        self.wait_lds()
        hint = self.hint_for(d, k)
        if hint is not None and hint != b and ("v", hint) not in self.at and ("vw", hint) not in self.at:
            bd = hint
        elif self.next_use(a, k + 1) == INF:
            bd = b
        else:
            bd = self.alloc_v(k, avoid=(b,))
        for j in range(14):
            self.e("v_sub_u32_e64 v%d, 0, v%d" % (vb(bd) + j, vb(b) + j))
        self.stats["arith"] += 14
        if self.loc.get(a) == ("v", bd):
            self.release(a)
        self.place(d, ("v", bd))
        self.bound[d] = self.bound[a].neg()

    def do_iszero(self, k, a, mask):
        """the reduced representative is the one nearest to zero, |v| < p: the value is 0 mod p iff every digit is 0"""
        b = self.to_vgpr(a, k)
        self.wait_lds()
        B = self.bound[a]
        red = B.vlo >= REDUCED.vlo and B.vhi <= REDUCED.vhi and B.dlo >= 0 and B.dhi <= M28
        if self.next_use(a, k + 1) != INF and not red:          # the operand lives on unreduced: test a copy
            nb = self.alloc_v(k, avoid=(b,))
            self.copy(("v", b), ("v", nb)); b = nb
        if not red:
            assert B.vabs() < (P << 16)
            for l in seq_reduce(lambda j: "v%d" % (vb(b) + j)):
                self.e(l)
            self.stats["reduce"] += 60
            if self.loc.get(a) == ("v", b):
                self.bound[a] = REDUCED
        self.e("v_or_b32_e64 %s, v%d, v%d" % (TMP, vb(b), vb(b) + 1))
        for j in range(2, 14, 2):
            self.e("v_or3_b32 %s, %s, v%d, v%d" % (TMP, TMP, vb(b) + j, vb(b) + j + 1))
        self.e("v_cmp_eq_u32_e64 %s, 0, %s" % (mask, TMP))
        self.stats["arith"] += 8

    # routines with the 12-word interface of tools/gen_fp_asm.py: (symbol, VGPR blocks, AGPR blocks they overwrite)
    EXT = {"inv": ("mbls_fp_inv_gcd_asm_fn", range(0, 7), range(0, 0)),              # v0..v84, no AGPRs
           "pow34": ("mbls_fp_pow_pm3d4_asm_fn", range(0, 7), range(0, 16))}         # v0..v84, a0..a223 (the window table)

    def do_sgn0(self, k, x0, x1, masks):
        """x0, x1: plain integers mod p in digit form -> canonical digits -> s0 | (z0 & s1) as a lane mask"""
        mask, tmp = masks
        regs = []
        for x in (x0, x1):
            b = self.to_vgpr(x, k, avoid=tuple(regs))
            self.wait_lds()
            assert self.next_use(x, k + 1) == INF
            reg = lambda j, b=b: "v%d" % (vb(b) + j)
            B = self.bound[x]
            red = B.vlo >= REDUCED.vlo and B.vhi <= REDUCED.vhi and B.dlo >= 0 and B.dhi <= M28
            for l in ([] if red else seq_reduce(reg)) + seq_canonical(reg):
                self.e(l)
            regs.append(b)
        b0, b1 = regs
        self.e("v_or_b32_e64 %s, v%d, v%d" % (TMP, vb(b0), vb(b0) + 1))
        for j in range(2, 14, 2):
            self.e("v_or3_b32 %s, %s, v%d, v%d" % (TMP, TMP, vb(b0) + j, vb(b0) + j + 1))
        self.e("v_cmp_eq_u32_e64 %s, 0, %s" % (tmp, TMP))                               # c0 = 0
        self.e("v_and_b32_e64 %s, 1, v%d" % (TMP, vb(b1)))
        self.e("v_cmp_ne_u32_e64 %s, 0, %s" % (mask, TMP))                              # parity of c1
        self.e("s_and_b64 %s, %s, %s" % (mask, mask, tmp))
        self.e("v_and_b32_e64 %s, 1, v%d" % (TMP, vb(b0)))
        self.e("v_cmp_ne_u32_e64 %s, 0, %s" % (tmp, TMP))                               # parity of c0
        self.e("s_or_b64 %s, %s, %s" % (mask, mask, tmp))
        self.stats["arith"] += 14
        self.release(x0); self.release(x1)

    def do_inv(self, k, d, a, kind="inv"):
        """a: reduced value x * 2^384 (digit form) -> canonical words in v0..v11 -> the inversion routine -> words of x^-1 * 2^384,
        cut into digits of the 2^392 domain. Everything live leaves the registers that routine uses."""
        b = self.to_vgpr(a, k)
        self.wait_lds()
        B = self.bound[a]
        assert B.vlo >= REDUCED.vlo and B.vhi <= REDUCED.vhi and B.dlo >= 0 and B.dhi <= M28, ("inv: operand not reduced", B)
        assert self.next_use(a, k + 1) == INF
        reg = lambda j: "v%d" % (vb(b) + j)
        for l in seq_canonical(reg) + seq_to32(reg):
            self.e(l)
        for blk in range(NV):                                 # prefetched words are rematerialisable: drop them
            self.drop_prefetch(blk)
        sym, CLOB_V, CLOB_A = self.EXT[kind]
        safe_v = [x for x in self.free_v if x not in CLOB_V and x != b]
        safe_a = [x for x in self.a_pool if x not in CLOB_A]

        def park(w, src):
            """live value w, currently readable in VGPR block src: to a place the exponentiation leaves alone"""
            for blk in safe_v:
                if ("v", blk) not in self.at:
                    self.copy(("v", src), ("v", blk)); self.place(w, ("v", blk)); return
            for blk in safe_a:
                if ("a", blk) not in self.at:
                    self.copy(("v", src), ("a", blk)); self.place(w, ("a", blk)); return
            if w in self.home:
                self.place(w, self.home[w]); self.bound[w] = self.home_bound[w]; return
            ls = self.free_block("l", range(self.n_lds))
            if ls is None:
                raise RuntimeError("out of storage around the inversion")
            self.copy(("v", src), ("l", ls)); self.place(w, ("l", ls))
        for blk in CLOB_V:
            w = self.at.get(("v", blk))
            if w is None or w == a:
                continue
            if self.next_use(w, k + 1) == INF:
                self.release(w)
            else:
                park(w, blk)
        tmp = next(x for x in CLOB_V if x != b and ("v", x) not in self.at)
        for blk in CLOB_A:
            w = self.at.get(("a", blk))
            if w is None:
                continue
            assert not isinstance(w, tuple), "a stored output sits in the exponentiation's table registers"
            if self.next_use(w, k + 1) == INF:
                self.release(w)
            else:
                self.copy(("a", blk), ("v", tmp)); park(w, tmp)
        self.release(a)
        self.wait_lds()
        if b != 0:
            for j in range(12):
                self.e("v_mov_b32_e64 v%d, v%d" % (j, vb(b) + j))
        self.e("s_waitcnt vmcnt(0)")
        self.e("CALL " + sym)
        self.stats["calls"] += 1
        dst = lambda j: "v%d" % (vb(1) + j)
        for l in seq_conv(dst, ["v%d" % j for j in range(12)], True):
            self.e(l)
        self.place(d, ("v", 1))
        self.bound[d] = G_IN

    def do_storep(self, k, a, slot):
        """value -> workspace slot, packed (representative in (0.5 p, 1.5 p), 12 words, 2^392 domain); values that still call this
        slot home are fetched first"""
        kind = "gd"
        if isinstance(slot, tuple):                          # ('k', j): slot j of the record selected by the run-time offset
            kind, slot = "gk", slot[1]
        for w in [w for w, h in self.home.items() if h == (kind, slot)]:
            if w != a and self.next_use(w, k + 1) != INF and self.loc.get(w) == (kind, slot):
                self.to_vgpr(w, k)
            del self.home[w]
        b = self.to_vgpr(a, k)
        self.wait_lds()
        if self.next_use(a, k + 1) != INF:
            nb = self.alloc_v(k, avoid=(b,))
            self.copy(("v", b), ("v", nb)); b = nb
        else:
            self.release(a)
        assert self.bound[a].vabs() < (P << 16)
        reg = lambda j: "v%d" % (vb(b) + j)
        for l in seq_pack_pass(reg) + seq_to32(reg) + seq_gstore(reg, slot, koff=(kind == "gk")):
            self.e(l)
        self.stats["reduce"] += 62 + 21 + 38
        if self.next_use(a, k + 1) != INF:                  # from now on the slot is a home of the value: register copies of it can be
            self.home[a] = (kind, slot)                     # dropped and fetched again (same value, the packed representative)
            self.home_bound[a] = PACKED

    def call_limits_ok(self, kind, B):
        m = [x.mag() for x in B]
        if kind == "mul":                                   # Karatsuba on column sums: the third product is (a0 - a1)(b1 - b0)
            da, db = (B[0] - B[1]), (B[3] - B[2])
            return (da.fits() and db.fits() and column_ok([(m[0], m[2]), (m[1], m[3])]) and column_ok([(m[0], m[3]), (m[1], m[2])])
                    and column_ok([(da.mag(), db.mag())]))
        if kind == "sqr":
            s = B[0] + B[1]; dd = B[0] - B[1]
            return s.fits() and dd.fits() and 2 * m[1] < (1 << 31) and column_ok([(s.mag(), dd.mag())]) and column_ok([(m[0], 2 * m[1])])
        if kind == "mulpair":
            return column_ok([(m[0], m[2])]) and column_ok([(m[1], m[3])])
        if kind == "sqrpair":                               # cross products against the doubled digits: the same column sums as a product
            return 2 * m[0] < (1 << 31) and 2 * m[1] < (1 << 31) and column_ok([(m[0], m[0])]) and column_ok([(m[1], m[1])])
        if kind == "mul1":
            return column_ok([(m[0], m[1])])
        return column_ok([(m[0], m[2])]) and column_ok([(m[1], m[2])])

    # kinds that can be inlined: (scratch blocks, body) -- block order of the bodies: operands, scratch, results (see gen_fpd_asm.py)
    INLINE = {"mul": (2, lambda o, t, r: fp2_mul_d_body((o[0], o[1], o[2], o[3], t[0], r[0], r[1], t[1]))),
              "sqr": (3, lambda o, t, r: fp2_sqr_d_body((o[0], o[1], t[0], t[1], t[2], r[0], r[1]))),
              "mulfp": (0, lambda o, t, r: fp2_mulfp_d_body((o[0], o[1], o[2], r[0], r[1])))}

    def call_bounds(self, kind, B):
        if kind == "mul":
            return [product_bound([(B[0], B[2]), (B[1], B[3])]), product_bound([(B[0], B[3]), (B[1], B[2])])]
        if kind == "sqr":
            return [product_bound([(B[0] + B[1], B[0] - B[1])]), product_bound([(B[0], B[1] + B[1])])]
        if kind == "mulpair":
            return [product_bound([(B[0], B[2])]), product_bound([(B[1], B[3])])]
        if kind == "sqrpair":
            return [product_bound([(B[0], B[0])]), product_bound([(B[1], B[1])])]
        if kind == "mul1":
            return [product_bound([(B[0], B[1])])]
        return [product_bound([(B[0], B[2])]), product_bound([(B[1], B[2])])]

    def do_call_inline(self, k, kind, outs, ins):
        """The product scan emitted HERE, on the blocks its operands already occupy and into result blocks of the allocator's choice: no copies
        into a fixed operand window, no call. Operands, scratch and results are pairwise distinct blocks (the operands are read until the
        last column); block 7 holds the accumulators."""
        nscratch, bodyf = self.INLINE[kind]
        guard = 0
        while not self.call_limits_ok(kind, [self.bound[v] for v in ins]):
            v = max(ins, key=lambda x: (self.bound[x].mag(), -ins.index(x)))
            before = self.bound[v].mag()
            self.narrow(v, k)
            guard += 1
            assert self.bound[v].mag() < before or guard < 8, "cannot meet the routine's input limits"
        uniq = list(dict.fromkeys(ins))
        for attempt in range(4):                             # fetching one operand may evict another: repeat until all are in VGPR blocks
            for v in uniq:
                if self.loc[v][0] != "v":
                    self.to_vgpr(v, k, avoid=tuple(self.loc[w][1] for w in uniq if w != v and self.loc[w][0] == "v"))
            if all(self.loc[v][0] == "v" for v in uniq):
                break
        assert all(self.loc[v][0] == "v" for v in uniq), "operands do not fit the register blocks"
        opb = [self.loc[v][1] for v in ins]
        taken = list(dict.fromkeys(opb))
        extra = []
        for _ in range(nscratch + 2):
            b = self.alloc_v(k, avoid=tuple(taken))
            assert ("v", b) not in self.at and ("vw", b) not in self.at
            taken.append(b); extra.append(b)
        assert all(self.loc[v] == ("v", b) for v, b in zip(ins, opb)), "an operand moved while the result blocks were made"
        self.wait_lds()
        self.prefetch(k, avoid=tuple(taken))
        reg = lambda b: (lambda j: "v%d" % (vb(b) + j))
        for l in bodyf([reg(b) for b in opb], [reg(b) for b in extra[:nscratch]], [reg(b) for b in extra[nscratch:]]):
            self.e(l)
        self.stats["calls"] += 1
        self.stats["inlined"] = self.stats.get("inlined", 0) + 1
        ob = self.call_bounds(kind, [self.bound[v] for v in ins])
        for i, o in enumerate(outs):
            self.place(o, ("v", extra[nscratch + i]))
            self.bound[o] = ob[i]

    def do_call(self, k, kind, outs, ins):
        if self.inline and kind in self.INLINE:
            return self.do_call_inline(k, kind, outs, ins)
        R = ROUTINES[kind]
        slots = R["ins"]
        # 1. operand limits: renormalise the largest operand until the product columns fit
        guard = 0
        while not self.call_limits_ok(kind, [self.bound[v] for v in ins]):
            v = max(ins, key=lambda x: (self.bound[x].mag(), -ins.index(x)))
            before = self.bound[v].mag()
            self.narrow(v, k)
            guard += 1
            assert self.bound[v].mag() < before or guard < 8, "cannot meet the routine's input limits"
        want = {slots[i]: ins[i] for i in range(len(ins))}
        clob = set(R["clob"])
        # 2. vacate what the routine overwrites, and operand slots that hold something else
        for s in sorted(clob | set(slots)):
            w = self.at.get(("v", s))
            if w is None or (s in want and want[s] == w):
                continue
            live = self.next_use(w, k) != INF
            if not live:
                self.release(w)
                continue
            u = self.next_use(w, k + 1 if w not in ins else k)
            nk = self.p.ops[u][0] if u != INF else None
            busy = clob | set(slots)
            if w not in ins and nk is not None and is_product(nk) and self.free_block("a", self.a_pool) is not None:
                self.spill(w)                                    # next consumed as a call operand: waits in an AGPR for free
            else:
                b = self.alloc_v(k, avoid=busy)
                self.copy(("v", s), ("v", b)); self.place(w, ("v", b))
        # 3. operands into their slots. The routines preserve blocks 0..3, so an operand that lives on simply stays tracked there.
        # (operands that come from LDS first: their reads are in flight while the others are moved)
        for s, v in sorted(want.items(), key=lambda sv: (0 if self.loc[sv[1]][0] == "l" else 1, sv[0])):
            if self.loc[v] == ("v", s):
                continue
            if self.loc[v][0] == "vw":
                self.to_vgpr(v, k)
            src = self.loc[v]
            self.copy(src, ("v", s))
            if src[0] != "v" or self.next_use(v, k + 1) == INF or src[1] in clob:
                self.place(v, ("v", s))                          # moved (its old home is free again, or about to be overwritten) ...
            # ... or copied: the original stays where it was, the slot copy is untracked and dies with the next operand load
        for s in slots[:len(ins)]:
            w = self.at.get(("v", s))
            assert w is None or w == want[s], "operand slot holds a foreign live value"
        self.wait_lds()
        self.prefetch(k)
        self.e("CALL " + R["name"])
        self.stats["calls"] += 1
        for s in clob:
            w = self.at.get(("v", s))
            if w is not None:
                assert self.next_use(w, k + 1) == INF, "live value in a clobbered block across a call"
                self.release(w)
        ob = self.call_bounds(kind, [self.bound[v] for v in ins])
        for i, o in enumerate(outs):
            self.place(o, ("v", R["outs"][i]))
            self.bound[o] = ob[i]

    def do_callx2(self, k, kind, outs, ins):
        """Two independent products of one kind as ONE call on a lane pair (see pair_products): the first product's operands go into the
        routine's slots as for any call; the second product's operands overwrite them on the odd lanes (exec = PAIR_ROLE); after the call each
        lane fetches its neighbour's result (DPP) and the odd lanes swap the two, so that blocks 5, 6 hold the first product and two more blocks
        the second one on BOTH lanes. Everything outside the masked moves runs identically on both lanes: their states never differ."""
        base = kind[:-2]
        R = ROUTINES[base]
        slots = R["ins"]
        nin = len(slots)
        insA, insB, outsA, outsB = ins[:nin], ins[nin:], outs[:2], outs[2:]
        for half in (insA, insB):
            guard = 0
            while not self.call_limits_ok(base, [self.bound[v] for v in half]):
                v = max(half, key=lambda x: (self.bound[x].mag(), -half.index(x)))
                before = self.bound[v].mag()
                self.narrow(v, k)
                guard += 1
                assert self.bound[v].mag() < before or guard < 8, "cannot meet the routine's input limits"
        want = {slots[i]: insA[i] for i in range(nin)}
        clob = set(R["clob"])
        busy = clob | set(slots)
        for s_ in sorted(busy):                              # vacate what the routine overwrites, and operand slots that hold something else
            w = self.at.get(("v", s_))
            if w is None or (s_ in want and want[s_] == w):
                continue
            if self.next_use(w, k) == INF:
                self.release(w)
                continue
            u = self.next_use(w, k + 1 if w not in ins else k)
            nk = self.p.ops[u][0] if u != INF else None
            if w not in ins and nk is not None and is_product(nk) and self.free_block("a", self.a_pool) is not None:
                self.spill(w)
            else:
                b = self.alloc_v(k, avoid=busy)
                self.copy(("v", s_), ("v", b)); self.place(w, ("v", b))
        overwritten = {slots[i] for i in range(nin) if insB[i] != insA[i]}       # slots that hold the second product's operand on the odd lanes
        for s_, v in want.items():                           # the first product's operands into their slots
            keeps = self.next_use(v, k + 1) != INF or v in insB                   # the value is needed again (the second product counts)
            if self.loc[v] == ("v", s_):
                if s_ in overwritten and keeps:              # it must survive somewhere the odd lanes do not overwrite
                    self.spill(v)
                continue
            if self.loc[v][0] == "vw":
                self.to_vgpr(v, k)
            src = self.loc[v]
            self.copy(src, ("v", s_))
            if not (s_ in overwritten and keeps) and (src[0] != "v" or self.next_use(v, k + 1) == INF or src[1] in clob):
                self.place(v, ("v", s_))
        for v in insB:                                       # the second product's operands must be readable without side effects: registers or LDS
            if self.loc[v][0] not in ("v", "a", "l"):
                self.to_vgpr(v, k, avoid=tuple(busy))
        self.wait_lds()
        moves = [(slots[i], insB[i]) for i in range(nin) if insB[i] != insA[i]]
        if moves:
            self.e("s_mov_b64 exec, %s" % PAIR_ROLE)
            for s_, v in moves:
                src = self.loc[v]
                assert src != ("v", s_) and src[0] in ("v", "a", "l"), (src, s_)
                self.copy(src, ("v", s_))                    # (the odd lanes execute it; stats and pending-LDS bookkeeping as for any move)
            self.wait_lds()
            self.e("s_mov_b64 exec, %s" % PAIR_EXEC)
        for s_ in slots[:nin]:
            w = self.at.get(("v", s_))
            assert w is None or (w == want[s_] and not (s_ in overwritten and self.next_use(w, k + 1) != INF)), "operand slot holds a foreign or a half-overwritten live value"
        self.prefetch(k, avoid=tuple(busy))
        self.e("CALL " + R["name"])
        self.stats["calls"] += 1
        self.stats["pairs"] = self.stats.get("pairs", 0) + 1
        for s_ in clob:
            w = self.at.get(("v", s_))
            if w is not None:
                assert self.next_use(w, k + 1) == INF, "live value in a clobbered block across a call"
                self.release(w)
        for s_, v in moves:                                  # ... and the slots the odd lanes overwrote hold nothing that may be read again
            w = self.at.get(("v", s_))
            if w is not None:
                assert self.next_use(w, k + 1) == INF
                self.release(w)
        o5, o6 = R["outs"]
        self.at[("v", o5)] = ("tmp", 0); self.at[("v", o6)] = ("tmp", 1)          # keep the allocator off the results while it finds two more blocks
        p0 = self.alloc_v(k, avoid=tuple(busy))
        self.at[("v", p0)] = ("tmp", 2)
        p1 = self.alloc_v(k, avoid=tuple(busy) + (p0,))
        for key in (("v", o5), ("v", o6), ("v", p0)):
            del self.at[key]
        self.wait_lds()
        self.e("s_nop 1")                                    # the routine's last writes to blocks 5, 6 and the cross-lane reads below
        for (dst, src) in ((p0, o5), (p1, o6)):
            for j in range(14):
                self.e("v_mov_b32_dpp v%d, v%d %s" % (vb(dst) + j, vb(src) + j, DPP_SWAP))
        self.e("s_mov_b64 exec, %s" % PAIR_ROLE)
        for (x, y) in ((o5, p0), (o6, p1)):
            for j in range(14):
                self.e("v_swap_b32 v%d, v%d" % (vb(x) + j, vb(y) + j))
        self.e("s_mov_b64 exec, %s" % PAIR_EXEC)
        self.stats["vmov"] += 56
        obA = self.call_bounds(base, [self.bound[v] for v in insA])
        obB = self.call_bounds(base, [self.bound[v] for v in insB])
        for o, blk, bd in ((outsA[0], o5, obA[0]), (outsA[1], o6, obA[1]), (outsB[0], p0, obB[0]), (outsB[1], p1, obB[1])):
            self.place(o, ("v", blk))
            self.bound[o] = bd

    def do_store(self, k, a, dst):
        """dst: ('a', blk) home; the stored value must satisfy the routine's live-in bound (checked by the caller of run())"""
        w = self.at.get(dst)
        if w is not None and w != a:
            if not (isinstance(w, tuple)) and self.next_use(w, k + 1) != INF:
                b = self.alloc_v(k, avoid=(self.loc[a][1],) if self.loc[a][0] == "v" else ())
                self.copy(dst, ("v", b)); self.place(w, ("v", b))
                self.wait_lds()
            elif not isinstance(w, tuple):
                self.release(w)
        if self.loc[a] != dst:
            b = self.to_vgpr(a, k)
            self.wait_lds()
            self.copy(("v", b), dst)
        self.stored = getattr(self, "stored", {})
        self.stored[dst] = self.bound[a]
        if self.next_use(a, k + 1) == INF:
            self.release(a)
        self.at[dst] = ("stored", a)


# ---------------------------------------------------------------------------------------------- programs
def prog_reduce(p, a):
    d = p.new(); p.ops.append(("reduce", [d], [a], None)); return d


A_HOME = lambda i: ("a", i)


def cyc_sqr_formula(p, z, out):
    """Granger-Scott squaring in the cyclotomic subgroup (formulas of fp12_cyc_sqr in mbls_tower.h) on six Fp2 values in tower order
    (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2); out(e, value) receives result e (unreduced) as soon as it exists"""
    z0, z4, z3, z2, z1, z5 = z

    def fp4_sqr(a, b):
        t0 = p.sqr2(a); t1 = p.sqr2(b)
        c0 = p.add2(p.mul_xi2(t1), t0)
        s = p.sqr2(p.add2(a, b))
        c1 = p.sub2(p.sub2(s, t0), t1)
        return c0, c1
    t0, t1 = fp4_sqr(z0, z1)
    out(0, p.shadd2(p.sub2(t0, z0), 1, t0))          # 2 (t - z) + t
    out(4, p.shadd2(p.add2(t1, z1), 1, t1))
    t0, t1 = fp4_sqr(z2, z3)
    t2, t3 = fp4_sqr(z4, z5)
    out(1, p.shadd2(p.sub2(t0, z4), 1, t0))
    out(5, p.shadd2(p.add2(t1, z5), 1, t1))
    x = p.mul_xi2(t3)
    out(3, p.shadd2(p.add2(x, z2), 1, x))
    out(2, p.shadd2(p.sub2(t2, z3), 1, t2))


def prog_cyc_sqr_d():
    """one cyclotomic squaring, state in AGPR blocks 0..11 in tower order, in place (the loop body of the powers by |x|). Every output
    re-enters the next squaring through the linear terms 3 t -+ 2 z, so each is reduced."""
    p = Prog()
    z = [(p.live_in(A_HOME(2 * e)), p.live_in(A_HOME(2 * e + 1))) for e in range(6)]

    def out(e, v):
        for i in range(2):
            p.store(prog_reduce(p, v[i]), A_HOME(2 * e + i))
    cyc_sqr_formula(p, z, out)
    return p


def build_cyc_sqr_d():
    p = prog_cyc_sqr_d()
    inb = {v: STATE_IN for v in p.init_loc}          # first round: canonical values from LDS; later rounds: reduced ones
    al = AllocD(p, inb, n_lds=11, lds_base=0, a_pool=list(range(12, NA)))
    body = al.run()
    for dst, B in al.stored.items():
        assert B.vlo >= REDUCED.vlo and B.vhi <= REDUCED.vhi and B.dlo >= 0 and B.dhi <= M28, (dst, B)
    return body, al.stats


def wrap_loop_d(lines, count_sgpr, prologue, epilogue):
    back = ["s_sub_u32 %s, %s, 1" % (count_sgpr, count_sgpr), "s_cmp_lg_u32 %s, 0" % count_sgpr, "s_cbranch_scc0 2f",
            "s_getpc_b64 s[66:67]", "3:", "s_sub_u32 s66, s66, 3b-1b", "s_subb_u32 s67, s67, 0", "s_setpc_b64 s[66:67]", "2:"]
    return ["s_mov_b64 s[36:37], s[30:31]"] + shell_constants() + list(prologue) + [".p2align 6", "1:"] + lines + back + list(epilogue) + ["s_mov_b64 s[30:31], s[36:37]"]


def expand_calls_d(lines):
    out = []
    for l in lines:
        if l.startswith("CALL "):
            sym = l.split()[1]
            out += ["s_getpc_b64 s[66:67]", "s_add_u32 s66, s66, %s@rel32@lo+4" % sym, "s_addc_u32 s67, s67, %s@rel32@hi+12" % sym,
                    "s_swappc_b64 s[30:31], s[66:67]"]
        else:
            out.append(l)
    return out


# ---------------------------------------------------------------------------------------------- the Miller loop
G1_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
ONE_D = R392 % P                                   # 1, x, -x, y ... as 2^392-domain constants
NPX0_D = (P - G1_X) * R392 % P                     # pair 0 of a verification is (signature, -G1): the line needs -px and py of -G1
PY0_D = (P - G1_Y) * R392 % P
F_HOME = [("a", i) for i in range(12)]             # the Miller value between iterations, tower order
P1_HOME = [("g", i) for i in range(3)]             # -px, py, pz^3 of the second pair's G1 argument: workspace slots 0..2 (the caller
# writes them over the aggregate key), 2^384 domain
SKIP_MASK = ["s[48:49]", "s[54:55]"]               # lanes whose pair contributes 1 (a member is infinity)
T_SLOT = lambda k, e, i: 31 + 6 * k + 2 * e + i    # workspace slot of coordinate e (x, y, z) / component i of running point k (packed, 2^392 domain)
# workspace slots (mbls_lanes.h) of the fixed points Q_k: the signature's affine x, y (slots 3..6) and H(m) in homogeneous form, which
# the caller writes over the Jacobian H in slots 7..12 before the loop
Q_SLOT = [[(3, 4), (5, 6), None], [(7, 8), (9, 10), (11, 12)]]
F_IN = Bound.normalised(-16 * P, 16 * P)          # a coefficient of the Miller value between rounds: carry-normalised, a few p wide


def t_live_in(p, k):
    return [(p.live_in(("gd", T_SLOT(k, e, 0))), p.live_in(("gd", T_SLOT(k, e, 1)))) for e in range(3)]


def prog_norm(p, a):
    d = p.new(); p.ops.append(("norm", [d], [a], None)); return d


def storep2(p, a, k, e):
    for i in range(2):
        p.ops.append(("storep", [], [a[i]], T_SLOT(k, e, i)))


def line_into_f(p, f, c0, c2, c3, k):
    m = SKIP_MASK[k]
    c0 = (p.sel(m, c0[0], p.const(ONE_D)), p.sel(m, c0[1], p.const(0)))
    c2 = (p.sel(m, c2[0], p.const(0)), p.sel(m, c2[1], p.const(0)))
    c3 = (p.sel(m, c3[0], p.const(0)), p.sel(m, c3[1], p.const(0)))
    return p.mul12_line(f, c0, c2, c3)


def masked_line(p, c0, c2, c3, k):
    m = SKIP_MASK[k]
    return ((p.sel(m, c0[0], p.const(ONE_D)), p.sel(m, c0[1], p.const(0))),
            (p.sel(m, c2[0], p.const(0)), p.sel(m, c2[1], p.const(0))),
            (p.sel(m, c3[0], p.const(0)), p.sel(m, c3[1], p.const(0))))


def mul_lines(p, la, lb):
    """(a0 + a2 w^2 + a3 w^3)(b0 + b2 w^2 + b3 w^3) in tower form, 6 Fp2 multiplications:
    L0 = (a0 b0 + xi a3 b3, a0 b2 + a2 b0, a2 b2), L1 = (0, a0 b3 + a3 b0, a2 b3 + a3 b2)"""
    a0, a2, a3 = la
    b0, b2, b3 = lb
    t00, t22, t33 = p.mul2(a0, b0), p.mul2(a2, b2), p.mul2(a3, b3)
    t02 = p.sub2(p.sub2(p.mul2(p.add2(a0, a2), p.add2(b0, b2)), t00), t22)
    t03 = p.sub2(p.sub2(p.mul2(p.add2(a0, a3), p.add2(b0, b3)), t00), t33)
    t23 = p.sub2(p.sub2(p.mul2(p.add2(a2, a3), p.add2(b2, b3)), t22), t33)
    return [p.add2(t00, p.mul_xi2(t33)), t02, t22], [None, t03, t23]


def mul6_by_0yz(p, x, y1, y2):
    """(x0 + x1 v + x2 v^2)(y1 v + y2 v^2), 5 Fp2 multiplications"""
    t11, t22 = p.mul2(x[1], y1), p.mul2(x[2], y2)
    cross = p.sub2(p.sub2(p.mul2(p.add2(x[1], x[2]), p.add2(y1, y2)), t11), t22)       # x1 y2 + x2 y1
    return [p.mul_xi2(cross), p.add2(p.mul2(x[0], y1), p.mul_xi2(t22)), p.add2(p.mul2(x[0], y2), t11)]


def mul12_by_lines(p, f, L0, L1):
    """f * (L0 + L1 w) with L1 = (0, y1, y2): 6 + 5 + 6 = 17 Fp2 multiplications"""
    a, b = f
    t0 = p.mul6(a, L0)
    t1 = mul6_by_0yz(p, b, L1[1], L1[2])
    s = [L0[0], p.add2(L0[1], L1[1]), p.add2(L0[2], L1[2])]
    c1 = p.mul6(p.add6(a, b), s)
    c1 = p.sub6(p.sub6(c1, t0), t1)
    return (p.add6(t0, p.mul_v6(t1)), c1)


def f_live_in(p):
    fl = [p.live_in(h) for h in F_HOME]
    return ([(fl[0], fl[1]), (fl[2], fl[3]), (fl[4], fl[5])], [(fl[6], fl[7]), (fl[8], fl[9]), (fl[10], fl[11])])


def f_store(p, f):
    for v, h in zip([x for half in f for c in half for x in c], F_HOME):
        p.store(prog_norm(p, v), h)


def dbl_point_and_line(p, k, p1):
    """T_k <- 2 T_k (written back to its packed slots) and the coefficients (c0, c2, c3) of the tangent line at P_k
    (formulas of miller_dbl_step in mbls_pairing.h)"""
    Tx, Ty, Tz = t_live_in(p, k)
    B = p.sqr2(Ty); C = p.sqr2(Tz)
    E = p.mul12_2(p.mul_xi2(C))
    F = p.mul3_2(E)
    X2 = p.sqr2(Tx)
    YZ2 = p.sub2(p.sub2(p.sqr2(p.add2(Ty, Tz)), B), C)       # 2 Y Z = (Y + Z)^2 - Y^2 - Z^2: a squaring instead of a product
    c0 = p.sub2(B, E)
    if k == 0:
        c2 = p.mulfp2(p.mul3_2(X2), p.const(NPX0_D))
        c3 = p.mulfp2(YZ2, p.const(PY0_D))
    else:
        c0 = p.mulfp2(c0, p1[2])
        c2 = p.mulfp2(p.mul3_2(X2), p1[0])
        c3 = p.mulfp2(YZ2, p1[1])
    x3 = p.dbl2(p.mul2(p.mul2(Tx, Ty), p.sub2(B, F)))
    y3 = p.sub2(p.sqr2(p.add2(B, F)), p.mul12_2(p.sqr2(E)))
    z3 = p.mul4_2(p.mul2(B, YZ2))
    for e, v in enumerate((x3, y3, z3)):
        storep2(p, v, k, e)
    return c0, c2, c3


def prog_miller_first_d(pairs=(0, 1)):
    """The FIRST doubling iteration of the loop: f = 1, so f^2 = 1 and f * lines = the lines themselves -- the 12 products of the squaring and
    the 17 of the product with the lines do not exist. A line c0 + c2 w^2 + c3 w^3 is (c0, c2, 0) + (0, c3, 0) w in the tower (w^2 = v)."""
    p = Prog()
    p1 = [p.live_in(h) for h in P1_HOME]
    lines = [masked_line(p, *dbl_point_and_line(p, k, p1), k) for k in pairs]
    zero2 = lambda: (p.const(0), p.const(0))
    if len(pairs) == 2:
        L0, L1 = mul_lines(p, lines[0], lines[1])
        f = (L0, [zero2(), L1[1], L1[2]])
    else:
        red2 = lambda a: (prog_reduce(p, a[0]), prog_reduce(p, a[1]))     # products with the unreduced G1 argument are a few dozen p wide
        c0, c2, c3 = (red2(c) for c in lines[0])
        f = ([c0, c2, zero2()], [zero2(), c3, zero2()])
    f_store(p, f)
    return p


def add_point_and_line(p, k, p1):
    """T_k <- T_k + Q_k and the coefficients of the line through them at P_k (formulas of miller_add_step in mbls_pairing.h). Q_k comes from the
    HBM workspace (2^384 domain, converted on the way in); pair 0's Q is affine and its G1 argument the constant -G1."""
    Tx, Ty, Tz = t_live_in(p, k)
    Q = [None if sl is None else (p.live_in(("g", sl[0])), p.live_in(("g", sl[1]))) for sl in Q_SLOT[k]]
    Qx, Qy, Qz = Q
    if Qz is None:                                   # affine Q: Z2 = 1
        y1z2, x1z2, z1z2 = Ty, Tx, Tz
    else:
        y1z2, x1z2, z1z2 = p.mul2(Ty, Qz), p.mul2(Tx, Qz), p.mul2(Tz, Qz)
    u = p.sub2(p.mul2(Qy, Tz), y1z2)
    v = p.sub2(p.mul2(Qx, Tz), x1z2)
    c0 = p.sub2(p.mul2(u, Qx), p.mul2(v, Qy))
    if Qz is None:
        c2 = p.mulfp2(u, p.const(NPX0_D))            # -u Z2 xP with Z2 = 1, -xP a constant
        c3 = p.mulfp2(v, p.const(PY0_D))
    else:
        c0 = p.mulfp2(c0, p1[2])
        c2 = p.mulfp2(p.mul2(u, Qz), p1[0])
        c3 = p.mulfp2(p.mul2(v, Qz), p1[1])
    uu = p.sqr2(u); vv = p.sqr2(v)
    vvv = p.mul2(v, vv); R = p.mul2(vv, x1z2)
    A = p.sub2(p.sub2(p.mul2(uu, z1z2), vvv), p.dbl2(R))
    x3 = p.mul2(v, A)
    y3 = p.sub2(p.mul2(u, p.sub2(R, A)), p.mul2(vvv, y1z2))
    z3 = p.mul2(vvv, z1z2)
    for e, w in enumerate((x3, y3, z3)):
        storep2(p, w, k, e)
    return c0, c2, c3


def prog_miller_add_both_d():
    """The addition steps of BOTH pairs of a verification as one body: the two lines are multiplied together first (6 + 17 products instead
    of 13 + 13, like the doubling body) and f is fetched and stored once."""
    p = Prog()
    f = f_live_in(p)
    p1 = [p.live_in(h) for h in P1_HOME]
    lines = {k: masked_line(p, *add_point_and_line(p, k, p1), k) for k in (1, 0)}      # pair 1 first: its step holds more temporaries
    L0, L1 = mul_lines(p, lines[0], lines[1])
    f_store(p, mul12_by_lines(p, f, L0, L1))
    return p


def prog_miller_dbl_d(pairs=(0, 1)):
    """One doubling iteration of the Miller loop for the two pairs of a verification (or, pairs = (1,), for a single general pair): f <- f^2, then for each pair T <- 2T and f <- f * line
    (formulas of miller_dbl_step / fp12_sqr / fp12_mul_line in mbls_pairing.h / mbls_tower.h). f in AGPR homes; the running points and
    the second G1 argument live in the HBM workspace as packed words (fetched when needed, the points written back at the end of their
    step), which leaves the whole LDS allocation to the allocator as spill space."""
    p = Prog()
    f = f_live_in(p)
    p1 = [p.live_in(h) for h in P1_HOME]
    merge = MERGE_LINES and len(pairs) == 2
    if not merge:
        f = p.sqr12(f)
    elif F2_FIRST:                                   # f^2 needs nothing from the workspace: the fetches of the running points travel under it
        f2 = p.sqr12(f)
    lines = []
    for k in pairs:
        c0, c2, c3 = dbl_point_and_line(p, k, p1)
        if merge:
            lines.append(masked_line(p, c0, c2, c3, k))
        else:
            f = line_into_f(p, f, c0, c2, c3, k)
    if merge:                                        # the two lines are multiplied together first: 6 + 17 instead of 13 + 13 products
        L0, L1 = mul_lines(p, lines[0], lines[1])
        f = mul12_by_lines(p, f2 if F2_FIRST else p.sqr12(f), L0, L1)
    f_store(p, f)
    return p


MERGE_LINES = True
LDS_PREFETCH = os.environ.get("MBLS_GEN_LDS_PREFETCH", "1") == "1"
F2_FIRST = os.environ.get("MBLS_GEN_F2_FIRST", "0") == "1"


def prog_miller_add_d(k):
    """The addition step T_k <- T_k + Q_k, f <- f * line for ONE pair (formulas of miller_add_step in mbls_pairing.h). Q_k comes from the
    HBM workspace (2^384 domain, converted on the way in); pair 0's Q is affine and its G1 argument the constant -G1."""
    p = Prog()
    f = f_live_in(p)
    p1 = [p.live_in(h) for h in P1_HOME]
    c0, c2, c3 = add_point_and_line(p, k, p1)
    f = line_into_f(p, f, c0, c2, c3, k)
    f_store(p, f)
    return p


def prog_miller(which, pairs=(0, 1), pair_mode=False):
    p = {"dbl": lambda: prog_miller_dbl_d(pairs), "first": lambda: prog_miller_first_d(pairs), "add01": prog_miller_add_both_d}.get(
        which, lambda: prog_miller_add_d(which))()
    return pair_products(p) if pair_mode else p


def build_miller(which, pairs=(0, 1), pair_mode=False):
    """pair_mode: the body for lane PAIRS (two lanes per Miller loop): independent products of one kind share a call (pair_products)"""
    p = prog_miller(which, pairs, pair_mode)
    inb = {}
    for v, l in p.init_loc.items():
        inb[v] = PACKED if l[0] == "gd" else G_IN if l[0] == "g" else F_IN
    al = AllocD(p, inb, n_lds=11, lds_base=0, a_pool=list(range(NA)), inline=(which in INLINE_BODIES))
    body = al.run()
    for dst, B in al.stored.items():
        assert B.dlo >= F_IN.dlo and B.dhi <= F_IN.dhi and B.vlo >= F_IN.vlo and B.vhi <= F_IN.vhi and B.tlo >= F_IN.tlo and B.thi <= F_IN.thi, (dst, B)
    return body, al.stats


# Bodies whose Fp2 products are inlined scans. NONE for the Miller loop: measured on the MI355X, the inlined doubling body (82.7 k instructions
# = 660 KB of straight-line code per iteration instead of 16.9 k of glue around three 8-10 KB product routines) issues 2.6 % fewer
# instructions and runs 12 % SLOWER (k_miller 11.26 -> 12.62 ms, 2.00 -> 2.30 ns per instruction): the called routines stay resident in the
# 64 KB instruction cache and 80 % of the executed instructions hit there, while straight-line code streams from L2 at ~2.4 ns per
# instruction. Inlining pays only where the whole loop fits the cache (the compressed squaring of the final exponentiation: 57 KB).
INLINE_BODIES = ()
F_OUT = [108 + 12 * i for i in range(12)]          # register groups (12 words each) in which the Miller routine returns f
RUNS = [1, 2, 3, 9, 32, 16]                        # doubling iterations between the additions: |x| = 0xd201000000010000, bits 62..0


def far_back(label):
    """jump to an earlier numeric label from anywhere (the bodies exceed the reach of s_cbranch)"""
    return ["s_getpc_b64 s[66:67]", "7:", "s_sub_u32 s66, s66, 7b-%db" % label, "s_subb_u32 s67, s67, 0", "s_setpc_b64 s[66:67]"]


def f_out_epilogue(ret="s[36:37]"):
    """the Fp12 in AGPR blocks 0..11 (D-form, 2^392 domain) -> canonical words of the 2^384 domain in the register groups F_OUT"""
    epi = ["s_waitcnt vmcnt(0)"]                    # nothing may still be in flight into registers when the routine returns
    B0, B1, B2, B5, B6 = (lambda j: "v%d" % j), (lambda j: "v%d" % (14 + j)), (lambda j: "v%d" % (28 + j)), (lambda j: "v%d" % (70 + j)), (lambda j: "v%d" % (84 + j))
    epi += ["v_mov_b32_e32 %s, 0x%08x" % (B2(j), dgt) for j, dgt in enumerate(digits_of(K384))]
    for i in range(6):                              # pairs of coefficients: (x 2^392)(2^384) / 2^392 = x 2^384, then the canonical words
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B0(j), vb(2 * i) + j) for j in range(14)]
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B1(j), vb(2 * i + 1) + j) for j in range(14)]
        epi += ["CALL mbls_fp2_mulfp_d_asm_fn"]
        for h, B in ((0, B5), (1, B6)):
            epi += seq_reduce(B) + seq_canonical(B) + seq_to32(B)
            epi += ["v_mov_b32_e64 v%d, %s" % (F_OUT[2 * i + h] + j, B(j)) for j in range(12)]
    epi += ["s_mov_b64 s[30:31], %s" % ret]
    return epi


def pair_prologue():
    """pair mode: PAIR_EXEC = the lanes that entered the routine, PAIR_ROLE = the odd ones among them"""
    return ["s_mov_b64 %s, exec" % PAIR_EXEC, "v_mbcnt_lo_u32_b32 v254, -1, 0", "v_mbcnt_hi_u32_b32 v254, -1, v254", "v_and_b32_e64 v254, 1, v254",
            "v_cmp_ne_u32_e64 %s, 0, v254" % PAIR_ROLE]


def miller_loop_d_routine(pairs=(0, 1), pair_mode=False):
    """The whole two-pair Miller loop of a verification as ONE routine: f = prod_k f_{|x|,Q_k}(P_k) (the caller conjugates); with
    pairs = (1,) the loop of a single general pair (Q_1, P_1) for the n-pairing paths (pair 0's slots are then unused).
    In:  v252 LDS byte address of the lane's column (11 spill slots), v253 skip flags (bit k: pair k contributes 1),
         s[68:69] workspace base adjusted so that v252 is the lane offset, s70 bytes between consecutive words of a value;
         workspace slots 0..2 = (-px, py, pz^3) of pair 1, 3..6 = Q0 (affine x, y), 7..12 = Q1 (homogeneous x, y, z), 2^384 domain.
    Out: f in v108..v251 (twelve groups of 12 words, tower order, canonical, 2^384 domain). Workspace slots 31..42 are scratch."""
    dbl, st_dbl = build_miller("dbl", pairs, pair_mode)
    first, st_first = build_miller("first", pairs, pair_mode)
    two = len(pairs) == 2
    add, st_add = build_miller("add01", pairs, pair_mode) if two else build_miller(1, pairs, pair_mode)
    add_name = "add01" if two else "add1"
    W = lambda j: "v%d" % (vb(8) + j)              # work block of the shell
    pro = ["s_mov_b64 s[36:37], s[30:31]", "s_waitcnt vmcnt(0)"] + shell_constants() + (pair_prologue() if pair_mode else [])
    pro += ["v_and_b32_e64 v254, 1, v253", "v_cmp_ne_u32_e64 %s, 0, v254" % SKIP_MASK[0], "v_and_b32_e64 v254, 2, v253", "v_cmp_ne_u32_e64 %s, 0, v254" % SKIP_MASK[1]]
    for k in pairs:                                 # T_k = Q_k, into the packed 2^392-domain slots
        for e in range(3):
            for i in range(2):
                sl = Q_SLOT[k][e]
                if sl is None:
                    pro += ["v_mov_b32_e32 %s, 0x%08x" % (W(j), dgt) for j, dgt in enumerate(digits_of(ONE_D if i == 0 else 0))]
                else:
                    pro += seq_gload(W, sl[i], True)
                pro += seq_pack_pass(W) + seq_to32(W) + seq_gstore(W, T_SLOT(k, e, i))
    # control flow (every far jump goes backwards): the first iteration (f = 1: the lines ARE f), then
    #   5: addition step(s); phase += 1;  4: RUNS[phase] doubling iterations;  phase == 5 ? done : back to 5
    assert RUNS[0] == 1
    pro += ["s_waitcnt vmcnt(0)", "s_mov_b32 s78, 0"]
    main = expand_calls_d(first)
    main += [".p2align 6", "5:"] + expand_calls_d(add) + ["s_add_u32 s78, s78, 1"]
    main += ["s_mov_b32 s39, %d" % RUNS[5]]
    for ph in range(1, 5):
        main += ["s_cmp_eq_u32 s78, %d" % ph, "s_cselect_b32 s39, %d, s39" % RUNS[ph]]
    main += [".p2align 6", "1:"] + expand_calls_d(dbl)
    main += ["s_sub_u32 s39, s39, 1", "s_cmp_lg_u32 s39, 0", "s_cbranch_scc0 2f"] + far_back(1) + ["2:"]
    main += ["s_cmp_eq_u32 s78, 5", "s_cbranch_scc1 9f"] + far_back(5) + ["9:"]
    epi = f_out_epilogue()
    pieces = dict(pro=pro, first=first, dbl=dbl, epi=epi)
    pieces[add_name] = add
    return pro + main + expand_calls_d(epi), pieces, {"first": st_first, "dbl": st_dbl, add_name: st_add}


# ---------------------------------------------------------------------------------------------- the final exponentiation as ONE routine
# f^(3 (p^12 - 1) / r) with the hard part (x-1)^2 (x+p) (x^2+p^2-1) + 3 (Hayashida-Hayasaka-Teruya), exactly the sequence of final_exp in
# mbls_pairing.h: easy part (one Fp12 inversion, Frobenius^2), five powers by |x| (conjugated: x < 0) with Fp12 products in between.
# Between the bodies below the running value lives in AGPR blocks 0..11; three Fp12 temporaries live in the workspace as packed
# 2^392-domain words: Y (the base of the running power), M (f after the easy part) and B.
FEXP_IN_SLOT, Y_SLOT, M_SLOT, B_SLOT = 13, 31, 0, 13     # the input f is dead when B is first written


def f2mul_py(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2pow_py(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2mul_py(r, a)
        a = f2mul_py(a, a); e >>= 1
    return r


FROB_W = [f2pow_py((1, 1), k * (P - 1) // 6) for k in range(6)]                 # (c w^k)^p = conj(c) FROB_W[k] w^k
FROB2_W = [(g[0] * g[0] + g[1] * g[1]) % P for g in FROB_W]                      # (c w^k)^(p^2) = c FROB2_W[k] w^k, in Fp
D392 = lambda x: x * R392 % P


def six(l):
    return ([(l[0], l[1]), (l[2], l[3]), (l[4], l[5])], [(l[6], l[7]), (l[8], l[9]), (l[10], l[11])])


def flat12(f):
    return [x for h in f for c in h for x in c]


def mul12(p, a, b, conj_b=False):
    """(a0 + a1 w)(b0 +- b1 w), Karatsuba over Fp6: 18 Fp2 multiplications; conj_b folds the conjugation of b into the signs"""
    t0 = None
    if not conj_b:
        c1 = p.mul6(p.add6(a[0], a[1]), p.add6(b[0], b[1]))
        t0 = p.mul6(a[0], b[0]); t1 = p.mul6(a[1], b[1])
        return (p.add6(t0, p.mul_v6(t1)), p.sub6(p.sub6(c1, t0), t1))
    c1 = p.mul6(p.add6(a[0], a[1]), p.sub6(b[0], b[1]))
    t0 = p.mul6(a[0], b[0]); t1 = p.mul6(a[1], b[1])
    return (p.sub6(t0, p.mul_v6(t1)), p.add6(p.sub6(c1, t0), t1))


def sqr6(p, a):
    """(a0 + a1 v + a2 v^2)^2, Chung-Hasan: 2 multiplications + 3 squarings"""
    s0, s4 = p.sqr2(a[0]), p.sqr2(a[2])
    s1 = p.dbl2(p.mul2(a[0], a[1])); s3 = p.dbl2(p.mul2(a[1], a[2]))
    s2 = p.sqr2(p.add2(p.sub2(a[0], a[1]), a[2]))
    return [p.add2(s0, p.mul_xi2(s3)), p.add2(s1, p.mul_xi2(s4)), p.sub2(p.sub2(p.add2(p.add2(s1, s2), s3), s0), s4)]


def frob12(p, a):
    out = ([None] * 3, [None] * 3)
    for k in range(6):
        c = p.conj2(a[k & 1][k >> 1])
        out[k & 1][k >> 1] = c if k == 0 else p.mul2(c, (p.const(D392(FROB_W[k][0])), p.const(D392(FROB_W[k][1]))))
    return out


def frob12_2(p, a):
    out = ([None] * 3, [None] * 3)
    for k in range(6):
        c = a[k & 1][k >> 1]
        g = FROB2_W[k]
        out[k & 1][k >> 1] = c if g == 1 else p.neg2(c) if g == P - 1 else p.mulfp2(c, p.const(D392(g)))
    return out


def acc_live_in(p):
    return six([p.live_in(("a", i)) for i in range(12)])


def gd_live_in(p, slot):
    return six([p.live_in(("gd", slot + i)) for i in range(12)])


def acc_store(p, f, also=()):
    """the Fp12 f -> the AGPR state (reduced) and, packed, to the workspace temporaries listed in `also`"""
    for i, v in enumerate(flat12(f)):
        r = prog_reduce(p, v)
        for slot in also:
            p.ops.append(("storep", [], [r], slot + i))
        p.store(r, ("a", i))


def park12(p, f, slot):
    """write the Fp12 to workspace slots slot..slot+11: its twelve values can be dropped from the registers and fetched again"""
    for i, v in enumerate(flat12(f)):
        p.ops.append(("storep", [], [v], slot + i))
    return f


def prog_fexp_easy():
    """m = f^((p^6 - 1)(p^2 + 1)): f from the workspace (2^384 domain); m -> state, Y and M"""
    p = Prog()
    # the words from the workspace are a 2^392-domain value up to 2^11 p wide: reduce each coefficient once and give it a packed home
    f = park12(p, six([prog_reduce(p, p.live_in(("g", FEXP_IN_SLOT + i))) for i in range(12)]), M_SLOT)
    a0, a1 = f
    t = p.sub6(sqr6(p, a0), p.mul_v6(sqr6(p, a1)))                     # a0^2 - v a1^2 (the norm to Fp6)
    c0, c1, c2 = t
    A = p.sub2(p.sqr2(c0), p.mul_xi2(p.mul2(c1, c2)))
    B = p.sub2(p.mul_xi2(p.sqr2(c2)), p.mul2(c0, c1))
    C = p.sub2(p.sqr2(c1), p.mul2(c0, c2))
    F = p.add2(p.mul_xi2(p.add2(p.mul2(c2, B), p.mul2(c1, C))), p.mul2(c0, A))      # the norm to Fp2
    Fc = p.conj2(F)
    n = p.mul2(F, (p.neg(p.neg(F[0])), Fc[1]))[0]                        # F0^2 + F1^2 (the routine wants four distinct operands)
    Fi = p.mulfp2(Fc, p.inv(n))
    T = [p.mul2(A, Fi), p.mul2(B, Fi), p.mul2(C, Fi)]                    # 1 / t
    fi = (p.mul6(a0, T), p.neg6(p.mul6(a1, T)))                          # 1 / f
    t = park12(p, mul12(p, fi, f, conj_b=True), Y_SLOT)                  # f^(p^6 - 1) = conj(f) / f
    m = mul12(p, park12(p, frob12_2(p, t), M_SLOT), t)
    acc_store(p, m, also=(Y_SLOT, M_SLOT))
    return p


# A power by |x| = 2^63 + 2^62 + 2^60 + 2^57 + 2^48 + 2^16 in the cyclotomic subgroup with Karabina's compressed squarings: the
# Granger-Scott formulas for the coefficients (z2, z3, z4, z5) do not involve (z0, z1), so a chain of squarings carries only those
# four (6 Fp2 squarings instead of 9); the six powers y^(2^i) the exponent needs are saved compressed, decompressed together
#     z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2)     [z2 = 0:  z1 = 2 z4 z5 / z3],     z0 = xi (2 z1^2 + z2 z5 - 3 z3 z4) + 1
# with ONE Fp inversion for the six denominators (Montgomery's trick on their norms), and multiplied. A zero denominator (the element
# 1: all four coefficients zero) is replaced by 1 inside the trick and its inverse by 0 afterwards, which decompresses to 1.
K_SLOT, K_REC = 49, 10                              # six records: z2, z3, z4, z5 (8 Fp) + the inverse of the denominator (2 Fp)
POW_RUNS = [16, 32, 9, 3, 2, 1]                     # squarings before each save: y^(2^16), ^(2^48), ^(2^57), ^(2^60), ^(2^62), ^(2^63)
C_IDX = [6, 7, 4, 5, 2, 3, 10, 11]                  # z2, z3, z4, z5 in the flat tower order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2)
ZMASK = ["s[48:49]", "s[50:51]", "s[52:53]", "s[54:55]", "s[84:85]", "s[86:87]"]
TMASK, TMASK2 = "s[88:89]", "s[90:91]"


# Homes of the compressed state (z2, z3, z4, z5) between the bodies of a power (pstart, csqr, psave): z2 and z3 stay in VGPR blocks 14..17 --
# the squaring body reads them first and writes them last, so they never travel through the AGPRs (112 of the body's 7 182 instructions) --,
# z4 and z5 in AGPR blocks 4..7. The three bodies withhold blocks 14..17 from their allocators.
CSTATE_HOME = [("v", 17), ("v", 16), ("v", 15), ("v", 14), ("a", 4), ("a", 5), ("a", 6), ("a", 7)]
CSTATE_BODIES = ("pstart", "csqr", "psave")
CSTATE_FREE_V = [b for b in FREE_V if b not in (14, 15, 16, 17)]


def prog_fexp_pstart():
    """compressed state <- the base Y"""
    p = Prog()
    y = [p.live_in(("gd", Y_SLOT + i)) for i in range(12)]
    for i, idx in enumerate(C_IDX):
        p.store(prog_reduce(p, y[idx]), CSTATE_HOME[i])
    return p


def prog_fexp_csqr():
    """one compressed squaring: (z2, z3, z4, z5) in AGPR blocks 0..7, in place"""
    p = Prog()
    l = [p.live_in(CSTATE_HOME[i]) for i in range(8)]
    z2, z3, z4, z5 = (l[0], l[1]), (l[2], l[3]), (l[4], l[5]), (l[6], l[7])

    def fp4_sqr(a, b):
        t0 = p.sqr2(a); t1 = p.sqr2(b)
        return p.add2(p.mul_xi2(t1), t0), p.sub2(p.sub2(p.sqr2(p.add2(a, b)), t0), t1)

    def out(base, v):
        for i in range(2):
            p.store(prog_reduce(p, v[i]), CSTATE_HOME[base + i])
    t0, t1 = fp4_sqr(z2, z3)
    out(4, p.shadd2(p.sub2(t0, z4), 1, t0))
    out(6, p.shadd2(p.add2(t1, z5), 1, t1))
    t2, t3 = fp4_sqr(z4, z5)
    x = p.mul_xi2(t3)
    out(0, p.shadd2(p.add2(x, z2), 1, x))
    out(2, p.shadd2(p.sub2(t2, z3), 1, t2))
    return p


def prog_fexp_psave():
    """the compressed state -> the record selected by the run-time offset (the state stays)"""
    p = Prog()
    l = [p.live_in(CSTATE_HOME[i]) for i in range(8)]
    for i, v in enumerate(l):
        p.ops.append(("storep", [], [v], ("k", K_SLOT + i)))         # record 0's slot: the run-time offset selects the record
    for i, v in enumerate(l):
        p.store(v, CSTATE_HOME[i])
    return p


def prog_fexp_pinv():
    """the inverses of the six decompression denominators (4 z2, or z3 where z2 = 0) into slots 8, 9 of their records"""
    p = Prog()
    dens, norms = [], []
    for i in range(6):
        l = [p.live_in(("gd", K_SLOT + K_REC * i + j)) for j in range(4)]
        z2, z3 = (l[0], l[1]), (l[2], l[3])
        p.iszero2(z2, TMASK, TMASK2)
        den = p.sel2(TMASK, p.scale2(z2, 4), z3)
        sq = p.mulpair(den[0], den[0], den[1], den[1])
        n = p.add(sq[0], sq[1])
        p.iszero(n, ZMASK[i])
        dens.append(den); norms.append(p.sel(ZMASK[i], n, p.const(ONE_D)))
    pre = [norms[0]]
    for i in range(1, 6):
        pre.append(p.mul1(pre[-1], norms[i]))
    inv = p.inv(pre[5])
    for i in range(5, -1, -1):
        if i:
            invn, inv = p.mulpair(inv, pre[i - 1], inv, norms[i])
        else:
            invn = inv
        invn = p.sel(ZMASK[i], invn, p.const(0))
        r = p.mulfp2(p.conj2(dens[i]), invn)
        for j in range(2):
            p.ops.append(("storep", [], [r[j]], K_SLOT + K_REC * i + 8 + j))
    return p


def decompress12(p):
    """the Fp12 of the record selected by the run-time offset, in six-form"""
    l = [p.live_in(("gk", K_SLOT + j)) for j in range(K_REC)]
    z2, z3, z4, z5, iden = (l[0], l[1]), (l[2], l[3]), (l[4], l[5]), (l[6], l[7]), (l[8], l[9])
    p.iszero2(z2, TMASK, TMASK2)
    n_gen = p.sub2(p.add2(p.mul_xi2(p.sqr2(z5)), p.scale2(p.sqr2(z4), 3)), p.scale2(z3, 2))
    n_alt = p.scale2(p.mul2(z4, z5), 2)
    z1 = p.mul2(p.sel2(TMASK, n_gen, n_alt), iden)
    t = p.sub2(p.add2(p.scale2(p.sqr2(z1), 2), p.mul2(z2, z5)), p.scale2(p.mul2(z3, z4), 3))
    t = p.mul_xi2(t)
    z0 = (p.add(t[0], p.const(ONE_D)), t[1])
    return ([z0, z4, z3], [z2, z1, z5])


def prog_fexp_pfirst():
    """state <- the decompressed first record"""
    p = Prog()
    acc_store(p, decompress12(p))
    return p


def prog_fexp_pmul():
    """state <- state * (the decompressed record)"""
    p = Prog()
    a = acc_live_in(p)
    acc_store(p, mul12(p, a, decompress12(p)))
    return p


def prog_fexp_step_conj():
    """after the first and the second power: state <- conj(state) * conj(Y), also the next base (x < 0: the power is conjugated)"""
    p = Prog()
    a = acc_live_in(p)
    r = mul12(p, a, gd_live_in(p, Y_SLOT))                               # conj(a) conj(y) = conj(a y)
    acc_store(p, (r[0], p.neg6(r[1])), also=(Y_SLOT,))
    return p


def prog_fexp_step_frob():
    """after the third power: b = conj(state) * frob(Y) -> state, Y, B"""
    p = Prog()
    a = acc_live_in(p)
    r = mul12(p, park12(p, frob12(p, gd_live_in(p, Y_SLOT)), B_SLOT), a, conj_b=True)
    acc_store(p, r, also=(Y_SLOT, B_SLOT))
    return p


def prog_fexp_step_base():
    """after the fourth power: conj(state) is the base of the fifth"""
    p = Prog()
    a = acc_live_in(p)
    acc_store(p, (a[0], p.neg6(a[1])), also=(Y_SLOT,))
    return p


def prog_fexp_tail():
    """after the fifth power: conj(state) * frob^2(B) * conj(B) * M^3"""
    p = Prog()
    a = acc_live_in(p)
    b = gd_live_in(p, B_SLOT)
    c = mul12(p, park12(p, frob12_2(p, b), Y_SLOT), a, conj_b=True)
    c = park12(p, mul12(p, c, b, conj_b=True), B_SLOT)
    m = gd_live_in(p, M_SLOT)
    m2 = [None] * 6
    cyc_sqr_formula(p, m[0] + m[1], lambda e, v: m2.__setitem__(e, v))
    c = mul12(p, c, park12(p, mul12(p, (m2[:3], m2[3:]), m), Y_SLOT))
    acc_store(p, c)
    return p


FEXP_BODIES = dict(easy=prog_fexp_easy, pstart=prog_fexp_pstart, csqr=prog_fexp_csqr, psave=prog_fexp_psave, pinv=prog_fexp_pinv,
                   pfirst=prog_fexp_pfirst, pmul=prog_fexp_pmul, step_conj=prog_fexp_step_conj, step_frob=prog_fexp_step_frob,
                   step_base=prog_fexp_step_base, tail=prog_fexp_tail)


FEXP_INLINE = ("csqr",)                          # the loop body that runs 315 times per item and fits the instruction cache with its products inlined (57 KB)


def build_fexp(which, pair_mode=False):
    """pair_mode: the body for lane pairs -- independent products of one kind share a call (pair_products)"""
    p = FEXP_BODIES[which]()
    if pair_mode:
        pair_products(p)
    inb = {v: (STATE_IN if l[0] in ("a", "v") else PACKED if l[0] in ("gd", "gk") else G_IN) for v, l in p.init_loc.items()}
    al = AllocD(p, inb, n_lds=11, lds_base=0, a_pool=list(range(NA)), inline=(which in FEXP_INLINE),
                free_v=(CSTATE_FREE_V if which in CSTATE_BODIES else None))
    body = al.run()
    for dst, B in getattr(al, "stored", {}).items():
        assert B.vlo >= STATE_IN.vlo and B.vhi <= STATE_IN.vhi and B.dhi <= M28, (dst, B)
    return body, al.stats


def far_fwd(label):
    return ["s_getpc_b64 s[66:67]", "7:", "s_add_u32 s66, s66, %df-7b" % label, "s_addc_u32 s67, s67, 0", "s_setpc_b64 s[66:67]"]


# ---- two lanes per item (batches that fill at most half of the SIMDs: k_final2). The two lanes of an item (lanes 2 i, 2 i + 1 of a wave: the
# caller gives both the same LDS column and the same workspace item) run the whole routine side by side on identical values -- except the
# 315 compressed squarings, 57 % of the routine: the two Fp4 squarings of a squaring do not depend on each other, so each lane does ONE.
# A lane in role A holds (u, v) = (z2, z3), a lane in role B holds (z4, z5); with (first, second) = fp4_sqr(u, v) = (xi v^2 + u^2, 2 u v)
#     role A:  z4' = 3 first - 2 z4,       z5' = 3 second + 2 z5      (the partner's u, v)
#     role B:  z2' = 3 xi second + 2 z2,   z3' = 3 first - 2 z3
# so after the step the lane that was A holds (z4', z5') and IS role B, and vice versa: the roles swap every squaring (ROLE = the lanes in
# role B), the partner's (u, v) come through DPP (quad_perm [1,0,3,2]), and one instruction stream serves both roles through selections.
# 3.7 k instructions per squaring and lane instead of 7.1 k. State: u in VGPR blocks 0, 1, v in 2, 3; everything is explicit (no allocator).
ROLE = "s[92:93]"
DPP_SWAP = "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
VBLK = lambda b: (lambda j: "v%d" % (vb(b) + j))
CS2_CU, CS2_CV, CS2_LANE = "v84", "v85", "v86"          # block 6: scratch of the squarings, free between them


def csqr2_body():
    """one compressed squaring on a pair of lanes (see above); roles swap at the end"""
    U0, U1, V0, V1 = VBLK(0), VBLK(1), VBLK(2), VBLK(3)
    B = [VBLK(i) for i in range(18)]
    L = []
    L += fp2_sqr_d_body((U0, U1, B[4], B[5], B[6], B[8], B[9]))                  # t0 = u^2
    L += fp2_sqr_d_body((V0, V1, B[4], B[5], B[6], B[10], B[11]))                # t1 = v^2
    for j in range(14):                                                          # w = u + v (digits below 2^29: inside the squaring's limits)
        L += ["v_add_u32_e64 %s, %s, %s" % (B[12](j), U0(j), V0(j)), "v_add_u32_e64 %s, %s, %s" % (B[13](j), U1(j), V1(j))]
    L += fp2_sqr_d_body((B[12], B[13], B[4], B[5], B[6], B[14], B[15]))          # (u + v)^2
    for j in range(14):                                                          # second = (u + v)^2 - t0 - t1 (in blocks 14, 15)
        L += ["v_sub_u32_e64 %s, %s, %s" % (B[14](j), B[14](j), B[8](j)), "v_sub_u32_e64 %s, %s, %s" % (B[15](j), B[15](j), B[9](j)),
              "v_sub_u32_e64 %s, %s, %s" % (B[14](j), B[14](j), B[10](j)), "v_sub_u32_e64 %s, %s, %s" % (B[15](j), B[15](j), B[11](j))]
    for j in range(14):                                                          # first = xi t1 + t0 (blocks 12, 13); xi (c0, c1) = (c0 - c1, c0 + c1)
        L += ["v_sub_u32_e64 %s, %s, %s" % (B[12](j), B[10](j), B[11](j)), "v_add_u32_e64 %s, %s, %s" % (B[12](j), B[12](j), B[8](j)),
              "v_add3_u32 %s, %s, %s, %s" % (B[13](j), B[10](j), B[11](j), B[9](j))]
    for j in range(14):                                                          # xi second (blocks 4, 5)
        L += ["v_sub_u32_e64 %s, %s, %s" % (B[4](j), B[14](j), B[15](j)), "v_add_u32_e64 %s, %s, %s" % (B[5](j), B[14](j), B[15](j))]
    for (dst, src) in ((B[16], U0), (B[17], U1), (B[10], V0), (B[11], V1)):      # the partner's u, v (both lanes still hold their old state here)
        L += ["v_mov_b32_dpp %s, %s %s" % (dst(j), src(j), DPP_SWAP) for j in range(14)]
    for j in range(14):                                                          # what is tripled: role B ? (xi second, first) : (first, second)
        L += ["v_cndmask_b32_e64 %s, %s, %s, %s" % (B[8](j), B[12](j), B[4](j), ROLE), "v_cndmask_b32_e64 %s, %s, %s, %s" % (B[9](j), B[13](j), B[5](j), ROLE)]
    for j in range(14):
        L += ["v_cndmask_b32_e64 %s, %s, %s, %s" % (B[14](j), B[14](j), B[12](j), ROLE), "v_cndmask_b32_e64 %s, %s, %s, %s" % (B[15](j), B[15](j), B[13](j), ROLE)]
    L += ["v_cndmask_b32_e64 %s, -2, 2, %s" % (CS2_CU, ROLE), "v_cndmask_b32_e64 %s, 2, -2, %s" % (CS2_CV, ROLE)]
    for (w, part, dst, c) in ((B[8], B[16], U0, CS2_CU), (B[9], B[17], U1, CS2_CU), (B[14], B[10], V0, CS2_CV), (B[15], B[11], V1, CS2_CV)):
        L += seq_norm(w)                                                         # digits into [0, 2^28): 3 w + 2 p stays far inside 32 bits
        for j in range(14):
            L += ["v_mul_lo_u32 %s, %s, %s" % (part(j), part(j), c), "v_lshl_add_u32 %s, %s, 1, %s" % (dst(j), w(j), part(j)),
                  "v_add_u32_e64 %s, %s, %s" % (dst(j), dst(j), w(j))]
        L += seq_reduce(dst)
    L.append("s_not_b64 %s, %s" % (ROLE, ROLE))
    return L


def pstart2_body():
    """u, v <- the lane's half of the compressed base Y: even lanes (role A) z2, z3, odd lanes (role B) z4, z5; ROLE <- the odd lanes"""
    L = ["v_mbcnt_lo_u32_b32 %s, -1, 0" % CS2_LANE, "v_mbcnt_hi_u32_b32 %s, -1, %s" % (CS2_LANE, CS2_LANE),      # the lane's number: its parity is its first role
         "v_and_b32_e64 %s, 1, %s" % (CS2_LANE, CS2_LANE), "v_cmp_ne_u32_e64 %s, 0, %s" % (ROLE, CS2_LANE)]
    T = VBLK(4)
    for k in range(4):
        dst = VBLK(k)
        L += seq_gload(dst, Y_SLOT + C_IDX[k], aform=False) + seq_gload(T, Y_SLOT + C_IDX[4 + k], aform=False)
        L += ["v_cndmask_b32_e64 %s, %s, %s, %s" % (dst(j), dst(j), T(j), ROLE) for j in range(14)]
        L += seq_reduce(dst)
    return L


def psave2_body():
    """the lane's half of the compressed state -> the record the run-time offset selects: role A lanes write z2, z3 (slots 0..3 of the
    record), role B lanes z4, z5 (slots 4..7): the lane offset of the stores carries the difference. The state stays."""
    T = VBLK(4)
    L = ["v_mov_b32_e32 %s, %s" % (CS2_LANE, GSTRIDE), "v_mul_lo_u32 %s, %s, 48" % (CS2_LANE, CS2_LANE),          # 4 slots x 12 words
         "v_cndmask_b32_e64 %s, 0, %s, %s" % (CS2_LANE, CS2_LANE, ROLE), "v_add_u32_e64 %s, %s, %s" % (CS2_LANE, CS2_LANE, LADDR)]
    for k in range(4):
        src = VBLK(k)
        L += ["v_mov_b64_e64 v[%d:%d], v[%d:%d]" % (vb(4) + j, vb(4) + j + 1, vb(k) + j, vb(k) + j + 1) for j in range(0, 14, 2)]
        L += seq_pack_pass(T) + seq_to32(T) + seq_gstore(T, K_SLOT + k, koff=True, lane=CS2_LANE)
    return L


def final_exp_d_routine(two_lane=False):
    """In:  workspace slots 13..24 = f (2^384 domain, canonical); v252 LDS byte address of the lane's column (11 spill slots);
         s[68:69] workspace base adjusted so that v252 is the lane offset, s70 bytes between consecutive words of a value.
    Out: f^(3 (p^12 - 1) / r) in v108..v251 (twelve groups of 12 words, tower order, canonical, 2^384 domain).
    Workspace slots 0..24, 31..42 and 49..108 are overwritten.
    two_lane: the variant for lane PAIRS (see above): lanes 2 i and 2 i + 1 come with the same v252 and the same workspace item."""
    bodies, stats = {}, {}
    for name in FEXP_BODIES:
        bodies[name], stats[name] = build_fexp(name, pair_mode=two_lane)
    if two_lane:
        bodies.update(pstart=pstart2_body(), csqr=csqr2_body(), psave=psave2_body())
        for name in ("pstart", "csqr", "psave"):
            stats[name] = dict(lines=len(bodies[name]))
    X = lambda name: expand_calls_d(bodies[name])
    POWER, CSQR = 60, 61
    next_rec = ["s_add_u32 %s, %s, s72" % (GKOFF, GKOFF)]
    pro = ["s_mov_b64 s[80:81], s[30:31]", "s_waitcnt vmcnt(0)"] + shell_constants() + ["s_mul_i32 s72, %s, %d" % (GSTRIDE, 12 * K_REC)]
    if two_lane:
        pro += pair_prologue()                      # the bodies outside the squaring chains have their products in pairs
    main = X("easy") + ["s_mov_b32 s79, 0", "5:"] + call_sub(POWER)
    main += ["s_cmp_lt_u32 s79, 2", "s_cbranch_scc1 13f"] + far_fwd(11) + ["13:"] + X("step_conj") + far_fwd(20)
    main += ["11:", "s_cmp_eq_u32 s79, 2", "s_cbranch_scc1 13f"] + far_fwd(12) + ["13:"] + X("step_frob") + far_fwd(20)
    main += ["12:", "s_cmp_eq_u32 s79, 3", "s_cbranch_scc1 13f"] + far_fwd(14) + ["13:"] + X("step_base")
    main += ["20:", "s_add_u32 s79, s79, 1"] + far_back(5)
    main += ["14:"] + X("tail")
    epi = f_out_epilogue("s[80:81]")
    # state <- Y^|x| (Y in its workspace home): compressed chain with six saves, joint decompression, five products
    power = ["%d:" % POWER] + X("pstart") + ["s_mov_b32 %s, 0" % GKOFF]
    for ph, n in enumerate(POW_RUNS):
        power += ["s_mov_b32 s39, %d" % n, "s_getpc_b64 s[96:97]", "7:", "s_add_u32 s96, s96, 8f-7b", "s_addc_u32 s97, s97, 0"] + far_fwd(CSQR) + ["8:"]
        power += X("psave") + (next_rec if ph < 5 else [])
    power += ["s_waitcnt vmcnt(0)"] + X("pinv") + ["s_mov_b32 %s, 0" % GKOFF] + X("pfirst") + ["s_mov_b32 s39, 5", "1:"] + next_rec + X("pmul")
    power += ["s_sub_u32 s39, s39, 1", "s_cmp_lg_u32 s39, 0", "s_cbranch_scc0 2f"] + far_back(1) + ["2:", "s_setpc_b64 s[98:99]"]
    csqr = ["%d:" % CSQR, ".p2align 6", "1:"] + X("csqr") + ["s_sub_u32 s39, s39, 1", "s_cmp_lg_u32 s39, 0", "s_cbranch_scc0 2f"] + far_back(1) + ["2:", "s_setpc_b64 s[96:97]"]
    pieces = dict(bodies, pro=pro, epi=epi)
    return pro + main + expand_calls_d(epi) + ["s_setpc_b64 s[30:31]"] + power + csqr, pieces, stats


# ---------------------------------------------------------------------------------------------- the sum of an item's public keys
# AggregatePublicKey::aggregate (reference src/aggregates.rs:29-39) as ONE routine per key format: a loop over the lane's keys, each
# iteration = fetch (one key ahead) + decode + Jacobian mixed addition (formulas and case handling of g1_madd_inl in mbls_curve.h),
# the running sum in AGPR blocks 0..2. The Fp products come in independent pairs (mbls_fp_mulpair_d_asm_fn).
#   raw:     96-byte uncompressed keys (big-endian x || y, flag bits in byte 0) at a per-lane address: byte order, flag and range
#            checks in the shell, conversion to the Montgomery domain and the on-curve check y^2 = x^3 + 4 in the body (16 products)
#   indexed: 32-bit indices at a per-lane address into a resident table of 128-byte records (x, y in the 2^384 Montgomery domain,
#            a flag word; see MBLS_KEYREC_DWORDS in mbls_lanes.h): 11 products
# Shell registers (blocks 15..17 are withheld from the allocator): v224..v247 the 24 words of the key in flight, v[248:249] running
# address, v250 the lane's key count, v251 status out (bit 0: a key was infinity, bit 1: a key was undecodable, bit 2: the sum is
# infinity), v210..v216 temporaries. s39 counts keys; exec = lanes that still have a key; s[82:83] the full exec mask.
G1_FREE_V = list(range(8, 14))             # block 14: v196..v207 hold the limbs of p (range check of the raw coordinates)
PLIMB = lambda j: "v%d" % (196 + j)
KW = lambda j: "v%d" % (224 + j)
M_INF2, M_CURVE, M_H0, M_R0, M_INF1, M_INF2F, M_BAD = "s[48:49]", "s[50:51]", "s[52:53]", "s[54:55]", "s[84:85]", "s[86:87]", "s[88:89]"
EXEC_ALL, EXEC_ACT = "s[82:83]", "s[90:91]"
R392SQ = R392 * R392 % P             # x (plain) times this, Montgomery-multiplied, is x in the 2^392 domain
FOUR_D = 4 * R392 % P
G1_RAW_IN = Bound.normalised(0, P - 1)


def prog_g1_step(mode):
    """acc <- acc + key. Live in: acc = (X, Y, Z) in AGPR blocks 0..2; the key's coordinates in VGPR blocks 8, 9 -- plain integers
    below p (raw) or Montgomery words cut as 2^392-domain digits (indexed); mask M_INF2 = the key counts as infinity. Out: the new sum
    in AGPR blocks 0..2, the old one in 5..7 (for the doubling case), masks M_H0, M_R0, M_INF1, M_INF2F (raw: includes off-curve)."""
    p = Prog()
    X, Y, Z = [p.live_in(("a", i)) for i in range(3)]
    kx, ky = p.live_in(("v", 8)), p.live_in(("v", 9))
    if mode == "raw":
        r2 = p.const(R392SQ)
        x2, y2 = p.mulpair(kx, r2, ky, r2)
        xx, yy = p.mulpair(x2, x2, y2, y2)
        x3, Z1Z1 = p.mulpair(xx, x2, Z, Z)
        p.iszero(p.sub(p.sub(yy, x3), p.const(FOUR_D)), M_CURVE)           # y^2 = x^3 + 4
        p.mask_orn2(M_INF2F, M_INF2, M_CURVE)
        T, U2 = p.mulpair(y2, Z, x2, Z1Z1)
        H = p.sub(U2, X)
        S2, ZH = p.mulpair(T, Z1Z1, Z, H)
    else:
        x2, y2 = kx, ky
        Z1Z1, T = p.mulpair(Z, Z, y2, Z)
        U2, S2 = p.mulpair(x2, Z1Z1, T, Z1Z1)
        H = p.sub(U2, X)
        ZH = None
    RR = p.scale(p.sub(S2, Y), 2)
    p.iszero(H, M_H0); p.iszero(RR, M_R0); p.iszero(Z, M_INF1)
    HH, RR2 = p.mulpair(H, H, RR, RR)
    if ZH is None:
        ZH = p.mul1(Z, H)
    I4 = p.scale(HH, 4)
    J, V = p.mulpair(H, I4, X, I4)
    X3 = p.sub(p.sub(RR2, J), p.scale(V, 2))
    M0, M1 = p.mulpair(RR, p.sub(V, X3), Y, J)
    Y3 = p.sub(M0, p.scale(M1, 2))
    Z3 = p.scale(ZH, 2)                                                    # (Z + H)^2 - Z^2 - H^2
    inf2 = M_INF2F if mode == "raw" else M_INF2
    out = [p.sel(M_INF1, X3, x2), p.sel(M_INF1, Y3, y2), p.sel(M_INF1, Z3, p.const(ONE_D))]
    out = [p.sel(inf2, out[0], X), p.sel(inf2, out[1], Y), p.sel(inf2, out[2], Z)]
    for i, v in enumerate((X, Y, Z)):
        p.store(v, ("a", 5 + i))
    for i, v in enumerate(out):
        p.store(prog_reduce(p, v), ("a", i))
    return p


def prog_g1_dbl(src=5):
    """the sum as it was before this key (AGPR blocks 5..7), doubled, into blocks 0..2: the key equals the running sum (g1_dbl's formulas);
    src = 0: the running point doubled in place"""
    p = Prog()
    X, Y, Z = [p.live_in(("a", src + i)) for i in range(3)]
    A, B = p.mulpair(X, X, Y, Y)
    XB = p.add(X, B)
    C, S = p.mulpair(B, B, XB, XB)
    D = p.scale(p.sub(p.sub(S, A), C), 2)
    E = p.scale(A, 3)
    F, YZ = p.mulpair(E, E, Y, Z)
    X3 = p.sub(F, p.scale(D, 2))
    Y3 = p.sub(p.mul1(E, p.sub(D, X3)), p.scale(C, 8))
    for i, v in enumerate((X3, Y3, p.scale(YZ, 2))):
        p.store(prog_reduce(p, v), ("a", i))
    return p


def build_g1(which):
    p = prog_g1_dbl() if which == "dbl" else prog_g1_step(which)
    key_in = G1_RAW_IN if which == "raw" else G_IN
    inb = {v: (STATE_IN if l[0] == "a" else key_in) for v, l in p.init_loc.items()}
    al = AllocD(p, inb, n_lds=0, a_pool=list(range(8, NA)), free_v=G1_FREE_V)
    body = al.run()
    for dst, B in al.stored.items():
        assert B.vlo >= STATE_IN.vlo and B.vhi <= STATE_IN.vhi and B.dlo >= 0 and B.dhi <= M28, (dst, B)
    return body, al.stats


P_LIMBS = [(P >> (32 * j)) & 0xFFFFFFFF for j in range(12)]


def g1_decode_raw():
    """the key in flight (KW: 24 memory-order words of the 96 big-endian bytes) -> plain digits of x, y in blocks 8, 9; M_INF2 = it
    counts as infinity, M_BAD = it is undecodable (g1_decode_uncompressed_w: the compression flag, a non-canonical infinity, the sign
    flag, a coordinate >= p; the on-curve test follows in the body). s77 holds the byte-swap selector."""
    T0, T1 = "v210", "v211"
    L = ["v_and_b32_e32 %s, 0xff, %s" % (T0, KW(0)), "v_lshrrev_b32_e64 %s, 8, %s" % (T1, KW(0))]
    for j in range(1, 23, 2):
        L.append("v_or3_b32 %s, %s, %s, %s" % (T1, T1, KW(j), KW(j + 1)))
    L += ["v_or_b32_e64 %s, %s, %s" % (T1, T1, KW(23)),                                         # everything after byte 0
          "v_and_b32_e64 v212, 0x3f, %s" % T0, "v_or_b32_e64 %s, %s, v212" % (T1, T1),              # ... and the low six bits of byte 0
          "v_cmp_ne_u32_e64 s[92:93], 0, %s" % T1,                                              # an infinity flag with anything else set
          "v_and_b32_e64 v212, 0x40, %s" % T0, "v_cmp_ne_u32_e64 %s, 0, v212" % M_INF2,
          "s_and_b64 %s, %s, s[92:93]" % (M_BAD, M_INF2),
          "v_and_b32_e32 v212, 0x80, %s" % T0, "v_cmp_ne_u32_e64 s[92:93], 0, v212", "s_or_b64 %s, %s, s[92:93]" % (M_BAD, M_BAD),
          "v_and_b32_e64 v212, 0x20, %s" % T0, "v_cmp_ne_u32_e64 s[92:93], 0, v212"]                # sign flag: only an error when finite
    for j in range(24):
        L.append("v_perm_b32 %s, %s, %s, s77" % (KW(j), KW(j), KW(j)))
    for base in (0, 12):                                                                          # borrow of (coordinate - p): set iff it is < p
        for j in range(12):
            w = KW(base + 11 - j)
            L.append(("v_sub_co_u32_e32 %s, vcc, %s, %s" if j == 0 else "v_subb_co_u32_e32 %s, vcc, %s, %s, vcc") % (T1, w, PLIMB(j)))
        L += ["s_orn2_b64 s[92:93], s[92:93], vcc"]
    L += ["s_andn2_b64 s[92:93], s[92:93], %s" % M_INF2, "s_or_b64 %s, %s, s[92:93]" % (M_BAD, M_BAD), "s_and_b64 %s, %s, exec" % (M_BAD, M_BAD),
          "s_or_b64 %s, %s, %s" % (M_INF2, M_INF2, M_BAD)]
    L += seq_conv(lambda j: "v%d" % (vb(8) + j), [KW(11 - j) for j in range(12)], False)
    L += seq_conv(lambda j: "v%d" % (vb(9) + j), [KW(23 - j) for j in range(12)], False)
    return L


def g1_status(inf_mask, bad_mask):
    return ["v_cndmask_b32_e64 v211, 0, 1, %s" % inf_mask, "v_cndmask_b32_e64 v212, 0, 2, %s" % bad_mask, "v_or3_b32 v251, v251, v211, v212"]


def g1_aggregate_d_routine(mode):
    """In:  v[248:249] address of the lane's first key (raw: 96-byte records) or first index (indexed: uint32), v250 its key count;
         indexed: s[94:95] the table's records, s96 its size; v252, s[68:69], s70 the workspace addressing of the other routines.
    Out: the sum in workspace slots 0..2 (Jacobian X, Y, Z; canonical, 2^384 domain); v251 status bits."""
    step, st_step = build_g1(mode)
    dbl, st_dbl = build_g1("dbl")
    pro = ["s_mov_b64 s[80:81], s[30:31]", "s_waitcnt vmcnt(0)"] + shell_constants()
    pro += ["s_mov_b64 %s, exec" % EXEC_ALL, "s_mov_b32 s77, 0x00010203", "v_mov_b32_e32 v251, 0", "s_mov_b32 s39, 0", "v_mov_b32_e32 v254, 0"]
    for i in range(3):                                  # the sum starts at infinity: (0, 1, 0)
        for j, dgt in enumerate(digits_of(ONE_D if i == 1 else 0)):
            pro += (["v_mov_b32_e32 v255, 0x%08x" % dgt, "v_accvgpr_write_b32 a%d, v255" % (vb(i) + j)] if dgt else ["v_accvgpr_write_b32 a%d, v254" % (vb(i) + j)])

    def lanes_with_key(offset):
        """exec <- lanes whose key count exceeds s39 + offset"""
        src = "s39" if offset == 0 else "s76"
        pre = [] if offset == 0 else ["s_add_u32 s76, s39, %d" % offset]
        # the comparison itself must see every lane: a v_cmp writes zeros for lanes that are switched off
        return pre + ["s_mov_b64 exec, %s" % EXEC_ALL, "v_cmp_gt_u32_e64 vcc, v250, %s" % src, "s_mov_b64 exec, vcc"]

    if mode == "raw":
        fetch = ["global_load_dwordx4 v[%d:%d], v[248:249], off offset:%d" % (224 + 4 * q, 227 + 4 * q, 16 * q) for q in range(6)]
        advance = ["v_add_co_u32_e32 v248, vcc, 0x60, v248", "v_addc_co_u32_e32 v249, vcc, 0, v249, vcc"]
        pro += ["v_mov_b32_e32 %s, 0x%08x" % (PLIMB(j), P_LIMBS[j]) for j in range(12)]
        pro += lanes_with_key(0) + fetch + advance
        decode = g1_decode_raw()
        nxt = lanes_with_key(1) + fetch + advance
    else:
        ID, IDHI, RA, FL, OOB = "v214", "v215", "v[212:213]", "v210", "v216"
        addr = ["v_cmp_le_u32_e64 vcc, s96, %s" % ID, "v_cndmask_b32_e64 %s, 0, 3, vcc" % OOB, "v_cndmask_b32_e64 %s, %s, 0, vcc" % (ID, ID),
                "v_mov_b32_e32 %s, 0" % IDHI, "v_lshlrev_b64 %s, 7, v[214:215]" % RA,
                "v_add_co_u32_e64 v212, vcc, s94, v212", "v_mov_b32_e32 v211, s95", "v_addc_co_u32_e64 v213, vcc, v211, v213, vcc"]
        fetch = ["global_load_dwordx4 v[%d:%d], %s, off offset:%d" % (224 + 4 * q, 227 + 4 * q, RA, 16 * q) for q in range(6)]
        fetch += ["global_load_dword v217, %s, off offset:96" % RA]
        load_id = ["global_load_dword %s, v[248:249], off" % ID, "v_add_co_u32_e64 v248, vcc, 4, v248", "v_addc_co_u32_e64 v249, vcc, 0, v249, vcc"]
        # prologue: index 0 -> record 0 in flight, index 1 in flight
        pro += lanes_with_key(0) + load_id + ["s_waitcnt vmcnt(0)"] + addr + fetch + ["v_mov_b32_e32 v218, %s" % OOB] + lanes_with_key(1) + load_id
        # top of an iteration (everything arrived): flags of this key (an index outside the table counts as undecodable)
        decode = ["v_or_b32_e64 %s, v217, v218" % FL,
                  "v_and_b32_e64 v211, 1, %s" % FL, "v_cmp_ne_u32_e64 %s, 0, v211" % M_INF2,
                  "v_and_b32_e64 v211, 2, %s" % FL, "v_cmp_ne_u32_e64 %s, 0, v211" % M_BAD]
        decode += seq_conv(lambda j: "v%d" % (vb(8) + j), [KW(j) for j in range(12)], True)
        decode += seq_conv(lambda j: "v%d" % (vb(9) + j), [KW(12 + j) for j in range(12)], True)
        nxt = lanes_with_key(1) + addr + fetch + ["v_mov_b32_e32 v218, %s" % OOB] + lanes_with_key(2) + load_id
    inf_final = M_INF2F if mode == "raw" else M_INF2
    loop = ["1:"] + lanes_with_key(0) + ["s_mov_b64 %s, exec" % EXEC_ACT, "s_cbranch_execnz 3f"] + far_fwd(9) + ["3:", "s_waitcnt vmcnt(0)"]
    loop += decode + nxt + ["s_mov_b64 exec, %s" % EXEC_ACT]
    loop += expand_calls_d(step)
    post = []
    if mode == "raw":                                   # an off-curve key is undecodable too
        post += ["s_andn2_b64 s[92:93], %s, %s" % (M_INF2F, M_INF2), "s_or_b64 %s, %s, s[92:93]" % (M_BAD, M_BAD)]
    post += g1_status(inf_final, M_BAD)
    loop += post
    # same x and same y as the running sum, neither infinite: the doubling, on those lanes only
    loop += ["s_and_b64 s[92:93], %s, %s" % (M_H0, M_R0), "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF1, "s_andn2_b64 s[92:93], s[92:93], %s" % inf_final,
             "s_and_b64 s[92:93], s[92:93], exec", "s_cbranch_scc1 3f"] + far_fwd(4) + ["3:", "s_mov_b64 exec, s[92:93]"]
    loop += expand_calls_d(dbl) + ["4:", "s_add_u32 s39, s39, 1"] + far_back(1)
    # epilogue: canonical 2^384-domain words of X, Y, Z to the workspace; is the sum infinity?
    epi = ["9:", "s_mov_b64 exec, %s" % EXEC_ALL, "s_waitcnt vmcnt(0)"]
    B0, B1, B2, B5, B6 = (lambda j: "v%d" % j), (lambda j: "v%d" % (14 + j)), (lambda j: "v%d" % (28 + j)), (lambda j: "v%d" % (70 + j)), (lambda j: "v%d" % (84 + j))
    epi += ["v_mov_b32_e32 %s, 0x%08x" % (B2(j), dgt) for j, dgt in enumerate(digits_of(K384))]
    for half in range(2):
        srcs = (0, 1) if half == 0 else (2, 2)
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B0(j), vb(srcs[0]) + j) for j in range(14)]
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B1(j), vb(srcs[1]) + j) for j in range(14)]
        epi += ["CALL mbls_fp2_mulfp_d_asm_fn"]
        for h, B in ((0, B5), (1, B6))[:2 if half == 0 else 1]:
            epi += seq_reduce(B) + seq_canonical(B) + seq_to32(B)
            if half == 1:
                epi += ["v_or_b32_e64 v211, %s, %s" % (B(0), B(1))] + ["v_or3_b32 v211, v211, %s, %s" % (B(j), B(j + 1)) for j in range(2, 12, 2)]
                epi += ["v_cmp_eq_u32_e64 vcc, 0, v211", "v_cndmask_b32_e64 v211, 0, 4, vcc", "v_or_b32_e64 v251, v251, v211"]
            epi += seq_gstore(B, 2 * half + h)
    epi += ["s_waitcnt vmcnt(0)", "s_mov_b64 s[30:31], s[80:81]"]
    pieces = dict(pro=pro, decode=decode, nxt=nxt, step=step, post=post, dbl=dbl, epi=epi)
    return pro + loop + expand_calls_d(epi), pieces, dict(step=st_step, dbl=st_dbl)


# ---------------------------------------------------------------------------------------------- G2 doubling (subgroup check, cofactor clearing)
G2D_ARG = [108 + 12 * i for i in range(6)]         # X.c0, X.c1, Y.c0, Y.c1, Z.c0, Z.c1 as six groups of 12 words, in and out
G2_IN = Bound.normalised(-16 * P, 16 * P)          # every round, the first included: the shell reduces the converted inputs


def prog_g2_dbl_d(src=0, dst=0):
    """Jacobian doubling in E'(Fp2) (formulas of g2_dbl in mbls_curve.h; valid for every curve point including infinity). X, Y, Z in
    AGPR blocks src..src+5, result in dst..dst+5. Every output is a combination of products only, so a carry pass at the store keeps
    the state bounded."""
    p = Prog()
    l = [p.live_in(("a", src + i)) for i in range(6)]
    X, Y, Z = (l[0], l[1]), (l[2], l[3]), (l[4], l[5])
    A = p.sqr2(X); B = p.sqr2(Y); C = p.sqr2(B)
    D = p.dbl2(p.sub2(p.sub2(p.sqr2(p.add2(X, B)), A), C))
    E = p.mul3_2(A); F = p.sqr2(E)
    Z3 = p.dbl2(p.mul2(Y, Z))
    X3 = p.sub2(F, p.dbl2(D))
    Y3 = p.sub2(p.mul2(E, p.sub2(D, X3)), p.mul8_2(C))
    for v, i in zip((X3[0], X3[1], Y3[0], Y3[1], Z3[0], Z3[1]), range(6)):
        p.store(prog_norm(p, v), ("a", dst + i))
    return p


def g2_dbl_d_routine():
    """s38 doublings of the point given in six groups of 12 words (v108..v179, Montgomery-2^384 form, canonical), in place."""
    p = prog_g2_dbl_d()
    al = AllocD(p, {v: G2_IN for v in p.init_loc}, n_lds=0, lds_base=0, a_pool=list(range(NA)))
    body = al.run()
    for dst, B in al.stored.items():
        assert B.vlo >= G2_IN.vlo and B.vhi <= G2_IN.vhi and B.dlo >= 0 and B.dhi <= M28, (dst, B)
    W = lambda j: "v%d" % j
    pro = ["s_mov_b32 s39, s38"]
    for i in range(6):                              # words * 2^8 can be 256 p: reduce, so that the bounds the body was generated under hold
        pro += seq_conv(W, ["v%d" % (G2D_ARG[i] + q) for q in range(12)], True) + seq_reduce(W)        # from the first round on
        pro += ["v_accvgpr_write_b32 a%d, v%d" % (vb(i) + j, j) for j in range(14)]
    B0, B1, B2, B5, B6 = (lambda j: "v%d" % j), (lambda j: "v%d" % (14 + j)), (lambda j: "v%d" % (28 + j)), (lambda j: "v%d" % (70 + j)), (lambda j: "v%d" % (84 + j))
    epi = ["v_mov_b32_e32 %s, 0x%08x" % (B2(j), dgt) for j, dgt in enumerate(digits_of(K384))]
    for i in range(3):
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B0(j), vb(2 * i) + j) for j in range(14)]
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B1(j), vb(2 * i + 1) + j) for j in range(14)]
        epi += ["CALL mbls_fp2_mulfp_d_asm_fn"]
        for h, B in ((0, B5), (1, B6)):
            epi += seq_reduce(B) + seq_canonical(B) + seq_to32(B)
            epi += ["v_mov_b32_e64 v%d, %s" % (G2D_ARG[2 * i + h] + j, B(j)) for j in range(12)]
    full = wrap_loop_d(expand_calls_d(body), "s39", pro, expand_calls_d(epi))
    return full, dict(pro=pro, body=body, epi=epi), al.stats


# ---------------------------------------------------------------------------------------------- G2 group routines
# The group arithmetic of the signature and message phases as generated routines: the subgroup test psi(P) = [x]P of a decoded signature
# (g2_in_subgroup in mbls_curve.h; reference src/signature.rs:29, src/aggregates.rs:184) and everything after the two map_to_curve
# evaluations of hash_to_curve_g2 (q0 + q1, Budroni-Pintore cofactor clearing: g2_clear_cofactor; reference src/amcl_utils.rs:33-35).
# The running point lives in AGPR blocks 0..5 (Jacobian X, Y, Z), every other point in workspace slots (packed, 2^392 domain); an
# addition takes its second operand from the staging slots AD. Additions follow g2_add's case handling: operands at infinity by
# selection, equal operands by a doubling run on those lanes only (exec-masked), opposite operands give Z = 0 by the formulas.
G2M_TMP0, G2M_TMP1 = "s[86:87]", "s[88:89]"
PSI_CX = None


def f2inv_py(a):
    n = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
    return (a[0] * n % P, -a[1] * n % P)


PSI_CX = f2inv_py(f2pow_py((1, 1), (P - 1) // 3))
PSI_CY = f2inv_py(f2pow_py((1, 1), (P - 1) // 2))
PSI2_CX = (PSI_CX[0] * PSI_CX[0] + PSI_CX[1] * PSI_CX[1]) % P


def c2(p, c):
    return (p.const(D392(c[0])), p.const(D392(c[1])))


def pt_live_in(p, kind, base):
    l = [p.live_in((kind, base + i)) for i in range(6)]
    return [(l[0], l[1]), (l[2], l[3]), (l[4], l[5])]


def pt_store_acc(p, pt, base=0):
    for i, v in enumerate([x for c in pt for x in c]):
        p.store(prog_norm(p, v), ("a", base + i))


def pt_park(p, pt, slot):
    for i, v in enumerate([x for c in pt for x in c]):
        p.ops.append(("storep", [], [v], slot + i))


def pt_psi(p, pt):
    return [p.mul2(p.conj2(pt[0]), c2(p, PSI_CX)), p.mul2(p.conj2(pt[1]), c2(p, PSI_CY)), p.conj2(pt[2])]


def pt_neg(p, pt):
    return [pt[0], p.neg2(pt[1]), pt[2]]


# ---- map_to_curve (simplified SWU + 3-isogeny), formulas and case handling of map_to_curve_g2 in mbls_hash.h
def read_product_constants(names):
    """constants of milagro_bls_amd/csrc/mbls_constants.inc (32-bit limbs, 2^384 Montgomery form) as integers"""
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "milagro_bls_amd", "csrc", "mbls_constants.inc")
    txt = open(path).read()
    ri = pow(1 << 384, -1, P)
    out = {}
    for nm in names:
        m = re.search(r"MBLS_CONST uint32_t %s((?:\[\d+\])+) = (\{.*?\});" % nm, txt, re.S)
        words = [int(w, 16) for w in re.findall(r"0x([0-9a-fA-F]{8})u", m.group(2))]
        vals = [sum(words[12 * i + j] << (32 * j) for j in range(12)) * ri % P for i in range(len(words) // 12)]
        out[nm] = vals
    return out


SSWU_A, SSWU_B, SSWU_Z = (0, 240), (1012, 1012), (P - 2, P - 1)
SQRT_C5 = pow(5, (P + 1) // 4, P)                   # C5^2 = -5 = -norm(Z)
INV2 = pow(2, -1, P)
_ISO = read_product_constants(["MBLS_ISO3_XNUM", "MBLS_ISO3_YNUM", "MBLS_ISO3_K"])
ISO3_XNUM = [(_ISO["MBLS_ISO3_XNUM"][2 * i], _ISO["MBLS_ISO3_XNUM"][2 * i + 1]) for i in range(4)]
ISO3_YNUM = [(_ISO["MBLS_ISO3_YNUM"][2 * i], _ISO["MBLS_ISO3_YNUM"][2 * i + 1]) for i in range(4)]
ISO3_K = tuple(_ISO["MBLS_ISO3_K"])
M_SQ1, M_CHI, M_T0, M_S0, M_S1 = "s[48:49]", "s[50:51]", "s[52:53]", "s[54:55]", "s[84:85]"


def prog_sswu():
    """u (slots U, U + 1 of the record selected by the run-time offset; 2^384 domain) -> the point of E'(Fp2) in Jacobian coordinates
    (slots Q0 .. Q0 + 5 of the same record, packed)"""
    p = Prog()
    S = G2_SLOTS
    u = tuple(prog_reduce(p, p.live_in(("gka", S["U"] + i))) for i in range(2))
    out = sswu_formula(p, u)
    for i, v in enumerate([c for pt in out for c in pt]):
        p.ops.append(("storep", [], [v], ("k", S["Q0"] + i)))
    return p


def sswu_formula(p, u, masks=None):
    """simplified SWU + 3-isogeny on the Fp2 value u -> Jacobian point [X, Y, Z] (formulas and case handling of map_to_curve_g2 in
    mbls_hash.h); masks: the names of the six lane masks / flags it uses (M_T0, M_SQ1, M_CHI, M_S0, M_S1, tmp)"""
    M_T0, M_SQ1, M_CHI, M_S0, M_S1, G2M_TMP0 = masks or (globals()["M_T0"], globals()["M_SQ1"], globals()["M_CHI"], globals()["M_S0"], globals()["M_S1"], globals()["G2M_TMP0"])
    A, B, Z = c2(p, SSWU_A), c2(p, SSWU_B), c2(p, SSWU_Z)
    fpc = lambda x: p.const(D392(x))
    norm = lambda a: (lambda sq: p.add(sq[0], sq[1]))(p.mulpair(a[0], a[0], a[1], a[1]))
    tv1 = p.mul2(Z, p.sqr2(u))
    tv2 = p.add2(p.sqr2(tv1), tv1)
    xn = p.mul2(B, (p.add(tv2[0], fpc(1)), tv2[1]))
    p.iszero2(tv2, M_T0, G2M_TMP0)
    xd = p.mul2(A, p.sel2(M_T0, p.neg2(tv2), Z))
    xd2 = p.sqr2(xd); D = p.mul2(xd2, xd)
    N = p.add2(p.mul2(p.add2(p.sqr2(xn), p.mul2(A, xd2)), xn), p.mul2(B, D))
    g = p.mul2(N, D)
    nd, nN = norm(xd), norm(N)
    nd2 = p.mul1(nd, nd)
    t_a, t_b = p.mulpair(nN, nd2, nd2, nd)                       # nN nd^2, nd^3
    ng = p.mul1(nN, t_b)
    w1 = p.pow34(ng)
    s_, w1sq = p.mulpair(w1, ng, w1, w1)
    p.iszero(p.sub(p.mul1(s_, s_), ng), M_SQ1)                   # gx1 is a square in Fp2
    inv_ng = p.sel(M_SQ1, p.neg(w1sq), w1sq)                     # 1 / norm(g) = chi w1^2
    inv_nd = p.mul1(t_a, inv_ng)
    inv_xd = p.mulfp2(p.conj2(xd), inv_nd)
    G = p.sel2(M_SQ1, p.mul2(tv1, g), g)
    s_alt = p.mul1(p.mul1(norm(u), fpc(SQRT_C5)), s_)
    Sr = p.sel(M_SQ1, s_alt, s_)
    half = fpc(INV2)
    t, t_alt = p.mulpair(p.add(G[0], Sr), half, p.sub(G[0], Sr), half)
    p.iszero(t, M_CHI)
    t = p.sel(M_CHI, t, t_alt)
    w2 = p.pow34(t)
    x0, w2sq = p.mulpair(w2, t, w2, w2)
    p.iszero(p.sub(p.mul1(x0, x0), t), M_CHI)
    g1h, x0w = p.mulpair(G[1], half, x0, w2sq)
    other = p.mul1(g1h, x0w)
    r = (p.sel(M_CHI, other, x0), p.sel(M_CHI, x0, other))
    inv_D = p.mul2(p.sqr2(inv_xd), inv_xd)
    x = p.mul2(xn, inv_xd); y = p.mul2(r, inv_D)
    x = p.sel2(M_SQ1, p.mul2(tv1, x), x)
    y = p.sel2(M_SQ1, p.mul2(tv1, y), y)
    p.sgn0_2(u, M_S0, G2M_TMP0); p.sgn0_2(y, M_S1, G2M_TMP0); p.mask_xor(M_S0, M_S0, M_S1)
    y = p.sel2(M_S0, y, p.neg2(y))
    xnum, ynum = c2(p, ISO3_XNUM[3]), c2(p, ISO3_YNUM[3])
    for i in (2, 1, 0):
        xnum = p.add2(p.mul2(xnum, x), c2(p, ISO3_XNUM[i]))
        ynum = p.add2(p.mul2(ynum, x), c2(p, ISO3_YNUM[i]))
    return [xnum, p.mul2(y, ynum), p.add2(x, c2(p, ISO3_K))]


def prog_g2_add(ad_slot, negate, table=False, table_kind="gv"):
    """acc <- acc +- (the point in slots ad_slot..ad_slot+5); the old acc goes to AGPR blocks 6..11 for the doubling case; masks M_H0,
    M_R0 (same x / same y), M_INF1, M_INF2 (an operand at infinity). table: the second operand is the record each LANE selects (kind
    'gv', register VOFF) of a table that starts at ad_slot, negated on the lanes of M_NEGQ and ignored (as if infinite) on those of M_ZEROQ."""
    p = Prog()
    A = pt_live_in(p, "a", 0)
    Q = pt_live_in(p, table_kind if table else "gd", ad_slot)
    if table:
        Q = [Q[0], p.sel2(M_NEGQ, Q[1], p.neg2(Q[1])), Q[2]]
    if negate:
        Q = pt_neg(p, Q)
    z1z1, z2z2 = p.sqr2(A[2]), p.sqr2(Q[2])
    u1, u2 = p.mul2(A[0], z2z2), p.mul2(Q[0], z1z1)
    s1 = p.mul2(p.mul2(A[1], Q[2]), z2z2)
    s2 = p.mul2(p.mul2(Q[1], A[2]), z1z1)
    h = p.sub2(u2, u1)
    rr = p.dbl2(p.sub2(s2, s1))
    p.iszero2(h, M_H0, G2M_TMP0); p.iszero2(rr, M_R0, G2M_TMP0)
    p.iszero2(A[2], M_INF1, G2M_TMP0); p.iszero2(Q[2], M_INF2, G2M_TMP0)
    if table:
        p.mask_or(M_INF2, M_INF2, M_ZEROQ)
    i4 = p.sqr2(p.dbl2(h))
    j, v = p.mul2(h, i4), p.mul2(u1, i4)
    X3 = p.sub2(p.sub2(p.sqr2(rr), j), p.dbl2(v))
    Y3 = p.sub2(p.mul2(rr, p.sub2(v, X3)), p.dbl2(p.mul2(s1, j)))
    Z3 = p.mul2(p.sub2(p.sub2(p.sqr2(p.add2(A[2], Q[2])), z1z1), z2z2), h)
    out = [p.sel2(M_INF1, o, q) for o, q in zip((X3, Y3, Z3), Q)]
    out = [p.sel2(M_INF2, o, a) for o, a in zip(out, A)]
    for i, v_ in enumerate([x for c in A for x in c]):
        p.store(v_, ("a", 6 + i))
    pt_store_acc(p, out)
    return p


# slots 0..12 hold the phases' results (sum of keys, signature, H) and 25..30 the n-pairing paths' G2 accumulator, which other kernels may be
# writing or keeping meanwhile: the hash routine's scratch is 13..24 and 31..42, the signature routine's 43..48
G2_SLOTS = dict(AD=19, P=31, T1=37, T2=13, T3=7, Q0=7, Q1=13, SIGAD=43, SIG=3, H=7, U=31)      # U: u0 in 31, 32; u1 in 37, 38
# [r] sig of verify_multiple (g2_blind_routine): the table of 1..8 times the signature in slots 49..96 (the final exponentiation's records: not in
# use while the signature phase runs), the result in the n-pairing paths' accumulator slots 25..30
BL_TAB, BL_OUT = 49, 25
BL_SEL = 97                          # constant-time form (signing): the record the window selected, copied here by a scan over all eight (slots 97..102)
BL_FREE_V = list(range(9, 17))       # block 17 (v238..v251) belongs to the shell: v[248:249] the scalar, v247 its top digit, v250 = VOFF, v251 status
M_NEGQ, M_ZEROQ = "s[90:91]", "s[36:37]"              # (not s[94:95]: that pair is PAIR_ROLE in the two-lane routines)


def prog_g2_madd(ad_slot):
    """acc <- acc + (x, y, 1): the MIXED addition for a base point that is affine -- the signature in the ladder of its subgroup test and in
    the table of its blinding (slots ad_slot..ad_slot+3 hold x, y; the Z = 1 in +4, +5 is not read): 11 products instead of 16
    (madd-2007-bl). Same masks and old-acc copy as prog_g2_add; the affine operand is never infinite."""
    p = Prog()
    A = pt_live_in(p, "a", 0)
    l = [p.live_in(("gd", ad_slot + i)) for i in range(4)]
    Qx, Qy = (l[0], l[1]), (l[2], l[3])
    z1z1 = p.sqr2(A[2])
    u2 = p.mul2(Qx, z1z1)
    s2 = p.mul2(p.mul2(Qy, A[2]), z1z1)
    h = p.sub2(u2, A[0])
    rr = p.dbl2(p.sub2(s2, A[1]))
    p.iszero2(h, M_H0, G2M_TMP0); p.iszero2(rr, M_R0, G2M_TMP0)
    p.iszero2(A[2], M_INF1, G2M_TMP0)
    p.mask_and(M_INF2, M_INF1, M_INF1); p.mask_xor(M_INF2, M_INF2, M_INF1)           # the affine operand is finite: M_INF2 = 0
    i4 = p.sqr2(p.dbl2(h))
    j, v = p.mul2(h, i4), p.mul2(A[0], i4)
    X3 = p.sub2(p.sub2(p.sqr2(rr), j), p.dbl2(v))
    Y3 = p.sub2(p.mul2(rr, p.sub2(v, X3)), p.dbl2(p.mul2(A[1], j)))
    Z3 = p.dbl2(p.mul2(A[2], h))
    one = (p.const(ONE_D), p.const(0))
    out = [p.sel2(M_INF1, o, q) for o, q in zip((X3, Y3, Z3), (Qx, Qy, one))]
    for i, v_ in enumerate([x for c in A for x in c]):
        p.store(v_, ("a", 6 + i))
    pt_store_acc(p, out)
    return p


def prog_g2_glue(which):
    """the point moves between the additions / ladders of the two routines (see g2_hash_tail_d_routine, g2_subgroup_d_routine)"""
    p = Prog()
    S = G2_SLOTS
    acc = lambda: pt_live_in(p, "a", 0)
    gd = lambda name: pt_live_in(p, "gd", S[name])
    one = lambda: (p.const(ONE_D), p.const(0))
    if which == "h_start":                           # acc = q0, AD = q1 (the map_to_curve outputs, packed by the sswu body)
        pt_park(p, gd("Q1"), S["AD"]); pt_store_acc(p, gd("Q0"))
    elif which == "h_q1":                            # two lanes per message: AD = q1 = what the ODD lane's map left in its item's q0 slots, and
        pt_park(p, pt_live_in(p, "gv", S["Q0"]), S["AD"])      # (VOFF = that item's lane offset: the neighbour's for the even lane)
    elif which == "h_q0":                            # acc = q0 = what the EVEN lane's map left in its item (VOFF = that item)
        pt_store_acc(p, pt_live_in(p, "gv", S["Q0"]))
    elif which == "h_base1":                         # p = q0 + q1: remember it, and it is the ladder's base
        a = acc(); pt_park(p, a, S["P"]); pt_park(p, a, S["AD"]); pt_store_acc(p, a)
    elif which == "h_after1":                        # t1 = -[|x|]p; t2 = psi(p); acc = p (to be doubled)
        a = acc(); pt_park(p, pt_neg(p, a), S["T1"])
        pp = gd("P")
        pt_park(p, pt_psi(p, pp), S["T2"]); pt_store_acc(p, pp)
    elif which == "h_psi2":                          # acc = psi^2(2p), AD = t2
        a = acc()
        pt_park(p, gd("T2"), S["AD"])
        pt_store_acc(p, [p.mulfp2(a[0], p.const(D392(PSI2_CX))), p.neg2(a[1]), a[2]])
    elif which == "h_t3":                            # t3 = acc; acc = t1, AD = t2
        pt_park(p, acc(), S["T3"]); pt_park(p, gd("T2"), S["AD"]); pt_store_acc(p, gd("T1"))
    elif which == "h_base2":                         # t1 + t2 is the second ladder's base
        a = acc(); pt_park(p, a, S["AD"]); pt_store_acc(p, a)
    elif which == "h_after2":                        # acc = -[|x|](t1 + t2), AD = t3
        pt_park(p, gd("T3"), S["AD"]); pt_store_acc(p, pt_neg(p, acc()))
    elif which == "h_ad_t1":
        pt_park(p, gd("T1"), S["AD"]); pt_store_acc(p, acc())
    elif which == "h_ad_p":
        pt_park(p, gd("P"), S["AD"]); pt_store_acc(p, acc())
    elif which == "s_start":                         # the decoded signature (affine x, y; 2^384 domain words): acc = AD = (x, y, 1)
        l = [prog_reduce(p, p.live_in(("g", S["SIG"] + i))) for i in range(4)]
        pt = [(l[0], l[1]), (l[2], l[3]), one()]
        pt_park(p, pt, S["SIGAD"]); pt_store_acc(p, pt)
    elif which == "b_tab":                           # table record (the run-time offset selects it) <- acc; acc stays
        a = acc()
        for i, v in enumerate([x for c in a for x in c]):
            p.ops.append(("storep", [], [v], ("k", BL_TAB + i)))
        pt_store_acc(p, a)
    elif which == "b_start":                         # acc = the signature again (x, y, 1)
        pt_store_acc(p, gd("SIGAD"))
    elif which == "b_inf":                           # acc = infinity
        pt_store_acc(p, [(p.const(0), p.const(0)), one(), (p.const(0), p.const(0))])
    elif which == "s_compare":                       # g2_eq(psi(P), -acc) with P = (x, y, 1) -> mask M_H0
        b = pt_neg(p, acc())
        a = pt_psi(p, gd("SIGAD"))
        za2, zb2 = p.sqr2(a[2]), p.sqr2(b[2])
        p.iszero2(p.sub2(p.mul2(a[0], zb2), p.mul2(b[0], za2)), M_H0, G2M_TMP0)
        p.iszero2(p.sub2(p.mul2(p.mul2(a[1], zb2), b[2]), p.mul2(p.mul2(b[1], za2), a[2])), M_R0, G2M_TMP0)
        p.iszero2(a[2], M_INF1, G2M_TMP0); p.iszero2(b[2], M_INF2, G2M_TMP0)
    else:
        raise ValueError(which)
    return p


def build_g2(which, ad_slot=None, free_v=None, pair_mode=False):
    """pair_mode: the body for lane pairs -- independent products of one kind share a call (pair_products)"""
    if which in ("add", "sub"):
        p = prog_g2_add(ad_slot, which == "sub")
    elif which == "addt":
        p = prog_g2_add(BL_TAB, False, table=True)
    elif which == "addt_ct":         # the constant-time form: the window's record was selected into BL_SEL by a scan over ALL records (blind_scan_ct)
        p = prog_g2_add(BL_SEL, False, table=True, table_kind="gd")
    elif which == "madd":
        p = prog_g2_madd(ad_slot)
    elif which == "dbl":
        p = prog_g2_dbl_d()
    elif which == "fix":
        p = prog_g2_dbl_d(6, 0)
    elif which == "sswu":
        p = prog_sswu()
    else:
        p = prog_g2_glue(which)
    if pair_mode:
        pair_products(p)
    inb = {v: (G2_IN if l[0] == "a" else PACKED if l[0] in ("gd", "gk", "gv") else G_IN) for v, l in p.init_loc.items()}      # g, gka: 2^384-domain words
    al = AllocD(p, inb, n_lds=11, lds_base=0, a_pool=list(range(NA)), free_v=free_v)
    body = al.run()
    for dst, B in getattr(al, "stored", {}).items():
        assert B.vlo >= G2_IN.vlo and B.vhi <= G2_IN.vhi and B.dlo >= 0 and B.dhi <= M28, (which, dst, B)
    return body, al.stats


def call_sub(label):
    """internal subroutine call: the return address (the instruction after the jump) in s[98:99]"""
    return ["s_getpc_b64 s[98:99]", "7:", "s_add_u32 s98, s98, 8f-7b", "s_addc_u32 s99, s99, 0"] + far_fwd(label) + ["8:"]


def g2_group_routine(kind, two_lane=False):
    """kind 'hash': in  q0 in workspace slots 7..12, q1 in 19..24 (Jacobian, 2^384 domain, canonical); out: clear_cofactor(q0 + q1) in
    slots 7..12 (same form). kind 'sig': in  the signature's affine x, y in slots 3..6; out: v251 = 1 iff psi(P) = [x]P.
    v252 / s[68:69] / s70: LDS column (11 spill slots) and workspace addressing as in the other routines.
    kind 'hash', two_lane (k_hash2, batches of at most a quarter of a round): lanes 2 j and 2 j + 1 are ONE message with workspace items of
    their own, next to each other; the even lane comes with u0, the odd one with u1 in the slots of u0, each runs ONE map_to_curve; then
    both fetch q0 from the even lane's item and q1 from the odd lane's and walk the rest together on identical values -- the products of the
    addition, the doublings and the glue in pairs (pair_products): H in BOTH items."""
    S = G2_SLOTS
    ad = S["AD"] if kind == "hash" else S["SIGAD"]
    B = {}
    st = {}
    names = ["add", "dbl", "fix"] + (["sswu", "sub"] + (["h_q1", "h_q0"] if two_lane else ["h_start"]) + ["h_base1", "h_after1", "h_psi2", "h_t3", "h_base2", "h_after2", "h_ad_t1", "h_ad_p"] if kind == "hash" else ["madd", "s_start", "s_compare"])
    for nm in names:                                # (the maps differ between the lanes, the doubling fix-up runs under its own exec mask: no pairs there)
        B[nm], st[nm] = build_g2(nm, ad, free_v=(BL_FREE_V if nm in ("h_q1", "h_q0") else None), pair_mode=(two_lane and nm not in ("sswu", "fix")))
    X = lambda nm: expand_calls_d(B[nm])
    ADD, SUB, LADDER = 50, 51, 52
    pro = ["s_mov_b64 s[80:81], s[30:31]", "s_waitcnt vmcnt(0)"] + shell_constants() + ["s_mov_b64 %s, exec" % EXEC_ALL]
    if two_lane:
        assert PAIR_EXEC == EXEC_ALL                # one register: the lanes that entered the routine
        pro += pair_prologue()

    def fixup():                                    # equal operands (same x, same y, neither at infinity): double the old acc on those lanes
        return ["s_and_b64 s[92:93], %s, %s" % (M_H0, M_R0), "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF1, "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF2,
                "s_and_b64 s[92:93], s[92:93], exec", "s_cbranch_scc1 3f"] + far_fwd(6) + ["3:", "s_mov_b64 exec, s[92:93]"] + X("fix") + ["6:", "s_mov_b64 exec, %s" % EXEC_ALL]
    subs = ["%d:" % ADD] + X("add" if kind == "hash" else "madd") + fixup() + ["s_setpc_b64 s[98:99]"]      # sig: the base point is affine
    if kind == "hash":
        subs += ["%d:" % SUB] + X("sub") + fixup() + ["s_setpc_b64 s[98:99]"]
    # [|x|] acc with the base in AD: runs of doublings, an addition of the base after each run but the last
    lad = ["%d:" % LADDER, "s_mov_b32 s78, 0", "4:", "s_mov_b32 s39, %d" % RUNS[5]]
    for ph in range(5):
        lad += ["s_cmp_eq_u32 s78, %d" % ph, "s_cselect_b32 s39, %d, s39" % RUNS[ph]]
    lad += [".p2align 6", "1:"] + X("dbl") + ["s_sub_u32 s39, s39, 1", "s_cmp_lg_u32 s39, 0", "s_cbranch_scc0 2f"] + far_back(1) + ["2:"]
    lad += ["s_cmp_eq_u32 s78, 5", "s_cbranch_scc0 3f", "s_setpc_b64 s[96:97]", "3:"]
    lad += call_sub(ADD) + ["s_add_u32 s78, s78, 1"] + far_back(4)

    def call_ladder():
        return ["s_getpc_b64 s[96:97]", "7:", "s_add_u32 s96, s96, 8f-7b", "s_addc_u32 s97, s97, 0"] + far_fwd(LADDER) + ["8:"]
    if kind == "hash":
        # the two map_to_curve evaluations: u0 -> q0, u1 -> q1 (records six slots apart)
        if two_lane:                                # one map per lane; then the even lanes alone, their neighbour's point one item (4 bytes) further on
            main = ["s_mov_b32 %s, 0" % GKOFF] + X("sswu") + ["s_waitcnt vmcnt(0)"]
            # VOFF = the lane offset of the item that holds q1 (the odd lane's): own + 4 on even lanes, own on odd lanes; then of q0's: 4 less
            B["voff_q1"] = ["v_mbcnt_lo_u32_b32 v254, -1, 0", "v_mbcnt_hi_u32_b32 v254, -1, v254", "v_and_b32_e64 v254, 1, v254", "v_lshlrev_b32_e64 v254, 2, v254",
                            "v_sub_u32_e64 %s, %s, v254" % (VOFF, LADDR), "v_add_u32_e64 %s, 4, %s" % (VOFF, VOFF)]
            B["voff_q0"] = ["v_sub_u32_e64 %s, %s, 4" % (VOFF, VOFF)]
            main += B["voff_q1"] + X("h_q1") + B["voff_q0"] + X("h_q0")
        else:
            main = ["s_mul_i32 s72, %s, %d" % (GSTRIDE, 12 * 6), "s_mov_b32 %s, 0" % GKOFF] + X("sswu") + ["s_mov_b32 %s, s72" % GKOFF] + X("sswu")
            main += X("h_start")
        main += call_sub(ADD) + X("h_base1") + call_ladder() + X("h_after1") + X("dbl") + X("h_psi2") + call_sub(SUB)
        main += X("h_t3") + call_sub(ADD) + X("h_base2") + call_ladder() + X("h_after2") + call_sub(ADD)
        main += X("h_ad_t1") + call_sub(SUB) + X("h_ad_p") + call_sub(SUB)
        epi = ["s_waitcnt vmcnt(0)"]
        B0, B1, B2, B5, B6 = (lambda j: "v%d" % j), (lambda j: "v%d" % (14 + j)), (lambda j: "v%d" % (28 + j)), (lambda j: "v%d" % (70 + j)), (lambda j: "v%d" % (84 + j))
        epi += ["v_mov_b32_e32 %s, 0x%08x" % (B2(j), dgt) for j, dgt in enumerate(digits_of(K384))]
        for i in range(3):
            epi += ["v_accvgpr_read_b32 %s, a%d" % (B0(j), vb(2 * i) + j) for j in range(14)]
            epi += ["v_accvgpr_read_b32 %s, a%d" % (B1(j), vb(2 * i + 1) + j) for j in range(14)]
            epi += ["CALL mbls_fp2_mulfp_d_asm_fn"]
            for h, Bk in ((0, B5), (1, B6)):
                epi += seq_reduce(Bk) + seq_canonical(Bk) + seq_to32(Bk) + seq_gstore(Bk, S["H"] + 2 * i + h)
    else:
        main = X("s_start") + call_ladder() + X("s_compare")
        # (ia & ib) | (!ia & !ib & ex & ey)
        epi = ["s_and_b64 s[92:93], %s, %s" % (M_H0, M_R0), "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF1, "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF2,
               "s_and_b64 %s, %s, %s" % (G2M_TMP0, M_INF1, M_INF2), "s_or_b64 s[92:93], s[92:93], %s" % G2M_TMP0, "v_cndmask_b32_e64 v251, 0, 1, s[92:93]"]
    epi += ["s_waitcnt vmcnt(0)", "s_mov_b64 s[30:31], s[80:81]"]
    ret = ["s_setpc_b64 s[30:31]"]                  # the routine's own return: the subroutines follow it
    pieces = dict(B, pro=pro, epi=epi)
    return pro + main + expand_calls_d(epi) + ret + lad + subs, pieces, st          # every internal call is a forward jump


# ---- one level of the n-pairing paths' trees with ONE LANE per product (the wide levels; the narrow ones run on the cooperative engine):
# the lane's item <- its value (op) the value of the item `partner` further on; the partner's byte offset in GKOFF (kinds 'gka').
def prog_f12_treemul():
    p = Prog()
    a = six([p.live_in(("g", FEXP_IN_SLOT + i)) for i in range(12)])            # rematerialisable from their workspace homes
    b = six([p.live_in(("gka", FEXP_IN_SLOT + i)) for i in range(12)])
    acc_store(p, mul12(p, a, b))
    return p


def f12_tree_routine():
    """In:  slots 13..24 of this lane's item and of the item s71 bytes further on (Miller values, 2^384 domain); v252 / s[68:69] / s70 as in the
    other routines (11 LDS spill slots). Out: their product in v108..v251 (twelve groups of 12 words, canonical, 2^384 domain)."""
    p = prog_f12_treemul()
    inb = {v: G_IN for v in p.init_loc}
    al = AllocD(p, inb, n_lds=11, lds_base=0, a_pool=list(range(NA)))
    body = al.run()
    for dst, B in al.stored.items():
        assert B.vlo >= STATE_IN.vlo and B.vhi <= STATE_IN.vhi and B.dhi <= M28, (dst, B)
    pro = ["s_mov_b64 s[80:81], s[30:31]", "s_waitcnt vmcnt(0)"] + shell_constants()
    epi = f_out_epilogue("s[80:81]")
    return pro + expand_calls_d(body) + expand_calls_d(epi), dict(pro=pro, body=body, epi=epi), al.stats


def prog_g2_tree_start():
    """acc = the lane's own sum (slots 25..30), AD staging slots <- the partner's (both Jacobian, 2^384-domain words)"""
    p = Prog()
    own = [prog_reduce(p, p.live_in(("g", BL_OUT + i))) for i in range(6)]
    oth = [prog_reduce(p, p.live_in(("gka", BL_OUT + i))) for i in range(6)]
    pt_park(p, [(oth[0], oth[1]), (oth[2], oth[3]), (oth[4], oth[5])], G2_SLOTS["SIGAD"])
    pt_store_acc(p, [(own[0], own[1]), (own[2], own[3]), (own[4], own[5])])
    return p


def g2_tree_routine():
    """slots 25..30 of this lane's item <- that point + the same slots of the item s71 bytes further on (g2_add's case handling; slots
    43..48 are scratch)"""
    S = G2_SLOTS
    inb = lambda p: {v: (G2_IN if l[0] == "a" else PACKED if l[0] in ("gd", "gk") else G_IN) for v, l in p.init_loc.items()}
    bodies = {}
    for nm, p in (("start", prog_g2_tree_start()), ("add", prog_g2_add(S["SIGAD"], False)), ("fix", prog_g2_dbl_d(6, 0))):
        al = AllocD(p, inb(p), n_lds=11, lds_base=0, a_pool=list(range(NA)))
        bodies[nm] = al.run()
        for dst, B in getattr(al, "stored", {}).items():
            assert B.vlo >= G2_IN.vlo and B.vhi <= G2_IN.vhi and B.dlo >= 0 and B.dhi <= M28, (nm, dst, B)
    X = lambda nm: expand_calls_d(bodies[nm])
    pro = ["s_mov_b64 s[80:81], s[30:31]", "s_waitcnt vmcnt(0)"] + shell_constants() + ["s_mov_b64 %s, exec" % EXEC_ALL]
    main = X("start") + ["s_waitcnt vmcnt(0)"] + X("add")
    main += ["s_and_b64 s[92:93], %s, %s" % (M_H0, M_R0), "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF1, "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF2,
             "s_and_b64 s[92:93], s[92:93], exec", "s_cbranch_scc1 3f"] + far_fwd(6) + ["3:", "s_mov_b64 exec, s[92:93]"] + X("fix") + ["6:", "s_mov_b64 exec, %s" % EXEC_ALL]
    epi = ["s_waitcnt vmcnt(0)"]
    B0, B1, B2, B5, B6 = (lambda j: "v%d" % j), (lambda j: "v%d" % (14 + j)), (lambda j: "v%d" % (28 + j)), (lambda j: "v%d" % (70 + j)), (lambda j: "v%d" % (84 + j))
    epi += ["v_mov_b32_e32 %s, 0x%08x" % (B2(j), dgt) for j, dgt in enumerate(digits_of(K384))]
    for i in range(3):
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B0(j), vb(2 * i) + j) for j in range(14)]
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B1(j), vb(2 * i + 1) + j) for j in range(14)]
        epi += ["CALL mbls_fp2_mulfp_d_asm_fn"]
        for h, Bk in ((0, B5), (1, B6)):
            epi += seq_reduce(Bk) + seq_canonical(Bk) + seq_to32(Bk) + seq_gstore(Bk, BL_OUT + 2 * i + h)
    epi += ["s_waitcnt vmcnt(0)", "s_mov_b64 s[30:31], s[80:81]"]
    return pro + main + expand_calls_d(epi), dict(bodies, pro=pro, epi=epi), {}


# ---- [r] apk for verify_multiple (reference src/aggregates.rs:293): the same signed 4-bit windows in G1
G1B_TAB = 109                         # eight records of three slots: 1 P .. 8 P (slots 109..132, behind the final exponentiation's records)
G1B_FREE_V = list(range(8, 17))


# The scalar's signed 4-bit digits (shell code of both blinding routines): r' = r + 0x8888888888888888 = sum e_j 16^j + carry 2^64, so
# r = sum (e_j - 8) 16^j + carry 2^64 with digits in [-8, 7] and a 17th digit 0 / 1 (v247). Per window: M_NEGQ = the digit is negative,
# M_ZEROQ = it is zero, VOFF = lane offset of table record |digit| - 1.
BLIND_RPRIME = ["v_add_co_u32_e32 v248, vcc, 0x88888888, v248", "v_mov_b32_e32 v247, 0x88888888", "v_addc_co_u32_e32 v249, vcc, v249, v247, vcc",
                "v_cndmask_b32_e64 v247, 0, 1, vcc"]
BLIND_TOP = ["s_mov_b64 %s, 0" % M_NEGQ, "v_cmp_eq_u32_e64 %s, 0, v247" % M_ZEROQ, "v_mov_b32_e32 %s, %s" % (VOFF, LADDR)]
BLIND_DIGIT = ["v_lshrrev_b64 v[246:247], s38, v[248:249]", "v_and_b32_e32 v246, 15, v246", "v_subrev_u32_e32 v246, 8, v246",          # d = e - 8 in [-8, 7]
               "v_cmp_gt_i32_e64 %s, 0, v246" % M_NEGQ, "v_cmp_eq_u32_e64 %s, 0, v246" % M_ZEROQ,
               "v_sub_u32_e32 v247, 0, v246", "v_max_i32_e32 v246, v246, v247", "v_max_i32_e32 v246, 1, v246", "v_subrev_u32_e32 v246, 1, v246",
               "v_mul_lo_u32 v246, v246, s72", "v_add_u32_e32 %s, %s, v246" % (VOFF, LADDR)]


# CONSTANT-TIME table access (signing: the scalar is a secret key). The variable-time form above reads ONE record at an address computed from the
# window digit. Here every window reads ALL eight records of the lane's table in the same order and keeps the one it needs by selection
# (v_cndmask on |digit| - 1 = v246): addresses, instruction stream and the number of memory operations do not depend on the scalar. The selected
# record goes to slots BL_SEL.. of the lane's own item, where the table addition (build_g2 "addt_ct") fetches it like any other operand.
BLIND_TOP_CT = BLIND_TOP[:2] + ["v_mov_b32_e32 v246, 0"]
BLIND_DIGIT_CT = BLIND_DIGIT[:-2]                      # ... v246 = max(|d|, 1) - 1; no address arithmetic


def blind_scan_ct(tab=None, sel=None, words=6):
    """v246 = record index 0..7 per lane -> record (tab + 6 e .. + 5) of every e read, the lane's one written to sel.. (packed words, bit for bit).
    Between the routine's programs nothing lives in VGPR blocks 0..6 (v0..v97): v0..v83 take records 1..7 of one slot, v84..v95 record 0 and the result."""
    tab = BL_TAB if tab is None else tab
    sel = BL_SEL if sel is None else sel
    L = []
    for w in range(words):
        for e in range(8):
            base = 84 if e == 0 else 12 * (e - 1)
            L += seq_gaddr(tab + 6 * e + w)
            for j in range(12):
                L.append("global_load_dword v%d, %s, %s" % (base + j, LADDR, GADDR))
                if j < 11:
                    L += ["s_add_u32 s74, s74, %s" % GSTRIDE, "s_addc_u32 s75, s75, 0"]
        L += ["s_waitcnt vmcnt(0)"]
        for e in range(1, 8):
            L += ["v_cmp_eq_u32_e64 vcc, %d, v246" % e]
            L += ["v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (84 + j, 84 + j, 12 * (e - 1) + j) for j in range(12)]
        L += seq_gstore(lambda j: "v%d" % (84 + j), sel + w) + ["s_waitcnt vmcnt(0)"]
    return L


def prog_g1_jadd(slot, table):
    """acc <- acc + (the Jacobian point in slots slot..slot+2; table: the record each lane selects, negated / ignored by M_NEGQ / M_ZEROQ):
    add-2007-bl with g1_add's case handling by selection; the old acc goes to AGPR blocks 5..7 for the doubling fix-up"""
    p = Prog()
    X1, Y1, Z1 = [p.live_in(("a", i)) for i in range(3)]
    X2, Y2, Z2 = [p.live_in(("gv" if table else "gd", slot + i)) for i in range(3)]
    if table:
        Y2 = p.sel(M_NEGQ, Y2, p.neg(Y2))
    Z1Z1, Z2Z2 = p.mulpair(Z1, Z1, Z2, Z2)
    U1, U2 = p.mulpair(X1, Z2Z2, X2, Z1Z1)
    T1, T2 = p.mulpair(Y1, Z2, Y2, Z1)
    S1, S2 = p.mulpair(T1, Z2Z2, T2, Z1Z1)
    H = p.sub(U2, U1)
    RR = p.scale(p.sub(S2, S1), 2)
    p.iszero(H, M_H0); p.iszero(RR, M_R0); p.iszero(Z1, M_INF1); p.iszero(Z2, M_INF2)
    if table:
        p.mask_or(M_INF2, M_INF2, M_ZEROQ)
    H2 = p.scale(H, 2)
    I, RR2 = p.mulpair(H2, H2, RR, RR)
    J, V = p.mulpair(H, I, U1, I)
    X3 = p.sub(p.sub(RR2, J), p.scale(V, 2))
    ZZ = p.add(Z1, Z2)
    M0, ZS = p.mulpair(RR, p.sub(V, X3), ZZ, ZZ)
    M1, Z3 = p.mulpair(S1, J, p.sub(p.sub(ZS, Z1Z1), Z2Z2), H)
    Y3 = p.sub(M0, p.scale(M1, 2))
    out = [p.sel(M_INF1, o, q) for o, q in zip((X3, Y3, Z3), (X2, Y2, Z2))]
    out = [p.sel(M_INF2, o, a) for o, a in zip(out, (X1, Y1, Z1))]
    for i, v in enumerate((X1, Y1, Z1)):
        p.store(v, ("a", 5 + i))
    for i, v in enumerate(out):
        p.store(prog_reduce(p, v), ("a", i))
    return p


def prog_g1b_glue(which):
    p = Prog()
    acc = lambda: [p.live_in(("a", i)) for i in range(3)]
    if which == "start":                             # acc = the point in slots 0..2 (Jacobian, 2^384 domain words)
        for i in range(3):
            p.store(prog_reduce(p, p.live_in(("g", i))), ("a", i))
    elif which == "tab":                             # table record (run-time offset) <- acc; acc stays
        a = acc()
        for i, v in enumerate(a):
            p.ops.append(("storep", [], [v], ("k", G1B_TAB + i)))
        for i, v in enumerate(a):
            p.store(v, ("a", i))
    elif which == "inf":
        for i, c in enumerate((0, ONE_D, 0)):
            p.store(prog_reduce(p, p.const(c)), ("a", i))
    else:
        raise ValueError(which)
    return p


def build_g1b(which):
    p = {"add": lambda: prog_g1_jadd(G1B_TAB, False), "addt": lambda: prog_g1_jadd(G1B_TAB, True), "dbl": lambda: prog_g1_dbl(0),
         "fix": lambda: prog_g1_dbl(5)}.get(which, lambda: prog_g1b_glue(which))()
    inb = {v: (STATE_IN if l[0] == "a" else PACKED if l[0] in ("gd", "gv", "gk") else G_IN) for v, l in p.init_loc.items()}
    al = AllocD(p, inb, n_lds=0, a_pool=list(range(8, NA)), free_v=G1B_FREE_V)
    body = al.run()
    for dst, B in getattr(al, "stored", {}).items():
        assert B.vlo >= STATE_IN.vlo and B.vhi <= STATE_IN.vhi and B.dlo >= 0 and B.dhi <= M28, (which, dst, B)
    return body, al.stats


def g1_blind_routine():
    """[r] P for the Jacobian G1 point in workspace slots 0..2 (2^384 domain) and the lane's 64-bit scalar r in v[248:249], back into slots 0..2
    (canonical): signed 4-bit windows over a table 1 P .. 8 P in workspace records, each lane fetching its own (see g2_blind_routine).
    v252 / s[68:69] / s70: workspace addressing of the other routines (no LDS: the lane offset in v252 may be any address the caller folds
    into the base)."""
    B, st = {}, {}
    for nm in ["add", "addt", "dbl", "fix", "start", "tab", "inf"]:
        B[nm], st[nm] = build_g1b(nm)
    X = lambda nm: expand_calls_d(B[nm])
    ADD, ADDT, DBL4 = 50, 53, 54
    pro = ["s_mov_b64 s[80:81], s[30:31]", "s_waitcnt vmcnt(0)"] + shell_constants() + ["s_mov_b64 %s, exec" % EXEC_ALL,
           "s_mul_i32 s72, %s, %d" % (GSTRIDE, 12 * 3)]
    pro += BLIND_RPRIME

    def fixup():
        return ["s_and_b64 s[92:93], %s, %s" % (M_H0, M_R0), "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF1, "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF2,
                "s_and_b64 s[92:93], s[92:93], exec", "s_cbranch_scc1 3f"] + far_fwd(6) + ["3:", "s_mov_b64 exec, s[92:93]"] + X("fix") + ["6:", "s_mov_b64 exec, %s" % EXEC_ALL]
    subs = ["%d:" % ADD] + X("add") + fixup() + ["s_setpc_b64 s[98:99]"]
    subs += ["%d:" % ADDT] + X("addt") + fixup() + ["s_setpc_b64 s[98:99]"]
    dbl4 = ["%d:" % DBL4, "s_mov_b32 s39, 4", ".p2align 6", "1:"] + X("dbl") + ["s_sub_u32 s39, s39, 1", "s_cmp_lg_u32 s39, 0", "s_cbranch_scc0 2f"] + far_back(1) + ["2:", "s_setpc_b64 s[96:97]"]

    def call_far(label):
        return ["s_getpc_b64 s[96:97]", "7:", "s_add_u32 s96, s96, 8f-7b", "s_addc_u32 s97, s97, 0"] + far_fwd(label) + ["8:"]
    main = X("start") + ["s_mov_b32 %s, 0" % GKOFF] + X("tab") + X("dbl") + ["s_mov_b32 %s, s72" % GKOFF] + X("tab")
    main += ["s_mov_b32 s79, 6", "5:"] + call_sub(ADD) + ["s_add_u32 %s, %s, s72" % (GKOFF, GKOFF)] + X("tab")
    main += ["s_sub_u32 s79, s79, 1", "s_cmp_lg_u32 s79, 0", "s_cbranch_scc0 2f"] + far_back(5) + ["2:", "s_waitcnt vmcnt(0)"]
    main += X("inf")
    main += BLIND_TOP + call_sub(ADDT)
    main += ["s_mov_b32 s38, 60", "5:"] + call_far(DBL4)
    main += BLIND_DIGIT
    main += call_sub(ADDT)
    main += ["s_cmp_eq_u32 s38, 0", "s_cbranch_scc1 2f", "s_sub_u32 s38, s38, 4"] + far_back(5) + ["2:"]
    epi = ["s_waitcnt vmcnt(0)"]
    B0, B1, B2, B5, B6 = (lambda j: "v%d" % j), (lambda j: "v%d" % (14 + j)), (lambda j: "v%d" % (28 + j)), (lambda j: "v%d" % (70 + j)), (lambda j: "v%d" % (84 + j))
    epi += ["v_mov_b32_e32 %s, 0x%08x" % (B2(j), dgt) for j, dgt in enumerate(digits_of(K384))]
    for half in range(2):
        srcs = (0, 1) if half == 0 else (2, 2)
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B0(j), vb(srcs[0]) + j) for j in range(14)]
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B1(j), vb(srcs[1]) + j) for j in range(14)]
        epi += ["CALL mbls_fp2_mulfp_d_asm_fn"]
        for h, Bk in ((0, B5), (1, B6))[:2 if half == 0 else 1]:
            epi += seq_reduce(Bk) + seq_canonical(Bk) + seq_to32(Bk) + seq_gstore(Bk, 2 * half + h)
    epi += ["s_waitcnt vmcnt(0)", "s_mov_b64 s[30:31], s[80:81]"]
    ret = ["s_setpc_b64 s[30:31]"]
    pieces = dict(B, pro=pro, epi=epi, rprime=BLIND_RPRIME, top=BLIND_TOP, digit=BLIND_DIGIT)
    return pro + main + expand_calls_d(epi) + ret + dbl4 + subs, pieces, st


def g2_blind_routine(ct=False, two_lane=False):
    """ct: the constant-time form used by signing (blind_scan_ct) -- same arithmetic, table records by scan + selection instead of by address.
    two_lane (small batches of verify_multiple, k_blind_sig2_d): lanes 2 j and 2 j + 1 are ONE signature -- same workspace item, same LDS column, same
    scalar --, both walk the whole routine on identical values and the independent products of a doubling / an addition go in pairs (pair_products).
    The signature phase of verify_multiple_aggregate_signatures (reference src/aggregates.rs:274-276, :303) as ONE routine: the subgroup
    test psi(P) = [x]P of the decoded signature, then [r] P for the lane's 64-bit blinding scalar r by signed 4-bit windows:
    r + 0x8888888888888888 = sum e_j 16^j (+ a carry digit), r = sum (e_j - 8) 16^j + carry 2^64; the table 1 P .. 8 P lives in workspace
    records (BL_TAB, one per multiple) and each lane fetches ITS record (kind 'gv'); 64 doublings + 17 additions instead of 64 + 64.
    In:  slots 3..6 = the signature's affine x, y (2^384 domain); v[248:249] = r; v251 = 0 (wave-uniform; non-zero skips the subgroup test and
         is returned as is); v252 / s[68:69] / s70 as in the other routines.
    Out: v251 = 1 iff psi(P) = [x]P; slots 25..30 = [r] P (Jacobian, canonical, 2^384 domain). Slots 43..48, 49..96 are scratch."""
    S = G2_SLOTS
    B, st = {}, {}
    for nm in ["madd", "addt", "dbl", "fix", "s_start", "s_compare", "b_tab", "b_start", "b_inf"]:
        B[nm], st[nm] = build_g2("addt_ct" if (ct and nm == "addt") else nm, S["SIGAD"], free_v=BL_FREE_V, pair_mode=(two_lane and nm != "fix"))
    top, digit = (BLIND_TOP_CT, BLIND_DIGIT_CT) if ct else (BLIND_TOP, BLIND_DIGIT)
    scan = blind_scan_ct() if ct else []
    X = lambda nm: expand_calls_d(B[nm])
    ADD, ADDT, LADDER, DBL4 = 50, 53, 52, 54
    pro = ["s_mov_b64 s[80:81], s[30:31]", "s_waitcnt vmcnt(0)"] + shell_constants() + ["s_mov_b64 %s, exec" % EXEC_ALL,
           "s_mul_i32 s72, %s, %d" % (GSTRIDE, 12 * 6)]
    if two_lane:
        assert PAIR_EXEC == EXEC_ALL and not ct
        pro += pair_prologue()
    # r' = r + 0x8888888888888888, its carry is the 17th digit
    pro += BLIND_RPRIME

    def fixup():
        return ["s_and_b64 s[92:93], %s, %s" % (M_H0, M_R0), "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF1, "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF2,
                "s_and_b64 s[92:93], s[92:93], exec", "s_cbranch_scc1 3f"] + far_fwd(6) + ["3:", "s_mov_b64 exec, s[92:93]"] + X("fix") + ["6:", "s_mov_b64 exec, %s" % EXEC_ALL]
    subs = ["%d:" % ADD] + X("madd") + fixup() + ["s_setpc_b64 s[98:99]"]            # ladder and table: + P with P affine
    subs += ["%d:" % ADDT] + X("addt") + fixup() + ["s_setpc_b64 s[98:99]"]
    lad = ["%d:" % LADDER, "s_mov_b32 s78, 0", "4:", "s_mov_b32 s39, %d" % RUNS[5]]
    for ph in range(5):
        lad += ["s_cmp_eq_u32 s78, %d" % ph, "s_cselect_b32 s39, %d, s39" % RUNS[ph]]
    lad += [".p2align 6", "1:"] + X("dbl") + ["s_sub_u32 s39, s39, 1", "s_cmp_lg_u32 s39, 0", "s_cbranch_scc0 2f"] + far_back(1) + ["2:"]
    lad += ["s_cmp_eq_u32 s78, 5", "s_cbranch_scc0 3f", "s_setpc_b64 s[96:97]", "3:"]
    lad += call_sub(ADD) + ["s_add_u32 s78, s78, 1"] + far_back(4)
    dbl4 = ["%d:" % DBL4, "s_mov_b32 s39, 4", ".p2align 6", "1:"] + X("dbl") + ["s_sub_u32 s39, s39, 1", "s_cmp_lg_u32 s39, 0", "s_cbranch_scc0 2f"] + far_back(1) + ["2:", "s_setpc_b64 s[96:97]"]

    def call_far(label):
        return ["s_getpc_b64 s[96:97]", "7:", "s_add_u32 s96, s96, 8f-7b", "s_addc_u32 s97, s97, 0"] + far_fwd(label) + ["8:"]
    # the subgroup test (skipped when the caller passes v251 != 0 in its first active lane: a point known to be in G2 -- signing)
    main = X("s_start") + ["v_readfirstlane_b32 s38, v251", "s_cmp_lg_u32 s38, 0", "s_cbranch_scc0 3f"] + far_fwd(9) + ["3:"]
    main += call_far(LADDER) + X("s_compare")
    main += ["s_and_b64 s[92:93], %s, %s" % (M_H0, M_R0), "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF1, "s_andn2_b64 s[92:93], s[92:93], %s" % M_INF2,
             "s_and_b64 %s, %s, %s" % (G2M_TMP0, M_INF1, M_INF2), "s_or_b64 s[92:93], s[92:93], %s" % G2M_TMP0, "v_cndmask_b32_e64 v251, 0, 1, s[92:93]", "9:"]
    # the table: record e holds (e + 1) P
    main += X("b_start") + ["s_mov_b32 %s, 0" % GKOFF] + X("b_tab") + X("dbl") + ["s_mov_b32 %s, s72" % GKOFF] + X("b_tab")
    main += ["s_mov_b32 s79, 6", "5:"] + call_sub(ADD) + ["s_add_u32 %s, %s, s72" % (GKOFF, GKOFF)] + X("b_tab")
    main += ["s_sub_u32 s79, s79, 1", "s_cmp_lg_u32 s79, 0", "s_cbranch_scc0 2f"] + far_back(5) + ["2:", "s_waitcnt vmcnt(0)"]
    # the windows, top digit (0 or 1) first
    main += X("b_inf")
    main += top + scan + call_sub(ADDT)
    main += ["s_mov_b32 s38, 60", "5:"] + call_far(DBL4)
    main += digit + scan
    main += call_sub(ADDT)
    main += ["s_cmp_eq_u32 s38, 0", "s_cbranch_scc1 2f", "s_sub_u32 s38, s38, 4"] + far_back(5) + ["2:"]
    # [r] P -> slots 25..30, canonical words of the 2^384 domain
    epi = ["s_waitcnt vmcnt(0)"]
    B0, B1, B2, B5, B6 = (lambda j: "v%d" % j), (lambda j: "v%d" % (14 + j)), (lambda j: "v%d" % (28 + j)), (lambda j: "v%d" % (70 + j)), (lambda j: "v%d" % (84 + j))
    epi += ["v_mov_b32_e32 %s, 0x%08x" % (B2(j), dgt) for j, dgt in enumerate(digits_of(K384))]
    for i in range(3):
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B0(j), vb(2 * i) + j) for j in range(14)]
        epi += ["v_accvgpr_read_b32 %s, a%d" % (B1(j), vb(2 * i + 1) + j) for j in range(14)]
        epi += ["CALL mbls_fp2_mulfp_d_asm_fn"]
        for h, Bk in ((0, B5), (1, B6)):
            epi += seq_reduce(Bk) + seq_canonical(Bk) + seq_to32(Bk) + seq_gstore(Bk, BL_OUT + 2 * i + h)
    epi += ["s_waitcnt vmcnt(0)", "s_mov_b64 s[30:31], s[80:81]"]
    ret = ["s_setpc_b64 s[30:31]"]
    pieces = dict(B, pro=pro, epi=epi, rprime=BLIND_RPRIME, top=top, digit=digit, scan=scan)
    return pro + main + expand_calls_d(epi) + ret + lad + dbl4 + subs, pieces, st


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    # MBLS_GEN_OUT_DIR: write there instead of over the tracked file (the freshness tests generate into a temporary directory and compare)
    path = os.path.join(os.environ.get("MBLS_GEN_OUT_DIR") or os.path.join(os.path.dirname(here), "milagro_bls_amd", "csrc"), "mbls_towerd_asm.inc")
    txt = "// GENERATED by tools/gen_tower_d.py -- do not edit.\n"
    body, stats = build_cyc_sqr_d()
    print("cyc_sqr_d", len(body), "lines", stats)
    full, pieces, st = miller_loop_d_routine()
    txt += emit("MBLS_MILLER_LOOP_D_ASM", full) + "\n"
    for kname, v in st.items():
        print("miller", kname, len(pieces[kname]), "lines", v)
    full, pieces, st = miller_loop_d_routine((1,))
    txt += emit("MBLS_MILLER_LOOP_1P_D_ASM", full) + "\n"
    full, pieces, st = miller_loop_d_routine((1,), pair_mode=True)
    txt += emit("MBLS_MILLER_LOOP_1P_PAIR_D_ASM", full) + "\n"
    print("miller single pair on TWO lanes: dbl", len(pieces["dbl"]), "lines", st["dbl"])
    txt += "#define MBLS_PAIR_D_ASM_CLOBBERS \"s82\", \"s83\", \"s94\", \"s95\"\n"
    txt += "#define MBLS_PAIR_EXEC_ASM_CLOBBERS \"s82\", \"s83\"\n"
    print("miller single pair: dbl", len(pieces["dbl"]), "lines", st["dbl"])
    sgm = '"s30","s31","s36","s37","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s73","s74","s75","s76","s77","s78","vcc","scc","memory"'
    fout = set(r for b in F_OUT for r in range(b, b + 12))
    # what mbls_lanes.h (lane_sig_verdict) relies on: when the two-pair loop returns, pair 0's running point T = [|x|] Q_0 sits in these
    # workspace slots (x.c0, x.c1, y.c0, y.c1, z.c0, z.c1) as packed words of the 2^392 domain, representatives in (0.5 p, 1.5 p)
    txt += "#define MBLS_GEN_MILLER_T0_SLOT %d\n#define MBLS_GEN_MILLER_T_DOMAIN_BITS 392\n#define MBLS_GEN_MILLER_T_PACKED 1\n" % T_SLOT(0, 0, 0)
    assert [T_SLOT(0, e, i) for e in range(3) for i in range(2)] == list(range(T_SLOT(0, 0, 0), T_SLOT(0, 0, 0) + 6))
    assert T_SLOT(1, 0, 0) == T_SLOT(0, 0, 0) + 6            # pair 1's point (the one the one-pair routine walks) follows pair 0's
    txt += "// the Miller-loop routine returns f in twelve register groups\n"
    txt += "#define MBLS_MILLER_D_OUT_REGS(x) " + ", ".join('"={v[%d:%d]}"(x##%d)' % (b, b + 11, i) for i, b in enumerate(F_OUT)) + "\n"
    txt += "#define MBLS_MILLER_D_ASM_CLOBBERS %s,%s, \\\n    %s\n" % (
        ",".join('"v%d"' % i for i in range(256) if i not in fout and i not in (252, 253) and i not in UNTOUCHED_V), ",".join('"a%d"' % i for i in range(252)), sgm)
    full, pieces, st = final_exp_d_routine()
    txt += emit("MBLS_FINAL_EXP_D_ASM", full) + "\n"
    for kname, v in st.items():
        print("final_exp_d", kname, len(pieces[kname]), "lines", v)
    full2, pieces2, st2 = final_exp_d_routine(two_lane=True)
    txt += emit("MBLS_FINAL_EXP2_D_ASM", full2) + "\n"
    print("final_exp_d two lanes per item: csqr", len(pieces2["csqr"]), "pstart", len(pieces2["pstart"]), "psave", len(pieces2["psave"]), "lines")
    txt += "#define MBLS_FINAL_EXP_D_ASM_CLOBBERS MBLS_MILLER_D_ASM_CLOBBERS, \"v253\", \"s50\", \"s51\", \"s52\", \"s53\", \"s79\", \"s71\", \"s72\", \"s80\", \"s81\", " + ", ".join('\"s%d\"' % i for i in range(84, 100)) + "\n"
    for mode in ("raw", "indexed"):
        full, pieces, st = g1_aggregate_d_routine(mode)
        txt += emit("MBLS_G1_AGGREGATE_%s_D_ASM" % mode.upper(), full) + "\n"
        print("g1 aggregate", mode, "step", len(pieces["step"]), "lines", st["step"], "dbl", len(pieces["dbl"]))
    sga = sgm.replace('"vcc"', '"s50","s51","s52","s53","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","vcc"')
    txt += "#define MBLS_G1_AGG_D_ASM_CLOBBERS %s,%s, \\\n    %s\n" % (
        ",".join('"v%d"' % i for i in range(256) if i not in (248, 249, 250, 251, 252) and i not in UNTOUCHED_V), ",".join('"a%d"' % i for i in range(252)), sga)
    for kind, macro in (("sig", "MBLS_G2_SUBGROUP_D_ASM"), ("hash", "MBLS_G2_HASH_TAIL_D_ASM")):
        full, pieces, st = g2_group_routine(kind)
        txt += emit(macro, full) + "\n"
        print("g2 group routine", kind, len(full), "lines; add", len(pieces["add"]), st["add"])
    full, pieces, st = g2_group_routine("hash", two_lane=True)
    txt += emit("MBLS_G2_HASH_TAIL2_D_ASM", full) + "\n"
    print("g2 hash routine, two lanes per message:", len(full), "lines")
    full, pieces, st = g2_blind_routine()
    txt += emit("MBLS_G2_BLIND_D_ASM", full) + "\n"
    print("g2 blind routine", len(full), "lines; addt", len(pieces["addt"]), st["addt"])
    full, pieces, st = g2_blind_routine(two_lane=True)
    txt += emit("MBLS_G2_BLIND2_D_ASM", full) + "\n"
    print("g2 blind routine, two lanes per signature:", len(full), "lines; dbl", st["dbl"].get("pairs"), "addt", st["addt"].get("pairs"), "madd", st["madd"].get("pairs"), "pairs")
    full, pieces, st = g2_group_routine("sig", two_lane=True)
    txt += emit("MBLS_G2_SUBGROUP2_D_ASM", full) + "\n"
    print("g2 subgroup routine, two lanes per signature:", len(full), "lines")
    full, pieces, st = g2_blind_routine(ct=True)
    txt += emit("MBLS_G2_BLIND_CT_D_ASM", full) + "\n"
    print("g2 blind routine, constant-time table access:", len(full), "lines; scan", len(pieces["scan"]))
    sgb = sgm.replace('"vcc"', ",".join('"s%d"' % i for i in [38] + list(range(50, 54)) + [71, 72] + list(range(79, 100))) + ',"vcc"')
    txt += "#define MBLS_G2_BLIND_D_ASM_CLOBBERS %s,%s, \\\n    %s\n" % (
        ",".join('"v%d"' % i for i in range(256) if i not in (248, 249, 251, 252) and i not in UNTOUCHED_V), ",".join('"a%d"' % i for i in range(252)), sgb)
    full, pieces, st = f12_tree_routine()
    txt += emit("MBLS_F12_TREE_D_ASM", full) + "\n"
    print("f12 tree routine", len(full), "lines", st)
    full, pieces, st = g2_tree_routine()
    txt += emit("MBLS_G2_TREE_D_ASM", full) + "\n"
    print("g2 tree routine", len(full), "lines")
    # the tree routines take the partner's byte offset in s71 (an input of the call, so it is not in their clobber lists)
    txt += "#define MBLS_F12_TREE_D_ASM_CLOBBERS MBLS_MILLER_D_ASM_CLOBBERS, \"v253\", \"s50\", \"s51\", \"s52\", \"s53\", \"s79\", \"s72\", \"s80\", \"s81\", " + ", ".join('\"s%d\"' % i for i in range(84, 100)) + "\n"
    full, pieces, st = g1_blind_routine()
    txt += emit("MBLS_G1_BLIND_D_ASM", full) + "\n"
    print("g1 blind routine", len(full), "lines; addt", len(pieces["addt"]), st["addt"], "dbl", len(pieces["dbl"]))
    txt += "#define MBLS_G1_BLIND_D_ASM_CLOBBERS %s,%s, \\\n    %s\n" % (
        ",".join('"v%d"' % i for i in range(256) if i not in (248, 249, 252) and i not in UNTOUCHED_V), ",".join('"a%d"' % i for i in range(252)), sgb)
    sgt = sgm.replace('"vcc"', ",".join('"s%d"' % i for i in list(range(50, 54)) + [72] + list(range(79, 100))) + ',"vcc"')
    txt += "#define MBLS_G2_TREE_D_ASM_CLOBBERS %s,%s, \\\n    %s\n" % (
        ",".join('"v%d"' % i for i in range(256) if i != 252 and i not in UNTOUCHED_V), ",".join('"a%d"' % i for i in range(252)), sgt)
    sgg = sgm.replace('"vcc"', ",".join('"s%d"' % i for i in list(range(50, 54)) + [71, 72] + list(range(79, 100))) + ',"vcc"')
    txt += "#define MBLS_G2_GROUP_D_ASM_CLOBBERS %s,%s, \\\n    %s\n" % (
        ",".join('"v%d"' % i for i in range(256) if i not in (251, 252) and i not in UNTOUCHED_V), ",".join('"a%d"' % i for i in range(252)), sgg)
    full, pieces, st = g2_dbl_d_routine()
    txt += emit("MBLS_G2_DBL_D_ASM", full) + "\n"
    print("g2_dbl_d", len(pieces["body"]), "lines", st)
    g2r = set(r for b in G2D_ARG for r in range(b, b + 12))
    txt += "#define MBLS_G2D_ARG_REGS(x) " + ", ".join('"+{v[%d:%d]}"(x##%d)' % (b, b + 11, i) for i, b in enumerate(G2D_ARG)) + "\n"
    sg = '"s30","s31","s36","s37","s39","s40","s41","s42","s43","s44","s45","s46","s47","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s73","vcc","scc","memory"'
    txt += "#define MBLS_G2D_ASM_CLOBBERS %s,%s, \\\n    %s\n" % (
        ",".join('"v%d"' % i for i in range(256) if i not in g2r and i not in UNTOUCHED_V), ",".join('"a%d"' % i for i in range(252)), sg)
    with open(path, "w") as f:
        f.write(txt)
    print("wrote", path)


if __name__ == "__main__":
    main()
