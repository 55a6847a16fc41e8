#!/usr/bin/env python3
"""Digit-exact CPU model of the wave-cooperative engine (mbls_coop.h) running the microcode of tools/gen_coop.py: every step on every
lane with the integer semantics of the kernel -- 32-bit digit sums, the signed 64-bit Montgomery scan of mbls_fp_mul1_d_asm_fn
(tools/gen_fpd_asm.py signed_scan), the float quotient estimate of the reduction -- asserting on any 32 / 64-bit overflow. Test
infrastructure (tests/test_coop_cpu.py): the programs against the big-integer model of oracle/pymodel."""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_fp_asm import P, P28, NP28, M28  # noqa: E402
import gen_coop as G  # noqa: E402

PTOP = P >> 364
R384 = 1 << 384


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


RECIP = f32(1.0 / PTOP)


def i32(x):
    assert -(1 << 31) <= x < (1 << 31), "32-bit digit overflow"
    return x


def mont(a, b):
    """mbls_fp_mul1_d_asm_fn: a * b / 2^392 on signed digits (tools/gen_fpd_asm.py signed_scan)"""
    acc, q, res = 0, [0] * 14, [0] * 14
    for k in range(28):
        for i in range(max(0, k - 13), min(k, 13) + 1):
            acc += a[i] * b[k - i]
        if k < 14:
            for i in range(k):
                acc += P28[k - i] * q[i]
        else:
            for i in range(k - 13, 14):
                acc += P28[k - i] * q[i]
        assert -(1 << 63) <= acc < (1 << 63), "64-bit column overflow"
        if k < 14:
            q[k] = ((acc & 0xFFFFFFFF) * NP28) & M28
            acc += P28[0] * q[k]
            assert acc & M28 == 0
            acc >>= 28
        elif k < 27:
            res[k - 14] = acc & M28
            acc >>= 28
        else:
            res[13] = i32(acc)
    return res


def qpass(d, nq):
    """tools/gen_tower_d.py seq_qpass: acc = carry + nq p_j + d_j over a signed 64-bit running sum"""
    out, acc = [0] * 14, 0
    for j in range(14):
        acc = acc + nq * P28[j] + d[j]
        assert -(1 << 63) <= acc < (1 << 63)
        if j < 13:
            out[j] = acc & M28
            acc >>= 28
        else:
            out[13] = i32(acc)
    return out


def rndne(x):
    import math
    fl = math.floor(x)
    diff = x - fl
    if diff > 0.5 or (diff == 0.5 and fl % 2 == 1):
        return fl + 1
    return fl


RECIPD = 1.0 / PTOP


def reduce_digits(d):
    """the LIN step's reduction (coop_reduce in mbls_coop.h): 64-bit digit sums -> carry pass -> quotient estimate from the true top digit
    (double arithmetic) -> subtraction pass: the representative nearest to zero, digits 0..12 normalised"""
    n, acc = [0] * 14, 0
    for j in range(14):
        acc += d[j]
        assert -(1 << 63) <= acc < (1 << 63)
        if j < 13:
            n[j] = acc & M28
            acc >>= 28
        else:
            n[13] = acc
    q = int(rndne(float(n[13]) * RECIPD))
    assert abs(q) < (1 << 34)
    return qpass(n, -q)


def canonical(d):
    """seq_canonical on a reduced value: [0, p)"""
    nq = 1 if d[13] < 0 else 0
    out = qpass(d, nq)
    v = value(out)
    assert 0 <= v < P, "canonical form out of range"
    return out


def value(d):
    return sum(x << (28 * j) for j, x in enumerate(d))


def digits_shl8(w):
    """12 words (an integer below 2^384) -> the 14 digits of w * 2^8"""
    v = w << 8
    return [(v >> (28 * j)) & M28 for j in range(14)]


def unpack_lane(w):
    coefs = []
    for x in (w[0], w[1]):
        for i in range(4):
            c = (x >> (8 * i)) & 0xFF
            coefs.append(c - 256 if c >= 128 else c)
    idx = []
    for x in (w[2], w[3], w[4]):
        for i in range(3):
            idx.append((x >> (10 * i)) & 0x3FF)
    return coefs, idx[:8], idx[8], w[5] & 0xFF, (w[5] >> 8) & 0xFF, (w[5] >> 16) & 0xFF, bool(w[5] >> 31), w[6]


class Sim:
    def __init__(self, comp, ws, partner=None):
        """comp: gen_coop.compile_program output; ws: {workspace slot: integer (canonical 2^384-domain words)} of the wave's item;
        partner: the same for the item partner_step further on"""
        self.comp, self.ws, self.partner = comp, dict(ws), dict(partner or {})
        self.S = [[0] * 14 for _ in range(comp["n_slots"] + 1)]
        for slot, v in comp["consts"].items():
            self.S[slot] = G.T.digits_of(v)
        self.flags = [0] * 256
        self.flags[G.FLAG_TRUE] = 1
        self.result = None
        self.counts = {}

    def comb(self, coefs, idx, lo, hi, wide=False):
        v = [sum(coefs[t] * self.S[idx[t]][j] for t in range(lo, hi)) for j in range(14)]
        return v if wide else [i32(x) for x in v]

    def run(self, max_steps=None):
        rows = self.comp["rows"]
        for si, (info, r) in enumerate(self.comp["steps"]):
            if max_steps is not None and si >= max_steps:
                break
            kind, na, nb = info & 0xFF, (info >> 8) & 0xF, (info >> 12) & 0xF
            if kind == G.K_END:
                break
            self.counts[kind] = self.counts.get(kind, 0) + 1
            words = rows[r]
            writes, fwrites = [], []
            for lane in range(64):
                coefs, idx, dst, fl, fl2, fop, active, wslot = unpack_lane(words[8 * lane:8 * lane + 8])
                if not active:
                    continue
                if kind == G.K_MUL:
                    assert all(c == 0 for c in coefs[na:4]) and all(c == 0 for c in coefs[4 + nb:8])
                    writes.append((dst, mont(self.comb(coefs, idx, 0, 4), self.comb(coefs, idx, 4, 8))))
                elif kind == G.K_LIN:
                    if fop == 1:
                        v = self.comb(coefs, idx, 4, 8, True) if self.flags[fl] else self.comb(coefs, idx, 0, 4, True)
                    else:
                        assert all(c == 0 for c in coefs[na:4]) and all(c == 0 for c in coefs[4 + nb:8])
                        v = self.comb(coefs, idx, 0, 8, True)
                    writes.append((dst, reduce_digits(v)))
                elif kind == G.K_INV:
                    w = value(canonical(self.S[idx[0]]))
                    r_ = pow(w, -1, P) * pow(2, 768, P) % P if w else 0
                    writes.append((dst, reduce_digits(digits_shl8(r_))))
                elif kind == G.K_POW:
                    w = value(canonical(self.S[idx[0]]))                       # x * 2^384 mod p
                    x = w * pow(R384, -1, P) % P
                    r_ = pow(x, (P - 3) // 4, P) * R384 % P
                    writes.append((dst, reduce_digits(digits_shl8(r_))))
                elif kind == G.K_SGN:
                    c0, c1 = value(canonical(self.S[idx[0]])), value(canonical(self.S[idx[1]]))
                    fwrites.append((fl, (c0 & 1) | ((1 if c0 == 0 else 0) & (c1 & 1))))
                elif kind == G.K_ISZ:
                    fwrites.append((fl, 1 if all(x == 0 for x in self.S[idx[0]]) else 0))
                    assert (value(self.S[idx[0]]) % P == 0) == all(x == 0 for x in self.S[idx[0]])
                elif kind == G.K_FLG:
                    a, b = self.flags[fl2], self.flags[wslot] if fop != G.F_ALL else 0
                    if fop == G.F_AND: v = a & b
                    elif fop == G.F_OR: v = a | b
                    elif fop == G.F_ANDN: v = a & (1 - b)
                    elif fop == G.F_XOR: v = a ^ b
                    elif fop == G.F_ORN: v = a | (1 - b)
                    elif fop == G.F_ALL: v = 1 if all(self.flags[fl2 + t] for t in range(wslot)) else 0
                    else: raise ValueError(fop)
                    fwrites.append((fl, v))
                elif kind == G.K_LOADW:
                    src = self.partner if wslot & 0x10000 else self.ws
                    writes.append((dst, reduce_digits(digits_shl8(src[wslot & 0xFFFF]))))
                elif kind == G.K_STOREW:
                    self.ws[wslot] = value(canonical(self.S[idx[0]]))
                elif kind == G.K_RES:
                    self.result = bool(self.flags[fl]) and not self.flags[fl2]
                else:
                    raise ValueError(kind)
            for dst, v in writes:                      # a step reads all its operands before it writes
                if dst != G.SLOT_ZERO:
                    self.S[dst] = v
            for f, v in fwrites:
                self.flags[f] = v
        return self.result

    def state_value(self, machine, name):
        """the field element held in a state slot (D-form: value / 2^392 mod p)"""
        return value(self.S[machine.state[name]]) * pow(G.R392, -1, P) % P
