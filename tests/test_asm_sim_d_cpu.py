"""The digit-form ("D-form") routines of tools/gen_fpd_asm.py and tools/gen_tower_d.py interpreted on the CPU by tools/asm_sim.py and
compared with big-integer arithmetic in the 2^392 Montgomery domain: the three Fp2 product scans on signed unsaturated digits (with
redundant digit vectors at the routines' input limits), the carry / reduce / canonical passes, and the complete cyclotomic-squaring
routine (unpack from LDS, n squarings on AGPR-resident digits, canonical repack) against the formulas of fp12_cyc_sqr."""
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_fpd_asm as d          # noqa: E402
import gen_tower_d as t          # noqa: E402
from asm_sim import Machine, s32, digits_signed, from_digits_signed, limbs, from_limbs   # noqa: E402

P = d.P
RI = pow(1 << 392, -1, P)
ROUT = {k: f() for k, f in d.ROUTINE_BODIES.items()}


def mm(a, b):
    return a * b * RI % P


def f2mul(a, b):
    return ((mm(a[0], b[0]) - mm(a[1], b[1])) % P, (mm(a[0], b[1]) + mm(a[1], b[0])) % P)


def f2add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def xi(a):
    return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)


def test_product_scans_on_signed_redundant_digits():
    rng = random.Random(3)
    for trial in range(24):
        dm = [1 << 28, 1 << 29, 1 << 29][trial % 3]
        vals = [rng.randrange(-3 * P, 4 * P) for _ in range(4)]
        if trial == 0:
            vals = [0, 1, P - 1, -P]
        a0, a1, b0, b1 = vals
        m = Machine(); m.run(d.load_constants())
        for i, x in enumerate(vals):
            m.v[14 * i:14 * i + 14] = digits_signed(x, dm, rng)
        m.run(d.fp2_mul_d_body())
        c0, c1 = from_digits_signed(m.v[70:84]), from_digits_signed(m.v[84:98])
        assert (c0 - (a0 * b0 - a1 * b1) * RI) % P == 0 and (c1 - (a0 * b1 + a1 * b0) * RI) % P == 0
        assert all(0 <= s32(v) < (1 << 28) for v in m.v[70:83] + m.v[84:97])          # digits 0..12 normalised
        X = (abs(a0 * b0) + abs(a1 * b1)) // (1 << 392)
        assert -X - 1 <= c0 <= P + X + 1                                               # the bound gen_tower_d.product_bound assumes
        assert [from_digits_signed(m.v[14 * i:14 * i + 14]) for i in range(4)] == vals  # operands survive the call
        m = Machine(); m.run(d.load_constants())
        for i, x in enumerate(vals[:2]):
            m.v[14 * i:14 * i + 14] = digits_signed(x, 1 << 28, rng)
        m.run(d.fp2_sqr_d_body())
        c0, c1 = from_digits_signed(m.v[70:84]), from_digits_signed(m.v[84:98])
        assert (c0 - (a0 * a0 - a1 * a1) * RI) % P == 0 and (c1 - 2 * a0 * a1 * RI) % P == 0
        m = Machine(); m.run(d.load_constants())
        for i, x in enumerate(vals[:3]):
            m.v[14 * i:14 * i + 14] = digits_signed(x, dm, rng)
        m.run(d.fp2_mulfp_d_body())
        c0, c1 = from_digits_signed(m.v[70:84]), from_digits_signed(m.v[84:98])
        assert (c0 - a0 * b0 * RI) % P == 0 and (c1 - a1 * b0 * RI) % P == 0


def test_column_limit_is_what_the_scans_need():
    # operands at the generator's own limit must not overflow a column (the simulator asserts on 64-bit overflow)
    lim = 1 << 29
    assert d.column_ok([(lim, lim), (lim, lim)]) and not d.column_ok([(2 * lim, lim), (2 * lim, lim)])
    worst = [(lim - 1) & 0xFFFFFFFF] * 14
    m = Machine(); m.run(d.load_constants())
    for i in range(4):
        m.v[14 * i:14 * i + 14] = worst
    m.run(d.fp2_mul_d_body())


def test_carry_reduce_and_canonical_passes():
    rng = random.Random(4)
    reg = lambda j: "v%d" % (112 + j)
    for trial in range(40):
        x = rng.randrange(-200 * P, 200 * P)
        m = Machine(); m.run(t.shell_constants())
        m.v[112:126] = digits_signed(x, 1 << 30, rng)
        m.run(t.seq_norm(reg))
        assert from_digits_signed(m.v[112:126]) == x and all(0 <= s32(v) < (1 << 28) for v in m.v[112:125])
        m.v[112:126] = digits_signed(x, 1 << 30, rng)
        m.run(t.seq_reduce(reg))
        y = from_digits_signed(m.v[112:126])
        assert (y - x) % P == 0 and t.REDUCED.vlo <= y <= t.REDUCED.vhi and all(0 <= s32(v) < (1 << 28) for v in m.v[112:125])
        m.run(t.seq_canonical(reg) + t.seq_to32(reg))
        assert from_limbs(m.v[112:124]) == x % P
    for x in (0, P - 1, -1, -(P // 2), P // 2):                       # canonical pass at the edges of its domain
        m = Machine(); m.run(t.shell_constants())
        dd = [(x >> (28 * i)) & 0xFFFFFFF for i in range(13)]
        m.v[112:126] = dd + [((x - sum(v << (28 * i) for i, v in enumerate(dd))) >> 364) & 0xFFFFFFFF]
        m.run(t.seq_canonical(reg) + t.seq_to32(reg))
        assert from_limbs(m.v[112:124]) == x % P


def cyc_model(z):
    def fp4(a, b):
        t0 = f2mul(a, a); t1 = f2mul(b, b); s = f2add(a, b)
        return f2add(xi(t1), t0), f2sub(f2sub(f2mul(s, s), t0), t1)

    def dbl(a):
        return f2add(a, a)
    z0, z4, z3, z2, z1, z5 = z
    t0, t1 = fp4(z0, z1)
    n0 = f2add(dbl(f2sub(t0, z0)), t0); n1 = f2add(dbl(f2add(t1, z1)), t1)
    t0, t1 = fp4(z2, z3); t2, t3 = fp4(z4, z5)
    n4 = f2add(dbl(f2sub(t0, z4)), t0); n5 = f2add(dbl(f2add(t1, z5)), t1); x = xi(t3)
    n2 = f2add(dbl(f2add(x, z2)), x); n3 = f2add(dbl(f2sub(t2, z3)), t2)
    return [n0, n4, n3, n2, n1, n5]


def test_cyclotomic_squaring_routine():
    full, body, stats, pro, epi = t.cyc_sqr_d_routine()
    assert not any("scratch" in l or "buffer_" in l for l in full)
    rng = random.Random(9)
    for trial in range(4):
        z = [(rng.randrange(P), rng.randrange(P)) for _ in range(6)]
        if trial == 0:
            z = [(0, P - 1)] * 3 + [(P - 1, 0)] * 3
        m = Machine(ROUT); m.v[252] = 8192
        for e in range(6):
            for i in range(2):
                for j, w in enumerate(limbs(z[e][i])):
                    m.lds[8192 + ((2 * e + i) * 12 + j) * 256] = w
        m.run(t.shell_constants()); m.run(pro[1:])
        exp = z
        for r in range([1, 2, 3, 5][trial]):
            m.run(body); exp = cyc_model(exp)
            for e in range(12):                                      # loop invariant: reduced, normalised
                v = from_digits_signed(m.a[14 * e:14 * e + 14])
                assert t.REDUCED.vlo <= v <= t.REDUCED.vhi and all(0 <= s32(w) < (1 << 28) for w in m.a[14 * e:14 * e + 13])
        m.run(epi)
        for e in range(6):
            got = tuple(from_limbs([m.lds[8192 + ((2 * e + i) * 12 + j) * 256] for j in range(12)]) for i in range(2))
            assert got == exp[e], (trial, e)


def test_generated_d_files_up_to_date():
    for script, name in (("gen_fpd_asm.py", "mbls_fpd_asm.inc"), ("gen_tower_d.py", "mbls_towerd_asm.inc")):
        inc = os.path.join(ROOT, "milagro_bls_amd", "csrc", name)
        before = open(inc).read()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", script)], stdout=subprocess.DEVNULL)
        assert open(inc).read() == before, name
