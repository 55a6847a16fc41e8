"""The generated gfx950 routines on 12 x 32-bit limbs (tools/gen_fp_asm.py, tools/gen_tower_asm.py) interpreted on the CPU by
tools/asm_sim.py and compared with big-integer arithmetic: Montgomery products, the fused Fp2 routines, the fixed-exponent
exponentiations and the Fp12 multiplication routine. The digit-form routines have their own file, test_asm_sim_d_cpu.py."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_fp_asm as g          # noqa: E402
import gen_tower_asm as t       # noqa: E402
from asm_sim import Machine, limbs, from_limbs   # noqa: E402

P = g.P
RI = pow(1 << 384, -1, P)
ROUTINES = {"mbls_fp2_mul_asm_fn": g.fp2_mul_body(), "mbls_fp2_sqr_asm_fn": g.fp2_sqr_body(), "mbls_fp2_mulfp_asm_fn": g.fp2_mulfp_body()}


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


def edge_values(rng, n):
    vals = [0, 1, P - 1, P - 2, (1 << 383) % P]
    return [vals[i] if i < len(vals) else rng.randrange(P) for i in range(n)]


def test_fp_routines():
    rng = random.Random(1)
    for trial in range(12):
        a0, a1, b0, b1 = (edge_values(rng, 4) if trial == 0 else [rng.randrange(P) for _ in range(4)])
        if trial == 1:
            a0 = a1 = b0 = b1 = P - 1
        if trial == 2:
            b1 = 0
        if trial == 3:                            # first factors may arrive unreduced (< 2p)
            a0, a1 = 2 * P - 1, 2 * P - 2
        m = Machine()
        for i, x in enumerate([a0, a1, b0, b1]):
            m.v[12 * i:12 * i + 12] = limbs(x)
        m.run(g.fp2_mul_body())
        assert (from_limbs(m.v[48:60]), from_limbs(m.v[60:72])) == f2mul((a0, a1), (b0, b1))
        a0, a1 = a0 % P, a1 % P                   # the other routines take reduced operands
        m = Machine(); m.v[0:12] = limbs(a0); m.v[12:24] = limbs(a1)
        m.run(g.fp2_sqr_body())
        assert (from_limbs(m.v[24:36]), from_limbs(m.v[36:48])) == f2mul((a0, a1), (a0, a1))
        m = Machine(); m.v[0:12] = limbs(a0); m.v[12:24] = limbs(a1); m.v[24:36] = limbs(b0)
        m.run(g.fp2_mulfp_body())
        assert (from_limbs(m.v[36:48]), from_limbs(m.v[48:60])) == (mm(a0, b0), mm(a1, b0))
        m = Machine()
        for i, x in enumerate([a0, b0, a1, b1]):
            m.v[12 * i:12 * i + 12] = limbs(x)
        m.run(g.fp_mulpair_body())
        assert (from_limbs(m.v[48:60]), from_limbs(m.v[60:72])) == (mm(a0, b0), mm(a1, b1))
        m = Machine(); m.v[0:12] = limbs(a0); m.v[12:24] = limbs(b0)
        m.run(g.fp_mul_body())
        assert from_limbs(m.v[0:12]) == mm(a0, b0)


def lds_put(m, base, slot, x):
    for j, w in enumerate(limbs(x)):
        m.lds[base + (slot * 12 + j) * 256] = w


def lds_get(m, base, slot):
    return from_limbs([m.lds[base + (slot * 12 + j) * 256] for j in range(12)])


def test_generated_files_up_to_date():
    import io
    import contextlib
    for mod, name in ((g, "mbls_fp_asm.inc"), (t, "mbls_tower_asm.inc")):
        path = os.path.join(ROOT, "milagro_bls_amd", "csrc", name)
        before = open(path).read()
        with contextlib.redirect_stdout(io.StringIO()):
            mod.main()
        assert open(path).read() == before, name + " is stale: run tools/" + mod.__name__ + ".py"


# ---- big-integer mirror of one doubling iteration (Montgomery-domain values, formulas of mbls_pairing.h / mbls_tower.h)
def f2mulfp(a, s):
    return (mm(a[0], s), mm(a[1], s))


def f2k(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f6add(a, b):
    return [f2add(a[i], b[i]) for i in range(3)]


def f6sub(a, b):
    return [f2sub(a[i], b[i]) for i in range(3)]


def f6mulv(a):
    return [xi(a[2]), a[0], a[1]]


def f6mul(a, b):
    t0, t1, t2 = f2mul(a[0], b[0]), f2mul(a[1], b[1]), f2mul(a[2], b[2])
    c0 = f2add(xi(f2sub(f2sub(f2mul(f2add(a[1], a[2]), f2add(b[1], b[2])), t1), t2)), t0)
    c1 = f2add(f2sub(f2sub(f2mul(f2add(a[0], a[1]), f2add(b[0], b[1])), t0), t1), xi(t2))
    c2 = f2add(f2sub(f2sub(f2mul(f2add(a[0], a[2]), f2add(b[0], b[2])), t0), t2), t1)
    return [c0, c1, c2]


def test_fp12_mul_routine():
    lines, stats = t.build("fp12_mul")
    assert not any("scratch" in l for l in lines)
    rng = random.Random(21)
    r2 = lambda: (rng.randrange(P), rng.randrange(P))
    for trial in range(3):
        a = ([r2(), r2(), r2()], [r2(), r2(), r2()])
        b = ([r2(), r2(), r2()], [r2(), r2(), r2()])
        if trial == 0:
            b = ([((1 << 384) % P, 0), (0, 0), (0, 0)], [(0, 0), (0, 0), (0, 0)])
        m = Machine(ROUTINES); m.v[252] = 512
        for i, x in enumerate([x for h in a for c in h for x in c]):
            lds_put(m, 512, i, x)
        for i, x in enumerate([x for h in b for c in h for x in c]):
            m.v[t.F12_ARG[i]:t.F12_ARG[i] + 12] = limbs(x)
        m.run(lines)
        t0, t1 = f6mul(a[0], b[0]), f6mul(a[1], b[1])
        c1 = f6sub(f6sub(f6mul(f6add(a[0], a[1]), f6add(b[0], b[1])), t0), t1)
        c0 = f6add(t0, f6mulv(t1))
        exp = [x for h in (c0, c1) for c in h for x in c]
        assert [lds_get(m, 512, i) for i in range(12)] == exp
        if trial == 0:
            assert exp == [x for h in a for c in h for x in c]


@pytest.mark.parametrize("which", ["pm3d4", "pm2"])
def test_fp_pow_routines(which):
    e = g.EXP_PM3D4 if which == "pm3d4" else g.EXP_PM2
    body = g.pow_body(e)
    subs = g.pow_subroutines()
    rng = random.Random(41)
    R = 1 << 384
    for a in (0, 1, P - 1, rng.randrange(P), rng.randrange(P)):
        am = a * R % P
        m = Machine(subs); m.s[("pair", 30)] = 5
        m.v[0:12] = limbs(am)
        m.run(body)
        assert from_limbs(m.v[0:12]) == pow(a, e, P) * R % P, (which, a)
