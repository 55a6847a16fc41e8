"""The generated gfx950 routines (tools/gen_fp_asm.py, tools/gen_tower_asm.py) interpreted on the CPU by tools/asm_sim.py and
compared with big-integer arithmetic: Montgomery products, the fused Fp2 routines, and the straight-line tower routines."""
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


def test_cyc_sqr_routine():
    lines, stats = t.build("cyc_sqr")
    assert not any("scratch" in l for l in lines)
    rng = random.Random(5)

    def fp4(a, b):
        t0 = f2mul(a, a); t1 = f2mul(b, b)
        s = f2add(a, b)
        return f2add(xi(t1), t0), f2sub(f2sub(f2mul(s, s), t0), t1)

    def dbl(a):
        return f2add(a, a)
    for trial in range(4):
        z = [(rng.randrange(P), rng.randrange(P)) for _ in range(6)]
        if trial == 0:
            z = [(0, 0)] * 5 + [(P - 1, 1)]
        m = Machine(ROUTINES); m.v[252] = 4096; m.s[("pair", 30)] = 7
        for e in range(6):
            lds_put(m, 4096, 2 * e, z[e][0]); lds_put(m, 4096, 2 * e + 1, z[e][1])
        m.run(lines)
        z0, z4, z3, z2, z1, z5 = z
        t0, t1 = fp4(z0, z1)
        n0 = f2add(dbl(f2sub(t0, z0)), t0); n1 = f2add(dbl(f2add(t1, z1)), t1)
        t0, t1 = fp4(z2, z3); t2, t3 = fp4(z4, z5)
        n4 = f2add(dbl(f2sub(t0, z4)), t0); n5 = f2add(dbl(f2add(t1, z5)), t1); x = xi(t3)
        n2 = f2add(dbl(f2add(x, z2)), x); n3 = f2add(dbl(f2sub(t2, z3)), t2)
        exp = [n0, n4, n3, n2, n1, n5]
        for e in range(6):
            assert (lds_get(m, 4096, 2 * e), lds_get(m, 4096, 2 * e + 1)) == exp[e], (trial, e)


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


def f12sqr(f):
    a, b = f
    ab = f6mul(a, b)
    st = f6sub(f6mul(f6add(a, b), f6add(a, f6mulv(b))), ab)
    return (f6sub(st, f6mulv(ab)), f6add(ab, ab))


def f12mul_line(f, c0, c2, c3):
    a, b = f
    zero = (0, 0)
    l0, l1 = [c0, c2, zero], [zero, c3, zero]
    t0, t1 = f6mul(a, l0), f6mul(b, l1)
    c1 = f6sub(f6sub(f6mul(f6add(a, b), f6add(l0, l1)), t0), t1)
    return (f6add(t0, f6mulv(t1)), c1)


def dbl_step_model(f, T, npx, py, pz3, skip):
    Tx, Ty, Tz = T
    B = f2mul(Ty, Ty); C = f2mul(Tz, Tz)
    E = f2k(xi(C), 12); F = f2k(E, 3)
    X2 = f2mul(Tx, Tx); YZ = f2mul(Ty, Tz)
    c0 = f2sub(B, E)
    if pz3 is not None:
        c0 = f2mulfp(c0, pz3)
    c2 = f2mulfp(f2k(X2, 3), npx)
    c3 = f2mulfp(f2k(YZ, 2), py)
    x3 = f2k(f2mul(f2mul(Tx, Ty), f2sub(B, F)), 2)
    BF = f2add(B, F)
    y3 = f2sub(f2mul(BF, BF), f2k(f2mul(E, E), 12))
    z3 = f2k(f2mul(B, YZ), 8)
    if skip:
        c0, c2, c3 = (t.ONE_M, 0), (0, 0), (0, 0)
    return f12mul_line(f, c0, c2, c3), (x3, y3, z3)


@pytest.mark.parametrize("skips", [(0, 0), (1, 0), (0, 1)])
def test_miller_dbl_routine(skips):
    lines, stats = t.build("miller_dbl")
    assert not any("scratch" in l for l in lines)
    rng = random.Random(11 + skips[0] + 2 * skips[1])
    r2 = lambda: (rng.randrange(P), rng.randrange(P))
    f = ([r2(), r2(), r2()], [r2(), r2(), r2()])
    T = [(r2(), r2(), r2()), (r2(), r2(), r2())]
    p1 = [rng.randrange(P) for _ in range(3)]           # -px, py, pz^3
    m = Machine(ROUTINES); m.v[252] = 8192
    m.s[("pair", 48)], m.s[("pair", 54)] = skips
    flat = [x for h in f for c in h for x in c]
    for i, x in enumerate(flat):
        m.a[12 * i:12 * i + 12] = limbs(x)
    for i, x in enumerate(p1):
        m.a[144 + 12 * i:144 + 12 * i + 12] = limbs(x)
    for k in range(2):
        for e in range(3):
            lds_put(m, 8192, 6 * k + 2 * e, T[k][e][0]); lds_put(m, 8192, 6 * k + 2 * e + 1, T[k][e][1])
    m.run(lines)
    g1 = f12sqr(f)
    g1, T0n = dbl_step_model(g1, T[0], t.NPX0_M, t.PY0_M, None, skips[0])
    g1, T1n = dbl_step_model(g1, T[1], p1[0], p1[1], p1[2], skips[1])
    exp = [x for h in g1 for c in h for x in c]
    got = [from_limbs(m.a[12 * i:12 * i + 12]) for i in range(12)]
    assert got == exp
    for k, Tn in enumerate((T0n, T1n)):
        for e in range(3):
            assert (lds_get(m, 8192, 6 * k + 2 * e), lds_get(m, 8192, 6 * k + 2 * e + 1)) == Tn[e], (k, e)
    assert [from_limbs(m.a[144 + 12 * i:144 + 12 * i + 12]) for i in range(3)] == p1      # the G1 argument stays in its home


def test_fp12_mul_routine():
    lines, stats = t.build("fp12_mul")
    assert not any("scratch" in l for l in lines)
    rng = random.Random(21)
    r2 = lambda: (rng.randrange(P), rng.randrange(P))
    for trial in range(3):
        a = ([r2(), r2(), r2()], [r2(), r2(), r2()])
        b = ([r2(), r2(), r2()], [r2(), r2(), r2()])
        if trial == 0:
            b = ([(t.ONE_M, 0), (0, 0), (0, 0)], [(0, 0), (0, 0), (0, 0)])
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


def test_g2_dbl_routine():
    lines, stats = t.build("g2_dbl")
    rng = random.Random(31)
    r2 = lambda: (rng.randrange(P), rng.randrange(P))
    for trial in range(3):
        X, Y, Z = r2(), r2(), r2()
        if trial == 0:
            Z = (0, 0)
        m = Machine(ROUTINES)
        for i, x in enumerate([X[0], X[1], Y[0], Y[1], Z[0], Z[1]]):
            m.a[12 * i:12 * i + 12] = limbs(x)
        m.run(lines)
        sq = lambda a: f2mul(a, a)
        A, B = sq(X), sq(Y)
        C = sq(B)
        D = f2k(f2sub(f2sub(sq(f2add(X, B)), A), C), 2)
        E = f2k(A, 3); F = sq(E)
        Z3 = f2k(f2mul(Y, Z), 2)
        X3 = f2sub(F, f2k(D, 2))
        Y3 = f2sub(f2mul(E, f2sub(D, X3)), f2k(C, 8))
        got = [from_limbs(m.a[12 * i:12 * i + 12]) for i in range(6)]
        assert got == [X3[0], X3[1], Y3[0], Y3[1], Z3[0], Z3[1]]


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
