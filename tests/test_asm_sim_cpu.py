"""The generated gfx950 routines on 12 x 32-bit limbs (tools/gen_fp_asm.py) interpreted on the CPU by tools/asm_sim.py and compared
with big-integer arithmetic: Montgomery products, the fused Fp2 routines and the fixed-exponent exponentiations. The digit-form
routines have their own file, test_asm_sim_d_cpu.py."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_fp_asm as g          # noqa: E402
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
    for mod, name in ((g, "mbls_fp_asm.inc"),):
        path = os.path.join(ROOT, "milagro_bls_amd", "csrc", name)
        before = open(path).read()
        with contextlib.redirect_stdout(io.StringIO()):
            mod.main()
        assert open(path).read() == before, name + " is stale: run tools/" + mod.__name__ + ".py"


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
