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
    """the imported generator module (the one the tests below simulate) emits exactly the tracked file -- into a temporary directory, never in place"""
    import io
    import contextlib
    import tempfile
    for mod, name in ((g, "mbls_fp_asm.inc"),):
        path = os.path.join(ROOT, "milagro_bls_amd", "csrc", name)
        with tempfile.TemporaryDirectory() as d:
            old = os.environ.get("MBLS_GEN_OUT_DIR")
            os.environ["MBLS_GEN_OUT_DIR"] = d
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    mod.main()
            finally:
                if old is None:
                    os.environ.pop("MBLS_GEN_OUT_DIR")
                else:
                    os.environ["MBLS_GEN_OUT_DIR"] = old
            assert open(os.path.join(d, name)).read() == open(path).read(), name + " is stale: run tools/" + mod.__name__ + ".py"


@pytest.mark.parametrize("which", ["pm3d4", "pm2"])
def test_fp_pow_routines(which):
    """sliding-window exponentiations: (p-3)/4 is the one the kernels use (square roots); p-2 exercises another bit pattern"""
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


def test_fp_inversion_by_divsteps():
    """the safegcd inversion routine (flat form of its two loops) against a^(p-2), operands at the edges included; and its pieces
    against the limb-level model of the same algorithm in plain Python (transition matrix of one round, state after the round)"""
    body = g.fp_inv_gcd_body(unrolled=True)
    rng = random.Random(43)
    R = 1 << 384
    for a in [0, 1, 2, P - 1, P - 2, (P + 1) // 2] + [rng.randrange(P) for _ in range(6)]:
        m = Machine()
        m.v[0:12] = limbs(a * R % P)
        m.run(body)
        assert from_limbs(m.v[0:12]) == pow(a, P - 2, P) * R % P, a
    # one round in isolation: f, g must stay exact multiples (u f + v g) / 2^30 and d, e congruent modulo p
    pc = g.fp_inv_gcd_pieces()
    val = lambda regs: sum((x - (1 << 32) if x >> 31 else x) << (30 * i) for i, x in enumerate(regs))
    for trial in range(4):
        x = rng.randrange(P)
        m = Machine(); m.v[0:12] = limbs(x); m.run(pc["pro"])
        for rounds in range(3):
            f0, g0, d0, e0 = val(m.v[12:25]), val(m.v[25:38]), val(m.v[38:51]), val(m.v[51:64])
            m.run(pc["head"])
            for _ in range(30):
                m.run(pc["step"])
            s32 = lambda v: v - (1 << 32) if v >> 31 else v
            u, v_, q, r = (s32(m.v[i]) for i in range(4))
            m.run(pc["tail"])
            f1, g1, d1, e1 = val(m.v[12:25]), val(m.v[25:38]), val(m.v[38:51]), val(m.v[51:64])
            assert (f1 << 30) == u * f0 + v_ * g0 and (g1 << 30) == q * f0 + r * g0
            assert (d1 * (1 << 30) - (u * d0 + v_ * e0)) % P == 0 and (e1 * (1 << 30) - (q * d0 + r * e0)) % P == 0
            assert -2 * P < d1 < P and -2 * P < e1 < P
