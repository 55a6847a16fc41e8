"""The digit-form ("D-form") routines of tools/gen_fpd_asm.py and tools/gen_tower_d.py interpreted on the CPU by tools/asm_sim.py and
compared with big-integer arithmetic in the 2^392 Montgomery domain: the three Fp2 product scans on signed unsaturated digits (with
redundant digit vectors at the routines' input limits), the carry / reduce / canonical passes, the cyclotomic squaring body against
the formulas of fp12_cyc_sqr, and the Miller-loop, exponentiation and G2-doubling routines against the same programs run on field values."""
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "pymodel"))
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
    """the squaring body of the exponentiation routine (state in AGPRs, reduced and normalised between iterations) against the formulas
    of fp12_cyc_sqr written out above -- independent of the program recorder the other tests share with the generator"""
    body, stats = t.build_cyc_sqr_d()
    assert not any("scratch" in l or "buffer_" in l for l in body)
    rng = random.Random(9)
    for trial in range(4):
        z = [(rng.randrange(P), rng.randrange(P)) for _ in range(6)]
        if trial == 0:
            z = [(0, P - 1)] * 3 + [(P - 1, 0)] * 3
        m = miller_machine(0); m.run(t.shell_constants())
        for e in range(6):
            for i in range(2):
                rep = z[e][i] + rng.choice([-15, -1, 0, 14]) * P
                m.a[14 * (2 * e + i):14 * (2 * e + i) + 14] = normalised_digits(rep)
        exp = z
        for r in range([1, 2, 3, 5][trial]):
            m.run(body); exp = cyc_model(exp)
            for e in range(12):                                      # loop invariant: reduced, normalised
                v = from_digits_signed(m.a[14 * e:14 * e + 14])
                assert t.REDUCED.vlo <= v <= t.REDUCED.vhi and all(0 <= s32(w) < (1 << 28) for w in m.a[14 * e:14 * e + 13])
                assert v % P == exp[e // 2][e % 2], (trial, r, e)


def test_generated_d_files_up_to_date():
    """the generators' output, written to a temporary directory (never over the tracked files), equals what the build compiles"""
    import filecmp
    import tempfile
    for script, name in (("gen_fpd_asm.py", "mbls_fpd_asm.inc"), ("gen_tower_d.py", "mbls_towerd_asm.inc")):
        inc = os.path.join(ROOT, "milagro_bls_amd", "csrc", name)
        with tempfile.TemporaryDirectory() as d:
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", script)], stdout=subprocess.DEVNULL, env=dict(os.environ, MBLS_GEN_OUT_DIR=d))
            assert filecmp.cmp(os.path.join(d, name), inc, shallow=False), name + " is stale: run tools/" + script


# ---------------------------------------------------------------------------------------------- the Miller loop and the G2 doubling runs
# The programs of tools/gen_tower_d.py are executed twice: by the allocator (-> instructions, interpreted by asm_sim) and by ModelProg,
# which implements the same Prog interface on true field values. Agreement checks the allocator, the bound tracking (no 32-bit /
# 64-bit overflow is possible: the simulator asserts), the packing / HBM fetch sequences and the shells.
from gen_tower_d import Prog     # noqa: E402  (the program recorder)

R392, R384 = 1 << 392, 1 << 384
RI392 = pow(R392, -1, P)


class _Ops(list):
    def __init__(self, owner):
        super().__init__(); self.o = owner

    def append(self, op):
        kind, outs, ins, aux = op
        if kind in ("norm", "reduce"):
            self.o.val[outs[0]] = self.o.val[ins[0]]
        elif kind == "scale":
            self.o.val[outs[0]] = self.o.val[ins[0]] * aux % P
        elif kind == "shadd":
            self.o.val[outs[0]] = ((self.o.val[ins[0]] << aux) + self.o.val[ins[1]]) % P
        elif kind == "storep":                               # aux: a slot number, or ('k', j) = slot j of the run-time record
            self.o.out_g[aux] = self.o.val[ins[0]]
        elif kind == "iszero":
            self.o.masks[aux] = 1 if self.o.val[ins[0]] % P == 0 else 0
        elif kind == "mask_xor":
            self.o.masks[aux[0]] = self.o.masks[aux[1]] ^ self.o.masks[aux[2]]
        elif kind == "sgn0":                                 # the operands are plain integers: representation = value * 2^392
            x0, x1 = (self.o.val[v] * R392 % P for v in ins)
            self.o.masks[aux[0]] = (x0 & 1) | ((1 if x0 == 0 else 0) & (x1 & 1))
        elif kind == "pow34":                                # like inv: the operand is x 2^-8
            self.o.val[outs[0]] = pow(self.o.val[ins[0]] * 256, (P - 3) // 4, P)
        elif kind == "mask_and":
            self.o.masks[aux[0]] = self.o.masks[aux[1]] & self.o.masks[aux[2]]
        elif kind == "mask_orn2":
            self.o.masks[aux[0]] = self.o.masks[aux[1]] | (1 - self.o.masks[aux[2]])
        elif kind == "neg":
            self.o.val[outs[0]] = -self.o.val[ins[0]] % P
        elif kind == "inv":                                  # the operand is x 2^-8 (see Prog.inv); 0 -> 0 like a^(p-2)
            self.o.val[outs[0]] = pow(self.o.val[ins[0]] * 256, P - 2, P)
        else:
            raise ValueError(kind)


class ModelProg(Prog):
    def __init__(self, init, masks):
        self.val, self.nval, self.init, self.masks, self.out_g, self.out_home, self.init_loc = {}, 0, init, masks, {}, {}, {}
        self.ops = _Ops(self)

    def new(self):
        self.nval += 1; return self.nval - 1

    def live_in(self, loc):
        v = self.new(); self.val[v] = self.init[loc] % P; self.init_loc[v] = loc; return v

    def add(self, a, b):
        d_ = self.new(); self.val[d_] = (self.val[a] + self.val[b]) % P; return d_

    def sub(self, a, b):
        d_ = self.new(); self.val[d_] = (self.val[a] - self.val[b]) % P; return d_

    def const(self, c):
        d_ = self.new(); self.val[d_] = c * RI392 % P; return d_

    def sel(self, mask, a, b):
        d_ = self.new(); self.val[d_] = self.val[b] if self.masks[mask] else self.val[a]; return d_

    def pair(self, k0, a0, b0, k1, a1, b1):
        f = {"add": self.add, "sub": self.sub}
        return (f[k0](a0, b0), f[k1](a1, b1))

    def call(self, kind, ins):
        x = [self.val[i] for i in ins]; o = (self.new(), self.new())
        if kind == "mul1":
            self.val[o[0]] = x[0] * x[1] % P
            return (o[0],)
        if kind == "sqrpair":
            r = (x[0] * x[0] % P, x[1] * x[1] % P)
        elif kind == "mulpair":
            r = (x[0] * x[2] % P, x[1] * x[3] % P)
        elif kind == "mul":
            r = ((x[0] * x[2] - x[1] * x[3]) % P, (x[0] * x[3] + x[1] * x[2]) % P)
        elif kind == "sqr":
            r = ((x[0] * x[0] - x[1] * x[1]) % P, (2 * x[0] * x[1]) % P)
        else:
            r = (x[0] * x[2] % P, x[1] * x[2] % P)
        self.val[o[0]], self.val[o[1]] = r
        return o

    def store(self, a, loc):
        self.out_home[loc] = self.val[a]

    def keep(self, vals):
        pass


def run_model(progf, init, masks):
    mp = ModelProg(init, masks)
    saved = t.Prog
    t.Prog = lambda: mp
    try:
        progf()
    finally:
        t.Prog = saved
    return mp


STRIDE, GBASE, LADDR = 4096, 0x7F0000001000, 8192


def ws_addr(slot, j):
    return GBASE + ((slot * 12 + j) * STRIDE) * 4 + LADDR      # the caller folds (item offset - LADDR) into the base


def ws_put(m, slot, x):
    for j, w in enumerate(limbs(x)):
        m.mem[ws_addr(slot, j)] = w


def ws_get(m, slot):
    return from_limbs([m.mem[ws_addr(slot, j)] for j in range(12)])


def normalised_digits(x):
    dd = [(x >> (28 * i)) & 0xFFFFFFF for i in range(13)]
    return [v & 0xFFFFFFFF for v in dd + [(x - sum(v << (28 * i) for i, v in enumerate(dd))) >> 364]]


def miller_machine(masks_bits):
    m = Machine(ROUT); m.v[252] = LADDR; m.v[253] = masks_bits
    m.s[68] = GBASE & 0xFFFFFFFF; m.s[69] = GBASE >> 32; m.s[70] = STRIDE * 4
    return m


@pytest.mark.parametrize("which", ["dbl", "first", "add01", 0, 1])
def test_miller_bodies(which):
    """one doubling iteration / the first iteration (f = 1) / the addition steps (both pairs in one body, one pair each) from random state at the
    edges of the declared bounds, with and without skipped pairs"""
    rng = random.Random(11)
    body, _ = t.build_miller(which)
    progf = lambda: t.prog_miller(which)
    for trial in range(3):
        masks = {"s[48:49]": 1 if trial == 1 else 0, "s[54:55]": 1 if trial == 2 else 0}
        m = miller_machine(0); m.run(t.shell_constants())
        m.s[("pair", 48)] = masks["s[48:49]"]; m.s[("pair", 54)] = masks["s[54:55]"]
        init = {}
        for i in range(12):
            x = rng.randrange(P); rep = (x * R392 % P) + rng.choice([-15, -1, 0, 14]) * P
            init[("a", i)] = x; m.a[14 * i:14 * i + 14] = normalised_digits(rep)
        for sl in range(13):
            x = rng.randrange(P); init[("g", sl)] = x; ws_put(m, sl, x * R384 % P)
        for sl in range(31, 43):
            x = rng.randrange(P); rep = x * R392 % P
            init[("gd", sl)] = x; ws_put(m, sl, rep + P if rep < P // 2 else rep)
        m.run(body)
        mp = run_model(progf, init, masks)
        for loc, x in mp.out_home.items():
            got = from_digits_signed(m.a[14 * loc[1]:14 * loc[1] + 14])
            assert (got - x * R392) % P == 0 and t.F_IN.vlo <= got <= t.F_IN.vhi, (which, trial, loc)
            assert all(0 <= s32(w) < (1 << 28) for w in m.a[14 * loc[1]:14 * loc[1] + 13])
        for slot, x in mp.out_g.items():
            got = ws_get(m, slot)
            assert (got - x * R392) % P == 0 and t.PACKED.vlo <= got <= t.PACKED.vhi, (which, trial, slot)


def pair_machines(masks_bits=0, routines=None):
    """the two lanes of an item in pair mode (tools/gen_tower_d.py, pair_products): same LDS column and workspace item, lane numbers 2 j and
    2 j + 1; PAIR_EXEC = both lanes, PAIR_ROLE = the odd one; the exec writes of the masked moves are modelled"""
    ma, mb = miller_machine(masks_bits), miller_machine(masks_bits)
    mb.lds = ma.lds; mb.mem = ma.mem
    for m, lane in ((ma, 4), (mb, 5)):
        if routines is not None:
            m.routines = routines
        m.lane = lane; m.model_exec = True
        m.s[("pair", 82)] = 1; m.s[("pair", 94)] = lane & 1
    return ma, mb


@pytest.mark.parametrize("which", ["dbl", "first", 1])
def test_miller_bodies_in_pair_mode(which):
    """the one-pair Miller bodies with their products in pairs on two lanes (masked operand moves, DPP exchange, masked swap): both lanes must
    end with the model's values in their AGPRs, and the pair's workspace slots with the model's running point"""
    from asm_sim import run_pair
    rng = random.Random(19)
    body, st = t.build_miller(which, (1,), pair_mode=True)
    assert st.get("pairs", 0) >= (4 if which == "first" else 14)
    for trial in range(2):
        masks = {"s[48:49]": 0, "s[54:55]": 1 if trial == 1 else 0}
        ma, mb = pair_machines()
        init = {}
        for m in (ma, mb):
            m.run(t.shell_constants())
            m.s[("pair", 48)] = 0; m.s[("pair", 54)] = masks["s[54:55]"]
        rng2 = random.Random(100 + trial)
        for i in range(12):
            x = rng2.randrange(P); rep = (x * R392 % P) + rng2.choice([-15, -1, 0, 14]) * P
            init[("a", i)] = x
            for m in (ma, mb):
                m.a[14 * i:14 * i + 14] = normalised_digits(rep)
        for sl in range(13):
            x = rng2.randrange(P); init[("g", sl)] = x; ws_put(ma, sl, x * R384 % P)
        for sl in range(31, 43):
            x = rng2.randrange(P); rep = x * R392 % P
            init[("gd", sl)] = x; ws_put(ma, sl, rep + P if rep < P // 2 else rep)
        run_pair(ma, mb, body)
        mp = run_model(lambda: t.prog_miller(which, (1,)), init, masks)
        for m in (ma, mb):
            for loc, x in mp.out_home.items():
                got = from_digits_signed(m.a[14 * loc[1]:14 * loc[1] + 14])
                assert (got - x * R392) % P == 0 and t.F_IN.vlo <= got <= t.F_IN.vhi, (which, trial, loc)
        assert ma.a[:168] == mb.a[:168]
        for slot, x in mp.out_g.items():
            got = ws_get(ma, slot)
            assert (got - x * R392) % P == 0 and t.PACKED.vlo <= got <= t.PACKED.vhi, (which, trial, slot)


def miller_loop_sim(runs, seed, masks_bits=0, pairs=(0, 1), q0=None, pair_mode=False):
    """q0: an affine G2 point ((x0, x1), (y0, y1)) for pair 0's Q (workspace slots 3..6) instead of random field elements.
    pair_mode: the routine for lane pairs, on two machines in lockstep (both must come back with the model's value)"""
    rng = random.Random(seed)
    full, pieces, st = t.miller_loop_d_routine(pairs, pair_mode) if pair_mode else t.miller_loop_d_routine(pairs)
    if pair_mode:
        from asm_sim import run_pair
        m, m_b = pair_machines(masks_bits)

        class _Both:                                    # run(lines) on the pair; everything else is lane A's (shared memory, same values)
            def __getattr__(self, name):
                return getattr(m_a, name)

            def run(self, lines):
                run_pair(m_a, m_b, lines)
        m_a = m; m = _Both()
    else:
        m = miller_machine(masks_bits)
    true = {}
    for sl in range(13):
        x = rng.randrange(P)
        if q0 is not None and 3 <= sl <= 6:
            x = q0[(sl - 3) >> 1][(sl - 3) & 1]
        true[sl] = x; ws_put(m, sl, x * R384 % P)
    m.run(pieces["pro"])
    masks = {"s[48:49]": masks_bits & 1, "s[54:55]": (masks_bits >> 1) & 1}
    f = [1] + [0] * 11
    T = {}
    for k in pairs:
        for e in range(3):
            for i in range(2):
                sl = t.Q_SLOT[k][e]
                T[t.T_SLOT(k, e, i)] = (1 if i == 0 else 0) if sl is None else true[sl[i]]

    def step(progf):
        init = {("a", i): f[i] for i in range(12)}
        init.update({("g", sl): true[sl] for sl in range(13)})
        init.update({("gd", sl): T[sl] for sl in T})
        mp = run_model(progf, init, masks)
        for loc, x in mp.out_home.items():
            f[loc[1]] = x
        T.update(mp.out_g)
    # the routine's control flow: the first iteration (f = 1: the lines are f), then per phase the addition step(s) and the next run of doublings
    assert runs[0] == 1
    add_name = "add01" if len(pairs) == 2 else "add1"
    assert set(pieces) == {"pro", "first", "dbl", add_name, "epi"}
    for ph, n in enumerate(runs):
        for it in range(n):
            if ph == 0:
                m.run(pieces["first"]); step(lambda: t.prog_miller("first", pairs))
            else:
                m.run(pieces["dbl"]); step(lambda: t.prog_miller("dbl", pairs))
        if ph < len(runs) - 1:
            m.run(pieces[add_name]); step(lambda: t.prog_miller(add_name if len(pairs) == 2 else 1, pairs))
    m.run(pieces["epi"][:-1])
    for i in range(12):
        assert from_limbs(m.v[t.F_OUT[i]:t.F_OUT[i] + 12]) == f[i] * R384 % P, ("f", i)
        if pair_mode:
            assert m_b.v[t.F_OUT[i]:t.F_OUT[i] + 12] == m_a.v[t.F_OUT[i]:t.F_OUT[i] + 12]
    assert not any("scratch" in l or "buffer_" in l for l in full)
    # the running points as the routine leaves them in the workspace: packed words of the 2^392 domain, representatives in (0.5 p, 1.5 p)
    # -- what lane_sig_verdict (mbls_lanes.h) reads for pair 0 through MBLS_GEN_MILLER_T0_SLOT
    for sl, x in T.items():
        got = ws_get(m, sl)
        assert (got - x * R392) % P == 0 and t.PACKED.vlo <= got <= t.PACKED.vhi, ("T", sl)
    return T


def test_miller_loop_routine_short_schedules():
    """prologue (f = 1, T = Q into the packed slots), doubling runs, addition steps, epilogue (back to canonical 2^384-domain words);
    the schedule of the real loop is the same sequence with runs of 1, 2, 3, 9, 32 and 16 iterations"""
    assert t.RUNS == [1, 2, 3, 9, 32, 16] and sum(t.RUNS) == 63
    miller_loop_sim([1, 2, 1], 5)
    miller_loop_sim([1, 1], 6, masks_bits=1)          # pair 0 contributes 1 (infinite signature)
    miller_loop_sim([1, 1], 7, masks_bits=2)
    miller_loop_sim([1, 1, 1], 8, pairs=(1,))           # the single-pair routine of the n-pairing paths
    miller_loop_sim([1, 1], 9, masks_bits=2, pairs=(1,))
    miller_loop_sim([1, 2, 1], 10, pairs=(1,), pair_mode=True)       # the same routine for lane pairs (products in pairs)
    miller_loop_sim([1, 1], 11, masks_bits=2, pairs=(1,), pair_mode=True)
    # the shell around the bodies: one forward exit, every far jump backwards, the phase counter picks RUNS[1..5]
    full, pieces, _ = t.miller_loop_d_routine()
    assert sum(1 for l in full if l.startswith("s_setpc_b64")) == 2 and "5:" in full and "4:" not in full


def test_g2_doubling_runs():
    rng = random.Random(21)
    full, pieces, st = t.g2_dbl_d_routine()
    for trial in range(4):
        true = [rng.randrange(P) for _ in range(6)]
        if trial == 0:
            true = [0, 0, 1, 0, 0, 0]                   # the point at infinity (0 : 1 : 0)
        m = Machine(ROUT); m.run(t.shell_constants())
        for i in range(6):
            m.v[t.G2D_ARG[i]:t.G2D_ARG[i] + 12] = limbs(true[i] * R384 % P)
        m.run(pieces["pro"][1:])
        cur = list(true)
        for r in range([1, 2, 3, 4][trial]):
            m.run(pieces["body"])
            mp = run_model(t.prog_g2_dbl_d, {("a", i): cur[i] for i in range(6)}, {})
            cur = [mp.out_home[("a", i)] for i in range(6)]
        m.run(pieces["epi"])
        for i in range(6):
            assert from_limbs(m.v[t.G2D_ARG[i]:t.G2D_ARG[i] + 12]) == cur[i] * R384 % P, (trial, i)


def test_miller_loop_routine_full_schedule():
    """the complete loop as the kernel runs it: 63 doubling iterations and 5 addition steps per pair, ~6 million interpreted instructions"""
    import bls12_381 as M
    q = M.g2_mul(M.G2, 0x1234567)
    T = miller_loop_sim(t.RUNS, 7, q0=q)
    # pair 0's running point starts at Q_0 and walks the bits of |x|: when the loop returns, slots T0 .. T0 + 5 hold [|x|] Q_0 (X : Y : Z)
    t0 = t.T_SLOT(0, 0, 0)
    X, Y, Z = ((T[t0], T[t0 + 1]), (T[t0 + 2], T[t0 + 3]), (T[t0 + 4], T[t0 + 5]))
    zi = M.f2_inv(Z)
    assert (M.f2_mul(X, zi), M.f2_mul(Y, zi)) == M.g2_mul(q, M.X_ABS)
    gen = open(os.path.join(ROOT, "milagro_bls_amd", "csrc", "mbls_towerd_asm.inc")).read()
    assert "#define MBLS_GEN_MILLER_T0_SLOT %d\n" % t0 in gen


# ---------------------------------------------------------------------------------------------- the final exponentiation routine
import gen_fp_asm as g1          # noqa: E402  (the inversion routine the easy part calls)

FEXP_ROUT = dict(ROUT)
FEXP_ROUT["mbls_fp_inv_gcd_asm_fn"] = g1.fp_inv_gcd_body(unrolled=True)


def fexp_sim(runs, seed, f=None):
    """the routine's bodies in the order its control skeleton runs them, with `runs` as the squarings before each of the six saves of
    every power; returns (result of the instruction stream, result of the same programs on field values)"""
    full, pieces, st = t.final_exp_d_routine()
    assert not any("scratch" in l or "buffer_" in l for l in full)
    rng = random.Random(seed)
    f = f or [rng.randrange(P) for _ in range(12)]
    m = Machine(FEXP_ROUT); m.v[252] = LADDR
    m.s[68] = GBASE & 0xFFFFFFFF; m.s[69] = GBASE >> 32; m.s[70] = STRIDE * 4
    for i in range(12):
        ws_put(m, t.FEXP_IN_SLOT + i, f[i] * R384 % P)
    m.run(pieces["pro"])
    state = {("g", t.FEXP_IN_SLOT + i): f[i] for i in range(12)}
    masks = {}

    def step(name, rec=0):
        m.s[71] = rec * m.s[72]                                        # the run-time record offset the skeleton maintains
        for j in range(t.K_REC):
            if ("gd", t.K_SLOT + t.K_REC * rec + j) in state:
                state[("gk", t.K_SLOT + j)] = state[("gd", t.K_SLOT + t.K_REC * rec + j)]
        m.run(pieces[name])
        mp = run_model(t.FEXP_BODIES[name], state, masks)
        for loc, v in mp.out_home.items():
            state[loc] = v
        for slot, v in mp.out_g.items():
            state[("gd", t.K_REC * rec + slot[1]) if isinstance(slot, tuple) else ("gd", slot)] = v

    def power():
        step("pstart")
        for ph, n in enumerate(runs):
            for _ in range(n):
                step("csqr")
            step("psave", ph)
        step("pinv"); step("pfirst", 0)
        for rec in range(1, 6):
            step("pmul", rec)
    step("easy")
    for k in range(5):
        power()
        if k < 4:
            step(["step_conj", "step_conj", "step_frob", "step_base"][k])
    step("tail")
    m.run(pieces["epi"][:-1])
    got = [from_limbs(m.v[t.F_OUT[i]:t.F_OUT[i] + 12]) for i in range(12)]
    return got, [state[("a", i)] * R384 % P for i in range(12)], f


def test_final_exponentiation_routine_short_schedule():
    got, model, _ = fexp_sim([1, 2, 1, 1, 1, 1], 41)
    assert got == model
    # a Miller value in Fp6 (here 1: both pairs skipped) goes to 1 in the easy part: every compressed coefficient is zero, the
    # decompression denominators vanish and the zero handling must give back 1 -- for any schedule
    one = [R384 % P] + [0] * 11
    for f in ([1] + [0] * 11, [5, 7, 11, 13, 17, 19] + [0] * 6):
        got, model, _ = fexp_sim([1, 2, 1, 1, 1, 1], 42, f=f)
        assert got == model == one


def test_final_exponentiation_routine_full_schedule():
    """the complete routine (~5 million interpreted instructions) against the Python model's final exponentiation -- an independent
    statement of the same function: f^((p^12-1)/r) by Frobenius maps, a generic inversion and one literal 1269-bit power. The routine
    computes the cube of that value (its hard part is 3 (p^4-p^2+1)/r, see mbls_pairing.h)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pymodel import bls12_381 as M
    got, model, f = fexp_sim(t.POW_RUNS, 43)
    assert got == model
    ri = pow(R384, -1, P)
    tower = [(got[2 * e] * ri % P, got[2 * e + 1] * ri % P) for e in range(6)]          # c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2
    fin = [(f[2 * e], f[2 * e + 1]) for e in range(6)]
    wpow = lambda x: [x[0], x[3], x[1], x[4], x[2], x[5]]                                # coefficients of w^0 .. w^5
    want = M.f12_pow(M.final_exp(wpow(fin)), 3)
    assert M.f12_eq(wpow(tower), want)


def fexp2_sim(runs, seed, lanes=(6, 7)):
    """the two-lane variant (final_exp_d_routine(two_lane=True), k_final2): two machines in lockstep that share the LDS column and the workspace
    item, as the kernel sets the two lanes of an item up; the cross-lane moves of the compressed squaring exchange their registers"""
    from asm_sim import run_pair
    full, pieces, st = t.final_exp_d_routine(two_lane=True)
    assert not any("scratch" in l or "buffer_" in l for l in full)
    rng = random.Random(seed)
    f = [rng.randrange(P) for _ in range(12)]
    ma, mb = Machine(FEXP_ROUT), Machine(FEXP_ROUT)
    mb.lds = ma.lds; mb.mem = ma.mem
    for m, lane in ((ma, lanes[0]), (mb, lanes[1])):
        m.lane = lane; m.v[252] = LADDR; m.model_exec = True               # (the bodies outside the squaring chains have their products in pairs)
        m.s[68] = GBASE & 0xFFFFFFFF; m.s[69] = GBASE >> 32; m.s[70] = STRIDE * 4
    for i in range(12):
        ws_put(ma, t.FEXP_IN_SLOT + i, f[i] * R384 % P)
    run_pair(ma, mb, pieces["pro"])

    def step(name, rec=0):
        for m in (ma, mb):
            m.s[71] = rec * m.s[72]
        run_pair(ma, mb, pieces[name])

    def power():
        step("pstart")
        for ph, n in enumerate(runs):
            for _ in range(n):
                step("csqr")
            step("psave", ph)
        step("pinv"); step("pfirst", 0)
        for rec in range(1, 6):
            step("pmul", rec)
    step("easy")
    for k in range(5):
        power()
        if k < 4:
            step(["step_conj", "step_conj", "step_frob", "step_base"][k])
    step("tail")
    run_pair(ma, mb, pieces["epi"][:-1])
    outs = [[from_limbs(m.v[t.F_OUT[i]:t.F_OUT[i] + 12]) for i in range(12)] for m in (ma, mb)]
    assert outs[0] == outs[1], "the two lanes of an item must end with the same value"
    return outs[0]


def test_two_lane_compressed_squaring_body():
    """csqr2_body on a lane pair against the Granger-Scott formulas (cyc_sqr_formula restricted to z2..z5) for several squarings in a row:
    the roles swap every squaring and the pair's four values always make up the state of the one-lane body"""
    from asm_sim import run_pair
    rng = random.Random(77)
    body = t.csqr2_body()
    assert len(body) < 4000 and sum(1 for l in body if "v_mov_b32_dpp" in l) == 56
    z = [rng.randrange(P) for _ in range(8)]                       # z2.0, z2.1, z3.0, z3.1, z4.0, z4.1, z5.0, z5.1 (true values)
    ma, mb = Machine(ROUT), Machine(ROUT)
    for m in (ma, mb):
        m.run(t.shell_constants())
    ma.s[("pair", 92)] = 0; mb.s[("pair", 92)] = 1
    for k in range(4):
        ma.v[14 * k:14 * k + 14] = normalised_digits(z[k] * R392 % P)
        mb.v[14 * k:14 * k + 14] = normalised_digits((z[4 + k] * R392 % P) - (P if k % 2 else 0))      # reduced representatives of either sign
    f2 = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    xi2 = lambda a: ((a[0] - a[1]) % P, (a[0] + a[1]) % P)
    for it in range(5):
        z2, z3, z4, z5 = (z[0], z[1]), (z[2], z[3]), (z[4], z[5]), (z[6], z[7])
        def fp4(a, b):
            t0, t1 = f2(a, a), f2(b, b)
            s_ = f2(((a[0] + b[0]) % P, (a[1] + b[1]) % P), ((a[0] + b[0]) % P, (a[1] + b[1]) % P))
            return ((xi2(t1)[0] + t0[0]) % P, (xi2(t1)[1] + t0[1]) % P), ((s_[0] - t0[0] - t1[0]) % P, (s_[1] - t0[1] - t1[1]) % P)
        t0, t1 = fp4(z2, z3); t2, t3 = fp4(z4, z5)
        n4 = tuple((3 * t0[i] - 2 * z4[i]) % P for i in range(2)); n5 = tuple((3 * t1[i] + 2 * z5[i]) % P for i in range(2))
        x = xi2(t3)
        n2 = tuple((3 * x[i] + 2 * z2[i]) % P for i in range(2)); n3 = tuple((3 * t2[i] - 2 * z3[i]) % P for i in range(2))
        z = [n2[0], n2[1], n3[0], n3[1], n4[0], n4[1], n5[0], n5[1]]
        run_pair(ma, mb, body)
        # the lane that was in role A now holds (z4', z5'), the other one (z2', z3'); the role bits have swapped
        holder_a, holder_b = (mb, ma) if it % 2 == 0 else (ma, mb)          # holder_a: the lane that now holds (z2, z3)
        assert holder_a.s[("pair", 92)] == 0 and holder_b.s[("pair", 92)] == 1
        for k in range(4):
            for m, base in ((holder_a, 0), (holder_b, 4)):
                got = from_digits_signed(m.v[14 * k:14 * k + 14])
                assert (got - z[base + k] * R392) % P == 0 and t.REDUCED.vlo <= got <= t.REDUCED.vhi, (it, k, base)
                assert all(0 <= s32(w) < (1 << 28) for w in m.v[14 * k:14 * k + 13])


def test_two_lane_final_exponentiation_routine():
    """the whole two-lane routine on a lane pair (short schedule, then the real one) gives, on both lanes, what the one-lane routine gives"""
    got, model, f = fexp_sim([1, 2, 1, 1, 1, 1], 41)
    assert fexp2_sim([1, 2, 1, 1, 1, 1], 41) == got == model
    got, model, f = fexp_sim(t.POW_RUNS, 43)
    assert fexp2_sim(t.POW_RUNS, 43, lanes=(33, 32)) == got == model


# ---------------------------------------------------------------------------------------------- the public-key sum
def g1_model_masks(which, init, masks):
    """the program on field values; returns the model (its masks dictionary is updated in place by iszero / mask_orn2)"""
    prog = t.prog_g1_dbl if which == "dbl" else (lambda: t.prog_g1_step(which))
    return run_model(prog, init, masks)


def jac_affine(X, Y, Z):
    if Z % P == 0:
        return None
    zi = pow(Z, -1, P)
    return (X * zi * zi % P, Y * zi * zi * zi % P)


def g1_add_affine(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0: return None
        l = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        l = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (l * l - a[0] - b[0]) % P
    return (x, (l * (a[0] - x) - a[1]) % P)


G1_GEN = (t.G1_X, t.G1_Y)


def g1_mul_affine(pt, k):
    r = None
    while k:
        if k & 1: r = g1_add_affine(r, pt)
        pt = g1_add_affine(pt, pt); k >>= 1
    return r


@pytest.mark.parametrize("mode", ["raw", "indexed"])
def test_g1_sum_step_and_doubling(mode):
    """one key into the running sum, against affine chord-and-tangent arithmetic: general position, sum at infinity, key flagged infinite,
    key = -sum (result infinity), key = sum (the masks must select the doubling body, which is then run), off-curve key (raw)"""
    step, _ = t.build_g1(mode)
    dbl, _ = t.build_g1("dbl")
    rng = random.Random(77)
    for case in ("general", "acc_inf", "key_inf", "inverse", "double", "offcurve"):
        if case == "offcurve" and mode != "raw":
            continue
        accp = g1_mul_affine(G1_GEN, rng.randrange(1, 1 << 64))
        key = g1_mul_affine(G1_GEN, rng.randrange(1, 1 << 64))
        if case == "inverse": key = (accp[0], P - accp[1])
        if case == "double": key = accp
        if case == "offcurve": key = (key[0], (key[1] + 1) % P)
        z = rng.randrange(1, P)
        acc = (accp[0] * z * z % P, accp[1] * z * z * z % P, z) if case != "acc_inf" else (rng.randrange(P), rng.randrange(P), 0)
        m = Machine(ROUT); m.run(t.shell_constants())
        init, masks = {}, {t.M_INF2: 1 if case == "key_inf" else 0}
        m.s[("pair", 48)] = masks[t.M_INF2]
        for i in range(3):
            init[("a", i)] = acc[i]
            m.a[14 * i:14 * i + 14] = normalised_digits(acc[i] * R392 % P + rng.choice([-1, 0]) * 0)
        if mode == "raw":                                   # plain integers below p
            init[("v", 8)], init[("v", 9)] = key[0] * RI392 % P, key[1] * RI392 % P      # the model's value is rep / 2^392
            m.v[112:126] = normalised_digits(key[0]); m.v[126:140] = normalised_digits(key[1])
        else:                                               # Montgomery words (2^384) cut as digits of words * 2^8
            init[("v", 8)], init[("v", 9)] = key[0], key[1]
            m.v[112:126] = normalised_digits((key[0] * R384 % P) << 8); m.v[126:140] = normalised_digits((key[1] * R384 % P) << 8)
        m.run(step)
        mp = g1_model_masks(mode, init, masks)
        got = [from_digits_signed(m.a[14 * i:14 * i + 14]) * RI392 % P for i in range(3)]
        assert got == [mp.out_home[("a", i)] for i in range(3)], case
        for name, idx in ((t.M_H0, 52), (t.M_R0, 54), (t.M_INF1, 84)):
            assert m.s[("pair", idx)] == masks[name], (case, name)
        infk = masks[t.M_INF2F] if mode == "raw" else masks[t.M_INF2]
        if mode == "raw":
            assert m.s[("pair", 86)] == masks[t.M_INF2F] == (1 if case in ("key_inf", "offcurve") else 0)
        need_dbl = masks[t.M_H0] and masks[t.M_R0] and not masks[t.M_INF1] and not infk
        assert bool(need_dbl) == (case == "double")
        if need_dbl:
            m.run(dbl)
            old = {("a", 5 + i): mp.out_home[("a", 5 + i)] for i in range(3)}
            assert [old[("a", 5 + i)] for i in range(3)] == list(acc)
            mp = g1_model_masks("dbl", old, {})
            got = [from_digits_signed(m.a[14 * i:14 * i + 14]) * RI392 % P for i in range(3)]
            assert got == [mp.out_home[("a", i)] for i in range(3)]
        want = accp if case != "acc_inf" else None
        if case not in ("key_inf", "offcurve"):
            want = g1_add_affine(want, key)
        assert jac_affine(*got) == want, case


def test_g1_raw_key_decoding():
    """byte order, flag bits and range checks of the 96-byte format (g1_decode_uncompressed_w in mbls_curve.h)"""
    dec = t.g1_decode_raw()
    rng = random.Random(5)
    pt = g1_mul_affine(G1_GEN, 12345)
    cases = []
    enc = lambda x, y: x.to_bytes(48, "big") + y.to_bytes(48, "big")
    cases.append((enc(*pt), 0, 0, pt))
    cases.append((bytes([0x40]) + bytes(95), 1, 0, None))                                    # canonical infinity
    cases.append((bytes([0x40]) + bytes(94) + b"\x01", 1, 1, None))                           # infinity with a stray bit
    cases.append((bytes([0x41]) + bytes(95), 1, 1, None))
    cases.append((bytes([enc(*pt)[0] | 0x80]) + enc(*pt)[1:], 1, 1, None))                    # compression flag on a 96-byte key
    cases.append((bytes([enc(*pt)[0] | 0x20]) + enc(*pt)[1:], 1, 1, None))                    # sign flag
    cases.append((enc(P, pt[1]), 1, 1, None))                                                 # x = p
    cases.append((enc(pt[0], P + 5), 1, 1, None))
    cases.append((enc(P - 1, P - 1), 0, 0, (P - 1, P - 1)))                                   # in range (off curve: the body's business)
    for blob, inf, bad, xy in cases:
        m = Machine(ROUT); m.run(t.shell_constants()); m.s[77] = 0x00010203
        for j in range(24):
            m.v[224 + j] = int.from_bytes(blob[4 * j:4 * j + 4], "little")
        for j in range(12):
            m.v[196 + j] = t.P_LIMBS[j]
        m.run(dec)
        assert (m.s[("pair", 48)], m.s[("pair", 88)]) == (inf, bad), blob[:2].hex()
        if xy:
            assert from_digits_signed(m.v[112:126]) == xy[0] and from_digits_signed(m.v[126:140]) == xy[1]


@pytest.mark.parametrize("mode", ["raw", "indexed"])
def test_g1_sum_routine_shell(mode):
    """prologue, per-key fetch / decode / step / status, epilogue in the order the routine's loop runs them, for one lane with five keys
    (a repeated key, an infinite key, an undecodable key) -> the sum in the workspace and the status bits"""
    full, pieces, _ = t.g1_aggregate_d_routine(mode)
    assert not any("scratch" in l or "buffer_" in l for l in full)
    pts = [g1_mul_affine(G1_GEN, s) for s in (5, 7, 11)]
    keys = [pts[0], pts[1], "inf", pts[1], "bad", pts[2]]
    KEYS, IDX, TAB = 0x7E0000100000, 0x7E0000200000, 0x7E0000300000
    m = miller_machine(0)
    m.v[252] = LADDR
    words = lambda b: [int.from_bytes(b[4 * j:4 * j + 4], "little") for j in range(len(b) // 4)]
    if mode == "raw":
        for n_, kx in enumerate(keys + [pts[0]]):                  # one more record: the fetch runs a key ahead
            blob = bytes([0x40]) + bytes(95) if kx == "inf" else bytes([0x20]) + bytes(95) if kx == "bad" else kx[0].to_bytes(48, "big") + kx[1].to_bytes(48, "big")
            for j, w in enumerate(words(blob)):
                m.mem[KEYS + 96 * n_ + 4 * j] = w
        m.v[248], m.v[249] = KEYS & 0xFFFFFFFF, KEYS >> 32
    else:
        table = [pts[0], pts[1], "inf", pts[2]]
        for n_, kx in enumerate(table):
            rec = [0] * 32
            if kx == "inf":
                rec[24] = 1
            else:
                rec[0:12] = limbs(kx[0] * R384 % P); rec[12:24] = limbs(kx[1] * R384 % P)
            for j, w in enumerate(rec):
                m.mem[TAB + 128 * n_ + 4 * j] = w
        ids = [0, 1, 2, 1, 99, 3, 0, 0]                            # 99: outside the table = undecodable
        for n_, w in enumerate(ids):
            m.mem[IDX + 4 * n_] = w
        m.v[248], m.v[249] = IDX & 0xFFFFFFFF, IDX >> 32
        m.s[94], m.s[95], m.s[96] = TAB & 0xFFFFFFFF, TAB >> 32, len(table)
    m.v[250] = len(keys)
    m.run(pieces["pro"])
    want = None
    for n_, kx in enumerate(keys):
        m.s[39] = n_
        m.run(pieces["decode"]); m.run(pieces["nxt"]); m.run(pieces["step"]); m.run(pieces["post"])
        h0, r0, i1 = m.s[("pair", 52)], m.s[("pair", 54)], m.s[("pair", 84)]
        i2 = m.s[("pair", 86)] if mode == "raw" else m.s[("pair", 48)]
        if h0 and r0 and not i1 and not i2:
            m.run(pieces["dbl"])
        if kx not in ("inf", "bad"):
            want = g1_add_affine(want, kx)
    m.run(pieces["epi"][1:-1])
    ri = pow(R384, -1, P)
    X, Y, Z = [ws_get(m, sl) * ri % P for sl in range(3)]
    assert jac_affine(X, Y, Z) == want
    assert m.v[251] == 3                                            # an infinite and an undecodable key were seen, the sum is finite


# ---------------------------------------------------------------------------------------------- G2 group routines
def g2_piece_runner(kind, two_lane=False):
    """(machine, state, step): step(name) runs a body of g2_group_routine(kind) on the machine and the same program on field values"""
    full, pieces, st = t.g2_group_routine(kind, two_lane) if two_lane else t.g2_group_routine(kind)
    assert not any("scratch" in l or "buffer_" in l for l in full)
    m = miller_machine(0)
    m.run(pieces["pro"])
    state, masks = {}, {}
    ad = t.G2_SLOTS["AD"] if kind == "hash" else t.G2_SLOTS["SIGAD"]
    progs = {"add": lambda: t.prog_g2_add(ad, False), "sub": lambda: t.prog_g2_add(ad, True), "dbl": t.prog_g2_dbl_d, "fix": lambda: t.prog_g2_dbl_d(6, 0),
             "madd": lambda: t.prog_g2_madd(ad)}

    def step(name):
        m.run(pieces[name])
        model_name = "h_start" if name == "h_start2" else name         # the same step on field values: where q1 comes from is the machine's business
        mp = run_model(progs.get(model_name, lambda: t.prog_g2_glue(model_name)), state, masks)
        for loc, v in mp.out_home.items():
            state[loc] = v
        for slot, v in mp.out_g.items():
            state[("gd", slot)] = v
        for i in range(12):                                              # the instruction stream agrees with the model after every body
            if ("a", i) in state and name != "s_compare":
                assert from_digits_signed(m.a[14 * i:14 * i + 14]) * RI392 % P == state[("a", i)], (name, i)

    def add(name="add"):
        step(name)
        for nm, idx in ((t.M_H0, 52), (t.M_R0, 54), (t.M_INF1, 84), (t.M_INF2, 48)):
            assert m.s[("pair", idx)] == masks[nm], (name, nm)
        if masks[t.M_H0] and masks[t.M_R0] and not masks[t.M_INF1] and not masks[t.M_INF2]:
            step("fix")

    def ladder(runs):
        for ph, n_ in enumerate(runs):
            for _ in range(n_):
                step("dbl")
            if ph < len(runs) - 1:
                add("add" if kind == "hash" else "madd")             # the signature routine adds its affine base point
    return m, state, masks, step, add, ladder


def _g2m():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pymodel import bls12_381 as M
    return M


def jac2_affine(M, X, Y, Z):
    if M.f2_is_zero(Z):
        return None
    zi = M.f2_inv(Z); zi2 = M.f2_sqr(zi)
    return (M.f2_mul(X, zi2), M.f2_mul(Y, M.f2_mul(zi2, zi)))


def test_g2_subgroup_routine():
    """psi(P) = [x]P through the routine's bodies in its control order (full ladder) for a point of G2, a curve point outside G2 and
    a point whose ladder meets the doubling case; the verdict against the model's [r]P test"""
    M = _g2m()
    rng = random.Random(8)
    def curve_point():
        while True:
            x = (rng.randrange(P), rng.randrange(P))
            y = M.f2_sqrt(M.f2_add(M.f2_mul(M.f2_sqr(x), x), M.B2))
            if y is not None:
                return (x, y)
    outside = curve_point()
    inside = M.g2_mul(M.G2, rng.randrange(1, M.R))
    for pt, want in ((inside, True), (outside, False)):
        assert M.subgroup_check_g2(pt) == want
        m, state, masks, step, add, ladder = g2_piece_runner("sig")
        for i, c in enumerate((pt[0][0], pt[0][1], pt[1][0], pt[1][1])):
            state[("g", t.G2_SLOTS["SIG"] + i)] = c
            ws_put(m, t.G2_SLOTS["SIG"] + i, c * R384 % P)
        step("s_start")
        ladder(t.RUNS)
        acc = [(state[("a", 2 * e)], state[("a", 2 * e + 1)]) for e in range(3)]
        assert jac2_affine(M, *acc) == M.g2_mul(pt, 0xd201000000010000)
        step("s_compare")
        ex, ey, ia, ib = masks[t.M_H0], masks[t.M_R0], masks[t.M_INF1], masks[t.M_INF2]
        assert (m.s[("pair", 52)], m.s[("pair", 54)], m.s[("pair", 84)], m.s[("pair", 48)]) == (ex, ey, ia, ib)
        assert bool((ia and ib) or (not ia and not ib and ex and ey)) == want


def test_g2_addition_cases():
    """the addition body and its doubling fix-up: general position, equal operands, opposite operands, either operand at infinity"""
    M = _g2m()
    rng = random.Random(9)
    A = M.g2_mul(M.G2, 7); Bp = M.g2_mul(M.G2, 11)
    for case in ("general", "equal", "opposite", "acc_inf", "ad_inf"):
        m, state, masks, step, add, ladder = g2_piece_runner("hash")
        a = A if case != "acc_inf" else None
        b = {"general": Bp, "equal": A, "opposite": M.g2_neg(A), "acc_inf": Bp, "ad_inf": None}[case]
        def jac(pt):
            if pt is None:
                return [(rng.randrange(P), rng.randrange(P)), (rng.randrange(P), rng.randrange(P)), (0, 0)]
            z = (rng.randrange(1, P), rng.randrange(P)); z2 = M.f2_sqr(z)
            return [M.f2_mul(pt[0], z2), M.f2_mul(pt[1], M.f2_mul(z2, z)), z]
        ja, jb = jac(a), jac(b)
        for e in range(3):
            for i in range(2):
                state[("a", 2 * e + i)] = ja[e][i]
                m.a[14 * (2 * e + i):14 * (2 * e + i) + 14] = normalised_digits(ja[e][i] * R392 % P)
                state[("gd", t.G2_SLOTS["AD"] + 2 * e + i)] = jb[e][i]
                rep = jb[e][i] * R392 % P
                ws_put(m, t.G2_SLOTS["AD"] + 2 * e + i, rep + P if rep < P // 2 else rep)
        add()
        acc = [(state[("a", 2 * e)], state[("a", 2 * e + 1)]) for e in range(3)]
        assert jac2_affine(M, *acc) == M.g2_add(a, b), case


def test_g2_hash_routine():
    """The message phase's generated routine through its bodies in its control order, from the two field elements u0, u1 to H: two
    map_to_curve evaluations (run-time records 0 and 1), q0 + q1, the cofactor clearing with two full ladders -- against the Python
    model (sswu_g2, iso3_g2, g2_add, clear_cofactor_g2)"""
    M = _g2m()
    import gen_fp_asm as gf
    rng = random.Random(10)
    S = t.G2_SLOTS
    u = [(rng.randrange(P), rng.randrange(P)) for _ in range(2)]
    m, state, masks, step, add, ladder = g2_piece_runner("hash")
    m.routines.update(gf.pow_subroutines()); m.routines["mbls_fp_pow_pm3d4_asm_fn"] = gf.pow_body(gf.EXP_PM3D4)
    full, pieces, _ = t.g2_group_routine("hash")
    for rec in range(2):
        for i in range(2):
            ws_put(m, S["U"] + 6 * rec + i, u[rec][i] * R384 % P)
        m.s[71] = rec * (STRIDE * 4 * 12 * 6)
        m.run(pieces["sswu"])
        mp = run_model(t.prog_sswu, {("gka", S["U"] + i): u[rec][i] for i in range(2)}, masks)
        for slot, v in mp.out_g.items():
            state[("gd", slot[1] + 6 * rec)] = v
    step("h_start"); add(); step("h_base1"); ladder(t.RUNS); step("h_after1"); step("dbl"); step("h_psi2"); add("sub")
    step("h_t3"); add(); step("h_base2"); ladder(t.RUNS); step("h_after2"); add()
    step("h_ad_t1"); add("sub"); step("h_ad_p"); add("sub")
    acc = [(state[("a", 2 * e)], state[("a", 2 * e + 1)]) for e in range(3)]
    want = M.clear_cofactor_g2(M.g2_add(M.iso3_g2(M.sswu_g2(u[0])), M.iso3_g2(M.sswu_g2(u[1]))))
    assert jac2_affine(M, *acc) == want and M.subgroup_check_g2(want)
    m.run(pieces["epi"][:-1])
    ri = pow(R384, -1, P)
    got = [(ws_get(m, S["H"] + 2 * e) * ri % P, ws_get(m, S["H"] + 2 * e + 1) * ri % P) for e in range(3)]
    assert jac2_affine(M, *got) == want



def test_g2_hash_routine_two_lanes_per_message():
    """k_hash2's routine on a lane pair: the even lane maps u0, the odd lane (a workspace item of its own, 4 bytes further on, u1 in the slots
    of u0) maps u1 with the SAME body; then both fetch q0 / q1 from the even / odd lane's item (per-lane offsets) and walk the addition and
    the cofactor clearing together, products in pairs -- the model's H in BOTH items"""
    from asm_sim import run_pair
    M = _g2m()
    import gen_fp_asm as gf
    rng = random.Random(12)
    S = t.G2_SLOTS
    u = [(rng.randrange(P), rng.randrange(P)) for _ in range(2)]
    full, pieces, st = t.g2_group_routine("hash", two_lane=True)
    assert st["dbl"].get("pairs") == 3 and st["add"].get("pairs") == 7 and "pairs" not in st["sswu"] and "pairs" not in st["fix"]
    # the control skeleton is the one-lane routine's minus one map_to_curve body (a shape check: the bodies themselves are run piece by piece below)
    full1, pieces1, _ = t.g2_group_routine("hash")
    assert full.count("CALL") == 0 and sum(1 for l in full if l.startswith("s_mov_b64 exec, %s" % t.PAIR_ROLE)) >= 2 * (3 + 7 + 7)
    ma, mb = miller_machine(0), miller_machine(0)
    mb.mem = ma.mem                                                # own registers and LDS columns, the same memory
    mb.s[68] = (GBASE + 4) & 0xFFFFFFFF; mb.s[69] = (GBASE + 4) >> 32   # the odd lane's item lies one word further on; v252 is the same column
    for lane, mach in ((0, ma), (1, mb)):                          # address in both (LDS is per machine here)
        mach.lane = 6 + lane; mach.model_exec = True
        mach.routines = dict(ROUT); mach.routines.update(gf.pow_subroutines()); mach.routines["mbls_fp_pow_pm3d4_asm_fn"] = gf.pow_body(gf.EXP_PM3D4)
    run_pair(ma, mb, pieces["pro"])
    assert (ma.s[("pair", 94)], mb.s[("pair", 94)]) == (0, 1)
    masks = {}
    q = []
    for lane, mach in ((0, ma), (1, mb)):
        for i in range(2):
            for j, w in enumerate(limbs(u[lane][i] * R384 % P)):
                mach.mem[ws_addr(S["U"] + i, j) + 4 * lane] = w
        mach.s[71] = 0
    run_pair(ma, mb, pieces["sswu"])
    # from here on the model of the one-lane routine applies to BOTH lanes: the machine's only own business is where q0 and q1 come from
    state = {}
    for lane in (0, 1):
        mp = run_model(t.prog_sswu, {("gka", S["U"] + i): u[lane][i] for i in range(2)}, masks)
        for slot, v in mp.out_g.items():
            state[("gd", slot[1] + 6 * lane)] = v
    ad = S["AD"]
    progs = {"add": lambda: t.prog_g2_add(ad, False), "sub": lambda: t.prog_g2_add(ad, True), "dbl": t.prog_g2_dbl_d, "fix": lambda: t.prog_g2_dbl_d(6, 0)}

    def step(name, model=None):
        run_pair(ma, mb, pieces[name])
        model = model or name
        if model == "none":
            return
        mp = run_model(progs.get(model, lambda: t.prog_g2_glue(model)), state, masks)
        for loc, v in mp.out_home.items():
            state[loc] = v
        for slot, v in mp.out_g.items():
            state[("gd", slot)] = v
        for m_ in (ma, mb):
            for i in range(6):
                if ("a", i) in state:
                    assert from_digits_signed(m_.a[14 * i:14 * i + 14]) * RI392 % P == state[("a", i)], (name, i)

    def add(name="add"):
        step(name)
        for m_ in (ma, mb):
            for nm, idx in ((t.M_H0, 52), (t.M_R0, 54), (t.M_INF1, 84), (t.M_INF2, 48)):
                assert m_.s[("pair", idx)] == masks[nm], (name, nm)
        if masks[t.M_H0] and masks[t.M_R0] and not masks[t.M_INF1] and not masks[t.M_INF2]:
            step("fix")

    def ladder(runs):
        for ph, n_ in enumerate(runs):
            for _ in range(n_):
                step("dbl")
            if ph < len(runs) - 1:
                add("add")
    step("voff_q1", "none"); step("h_q1", "none"); step("voff_q0", "none"); step("h_q0", "h_start")      # together they are the one-lane routine's h_start
    add(); step("h_base1"); ladder(t.RUNS); step("h_after1"); step("dbl"); step("h_psi2"); add("sub")
    step("h_t3"); add(); step("h_base2"); ladder(t.RUNS); step("h_after2"); add()
    step("h_ad_t1"); add("sub"); step("h_ad_p"); add("sub")
    want = M.clear_cofactor_g2(M.g2_add(M.iso3_g2(M.sswu_g2(u[0])), M.iso3_g2(M.sswu_g2(u[1]))))
    run_pair(ma, mb, pieces["epi"][:-1])
    ri = pow(R384, -1, P)
    for lane in (0, 1):
        got = [tuple(from_limbs([ma.mem[ws_addr(S["H"] + 2 * e + h, j) + 4 * lane] for j in range(12)]) * ri % P for h in range(2)) for e in range(3)]
        assert jac2_affine(M, *got) == want, lane


def test_compressed_squaring_decompression_formulas():
    """The identities the decompression rests on, on a random element of the cyclotomic subgroup (tower coefficients z0, z4, z3, z2,
    z1, z5 as in fp12_cyc_sqr):  4 z2 z1 = xi z5^2 + 3 z4^2 - 2 z3,  z3 z1 - 2 z4 z5 = z2 (1 - z0) / xi  (so z1 = 2 z4 z5 / z3 where
    z2 = 0),  z0 = xi (2 z1^2 + z2 z5 - 3 z3 z4) + 1; and the decompression body on a record with z2 = 0 (the branch no random input
    reaches) against the same program on field values."""
    M = _g2m()
    rng = random.Random(12)
    f = [(rng.randrange(P), rng.randrange(P)) for _ in range(6)]
    tt = M.f12_mul(M.f12_conj(f), M.f12_inv(f))
    m = M.f12_mul(M.f12_frob(M.f12_frob(tt)), tt)
    z0, z4, z3, z2, z1, z5 = m[0], m[2], m[4], m[1], m[3], m[5]          # w^0, w^2, w^4, w^1, w^3, w^5
    xi = (1, 1)
    mxi = lambda a: M.f2_mul(a, xi)
    assert M.f2_eq(M.f2_muls(M.f2_mul(z2, z1), 4), M.f2_sub(M.f2_add(mxi(M.f2_sqr(z5)), M.f2_muls(M.f2_sqr(z4), 3)), M.f2_muls(z3, 2)))
    assert M.f2_eq(mxi(M.f2_sub(M.f2_mul(z3, z1), M.f2_muls(M.f2_mul(z4, z5), 2))), M.f2_mul(z2, M.f2_sub((1, 0), z0)))
    assert M.f2_eq(z0, M.f2_add(mxi(M.f2_sub(M.f2_add(M.f2_muls(M.f2_sqr(z1), 2), M.f2_mul(z2, z5)), M.f2_muls(M.f2_mul(z3, z4), 3))), (1, 0)))
    # the body with z2 = 0: masks select the second numerator
    body, _ = t.build_fexp("pfirst")
    for zero_z2 in (True, False):
        mch = miller_machine(0); mch.run(t.shell_constants()); mch.s[71] = 0
        init = {}
        for j in range(t.K_REC):
            x = 0 if (zero_z2 and j < 2) else rng.randrange(P)
            init[("gk", t.K_SLOT + j)] = x
            rep = x * R392 % P
            ws_put(mch, t.K_SLOT + j, rep + P if rep < P // 2 else rep)
        masks = {}
        mp = run_model(t.prog_fexp_pfirst, init, masks)
        mch.run(body)
        assert masks[t.TMASK] == (1 if zero_z2 else 0)
        for i in range(12):
            assert from_digits_signed(mch.a[14 * i:14 * i + 14]) * RI392 % P == mp.out_home[("a", i)], (zero_z2, i)


def test_map_to_curve_body():
    """simplified SWU + 3-isogeny on u (run-time record 0 and 1) against the Python model's sswu_g2 / iso3_g2: both square classes of
    gx1 occur among the samples, and the degenerate u = 0 (tv2 = 0: the exceptional denominator)"""
    M = _g2m()
    import gen_fp_asm as gf
    rout = dict(ROUT); rout.update(gf.pow_subroutines()); rout["mbls_fp_pow_pm3d4_asm_fn"] = gf.pow_body(gf.EXP_PM3D4)
    body, _ = t.build_g2("sswu")
    rng = random.Random(21)
    S = t.G2_SLOTS
    seen = set()
    for trial, rec in ((0, 0), (1, 1), (2, 0), (3, 1), (4, 0)):
        u = (rng.randrange(P), rng.randrange(P)) if trial < 4 else (0, 0)
        m = Machine(rout); m.v[252] = LADDR
        m.s[68] = GBASE & 0xFFFFFFFF; m.s[69] = GBASE >> 32; m.s[70] = STRIDE * 4
        m.run(t.shell_constants())
        m.s[71] = rec * (STRIDE * 4 * 12 * 6)
        for i in range(2):
            ws_put(m, S["U"] + 6 * rec + i, u[i] * R384 % P)
        init, masks = {("gka", S["U"] + i): u[i] for i in range(2)}, {}
        m.run(body)
        mp = run_model(t.prog_sswu, init, masks)
        seen.add(masks[t.M_SQ1])
        got = []
        for i in range(6):
            rep = ws_get(m, S["Q0"] + 6 * rec + i)
            assert (rep * RI392 - mp.out_g[("k", S["Q0"] + i)]) % P == 0, (trial, i)
            got.append(rep * RI392 % P)
        X, Y, Z = (got[0], got[1]), (got[2], got[3]), (got[4], got[5])
        want = M.iso3_g2(M.sswu_g2(u))
        assert jac2_affine(M, X, Y, Z) == want, trial
    assert seen == {0, 1}


def test_three_product_and_four_scan_fp2_multiplication_agree():
    """Karatsuba on the column sums must leave exactly the digits the four plain scans leave (same quotient digits, same results)"""
    rng = random.Random(77)
    for trial in range(6):
        vals = [rng.randrange(-3 * P, 4 * P) for _ in range(4)]
        out = []
        for body in (d.fp2_mul_d_body(), d.fp2_mul_d4_body()):
            m = Machine(); m.run(d.load_constants())
            r2 = random.Random(trial)
            for i, x in enumerate(vals):
                m.v[14 * i:14 * i + 14] = digits_signed(x, 1 << 28, r2)
            m.run(body)
            out.append(list(m.v[70:98]))
        assert out[0] == out[1], trial


def test_paired_fp_products_and_squarings():
    """fp_mulpair_d, fp_mul1_d, fp_sqrpair_d on redundant signed digits at the generator's limits; the squaring scan (cross products once,
    against doubled digits) must leave exactly the digits of the product scan"""
    rng = random.Random(78)
    for trial in range(8):
        dm = [1 << 28, 1 << 29][trial % 2]
        vals = [rng.randrange(-3 * P, 4 * P) for _ in range(4)]
        m = Machine(); m.run(d.load_constants())
        for i, x in enumerate(vals):
            m.v[14 * i:14 * i + 14] = digits_signed(x, dm, rng)
        a_digits = [list(m.v[0:14]), list(m.v[14:28])]
        m.run(d.fp_mulpair_d_body())
        c0, c1 = from_digits_signed(m.v[70:84]), from_digits_signed(m.v[84:98])
        assert (c0 - vals[0] * vals[2] * RI) % P == 0 and (c1 - vals[1] * vals[3] * RI) % P == 0
        m.run(d.fp_mul1_d_body())
        assert (from_digits_signed(m.v[70:84]) - vals[0] * vals[2] * RI) % P == 0
        m2 = Machine(); m2.run(d.load_constants())
        m2.v[0:14], m2.v[14:28] = a_digits
        m2.run(d.fp_sqrpair_d_body())
        s0, s1 = from_digits_signed(m2.v[70:84]), from_digits_signed(m2.v[84:98])
        assert (s0 - vals[0] * vals[0] * RI) % P == 0 and (s1 - vals[1] * vals[1] * RI) % P == 0
        m3 = Machine(); m3.run(d.load_constants())
        m3.v[0:14], m3.v[14:28], m3.v[28:42], m3.v[42:56] = a_digits[0], a_digits[1], a_digits[0], a_digits[1]
        m3.run(d.fp_mulpair_d_body())
        assert list(m3.v[70:98]) == list(m2.v[70:98])


# ---------------------------------------------------------------------------------------------- the blinding routines of verify_multiple
class _LanePair:
    """the two lanes of one signature in the two-lane routines: two machines in lockstep (run_pair) with the same workspace item and, being
    two lanes of one item, the same LDS column -- only `v`, `a`, the lane masks and the lane number are their own. Attribute reads go to the
    even lane; `run` drives both."""

    def __init__(self):
        from asm_sim import run_pair
        self._run_pair = run_pair
        self.ma, self.mb = miller_machine(0), miller_machine(0)
        self.mb.mem = self.ma.mem; self.mb.lds = self.ma.lds
        for lane, mach in ((0, self.ma), (1, self.mb)):
            mach.lane = 6 + lane; mach.model_exec = True

    def run(self, lines):
        self._run_pair(self.ma, self.mb, lines)

    def both(self):
        return (self.ma, self.mb)


def _blind_run(kind, r, ws_init, out_slots, ct=False, two_lane=False, skip_test=False):
    """the instruction streams of g1_blind_routine / g2_blind_routine in their control order (the skeleton's loops mirrored here: table
    of 1 P .. 8 P, top digit, sixteen windows of four doublings + one table addition), returning the words left in out_slots.
    two_lane: the routine for lane pairs on two machines in lockstep; every check below then holds on BOTH lanes."""
    full, pieces, st = t.g1_blind_routine() if kind == "g1" else t.g2_blind_routine(ct=ct, two_lane=two_lane)
    assert not any("scratch" in l or "buffer_" in l for l in full)
    if two_lane:
        return _blind_run_pair(pieces, st, r, ws_init, out_slots)
    if ct:      # constant-time table access: no memory instruction takes its lane offset from anything but the item's own (LADDR), no per-lane record offset
        mem = [l for l in full if l.startswith("global_")]
        assert mem and all(("%s, s[74:75]" % t.LADDR in l) or (l.startswith("global_store") and l.split()[1].rstrip(",") == t.LADDR) for l in mem)
        assert not any(t.VOFF in l.replace(",", " ").split() for l in full)
    m = miller_machine(0)
    m.v[248], m.v[249] = r & 0xFFFFFFFF, r >> 32
    for slot, x in ws_init.items():
        ws_put(m, slot, x)
    m.run(pieces["pro"])
    rp = r + 0x8888888888888888
    assert (m.v[248] | (m.v[249] << 32), m.v[247]) == (rp & 0xFFFFFFFFFFFFFFFF, rp >> 64)

    def add(name):
        m.run(pieces[name])
        pr = lambda nm: m.s[("pair", int(nm[2:nm.index(":")]))]
        if pr(t.M_H0) and pr(t.M_R0) and not pr(t.M_INF1) and not pr(t.M_INF2):
            m.run(pieces["fix"])
    verdict = None
    if kind == "g2":
        m.run(pieces["s_start"])
        for ph, n_ in enumerate(t.RUNS):
            for _ in range(n_):
                m.run(pieces["dbl"])
            if ph < 5:
                add("madd")
        m.run(pieces["s_compare"])
        pr = lambda nm: m.s[("pair", int(nm[2:nm.index(":")]))]
        ex, ey, ia, ib = pr(t.M_H0), pr(t.M_R0), pr(t.M_INF1), pr(t.M_INF2)
        verdict = bool((ia and ib) or (not ia and not ib and ex and ey))
    start, tab, inf = ("start", "tab", "inf") if kind == "g1" else ("b_start", "b_tab", "b_inf")
    m.run(pieces[start]); m.s[71] = 0; m.run(pieces[tab]); m.run(pieces["dbl"]); m.s[71] = m.s[72]; m.run(pieces[tab])
    for _ in range(6):
        add("add" if kind == "g1" else "madd"); m.s[71] += m.s[72]; m.run(pieces[tab])
    def selected(e):
        """ct: the scan left record e of the table in the BL_SEL slots, bit for bit"""
        m.run(pieces["scan"])
        assert [ws_get(m, t.BL_SEL + i) for i in range(6)] == [ws_get(m, t.BL_TAB + 6 * e + i) for i in range(6)]
    m.run(pieces[inf]); m.run(pieces["top"])
    if ct:
        selected(0)
    add("addt")
    for shift in range(60, -4, -4):
        for _ in range(4):
            m.run(pieces["dbl"])
        m.s[38] = shift
        m.run(pieces["digit"])
        d = ((rp >> shift) & 15) - 8
        if ct:
            assert m.v[246] == max(abs(d), 1) - 1
            selected(max(abs(d), 1) - 1)
        else:
            assert m.v[250] == LADDR + (max(abs(d), 1) - 1) * m.s[72]
        add("addt")
    m.run(pieces["epi"][:-1])
    return [ws_get(m, sl) for sl in out_slots], verdict


def _blind_run_pair(pieces, st, r, ws_init, out_slots):
    assert st["dbl"].get("pairs") == 3 and st["addt"].get("pairs") == 7 and st["madd"].get("pairs") == 5 and "pairs" not in st["fix"]
    lp = _LanePair()
    for m in lp.both():
        m.v[248], m.v[249] = r & 0xFFFFFFFF, r >> 32
    for slot, x in ws_init.items():
        ws_put(lp.ma, slot, x)
    lp.run(pieces["pro"])
    assert (lp.ma.s[("pair", 94)], lp.mb.s[("pair", 94)]) == (0, 1)
    rp = r + 0x8888888888888888
    for m in lp.both():
        assert (m.v[248] | (m.v[249] << 32), m.v[247]) == (rp & 0xFFFFFFFFFFFFFFFF, rp >> 64)
    pr = lambda m, nm: m.s[("pair", int(nm[2:nm.index(":")]))]

    def add(name):
        lp.run(pieces[name])
        flags = [(pr(m, t.M_H0), pr(m, t.M_R0), pr(m, t.M_INF1), pr(m, t.M_INF2)) for m in lp.both()]
        assert flags[0] == flags[1], (name, flags)
        h0, r0, i1, i2 = flags[0]
        if h0 and r0 and not i1 and not i2:
            lp.run(pieces["fix"])
    lp.run(pieces["s_start"])
    for ph, n_ in enumerate(t.RUNS):
        for _ in range(n_):
            lp.run(pieces["dbl"])
        if ph < 5:
            add("madd")
    lp.run(pieces["s_compare"])
    verdicts = []
    for m in lp.both():
        ex, ey, ia, ib = pr(m, t.M_H0), pr(m, t.M_R0), pr(m, t.M_INF1), pr(m, t.M_INF2)
        verdicts.append(bool((ia and ib) or (not ia and not ib and ex and ey)))
    assert verdicts[0] == verdicts[1]
    lp.run(pieces["b_start"])
    for m in lp.both():
        m.s[71] = 0
    lp.run(pieces["b_tab"]); lp.run(pieces["dbl"])
    for m in lp.both():
        m.s[71] = m.s[72]
    lp.run(pieces["b_tab"])
    for _ in range(6):
        add("madd")
        for m in lp.both():
            m.s[71] += m.s[72]
        lp.run(pieces["b_tab"])
    lp.run(pieces["b_inf"]); lp.run(pieces["top"])
    add("addt")
    for shift in range(60, -4, -4):
        for _ in range(4):
            lp.run(pieces["dbl"])
        for m in lp.both():
            m.s[38] = shift
        lp.run(pieces["digit"])
        d = ((rp >> shift) & 15) - 8
        for m in lp.both():
            assert m.v[250] == LADDR + (max(abs(d), 1) - 1) * m.s[72]
        add("addt")
    lp.run(pieces["epi"][:-1])
    return [ws_get(lp.ma, sl) for sl in out_slots], verdicts[0]


def test_g2_blinding_routine_two_lanes_per_signature():
    """k_blind_sig2_d's routine (small batches of verify_multiple): the two lanes of a signature -- same workspace item, same LDS column, same scalar -- walk
    the subgroup test and the windowed [r] sig together, the products of every doubling / addition in pairs; same verdict and the same [r] sig as the model,
    on both lanes, for a signature in G2 (random scalar, extreme digits) and a curve point outside it"""
    M = _g2m()
    rng = random.Random(34)
    ri = pow(R384, -1, P)
    inside = M.g2_mul(M.G2, rng.randrange(1, M.R))
    while True:
        x = (rng.randrange(P), rng.randrange(P))
        y = M.f2_sqrt(M.f2_add(M.f2_mul(M.f2_sqr(x), x), M.B2))
        if y is not None:
            outside = (x, y)
            break
    for pt, r in ((inside, rng.randrange(1, 1 << 64)), (inside, 0x0807060504030201), (outside, 0xF00000000000000F)):
        ws = {t.G2_SLOTS["SIG"] + i: c * R384 % P for i, c in enumerate((pt[0][0], pt[0][1], pt[1][0], pt[1][1]))}
        out, verdict = _blind_run("g2", r, ws, range(t.BL_OUT, t.BL_OUT + 6), two_lane=True)
        assert verdict == M.subgroup_check_g2(pt)
        c = [w * ri % P for w in out]
        assert jac2_affine(M, (c[0], c[1]), (c[2], c[3]), (c[4], c[5])) == M.g2_mul(pt, r), hex(r)


def test_g1_blinding_routine():
    """[r] P in G1 by signed 4-bit windows over a per-lane table (g1_blind_routine) against the model's scalar multiplication: random 64-bit
    scalars, scalars with zero / extreme digits, a point at infinity"""
    M = _g2m()
    rng = random.Random(31)
    ri = pow(R384, -1, P)
    pt = M.g1_mul(M.G1, rng.randrange(1, M.R))
    z = rng.randrange(1, P)
    jac = (pt[0] * z * z % P, pt[1] * z * z * z % P, z)
    for r in (rng.randrange(1, 1 << 64), 1, (1 << 64) - 1, 0x8000000000000000, 0x0807060504030201, 0x7777777777777778):
        out, _ = _blind_run("g1", r, {i: jac[i] * R384 % P for i in range(3)}, range(3))
        X, Y, Z = [w * ri % P for w in out]
        assert jac_affine(X, Y, Z) == M.g1_mul(pt, r), hex(r)
    out, _ = _blind_run("g1", 12345, {0: 0, 1: R384 % P, 2: 0}, range(3))
    assert out[2] == 0


def test_g2_blinding_routine():
    """the signature phase of verify_multiple (g2_blind_routine): subgroup verdict + [r] sig against the model, for a signature in G2 and a
    curve point outside it"""
    M = _g2m()
    rng = random.Random(32)
    ri = pow(R384, -1, P)
    inside = M.g2_mul(M.G2, rng.randrange(1, M.R))
    while True:
        x = (rng.randrange(P), rng.randrange(P))
        y = M.f2_sqrt(M.f2_add(M.f2_mul(M.f2_sqr(x), x), M.B2))
        if y is not None:
            outside = (x, y)
            break
    for pt, r in ((inside, rng.randrange(1, 1 << 64)), (outside, 0xF00000000000000F)):
        ws = {t.G2_SLOTS["SIG"] + i: c * R384 % P for i, c in enumerate((pt[0][0], pt[0][1], pt[1][0], pt[1][1]))}
        out, verdict = _blind_run("g2", r, ws, range(t.BL_OUT, t.BL_OUT + 6))
        assert verdict == M.subgroup_check_g2(pt)
        c = [w * ri % P for w in out]
        assert jac2_affine(M, (c[0], c[1]), (c[2], c[3]), (c[4], c[5])) == M.g2_mul(pt, r)


def test_g2_blinding_routine_constant_time_table_access():
    """the form signing uses (g2_blind_routine(ct=True), reference src/signature.rs:17-21 -- amcl's g2mul selects in constant time): every window reads all eight
    table records and keeps its own by selection. Same [r] P as the model (digits 0, +-8, +-1 and random ones), the selected record checked bit for bit per window,
    and no memory instruction of the routine takes a per-lane record offset."""
    M = _g2m()
    rng = random.Random(33)
    ri = pow(R384, -1, P)
    pt = M.g2_mul(M.G2, rng.randrange(1, M.R))
    ws = {t.G2_SLOTS["SIG"] + i: c * R384 % P for i, c in enumerate((pt[0][0], pt[0][1], pt[1][0], pt[1][1]))}
    for r in (rng.randrange(1, 1 << 64), 0x0807060504030201, 0xF00000000000000F):
        out, verdict = _blind_run("g2", r, dict(ws), range(t.BL_OUT, t.BL_OUT + 6), ct=True)
        assert verdict is True
        c = [w * ri % P for w in out]
        assert jac2_affine(M, (c[0], c[1]), (c[2], c[3]), (c[4], c[5])) == M.g2_mul(pt, r), hex(r)
