#!/usr/bin/env python3
"""Generate milagro_bls_amd/csrc/mbls_coop_prog.inc: microprograms for the WAVE-COOPERATIVE engine (mbls_coop.h).

The pipeline kernels give every item one lane, which is what throughput wants -- and what makes ONE item cost 20 ms however small the
batch is (a lane walks 14 M dependent instructions), and what makes the single final exponentiation of verify_multiple (reference
src/aggregates.rs:313-315) an 8 ms tail. The cooperative engine turns that around for the latency-bound cases: ONE item per WAVE, its
Fp values in LDS slots shared by the 64 lanes, and the computation cut into steps in which every lane performs one Fp operation of its
own on operands it picks from the slots:
    MUL   S[dst] <- (sum of up to 4 slots with small coefficients) * (sum of up to 4 slots) / 2^392      (mbls_fp_mul1_d_asm_fn: 461 instr.)
    LIN   S[dst] <- reduce(sum of up to 8 slots with small coefficients), or a flag-selected one of two 4-term sums
    INV / ISZ / FLG / LOADW / STOREW / RES   inversion, zero test -> flag, flag logic, workspace words in / out, result out
An Fp12 product is then three steps deep instead of 54 multiplications long: the final exponentiation takes ~0.9 M instruction
slots of ONE wave ... in ~1700 steps of ~500 instructions, i.e. under a millisecond instead of 8.

This file is the "compiler": the formulas are the ones of tools/gen_tower_d.py (same Prog layer: Fp2 / Fp6 / Fp12 methods, line
products, Frobenius constants), evaluated over LAZY LINEAR COMBINATIONS -- additions, subtractions, small multiples and
multiplications by xi cost nothing until a product or a store needs the value; an operand that has more than 4 terms, or whose digit
bound would overflow a product column, is materialised by a LIN step first. Bounds are tracked exactly as in gen_tower_d.py
(class Bound). A list scheduler packs the Fp operations of a block into steps (64 lanes each), slots are assigned by liveness, loop
bodies are blocks whose microcode is emitted once and referenced by every iteration.

Digit-exact simulation of the microcode: tools/coop_sim.py (tests/test_coop_cpu.py checks the programs against the big-integer model).
Run:  python3 tools/gen_coop.py     (output committed; tests check it is up to date)
"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_fp_asm import P, M28  # noqa: E402
from gen_fpd_asm import column_ok  # noqa: E402
import gen_tower_d as T  # noqa: E402
from gen_tower_d import Bound, product_bound, REDUCED, G_IN, R392, ONE_D, K384, D392  # noqa: E402

K_END, K_MUL, K_LIN, K_INV, K_ISZ, K_FLG, K_LOADW, K_STOREW, K_RES, K_POW, K_SGN = range(11)
KIND_NAMES = ["END", "MUL", "LIN", "INV", "ISZ", "FLG", "LOADW", "STOREW", "RES", "POW", "SGN"]
NLANE = 64
MAXT = 4                     # terms per product operand
SLOT_ZERO = 0                # slot 0 holds zero: unused terms point there (coefficient 0), idle lanes write there
FLAG_FALSE, FLAG_TRUE = 0, 1  # flag words 0 / 1 are constants
F_AND, F_OR, F_ANDN, F_XOR, F_ORN, F_ALL = range(6)        # d = a & b, a | b, a & ~b, a ^ b, a | ~b, AND of flags a .. a + b - 1


class LC:
    """lazy linear combination: {node: coefficient}"""
    __slots__ = ("t",)

    def __init__(self, t=None):
        self.t = {k: v for k, v in (t or {}).items() if v}

    def key(self):
        return tuple(sorted((n.id, c) for n, c in self.t.items()))

    def __add__(self, o):
        d = dict(self.t)
        for n, c in o.t.items():
            d[n] = d.get(n, 0) + c
        return LC(d)

    def __sub__(self, o):
        d = dict(self.t)
        for n, c in o.t.items():
            d[n] = d.get(n, 0) - c
        return LC(d)

    def scaled(self, k):
        return LC({n: c * k for n, c in self.t.items()})

    def bound(self):
        b = None
        for n, c in self.t.items():
            x = n.bound.scaled(c) if c > 0 else n.bound.scaled(-c).neg()
            b = x if b is None else b + x
        return b if b is not None else Bound(0, 0, 0, 0, 0, 0)

    def nterms(self):
        return len(self.t)

    def single(self):
        """the node if this is exactly 1 * node"""
        if len(self.t) == 1:
            (n, c), = self.t.items()
            if c == 1:
                return n
        return None


class Node:
    __slots__ = ("id", "kind", "a", "b", "bound", "slot", "step", "lane", "pin", "aux", "last_use", "flag")

    def __init__(self, nid, kind, bound, a=None, b=None, aux=None):
        self.id, self.kind, self.bound, self.a, self.b, self.aux = nid, kind, bound, a, b, aux
        self.slot = self.step = self.lane = self.pin = self.flag = None
        self.last_use = -1


# Product shapes per (program, block): (Fp2 products schoolbook instead of Karatsuba, Fp6 products schoolbook instead of Karatsuba). A wave
# has 64 product lanes a step and the formulas rarely fill them, while every operand that is a sum of more than four slots costs a LIN
# operation first: spending idle product lanes on schoolbook forms (operands are the inputs themselves, results have fewer terms) shortens
# the step lists. A shape may go on with (Fp12 squaring as a^2 + v b^2 and 2ab instead of the complex form, Fp12 products -- also by
# lines -- with four Fp6 products instead of three, the product of two lines with nine Fp2 products instead of six). The table is the result
# of a search (tools/coop_shapes.py: per block the cheapest step list, the program within the slot budget of eight waves per CU); blocks not
# listed use Karatsuba everywhere.
SHAPES = {
    ('f12mul', 'mul'): (1, 1),
    ('g2add', 'add'): (2,),
    ('hashg2', 'ladd'): (1,),
    ('hashg2', 'lsub'): (1,),
    ('hashg2', 'sswu'): (2,),
    ('hashg2x4', 'ladd'): (1,),
    ('hashg2x4', 'lsub'): (1,),
    ('hashg2x4', 'sswu'): (2,),
    ('miller1', 'add'): (2,),
    ('miller1', 'dbl'): (2, 0, 0, 1),
    ('pairing2', 'add'): (2,),
    ('pairing2', 'dbl'): (0, 1),
    ('pairing2', 'easy'): (2,),
    ('pairing2', 'mulbase'): (1, 1),
    ('pairing2', 'step_conj'): (1, 1),
    ('pairing2', 'step_frob'): (1, 1),
    ('pairing2', 'tail'): (2, 1),
    ('pairing2x2', 'add'): (1,),
    ('pairing2x2', 'dbl'): (1, 1, 1),
    ('pairing2x2', 'easy'): (0, 1, 0, 1),
    ('pairing2x2', 'step_frob'): (1,),
    ('pairing2x2', 'tail'): (2, 0, 0, 1),
    ('smiller', 'add'): (2,),
    ('smiller', 'dbl'): (2, 0, 0, 1),
    ('vmfinal', 'easy'): (2,),
    ('vmfinal', 'join'): (0, 1),
    ('vmfinal', 'mulbase'): (1, 1),
    ('vmfinal', 'step_conj'): (1, 1),
    ('vmfinal', 'step_frob'): (1, 1),
    ('vmfinal', 'tail'): (0, 1, 0, 1),    # 288 slots, the largest of all programs = the static LDS of every 64-lane instance: 8 waves per CU, two per SIMD. (With 275 slots a CU takes
                                          # 9 waves, one SIMD then holds three and sets the pace: 3 072 verifications 6.9 -> 7.8 ms. Measured, reverted.)
    ('vmtail', 'add'): (2,),
    ('vmtail', 'dbl'): (2,),
    ('vmtail', 'easy'): (2,),
    ('vmtail', 'join'): (0, 1),
    ('vmtail', 'mulbase'): (1, 1),
    ('vmtail', 'step_conj'): (1, 1),
    ('vmtail', 'step_frob'): (1, 1),
    ('vmtail', 'tail'): (2, 1),
}


class Block(T.Prog):
    """One block of a cooperative program: a DAG of Fp operations over lazy linear combinations, with the Fp2 / Fp6 / Fp12 layers of
    gen_tower_d.Prog on top (its primitives are overridden below). Live-in values are pinned slots of the program's state."""

    def __init__(self, machine, name):
        self.m, self.name = machine, name
        self.shape = (tuple(SHAPES.get((machine.name, name), ())) + (0, 0, 0, 0, 0))[:5]
        self.nodes = []
        self.mat_cache, self.mul_cache, self.const_cache = {}, {}, {}
        self.items = []                  # schedulable operations in creation order
        self.outs = []                   # (state name, node) written at the end of the block
        self.in_nodes = {}

    # ---- leaves
    def _node(self, kind, bound, a=None, b=None, aux=None):
        n = Node(len(self.nodes), kind, bound, a, b, aux)
        self.nodes.append(n)
        return n

    def inp(self, name, bound=REDUCED):
        """a state value (pinned slot), as it is when the block starts"""
        if name not in self.in_nodes:
            n = self._node("in", bound)
            n.pin = self.m.state_slot(name)
            self.in_nodes[name] = n
        return LC({self.in_nodes[name]: 1})

    def const(self, value):
        if value == 0:
            return LC()
        if value not in self.const_cache:
            n = self._node("in", Bound(0, M28, value >> 364, value >> 364, value, value))
            n.pin = self.m.const_slot(value)
            self.const_cache[value] = n
        return LC({self.const_cache[value]: 1})

    # ---- Prog primitives on lazy values
    def add(self, a, b): return a + b
    def sub(self, a, b): return a - b
    def neg(self, a): return a.scaled(-1)
    def scale(self, a, k): return a.scaled(k)
    def shadd(self, a, s, b): return a.scaled(1 << s) + b

    def pair(self, k0, a0, b0, k1, a1, b1):
        return ((a0 + b0) if k0 == "add" else (a0 - b0), (a1 + b1) if k1 == "add" else (a1 - b1))

    def materialise(self, a, force=False):
        """the value as ONE slot holding its reduced representative"""
        if not force and a.single() is not None and _is_reduced(a.single().bound):
            return a
        k = a.key()
        if k in self.mat_cache:
            return LC({self.mat_cache[k]: 1})
        terms = list(a.t.items())
        for i, (nd, c) in enumerate(terms):                  # a coefficient outside int8: 127 * (node) is materialised on its own
            if not -127 <= c <= 127:
                big = self.materialise(LC({nd: 127}), True).single()
                k_, r_ = divmod(abs(c), 127)
                sg = 1 if c > 0 else -1
                assert k_ <= 127
                terms[i] = (big, sg * k_)
                if r_:
                    terms.append((nd, sg * r_))
        while len(terms) > 2 * MAXT:                         # more than a LIN step takes: partial sums first, side by side (a tree, not a chain)
            groups = [terms[i:i + 2 * MAXT] for i in range(0, len(terms), 2 * MAXT)]
            terms = []
            for g in groups:
                terms += list(self.materialise(LC(dict(g)), True).t.items()) if len(g) > 1 else g
        lc = LC(dict(terms))
        B = lc.bound()
        # a LIN step sums in 64 bits, carries, estimates the quotient from the true top digit and subtracts: any digit bound below 2^50 works
        assert B.mag() < (1 << 50) and B.vabs() < (P << 30), ("cannot reduce", B)
        for n, c in lc.t.items():
            assert -128 <= c <= 127
        n = self._node("lin", REDUCED, a=lc)
        self.items.append(n)
        self.mat_cache[k] = n
        return LC({n: 1})

    def _operand(self, a):
        if a.nterms() > MAXT or any(not -128 <= c <= 127 for c in a.t.values()) or not a.bound().fits():
            a = self.materialise(a)
        return a

    def _sel_operand(self, a):
        if a.nterms() > MAXT or any(not -128 <= c <= 127 for c in a.t.values()):
            a = self.materialise(a)
        return a

    def mul(self, a, b):
        """one Fp product (Montgomery, radix 2^392) of two lazy values"""
        if not a.t or not b.t:
            return LC()
        a, b = self._operand(a), self._operand(b)
        guard = 0
        while not column_ok([(a.bound().mag(), b.bound().mag())]):
            if a.bound().mag() >= b.bound().mag() and not (a.single() is not None and _is_reduced(a.single().bound)):
                a = self.materialise(a)
            else:
                b = self.materialise(b)
            guard += 1
            assert guard < 4, "operands cannot be brought inside the column limit"
        k = (a.key(), b.key())
        if k not in self.mul_cache and (k[1], k[0]) in self.mul_cache:
            k = (k[1], k[0])
        if k not in self.mul_cache:
            n = self._node("mul", product_bound([(a.bound(), b.bound())]), a=a, b=b)
            self.items.append(n)
            self.mul_cache[k] = n
        return LC({self.mul_cache[k]: 1})

    def call(self, kind, ins):
        if kind == "mul":
            a0, a1, b0, b1 = ins
            if self.shape[0]:                                # schoolbook: 4 products on the operands themselves, two-term results (see SHAPES)
                return (self.mul(a0, b0) - self.mul(a1, b1), self.mul(a0, b1) + self.mul(a1, b0))
            t0, t1, t2 = self.mul(a0, b0), self.mul(a1, b1), self.mul(a0 + a1, b0 + b1)      # Karatsuba: 3 products
            return (t0 - t1, t2 - t0 - t1)
        if kind == "sqr":
            a0, a1 = ins
            if self.shape[0] == 2:                           # three products on the operands themselves
                return (self.mul(a0, a0) - self.mul(a1, a1), self.mul(a0, a1).scaled(2))
            return (self.mul(a0 + a1, a0 - a1), self.mul(a0, a1).scaled(2))
        if kind == "mulfp":
            a0, a1, s = ins
            return (self.mul(a0, s), self.mul(a1, s))
        if kind == "mulpair":
            a0, a1, b0, b1 = ins
            return (self.mul(a0, b0), self.mul(a1, b1))
        if kind == "sqrpair":
            return (self.mul(ins[0], ins[0]), self.mul(ins[1], ins[1]))
        if kind == "mul1":
            return (self.mul(ins[0], ins[1]),)
        raise ValueError(kind)

    def mulpair(self, a0, b0, a1, b1):
        return (self.mul(a0, b0), self.mul(a1, b1))

    def mul1(self, a, b):
        return self.mul(a, b)

    def mul2(self, a, b):
        return self.sqr2(a) if (a[0] is b[0] and a[1] is b[1]) else self.call("mul", [a[0], a[1], b[0], b[1]])

    def mul6(self, a, b):
        if not self.shape[1]:
            return T.Prog.mul6(self, a, b)                   # Karatsuba over Fp2: 6 products
        m = self.mul2                                        # schoolbook: 9 products, no operand sums
        c0 = self.add2(m(a[0], b[0]), self.mul_xi2(self.add2(m(a[1], b[2]), m(a[2], b[1]))))
        c1 = self.add2(self.add2(m(a[0], b[1]), m(a[1], b[0])), self.mul_xi2(m(a[2], b[2])))
        c2 = self.add2(self.add2(m(a[0], b[2]), m(a[1], b[1])), m(a[2], b[0]))
        return [c0, c1, c2]

    def sqr12(self, f):
        if not self.shape[2]:
            return T.Prog.sqr12(self, f)                     # complex squaring: 2 Fp6 products of sums
        a, b = f                                             # a^2 + v b^2, 2 a b: 3 Fp6 products of the coefficients themselves
        ab = self.mul6(a, b)
        return (self.add6(self.mul6(a, a), self.mul_v6(self.mul6(b, b))), self.add6(ab, ab))

    def mul12_line(self, f, c0, c2, c3):
        if not self.shape[3]:
            return T.Prog.mul12_line(self, f, c0, c2, c3)
        a, b = f                                             # four sparse products instead of three with a sum operand
        t0, t1 = self.mul6_01(a, c0, c2), self.mul6_1(b, c3)
        return (self.add6(t0, self.mul_v6(t1)), self.add6(self.mul6_1(a, c3), self.mul6_01(b, c0, c2)))

    def sel(self, flag, a, b):
        """flag ? b : a   (flag: a flag word index)"""
        if flag == FLAG_FALSE:
            return a
        if flag == FLAG_TRUE:
            return b
        a, b = self._sel_operand(a), self._sel_operand(b)
        n = self._node("sel", REDUCED, a=a, b=b, aux=flag)
        self.items.append(n)
        return LC({n: 1})

    def reduce(self, a):
        return self.materialise(a)

    def inv(self, a):
        """1 / a (0 -> 0): a * 2^384-domain words through the safegcd routine of tools/gen_fp_asm.py, like gen_tower_d.Prog.inv"""
        w = self.materialise(self.mul(a, self.const(K384)), True)
        n = self._node("inv", REDUCED, a=w)                  # the kernel reduces the converted words
        self.items.append(n)
        return LC({n: 1})

    def pow34(self, a):
        """a^((p-3)/4) by the fixed-exponent routine of tools/gen_fp_asm.py (same word interface as inv)"""
        w = self.materialise(self.mul(a, self.const(K384)), True)
        n = self._node("pow", REDUCED, a=w)
        self.items.append(n)
        return LC({n: 1})

    def sgn0_2(self, a, flag, tmp):
        """flag <- sgn0 of the Fp2 value a (RFC 9380 section 4.1: parity of the plain representative of c0, or of c1 when c0 = 0): the
        plain integers come out of a Montgomery product with the constant whose representation is 1"""
        one = self.const(1)
        x0 = self.materialise(self.mul(a[0], one), True)
        x1 = self.materialise(self.mul(a[1], one), True)
        n = self._node("sgn", None, a=x0, b=x1, aux=flag)
        self.items.append(n)

    def mask_xor(self, dst, a, b):
        self.flagop(F_XOR, dst, a, b)

    def mask_and(self, dst, a, b):
        self.flagop(F_AND, dst, a, b)

    def iszero(self, a, flag):
        """flag word <- (a = 0 mod p)"""
        w = self.materialise(a)
        n = self._node("isz", None, a=w, aux=flag)
        self.items.append(n)

    def iszero2(self, a, flag, tmp):
        self.iszero(a[0], flag); self.iszero(a[1], tmp); self.flagop(F_AND, flag, flag, tmp)

    def flagop(self, op, dst, a, b):
        n = self._node("flg", None, aux=(op, dst, a, b))
        self.items.append(n)

    def out(self, name, a):
        """state[name] <- a (reduced) when the block ends"""
        fresh = a.single()
        if fresh is not None and fresh.kind in ("lin", "sel", "inv", "pow") and fresh.pin is None and _is_reduced(fresh.bound) and not any(fresh is o for _, o in self.outs):
            w = a                                                # a reduced value of its own that lives nowhere yet: it becomes the state slot's content
        else:
            w = self.materialise(a, True)
        node = w.single()
        if node.pin is not None or any(node is o for _, o in self.outs):       # one node cannot live in two pinned slots: a fresh copy
            node = self._node("lin", REDUCED, a=LC({node: 1}))
            self.items.append(node)
        node.pin = self.m.state_slot(name)
        self.outs.append((name, node))

    def loadw(self, name, ws_slot, partner=False):
        """state[name] <- the 12 words of workspace slot ws_slot of this wave's item (partner: of the item partner_step further on)
        (2^384 domain) as 2^392-domain digits, reduced"""
        n = self._node("loadw", REDUCED, aux=ws_slot | (0x10000 if partner else 0))
        n.pin = self.m.state_slot(name)
        self.items.append(n)
        self.in_nodes[name] = n
        return LC({n: 1})

    def storew(self, a, ws_slot):
        """workspace slot <- canonical 2^384-domain words of a"""
        w = self.materialise(self.mul(a, self.const(K384)), True)
        n = self._node("storew", None, a=w, aux=ws_slot)
        self.items.append(n)

    def result(self, flag_ok, flag_skip=FLAG_FALSE):
        n = self._node("res", None, aux=(flag_ok, flag_skip))
        self.items.append(n)


def _is_reduced(B):
    return B is not None and B.dlo >= 0 and B.dhi <= M28 and B.vlo >= REDUCED.vlo and B.vhi <= REDUCED.vhi


class Machine:
    """slot and flag numbering of one program; blocks are scheduled against it"""

    def __init__(self, name, n_slots=600, nlane=NLANE):
        self.name = name
        self.nlane = nlane               # lanes of a step one item may use: 64 (one item per wave), 32 or 16 (two / four items per wave)
        self.state = {}                  # name -> slot
        self.consts = {}                 # value -> slot
        self.next_slot = 1               # slot 0 = zero
        self.flags = {"false": FLAG_FALSE, "true": FLAG_TRUE}
        self.blocks = {}
        self.order = []                  # [(block name, repeat)]
        self.n_slots = n_slots

    def state_slot(self, name):
        if name not in self.state:
            self.state[name] = self.next_slot; self.next_slot += 1
        return self.state[name]

    def const_slot(self, value):
        if value not in self.consts:
            self.consts[value] = self.next_slot; self.next_slot += 1
        return self.consts[value]

    def flag(self, name):
        if name not in self.flags:
            self.flags[name] = len(self.flags)
        return self.flags[name]

    def block(self, name):
        b = Block(self, name)
        self.blocks[name] = b
        return b

    def run(self, name, repeat=1):
        self.order.append((name, repeat))


# ---------------------------------------------------------------------------------------------- scheduling
BY_HEIGHT = os.environ.get("MBLS_COOP_BY_HEIGHT", "1") == "1"


class Step:
    def __init__(self, kind):
        self.kind = kind
        self.lanes = []                  # nodes


def node_inputs(n):
    out = []
    for lc in (n.a, n.b):
        if isinstance(lc, LC):
            out += list(lc.t.keys())
    return out


STEP_KIND = {"mul": K_MUL, "lin": K_LIN, "sel": K_LIN, "inv": K_INV, "isz": K_ISZ, "flg": K_FLG, "loadw": K_LOADW, "storew": K_STOREW, "res": K_RES,
             "pow": K_POW, "sgn": K_SGN}


def schedule(block):
    """list scheduling in creation order (a topological order): every operation goes into the earliest step of its kind after its
    operands. Flag producers / consumers and stores to the state keep their program order through explicit ready times."""
    steps = []
    flag_ready = {}                      # flag -> first step index at which it can be read
    flag_last_read = {}
    last_read = {}                       # node -> last step that reads it
    pin_out = {node: name for name, node in block.outs}
    for n in block.items:
        kind = STEP_KIND[n.kind]
        ready = 0
        for x in node_inputs(n):
            if x.kind != "in":
                ready = max(ready, x.step + 1)
        if n.kind == "sel":
            ready = max(ready, flag_ready.get(n.aux, 0))
        if n.kind in ("isz", "sgn"):
            ready = max(ready, flag_last_read.get(n.aux, -1) + 0, flag_ready.get(n.aux, 0))
        if n.kind == "flg":
            op, d, a, b = n.aux
            srcs = list(range(a, a + b)) if op == F_ALL else [a, b]
            for f in srcs:
                ready = max(ready, flag_ready.get(f, 0))
            # every lane of a step reads its flags before any lane writes: an earlier reader of d may share the step, an earlier writer may not
            ready = max(ready, flag_last_read.get(d, -1), flag_ready.get(d, 0))
        if n.kind == "res":
            ready = max([ready, len(steps)] + [flag_ready.get(f, 0) for f in n.aux])
        if n in pin_out or n.kind == "loadw":
            # the write of a state slot comes after every read of the value the slot held when the block started (same step is fine: a
            # step reads all its operands before it writes)
            old = block.in_nodes.get(pin_out.get(n))
            if old is not None and old is not n:
                ready = max(ready, last_read.get(old, -1))
            for other in block.nodes:                        # ... and of any other live-in that shares the slot (it does not happen: names are unique)
                pass
        if n.kind in ("storew", "res"):
            ready = max(ready, 0)
        # earliest step of this kind with a free lane; MUL / LIN steps take 64 operations, the serial kinds as well
        k = None
        for i in range(ready, len(steps)):
            if steps[i].kind == kind and len(steps[i].lanes) < block.m.nlane:
                k = i
                break
        if k is None:
            steps.append(Step(kind)); k = len(steps) - 1
        n.step, n.lane = k, len(steps[k].lanes)
        steps[k].lanes.append(n)
        for x in node_inputs(n):
            last_read[x] = max(last_read.get(x, -1), k)
        if n.kind == "sel":
            flag_last_read[n.aux] = max(flag_last_read.get(n.aux, -1), k)
        if n.kind in ("isz", "sgn"):
            flag_ready[n.aux] = k + 1
        if n.kind == "flg":
            op, d, a, b = n.aux
            for f in (list(range(a, a + b)) if op == F_ALL else [a, b]):
                flag_last_read[f] = max(flag_last_read.get(f, -1), k)
            flag_ready[d] = k + 1
        if n.kind == "res":
            for f in n.aux:
                flag_last_read[f] = max(flag_last_read.get(f, -1), k)
    # a state write must not land before a LATER-scheduled read of the old value: verify (creation order makes this hold; assert it)
    for name, node in block.outs:
        old = block.in_nodes.get(name)
        if old is not None and old is not node:
            assert last_read.get(old, -1) <= node.step, ("state slot rewritten before its last read", block.name, name)
    for n in block.nodes:
        n.last_use = last_read.get(n, -1)
    return steps


def schedule_by_height(block):
    """The same packing problem solved step by step for blocks of nothing but products and linear combinations (the loop bodies): at every
    step the kind of the most urgent ready operation is chosen -- urgency = the longest chain of operations that still hangs on it -- and the
    step is filled with the ready operations of that kind in that order. Creation-order scheduling (schedule) serves whichever chain was
    written first and can leave the block's critical chain waiting for lanes. A state slot is written no earlier than the last read of the
    value it held when the block started (a step reads before it writes: the same step is fine)."""
    items = list(block.items)
    if any(n.kind not in ("mul", "lin") for n in items):
        return None
    pos = {n: i for i, n in enumerate(items)}
    preds = {n: [x for x in node_inputs(n) if x.kind != "in"] for n in items}
    succs = {n: [] for n in items}
    for n in items:
        for x in preds[n]:
            if x in succs:
                succs[x].append(n)
    readers = {}
    for n in items:
        for x in node_inputs(n):
            if x.kind == "in":
                readers.setdefault(x, []).append(n)
    after = {n: [] for n in items}                     # zero-latency predecessors: the readers of the state value a node overwrites
    for name, node in block.outs:
        old = block.in_nodes.get(name)
        if old is not None and old is not node and node in after:
            after[node] += [r for r in readers.get(old, []) if r is not node]
    height = {}
    for n in reversed(items):                          # creation order is topological
        height[n] = 1 + max([height[x] for x in succs[n]] + [0])
    for n in items:                                    # a writer's urgency carries over to the readers it waits for
        for r in after[n]:
            height[r] = max(height[r], height[n])
    done, steps = {}, []
    left = set(items)
    while left:
        t = len(steps)
        def ready(n):
            return all(x in done and done[x] < t for x in preds[n]) and all(r in done for r in after[n])
        cand = [n for n in left if ready(n)]
        assert cand, "scheduling deadlock"
        cand.sort(key=lambda n: (-height[n], pos[n]))
        kind = STEP_KIND[cand[0].kind]
        st = Step(kind)
        while len(st.lanes) < block.m.nlane:
            pick = [n for n in left if STEP_KIND[n.kind] == kind and ready(n)]
            if not pick:
                break
            pick.sort(key=lambda n: (-height[n], pos[n]))
            for n in pick[:block.m.nlane - len(st.lanes)]:
                n.step, n.lane = t, len(st.lanes)
                st.lanes.append(n); done[n] = t; left.discard(n)
        steps.append(st)
    last_read = {}
    for n in items:
        for x in node_inputs(n):
            last_read[x] = max(last_read.get(x, -1), n.step)
    for name, node in block.outs:
        old = block.in_nodes.get(name)
        if old is not None and old is not node:
            assert last_read.get(old, -1) <= node.step, ("state slot rewritten before its last read", block.name, name)
    for n in block.nodes:
        n.last_use = last_read.get(n, -1)
    return steps


def assign_slots(machine, block, steps, temp_base):
    """temporaries by liveness from temp_base upwards (a slot read for the last time in step t may be written in step t)"""
    free, nxt, peak = [], temp_base, temp_base
    release_at = {}
    for si, st in enumerate(steps):
        for n in release_at.pop(si, []):
            free.append(n.slot)
        for n in st.lanes:
            if n.kind in ("isz", "flg", "storew", "res", "sgn"):
                continue
            if n.pin is not None:
                n.slot = n.pin
                continue
            if free:
                n.slot = free.pop()
            else:
                n.slot = nxt; nxt += 1
            peak = max(peak, nxt)
            lu = n.last_use if n.last_use >= 0 else si
            release_at.setdefault(max(lu, si + 1), []).append(n)
    for n in block.nodes:
        if n.kind == "in":
            n.slot = n.pin
    assert peak <= machine.n_slots, ("out of LDS slots", peak)
    return peak


# ---------------------------------------------------------------------------------------------- microcode
# per lane and step: 8 dwords
#   w0, w1: eight signed 8-bit coefficients (terms 0..3 = operand A / first sum, 4..7 = operand B / second sum)
#   w2, w3, w4: nine 10-bit slot indices (3 per dword): the eight terms, then dst
#   w5: bits 0..7 flag index (sel: flag ? second sum : first sum; isz: flag to write; res: ok flag), bits 8..15 second flag, bits 16..23
#       flag op, bit 31 lane is active
#   w6: workspace slot (loadw / storew), w7: reserved
def pack_lane(coefs, idxs, dst, flag=0, flag2=0, fop=0, active=True, wslot=0):
    assert len(coefs) == 8 and len(idxs) == 8
    w0 = sum((c & 0xFF) << (8 * i) for i, c in enumerate(coefs[:4]))
    w1 = sum((c & 0xFF) << (8 * i) for i, c in enumerate(coefs[4:]))
    ii = list(idxs) + [dst]
    assert all(0 <= x < 1024 for x in ii)
    w2 = ii[0] | (ii[1] << 10) | (ii[2] << 20)
    w3 = ii[3] | (ii[4] << 10) | (ii[5] << 20)
    w4 = ii[6] | (ii[7] << 10) | (ii[8] << 20)
    w5 = (flag & 0xFF) | ((flag2 & 0xFF) << 8) | ((fop & 0xFF) << 16) | ((1 << 31) if active else 0)
    return [w0, w1, w2, w3, w4, w5, wslot, 0]


IDLE_LANE = pack_lane([0] * 8, [0] * 8, SLOT_ZERO, active=False)


def lane_words(n):
    def terms(lc, cnt):
        t = [(x.slot, c) for x, c in lc.t.items()] if lc is not None else []
        assert len(t) <= cnt
        t += [(SLOT_ZERO, 0)] * (cnt - len(t))
        return t
    if n.kind == "mul":
        t = terms(n.a, 4) + terms(n.b, 4)
        return pack_lane([c for _, c in t], [s for s, _ in t], n.slot)
    if n.kind == "lin":
        t = terms(n.a, 8)
        return pack_lane([c for _, c in t], [s for s, _ in t], n.slot)
    if n.kind == "sel":
        t = terms(n.a, 4) + terms(n.b, 4)
        return pack_lane([c for _, c in t], [s for s, _ in t], n.slot, flag=n.aux, fop=1)
    if n.kind in ("inv", "pow"):
        return pack_lane([1] + [0] * 7, [n.a.single().slot] + [0] * 7, n.slot)
    if n.kind == "sgn":
        return pack_lane([1, 1] + [0] * 6, [n.a.single().slot, n.b.single().slot] + [0] * 6, SLOT_ZERO, flag=n.aux)
    if n.kind == "isz":
        return pack_lane([1] + [0] * 7, [n.a.single().slot] + [0] * 7, SLOT_ZERO, flag=n.aux)
    if n.kind == "flg":
        op, d, a, b = n.aux
        return pack_lane([0] * 8, [0] * 8, SLOT_ZERO, flag=d, flag2=a, fop=op, wslot=b)
    if n.kind == "loadw":
        return pack_lane([0] * 8, [0] * 8, n.slot, wslot=n.aux)
    if n.kind == "storew":
        return pack_lane([1] + [0] * 7, [n.a.single().slot] + [0] * 7, SLOT_ZERO, wslot=n.aux)
    if n.kind == "res":
        return pack_lane([0] * 8, [0] * 8, SLOT_ZERO, flag=n.aux[0], flag2=n.aux[1])
    raise ValueError(n.kind)


def step_info(st):
    """descriptor word: kind | nA << 8 | nB << 12 (the largest term counts over the lanes: the kernel skips the rest)"""
    na = nb = 0
    for n in st.lanes:
        if n.kind == "mul":
            na, nb = max(na, n.a.nterms()), max(nb, n.b.nterms())
        elif n.kind == "lin":
            t = n.a.nterms()
            na, nb = max(na, min(t, 4)), max(nb, max(0, t - 4))
        elif n.kind == "sel":
            na, nb = max(na, n.a.nterms()), max(nb, n.b.nterms())
        elif n.kind in ("inv", "isz", "storew", "pow"):
            na = max(na, 1)
        elif n.kind == "sgn":
            na = max(na, 2)
    return st.kind | (na << 8) | (nb << 12)


def compile_program(machine):
    """-> dict(steps=[(info, microcode row)], rows=[[64 x 8 words]], consts={slot: value}, n_slots, n_flags, stats)"""
    temp_base = None
    sched = {}
    # pinned slots (state + constants) are numbered while the blocks are built; temporaries live above all of them
    for name, blk in machine.blocks.items():
        sched[name] = schedule(blk)
    temp_base = machine.next_slot
    peak = temp_base
    for name, blk in machine.blocks.items():
        peak = max(peak, assign_slots(machine, blk, sched[name], temp_base))
    if BY_HEIGHT:       # a block takes the other schedule when that is shorter and needs no more LDS slots than the program already has
        for name, blk in machine.blocks.items():
            alt = schedule_by_height(blk)
            if alt is not None and len(alt) < len(sched[name]) and assign_slots(machine, blk, alt, temp_base) <= peak:
                sched[name] = alt
            else:
                sched[name] = schedule(blk)              # (the nodes carry their step / lane / last use / slot: assign them again)
                assign_slots(machine, blk, sched[name], temp_base)
    rows, row_of = [], {}
    block_rows = {}
    for name, blk in machine.blocks.items():
        br = []
        for st in sched[name]:
            words = []
            for n in st.lanes:
                words += lane_words(n)
            for _ in range(NLANE - len(st.lanes)):
                words += IDLE_LANE
            key = (step_info(st), tuple(words))
            if key not in row_of:
                row_of[key] = len(rows); rows.append(words)
            br.append((step_info(st), row_of[key]))
        block_rows[name] = br
    steps = []
    for name, rep in machine.order:
        for _ in range(rep):
            steps += block_rows[name]
    steps.append((K_END, 0))
    stats = {name: dict(steps=len(sched[name]), mul=sum(1 for s in sched[name] if s.kind == K_MUL), lin=sum(1 for s in sched[name] if s.kind == K_LIN),
                        ops=len(machine.blocks[name].items)) for name in machine.blocks}
    return dict(nlane=machine.nlane, steps=steps, rows=rows, consts=dict((s, v) for v, s in machine.consts.items()), n_slots=peak, n_flags=len(machine.flags),
                stats=stats, total_steps=len(steps), sched=sched, block_rows=block_rows)


# ---------------------------------------------------------------------------------------------- the programs
RUNS = T.RUNS                                   # doublings between the additions of the Miller loop / squarings between the products of a power by |x|
F_NAMES = ["f%d" % i for i in range(12)]
WS_APK, WS_SIG, WS_H, WS_F, WS_S = 0, 3, 7, 13, 25      # workspace slots of the pipeline (mbls_lanes.h, mbls_kernels.hip)
NPX0_D, PY0_D = T.NPX0_D, T.PY0_D


def six(l):
    return T.six(l)


def flat12(f):
    return T.flat12(f)


def st12(b, f, prefix="f"):
    for i, v in enumerate(flat12(f)):
        b.out("%s%d" % (prefix, i), v)


def ld12(b, prefix="f"):
    return six([b.inp("%s%d" % (prefix, i)) for i in range(12)])


def pt_in(b, prefix):
    l = [b.inp("%s%d" % (prefix, i)) for i in range(6)]
    return [(l[0], l[1]), (l[2], l[3]), (l[4], l[5])]


def pt_out(b, prefix, pt):
    for i, v in enumerate([x for c in pt for x in c]):
        b.out("%s%d" % (prefix, i), v)


def masked_p(b, m, skip, npx, py, pz3, prefix):
    """A pair with a member at infinity contributes 1: its lines must be (1, 0, 0). No selection inside the loop: the G1 argument is
    masked ONCE -- (-px, py, pz3) <- 0 and `one` <- 1 on a skipped pair -- and every line is (c0 pz3 + one, c2 (-px), c3 py)."""
    one, zero = b.const(ONE_D), LC()
    b.out(prefix + "npx", b.sel(skip, npx, zero)); b.out(prefix + "py", b.sel(skip, py, zero))
    b.out(prefix + "pz3", b.sel(skip, pz3, zero)); b.out(prefix + "one", b.sel(skip, zero, one))


def dbl_step(b, Tn, pxyz):
    """T <- 2T and the line through it (formulas of prog_miller_dbl_d in tools/gen_tower_d.py / miller_dbl_step in mbls_pairing.h);
    pxyz = (-px, py, pz3, one): the masked G1 argument (masked_p)"""
    Tx, Ty, Tz = Tn
    B = b.sqr2(Ty); C = b.sqr2(Tz)
    E = b.mul12_2(b.mul_xi2(C))
    F = b.mul3_2(E)
    X2 = b.sqr2(Tx)
    YZ2 = b.sub2(b.sub2(b.sqr2(b.add2(Ty, Tz)), B), C)
    c0 = b.mulfp2(b.sub2(B, E), pxyz[2])
    c0 = (b.add(c0[0], pxyz[3]), c0[1])
    c2 = b.mulfp2(b.mul3_2(X2), pxyz[0])
    c3 = b.mulfp2(YZ2, pxyz[1])
    x3 = b.dbl2(b.mul2(b.mul2(Tx, Ty), b.sub2(B, F)))
    y3 = b.sub2(b.sqr2(b.add2(B, F)), b.mul12_2(b.sqr2(E)))
    z3 = b.mul4_2(b.mul2(B, YZ2))
    return [x3, y3, z3], (c0, c2, c3)


def add_step(b, Tn, Q, pxyz):
    """T <- T + Q and the line through them (prog_miller_add_d / miller_add_step); Q = (x, y, z or None for an affine point)"""
    Tx, Ty, Tz = Tn
    Qx, Qy, Qz = Q
    if Qz is None:
        y1z2, x1z2, z1z2 = Ty, Tx, Tz
    else:
        y1z2, x1z2, z1z2 = b.mul2(Ty, Qz), b.mul2(Tx, Qz), b.mul2(Tz, Qz)
    u = b.sub2(b.mul2(Qy, Tz), y1z2)
    v = b.sub2(b.mul2(Qx, Tz), x1z2)
    c0 = b.mulfp2(b.sub2(b.mul2(u, Qx), b.mul2(v, Qy)), pxyz[2])
    c0 = (b.add(c0[0], pxyz[3]), c0[1])
    uq, vq = (u, v) if Qz is None else (b.mul2(u, Qz), b.mul2(v, Qz))
    c2 = b.mulfp2(uq, pxyz[0])
    c3 = b.mulfp2(vq, pxyz[1])
    uu = b.sqr2(u); vv = b.sqr2(v)
    vvv = b.mul2(v, vv); R = b.mul2(vv, x1z2)
    A = b.sub2(b.sub2(b.mul2(uu, z1z2), vvv), b.dbl2(R))
    x3 = b.mul2(v, A)
    y3 = b.sub2(b.mul2(u, b.sub2(R, A)), b.mul2(vvv, y1z2))
    z3 = b.mul2(vvv, z1z2)
    return [x3, y3, z3], (c0, c2, c3)


def mul_lines(b, la, lb):
    """T.mul_lines, or (shape[4]) the nine plain products"""
    if not b.shape[4]:
        return T.mul_lines(b, la, lb)
    (a0, a2, a3), (b0, b2, b3) = la, lb
    m = b.mul2
    return ([b.add2(m(a0, b0), b.mul_xi2(m(a3, b3))), b.add2(m(a0, b2), m(a2, b0)), m(a2, b2)],
            [None, b.add2(m(a0, b3), m(a3, b0)), b.add2(m(a2, b3), m(a3, b2))])


def mul12_by_lines(b, f, L0, L1):
    """T.mul12_by_lines, or (shape[3]) four products without the sum operands"""
    if not b.shape[3]:
        return T.mul12_by_lines(b, f, L0, L1)
    x, y = f
    t0 = b.mul6(x, L0)
    t1 = T.mul6_by_0yz(b, y, L1[1], L1[2])
    return (b.add6(t0, b.mul_v6(t1)), b.add6(T.mul6_by_0yz(b, x, L1[1], L1[2]), b.mul6(y, L0)))


def mul12(b, x, y, conj_b=False):
    """T.mul12, or (shape[3]) the four Fp6 products"""
    if not b.shape[3]:
        return T.mul12(b, x, y, conj_b)
    t0, t1 = b.mul6(x[0], y[0]), b.mul6(x[1], y[1])
    cross = b.add6(b.mul6(x[0], y[1]), b.mul6(x[1], y[0])) if not conj_b else b.sub6(b.mul6(x[1], y[0]), b.mul6(x[0], y[1]))
    return ((b.add6 if not conj_b else b.sub6)(t0, b.mul_v6(t1)), cross)


def build_miller_plain(m, pairs, prefix=""):
    """blocks `dbl` and `add` of the Miller loop over the listed pairs, every block multiplying its own lines into f"""
    def parg(b, pr):
        return [b.inp(pr["P"] + x) for x in ("npx", "py", "pz3", "one")]
    b = m.block(prefix + "dbl")
    f = b.sqr12(ld12(b))
    lines = []
    for pr in pairs:
        Tn, line = dbl_step(b, pt_in(b, pr["T"]), parg(b, pr))
        pt_out(b, pr["T"], Tn)
        lines.append(line)
    if len(lines) == 2:
        L0, L1 = mul_lines(b, lines[0], lines[1])
        f = mul12_by_lines(b, f, L0, L1)
    else:
        f = b.mul12_line(f, *lines[0])
    st12(b, f)
    b = m.block(prefix + "add")
    f = ld12(b)
    for pr in pairs:
        q = [b.inp("%s%d" % (pr["Q"], i)) for i in range(4 if pr["affine"] else 6)]
        Q = [(q[0], q[1]), (q[2], q[3]), None if pr["affine"] else (q[4], q[5])]
        Tn, line = add_step(b, pt_in(b, pr["T"]), Q, parg(b, pr))
        pt_out(b, pr["T"], Tn)
        f = b.mul12_line(f, *line)
    st12(b, f)


def build_miller(m, pairs, prefix="", pipelined=None):
    """blocks `mlinit`, `dbl`, `add`, `flush` of the Miller loop over the listed pairs. Pair descriptor: dict(T=state prefix of the running
    point, Q=state prefix of the fixed point, affine=bool, P=state prefix of the masked G1 argument (masked_p)).
    The loop is software-pipelined by one block: the lines a block computes are NOT multiplied into f by that block but left as the pending
    factor Lp (the merged lines of two pairs, or the one line), and every block starts with g = f * Lp -- the factor of the block before.
    The state f is therefore the Miller value without its latest lines (F = f * Lp): a doubling block leaves f <- g^2, an addition block
    f <- g, `flush` after the last block multiplies the last lines in. The chain through the running point (doubling -> line coefficients ->
    merged lines: six steps) and the chain through f (f * Lp -> square: four steps) then run side by side instead of one after the other
    (eleven steps a doubling block before)."""
    two = len(pairs) == 2
    npend = 10 if two else 6
    if pipelined is None:      # measured in steps: two pairs on 64 lanes 1558 -> 1428; one pair (8 steps a doubling either way) and the 32-lane form lose
        pipelined = two and m.nlane == NLANE
    m.miller_pipelined = pipelined
    if not pipelined:
        return build_miller_plain(m, pairs, prefix)

    def parg(b, pr):
        return [b.inp(pr["P"] + x) for x in ("npx", "py", "pz3", "one")]

    def pend_in(b):
        l = [b.inp("%slp%d" % (prefix, i)) for i in range(npend)]
        if two:
            return [(l[0], l[1]), (l[2], l[3]), (l[4], l[5])], [None, (l[6], l[7]), (l[8], l[9])]
        return [(l[0], l[1]), (l[2], l[3]), (l[4], l[5])]

    def pend_out(b, L):
        vals = [x for c in (L[0] + L[1][1:] if two else L) for x in c]
        assert len(vals) == npend
        for i, v in enumerate(vals):
            b.out("%slp%d" % (prefix, i), v)

    def times_pending(b, f, L=None):
        L = L or pend_in(b)
        return mul12_by_lines(b, f, L[0], L[1]) if two else b.mul12_line(f, *L)

    def merged(b, lines):
        return mul_lines(b, lines[0], lines[1]) if two else lines[0]

    b = m.block(prefix + "mlinit")                                # Lp <- 1
    one = b.const(ONE_D)
    for i in range(npend):
        b.out("%slp%d" % (prefix, i), one if i == 0 else LC())
    # the operations of the longer chain (the points and their lines) are created first: the list scheduler serves them first
    b = m.block(prefix + "dbl")
    Lp = pend_in(b)
    lines = []
    for pr in pairs:
        Tn, line = dbl_step(b, pt_in(b, pr["T"]), parg(b, pr))
        pt_out(b, pr["T"], Tn)
        lines.append(line)
    Ln = merged(b, lines)
    f = b.sqr12(times_pending(b, ld12(b), Lp))
    pend_out(b, Ln)
    st12(b, f)
    b = m.block(prefix + "add")
    Lp = pend_in(b)
    lines = []
    for pr in pairs:
        q = [b.inp("%s%d" % (pr["Q"], i)) for i in range(4 if pr["affine"] else 6)]
        Q = [(q[0], q[1]), (q[2], q[3]), None if pr["affine"] else (q[4], q[5])]
        Tn, line = add_step(b, pt_in(b, pr["T"]), Q, parg(b, pr))
        pt_out(b, pr["T"], Tn)
        lines.append(line)
    Ln = merged(b, lines)
    f = times_pending(b, ld12(b), Lp)
    pend_out(b, Ln)
    st12(b, f)
    b = m.block(prefix + "flush")
    st12(b, times_pending(b, ld12(b)))


def run_miller(m, prefix=""):
    if m.miller_pipelined:
        m.run(prefix + "mlinit")
    for ph in range(6):
        m.run(prefix + "dbl", RUNS[ph])
        if ph < 5:
            m.run(prefix + "add")
    if m.miller_pipelined:
        m.run(prefix + "flush")


def cyc_sqr_block(m, name="csq"):
    b = m.block(name)
    z = ld12(b)
    zz = z[0] + z[1]
    outv = [None] * 6
    T.cyc_sqr_formula(b, zz, lambda e, v: outv.__setitem__(e, v))
    st12(b, (outv[:3], outv[3:]))


def build_final_exp(m):
    """f (state f0..f11) <- f^(3 (p^12 - 1) / r), the sequence of final_exp in mbls_pairing.h / final_exp_d_routine in gen_tower_d.py, with
    plain Granger-Scott squarings (all 18 products of a squaring sit in ONE step here: compression would only add the decompressions).
    State: f (running value), y (base of the running power), mm (f after the easy part), bb."""
    b = m.block("easy")
    f = ld12(b)
    a0, a1 = f
    t = b.sub6(T.sqr6(b, a0), b.mul_v6(T.sqr6(b, a1)))
    c0, c1, c2 = t
    A = b.sub2(b.sqr2(c0), b.mul_xi2(b.mul2(c1, c2)))
    Bv = b.sub2(b.mul_xi2(b.sqr2(c2)), b.mul2(c0, c1))
    C = b.sub2(b.sqr2(c1), b.mul2(c0, c2))
    F = b.add2(b.mul_xi2(b.add2(b.mul2(c2, Bv), b.mul2(c1, C))), b.mul2(c0, A))
    n = b.add(b.mul(F[0], F[0]), b.mul(F[1], F[1]))
    ni = b.inv(n)
    Fi = b.mulfp2(b.conj2(F), ni)
    Tt = [b.mul2(A, Fi), b.mul2(Bv, Fi), b.mul2(C, Fi)]
    fi = (b.mul6(a0, Tt), b.neg6(b.mul6(a1, Tt)))
    t = mul12(b, fi, f, conj_b=True)                          # f^(p^6 - 1)
    mm = mul12(b, T.frob12_2(b, t), t)                        # ^(p^2 + 1)
    st12(b, mm); st12(b, mm, "y"); st12(b, mm, "m")
    cyc_sqr_block(m)
    b = m.block("mulbase")                                       # f <- f * y
    st12(b, mul12(b, ld12(b), ld12(b, "y")))
    b = m.block("pstart")                                        # f <- y
    st12(b, ld12(b, "y"))
    b = m.block("step_conj")                                     # after powers 1, 2: f <- conj(f) conj(y) = conj(f y); also the next base
    r = mul12(b, ld12(b), ld12(b, "y"))
    r = (r[0], b.neg6(r[1]))
    st12(b, r); st12(b, r, "y")
    b = m.block("step_frob")                                     # after power 3: b = conj(f) frob(y) -> f, y, bb
    r = mul12(b, T.frob12(b, ld12(b, "y")), ld12(b), conj_b=True)
    st12(b, r); st12(b, r, "y"); st12(b, r, "b")
    b = m.block("step_base")                                     # after power 4: conj(f) is the base of the fifth
    a = ld12(b)
    r = (a[0], b.neg6(a[1]))
    st12(b, r); st12(b, r, "y")
    b = m.block("tail")                                          # conj(f) * frob^2(b) * conj(b) * m^3
    a = ld12(b); bb = ld12(b, "b"); mmv = ld12(b, "m")
    c = mul12(b, T.frob12_2(b, bb), a, conj_b=True)
    c = mul12(b, c, bb, conj_b=True)
    m2 = [None] * 6
    T.cyc_sqr_formula(b, mmv[0] + mmv[1], lambda e, v: m2.__setitem__(e, v))
    c = mul12(b, c, mul12(b, (m2[:3], m2[3:]), mmv))
    st12(b, c)


def run_power(m):
    """f <- y^|x| (|x| = 2^63 + 2^62 + 2^60 + 2^57 + 2^48 + 2^16): start from y, squarings with a product by y after runs of 1, 2, 3, 9, 32"""
    m.run("pstart")
    for ph in range(6):
        m.run("csq", RUNS[ph])
        if ph < 5:
            m.run("mulbase")


def run_final_exp(m):
    m.run("easy")
    for k in range(5):
        run_power(m)
        if k < 2:
            m.run("step_conj")
        elif k == 2:
            m.run("step_frob")
        elif k == 3:
            m.run("step_base")
    m.run("tail")


def is_one_block(m, name, ok_flag, extra_ok=None):
    """ok_flag <- (f == 1) [& extra flag]"""
    b = m.block(name)
    f = flat12(ld12(b))
    fl = [m.flag("z%d" % i) for i in range(12)]
    assert fl == list(range(fl[0], fl[0] + 12))
    b.iszero(b.sub(f[0], b.const(ONE_D)), fl[0])
    for i in range(1, 12):
        b.iszero(f[i], fl[i])
    b.flagop(F_ALL, ok_flag, fl[0], 12)
    return b


def prog_pairing2(name="pairing2", nlane=NLANE):
    """The pairing check of ONE verification (Signature::verify / fast_aggregate_verify, reference src/amcl_utils.rs:38-42) on a wave:
    pair 0 = (signature, -G1), pair 1 = (H(m), apk), from the workspace slots the phase kernels filled (APK 0..2 Jacobian, SIG 3..6 affine,
    H 7..12 Jacobian): Miller loop over both pairs with a shared squaring, final exponentiation, == 1. Result flag; the status word
    receives MBLS_ST_PAIRING_FAILED through the kernel."""
    m = Machine(name, nlane=nlane)
    skip0, skip1, tmp, tmp2 = m.flag("skip0"), m.flag("skip1"), m.flag("tmp"), m.flag("tmp2")
    b = m.block("load")
    apk = [b.loadw("apk%d" % i, WS_APK + i) for i in range(3)]
    sig = [b.loadw("q0_%d" % i, WS_SIG + i) for i in range(4)]
    h = [b.loadw("hj%d" % i, WS_H + i) for i in range(6)]
    b.iszero2((sig[2], sig[3]), skip0, tmp)                       # infinity is stored as y = 0 (lane_sig)
    b.iszero2((h[4], h[5]), skip1, tmp); b.iszero(apk[2], tmp2); b.flagop(F_OR, skip1, skip1, tmp2)
    # pair 1: Q1 = (X Z : Y : Z^3) homogeneous, P1 = (-(X Z), Y, Z^3) (g2h_from_jacobian / g1arg_from_jacobian, mbls_pairing.h)
    H = [(h[0], h[1]), (h[2], h[3]), (h[4], h[5])]
    qx = b.mul2(H[0], H[2]); qz = b.mul2(b.sqr2(H[2]), H[2])
    for i, v in enumerate([qx[0], qx[1], H[1][0], H[1][1], qz[0], qz[1]]):
        b.out("q1_%d" % i, v); b.out("t1_%d" % i, v)
    one = b.const(ONE_D)
    for i, v in enumerate([sig[0], sig[1], sig[2], sig[3], one, LC()]):      # T0 = (x, y, 1)
        b.out("t0_%d" % i, v)
    zz = b.mul(apk[2], apk[2])
    masked_p(b, m, skip0, b.const(NPX0_D), b.const(PY0_D), one, "p0")
    masked_p(b, m, skip1, b.neg(b.mul(apk[0], apk[2])), apk[1], b.mul(zz, apk[2]), "p1")
    for i in range(12):
        b.out("f%d" % i, one if i == 0 else LC())
    pairs = [dict(T="t0_", Q="q0_", affine=True, P="p0"), dict(T="t1_", Q="q1_", affine=False, P="p1")]
    build_miller(m, pairs)
    build_final_exp(m)
    ok = m.flag("ok")
    b = is_one_block(m, "isone", ok)
    b.result(ok)
    m.run("load"); run_miller(m); run_final_exp(m); m.run("isone")
    return m


def prog_vmtail():
    """The tail of verify_multiple_aggregate_signatures / aggregate_verify (reference src/aggregates.rs:307-315, :158-169) on ONE wave:
    F = the product of the per-set Miller values (workspace slots 13..24 of item 0, conjugated values as k_miller_single leaves them),
    S = sum r_i sig_i (slots 25..30, Jacobian). result = (final_exp(F * conj(f_{|x|,S}(-G1))) == 1); S = infinity contributes 1."""
    m = Machine("vmtail")
    skip, tmp = m.flag("skip"), m.flag("tmp")
    b = m.block("load")
    F = [b.loadw("g%d" % i, WS_F + i) for i in range(12)]        # kept aside in g0..g11 until the loop is done
    s = [b.loadw("sj%d" % i, WS_S + i) for i in range(6)]
    b.iszero2((s[4], s[5]), skip, tmp)
    Sx, Sy, Sz = (s[0], s[1]), (s[2], s[3]), (s[4], s[5])
    qx = b.mul2(Sx, Sz); qz = b.mul2(b.sqr2(Sz), Sz)
    for i, v in enumerate([qx[0], qx[1], Sy[0], Sy[1], qz[0], qz[1]]):
        b.out("q%d" % i, v); b.out("t%d" % i, v)
    one = b.const(ONE_D)
    masked_p(b, m, skip, b.const(NPX0_D), b.const(PY0_D), one, "p")
    for i in range(12):
        b.out("f%d" % i, one if i == 0 else LC())
    build_miller(m, [dict(T="t", Q="q", affine=False, P="p")])
    b = m.block("join")                                           # f <- F * conj(f)
    st12(b, mul12(b, ld12(b, "g"), ld12(b), conj_b=True))
    build_final_exp(m)
    ok = m.flag("ok")
    b = is_one_block(m, "isone", ok)
    b.result(ok)
    m.run("load"); run_miller(m); m.run("join"); run_final_exp(m); m.run("isone")
    return m


WS_G = 97                                # slots 97..108 of item 0: the Miller value of (S, -G1) between `smiller` and `vmfinal`


def prog_smiller():
    """The first half of vmtail on its own, for batches that run their chains side by side: as soon as S = sum r_i sig_i exists (slots
    25..30 of item 0), one wave walks the Miller loop of (S, -G1) and leaves its value in slots 97..108 -- beside the sets' Miller loops."""
    m = Machine("smiller")
    skip, tmp = m.flag("skip"), m.flag("tmp")
    b = m.block("load")
    s = [b.loadw("sj%d" % i, WS_S + i) for i in range(6)]
    b.iszero2((s[4], s[5]), skip, tmp)
    Sx, Sy, Sz = (s[0], s[1]), (s[2], s[3]), (s[4], s[5])
    qx = b.mul2(Sx, Sz); qz = b.mul2(b.sqr2(Sz), Sz)
    for i, v in enumerate([qx[0], qx[1], Sy[0], Sy[1], qz[0], qz[1]]):
        b.out("q%d" % i, v); b.out("t%d" % i, v)
    one = b.const(ONE_D)
    masked_p(b, m, skip, b.const(NPX0_D), b.const(PY0_D), one, "p")
    for i in range(12):
        b.out("f%d" % i, one if i == 0 else LC())
    build_miller(m, [dict(T="t", Q="q", affine=False, P="p")])
    b = m.block("store")
    f = flat12(ld12(b))
    for i in range(12):
        b.storew(f[i], WS_G + i)
    m.run("load"); run_miller(m); m.run("store")
    return m


def prog_miller1():
    """One pair of the n-pairing paths (aggregate_verify: (H(m_i), pk_i), reference src/aggregates.rs:144-155; verify_multiple: (H(m_i), [r_i] apk_i),
    :290-300) on one wave, for small batches: the Miller loop of (H, P) from workspace slots 7..12 / 0..2 (both Jacobian), its conjugated
    value into slots 13..24 -- what k_miller_single leaves with one lane per pair."""
    m = Machine("miller1")
    skip, tmp, tmp2 = m.flag("skip"), m.flag("tmp"), m.flag("tmp2")
    b = m.block("load")
    apk = [b.loadw("apk%d" % i, WS_APK + i) for i in range(3)]
    h = [b.loadw("hj%d" % i, WS_H + i) for i in range(6)]
    b.iszero2((h[4], h[5]), skip, tmp); b.iszero(apk[2], tmp2); b.flagop(F_OR, skip, skip, tmp2)
    H = [(h[0], h[1]), (h[2], h[3]), (h[4], h[5])]
    qx = b.mul2(H[0], H[2]); qz = b.mul2(b.sqr2(H[2]), H[2])
    for i, v in enumerate([qx[0], qx[1], H[1][0], H[1][1], qz[0], qz[1]]):
        b.out("q%d" % i, v); b.out("t%d" % i, v)
    one = b.const(ONE_D)
    zz = b.mul(apk[2], apk[2])
    masked_p(b, m, skip, b.neg(b.mul(apk[0], apk[2])), apk[1], b.mul(zz, apk[2]), "p")
    for i in range(12):
        b.out("f%d" % i, one if i == 0 else LC())
    build_miller(m, [dict(T="t", Q="q", affine=False, P="p")])
    b = m.block("store")
    f = ld12(b)
    f = (f[0], b.neg6(f[1]))                                      # the conjugate (x < 0)
    for i, v in enumerate(flat12(f)):
        b.storew(v, WS_F + i)
    m.run("load"); run_miller(m); m.run("store")
    return m


def prog_vmfinal():
    """The second half: (product of the sets' Miller values, slots 13..24) * conj(the value `smiller` left in 97..108), ONE final
    exponentiation, == 1, the batch's status bits"""
    m = Machine("vmfinal")
    b = m.block("load")
    for i in range(12):
        b.loadw("g%d" % i, WS_F + i)
        b.loadw("f%d" % i, WS_G + i)
    b = m.block("join")
    st12(b, mul12(b, ld12(b, "g"), ld12(b), conj_b=True))
    build_final_exp(m)
    ok = m.flag("ok")
    b = is_one_block(m, "isone", ok)
    b.result(ok)
    m.run("load"); m.run("join"); run_final_exp(m); m.run("isone")
    return m


def prog_f12tree():
    """item 0's Miller value <- the product of the Miller values of up to 64 consecutive items (workspace slots 13..24), for the product
    trees of the n-pairing paths: the kernel maps lane groups to items; here: state f <- f * g, a block the kernel runs log-many times with
    g loaded from the partner item. (See k_coop_f12tree in mbls_coop.h.)"""
    m = Machine("f12mul")
    b = m.block("load")
    for i in range(12):
        b.loadw("f%d" % i, WS_F + i)
    b = m.block("loadg")
    for i in range(12):
        b.loadw("g%d" % i, WS_F + i, partner=True)
    b = m.block("mul")
    st12(b, mul12(b, ld12(b), ld12(b, "g")))
    b = m.block("store")
    f = flat12(ld12(b))
    for i in range(12):
        b.storew(f[i], WS_F + i)
    m.run("load"); m.run("loadg"); m.run("mul"); m.run("store")
    return m


def g2_dbl_formula(b, X, Y, Z):
    """Jacobian doubling on the twist (g2_dbl in mbls_curve.h / prog_g2_dbl_d in gen_tower_d.py; valid for every point incl. infinity)"""
    A = b.sqr2(X); B = b.sqr2(Y); C = b.sqr2(B)
    D = b.scale2(b.mul2(X, B), 4)                          # 2 ((X + B)^2 - A - C) = 4 X B as a product: no sum to materialise on the critical path
    E = b.mul3_2(A); F = b.scale2(b.sqr2(A), 9)            # (3 A)^2 as 9 A^2: the operands of the squaring stay single products
    Z3 = b.dbl2(b.mul2(Y, Z))
    X3 = b.sub2(F, b.dbl2(D))
    Y3 = b.sub2(b.mul2(E, b.sub2(D, X3)), b.mul8_2(C))
    return [X3, Y3, Z3]


def g2_add_formula(b, m, A, Q):
    """A + Q in Jacobian coordinates with g2_add's case handling (mbls_curve.h; prog_g2_add in gen_tower_d.py): an operand at infinity
    gives the other one, equal operands the doubling, opposite operands Z = 0 by the formulas"""
    fh, fr, fi1, fi2 = m.flag("h0"), m.flag("r0"), m.flag("inf1"), m.flag("inf2")
    tmps = [m.flag("tmp%d" % i) for i in range(4)]               # one scratch flag per zero test: the four tests share their steps
    z1z1, z2z2 = b.sqr2(A[2]), b.sqr2(Q[2])
    u1, u2 = b.mul2(A[0], z2z2), b.mul2(Q[0], z1z1)
    s1 = b.mul2(b.mul2(A[1], Q[2]), z2z2)
    s2 = b.mul2(b.mul2(Q[1], A[2]), z1z1)
    h = b.sub2(u2, u1)
    rr = b.dbl2(b.sub2(s2, s1))
    b.iszero2(h, fh, tmps[0]); b.iszero2(rr, fr, tmps[1])
    b.iszero2(A[2], fi1, tmps[2]); b.iszero2(Q[2], fi2, tmps[3])
    i4 = b.sqr2(b.dbl2(h))
    j, v = b.mul2(h, i4), b.mul2(u1, i4)
    X3 = b.sub2(b.sub2(b.sqr2(rr), j), b.dbl2(v))
    Y3 = b.sub2(b.mul2(rr, b.sub2(v, X3)), b.dbl2(b.mul2(s1, j)))
    Z3 = b.mul2(b.sub2(b.sub2(b.sqr2(b.add2(A[2], Q[2])), z1z1), z2z2), h)
    D = g2_dbl_formula(b, *A)
    feq = m.flag("eq")
    b.flagop(F_AND, feq, fh, fr); b.flagop(F_ANDN, feq, feq, fi1); b.flagop(F_ANDN, feq, feq, fi2)
    out = [b.sel2(feq, o, d) for o, d in zip((X3, Y3, Z3), D)]
    out = [b.sel2(fi1, o, q) for o, q in zip(out, Q)]
    out = [b.sel2(fi2, o, a) for o, a in zip(out, A)]
    return out


def prog_g2add():
    """slots 25..30 (Jacobian G2 accumulator of the n-pairing paths) of this wave's item += the same slots of its partner item: one level
    of the sum tree of verify_multiple's sum r_i sig_i (reference src/aggregates.rs:303) on one wave"""
    m = Machine("g2add")
    b = m.block("load")
    for i in range(6):
        b.loadw("a%d" % i, WS_S + i)
        b.loadw("q%d" % i, WS_S + i, partner=True)
    b = m.block("add")
    out = g2_add_formula(b, m, pt_in(b, "a"), pt_in(b, "q"))
    pt_out(b, "a", out)
    b = m.block("store")
    for i, v in enumerate([x for c in pt_in(b, "a") for x in c]):
        b.storew(v, WS_S + i)
    m.run("load"); m.run("add"); m.run("store")
    return m


WS_U0, WS_U1 = 31, 37                   # hash_to_field leaves u0 in slots 31, 32 and u1 in 37, 38 (hash_fields_to_ws, mbls_lanes.h)


def pt_psi(b, pt):
    return [b.mul2(b.conj2(pt[0]), T.c2(b, T.PSI_CX)), b.mul2(b.conj2(pt[1]), T.c2(b, T.PSI_CY)), b.conj2(pt[2])]


def prog_hashg2(name="hashg2", nlane=NLANE):
    """hash_to_curve_g2 after hash_to_field (reference src/amcl_utils.rs:33-35; the generated one-lane routine is g2_group_routine('hash') in
    tools/gen_tower_d.py) on ONE wave: both map_to_curve evaluations side by side (simplified SWU with its two fixed-exponent calls each,
    3-isogeny), q0 + q1, and the Budroni-Pintore cofactor clearing [x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2 P) with two ladders by |x|.
    In: u0, u1 in workspace slots 31, 32 / 37, 38; out: H (Jacobian) in slots 7..12."""
    m = Machine(name, nlane=nlane)
    b = m.block("sswu")
    outs = []
    for k, ws in enumerate((WS_U0, WS_U1)):
        u = (b.loadw("u%d_0" % k, ws), b.loadw("u%d_1" % k, ws + 1))
        masks = tuple(m.flag("%s%d" % (nm, k)) for nm in ("t0_", "sq1_", "chi_", "s0_", "s1_", "tmp_"))
        outs.append(T.sswu_formula(b, u, masks))
    pt_out(b, "a", outs[0]); pt_out(b, "bs", outs[1])
    b = m.block("ladd")                                          # acc <- acc + base
    pt_out(b, "a", g2_add_formula(b, m, pt_in(b, "a"), pt_in(b, "bs")))
    b = m.block("lsub")                                          # acc <- acc - base
    q = pt_in(b, "bs")
    pt_out(b, "a", g2_add_formula(b, m, pt_in(b, "a"), [q[0], b.neg2(q[1]), q[2]]))
    b = m.block("ldbl")
    pt_out(b, "a", g2_dbl_formula(b, *pt_in(b, "a")))
    b = m.block("h_base1")                                       # p = q0 + q1: remembered, and the first ladder's base
    a = pt_in(b, "a"); pt_out(b, "pp", a); pt_out(b, "bs", a)
    b = m.block("h_after1")                                      # t1 = -[|x|] p = [x] p; t2 = psi(p); acc = p (to be doubled)
    a = pt_in(b, "a"); pp = pt_in(b, "pp")
    pt_out(b, "ta", [a[0], b.neg2(a[1]), a[2]]); pt_out(b, "tb", pt_psi(b, pp)); pt_out(b, "a", pp)
    b = m.block("h_psi2")                                        # acc = psi^2(2 p); base = t2
    a = pt_in(b, "a")
    pt_out(b, "a", [b.mulfp2(a[0], b.const(D392(T.PSI2_CX))), b.neg2(a[1]), a[2]]); pt_out(b, "bs", pt_in(b, "tb"))
    b = m.block("h_t3")                                          # t3 = acc; acc = t1; base = t2
    pt_out(b, "tc", pt_in(b, "a")); pt_out(b, "a", pt_in(b, "ta")); pt_out(b, "bs", pt_in(b, "tb"))
    b = m.block("h_base2")                                       # t1 + t2 is the second ladder's base
    pt_out(b, "bs", pt_in(b, "a"))
    b = m.block("h_after2")                                      # acc = [x](t1 + t2); base = t3
    a = pt_in(b, "a")
    pt_out(b, "a", [a[0], b.neg2(a[1]), a[2]]); pt_out(b, "bs", pt_in(b, "tc"))
    b = m.block("h_ad_t1")
    pt_out(b, "bs", pt_in(b, "ta"))
    b = m.block("h_ad_p")
    pt_out(b, "bs", pt_in(b, "pp"))
    b = m.block("store")
    for i, v in enumerate([x for c in pt_in(b, "a") for x in c]):
        b.storew(v, WS_H + i)

    def ladder():
        for ph in range(6):
            m.run("ldbl", RUNS[ph])
            if ph < 5:
                m.run("ladd")
    m.run("sswu"); m.run("ladd"); m.run("h_base1"); ladder(); m.run("h_after1"); m.run("ldbl"); m.run("h_psi2"); m.run("lsub")
    m.run("h_t3"); m.run("ladd"); m.run("h_base2"); ladder(); m.run("h_after2"); m.run("ladd")
    m.run("h_ad_t1"); m.run("lsub"); m.run("h_ad_p"); m.run("lsub"); m.run("store")
    return m


PROGRAMS = {"hashg2": prog_hashg2, "pairing2": prog_pairing2, "vmtail": prog_vmtail, "f12mul": prog_f12tree, "g2add": prog_g2add, "smiller": prog_smiller, "vmfinal": prog_vmfinal,
            "miller1": prog_miller1,
            # the same programs for batches that fill the chip: a step serves two / four items side by side (mbls_coop.h), each on 32 / 16 lanes --
            # more steps per wave, fewer per item (the formulas rarely use more than a third of a step's 64 lanes)
            "pairing2x2": lambda: prog_pairing2("pairing2x2", 32), "hashg2x4": lambda: prog_hashg2("hashg2x4", 16)}


def emit_c(name, comp):
    U = name.upper()
    L = ["// program %s: %d steps, %d microcode rows, %d slots, %d flags" % (name, len(comp["steps"]), len(comp["rows"]), comp["n_slots"], comp["n_flags"])]
    L.append("static const uint32_t MBLS_COOP_%s_STEPS[%d] = {" % (U, 2 * len(comp["steps"])))
    row = []
    for info, r in comp["steps"]:
        row.append("0x%x,%d" % (info, r))
    for i in range(0, len(row), 16):
        L.append("  " + ",".join(row[i:i + 16]) + ",")
    L.append("};")
    L.append("static const uint32_t MBLS_COOP_%s_ROWS[%d] = {" % (U, 512 * len(comp["rows"])))
    for words in comp["rows"]:
        for i in range(0, 512, 32):
            L.append("  " + ",".join("0x%x" % w for w in words[i:i + 32]) + ",")
    L.append("};")
    cs = sorted(comp["consts"].items())
    L.append("static const uint32_t MBLS_COOP_%s_CONSTS[%d] = {      // slot, 14 digits" % (U, 15 * max(1, len(cs))))
    for slot, v in cs:
        L.append("  %d," % slot + ",".join("0x%x" % d for d in T.digits_of(v)) + ",")
    if not cs:
        L.append("  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,")
    L.append("};")
    L.append("#define MBLS_COOP_%s_NSTEPS %d" % (U, len(comp["steps"])))
    L.append("#define MBLS_COOP_%s_NROWS %d" % (U, len(comp["rows"])))
    L.append("#define MBLS_COOP_%s_NCONSTS %d" % (U, len(cs)))
    L.append("#define MBLS_COOP_%s_NSLOTS %d" % (U, comp["n_slots"]))
    L.append("#define MBLS_COOP_%s_LPI %d" % (U, comp["nlane"]))
    return "\n".join(L) + "\n"


def digit_constants():
    from gen_fp_asm import P28, NP28
    L = ["// the resident constants of mbls_fp_mul1_d_asm_fn (tools/gen_fpd_asm.py load_constants): digits of p, -1/p mod 2^28"]
    L.append("#define MBLS_COOP_P28 {" + ",".join("0x%x" % d for d in P28) + "}")
    L.append("#define MBLS_COOP_NP28 0x%xu" % NP28)
    L.append("#define MBLS_COOP_RECIP_PTOP %r" % (1.0 / (P >> 364)))
    return "\n".join(L) + "\n"


def build_all():
    out = {}
    for name, fn in PROGRAMS.items():
        out[name] = compile_program(fn())
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    # MBLS_GEN_OUT_DIR: write there instead of over the tracked file (the freshness tests generate into a temporary directory and compare)
    path = os.path.join(os.environ.get("MBLS_GEN_OUT_DIR") or os.path.join(os.path.dirname(here), "milagro_bls_amd", "csrc"), "mbls_coop_prog.inc")
    txt = "// GENERATED by tools/gen_coop.py -- do not edit. Microprograms of the wave-cooperative engine (mbls_coop.h).\n"
    mx = 0
    for name, comp in build_all().items():
        txt += emit_c(name, comp)
        if comp["nlane"] == NLANE:                           # the static allocation of the one-item-per-wave kernel instances
            mx = max(mx, comp["n_slots"])
        print(name, "steps", comp["total_steps"], "rows", len(comp["rows"]), "slots", comp["n_slots"], "flags", comp["n_flags"])
        for bn, s in comp["stats"].items():
            print("   ", bn, s)
    txt += "#define MBLS_COOP_MAX_SLOTS %d\n" % mx
    txt += digit_constants()
    with open(path, "w") as f:
        f.write(txt)
    print("wrote", path)


if __name__ == "__main__":
    main()
