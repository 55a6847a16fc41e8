#!/usr/bin/env python3
"""Generate milagro_bls_amd/csrc/mbls_tower_asm.inc: the Fp12 multiplication as a straight-line gfx950 routine on 12 x 32-bit limbs
(first generation of the tower generator; the loop bodies -- Miller loop, cyclotomic squaring, G2 doubling runs -- are generated in
digit form by tools/gen_tower_d.py, which reuses the Prog recorder of this file).

Why: inside these loops hipcc cannot keep the working set (an Fp12 = 144 registers, plus Fp6 temporaries, plus the fixed
register window of the Fp2 multiplication routines) in registers; it spills to lane-private scratch memory, and with one wave
per SIMD every reload is an exposed HBM round trip (measured: 23-27 % of k_final / k_miller wave cycles waiting). Here the
program is known in full, so values are placed by hand-rolled allocation with exact next-use knowledge (Belady eviction):
VGPR blocks first, AGPRs as the spill space (one v_accvgpr move per limb instead of a memory round trip), LDS for state that
crosses the routine boundary. No scratch memory is touched.

Structure:  Prog  -- records a straight-line program over Fp values (add, sub, select, Fp2 mul/sqr/mul-by-Fp calls, LDS moves)
            Alloc -- walks it, assigns 12-register blocks, emits instructions
The emitted text is checked on the CPU by tools/asm_sim.py against the big-integer model (tests/test_asm_sim_cpu.py).
Run:  python3 tools/gen_tower_asm.py    (output committed; tests check it is up to date)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_fp_asm import P, PL, emit  # noqa: E402

INF = 1 << 60
# Register map (12-register blocks). v0..v119 is the window of the multiplication routines (they overwrite all of it except
# their result blocks, operands included); between calls its first eight blocks are ordinary storage and v96..v119 is the
# scratch of the modular additions.
WIN = [0, 12, 24, 36, 48, 60, 72, 84]
FREE_V = [120 + 12 * i for i in range(10)]         # v120..v239: survive calls
ALL_V = FREE_V + [84, 72, 36, 24, 12, 0, 60, 48]   # preference order for fresh values
PB = 240                                           # v240..v251: the modulus
U = 96                                             # v96..v107: scratch of the modular add/sub (dead between calls)
U2 = 108                                           # v108..v119: scratch of the second chain of an interleaved pair
LADDR = "v252"                                     # byte address of this lane's column in the LDS state
AG = [12 * i for i in range(21)]                   # a0..a251
NLDS = 12                                          # Fp slots of the LDS state (the kernels allocate 144 dwords per lane)
CARRY_B = "s[62:63]"
CARRY_C, CARRY_D = "s[50:51]", "s[52:53]"          # carries of the second chain of an interleaved pair

F12_ARG = FREE_V + [72, 84]                         # the twelve blocks in which a routine receives / returns an Fp12 operand

ROUTINES = {
    "mul": dict(name="mbls_fp2_mul_asm_fn", ins=[0, 12, 24, 36], outs=[48, 60], clob=WIN),
    "sqr": dict(name="mbls_fp2_sqr_asm_fn", ins=[0, 12], outs=[24, 36], clob=WIN),
    "mulfp": dict(name="mbls_fp2_mulfp_asm_fn", ins=[0, 12, 24], outs=[36, 48], clob=WIN),
}


class Prog:
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
        outs = [self.new(), self.new()]
        assert len(set(ins)) == len(ins)
        self.ops.append((kind, outs, list(ins), None))
        return tuple(outs)

    def store(self, a, loc):
        self.ops.append(("store", [], [a], loc))

    def keep(self, vals):
        """values that stay in their live-in homes for the next round of the loop (never moved, only copied)"""
        self.ops.append(("keep", [], list(vals), None))

    # ---- Fp2 layer (an Fp2 is a pair of values)
    def pair(self, k0, a0, b0, k1, a1, b1):
        """two independent Fp additions/subtractions, emitted as interleaved carry chains (a lone wave issues a dependent
        v_addc chain at ~6.8 clk per instruction, two interleaved chains at ~4.5)"""
        d0, d1 = self.new(), self.new()
        self.ops.append(("pair", [d0, d1], [a0, b0, a1, b1], (k0, k1)))
        return (d0, d1)

    def add2(self, a, b): return self.pair("add", a[0], b[0], "add", a[1], b[1])
    def sub2(self, a, b): return self.pair("sub", a[0], b[0], "sub", a[1], b[1])
    def dbl2(self, a): return self.add2(a, a)
    def mul2(self, a, b): return self.call("mul", [a[0], a[1], b[0], b[1]])
    def sqr2(self, a): return self.call("sqr", [a[0], a[1]])
    def mulfp2(self, a, s): return self.call("mulfp", [a[0], a[1], s])
    def mul_xi2(self, a): return self.pair("sub", a[0], a[1], "add", a[0], a[1])    # (1 + i) a
    def sel2(self, mask, a, b): return (self.sel(mask, a[0], b[0]), self.sel(mask, a[1], b[1]))
    def store2(self, a, slot2): self.store(a[0], ("l", 2 * slot2)); self.store(a[1], ("l", 2 * slot2 + 1))
    def mul3_2(self, a): return self.add2(self.dbl2(a), a)
    def mul4_2(self, a): return self.dbl2(self.dbl2(a))
    def mul8_2(self, a): return self.dbl2(self.mul4_2(a))
    def mul12_2(self, a): return self.add2(self.mul8_2(a), self.mul4_2(a))

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


class Alloc:
    def __init__(self, prog):
        self.p = prog
        self.uses = {}
        for k, (kind, outs, ins, aux) in enumerate(prog.ops):
            for v in ins:
                self.uses.setdefault(v, []).append(k)
        # sums that only ever enter a multiplication as a first factor stay unreduced (< 2p): the routines convert first factors
        # as a * 2^8 < 2^392 and their results stay below 2p for first factors up to 8p
        self.unreduced = set()
        first_factor = {"mul": (0, 1), "mulfp": (0, 1)}
        for k, (kind, outs, ins, aux) in enumerate(prog.ops):
            cands = []
            if kind == "add":
                cands = [outs[0]]
            elif kind == "pair":
                cands = [outs[i] for i in range(2) if aux[i] == "add"]
            for d in cands:
                us = self.uses.get(d, [])
                ok = bool(us)
                for u in us:
                    uk, uo, ui, ua = prog.ops[u]
                    if uk not in first_factor or any(ui[j] == d for j in range(len(ui)) if j not in first_factor[uk]):
                        ok = False
                if ok:
                    self.unreduced.add(d)
        self.loc = dict(prog.init_loc)
        self.at = {l: v for v, l in self.loc.items()}
        self.out = []
        self.pending_lds = False
        self.stats = dict(vmov=0, acc=0, lds=0, arith=0, calls=0)

    # ---- liveness
    def next_use(self, v, k):
        for u in self.uses.get(v, ()):
            if u >= k:
                return u
        return INF

    # ---- location bookkeeping
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

    def free_block(self, kind, pool, avoid=()):
        for b in pool:
            if (kind, b) not in self.at and b not in avoid:
                return b
        return None

    # ---- instruction emission
    def e(self, s):
        self.out.append(s)

    def wait_lds(self):
        if self.pending_lds:
            self.e("s_waitcnt lgkmcnt(0)")
            self.e("s_nop 0")                 # keeps the 8-byte instructions that follow 8-byte aligned
            self.pending_lds = False

    def copy(self, src, dst):
        """copy one Fp between locations; dst is ('v', b), ('a', b) or ('l', slot)"""
        (sk, sb), (dk, db) = src, dst
        if sk == "v" and dk == "v":
            for j in range(0, 12, 2):
                self.e("v_mov_b64_e64 v[%d:%d], v[%d:%d]" % (db + j, db + j + 1, sb + j, sb + j + 1))
            self.stats["vmov"] += 6
        elif sk == "v" and dk == "a":
            for j in range(12):
                self.e("v_accvgpr_write_b32 a%d, v%d" % (db + j, sb + j))
            self.stats["acc"] += 12
        elif sk == "a" and dk == "v":
            for j in range(12):
                self.e("v_accvgpr_read_b32 v%d, a%d" % (db + j, sb + j))
            self.stats["acc"] += 12
        elif sk == "l" and dk == "v":
            for j in range(0, 12, 2):
                o = sb * 12 + j
                self.e("ds_read2st64_b32 v[%d:%d], %s offset0:%d offset1:%d" % (db + j, db + j + 1, LADDR, o, o + 1))
            self.pending_lds = True
            self.stats["lds"] += 6
        elif sk == "v" and dk == "l":
            for j in range(0, 12, 2):
                o = db * 12 + j
                self.e("ds_write2st64_b32 %s, v%d, v%d offset0:%d offset1:%d" % (LADDR, sb + j, sb + j + 1, o, o + 1))
            self.stats["lds"] += 6
        else:
            raise ValueError((src, dst))

    # ---- getting a VGPR block
    def alloc_v(self, k, avoid=(), hint=None):
        if hint is not None and ("v", hint) not in self.at and hint not in avoid:
            return hint
        b = self.free_block("v", ALL_V, avoid)
        if b is not None:
            return b
        # evict the VGPR-resident value whose next use is farthest away
        best, bu = None, -1
        for blk in ALL_V:
            if blk in avoid:
                continue
            w = self.at[("v", blk)]
            u = self.next_use(w, k)
            if u > bu:
                best, bu = blk, u
        w = self.at[("v", best)]
        if bu == INF:
            self.release(w)
            return best
        self.spill(w)
        return best

    def spill(self, w):
        """move a VGPR-resident value out of the VGPR file (AGPR first, LDS if the AGPRs are full)"""
        src = self.loc[w]
        ab = self.free_block("a", AG)
        if ab is not None:
            self.copy(src, ("a", ab)); self.place(w, ("a", ab)); return
        ls = self.free_block("l", range(NLDS))
        if ls is None:
            raise RuntimeError("out of storage")
        self.copy(src, ("l", ls)); self.place(w, ("l", ls))

    def to_vgpr(self, v, k, avoid=()):
        l = self.loc[v]
        if l[0] == "v":
            return l[1]
        b = self.alloc_v(k, avoid)
        self.copy(l, ("v", b))
        self.place(v, ("v", b))
        return b

    def hint_for(self, d, k):
        """if the next use of d is an operand slot of the very next call, compute it there"""
        u = self.next_use(d, k + 1)
        if u == INF:
            return None
        for j in range(k + 1, u):
            if self.p.ops[j][0] in ROUTINES:
                return None
        kind, outs, ins, aux = self.p.ops[u]
        if kind in ROUTINES:
            return ROUTINES[kind]["ins"][ins.index(d)]
        return None

    # ---- the walk
    def run(self):
        for j in range(12):
            self.e("v_mov_b32_e32 v%d, 0x%08x" % (PB + j, PL[j]))
        for k, (kind, outs, ins, aux) in enumerate(self.p.ops):
            if kind in ("add", "sub", "sel"):
                self.do_arith(k, kind, outs[0], ins, aux)
            elif kind == "pair":
                self.do_pair(k, outs, ins, aux)
            elif kind == "const":
                b = self.alloc_v(k, hint=self.hint_for(outs[0], k))
                for j in range(12):
                    self.e("v_mov_b32_e32 v%d, 0x%08x" % (b + j, (aux >> (32 * j)) & 0xFFFFFFFF))    # 4 + 4 bytes with the literal
                self.place(outs[0], ("v", b))
            elif kind in ROUTINES:
                self.do_call(k, kind, outs, ins)
            elif kind == "store":
                self.do_store(k, ins[0], aux)
            elif kind == "keep":
                for v in ins:
                    assert self.loc[v] == self.p.init_loc[v], "pinned value moved"
                continue
            for v in set(ins):
                if self.next_use(v, k + 1) == INF:
                    self.release(v)
        self.wait_lds()
        self.e("s_waitcnt lgkmcnt(0)")
        self.e("s_nop 0")
        return self.out

    def pick_dst(self, k, d, a, b, ba, bb, avoid):
        """destination block of d = a op b: the operand slot of the next call if d goes straight there, else in place over an
        operand that dies here, else None (a fresh block)"""
        hint = self.hint_for(d, k)
        if hint is not None and hint not in (ba, bb) and hint not in avoid and ("v", hint) not in self.at:
            return hint
        for v, blk in ((a, ba), (b, bb)):
            if self.next_use(v, k + 1) == INF and blk not in avoid:
                return blk
        return None

    @staticmethod
    def gen_arith(kind, D, A, B, Ub, c1, c2, aux=None):
        """instruction list of one modular operation on 12-register blocks; c1/c2: carry registers (vcc or an SGPR pair)"""
        # 8-byte (VOP3) encodings throughout: a lone wave fetches 8-byte instructions that sit at 4 mod 8 markedly slower,
        # and with no 4-byte instructions in the stream nothing ever does
        def co(op, d, x, y, c):
            return ("%s_e64 v%d, " + c + ", v%d, v%d") % (op, d, x, y)

        def cc(op, d, x, y, c):
            return ("%s_e64 v%d, " + c + ", v%d, v%d, " + c) % (op, d, x, y)

        def cm(d, x, y, c):
            return ("v_cndmask_b32_e64 v%d, v%d, v%d, " + c) % (d, x, y)
        L = []
        if kind == "addnr":                                                      # plain sum, no reduction
            L.append(co("v_add_co_u32", D, A, B, c1))
            L += [cc("v_addc_co_u32", D + j, A + j, B + j, c1) for j in range(1, 12)]
        elif kind == "add":
            L.append(co("v_add_co_u32", D, A, B, c1))
            L += [cc("v_addc_co_u32", D + j, A + j, B + j, c1) for j in range(1, 12)]
            L.append(co("v_sub_co_u32", Ub, D, PB, c1))
            L += [cc("v_subb_co_u32", Ub + j, D + j, PB + j, c1) for j in range(1, 12)]
            L += [cm(D + j, Ub + j, D + j, c1) for j in range(12)]            # borrow ? sum : sum - p
        elif kind == "sub":
            L.append(co("v_sub_co_u32", D, A, B, c1))
            L += [cc("v_subb_co_u32", D + j, A + j, B + j, c1) for j in range(1, 12)]
            L.append(co("v_add_co_u32", Ub, D, PB, c2))                          # the borrow stays in c1
            L += [cc("v_addc_co_u32", Ub + j, D + j, PB + j, c2) for j in range(1, 12)]
            L += [cm(D + j, D + j, Ub + j, c1) for j in range(12)]            # borrow ? diff + p : diff
        else:                                                                    # sel: mask ? b : a
            L += ["v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (D + j, A + j, B + j, aux) for j in range(12)]
        return L

    def finish_arith(self, d, bd, operands):
        # d may have been written over a dying operand's block: drop that operand first
        for v in operands:
            if self.loc.get(v) == ("v", bd):
                self.release(v)
        self.place(d, ("v", bd))

    def do_arith(self, k, kind, d, ins, aux):
        a, b = ins
        self.to_vgpr(a, k)
        if b != a:
            self.to_vgpr(b, k, avoid=(self.loc[a][1],))
        ba, bb = self.loc[a][1], self.loc[b][1]
        self.wait_lds()
        bd = self.pick_dst(k, d, a, b, ba, bb, ())
        if bd is None:
            bd = self.alloc_v(k, avoid=(ba, bb))
        if kind == "add" and d in self.unreduced:
            kind = "addnr"
        for l in self.gen_arith(kind, bd, ba, bb, U, "vcc", CARRY_B, aux):
            self.e(l)
        self.stats["arith"] += 12 if kind in ("sel", "addnr") else 36
        self.finish_arith(d, bd, (a, b))

    def do_pair(self, k, outs, ins, kinds):
        a0, b0, a1, b1 = ins
        got = []
        for v in (a0, b0, a1, b1):
            if v not in got:
                self.to_vgpr(v, k, avoid=tuple(self.loc[w][1] for w in got))
                got.append(v)
        blk = {v: self.loc[v][1] for v in got}
        assert all(self.loc[v][0] == "v" for v in got)
        self.wait_lds()
        # an operand of one operation must not be overwritten by the other one's result (the chains run interleaved)
        d0 = self.pick_dst(k, outs[0], a0, b0, blk[a0], blk[b0], avoid=(blk[a1], blk[b1]))
        if d0 is None:
            d0 = self.alloc_v(k, avoid=tuple(blk.values()))
        d1 = self.pick_dst(k, outs[1], a1, b1, blk[a1], blk[b1], avoid=(blk[a0], blk[b0], d0))
        if d1 is None:
            d1 = self.alloc_v(k, avoid=tuple(blk.values()) + (d0,))
        kinds = tuple("addnr" if kinds[i] == "add" and outs[i] in self.unreduced else kinds[i] for i in range(2))
        L0 = self.gen_arith(kinds[0], d0, blk[a0], blk[b0], U, "vcc", CARRY_B)
        L1 = self.gen_arith(kinds[1], d1, blk[a1], blk[b1], U2, CARRY_C, CARRY_D)
        for i in range(max(len(L0), len(L1))):
            if i < len(L0):
                self.e(L0[i])
            if i < len(L1):
                self.e(L1[i])
        self.stats["arith"] += len(L0) + len(L1)
        for bd in (d0, d1):
            for v in got:
                if self.loc.get(v) == ("v", bd):
                    self.release(v)
        self.place(outs[0], ("v", d0)); self.place(outs[1], ("v", d1))

    def do_call(self, k, kind, outs, ins):
        R = ROUTINES[kind]
        slots = R["ins"]
        touched = set(slots) | set(R["clob"])
        want = {slots[i]: ins[i] for i in range(len(ins))}
        ready = set()
        # 1. everything in the window is overwritten by the routine (its operands too): move out what is still needed
        for s in sorted(touched):
            w = self.at.get(("v", s))
            if w is None:
                continue
            in_place = want.get(s) == w
            is_operand = w in ins
            live_after = self.next_use(w, k + 1) != INF
            if in_place:
                ready.add(s)
                if live_after:                      # the slot keeps the operand copy; the value itself continues elsewhere
                    b = self.alloc_v(k, avoid=touched)
                    self.copy(("v", s), ("v", b)); self.place(w, ("v", b))
                continue
            if is_operand or live_after:
                u = self.next_use(w, k if is_operand else k + 1)
                nk = self.p.ops[u][0] if u != INF else None
                # a value that is next consumed as an operand of a call loses nothing by waiting in an AGPR
                if not is_operand and nk in ROUTINES and self.free_block("a", AG) is not None:
                    self.spill(w)
                else:
                    b = self.alloc_v(k, avoid=touched)
                    self.copy(("v", s), ("v", b)); self.place(w, ("v", b))
            else:
                self.release(w)
        # 2. operands into their slots (untracked copies: they die with the call)
        for s, v in want.items():
            if s in ready:
                continue
            self.copy(self.loc[v], ("v", s))
        self.wait_lds()
        self.e("CALL " + R["name"])
        self.stats["calls"] += 1
        for s in touched:
            w = self.at.get(("v", s))
            if w is not None:
                assert self.next_use(w, k + 1) == INF, "live value in the window across a call"
                self.release(w)
        for i, o in enumerate(outs):
            self.place(o, ("v", R["outs"][i]))

    def do_store(self, k, a, dst):
        w = self.at.get(dst)
        if w is not None and w != a:
            if self.next_use(w, k + 1) != INF:
                b = self.alloc_v(k, avoid=(self.loc[a][1],) if self.loc[a][0] == "v" else ())
                self.copy(dst, ("v", b)); self.place(w, ("v", b))
                self.wait_lds()
            else:
                self.release(w)
        if self.loc[a] != dst:
            b = self.to_vgpr(a, k)
            self.wait_lds()
            self.copy(("v", b), dst)
        if self.next_use(a, k + 1) == INF:
            self.release(a)
        self.at[dst] = ("stored", a)  # the slot now holds a result: never reused as spill space


# ------------------------------------------------------------------------------------------ the program
# Fp2 coefficient e2 of an Fp12 in tower order: 0 c0.c0, 1 c0.c1, 2 c0.c2, 3 c1.c0, 4 c1.c1, 5 c1.c2  (LDS slots 2*e2, 2*e2+1)
def prog_fp12_mul():
    """acc <- acc * g: acc in LDS slots 0..11 (tower order), g arriving in the twelve blocks F12_ARG (asm operands)."""
    p = Prog()
    al = [p.live_in(("l", i)) for i in range(12)]
    gl = [p.live_in(("v", b)) for b in F12_ARG]
    six = lambda l: ([(l[0], l[1]), (l[2], l[3]), (l[4], l[5])], [(l[6], l[7]), (l[8], l[9]), (l[10], l[11])])
    a, g = six(al), six(gl)
    t0 = p.mul6(a[0], g[0])
    t1 = p.mul6(a[1], g[1])
    c1 = p.mul6(p.add6(a[0], a[1]), p.add6(g[0], g[1]))
    c1 = p.sub6(p.sub6(c1, t0), t1)
    c0 = p.add6(t0, p.mul_v6(t1))
    for i, v in enumerate([x for h in (c0, c1) for c in h for x in c]):
        p.store(v, ("l", i))
    return p


def plain_shell(body):
    """a routine without a loop: only the return address needs saving around the nested calls"""
    return ["s_mov_b64 s[36:37], s[30:31]", ".p2align 6"] + body + ["s_mov_b64 s[30:31], s[36:37]"]


def expand_calls(lines):
    out = []
    for l in lines:
        if l.startswith("CALL "):
            sym = l.split()[1]
            out += ["s_getpc_b64 s[40:41]", "s_add_u32 s40, s40, %s@rel32@lo+4" % sym, "s_addc_u32 s41, s41, %s@rel32@hi+12" % sym,
                    "s_swappc_b64 s[30:31], s[40:41]"]
        else:
            out.append(l)
    return out


def build(name):
    prog = {"fp12_mul": prog_fp12_mul}[name]()
    al = Alloc(prog)
    lines = al.run()
    return lines, al.stats


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.path.dirname(here), "milagro_bls_amd", "csrc", "mbls_tower_asm.inc")
    txt = "// GENERATED by tools/gen_tower_asm.py -- do not edit.\n"
    lines, stats = build("fp12_mul")
    txt += emit("MBLS_FP12_MUL_ASM", plain_shell(expand_calls(lines))) + "\n"
    print("fp12_mul", len(lines), "lines", stats)
    sg = '"s30","s31","s36","s37","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","vcc","scc","memory"'
    argregs = set(r for b in F12_ARG for r in range(b, b + 12))
    other = [i for i in list(range(0, 120)) + list(range(240, 252)) + [254] if i not in argregs]
    txt += "// the routine takes its second operand as twelve register blocks, in tower order (v252 carries the LDS address and is preserved):\n"
    txt += "#define MBLS_F12_ARG_REGS(x) " + ", ".join('"+{v[%d:%d]}"(x##%d)' % (b, b + 11, i) for i, b in enumerate(F12_ARG)) + "\n"
    txt += "#define MBLS_FP12_ARG_ASM_CLOBBERS %s,%s, \\\n    %s\n" % (
        ",".join('"v%d"' % i for i in other), ",".join('"a%d"' % i for i in range(252)), sg)
    with open(path, "w") as f:
        f.write(txt)
    print("wrote", path)


if __name__ == "__main__":
    main()
