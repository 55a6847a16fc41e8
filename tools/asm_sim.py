#!/usr/bin/env python3
"""Single-lane interpreter for the small gfx950 instruction subset that tools/gen_fp_asm.py, tools/gen_fpd_asm.py and tools/gen_tower_d.py emit.

Development/test infrastructure only (tests/test_asm_sim_cpu.py): it lets the generated routines be checked against the
big-integer model on the CPU before they ever run on a GPU. One lane is simulated, so VCC and the SGPR-pair carry registers
are single bits and an LDS address is just a key. Unknown instructions raise.
"""
import re

M32 = 0xFFFFFFFF


class Machine:
    def __init__(self, routines=None):
        self.v = [0] * 256
        self.a = [0] * 256
        self.s = {}
        self.vcc = 0
        self.lds = {}
        self.mem = {}
        self.routines = routines or {}
        self.count = 0
        self.calls = 0
        self.scc = 0
        self.lane = 0                                # this lane's number within its wave (v_mbcnt_*)
        self.pair_sync = False                       # set by run_pair
        self.model_exec = False                      # True: writes to exec switch THIS lane on / off (its bit = the value written, as for every
        self.exec_on = 1                             # lane mask here) and an inactive lane skips vector and memory instructions

    # ---- operand helpers
    _TOK = {}                                        # operand text -> (kind, index / value): every distinct operand is parsed once

    @classmethod
    def _classify(cls, tok):
        tok = tok.strip()
        if tok == "vcc":
            r = ("vcc", 0)
        elif tok == "exec":                          # one lane: active unless the exec model says otherwise
            r = ("exec", 0)
        else:
            m = re.fullmatch(r"([vsa])(\d+)", tok)
            m2 = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
            if m:
                r = (m.group(1), int(m.group(2)))
            elif m2:
                r = (m2.group(1) + "2", int(m2.group(2)))
            elif tok.startswith("0x"):
                r = ("imm", int(tok, 16))
            elif "." in tok:                         # inline floating-point constant
                r = ("imm", f32_bits(float(tok)))
            else:
                r = ("imm", int(tok) & M32)
        cls._TOK[tok] = r
        return r

    def rd(self, tok):
        k, i = self._TOK.get(tok) or self._classify(tok)
        if k == "v":
            return self.v[i]
        if k == "imm":
            return i
        if k == "s":
            return self.s[i]
        if k == "a":
            return self.a[i]
        if k == "vcc":
            return self.vcc
        if k == "s2":
            return self.s.get(("pair", i), 0)
        if k == "exec":
            return self.exec_on if self.model_exec else 1
        return self.v[i] | (self.v[i + 1] << 32)     # v2

    def wr_carry(self, tok, val):
        tok = tok.strip()
        if tok == "exec":                            # by default the one simulated lane stays active: masked regions are the caller's business
            if self.model_exec:
                self.exec_on = val & 1
            return
        if tok == "vcc":
            self.vcc = val
        else:
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
            self.s[("pair", int(m.group(1)))] = val

    def wr(self, tok, val):
        k, i = self._TOK.get(tok) or self._classify(tok)
        if k == "v":
            self.v[i] = val & M32
        elif k == "a":
            self.a[i] = val & M32
        elif k == "s":
            self.s[i] = val & M32
        elif k == "v2":
            assert i % 2 == 0, "64-bit VGPR operands must be even-aligned: " + tok
            self.v[i] = val & M32
            self.v[i + 1] = (val >> 32) & M32
        elif k == "s2":
            self.s[("pair", i)] = val
        else:
            raise ValueError("bad destination " + tok)

    # ---- execution
    _LINES = {}                                      # instruction text -> (op, args, rest): parsed once

    def run(self, lines):
        for ev in self.run_gen(lines):
            raise RuntimeError("cross-lane instruction outside a lane pair (use run_pair): %r" % (ev,))

    def run_gen(self, lines):
        """the interpreter proper, as a generator: it yields (dst, src) at every cross-lane move (v_mov_b32_dpp) and expects the driver to
        have written dst before it is resumed (run_pair below); for single-lane code it simply runs to the end"""
        pending_call = None
        for line in lines:
            parsed = self._LINES.get(line)
            if parsed is None:
                text = line.split("//")[0].strip()
                if not text or text.startswith("."):  # assembler directives (.p2align)
                    parsed = ()
                else:
                    op, _, rest = text.partition(" ")
                    parsed = (op, [x.strip() for x in self._split(rest)], rest, text)
                self._LINES[line] = parsed
            if not parsed:
                continue
            op, args, rest, line = parsed
            self.count += 1
            if self.pair_sync and op[:3] in ("glo", "ds_"):
                yield None                        # a lane pair stays in step at every memory instruction (see run_pair)
            if self.model_exec and not self.exec_on and op[:2] in ("v_", "ds", "gl"):
                assert "dpp" not in op, "cross-lane move with a lane of the pair switched off"
                continue                          # this lane is switched off: vector and memory instructions do nothing here
            if op == "CALL":                      # pseudo-instruction of the generator: s_getpc/s_add/s_addc/s_swappc to a routine
                self.calls += 1
                yield from self.run_gen(self.routines[args[0]])
            elif op in ("s_mov_b32",):
                self.wr(args[0], self.rd(args[1]))
            elif op == "s_mov_b64":
                self.wr_carry(args[0], self.rd(args[1])) if args[0] in ("vcc", "exec") else self.wr(args[0], self.rd(args[1]))
            elif op in ("v_mov_b32_e32", "v_mov_b32_e64"):
                self.wr(args[0], self.rd(args[1]))
            elif op in ("v_mov_b64_e32", "v_mov_b64_e64"):
                self.wr(args[0], self.rd(args[1]))
            elif op in ("v_accvgpr_write_b32", "v_accvgpr_read_b32"):
                self.wr(args[0], self.rd(args[1]))
            elif op in ("v_add_co_u32_e32", "v_add_co_u32_e64"):
                t = self.rd(args[2]) + self.rd(args[3])
                self.wr(args[0], t); self.wr_carry(args[1], t >> 32)
            elif op in ("v_addc_co_u32_e32", "v_addc_co_u32_e64"):
                t = self.rd(args[2]) + self.rd(args[3]) + self.rd(args[4])
                self.wr(args[0], t); self.wr_carry(args[1], t >> 32)
            elif op in ("v_sub_co_u32_e32", "v_sub_co_u32_e64"):
                t = self.rd(args[2]) - self.rd(args[3])
                self.wr(args[0], t); self.wr_carry(args[1], 1 if t < 0 else 0)
            elif op in ("v_subb_co_u32_e32", "v_subb_co_u32_e64"):
                t = self.rd(args[2]) - self.rd(args[3]) - self.rd(args[4])
                self.wr(args[0], t); self.wr_carry(args[1], 1 if t < 0 else 0)
            elif op == "v_mad_u64_u32":
                t = self.rd(args[2]) * self.rd(args[3]) + self.rd(args[4])
                self.wr(args[0], t & 0xFFFFFFFFFFFFFFFF); self.wr_carry(args[1], t >> 64)
            elif op == "v_mad_i64_i32":
                t = s32(self.rd(args[2])) * s32(self.rd(args[3])) + s64(self.rd(args[4]))
                assert -(1 << 63) <= t < (1 << 63), "signed 64-bit accumulator overflow: " + line
                self.wr(args[0], t & 0xFFFFFFFFFFFFFFFF)
            elif op == "v_ashrrev_i64":
                self.wr(args[0], (s64(self.rd(args[2])) >> self.rd(args[1])) & 0xFFFFFFFFFFFFFFFF)
            elif op == "v_ashrrev_i32_e64":
                self.wr(args[0], s32(self.rd(args[2])) >> self.rd(args[1]))
            elif op == "v_cvt_f32_i32_e64":
                self.wr(args[0], f32_bits(float(s32(self.rd(args[1])))))
            elif op == "v_mul_f32_e64":
                self.wr(args[0], f32_bits(bits_f32(self.rd(args[1])) * bits_f32(self.rd(args[2]))))
            elif op == "v_rndne_f32_e64":
                self.wr(args[0], f32_bits(float(round(bits_f32(self.rd(args[1]))))))       # Python rounds half to even, like v_rndne
            elif op == "v_cvt_i32_f32_e64":
                self.wr(args[0], int(bits_f32(self.rd(args[1]))))
            elif op in ("v_add_u32_e32", "v_add_u32_e64"):
                self.wr(args[0], self.rd(args[1]) + self.rd(args[2]))
            elif op == "v_sub_u32_e64":
                self.wr(args[0], self.rd(args[1]) - self.rd(args[2]))
            elif op == "v_mul_lo_u32":
                self.wr(args[0], self.rd(args[1]) * self.rd(args[2]))
            elif op in ("v_cndmask_b32_e32", "v_cndmask_b32_e64"):
                self.wr(args[0], self.rd(args[2]) if self.rd(args[3]) else self.rd(args[1]))
            elif op == "v_bfe_i32":
                w = self.rd(args[3]); t = (self.rd(args[1]) >> self.rd(args[2])) & ((1 << w) - 1)
                self.wr(args[0], t - (1 << w) if t >> (w - 1) else t)
            elif op in ("v_xor_b32_e32", "v_xor_b32_e64"):
                self.wr(args[0], self.rd(args[1]) ^ self.rd(args[2]))
            elif op == "v_bfe_u32":
                self.wr(args[0], (self.rd(args[1]) >> self.rd(args[2])) & ((1 << self.rd(args[3])) - 1))
            elif op in ("v_and_b32_e32", "v_and_b32_e64"):
                self.wr(args[0], self.rd(args[1]) & self.rd(args[2]))
            elif op == "v_lshrrev_b32_e64":
                self.wr(args[0], self.rd(args[2]) >> self.rd(args[1]))
            elif op == "v_lshlrev_b32_e64":
                self.wr(args[0], self.rd(args[2]) << self.rd(args[1]))
            elif op == "v_lshrrev_b64":
                self.wr(args[0], self.rd(args[2]) >> self.rd(args[1]))
            elif op == "v_lshl_add_u64":
                self.wr(args[0], ((self.rd(args[1]) << self.rd(args[2])) + self.rd(args[3])) & 0xFFFFFFFFFFFFFFFF)
            elif op == "v_lshl_add_u32":
                self.wr(args[0], (self.rd(args[1]) << self.rd(args[2])) + self.rd(args[3]))
            elif op == "v_mbcnt_lo_u32_b32":      # with the mask -1: the number of lanes below this one among lanes 0..31
                assert args[1] == "-1"
                self.wr(args[0], min(self.lane, 32) + self.rd(args[2]))
            elif op == "v_mbcnt_hi_u32_b32":      # ... among lanes 32..63, added to the operand: together the lane's number
                assert args[1] == "-1"
                self.wr(args[0], max(self.lane - 32, 0) + self.rd(args[2]))
            elif op == "v_swap_b32":
                a_, b_ = self.rd(args[0]), self.rd(args[1])
                self.wr(args[0], b_); self.wr(args[1], a_)
            elif op == "v_add3_u32":
                self.wr(args[0], self.rd(args[1]) + self.rd(args[2]) + self.rd(args[3]))
            elif op == "v_mov_b32_dpp":           # v_mov_b32_dpp vdst, vsrc quad_perm:[1,0,3,2] ...: the value of the NEIGHBOUR lane (lane ^ 1)
                assert "quad_perm:[1,0,3,2]" in rest and "row_mask:0xf" in rest and "bank_mask:0xf" in rest, line
                yield (args[0], args[1].split()[0])
            elif op == "v_lshl_or_b32":
                self.wr(args[0], (self.rd(args[1]) << self.rd(args[2])) | self.rd(args[3]))
            elif op == "v_sub_u32_e32":
                self.wr(args[0], self.rd(args[1]) - self.rd(args[2]))
            elif op == "v_subrev_u32_e32":
                self.wr(args[0], self.rd(args[2]) - self.rd(args[1]))
            elif op == "v_max_i32_e32":
                self.wr(args[0], max(s32(self.rd(args[1])), s32(self.rd(args[2]))))
            elif op == "v_cmp_gt_i32_e64":
                self.wr_carry(args[0], 1 if s32(self.rd(args[1])) > s32(self.rd(args[2])) else 0)
            elif op == "v_lshlrev_b32_e32":
                self.wr(args[0], self.rd(args[2]) << self.rd(args[1]))
            elif op == "v_alignbit_b32":
                t = (self.rd(args[1]) << 32) | self.rd(args[2])
                self.wr(args[0], t >> self.rd(args[3]))
            elif op in ("v_or_b32_e32", "v_or_b32_e64"):
                self.wr(args[0], self.rd(args[1]) | self.rd(args[2]))
            elif op == "v_or3_b32":
                self.wr(args[0], self.rd(args[1]) | self.rd(args[2]) | self.rd(args[3]))
            elif op == "v_perm_b32":                 # byte i of the result = byte sel[i] of {src0 (4..7), src1 (0..3)}
                pool = self.rd(args[2]) | (self.rd(args[1]) << 32)
                sel = self.rd(args[3])
                r = 0
                for i in range(4):
                    q = (sel >> (8 * i)) & 0xFF
                    assert q < 8, "v_perm_b32 selector outside the plain byte range"
                    r |= ((pool >> (8 * q)) & 0xFF) << (8 * i)
                self.wr(args[0], r)
            elif op == "s_orn2_b64":
                self.wr_carry(args[0], self.rd(args[1]) | (1 - self.rd(args[2])))
            elif op == "s_xor_b64":
                self.wr_carry(args[0], self.rd(args[1]) ^ self.rd(args[2]))
            elif op == "s_andn2_b64":
                self.wr_carry(args[0], self.rd(args[1]) & (1 - self.rd(args[2])))
            elif op == "v_cmp_eq_u32_e64":
                self.wr_carry(args[0], 1 if self.rd(args[1]) == self.rd(args[2]) else 0)
            elif op == "v_cmp_ne_u32_e64":
                self.wr_carry(args[0], 1 if self.rd(args[1]) != self.rd(args[2]) else 0)
            elif op == "s_not_b64":
                self.wr_carry(args[0], 1 - self.rd(args[1]))
            elif op == "s_or_b64":
                self.wr_carry(args[0], self.rd(args[1]) | self.rd(args[2]))
            elif op == "s_and_b64":
                self.wr_carry(args[0], self.rd(args[1]) & self.rd(args[2]))
            elif op == "ds_read2st64_b32":       # ds_read2st64_b32 v[d:d+1], vaddr offset0:x offset1:y
                m = re.fullmatch(r"(v\[\d+:\d+\]|a\[\d+:\d+\]), (v\d+)(?: offset0:(\d+))?(?: offset1:(\d+))?", rest.strip())
                dst, addr = m.group(1), self.rd(m.group(2))
                o0, o1 = int(m.group(3) or 0), int(m.group(4) or 0)
                lo = int(re.match(r"[va]\[(\d+)", dst).group(1))
                bank = self.v if dst[0] == "v" else self.a
                bank[lo] = self.lds[addr + o0 * 256]; bank[lo + 1] = self.lds[addr + o1 * 256]
            elif op == "ds_write2st64_b32":      # ds_write2st64_b32 vaddr, vd0, vd1 offset0:x offset1:y
                m = re.fullmatch(r"(v\d+), ([va]\d+), ([va]\d+)(?: offset0:(\d+))?(?: offset1:(\d+))?", rest.strip())
                addr = self.rd(m.group(1))
                o0, o1 = int(m.group(4) or 0), int(m.group(5) or 0)
                self.lds[addr + o0 * 256] = self.rd(m.group(2)); self.lds[addr + o1 * 256] = self.rd(m.group(3))
            elif op == "v_add_f32_e64":
                self.wr(args[0], f32_bits(bits_f32(self.rd(args[1])) + bits_f32(self.rd(args[2]))))
            elif op == "s_mul_i32":
                self.wr(args[0], self.rd(args[1]) * self.rd(args[2]))
            elif op == "s_mul_hi_u32":
                self.wr(args[0], (self.rd(args[1]) * self.rd(args[2])) >> 32)
            elif op == "s_add_u32":
                t = self.rd(args[1]) + self.rd(args[2])
                self.wr(args[0], t); self.scc = t >> 32
            elif op == "s_addc_u32":
                t = self.rd(args[1]) + self.rd(args[2]) + self.scc
                self.wr(args[0], t); self.scc = t >> 32
            elif op in ("global_load_dword", "global_load_dwordx4") and args[2].startswith("off"):
                # global_load_dword[x4] vdst, v[lo:hi], off [offset:n]   (64-bit address in a VGPR pair)
                m = re.fullmatch(r"off(?: offset:(\d+))?", args[2])
                addr = self.rd(args[1]) + int(m.group(1) or 0)
                lo = int(re.match(r"v\[?(\d+)", args[0]).group(1))
                for q in range(4 if op.endswith("x4") else 1):
                    self.v[lo + q] = self.mem[addr + 4 * q]
            elif op == "v_lshlrev_b64":
                self.wr(args[0], (self.rd(args[2]) << self.rd(args[1])) & 0xFFFFFFFFFFFFFFFF)
            elif op == "v_cmp_gt_u32_e64":
                self.wr_carry(args[0], 1 if self.rd(args[1]) > self.rd(args[2]) else 0)
            elif op == "v_cmp_le_u32_e64":
                self.wr_carry(args[0], 1 if self.rd(args[1]) <= self.rd(args[2]) else 0)
            elif op in ("global_load_dword", "global_store_dword"):
                # global_load_dword vdst, voffset, s[lo:hi]  /  global_store_dword voffset, vdata, s[lo:hi]   (SADDR form)
                m = re.fullmatch(r"s\[(\d+):(\d+)\]", args[2])
                base = self.s[int(m.group(1))] | (self.s[int(m.group(2))] << 32)
                if op == "global_load_dword":
                    self.wr(args[0], self.mem[base + self.rd(args[1])])
                else:
                    self.mem[base + self.rd(args[0])] = self.rd(args[1])
            elif op in ("s_waitcnt", "s_nop"):
                pass
            else:
                raise NotImplementedError(line)

    @staticmethod
    def _split(rest):
        out, depth, cur = [], 0, ""
        for ch in rest:
            if ch == "[":
                depth += 1
            if ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                out.append(cur); cur = ""
            else:
                cur += ch
        if cur.strip():
            out.append(cur)
        return out


def run_pair(ma, mb, lines):
    """two lanes of one quad pair (lanes 2 i and 2 i + 1) in lockstep: the same instruction stream on both machines; at every v_mov_b32_dpp
    with quad_perm [1,0,3,2] each lane receives the other's source register. The machines may share their `lds` / `mem` dictionaries (the
    two-lane routines give both lanes of an item the same LDS column and workspace item)."""
    DONE = ("done",)
    ma.pair_sync = mb.pair_sync = True           # the lanes of a wave execute every instruction together: a load of lane B must not see the store
    ga, gb = ma.run_gen(lines), mb.run_gen(lines)      # that lane A issues LATER in the stream, so the two machines meet at every memory instruction
    while True:
        ea, eb = next(ga, DONE), next(gb, DONE)
        assert ea == eb, "the two lanes diverged"
        if ea is DONE:
            ma.pair_sync = mb.pair_sync = False
            return
        if ea is None:
            continue
        dst, src = ea
        va, vb = ma.rd(src), mb.rd(src)
        ma.wr(dst, vb); mb.wr(dst, va)


def f32_bits(x):
    import struct
    return struct.unpack("<I", struct.pack("<f", x))[0]


def bits_f32(b):
    import struct
    return struct.unpack("<f", struct.pack("<I", b & M32))[0]


def s32(x):
    x &= M32
    return x - (1 << 32) if x >> 31 else x


def s64(x):
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >> 63 else x


def digits_signed(x, dmax=None, rng=None):
    """a 14-digit signed representation of the integer x (|x| < 2^395): canonical digits, optionally randomised redundantly so that
    digits reach magnitudes up to dmax while the value stays x"""
    neg = x < 0
    ax = -x if neg else x
    d = [(ax >> (28 * i)) & 0xFFFFFFF for i in range(13)] + [ax >> 364]
    if neg:
        d = [-v for v in d]
    if dmax and rng:
        for i in range(13):
            c = rng.randrange(-(dmax >> 28), (dmax >> 28) + 1)      # move c * 2^28 from digit i+1 to digit i
            if abs(d[i] + c * (1 << 28)) < dmax and abs(d[i + 1] - c) < (1 << 31):
                d[i] += c * (1 << 28); d[i + 1] -= c
    assert sum(v << (28 * i) for i, v in enumerate(d)) == x
    return [v & M32 for v in d]


def from_digits_signed(d):
    return sum(s32(v) << (28 * i) for i, v in enumerate(d))


def limbs(x):
    return [(x >> (32 * i)) & M32 for i in range(12)]


def from_limbs(l):
    return sum(int(w) << (32 * i) for i, w in enumerate(l))
