#!/usr/bin/env python3
"""writes scripts/dbg/mixbench_gen.inc for mixbench.hip: the digit-form Fp2 product routine followed by blocks of the OTHER instruction classes of the
generated kernels (tools/instr_census.py: carry3 = 64-bit / carry / three-operand forms, simple2 = plain 32-bit two-operand operations and AGPR moves) in the
proportions of k_miller and k_final -- the instruction MIX of those kernels without their register pressure, to measure what a second wave per SIMD buys."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools"))
import gen_fpd_asm as d
from gen_fp_asm import emit


def simple2(n):
    """n independent plain operations on v112..v139 / a0..a27 (digit-wise additions, subtractions, masks, AGPR moves: what the linear parts of the routines are)"""
    L = []
    pat = ["v_add_u32_e64 v%d, v%d, v%d", "v_sub_u32_e64 v%d, v%d, v%d", "v_accvgpr_write_b32 a%d, v%d", "v_accvgpr_read_b32 v%d, a%d", "v_and_b32_e64 v%d, v%d, s65"]
    for i in range(n):
        k, j = i % 5, i % 14
        if k < 2:
            L.append(pat[k] % (112 + j, 112 + j, 126 + j))
        elif k == 2:
            L.append(pat[k] % (j, 126 + j))
        elif k == 3:
            L.append(pat[k] % (126 + j, 14 + j))
        else:
            L.append(pat[k] % (112 + j, 126 + j))
    return L


def carry3(n):
    L = []
    pat = ["v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]", "v_ashrrev_i64 v[%d:%d], 1, v[%d:%d]", "v_mov_b64_e64 v[%d:%d], v[%d:%d]"]
    for i in range(n):
        k, j = i % 3, 2 * (i % 7)
        a, b = 112 + j, 126 + j
        if k == 0:
            L.append(pat[0] % (a, a + 1, a, a + 1, b, b + 1))
        elif k == 1:
            L.append(pat[1] % (b, b + 1, b, b + 1))
        else:
            L.append(pat[2] % (a, a + 1, b, b + 1))
    return L


txt = emit("LOAD_CONST", d.load_constants()) + "\n" + emit("FP2_MUL_D", [".p2align 6"] + d.fp2_mul_d_body()) + "\n"
txt += emit("MIX_MILLER", carry3(25) + simple2(205)) + "\n" + emit("MIX_FINAL", simple2(263)) + "\n"
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mixbench_gen.inc"), "w").write(txt)
