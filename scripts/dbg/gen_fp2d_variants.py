#!/usr/bin/env python3
"""writes scripts/dbg/fp2d_variants.inc for fp2dbench.hip: the Fp2 product as four plain scans and as three products (Karatsuba on
the column sums), plus the constant loads"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools"))
import gen_fpd_asm as d
from gen_fp_asm import emit
txt = emit("FP2_MUL_D_OLD", [".p2align 6"] + d.fp2_mul_d4_body()) + "\n" + emit("FP2_MUL_D_KARA", [".p2align 6"] + d.fp2_mul_d_body()) + "\n" + emit("LOAD_CONST", d.load_constants()) + "\n"
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fp2d_variants.inc"), "w").write(txt)
