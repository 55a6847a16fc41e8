#!/usr/bin/env python3
"""Write the include files millerbench.hip needs (variants of the generated routines), next to this script."""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "..", "tools"))
import gen_fp_asm as g, gen_tower_asm as t
lines, _ = t.build("miller_dbl")
nocall = t.miller_dbl_shell([l for l in lines if not l.startswith("CALL")])
callonly = t.miller_dbl_shell(t.expand_calls([l for l in lines if l.startswith("CALL") or l.startswith("s_waitcnt")]))
open(os.path.join(here, "miller_variants.inc"), "w").write(g.emit("MILLER_NOCALL", nocall) + "\n" + g.emit("MILLER_CALLONLY", callonly) + "\n")
out = ""
for nm, sym in (("ONLY_MUL", "mbls_fp2_mul_asm_fn"), ("ONLY_SQR", "mbls_fp2_sqr_asm_fn"), ("ONLY_MULFP", "mbls_fp2_mulfp_asm_fn")):
    out += g.emit(nm, t.wrap_loop(t.expand_calls(["CALL " + sym] * 60), count_sgpr="s39", prologue=["s_mov_b32 s39, s38"])) + "\n"
open(os.path.join(here, "call_variants.inc"), "w").write(out)
open(os.path.join(here, "mac28.inc"), "w").write(g.emit("MUL28", g.fp2_mul_body()) + "\n" + g.emit("SQR28", g.fp2_sqr_body()) + "\n" + g.emit("MULFP28", g.fp2_mulfp_body()) + "\n")
print("wrote miller_variants.inc call_variants.inc mac28.inc")
