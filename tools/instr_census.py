#!/usr/bin/env python3
"""Exact instruction counts of the generated routines (the same generator code that emits them) -> profiles/instr_census.json.

Per routine: VALU / multiply-accumulate (v_mad_u64_u32 or v_mad_i64_i32) / SALU / memory instruction counts with the nested
multiplication calls expanded; per item: the dynamic counts of the generated part of the two dominant kernels (k_miller: the whole
digit-form loop -- prologue, the first iteration (f = 1), 62 doubling iterations, 5 addition steps (both pairs in one body), epilogue; k_final: the whole
final exponentiation routine -- easy part with its Fp inversion, 5 powers by |x| of 63 compressed squarings, 6 saves, a joint
decompression and 5 multiplications each, the products between them, epilogue). What the compiler schedules around them (k_miller's point set-up, the comparison with one) is not
counted. bench.py turns these into the valu_issue figure. Run:  python3 tools/instr_census.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_fp_asm as F  # noqa: E402
import gen_fpd_asm as FD  # noqa: E402
import gen_tower_d as TD  # noqa: E402


# VALU classes by how their issue rate depends on occupancy (profiles/r02_ubench.txt, scripts/dbg/ubench.hip): "simple2" = plain 32-bit two-operand operations and
# AGPR moves, which issue every ~2.3 clocks once TWO waves share a SIMD but every ~4.6 from one wave; "carry3" = everything else that is not a multiply-accumulate
# (64-bit shifts / moves / adds, carry chains, three-operand VOP3 forms, v_mul_lo_u32, compares, selections, conversions): ~4.2-4.9 clocks at any occupancy, like the
# multiply-accumulate itself. `valu` = mad_u64_u32 + carry3 + simple2.
SIMPLE2 = ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32",
           "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_max_i32", "v_min_i32", "v_max_u32", "v_min_u32", "v_not_b32", "v_add_f32", "v_mul_f32")


def classify(lines):
    c = dict(valu=0, mad_u64_u32=0, carry3=0, simple2=0, salu=0, lds=0, lds_only=0, global_mem=0)
    for l in lines:
        l = l.strip()
        if not l or l.startswith(".") or l.endswith(":") or l.startswith("CALL "):
            continue
        op = l.split()[0]
        base = op[:-4] if op.endswith(("_e32", "_e64")) else op
        if l.startswith("v_mad_u64_u32") or l.startswith("v_mad_i64_i32"):       # the 32 x 32 + 64 multiply-accumulate, either signedness
            c["mad_u64_u32"] += 1; c["valu"] += 1
        elif l.startswith("v_"):
            c["valu"] += 1
            c["simple2" if base in SIMPLE2 else "carry3"] += 1
        elif l.startswith("ds_") or l.startswith("global_"):
            c["lds"] += 1                                                       # memory instructions (LDS and the workspace loads / stores)
            c["lds_only" if l.startswith("ds_") else "global_mem"] += 1
        elif l.startswith("s_"):
            c["salu"] += 1
        else:
            raise ValueError(l)
    return c


# What a second wave per SIMD could buy, COMPUTED from the class counts instead of guessed: clocks per wave-instruction with one wave / with two or more
# (profiles/r02_ubench.txt; the multiply-accumulate in the scan pattern: scripts/dbg/icbench.hip) -- an upper bound that ignores what halving the registers and the
# LDS space per wave would add in spill traffic.
CLK_ONE_WAVE = dict(mad_u64_u32=4.4, carry3=4.7, simple2=4.6)
CLK_TWO_WAVES = dict(mad_u64_u32=4.4, carry3=4.2, simple2=2.3)


def two_wave_model(c):
    t1 = sum(c[k] * CLK_ONE_WAVE[k] for k in CLK_ONE_WAVE)
    t2 = sum(c[k] * CLK_TWO_WAVES[k] for k in CLK_TWO_WAVES)
    return {"clocks_per_item_one_wave": round(t1), "clocks_per_item_two_waves": round(t2), "upper_bound_gain": round(1 - t2 / t1, 4),
            "share_mad": round(c["mad_u64_u32"] / c["valu"], 4), "share_carry3": round(c["carry3"] / c["valu"], 4), "share_simple2": round(c["simple2"] / c["valu"], 4)}


def add(a, b, times=1):
    return {k: a[k] + times * b[k] for k in a}


def with_calls(lines, leaf):
    c = classify(lines)
    for l in lines:
        if l.startswith("CALL "):
            c = add(c, leaf[l.split()[1]])
            c["salu"] += 4           # s_getpc / s_add / s_addc / s_swappc
    return c


def main():
    leaf = {"mbls_fp2_mul_asm_fn": classify(F.fp2_mul_body()), "mbls_fp2_sqr_asm_fn": classify(F.fp2_sqr_body()),
            "mbls_fp2_mulfp_asm_fn": classify(F.fp2_mulfp_body())}
    leaf.update({name: classify(fn()) for name, fn in FD.ROUTINE_BODIES.items()})
    routines = dict(leaf)
    inv = classify(F.fp_inv_gcd_body(unrolled=True))                      # the Fp inversion of the easy part (its two counted loops unrolled)
    inv["salu"] += 3 * 30 * 31
    routines["fp_inv_gcd"] = inv
    leaf["mbls_fp_inv_gcd_asm_fn"] = inv
    # second generation (digit form): what the kernels run
    _, xp, _ = TD.final_exp_d_routine()
    for k in xp:
        routines["final_exp_d_" + k] = with_calls(xp[k], leaf)
    _, pieces, _ = TD.miller_loop_d_routine()
    for k in ("pro", "first", "dbl", "add01", "epi"):
        routines["miller_d_" + k] = with_calls(pieces[k], leaf)
    _, gp, _ = TD.g2_dbl_d_routine()
    routines["g2_dbl_d"] = with_calls(gp["body"], leaf)
    zero = dict(valu=0, mad_u64_u32=0, carry3=0, simple2=0, salu=0, lds=0, lds_only=0, global_mem=0)
    km = add(add(add(zero, routines["miller_d_first"]), routines["miller_d_dbl"], 62), routines["miller_d_add01"], 5)
    km = add(add(km, routines["miller_d_pro"]), routines["miller_d_epi"])
    kf = zero
    for k, times in (("pro", 1), ("easy", 1), ("pstart", 5), ("csqr", 315), ("psave", 30), ("pinv", 5), ("pfirst", 5), ("pmul", 25),
                     ("step_conj", 2), ("step_frob", 1), ("step_base", 1), ("tail", 1), ("epi", 1)):
        kf = add(kf, routines["final_exp_d_" + k], times)
    # the public-key sum (128 keys per item; 96-byte keys / table indices): prologue, per key fetch + decode + step + status, epilogue
    per_item = {"k_miller": km, "k_final": kf}
    for mode, kern in (("raw", "k_aggregate"), ("indexed", "k_aggregate_indexed")):
        _, gp, _ = TD.g1_aggregate_d_routine(mode)
        for k in gp:
            routines["g1_sum_%s_%s" % (mode, k)] = with_calls(gp[k], leaf)
        ka = add(routines["g1_sum_%s_pro" % mode], routines["g1_sum_%s_epi" % mode])
        for k in ("decode", "nxt", "step", "post"):
            ka = add(ka, routines["g1_sum_%s_%s" % (mode, k)], 128)
        per_item[kern] = ka
    out = {"_note": "generated by tools/instr_census.py from tools/gen_fp_asm.py, gen_fpd_asm.py, gen_tower_d.py; calls expanded; cold compiler-scheduled paths excluded; "
                    "valu = mad_u64_u32 + carry3 + simple2, lds = lds_only + global_mem (classes: see the tool)",
           "routines": routines, "per_item": per_item,
           "two_wave_model": dict({k: two_wave_model(v) for k, v in per_item.items()},
                                  _assumptions={"clocks_per_wave_instruction_one_wave": CLK_ONE_WAVE, "clocks_per_wave_instruction_two_or_more_waves": CLK_TWO_WAVES,
                                                "source": "profiles/r02_ubench.txt, scripts/dbg/icbench.hip", "ignores": "spill traffic of a 256-register allocation",
                                                "measured_instead": "profiles/r05_two_wave_mix.txt (scripts/dbg/mixbench.hip): the same mixes at 2 waves per SIMD gain 4.1-4.7 %, "
                                                                    "not the 12-13 % of this class model -- a lone wave already issues the mixed stream at ~3.9 clocks per instruction"})}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "instr_census.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(per_item))


if __name__ == "__main__":
    main()
