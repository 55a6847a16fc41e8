#!/usr/bin/env python3
"""Count Fp multiplications + squarings per item in the lane bodies (host emulation of the kernels) and write
profiles/fpmul_census.json, which bench.py uses for the integer-ALU roofline. Test infrastructure."""
import ctypes as C, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import helpers
E = helpers.load_emulator()
out = {}
for k, fmt, name in ((128, 1, "k128_uncompressed"), (128, 0, "k128_compressed"), (1, 0, "k1_compressed")):
    n = 4
    b = helpers.make_batch(n, k, fmt=fmt, seed=5, pool_n=max(64, 2 * k), negatives=False)
    m, s = C.c_uint64(), C.c_uint64()
    E.emul_op_counts(C.byref(m), C.byref(s), 1)
    res = helpers.ob(n); st = (C.c_uint32 * n)()
    E.emul_verify_batch(helpers.cb(b.sigs), helpers.cb(b.msgs), 32, helpers.cb(b.pks), fmt, None, C.c_uint64(n), k, 0 if k > 1 else 1, res, st)
    ph = (C.c_uint64 * 5)(); E.emul_phase_counts(ph)
    E.emul_op_counts(C.byref(m), C.byref(s), 1)
    assert all(bytes(res)[:n])
    cum = [0] + list(ph)
    out[name + "_per_phase"] = {nm: (cum[i + 1] - cum[i]) / n for i, nm in enumerate(("aggregate", "sig", "hash", "miller", "final"))}
    out[name] = (m.value + s.value) / n
    out[name + "_detail"] = {"fp_mul": m.value / n, "fp_sqr": s.value / n}
    print(name, out[name], out[name + "_detail"])
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "fpmul_census.json"), "w") as f:
    json.dump(out, f, indent=1)
