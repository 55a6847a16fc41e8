"""Randomised verify_multiple batches against the CPU oracle (test infrastructure, like tests/): random sizes 1 .. 48 sets (the lane-pair signature chain) and a few
mid-size ones, random members replaced by a wrong / infinite / non-subgroup signature, an infinite or wrong key, a zero scalar -- through the one-call entry with the
caller's scalar source (result AND number of scalars asked for, reference src/aggregates.rs:272-287), the entry that takes the scalars, and the device entry.
usage: stress_vm.py [n_batches] [seed]"""
import ctypes as C, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import helpers
import orc
from milagro_bls_amd import _native as N
import json
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
vec = json.load(open(os.path.join(helpers.ROOT, "tests", "golden", "vectors.json")))
PROBE = bytes.fromhex(vec["model"]["g2_subgroup_probes"][0]["compressed"])
G1_INF_U = bytes([0x40]) + bytes(95)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
POOL = 64
sks = [rnd.randrange(1, helpers.R) for _ in range(POOL)]
pks = [orc.sk_to_pk(s) for s in sks]
msgs_pool = [rnd.randbytes(32) for _ in range(POOL)]
sig_pool = [orc.g2_compress(orc.sign(m, s)) for m, s in zip(msgs_pool, sks)]
bad = 0; sets = 0; t0 = time.time()
for it in range(count):
    n = rnd.choice([rnd.randrange(1, 12), rnd.randrange(1, 49), rnd.randrange(1, 49), rnd.randrange(49, 300)])
    idx = [rnd.randrange(POOL) for _ in range(n)]
    sigs = [sig_pool[i] for i in idx]; apks = [pks[i] for i in idx]; msgs = [msgs_pool[i] for i in idx]
    rands = [rnd.randrange(1, 1 << 64) for _ in range(n)]
    for _ in range(rnd.choice([0, 0, 1, 1, 2, 3])):
        j = rnd.randrange(n); what = rnd.randrange(7)
        if what == 0: sigs[j] = sig_pool[(idx[j] + 1) % POOL]
        elif what == 1: sigs[j] = helpers.G2_INF
        elif what == 2: sigs[j] = PROBE
        elif what == 3: apks[j] = G1_INF_U
        elif what == 4: apks[j] = pks[(idx[j] + 1) % POOL]
        elif what == 5: sigs[j] = helpers.G2_INF; apks[j] = G1_INF_U
        else: rands[j] = 0
    dec = [orc.g2_from_compressed(s) for s in sigs]
    first_bad = next((i for i, (e, p) in enumerate(dec) if e or not orc.g2_subgroup_check(p)), n)
    want = bool(orc.verify_multiple([(d[1], a, m) for d, a, m in zip(dec, apks, msgs)], rands)) if all(r for r in rands) else False
    rr = (C.c_uint64 * n)(*rands)
    got_plain = bool(lib.mbls_verify_multiple_aggregate_signatures(ctx.handle, N.cbuf(b"".join(sigs)), N.cbuf(b"".join(apks)), N.cbuf(b"".join(msgs)), 32, None, rr, n))
    asked = []

    def draw(_u, out, cnt):
        C.memmove(out, rr, 8 * cnt); asked.append(int(cnt))
    cb = N.SCALAR_SOURCE(draw)
    got_rng = bool(lib.mbls_verify_multiple_aggregate_signatures_rng(ctx.handle, N.cbuf(b"".join(sigs)), N.cbuf(b"".join(apks)), N.cbuf(b"".join(msgs)), 32, None, n, cb, None))
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_s, d_a, d_m = t(b"".join(sigs)), t(b"".join(apks)), t(b"".join(msgs))
    d_r = torch.tensor([r - (1 << 64) if r >> 63 else r for r in rands], dtype=torch.int64, device=dev)
    d_res = torch.full((8,), 7, dtype=torch.uint8, device=dev)
    ctx.check(lib.mbls_verify_multiple_aggregate_signatures_device(ctx.handle, d_s.data_ptr(), d_a.data_ptr(), d_m.data_ptr(), 32, None, d_r.data_ptr(), n, d_res.data_ptr(), None, None))
    torch.cuda.synchronize()
    got_dev = bool(int(d_res[0].item()) == 1)
    ok = got_plain == want and got_rng == want and got_dev == want
    if True:
        ok = ok and sum(asked) == first_bad and len(asked) == (1 if first_bad else 0)
    if not ok:
        bad += 1; print("MISMATCH n =", n, "want", want, "plain", got_plain, "rng", got_rng, "device", got_dev, "asked", asked, "first_bad", first_bad, flush=True)
    sets += n
print("batches", count, "sets", sets, "mismatching batches", bad, "%.1f s" % (time.time() - t0))
