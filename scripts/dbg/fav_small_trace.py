"""dev probe: one fast_aggregate_verify (128 keys) and one Signature::verify through the Python mirror a few times (to be run under rocprofv3 --kernel-trace)"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import milagro_bls_amd as m
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
rnd = random.Random(21)
sks = [m.SecretKey.from_bytes(rnd.randrange(1, R).to_bytes(32, "big")) for _ in range(128)]
pks = [m.PublicKey.from_secret_key(s) for s in sks]
msg = rnd.randbytes(32)
agg = m.AggregateSignature.aggregate([m.Signature.new(msg, s) for s in sks])
sig1 = m.Signature.new(msg, sks[0])
which = sys.argv[1] if len(sys.argv) > 1 else "fav"
for _ in range(5):
    t = time.perf_counter()
    ok = agg.fast_aggregate_verify(msg, pks) if which == "fav" else sig1.verify(msg, pks[0])
    print(which, ok, round((time.perf_counter() - t) * 1e3, 2), "ms", flush=True)
