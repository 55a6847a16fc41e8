"""dev probe: the reference's criterion shape (10 sets x 3 keys) through the Python mirror a few times (to be run under rocprofv3 --kernel-trace)"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import milagro_bls_amd as m
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
rnd = random.Random(20)
n, mk = 10, 3
sets = []
for i in range(n):
    sks = [m.SecretKey.from_bytes(rnd.randrange(1, R).to_bytes(32, "big")) for _ in range(mk)]
    msg = rnd.randbytes(32)
    agg = m.AggregateSignature.aggregate([m.Signature.new(msg, s) for s in sks])
    apk = m.AggregatePublicKey.aggregate([m.PublicKey.from_secret_key(s) for s in sks])
    sets.append((agg, apk, msg))
for _ in range(5):
    t = time.perf_counter()
    ok = m.AggregateSignature.verify_multiple_aggregate_signatures(random.Random(1), sets)
    print(ok, round((time.perf_counter() - t) * 1e3, 2), "ms", flush=True)
