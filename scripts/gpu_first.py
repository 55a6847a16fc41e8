"""First GPU contact: parity probes against the oracle + first timings. Test/dev script (uses the oracle as checker)."""
import ctypes as C, os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import orc
import milagro_bls_amd as mb
from milagro_bls_amd import batch, _native as N

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
random.seed(5)
ctx = N.default_context()
t0 = time.time()
n = 1024
A = [random.randrange(P) for _ in range(n)]; B = [random.randrange(P) for _ in range(n)]
a = b"".join(x.to_bytes(48, "big") for x in A); b = b"".join(x.to_bytes(48, "big") for x in B)
o = batch.fp_mul_batch(a, b, n)
bad = sum(int.from_bytes(o[48*i:48*i+48], "big") != A[i]*B[i] % P for i in range(n))
o2 = batch.fp_mul_batch(a, b, n, square=True)
bad2 = sum(int.from_bytes(o2[48*i:48*i+48], "big") != A[i]*A[i] % P for i in range(n))
print("fp_mul mismatches", bad, "fp_sqr mismatches", bad2, "t=%.1f" % (time.time()-t0), flush=True)

msgs = [random.randbytes(32) for _ in range(8)]
h = batch.hash_to_g2_batch(b"".join(msgs), 8)
print("hash parity", all(h[96*i:96*i+96] == orc.g2_compress(orc.hash_to_g2(m)) for i, m in enumerate(msgs)), flush=True)
sks = [random.randrange(1, R) for _ in range(8)]
pk = batch.sk_to_pk_batch(b"".join(s.to_bytes(32, "big") for s in sks), 8)
print("sk_to_pk parity", all(pk[48*i:48*i+48] == orc.g1_compress(orc.sk_to_pk(s)) for i, s in enumerate(sks)), flush=True)
sg = batch.sign_batch(b"".join(s.to_bytes(32, "big") for s in sks), b"".join(msgs), 8)
print("sign parity", all(sg[96*i:96*i+96] == orc.g2_compress(orc.sign(m, s)) for i, (m, s) in enumerate(zip(msgs, sks))), flush=True)

def make_batch(Nn, K, pool_n=256, fmt=0):
    pool = [random.randrange(1, R) for _ in range(pool_n)]
    pkb = orc.batch_sk_to_pk(b"".join(s.to_bytes(32, "big") for s in pool), pool_n, fmt, nthreads=8)
    sz = 48 if fmt == 0 else 96
    msgs = random.randbytes(32 * Nn)
    aggs = []; pks = []
    for i in range(Nn):
        idx = random.sample(range(pool_n), K)
        aggs.append(sum(pool[j] for j in idx) % R)
        pks.append(b"".join(pkb[sz*j:sz*j+sz] for j in idx))
    sigs = orc.batch_sign(b"".join(x.to_bytes(32, "big") for x in aggs), msgs, Nn, nthreads=8)
    sigs = bytearray(sigs); msgs = bytearray(msgs)
    for i in range(Nn):
        if i % 16 == 7: msgs[32*i] ^= 1
    return bytes(sigs), bytes(msgs), b"".join(pks)

for (Nn, K, fmt) in [(64, 4, 0), (200, 128, 1), (128, 128, 0)]:
    sigs, msgs_, pks = make_batch(Nn, K, fmt=fmt)
    exp = orc.batch_fast_aggregate_verify(sigs, msgs_, pks, Nn, K, fmt, nthreads=8)
    t = time.time()
    res, st = batch.fast_aggregate_verify_batch(sigs, msgs_, pks, Nn, K, pk_format=fmt)
    print("verify N=%d K=%d fmt=%d parity=%s accepted=%d t=%.2fs" % (Nn, K, fmt, res == exp, sum(res), time.time()-t), flush=True)
    if res != exp:
        print("  exp", exp[:16], "\n  got", res[:16], st[:16])

# ALU calibration
for lanes in (65536, 524288):
    ms = batch.fp_mul_bench(lanes, 2000)
    print("fp_mul_bench lanes=%d iters=2000: %.2f ms -> %.3e Fp mul/s" % (lanes, ms, lanes*2000/(ms*1e-3)), flush=True)

# phase timing at a moderate size
import torch
Nn, K = 16384, 128
sigs, msgs_, pks = make_batch(Nn, K, fmt=1)
dev = torch.device("cuda:0")
d_s = torch.frombuffer(bytearray(sigs), dtype=torch.uint8).to(dev); d_m = torch.frombuffer(bytearray(msgs_), dtype=torch.uint8).to(dev)
d_p = torch.frombuffer(bytearray(pks), dtype=torch.uint8).to(dev)
d_r = torch.zeros(Nn, dtype=torch.uint8, device=dev); d_b = torch.zeros((Nn+63)//64, dtype=torch.int64, device=dev)
ctx.reserve(Nn)
N.lib().mbls_enable_phase_timing(ctx.handle, 1)
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    rc = N.lib().mbls_fast_aggregate_verify_batch_device(ctx.handle, d_s.data_ptr(), d_m.data_ptr(), 32, None, d_p.data_ptr(), 1, None, Nn, K,
                                                         d_r.data_ptr(), d_b.data_ptr(), None, None)
    torch.cuda.synchronize(); dt = time.time() - t
    ms = (C.c_float * 6)(); N.lib().mbls_last_phase_ms(ctx.handle, ms)
    print("N=%d K=%d uncompressed rc=%d total %.1f ms -> %.0f verif/s; phases(ms) %s accepted=%d" % (
        Nn, K, rc, dt*1e3, Nn/dt, ["%s=%.1f" % (nm, v) for nm, v in zip(N.PHASE_NAMES, ms)], int(d_r.sum())), flush=True)
