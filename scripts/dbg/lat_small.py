"""latency of small batches (device entries, 128 keys): n = 1, 64, 512, 1024, 2048, 4096 + the scalar API's Signature::verify"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
k = 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, 4096, k, N.PK_UNCOMPRESSED, rank=21)
out = {}
for n in (1, 64, 512, 1024, 2048, 4096):
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    def f():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k, d_res.data_ptr(), None, None, None))
        torch.cuda.synchronize()
    f(); f(); ts = []
    for _ in range(9):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    assert torch.equal(d_res.cpu(), expect[:n])
    out[n] = round(float(np.median(ts)), 3)
from milagro_bls_amd import PublicKey, Signature, SecretKey
sk = SecretKey.from_bytes(bytes([1] * 32)); pk = PublicKey.from_secret_key(sk); msg = b"Some msg"; sig = Signature.new(msg, sk)
assert sig.verify(msg, pk)
ts = []
for _ in range(9):
    t = time.perf_counter(); sig.verify(msg, pk); ts.append((time.perf_counter() - t) * 1e3)
out["Signature::verify"] = round(float(np.median(ts)), 3)
print(out)
