"""The operations either side of the verification path at batch size 2^16: compressed-key decode with and
without KeyValidate, secret key -> public key, signing. Dev script; writes gpurun_out/keyops.json (profiles/rNN_keyops.json)."""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from milagro_bls_amd import _native as N, batch
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
rnd = random.Random(11)
out = {"_unit": "ms for n = %d, median of 5 calls after 2 warm-ups" % n, "n": n}


def med(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


sks = b"".join(rnd.randrange(1, R).to_bytes(32, "big") for _ in range(n)); msgs = rnd.randbytes(32 * n)
d_sk = torch.frombuffer(bytearray(sks), dtype=torch.uint8).to(dev); d_msg = torch.frombuffer(bytearray(msgs), dtype=torch.uint8).to(dev)
d_pk = torch.zeros(48 * n, dtype=torch.uint8, device=dev); d_sig = torch.zeros(96 * n, dtype=torch.uint8, device=dev)
out["sk_to_pk_device_ms"] = med(lambda: ctx.check(lib.mbls_sk_to_pk_batch_device(ctx.handle, d_sk.data_ptr(), 0, n, d_pk.data_ptr(), None)))
out["sign_device_ms"] = med(lambda: ctx.check(lib.mbls_sign_batch_device(ctx.handle, d_sk.data_ptr(), d_msg.data_ptr(), 32, n, d_sig.data_ptr(), None)))
pks = bytes(d_pk.cpu().numpy())
import ctypes as C
c_in = (C.c_uint8 * (48 * n)).from_buffer_copy(pks); c_out = (C.c_uint8 * (96 * n))(); c_err = (C.c_uint8 * n)()
for name, val in (("pk_decode_keyvalidate_host_ms", 1), ("pk_decode_unchecked_host_ms", 0)):
    out[name] = med(lambda: ctx.check(lib.mbls_pk_decode_batch(ctx.handle, c_in, N.PK_COMPRESSED, val, n, c_out, c_err)))
    assert not any(bytes(c_err))
# the signatures verify against the keys (Signature::verify, reference src/signature.rs:27-40)
res, st = batch.verify_batch(bytes(d_sig.cpu().numpy()), msgs, pks, n, pk_format=N.PK_COMPRESSED)
assert all(res), "a device-made signature does not verify"
out["per_s"] = {k[:-3]: n / v * 1e3 for k, v in out.items() if k.endswith("_ms")}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/keyops.json", "w"), indent=1)
