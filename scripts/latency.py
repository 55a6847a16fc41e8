"""Latency of the entry points at small and large batch sizes (dev script; output committed as profiles/<tag>_latency.json).
Host-buffer entries (PCIe-inclusive: upload, kernels, download, synchronise) for fast_aggregate_verify at n = 1 .. 2^16 in the three
key representations, the scalar API (n = 1 objects), config 2 (2^16 x Signature::verify) and config 4 (verify_multiple, 2^14 sets x 128 keys
against the same sets verified one by one)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from milagro_bls_amd import _native as N, batch
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
out = {"_unit": "ms, median of 5 calls after 2 warm-up calls; host-buffer entries include PCIe transfers and the final synchronisation"}
nmax, k = 1 << 16, 128
d_sigs, d_msgs, d_pks, expect, d_idx, table = bench.build_inputs(ctx, dev, nmax, k, N.PK_UNCOMPRESSED, rank=21, return_indices=True)
sigs = d_sigs.cpu().numpy(); msgs = d_msgs.cpu().numpy(); pks = d_pks.cpu().numpy(); idx = d_idx.cpu().numpy()


def med(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


rows = {}
for n in (1, 64, 1024, 16384, 65536):
    res = (C.c_uint8 * n)()
    s_, m_, p_, i_ = sigs[:n].tobytes(), msgs[:n].tobytes(), pks[:n].tobytes(), np.ascontiguousarray(idx[:n]).ctypes.data_as(C.c_void_p)
    def f_bytes():
        ctx.check(lib.mbls_fast_aggregate_verify_batch(ctx.handle, s_, m_, 32, None, p_, N.PK_UNCOMPRESSED, None, n, k, res, None))
    def f_idx():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_indexed(ctx.handle, table.handle, s_, m_, 32, None, i_, None, n, k, res, None))
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    def f_dev():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              d_res.data_ptr(), None, None, None))
        torch.cuda.synchronize()
    def f_dev_idx():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_indexed_device(ctx.handle, table.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_idx.data_ptr(), None, n, k,
                                                                      d_res.data_ptr(), None, None, None))
        torch.cuda.synchronize()
    rows[str(n)] = {"host_bytes_ms": med(f_bytes), "host_indexed_ms": med(f_idx), "device_bytes_ms": med(f_dev), "device_indexed_ms": med(f_dev_idx)}
    assert bytes(res) == bytes(expect[:n].numpy().tobytes())
    print(n, rows[str(n)], flush=True)
out["fast_aggregate_verify_k128"] = rows
# scalar API: one object per call
from milagro_bls_amd import AggregateSignature, PublicKey, Signature, SecretKey
sk = SecretKey.from_bytes(bytes([1] * 32)); pk = PublicKey.from_secret_key(sk); msg = b"Some msg"; sig = Signature.new(msg, sk)
assert sig.verify(msg, pk)
out["scalar_api"] = {"Signature::verify": med(lambda: sig.verify(msg, pk)), "Signature::new": med(lambda: Signature.new(msg, sk)),
                     "PublicKey::from_bytes (decode + KeyValidate)": med(lambda: PublicKey.from_bytes(pk.as_bytes())),
                     "PublicKey::from_secret_key": med(lambda: PublicKey.from_secret_key(sk))}
keys = [PublicKey(bytes(pks[0, j])) for j in range(k)]
asig = AggregateSignature(bytes(sigs[0]))
assert asig.fast_aggregate_verify(bytes(msgs[0]), keys)
out["scalar_api"]["fast_aggregate_verify (128 keys)"] = med(lambda: asig.fast_aggregate_verify(bytes(msgs[0]), keys))
# AggregateSignature::aggregate_verify (distinct messages), 3 and 128 signers
import random as _r
_rnd = _r.Random(5)
for m in (3, 128):
    sks_ = [SecretKey.from_bytes(_rnd.randrange(1, 1 << 250).to_bytes(32, "big")) for _ in range(m)]
    pks_ = [PublicKey.from_secret_key(s_) for s_ in sks_]
    ms_ = [_rnd.randbytes(32) for _ in range(m)]
    ag = AggregateSignature.new()
    for s_, mm in zip(sks_, ms_):
        ag.add(Signature.new(mm, s_))
    assert ag.aggregate_verify(ms_, pks_)
    out["scalar_api"]["aggregate_verify (%d messages)" % m] = med(lambda: ag.aggregate_verify(ms_, pks_))
print(out["scalar_api"], flush=True)
# config 2: 2^16 x Signature::verify, compressed key, resident
n = 1 << 16
c_sigs, c_msgs, c_pks, c_exp = bench.build_inputs(ctx, dev, n, 1, N.PK_COMPRESSED, rank=11)
c_res = torch.zeros(n, dtype=torch.uint8, device=dev)
def f_c2():
    ctx.check(lib.mbls_verify_batch_device(ctx.handle, c_sigs.data_ptr(), c_msgs.data_ptr(), 32, None, c_pks.data_ptr(), N.PK_COMPRESSED, n, c_res.data_ptr(), None, None, None))
    torch.cuda.synchronize()
t = med(f_c2); assert torch.equal(c_res.cpu(), c_exp)
out["config2_verify_2_16"] = {"ms": t, "verify_per_s": n / t * 1e3}
# config 4: verify_multiple, 2^14 sets x 128 keys vs the same sets one by one
n = 1 << 14
v_sigs, v_msgs, v_pks, v_exp = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=12, negatives=False)
g = torch.Generator(device="cpu"); g.manual_seed(7)
rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
okv = []
def f_c4():
    okv.append(batch.verify_multiple_sets_device(v_sigs.data_ptr(), v_pks.data_ptr(), v_msgs.data_ptr(), rands.data_ptr(), n, k, pk_format=N.PK_UNCOMPRESSED))
t4 = med(f_c4); assert all(okv)
v_res = torch.zeros(n, dtype=torch.uint8, device=dev)
def f_c4b():
    ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, v_sigs.data_ptr(), v_msgs.data_ptr(), 32, None, v_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k, v_res.data_ptr(), None, None, None))
    torch.cuda.synchronize()
t4b = med(f_c4b)
out["config4_verify_multiple_2_14x128"] = {"ms": t4, "sets_per_s": n / t4 * 1e3, "same_sets_one_by_one_ms": t4b}
# the same comparison where throughput, not latency, decides: 2^16 sets (the chip is full in both forms)
n = 1 << 16
v_sigs, v_msgs, v_pks, v_exp = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=13, negatives=False)
rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
okv = []
t5 = med(f_c4, reps=3, warm=1); assert all(okv)
v_res = torch.zeros(n, dtype=torch.uint8, device=dev)
t5b = med(f_c4b, reps=3, warm=1)
out["verify_multiple_2_16x128"] = {"ms": t5, "sets_per_s": n / t5 * 1e3, "same_sets_one_by_one_ms": t5b}
print(json.dumps(out))
with open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/latency.json", "w") as f:
    json.dump(out, f, indent=1)
