"""Time the non-headline BASELINE configs on the GPU (dev script; numbers quoted in DESIGN.md)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from milagro_bls_amd import _native as N, batch
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
# config 2: 2^16 x Signature::verify (k = 1, compressed key)
n = 1 << 16
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, 1, N.PK_COMPRESSED, rank=11)
d_res = torch.zeros(n, dtype=torch.uint8, device=dev); d_bm = torch.zeros(n // 64, dtype=torch.int64, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.check(lib.mbls_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_COMPRESSED, n, d_res.data_ptr(), d_bm.data_ptr(), None, None))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print("config 2: 2^16 x verify: %.1f ms -> %.0f verify/s, correct=%s" % (dt * 1e3, n / dt, bool(torch.equal(d_res.cpu(), expect))))
# config 4: verify_multiple, 2^14 sets x 128 keys
n, k = 1 << 14, 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=12, negatives=False)
g = torch.Generator(device="cpu"); g.manual_seed(7)
rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    ok = batch.verify_multiple_sets_device(d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, k, pk_format=N.PK_UNCOMPRESSED)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print("config 4: verify_multiple 2^14 sets x 128 keys: %.1f ms -> %.0f sets/s, result=%s" % (dt * 1e3, n / dt, ok))
res = torch.zeros(n, dtype=torch.uint8, device=dev)
for it in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k, res.data_ptr(), None, None, None))
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t
print("  (same 2^14 sets through fast_aggregate_verify: %.1f ms)" % (dt2 * 1e3))
