"""experiment: pairing check one wave per item at several LDS allocations (= waves per CU); run with MBLS_COOP_LDS_MIN set"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
nmax, k = 1 << 13, 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, nmax, k, N.PK_UNCOMPRESSED, rank=21)
ctx.reserve(nmax)
BIG = 1 << 62
out = {}
for n in (2048, 4096, 8192):
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    def f():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k, d_res.data_ptr(), None, None, None))
        torch.cuda.synchronize()
    for name, pp in (("x1", BIG), ("x2", 0)):
        ctx.set_coop_max_items(BIG); ctx.set_coop_hash_max_items(0); ctx.set_coop_packing(pp, BIG, BIG)
        f(); f(); ts = []
        for _ in range(5):
            t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
        out[(n, name)] = round(float(np.median(ts)), 2)
print(os.environ.get("MBLS_COOP_LDS_MIN"), out)
