"""Latency of the device-resident fast_aggregate_verify entry at small batch sizes with the one-wave-per-item pairing check on and off
(dev script)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
nmax, k = 1 << 14, 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, nmax, k, N.PK_UNCOMPRESSED, rank=21)
ctx.reserve(nmax)


def med(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


out = {}
for n in (1, 64, 256, 1024, 2048, 4096, 8192, 16384):
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)

    def f_dev():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              d_res.data_ptr(), None, None, None))
        torch.cuda.synchronize()
    row = {}
    for name, lim in (("coop", 1 << 20), ("lane", 0)):
        ctx.set_coop_max_items(lim)
        row[name] = med(f_dev)
        assert torch.equal(d_res.cpu(), expect[:n]), (name, n)
    out[n] = row
    print(n, row, flush=True)
print(json.dumps(out))
