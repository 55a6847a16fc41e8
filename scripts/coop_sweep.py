"""Where the wave-cooperative paths stop paying: latency of the device-resident entry over n for the pairing check one lane per item / one
wave per item / two items per wave, and the message phase one lane per item / one wave per item / four items per wave (dev script)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
nmax, k = 3 << 13, 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, nmax, k, N.PK_UNCOMPRESSED, rank=21)
ctx.reserve(nmax)


def med(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


BIG = 1 << 62
for n in (1, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 10240, 12288, 16384, 20480, 24576):
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)

    def f_dev():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              d_res.data_ptr(), None, None, None))
        torch.cuda.synchronize()
    row = {}
    # (pairing limit, hash limit, pairing packed above, hash packed above)
    for name, lp, lh, pp, ph in (("wave+wavehash", BIG, BIG, BIG, BIG), ("wave2+wavehash4", BIG, BIG, 0, 0), ("wave+wavehash4", BIG, BIG, BIG, 0), ("wave", BIG, 0, BIG, BIG),
                                 ("wave2", BIG, 0, 0, BIG), ("lane", 0, 0, BIG, BIG)):
        if n > 8192 and name in ("wave+wavehash", "wave+wavehash4"):
            continue
        ctx.set_coop_max_items(lp); ctx.set_coop_hash_max_items(lh); ctx.set_coop_packing(pp, BIG, ph)
        row[name] = round(med(f_dev), 2)
        assert torch.equal(d_res.cpu(), expect[:n]), (name, n)
    print(n, row, flush=True)
