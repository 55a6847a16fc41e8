"""BASELINE configs[3]: verify_multiple_aggregate_signatures, 2^14 (and 2^16) sets x 128 keys, against the same sets verified one by one
(dev script; profiles/r03_config4.json)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from milagro_bls_amd import _native as N, batch
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
k = 128
sizes = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["16384", "65536"])]
out = {"_unit": "ms, median of 5 enqueue + synchronise calls after 2 warm-ups; inputs resident in HBM"}


def med(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


for n in sizes:
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=12, negatives=False)
    ctx.reserve(n)
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
    d_r = torch.full((8,), 7, dtype=torch.uint8, device=dev)

    def f_vm():
        batch.verify_multiple_sets_device(d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, k, pk_format=N.PK_UNCOMPRESSED, d_result=d_r.data_ptr())
    t = med(f_vm); assert int(d_r[0].item()) == 1
    d_msgs[n // 3, 5] ^= 0x10; f_vm(); torch.cuda.synchronize(); assert int(d_r[0].item()) == 0; d_msgs[n // 3, 5] ^= 0x10
    v_res = torch.zeros(n, dtype=torch.uint8, device=dev)

    def f_one():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k, v_res.data_ptr(), None, None, None))
    t1 = med(f_one); assert bool(v_res.all().item())
    out[str(n)] = {"verify_multiple_ms": t, "sets_per_s": n / t * 1e3, "same_sets_one_by_one_ms": t1}
    print(n, out[str(n)], flush=True)
print(json.dumps(out))
if len(sys.argv) > 1 and sys.argv[1] != "-":
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
