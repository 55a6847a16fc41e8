"""fast_aggregate_verify throughput over the batch size (device-resident inputs, 128 uncompressed keys, 32-byte messages), default routing:
n = 2^10 ... 2^18 and sizes that are no powers of two, either side of a round of the one-lane kernels (65 536 items on MI355X).
-> profiles/<tag>_throughput_vs_n.json (ms = median of 5 calls after 2 warm-up calls; every result compared with the expectation by construction)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "dev"
nbase, k = 1 << 16, 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, nbase, k, N.PK_UNCOMPRESSED, rank=21)
sizes = [1024, 2048, 4096, 8192, 10240, 12288, 16384, 24576, 32768, 40960, 49152, 57344, 65536, 65537, 65600, 66560, 67584, 69632, 71680, 73728, 75776, 76000, 81920, 90112,
         98304, 100000, 114688, 131072, 135168, 150000, 163840, 196608, 200000, 262144]
nmax = max(sizes)
reps = -(-nmax // nbase)
# items are independent: larger batches are the 2^16 items repeated
D_sigs = d_sigs.repeat(reps, 1)[:nmax].contiguous(); D_msgs = d_msgs.repeat(reps, 1)[:nmax].contiguous(); D_pks = d_pks.repeat(reps, 1, 1)[:nmax].contiguous()
E = expect.repeat(reps)[:nmax]
ctx.reserve(nmax)


def med(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


rows = {}
for n in sizes:
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)

    def f():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, D_sigs.data_ptr(), D_msgs.data_ptr(), 32, None, D_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              d_res.data_ptr(), None, None, None))
        torch.cuda.synchronize()
    ms = med(f)
    assert torch.equal(d_res.cpu(), E[:n]), n
    rows[str(n)] = {"ms": round(ms, 3), "items_per_s": round(n / ms * 1e3)}
    print(n, rows[str(n)], flush=True)
out = {"_what": "mbls_fast_aggregate_verify_batch_device, 128 uncompressed keys, default routing; ms = median of 5", "rows": rows,
       "t_65536_plus_4096_over_t_65536": round(rows["69632"]["ms"] / rows["65536"]["ms"], 3)}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/%s_throughput_vs_n.json" % tag, "w"), indent=1)
