"""How does the chip pack SEVERAL verification pipelines enqueued side by side (one context and one stream each)? Splits a batch of n items over
K contexts in several ways and times the whole (device-resident, 128 keys). Experiment behind DESIGN's batch-size curve: not a product path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from milagro_bls_amd import _native as N
lib = N.lib(); dev = torch.device("cuda:0")
nbase, k = 1 << 16, 128
KMAX = 4
ctxs = [N.Context(0) for _ in range(KMAX)]
streams = [torch.cuda.Stream() for _ in range(KMAX)]
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctxs[0], dev, nbase, k, N.PK_UNCOMPRESSED, rank=21)
sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [49152, 69632, 73728, 81920, 98304, 100000, 114688, 135168]
nmax = max(sizes); reps = -(-nmax // nbase)
D_sigs = d_sigs.repeat(reps, 1)[:nmax].contiguous(); D_msgs = d_msgs.repeat(reps, 1)[:nmax].contiguous(); D_pks = d_pks.repeat(reps, 1, 1)[:nmax].contiguous()
E = expect.repeat(reps)[:nmax]
for c in ctxs:
    c.reserve(nmax)


def med(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


def run(parts, mode):
    """parts: list of (lo, hi) item ranges, one context each"""
    n = parts[-1][1]
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    for c in ctxs:
        c.reset_tuning()
        if mode == "L1":
            c.set_coop_max_items(0); c.set_coop_hash_max_items(0); c.set_lane_shaping(0, 0)
        elif mode == "L1fork":
            c.set_coop_max_items(0); c.set_coop_hash_max_items(0); c.set_lane_shaping(0, 1 << 30)

    def f():
        for j, (lo, hi) in enumerate(parts):
            c = ctxs[j]
            c.check(lib.mbls_fast_aggregate_verify_batch_device(c.handle, D_sigs.data_ptr() + 96 * lo, D_msgs.data_ptr() + 32 * lo, 32, None, D_pks.data_ptr() + 96 * k * lo,
                                                                N.PK_UNCOMPRESSED, None, hi - lo, k, d_res.data_ptr() + lo, None, None, streams[j].cuda_stream))
        torch.cuda.synchronize()
    ms = med(f)
    assert torch.equal(d_res.cpu(), E[:n]), (parts, mode)
    return round(ms, 2)


def cuts(n, fr):
    at = [0]
    for x in fr:
        at.append(min(n, (int(at[-1] + x * n) + 63) // 64 * 64))
    at[-1] = n
    return [(at[i], at[i + 1]) for i in range(len(at) - 1)]


rows = {}
for n in sizes:
    r = {}
    r["one call"] = run([(0, n)], "default")
    for K in (2, 3, 4):
        for mode in ("default", "L1", "L1fork"):
            r["%d equal %s" % (K, mode)] = run(cuts(n, [1.0 / K] * K), mode)
    if n > 65536:
        for mode in ("default", "L1"):
            r["65536 + rest %s" % mode] = run([(0, 65536), (65536, n)], mode)
            r["rest + 65536 %s" % mode] = run([(0, n - 65536), (n - 65536, n)], mode)
    r["0.6/0.4 L1"] = run(cuts(n, [0.6, 0.4]), "L1")
    r["0.5/0.3/0.2 L1"] = run(cuts(n, [0.5, 0.3, 0.2]), "L1")
    best = min(r, key=r.get)
    rows[str(n)] = r
    print(n, "best:", best, r[best], "M/s %.2f" % (n / r[best] / 1e3), json.dumps(r), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/overlap_probe.json", "w"), indent=1)
