"""Where does the two-track route (mbls_ctx_set_tracks) beat rounds + remainder, and do the halves want their front phases side by side?
n = 65 536 + r over a grid of r: (A) tracks off, (B) tracks on with the default fork limit, (C) tracks on, front phases in a row."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
nbase, k = 1 << 16, 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, nbase, k, N.PK_UNCOMPRESSED, rank=21)
sizes = [65536 + r for r in range(2048, 65536, 4096)] + [131072 + 10240, 131072 + 20480, 131072 + 36864, 131072 + 49152]
nmax = max(sizes); reps = -(-nmax // nbase)
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
    r = {}
    for name, tracks, forkmax in (("off", 0, None), ("on", 1, None), ("on_nofork", 1, 0), ("on_fork_all", 1, 1 << 30)):
        ctx.reset_tuning(); ctx.set_tracks(tracks)
        if forkmax is not None:
            ctx.set_lane_shaping(32768, forkmax)
        r[name] = round(med(f), 2)
        assert torch.equal(d_res.cpu(), E[:n]), (n, name)
    rows[str(n)] = r
    print(n, n % 65536, r, "best", min(r, key=r.get), flush=True)
ctx.reset_tuning()
json.dump(rows, open("gpurun_out/tracks_probe.json", "w"), indent=1)
