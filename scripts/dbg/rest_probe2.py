"""n = 65 536 + r for small r: in a row (remainder on the wave engine) against the round and the remainder SIDE BY SIDE with the remainder forced onto lane forms"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from milagro_bls_amd import _native as N
lib = N.lib(); dev = torch.device("cuda:0")
nbase, k = 1 << 16, 128
ctxs = [N.Context(0), N.Context(0)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctxs[0], dev, nbase, k, N.PK_UNCOMPRESSED, rank=21)
rs = [1024, 2048, 2560, 3072, 3584, 4096, 4608, 5120, 5632, 6144]
nmax = 65536 + max(rs); reps = -(-nmax // nbase)
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


def call(c, s, lo, hi, d_res):
    c.check(lib.mbls_fast_aggregate_verify_batch_device(c.handle, D_sigs.data_ptr() + 96 * lo, D_msgs.data_ptr() + 32 * lo, 32, None, D_pks.data_ptr() + 96 * k * lo,
                                                        N.PK_UNCOMPRESSED, None, hi - lo, k, d_res.data_ptr() + lo, None, None, s))


for r in rs:
    n = 65536 + r
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    out = {}
    ctxs[0].reset_tuning(); ctxs[0].set_tracks(0)

    def f():
        call(ctxs[0], None, 0, n, d_res); torch.cuda.synchronize()
    out["in a row"] = round(med(f), 2)
    assert torch.equal(d_res.cpu(), E[:n])
    ctxs[0].reset_tuning(); ctxs[1].reset_tuning(); ctxs[1].set_coop_max_items(0); ctxs[1].set_coop_hash_max_items(0)
    parts = [(0, 65536), (65536, n)]

    def g():
        for j in (0, 1):
            call(ctxs[j], streams[j].cuda_stream, parts[j][0], parts[j][1], d_res)
        torch.cuda.synchronize()
    out["round | rest on lanes"] = round(med(g), 2)
    assert torch.equal(d_res.cpu(), E[:n])
    print(r, out, flush=True)
