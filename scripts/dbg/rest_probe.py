"""n = 65 536 + r: rounds + remainder in a row (tracks off) / two equal halves side by side (tracks on) / the round and the remainder SIDE BY SIDE (two contexts)"""
import json, os, sys, time
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
rs = [1024, 2048, 3072, 4096, 5120, 6144, 8192, 10240, 12288, 14336, 16384, 20480, 24576, 28672, 32768]
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
    for name, tracks in (("in a row", 0), ("halves", 1)):
        ctxs[0].reset_tuning(); ctxs[0].set_tracks(tracks)

        def f():
            call(ctxs[0], None, 0, n, d_res); torch.cuda.synchronize()
        out[name] = round(med(f), 2)
        assert torch.equal(d_res.cpu(), E[:n])
    for c in ctxs:
        c.reset_tuning()
    for name, order in (("round | rest", (0, 1)), ("rest | round", (1, 0))):
        parts = [(0, 65536), (65536, n)]

        def g():
            for j in order:
                call(ctxs[j], streams[j].cuda_stream, parts[j][0], parts[j][1], d_res)
            torch.cuda.synchronize()
        out[name] = round(med(g), 2)
        assert torch.equal(d_res.cpu(), E[:n])
    print(r, out, "best:", min(out, key=out.get), flush=True)
