"""Sustained throughput of a STREAM of batches (a server's view): B batches of n items submitted back to back through the device entries, which only enqueue --
on ONE context (calls serialise on its workspace) and alternating between TWO contexts on the same GPU (two workspaces, two sets of streams: while one
batch's kernels leave SIMDs idle -- any n that is not a multiple of a round -- the other batch's waves take them). Every result is checked.
-> gpurun_out/<tag>_pipelined.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from milagro_bls_amd import _native as N
lib = N.lib(); dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "dev"
nbase, k = 1 << 16, 128
KMAX = 8
ctxs = [N.Context(0) for _ in range(KMAX)]
streams = [torch.cuda.Stream() for _ in range(KMAX)]
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctxs[0], dev, nbase, k, N.PK_UNCOMPRESSED, rank=21)
sizes = [4096, 8192, 16384, 24576, 32768, 40960, 49152, 57344, 65536, 81920, 100000]
nmax = max(sizes); reps = -(-nmax // nbase)
D_sigs = d_sigs.repeat(reps, 1)[:nmax].contiguous(); D_msgs = d_msgs.repeat(reps, 1)[:nmax].contiguous(); D_pks = d_pks.repeat(reps, 1, 1)[:nmax].contiguous()
E = expect.repeat(reps)[:nmax]
for c in ctxs[:2]:
    c.reserve(nmax)
B = 8
rows = {}
for n in sizes:
    res = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(B)]

    def run(nctx):
        for b in range(B):
            c = ctxs[b % nctx]
            c.check(lib.mbls_fast_aggregate_verify_batch_device(c.handle, D_sigs.data_ptr(), D_msgs.data_ptr(), 32, None, D_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                                res[b].data_ptr(), None, None, streams[b % nctx].cuda_stream))
        torch.cuda.synchronize()
    r = {}
    for nctx in (1, 2):
        run(nctx)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); run(nctx); ts.append(time.perf_counter() - t)
        assert all(torch.equal(x.cpu(), E[:n]) for x in res), (n, nctx)
        r["%d context%s" % (nctx, "s" if nctx > 1 else "")] = {"ms_per_batch": round(float(np.median(ts)) / B * 1e3, 3), "items_per_s": round(B * n / float(np.median(ts)))}
    # THROUGHPUT MODE: as many contexts as batches fit a round of lanes, every batch on the one-lane kernels (the most efficient form: lane pairs cost 20-45 % more
    # lane-time for their shorter latency) -- what a server with a queue of small batches would configure
    K = min(KMAX, max(1, 65536 // n))
    if K > 1:
        for c in ctxs[:K]:
            c.reserve(n); c.set_coop_max_items(0); c.set_coop_hash_max_items(0); c.set_lane_shaping(0, 0)
        BB = 2 * K
        res2 = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(BB)]

        def run_t():
            for b in range(BB):
                c = ctxs[b % K]
                c.check(lib.mbls_fast_aggregate_verify_batch_device(c.handle, D_sigs.data_ptr(), D_msgs.data_ptr(), 32, None, D_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                                    res2[b].data_ptr(), None, None, streams[b % K].cuda_stream))
            torch.cuda.synchronize()
        run_t(); ts = []
        for _ in range(3):
            t = time.perf_counter(); run_t(); ts.append(time.perf_counter() - t)
        assert all(torch.equal(x.cpu(), E[:n]) for x in res2), (n, "throughput mode")
        r["%d contexts, one lane per item" % K] = {"ms_per_batch": round(float(np.median(ts)) / BB * 1e3, 3), "items_per_s": round(BB * n / float(np.median(ts)))}
        for c in ctxs[:K]:
            c.reset_tuning()
    rows[str(n)] = r
    print(n, r, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"_what": "%d batches of n items back to back (device entries, 128 uncompressed keys), one context vs two contexts on the same GPU; median of 3" % B, "rows": rows},
          open("gpurun_out/%s_pipelined.json" % tag, "w"), indent=1)
