"""Randomised sizes above a round through the device entries (two-track routes: the remainder beside the round, equal halves, rounds in front) against the expectation by
construction (bench.build_inputs: every 16th item corrupted over five rejection classes), byte keys and table indices, with and without a bitmap and a status array.
usage: stress_tracks.py [n_sizes]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
nbase, k = 1 << 16, 8
d_sigs, d_msgs, d_pks, expect, d_idx, table = bench.build_inputs(ctx, dev, nbase, k, N.PK_UNCOMPRESSED, rank=5, return_indices=True)
nmax = 3 * nbase + 5000; reps = -(-nmax // nbase)
D_sigs = d_sigs.repeat(reps, 1)[:nmax].contiguous(); D_msgs = d_msgs.repeat(reps, 1)[:nmax].contiguous(); D_pks = d_pks.repeat(reps, 1, 1)[:nmax].contiguous()
D_idx = d_idx.repeat(reps, 1)[:nmax].to(torch.int32).contiguous()
E = expect.repeat(reps)[:nmax]
ctx.reserve(nmax)
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0; items = 0; t0 = time.time()
for it in range(count):
    n = rnd.choice([rnd.randrange(65537, 70000), rnd.randrange(65537, 131072), rnd.randrange(131073, nmax)])
    ctx.reset_tuning()
    mode = rnd.randrange(4)
    if mode == 1:
        ctx.set_tracks(rnd.choice([1, 1000, 3584, 20000]), rnd.choice([0, 4096, 16384]))
    elif mode == 2:
        ctx.set_tracks(0)
    indexed = rnd.random() < 0.3
    want_bm = rnd.random() < 0.5
    d_res = torch.full((n,), 7, dtype=torch.uint8, device=dev)
    d_bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev) if want_bm else None
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev) if rnd.random() < 0.5 else None
    p = lambda x: x.data_ptr() if x is not None else None
    if indexed:
        ctx.check(lib.mbls_fast_aggregate_verify_batch_indexed_device(ctx.handle, table.handle, D_sigs.data_ptr(), D_msgs.data_ptr(), 32, None, D_idx.data_ptr(), None, n, k,
                                                                      d_res.data_ptr(), p(d_bm), p(d_st), None))
    else:
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, D_sigs.data_ptr(), D_msgs.data_ptr(), 32, None, D_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              d_res.data_ptr(), p(d_bm), p(d_st), None))
    torch.cuda.synchronize()
    ok = torch.equal(d_res.cpu(), E[:n])
    if want_bm:
        bits = ((d_bm.cpu().view(-1, 1) >> torch.arange(64)) & 1).reshape(-1)[:n].to(torch.uint8)
        ok = ok and torch.equal(bits, E[:n])
    if d_st is not None:
        st = d_st.cpu()
        ok = ok and bool(((st == 0) == (E[:n] == 1)).all())          # an accepted item has no status bit, a rejected one at least one
    if not ok:
        bad += 1; print("MISMATCH n =", n, "mode", mode, "indexed", indexed, flush=True)
    items += n
ctx.reset_tuning()
print("sizes", count, "items", items, "mismatching batches", bad, "%.1f s" % (time.time() - t0))
