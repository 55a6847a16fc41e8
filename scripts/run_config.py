"""Run one BASELINE config a few times (for rocprofv3 kernel statistics): python3 scripts/run_config.py <2|4|lat|mid> [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from milagro_bls_amd import _native as N, batch
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
which = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if which == "2":
    n = 1 << 16
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, 1, N.PK_COMPRESSED, rank=11)
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    for _ in range(reps):
        ctx.check(lib.mbls_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_COMPRESSED, n, d_res.data_ptr(), None, None, None))
    torch.cuda.synchronize(); assert torch.equal(d_res.cpu(), expect)
elif which == "4":
    n, k = 1 << 14, 128
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=12, negatives=False)
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
    d_r = torch.full((8,), 7, dtype=torch.uint8, device=dev)
    for _ in range(reps):
        batch.verify_multiple_sets_device(d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, k, pk_format=N.PK_UNCOMPRESSED, d_result=d_r.data_ptr())
        torch.cuda.synchronize()
    assert int(d_r[0].item()) == 1
elif which == "mid":                       # a quarter of a round: front phases side by side, the two Miller pairs of an item on two lanes
    n, k = 1 << 14, 128
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=14)
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    for _ in range(reps):
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              d_res.data_ptr(), None, None, None))
    torch.cuda.synchronize(); assert torch.equal(d_res.cpu(), expect)
else:
    n, k = 64, 128
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=13)
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    for _ in range(reps):
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k, d_res.data_ptr(), None, None, None))
        torch.cuda.synchronize()
    assert torch.equal(d_res.cpu(), expect)
print("ok", which)
