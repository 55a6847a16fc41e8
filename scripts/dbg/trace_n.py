"""dev probe: one batch size through the default routing a few times (to be run under rocprofv3 --kernel-trace):
trace_n.py fav N | vm N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from milagro_bls_amd import _native as N, batch
ctx = N.default_context(); dev = torch.device("cuda:0"); lib = N.lib()
mode, n = sys.argv[1], int(sys.argv[2]); k = 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=3, negatives=(mode == "fav"))
res = torch.zeros(n, dtype=torch.uint8, device=dev)
g = torch.Generator(device="cpu"); g.manual_seed(7)
rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
d_r = torch.full((8,), 7, dtype=torch.uint8, device=dev)
for _ in range(4):
    if mode == "fav":
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              res.data_ptr(), None, None, None))
    else:
        batch.verify_multiple_sets_device(d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, k, pk_format=N.PK_UNCOMPRESSED, d_result=d_r.data_ptr())
    torch.cuda.synchronize()
if mode == "fav":
    assert torch.equal(res.cpu(), expect)
else:
    assert int(d_r[0].item()) == 1
