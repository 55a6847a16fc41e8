"""dev probe: one batch size through the default routing a few times (to be run under rocprofv3 --kernel-trace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); dev = torch.device("cuda:0"); lib = N.lib()
n = int(sys.argv[1]); k = 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=3)
res = torch.zeros(n, dtype=torch.uint8, device=dev)
for _ in range(4):
    ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                          res.data_ptr(), None, None, None))
    torch.cuda.synchronize()
assert torch.equal(res.cpu(), expect)
