"""dev probe: latency of fast_aggregate_verify (128 keys) by batch size, default routing, results checked"""
import sys
sys.path.insert(0, ".")
import torch, bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); dev = torch.device("cuda:0"); lib = N.lib()
k = 128
sizes = [int(x) for x in sys.argv[1:]] or [1, 64, 1024, 2048, 3072, 4096, 5120]
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, max(sizes), k, N.PK_UNCOMPRESSED, rank=3)
for n in sizes:
    res = torch.zeros(n, dtype=torch.uint8, device=dev)
    def f():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              res.data_ptr(), None, None, None))
    t = bench._med_ms(f)
    assert torch.equal(res.cpu(), expect[:n])
    print(n, round(t, 3))
