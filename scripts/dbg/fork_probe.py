"""dev probe: the front phases side by side (fork) or in a row, by batch size"""
import sys
sys.path.insert(0, ".")
import torch, bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); dev = torch.device("cuda:0"); lib = N.lib()
k = 128
nmax = 65535
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, nmax, k, N.PK_UNCOMPRESSED, rank=3)
for n in (20480, 32768, 40960, 49152, 57344, 65535):
    res = torch.zeros(n, dtype=torch.uint8, device=dev)
    def f():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              res.data_ptr(), None, None, None))
    out = []
    for fork in (1 << 62, 0):
        ctx.set_lane_shaping(32768, fork)
        t = bench._med_ms(f)
        out.append(round(t, 2))
        assert torch.equal(res.cpu(), expect[:n])
    ctx.reset_tuning()
    print(n, "fork", out[0], "in a row", out[1])
