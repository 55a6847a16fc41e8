"""dev probe: pairing check one / two items per wave by batch size (default routing otherwise)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); dev = torch.device("cuda:0"); lib = N.lib()
k = 128
sizes = [int(x) for x in sys.argv[1:]] or [2560, 3072, 3584, 4096, 4608, 5120]
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, max(sizes), k, N.PK_UNCOMPRESSED, rank=3)
for n in sizes:
    res = torch.zeros(n, dtype=torch.uint8, device=dev)
    def f():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                              res.data_ptr(), None, None, None))
    out = []
    for pack in ((1 << 62, 1 << 62, 768), (0, 1 << 62, 768)):
        ctx.set_coop_max_items(1 << 20); ctx.set_coop_hash_max_items(1 << 20)
        ctx.set_coop_packing(*pack)
        out.append(round(bench._med_ms(f), 2))
        assert torch.equal(res.cpu(), expect[:n])
    ctx.reset_tuning()
    print(n, "one per wave", out[0], "two per wave", out[1])
