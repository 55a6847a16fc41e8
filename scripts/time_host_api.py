"""Time the host-buffer entry point (PCIe-inclusive): mbls_fast_aggregate_verify_batch on 2^16 items x 128 keys."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, milagro_bls_amd as m
from milagro_bls_amd import _native as N
ctx = m.default_context(); dev = torch.device("cuda:0")
n, k = 1 << 16, 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, 0)
sigs = d_sigs.cpu().numpy().tobytes(); msgs = d_msgs.cpu().numpy().tobytes(); pks = d_pks.cpu().numpy().tobytes()
res = (C.c_uint8 * n)()
for it in range(4):
    t0 = time.time()
    rc = N.lib().mbls_fast_aggregate_verify_batch(ctx.handle, sigs, msgs, 32, None, pks, 1, None, n, k, res, None)
    t1 = time.time()
    print("host-buffer call rc %d  %.1f ms  -> %.0f verify/s, accepted %d of %d" % (rc, (t1 - t0) * 1e3, n / (t1 - t0), sum(res), n))
