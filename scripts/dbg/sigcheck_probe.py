import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import torch
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib()
import bench
dev = torch.device("cuda:0")
n = 65536
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, 1, N.PK_UNCOMPRESSED, rank=5, negatives=False)
sig = d_sigs.cpu().numpy(); errs = np.zeros(n, np.uint8); g2 = np.zeros(n, np.uint8)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
def f(): ctx.check(lib.mbls_sig_check_batch(ctx.handle, vp(sig), n, vp(errs), vp(g2)))
for _ in range(2): f()
ts=[]
for _ in range(5):
    t=time.perf_counter(); f(); ts.append(time.perf_counter()-t)
t=float(np.median(ts)); print("sig_check_batch 2^16 (host entry, PCIe incl): %.2f ms = %.1f M/s ok=%s" % (t*1e3, n/t/1e6, bool(g2.all() and not errs.any())))
