"""dev probe (library built with MBLS_EXTRA_HIPCC_FLAGS=-DMBLS_COOP_PROFILE): where the steps of the wave engine spend their time -- one fast_aggregate_verify(128)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from milagro_bls_amd import _native as N
ctx = N.default_context(); dev = torch.device("cuda:0"); lib = N.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
k = 128
d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, max(n, 64), k, N.PK_UNCOMPRESSED, rank=3)
res = torch.zeros(n, dtype=torch.uint8, device=dev)
raw = C.CDLL(N.LIB_PATH) if hasattr(N, "LIB_PATH") else lib
def run():
    ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None, n, k,
                                                          res.data_ptr(), None, None, None))
    torch.cuda.synchronize()
for _ in range(3): run()
buf = (C.c_ulonglong * 64)()
f = lib.mbls_coop_profile_read; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int]
f(buf, 1)
t = time.perf_counter(); run(); ms = (time.perf_counter() - t) * 1e3
f(buf, 1)
names = ["END", "MUL", "LIN", "INV", "ISZ", "FLG", "LOADW", "STOREW", "RES", "POW", "SGN"]
print("one call: %.2f ms (profiled build; clocks per step = the ns/step column / 10)" % ms)
tot = 0
for kd, nm in enumerate(names):
    c, a, b, d = buf[4 * kd:4 * kd + 4]
    if c:
        print("%-7s steps %5d  fetch %7.1f us (%5.0f ns/step)  compute %7.1f us (%5.0f)  store+barrier %7.1f us (%5.0f)" % (nm, c, a / 100, a * 10 / c, b / 100, b * 10 / c, d / 100, d * 10 / c))
        tot += a + b + d
print("sum %.1f us" % (tot / 100))
