"""signing and sk -> pk, 2^16 each: constant-time table access (default) against the variable-time forms (mbls_ctx_set_secret_ops)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
n = 1 << 16
g = torch.Generator(); g.manual_seed(3)
sk = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g); sk[:, 0] &= 0x3f
d_sk = sk.to(dev); d_msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g).to(dev)
d_sig = torch.zeros((n, 96), dtype=torch.uint8, device=dev); d_pk = torch.zeros((n, 48), dtype=torch.uint8, device=dev)
res = {}
for vt in (0, 1):
    ctx.set_secret_ops(vt)
    for name, f in (("sign", lambda: ctx.check(lib.mbls_sign_batch_device(ctx.handle, d_sk.data_ptr(), d_msg.data_ptr(), 32, n, d_sig.data_ptr(), None))),
                    ("sk_to_pk", lambda: ctx.check(lib.mbls_sk_to_pk_batch_device(ctx.handle, d_sk.data_ptr(), 0, n, d_pk.data_ptr(), None)))):
        ts = []
        for it in range(6):
            torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        res[(name, vt)] = float(np.median(ts[1:]))
    res[("sig", vt)] = d_sig.cpu().clone(); res[("pk", vt)] = d_pk.cpu().clone()
ctx.set_secret_ops(0)
assert torch.equal(res[("sig", 0)], res[("sig", 1)]) and torch.equal(res[("pk", 0)], res[("pk", 1)])
for name in ("sign", "sk_to_pk"):
    print(name, "constant-time %.2f ms, variable-time %.2f ms (2^16)" % (res[(name, 0)], res[(name, 1)]))
