"""dev: the public-key sum routines on a tiny batch, with progress output (run under `timeout`)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "pymodel"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, q)
import helpers
import orc
from milagro_bls_amd import batch as mb
n, k = int(sys.argv[1]), int(sys.argv[2])
b = helpers.make_batch(n, k, fmt=1, seed=11, pool_n=16)
print("inputs built", flush=True)
apks, errs = mb.aggregate_public_keys_batch(b.pks, n, k, pk_format=1)
print("aggregate_public_keys_batch done", flush=True)
want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, 1, nthreads=8)
got, st = mb.fast_aggregate_verify_batch(b.sigs, b.msgs, b.pks, b.n, b.k, pk_format=1)
print("verify", got == want, [hex(x) for x in st[:8]], flush=True)
