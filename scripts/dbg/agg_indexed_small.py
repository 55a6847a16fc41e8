"""dev: indexed public-key sum on a tiny batch (run under `timeout`)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "pymodel"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, q)
import helpers, orc
from milagro_bls_amd import batch as mb
from milagro_bls_amd._native import KeyTable
n, k = int(sys.argv[1]), int(sys.argv[2])
b = helpers.make_batch(n, k, fmt=1, seed=11, pool_n=16)
tab = KeyTable()
first, errs = tab.append(b.pks, n * k, pk_format=1, validate=False)
print("table", len(tab), first, set(errs), flush=True)
got, st = mb.fast_aggregate_verify_batch_indexed(tab, b.sigs, b.msgs, list(range(n * k)), n, k)
want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, 1, nthreads=8)
print("verify", got == want, got[:8], want[:8], [hex(x) for x in st[:8]], flush=True)
os._exit(0)
