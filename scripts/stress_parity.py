"""Randomised GPU-vs-oracle stress of fast_aggregate_verify over many seeds, batch sizes and key counts (dev tool; the same
comparison as tests/test_gpu_parity.py on more data)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "pymodel"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, q)
import helpers
import orc
from milagro_bls_amd import batch as mb
bad = 0; total = 0; t0 = time.time()
for seed in range(100, 100 + int(sys.argv[1]) if len(sys.argv) > 1 else 108):
    for n, k, fmt in ((1, 3, 1), (63, 5, 0), (65, 2, 1), (129, 7, 1), (1000, 4, 1), (333, 16, 0)):
        b = helpers.make_batch(n, k, fmt=fmt, seed=seed * 7 + n, pool_n=64)
        got, st = mb.fast_aggregate_verify_batch(b.sigs, b.msgs, b.pks, b.n, b.k, pk_format=fmt)
        want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, fmt, nthreads=32)
        total += n
        if not (got == want == b.expect):
            bad += 1; print("MISMATCH seed", seed, n, k, fmt)
print("items", total, "mismatching batches", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
