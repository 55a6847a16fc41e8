"""Randomised GPU-vs-oracle stress over many seeds (dev tool; the same comparisons as tests/test_gpu_parity.py on more data):
fast_aggregate_verify with batch sizes either side of every engine crossover -- each batch once with the default engines, once forced onto
the lane-pair kernels, once onto the two-pair loop of the headline and once onto the cooperative engine --, signing and sk -> pk, verify_multiple with oracle-checked verdicts."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "pymodel"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, q)
import helpers
import orc
from milagro_bls_amd import _native as N, batch as mb
ctx = N.default_context()
bad = 0; total = 0; t0 = time.time()
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
big = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for seed in range(100, 100 + nseeds):
    shapes = [(1, 3, 1), (63, 5, 0), (65, 2, 1), (129, 7, 1), (1000, 4, 1), (333, 16, 0), (1537, 3, 1)]
    if big and seed % 4 == 0:
        shapes += [(8193, 2, 1), (2049, 9, 0)]
    for n, k, fmt in shapes:
        b = helpers.make_batch(n, k, fmt=fmt, seed=seed * 7 + n, pool_n=64)
        want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, fmt, nthreads=32)
        for engines in ("default", "pairs", "lanes2pair", "waves"):
            if engines == "pairs":                                   # one-lane kernels in the form the batch size takes: lane pairs / split
                ctx.set_coop_max_items(0); ctx.set_coop_hash_max_items(0)
            elif engines == "lanes2pair":                            # the two-pair loop of the headline
                ctx.set_coop_max_items(0); ctx.set_coop_hash_max_items(0); ctx.set_lane_shaping(0, (1 << 64) - 1)
            elif engines == "waves":
                ctx.set_coop_max_items(1 << 20); ctx.set_coop_hash_max_items(1 << 20 if n <= 8192 else 6144)
            try:
                got, st = mb.fast_aggregate_verify_batch(b.sigs, b.msgs, b.pks, b.n, b.k, pk_format=fmt)
            finally:
                ctx.reset_tuning()
            total += n
            if not (got == want == b.expect):
                bad += 1; print("MISMATCH fast_aggregate_verify seed", seed, n, k, fmt, engines)
    rnd = random.Random(seed)
    n = 257 + seed % 64
    sk = b"".join(rnd.randrange(1, helpers.R).to_bytes(32, "big") for _ in range(n)); msgs = rnd.randbytes(32 * n)
    sigs = mb.sign_batch(sk, msgs, n); pks = mb.sk_to_pk_batch(sk, n)
    if sigs != orc.batch_sign(sk, msgs, n, nthreads=32) or pks != orc.batch_sk_to_pk(sk, n, 0, nthreads=32):
        bad += 1; print("MISMATCH sign / sk_to_pk seed", seed)
    res, _ = mb.verify_batch(sigs, msgs, pks, n, pk_format=0)
    if not all(res):
        bad += 1; print("MISMATCH verify of device-made signatures seed", seed)
    total += 2 * n
print("items", total, "mismatching batches", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
