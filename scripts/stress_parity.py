"""Randomised GPU-vs-oracle stress over many seeds (dev tool; the same comparisons as tests/test_gpu_parity.py on more data):
fast_aggregate_verify with batch sizes either side of every engine crossover -- each batch once with the default engines, once forced onto
the lane-pair kernels, once onto the two-pair loop of the headline and once onto the cooperative engine --, signing and sk -> pk; verify_multiple (one call, cut into random shards, behind a two-context handle) and batched aggregate_verify with
random members spoiled, against the oracle."""
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
# ---- the n-pairing paths: verify_multiple (one call, cut into shards through the device entries, behind a two-context handle) and batched
# aggregate_verify, random members spoiled, against the oracle with the same scalars / item by item
import ctypes as C
import json
import torch
probe = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "vectors.json")))["model"]["g2_subgroup_probes"][0]["compressed"])
G1_INF_U = bytes([0x40]) + bytes(95)
dev = torch.device("cuda:0")
m2 = N.MultiContext([0, 0])
t = lambda b: torch.frombuffer(bytearray(b if b else b"\0"), dtype=torch.uint8).to(dev)
nvm = 0
for seed in range(100, 100 + max(1, nseeds // 4)):
    rnd = random.Random(seed * 13)
    for n in (1, 9, 70, 260):
        sks = [rnd.randrange(1, helpers.R) for _ in range(n)]
        pk96 = orc.batch_sk_to_pk(b"".join(x.to_bytes(32, "big") for x in sks), n, 1, nthreads=32)
        pks = [pk96[96 * i:96 * i + 96] for i in range(n)]
        msgs = [rnd.randbytes(32) for _ in range(n)]
        sg = orc.batch_sign(b"".join(x.to_bytes(32, "big") for x in sks), b"".join(msgs), n, nthreads=32)
        sigs = [sg[96 * i:96 * i + 96] for i in range(n)]
        rands = [rnd.randrange(1, 1 << 64) for _ in range(n)]
        kind = rnd.choice(["valid", "valid", "msg", "key", "swap", "outside", "sig inf", "both inf", "zero scalar"])
        i, j = rnd.randrange(n), rnd.randrange(n)
        if kind == "msg":
            msgs[i] = bytes([msgs[i][0] ^ 1]) + msgs[i][1:]
        elif kind == "key":
            pks[i] = pks[(i + 1) % n] if n > 1 else G1_INF_U
        elif kind == "swap" and n > 1 and i != j:
            sigs[i], sigs[j] = sigs[j], sigs[i]
        elif kind == "outside":
            sigs[i] = probe
        elif kind == "sig inf":
            sigs[i] = helpers.G2_INF
        elif kind == "both inf":
            sigs[i] = helpers.G2_INF; pks[i] = G1_INF_U
        elif kind == "zero scalar":
            rands[i] = 0
        want = False if kind == "zero scalar" else orc.verify_multiple([(orc.g2_from_compressed(s_)[1], a, m) for s_, a, m in zip(sigs, pks, msgs)], rands)
        rr = (C.c_uint64 * n)(*rands)
        got1 = bool(N.lib().mbls_verify_multiple_aggregate_signatures(ctx.handle, N.cbuf(b"".join(sigs)), N.cbuf(b"".join(pks)), N.cbuf(b"".join(msgs)), 32, None, rr, n))
        got2 = mb.multi_verify_multiple_aggregate_signatures(m2, b"".join(sigs), b"".join(pks), b"".join(msgs), rands, n)
        cuts = sorted([0, n] + [rnd.randrange(n + 1) for _ in range(rnd.randrange(4))])
        recs = torch.zeros((len(cuts) - 1) * N.VM_PARTIAL_BYTES, dtype=torch.uint8, device=dev)
        keep = []
        for g in range(len(cuts) - 1):
            lo, hi = cuts[g], cuts[g + 1]
            d = [t(b"".join(sigs[lo:hi])), t(b"".join(pks[lo:hi])), t(b"".join(msgs[lo:hi])),
                 torch.tensor([x - (1 << 64) if x >= (1 << 63) else x for x in rands[lo:hi]] or [0], dtype=torch.int64, device=dev)]
            keep += d
            mb.verify_multiple_partial_device(d[0].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), hi - lo, recs.data_ptr() + g * N.VM_PARTIAL_BYTES, d_apks=d[1].data_ptr())
        got3 = mb.verify_multiple_finish_device(recs.data_ptr(), len(cuts) - 1)
        nvm += n
        if not (got1 == got2 == got3 == want):
            bad += 1; print("MISMATCH verify_multiple seed", seed, n, kind, cuts, got1, got2, got3, want)
    # batched aggregate_verify: ragged items, spoiled members
    n = 40
    items = []
    for _ in range(n):
        k = rnd.choice([0, 1, 1, 2, 3, 5])
        sks = [rnd.randrange(1, helpers.R) for _ in range(k)]
        ms = [rnd.randbytes(rnd.choice([0, 7, 32, 100])) for _ in range(k)]
        pk = [orc.sk_to_pk(x) for x in sks]
        agg = None
        for x, m_ in zip(sks, ms):
            sgn = orc.sign(m_, x)
            agg = sgn if agg is None else orc.g2_add(agg, sgn)
        sig = orc.g2_compress(agg) if agg is not None else helpers.G2_INF
        kind = rnd.choice(["valid", "valid", "msg", "key", "outside"])
        if k and kind == "msg":
            ms[0] = ms[0] + b"x"
        elif k and kind == "key":
            pk[-1] = orc.sk_to_pk(sks[-1] % (helpers.R - 1) + 1)
        elif kind == "outside":
            sig = probe
        items.append((sig, ms, pk))
    off, moff, allm, allp = [0], [0], b"", b""
    for sig, ms, pk in items:
        for m_, q in zip(ms, pk):
            allm += m_; moff.append(len(allm)); allp += q
        off.append(len(moff) - 1)
    got, _ = mb.aggregate_verify_batch(b"".join(x[0] for x in items), allm, allp, n, pair_offsets=off, msg_len=0, msg_offsets=moff)
    want = []
    for sig, ms, pk in items:
        e, pt = orc.g2_from_compressed(sig)
        want.append(bool(not e and len(ms) and orc.aggregate_verify(pt, ms, pk)))
    total += sum(len(x[1]) + 1 for x in items)
    if got != want:
        bad += 1; print("MISMATCH aggregate_verify_batch seed", seed, [i_ for i_ in range(n) if got[i_] != want[i_]])
m2.close()
total += 3 * nvm
print("items", total, "mismatching batches", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
