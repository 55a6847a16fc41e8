"""The one-lane-per-item kernels the headline number is measured on (k_hash, k_miller + k_sig_verdict, k_final) against the oracle at batch
sizes where they are the default route, every routing boundary of the pipeline with the library's default settings, and the fuzz targets'
invariant. One seeded batch of 20 480 items x 2 keys is signed and verified by the oracle once; the tests below verify prefixes of it.

Routing (milagro_bls_amd/csrc/mbls_kernels.hip, verify_pipeline / launch_hash): n <= 768 hashg2 + pairing2, (768, 1024] hashg2x4 + pairing2,
(1024, 2048] hashg2x4 + pairing2x2, (2048, 3584] hashg2x4 + pairing2, (3584, 5120] k_hash2 (two lanes per message) + pairing2, (5120, 16384] k_hash2 (two lanes per message) + k_miller_split4 (four lanes per item: two per pair, products in pairs) + product + k_sig_verdict +
k_final2 (two lanes per item: the compressed squarings split, the other products in pairs), (16384, 20480] k_hash2 + k_miller_split (two lanes per item) + k_final2, (20480, 32768] k_hash + k_miller_split + k_final2, above 32768 k_hash + k_miller (two-pair loop) + k_sig_verdict + k_final; in both lane forms the signature's subgroup test is read off the
Miller loop's running point. Above a round (65 536 items) the remainder is routed as a batch of its own -- or, from 3 584 items up, the last round and the remainder run as two halves
side by side on two tracks (device entries)."""
import os
import random

import pytest

import helpers
import orc

pytestmark = pytest.mark.gpu

N_BIG = 20480
BOUNDARIES = [768, 769, 1024, 1025, 2048, 2049, 3584, 3585, 5120, 5121, 6144, 6145, 8192, 8193, 16384, 16385, 20480, 20481]
FLAG = {"sig_not_in_g2": 0x02, "apk_infinity": 0x08, "bad_sig_bytes": 0x01, "bad_pk_bytes": 0x04, "flip_msg": 0x40, "wrong_key": 0x40}


@pytest.fixture(scope="module")
def mb():
    from milagro_bls_amd import batch, _native
    _native.default_context()
    return batch


@pytest.fixture(scope="module")
def big():
    """20 480 items x 2 uncompressed keys, every fourth item one of the seven rejection classes; the oracle's verdict for every item"""
    import os
    nt = min(32, os.cpu_count() or 8)
    b = helpers.make_batch(N_BIG, 2, fmt=1, seed=4242, pool_n=64, nthreads=nt)
    b.want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, 1, nthreads=nt)
    assert b.want == b.expect
    assert set(b.kinds) == set(FLAG) | {"valid", "sig_infinity"}
    return b


def prefix(b, n):
    """the first n items (tiled beyond the batch: items are independent, reference src/aggregates.rs:177-215 keeps no state between calls)"""
    reps = -(-n // b.n)
    if reps == 1:
        return b.sigs[:96 * n], b.msgs[:32 * n], b.pks[:96 * b.k * n]
    return (b.sigs * reps)[:96 * n], (b.msgs * reps)[:32 * n], (b.pks * reps)[:96 * b.k * n]


def check(b, got, st, n):
    reps = -(-n // b.n)
    assert got == (b.want * reps)[:n]
    for i in range(n):
        kind = b.kinds[i % b.n]
        f = FLAG.get(kind)
        if f:
            assert st[i] & f, (i, kind, st[i])
        elif kind == "valid":
            assert st[i] == 0, (i, st[i])


@pytest.mark.parametrize("n", BOUNDARIES + [32768, 32769])
def test_routing_boundaries_with_default_settings_vs_oracle(mb, big, n):
    """either side of every crossover at which engine selection flips, default settings, every item compared with the oracle
    (reference src/aggregates.rs:177-215: the same bool whatever the batch size)"""
    from milagro_bls_amd import _native as N
    ctx = N.default_context()
    ctx.reset_tuning()
    s, m, p = prefix(big, n)
    got, st = mb.fast_aggregate_verify_batch(s, m, p, n, big.k, pk_format=1)
    check(big, got, st, n)


@pytest.mark.parametrize("k", [24, 32, 40, 128])
def test_wave_engine_batches_sum_their_keys_on_eight_lanes_vs_oracle(mb, k):
    """small batches (the wave engine's) cut an item's key sum into eight partial sums on lanes of their own when k is a multiple of 8 and at least 32
    (k_apk_combine adds them); 24 keys stay on one lane. Every item against the oracle -- the seven rejection classes (an aggregate key at infinity must still be
    seen on the TOTAL, a key that does not decode in any of the eight parts), a key repeated within an item (a doubling inside a partial sum), a key and its
    negative in different parts -- through byte keys and through a key table."""
    from milagro_bls_amd import _native as N
    N.default_context().reset_tuning()
    n = 70
    b = helpers.make_batch(n, k, fmt=1, seed=900 + k, pool_n=max(64, k + 8), nthreads=8)
    keys = [b.pks[96 * k * i:96 * k * (i + 1)] for i in range(n)]
    K = lambda i, j: keys[i][96 * j:96 * j + 96]
    def put(i, j, key):
        keys[i] = keys[i][:96 * j] + key + keys[i][96 * j + 96:]
    put(0, 1, K(0, 0)); put(0, k - 1, K(0, 0))                                  # item 0: one key three times (parts 0 and 7)
    put(4, k - 2, orc.g1_mul(K(4, 2), helpers.R - 1))                            # item 4: a key and its negative in different parts
    per = max(k // 8, 1)
    for j in range(k - per, k - 1, 2):                                           # item 8: the whole last part cancels (a partial sum at infinity; 3 keys: all but one)
        put(8, j + 1, orc.g1_mul(K(8, j), helpers.R - 1))
    pks = b"".join(keys)
    want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, pks, n, k, 1, nthreads=8)
    assert want[0] is False and want[4] is False and want[8] is False and want.count(True) > n // 2
    got, st = mb.fast_aggregate_verify_batch(b.sigs, b.msgs, pks, n, k, pk_format=1)
    assert got == want
    for i, kind in enumerate(b.kinds):
        if i not in (0, 4, 8) and FLAG.get(kind):
            assert st[i] & FLAG[kind], (i, kind, hex(st[i]))
    # the same items by table indices (keys that do not decode cannot enter a table: those items keep their other keys and are skipped in the comparison)
    tab = N.KeyTable()
    uniq = {}
    idx = []
    skip = set()
    for i in range(n):
        for j in range(k):
            key = K(i, j)
            if key not in uniq:
                first, errs = tab.append(key, 1, pk_format=1, validate=False)
                uniq[key] = None if errs[0] else first
            if uniq[key] is None:
                skip.add(i); idx.append(0)
            else:
                idx.append(uniq[key])
    got_i, _ = mb.fast_aggregate_verify_batch_indexed(tab, b.sigs, b.msgs, idx, n, k)
    assert [g for i, g in enumerate(got_i) if i not in skip] == [w for i, w in enumerate(want) if i not in skip] and len(skip) < n // 4
    tab.close()


def test_lane_kernels_20k_items_vs_oracle(mb, big):
    """the headline kernels (one lane per item; signature subgroup verdict out of the Miller loop) on 20 480 items, every one against the
    oracle: the two-pair loop k_miller forced (lane shaping off, front phases one after the other -- the shape of a full round), then the
    default route of a batch this size (two lanes per item in the Miller phase, front phases side by side); status words identical"""
    from milagro_bls_amd import _native as N
    ctx = N.default_context()
    try:
        ctx.set_coop_max_items(0); ctx.set_coop_hash_max_items(0); ctx.set_lane_shaping(0, 0)
        got0, st0 = mb.fast_aggregate_verify_batch(big.sigs, big.msgs, big.pks, big.n, big.k, pk_format=1)
    finally:
        ctx.reset_tuning()
    check(big, got0, st0, big.n)
    got1, st1 = mb.fast_aggregate_verify_batch(big.sigs, big.msgs, big.pks, big.n, big.k, pk_format=1)
    assert got1 == got0 and st1 == st0


def test_lane_and_wave_engines_agree_item_by_item_on_status_words(mb, big):
    """the first 4 096 items through both engines: identical accept bits AND identical status words (same rejection reason)"""
    from milagro_bls_amd import _native as N
    ctx = N.default_context()
    n = 4096
    s, m, p = prefix(big, n)
    out = {}
    try:
        for name, lim in (("lanes", 0), ("waves", 1 << 20)):
            ctx.set_coop_max_items(lim); ctx.set_coop_hash_max_items(lim)
            out[name] = mb.fast_aggregate_verify_batch(s, m, p, n, big.k, pk_format=1)
    finally:
        ctx.reset_tuning()
    assert out["lanes"] == out["waves"]
    check(big, out["lanes"][0], out["lanes"][1], n)


def test_tail_routing_above_one_round_of_lanes(mb, big):
    """n = 65 536 + r: the library cuts the batch into a full round of one-lane kernels and a tail that takes the route of an r-item batch.
    The oracle-checked 20 480 items are tiled (items are independent, src/aggregates.rs:177-215 keeps no state) to 65 537, 66 000 and 70 000
    items; every item must come back with the oracle's verdict of its source item."""
    for n in (65537, 66000, 70000, 80000):           # remainders on one wave per item (1, 464, 4 464) and on two lanes per item (14 464)
        s, m, p = prefix(big, n)
        got, st = mb.fast_aggregate_verify_batch(s, m, p, n, big.k, pk_format=1)
        check(big, got, st, n)


def test_round_cut_with_small_rounds_every_layout(mb, big):
    """the cut itself on small numbers (mbls_ctx_set_round_items(128)): n = 300 = two rounds + 44 items through the host entry (uniform
    and ragged layouts, byte keys and table indices) and the device entry (bitmap words, caller's status array) -- against the oracle."""
    import torch
    from milagro_bls_amd import _native as N
    ctx = N.default_context()
    n = 300
    s, m, p = prefix(big, n)
    try:
        ctx.set_round_items(128)
        for lim in (1 << 20, 0):                     # the remainder on waves / on lanes
            ctx.set_coop_max_items(lim)
            got, st = mb.fast_aggregate_verify_batch(s, m, p, n, big.k, pk_format=1)
            check(big, got, st, n)
        ctx.reset_tuning()
        # ragged: item i keeps its two keys, but through an offset table that starts at 5; messages through an offset table
        koff = [5 + 2 * i for i in range(n + 1)]
        moff = [32 * i for i in range(n + 1)]
        got, st = mb.fast_aggregate_verify_batch(s, m, bytes(96 * 5) + p, n, pk_format=1, pk_offsets=koff, msg_offsets=moff)
        check(big, got, st, n)
        # table indices: the 2 n keys as a table, item i = entries (2 i, 2 i + 1)
        tab = N.KeyTable(ctx, capacity_hint=2 * n)
        first, errs = tab.append(p, 2 * n, pk_format=1, validate=False)
        bad_keys = {2 * i for i in range(n) if big.kinds[i] == "bad_pk_bytes"}
        assert first == 0 and {j for j, e in enumerate(errs) if e} == bad_keys
        got, st = mb.fast_aggregate_verify_batch_indexed(tab, s, m, list(range(2 * n)), n, 2)
        check(big, got, st, n)
        # device entry with a bitmap
        dev = torch.device("cuda:0")
        t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        d_s, d_m, d_p = t(s), t(m), t(p)
        d_res = torch.full((n,), 9, dtype=torch.uint8, device=dev); d_bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
        d_st = torch.full((n,), 0x7fffffff, dtype=torch.int32, device=dev)
        ctx.check(N.lib().mbls_fast_aggregate_verify_batch_device(ctx.handle, d_s.data_ptr(), d_m.data_ptr(), 32, None, d_p.data_ptr(), 1, None, n, 2,
                                                                  d_res.data_ptr(), d_bm.data_ptr(), d_st.data_ptr(), None))
        torch.cuda.synchronize()
        got = [bool(x) for x in d_res.cpu().tolist()]
        check(big, got, [x & 0xffffffff for x in d_st.cpu().tolist()], n)
        bits = [(int(w) >> b) & 1 for w in d_bm.cpu().tolist() for b in range(64)][:n]
        assert bits == [int(x) for x in got]
        tab.close()
    finally:
        ctx.reset_tuning()


def _device_call(N, ctx, dev, s, m, p, n, k, *, koff=None, moff=None, table=None, idx=None, fmt=1):
    """mbls_fast_aggregate_verify_batch[_indexed]_device with a bitmap and the caller's status array -> (results, status, bitmap bits)"""
    import torch
    t = lambda b, dt=torch.uint8: torch.frombuffer(bytearray(b), dtype=dt).to(dev)
    d_s, d_m = t(s), t(m)
    d_res = torch.full((n,), 9, dtype=torch.uint8, device=dev); d_bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    d_st = torch.full((n,), 0x7fffffff, dtype=torch.int32, device=dev)
    d_ko = torch.tensor(koff, dtype=torch.int32, device=dev) if koff is not None else None
    d_mo = torch.tensor(moff, dtype=torch.int64, device=dev) if moff is not None else None
    ptr = lambda x: x.data_ptr() if x is not None else None
    if table is not None:
        d_i = torch.tensor(idx, dtype=torch.int32, device=dev)
        ctx.check(N.lib().mbls_fast_aggregate_verify_batch_indexed_device(ctx.handle, table.handle, d_s.data_ptr(), d_m.data_ptr(), 32, ptr(d_mo), d_i.data_ptr(), ptr(d_ko),
                                                                          n, 0 if koff is not None else k, d_res.data_ptr(), d_bm.data_ptr(), d_st.data_ptr(), None))
    else:
        d_p = t(p)
        ctx.check(N.lib().mbls_fast_aggregate_verify_batch_device(ctx.handle, d_s.data_ptr(), d_m.data_ptr(), 32, ptr(d_mo), d_p.data_ptr(), fmt, ptr(d_ko),
                                                                  n, 0 if koff is not None else k, d_res.data_ptr(), d_bm.data_ptr(), d_st.data_ptr(), None))
    torch.cuda.synchronize()
    got = [bool(x) for x in d_res.cpu().tolist()]
    bits = [(int(w) >> b) & 1 for w in d_bm.cpu().tolist() for b in range(64)][:n]
    return got, [x & 0xffffffff for x in d_st.cpu().tolist()], bits


def test_context_limits_are_what_the_plan_is_made_from(mb):
    """mbls_ctx_get_limits: the defaults follow the device's round (CUs x 4 x 64), the setters move them, and mbls_plan_batch on them is the plan the entries carry out
    (tests/test_plan_cpu.py checks the plan itself at every boundary without a GPU)"""
    import torch
    from milagro_bls_amd import _native as N
    ctx = N.default_context()
    ctx.reset_tuning()
    L = ctx.limits()
    R = torch.cuda.get_device_properties(0).multi_processor_count * 4 * 64
    D = N.default_limits(R)
    assert [getattr(L, f) for f, _ in N.Limits._fields_] == [getattr(D, f) for f, _ in N.Limits._fields_]
    try:
        ctx.set_tracks(1000, 2000); ctx.set_coop_max_items(7); ctx.set_lane_shaping(100, 200)
        L = ctx.limits()
        assert (L.tracks_min_rest, L.tracks_side_max, L.coop_max_items, L.split_max_items, L.fork_max_items) == (1000, 2000, 7, 100, 200)
        mode, ps = N.plan_batch(R + 1500, L)
        assert mode == N.BATCH_ROUND_BESIDE_REST and ps[1]["items"] == 1500 and ps[1]["pairing"] == N.PAIRING_LANE      # split_max 100: the remainder on one lane per item
    finally:
        ctx.reset_tuning()


def test_two_tracks_above_a_round_vs_oracle(mb, big):
    """n = q rounds + r with r >= mbls_ctx_set_tracks' limit (default 3 584): the last round and the remainder run on two tracks SIDE BY SIDE, each on its own part
    of the workspace and its own streams (verify_pipeline) -- the remainder beside the round up to a quarter of a round (73 728, 81 920 items; 150 000 with a whole
    round in front), two equal halves above (100 000) or when the side mode is off. The oracle-checked items, tiled (items are independent,
    src/aggregates.rs:177-215); 69 632 (remainder 4 096: beside the round on lane pairs, not on waves), 67 000 with the limit lowered (remainder 1 464) -- results, status words and bitmap words of every item, through the device entry."""
    import torch
    from milagro_bls_amd import _native as N
    ctx = N.default_context(); dev = torch.device("cuda:0")
    try:
        for n, lim in ((69632, None), (73728, None), (81920, None), (100000, None), (150000, None), (67000, 1000), (73728, (6144, 0))):
            if lim:                                  # (.., 0): no side mode -- 73 728 items as two equal halves; default: the remainder beside the round
                ctx.set_tracks(*lim) if isinstance(lim, tuple) else ctx.set_tracks(lim)
            s, m, p = prefix(big, n)
            got, st, bits = _device_call(N, ctx, dev, s, m, p, n, big.k)
            check(big, got, st, n)
            assert bits == [int(x) for x in got]
            ctx.reset_tuning(); ctx.set_tracks(0)   # the same batch as rounds + remainder: identical status words
            got0, st0, _ = _device_call(N, ctx, dev, s, m, p, n, big.k)
            assert (got0, st0) == (got, st)
            ctx.reset_tuning()
    finally:
        ctx.reset_tuning()


def test_two_tracks_with_small_rounds_every_layout(mb, big):
    """the two-track cut on small numbers (rounds of 128 items, limit 1, one lane per item): n = 300 = one round in front + halves of 128 and 44 items; n = 200 =
    halves of 128 and 72; n = 276 and 160 with the remainder BESIDE the round (side mode: 20 / 32 items on lane pairs next to a round of 128) -- uniform keys, ragged keys + ragged messages through offset tables that do not start at 0, table indices, and 48-byte keys (the staged
    decompression buffer is cut like the workspace) -- against the oracle."""
    import torch
    from milagro_bls_amd import _native as N
    ctx = N.default_context(); dev = torch.device("cuda:0")
    try:
        for n, side_max in ((300, 0), (200, 0), (276, 64), (160, 64)):          # halves (128 + 44 behind a round; 128 + 72) / the remainder beside the round (20 items behind a round, 32 items)
            ctx.reset_tuning(); ctx.set_round_items(128); ctx.set_coop_max_items(0); ctx.set_coop_hash_max_items(0); ctx.set_tracks(1, side_max)
            s, m, p = prefix(big, n)
            got, st, bits = _device_call(N, ctx, dev, s, m, p, n, big.k)
            check(big, got, st, n); assert bits == [int(x) for x in got]
            koff = [5 + 2 * i for i in range(n + 1)]
            moff = [7 + 32 * i for i in range(n + 1)]
            got, st, bits = _device_call(N, ctx, dev, s, bytes(7) + m, bytes(96 * 5) + p, n, big.k, koff=koff, moff=moff)
            check(big, got, st, n); assert bits == [int(x) for x in got]
            tab = N.KeyTable(ctx, capacity_hint=2 * n)
            first, errs = tab.append(p, 2 * n, pk_format=1, validate=False)
            assert first == 0
            got, st, bits = _device_call(N, ctx, dev, s, m, None, n, 2, table=tab, idx=list(range(2 * n)))
            check(big, got, st, n); assert bits == [int(x) for x in got]
            tab.close()
        # 48-byte keys: a batch of its own (the big fixture holds 96-byte keys)
        b = helpers.make_batch(276, 3, fmt=0, seed=77, nthreads=8)
        want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, 0, nthreads=8)
        assert want == b.expect
        for side_max in (0, 64):
            ctx.set_tracks(1, side_max)
            got, st, bits = _device_call(N, ctx, dev, b.sigs, b.msgs, b.pks, b.n, b.k, fmt=0)
            assert got == want and bits == [int(x) for x in got]
    finally:
        ctx.reset_tuning()


def test_fuzz_invariant_accepted_encodings_reencode_to_themselves(mb):
    """The reference's fuzz targets (fuzz/fuzz_targets/fuzz_serde_public_key.rs:5-10, fuzz_serde_signature.rs:5-10): from_bytes(data) is Ok
    => as_bytes() == data. 4 096 G1 and 4 096 G2 blobs -- canonical field elements with random flag bits, fully random bytes, valid
    encodings and single-bit mutations of them -- through decode and re-encode on the device; accept/reject classes against the oracle."""
    rnd = random.Random(0xF022)
    P = helpers.P
    n = 4096
    valid_pk = [orc.g1_compress(orc.sk_to_pk(rnd.randrange(1, helpers.R))) for _ in range(32)]
    blobs = []
    for i in range(n):
        t = i % 4
        if t == 0:                                   # x < p with random flags: about one in five decodes
            b = bytearray(rnd.randrange(P).to_bytes(48, "big")); b[0] = (b[0] & 0x1F) | (rnd.randrange(8) << 5)
        elif t == 1:
            b = bytearray(rnd.randbytes(48))
        elif t == 2:
            b = bytearray(valid_pk[rnd.randrange(32)])
        else:
            b = bytearray(valid_pk[rnd.randrange(32)]); b[rnd.randrange(48)] ^= 1 << rnd.randrange(8)
        blobs.append(bytes(b))
    blobs[0] = bytes([0xC0]) + bytes(47)             # infinity
    data = b"".join(blobs)
    for validate in (False, True):
        out96, errs = mb.pk_decode_batch(data, n, in_format=0, validate=validate)
        ok = [i for i in range(n) if errs[i] == 0]
        assert len(ok) > n // (4 if validate else 3)
        back, e2 = mb.pk_compress_batch(b"".join(out96[96 * i:96 * i + 96] for i in ok), len(ok))
        assert not any(e2)
        assert all(back[48 * j:48 * j + 48] == blobs[i] for j, i in enumerate(ok)), validate
        for i in range(0, n, 16):                    # the accept / reject class against the oracle on a subsample
            e, pt = orc.g1_from_compressed(blobs[i])
            good = e == 0 and (not validate or orc.g1_key_validate(pt))
            assert (errs[i] == 0) == good, (i, validate)
    # signatures: decode -> (one-member) aggregate -> compress
    valid_sig = [orc.g2_compress(orc.sign(bytes([j]) * 32, rnd.randrange(1, helpers.R))) for j in range(16)]
    blobs = []
    for i in range(n):
        t = i % 4
        if t == 0:
            b = bytearray(rnd.randrange(P).to_bytes(48, "big") + rnd.randrange(P).to_bytes(48, "big")); b[0] = (b[0] & 0x1F) | (rnd.randrange(8) << 5)
        elif t == 1:
            b = bytearray(rnd.randbytes(96))
        elif t == 2:
            b = bytearray(valid_sig[rnd.randrange(16)])
        else:
            b = bytearray(valid_sig[rnd.randrange(16)]); b[rnd.randrange(96)] ^= 1 << rnd.randrange(8)
        blobs.append(bytes(b))
    blobs[0] = helpers.G2_INF
    data = b"".join(blobs)
    errs, in_g2 = mb.sig_check_batch(data, n)
    sums, e2 = mb.aggregate_signatures_batch(data, n, 1)
    ok = [i for i in range(n) if errs[i] == 0]
    assert len(ok) > n // 3
    assert [x == 0 for x in e2] == [x == 0 for x in errs]
    assert all(sums[96 * i:96 * i + 96] == blobs[i] for i in ok)
    for i in range(0, n, 16):
        e, pt = orc.g2_from_compressed(blobs[i])
        assert (errs[i] == 0) == (e == 0), i
        if e == 0:
            assert in_g2[i] == orc.g2_subgroup_check(pt), i


def test_aggregate_verify_batch_vs_oracle(mb):
    """mbls_aggregate_verify_batch: n x AggregateSignature::aggregate_verify (reference src/aggregates.rs:130-170; its tests :808-929) in one
    call -- ragged pair counts (0, 1, 2 ... 9 pairs, one item of 70), ragged message lengths, and per item one of: valid, repeated message
    (:833-860), wrong signature (:864-891), a message swapped, a key at infinity whose signer is left out of / kept in the aggregate, an
    undecodable key, a signature outside G2, the infinite signature, no pairs at all -- every item against the oracle."""
    rnd = random.Random(77)
    pool = 40
    sks = [rnd.randrange(1, helpers.R) for _ in range(pool)]
    pks = [orc.sk_to_pk(s) for s in sks]
    inf_pk = bytes([0x40]) + bytes(95)
    probe = bytes.fromhex(helpers.load_vectors()["model"]["g2_subgroup_probes"][1]["compressed"])
    kinds = ["valid", "repeat", "wrong_sig", "swap_msg", "inf_key_out", "inf_key_in", "bad_key", "not_in_g2", "inf_sig", "empty", "valid", "valid"]
    n = 96
    sigs, all_msgs, all_keys, off, want, tags = [], [], [], [0], [], []
    for i in range(n):
        kind = kinds[i % len(kinds)]
        cnt = 70 if i == 50 else 0 if kind == "empty" else max(2 if kind in ("swap_msg", "inf_key_out", "inf_key_in") else 1, rnd.randrange(1, 10))
        who = [rnd.randrange(pool) for _ in range(cnt)]
        msgs = [rnd.randbytes(rnd.randrange(0, 70)) for _ in range(cnt)]
        if kind == "repeat" and cnt >= 2:
            msgs[1] = msgs[0]
        keys = [pks[j] for j in who]
        signers = list(range(cnt))
        if kind == "inf_key_out":
            keys[-1] = inf_pk; signers = signers[:-1]
        elif kind == "inf_key_in":
            keys[-1] = inf_pk
        agg = None
        for t in signers:
            p = orc.sign(msgs[t], sks[who[t]])
            agg = p if agg is None else orc.g2_add(agg, p)
        sigc = orc.g2_compress(agg) if agg is not None else helpers.G2_INF
        if kind == "wrong_sig":
            sigc = orc.g2_compress(orc.sign(b"something else", sks[who[0]]))
        elif kind == "swap_msg":
            msgs[0], msgs[1] = msgs[1], msgs[0]
            if msgs[0] == msgs[1]:
                msgs[0] = msgs[0] + b"x"
        elif kind == "bad_key":
            keys[0] = b"\xff" * 96
        elif kind == "not_in_g2":
            sigc = probe
        elif kind == "inf_sig":
            sigc = helpers.G2_INF
        e, pt = orc.g2_from_compressed(sigc)
        w = False if (e or cnt == 0 or kind == "bad_key") else orc.aggregate_verify(pt, msgs, keys)
        sigs.append(sigc); all_msgs += msgs; all_keys += keys; off.append(off[-1] + cnt); want.append(w); tags.append(kind)
    assert any(want) and want.count(False) > 30
    for kind, w in zip(tags, want):
        assert w == (kind in ("valid", "repeat", "inf_key_out")), kind
    moff = [0]
    for m in all_msgs:
        moff.append(moff[-1] + len(m))
    got, st = mb.aggregate_verify_batch(b"".join(sigs), b"".join(all_msgs), b"".join(all_keys), n, pair_offsets=off, msg_offsets=moff)
    assert got == want
    for i, kind in enumerate(tags):
        if kind == "empty":
            assert st[i] & 0x10
        elif kind == "bad_key":
            assert st[i] & 0x04
        elif kind == "not_in_g2":
            assert st[i] & 0x02
        elif want[i]:
            assert st[i] == 0
    # uniform layout: k pairs and 32-byte messages per item, 300 items (the pairs fill more than one wave per level)
    n2, k2 = 300, 4
    who = [[rnd.randrange(pool) for _ in range(k2)] for _ in range(n2)]
    msgs = [[rnd.randbytes(32) for _ in range(k2)] for _ in range(n2)]
    sigs2, want2 = [], []
    for i in range(n2):
        agg = None
        for t in range(k2):
            p = orc.sign(msgs[i][t], sks[who[i][t]])
            agg = p if agg is None else orc.g2_add(agg, p)
        if i % 5 == 3:
            msgs[i][2] = bytes([msgs[i][2][0] ^ 4]) + msgs[i][2][1:]
        sigs2.append(orc.g2_compress(agg)); want2.append(i % 5 != 3)
    got2, _ = mb.aggregate_verify_batch(b"".join(sigs2), b"".join(b"".join(m) for m in msgs), b"".join(pks[j] for w in who for j in w), n2, k=k2)
    assert got2 == want2
    assert mb.aggregate_verify_batch(b"", b"", b"", 0, k=3) == ([], [])


def test_aggregate_verify_batch_device_with_malformed_pair_offsets(mb):
    """The device entry takes the pair offset table as it is (the host entry refuses a bad one): a table that does not start at 0, ends below the total, runs
    backwards or makes two items' ranges overlap must never become an out-of-range index -- pairs no item owns are skipped, an item whose range is invalid or
    shares a pair with another item is rejected (status bit 0x04), and every OTHER item still gets the oracle's verdict. The staging buffer behind the
    pair -> item map is reused between calls: a first call with a long table leaves stale indices behind for the second (src/aggregates.rs:130-170)."""
    import torch
    from milagro_bls_amd import _native as N
    ctx = N.default_context(); dev = torch.device("cuda:0")
    rnd = random.Random(5)
    sks = [rnd.randrange(1, helpers.R) for _ in range(12)]
    pks = [orc.sk_to_pk(s) for s in sks]
    n, k = 8, 3
    who = [[rnd.randrange(12) for _ in range(k)] for _ in range(n)]
    msgs = [[rnd.randbytes(32) for _ in range(k)] for _ in range(n)]
    sigs = []
    for i in range(n):
        agg = None
        for t in range(k):
            p = orc.sign(msgs[i][t], sks[who[i][t]])
            agg = p if agg is None else orc.g2_add(agg, p)
        sigs.append(orc.g2_compress(agg))
    total = n * k
    t8 = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_s, d_m, d_p = t8(b"".join(sigs)), t8(b"".join(b"".join(m) for m in msgs)), t8(b"".join(pks[j] for w in who for j in w))

    def run(off, nn=n, tot=total):
        d_off = torch.tensor(off, dtype=torch.int32, device=dev)
        d_res = torch.full((nn,), 9, dtype=torch.uint8, device=dev); d_st = torch.zeros(nn, dtype=torch.int32, device=dev)
        ctx.check(N.lib().mbls_aggregate_verify_batch_device(ctx.handle, d_s.data_ptr(), d_m.data_ptr(), 32, None, d_p.data_ptr(), d_off.data_ptr(), 0, tot, nn,
                                                             d_res.data_ptr(), d_st.data_ptr(), None))
        torch.cuda.synchronize()
        return [bool(x) for x in d_res.cpu().tolist()], [x & 0xffffffff for x in d_st.cpu().tolist()]
    good = [k * i for i in range(n + 1)]
    got, st = run(good)
    assert got == [True] * n and st == [0] * n
    # a table that covers only pairs [6, 18): items 0, 1 empty (false: no pairs), items 2..5 as before, items 6, 7 empty; pairs [0, 6) and [18, 24) have no owner
    off = [6, 6, 6, 9, 12, 15, 18, 18, 18]
    got, st = run(off)
    assert got == [False, False, True, True, True, True, False, False]
    assert all(st[i] & 0x10 for i in (0, 1, 6, 7)) and st[2:6] == [0] * 4
    # item 3 runs backwards (rejected, never read); item 4 then starts below item 2's end: items 2 and 4 overlap in pairs [6, 9) -- at least one of them is rejected and
    # whichever is not rejected holds the right verdict for ITS range (item 4 over pairs [6, 15) with item 4's signature: false either way); the rest are untouched
    off = [0, 3, 6, 9, 6, 15, 18, 21, 24]
    got, st = run(off)
    assert st[3] & 0x04 and not got[3]
    assert (st[2] & 0x04) or (st[4] & 0x04)
    assert not got[4] and (got[2] or st[2] & 0x04)
    assert got[0] and got[1] and got[5] and got[6] and got[7]
    # offsets beyond the total: rejected without a read
    off = [0, 3, 6, 9, 12, 15, 18, 21, 99]
    got, st = run(off)
    assert got[:7] == [True] * 7 and not got[7] and st[7] & 0x04


@pytest.mark.parametrize("n,kp", [(13200, 4), (16384, 4), (20000, 3), (33000, 2)])
def test_aggregate_verify_batch_above_a_round_of_pairs(n, kp):
    """n + n kp one-pair Miller loops = more than a round of lanes (65 536): whole rounds first, what is left of the last round as a launch of its own --
    on lane pairs when it is at most half a round (launch_miller_single). All items valid, then one message flipped: only that item fails
    (bench.aggregate_verify_leg builds the signatures on the device)."""
    import torch
    import bench
    from milagro_bls_amd import _native as N
    ctx = N.default_context()
    r = bench.aggregate_verify_leg(ctx, N.lib(), torch.device("cuda:0"), None, n=n, kp=kp)
    assert r["correct"] is True, r


def test_a_remainder_on_the_wave_engine_never_grows_the_workspace_under_the_round(mb):
    """a pass on the wave engine with k >= 32 keys sums its keys on eight lanes per item (n + 8 n workspace items). In a rounds-then-remainder plan the
    remainder's 9 r items can exceed everything the round needs: the entry now sizes the workspace for the WHOLE plan before the first pass is queued
    (mbls_plan_workspace_items is that figure) and a later pass never reallocates -- on a FRESH context (nothing reserved), small rounds, against the oracle, byte keys
    and table indices; the same with the workspace pre-reserved for what the passes' own workspace_items say (too small: the call grows it ONCE, up front)."""
    import torch
    from milagro_bls_amd import _native as N
    nt = min(32, os.cpu_count() or 8)
    n, k = 168, 32
    b = helpers.make_batch(n, k, fmt=1, seed=77, pool_n=64, nthreads=nt)
    want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, 1, nthreads=nt)
    assert want == b.expect
    dev = torch.device("cuda:0")
    for prereserve in (False, True):
        ctx = N.Context(0)
        try:
            ctx.set_round_items(128); ctx.set_tracks(0); ctx.set_coop_max_items(100); ctx.set_coop_hash_max_items(100)
            L = ctx.limits()
            mode, ps = N.plan_batch(n, L)
            assert mode == N.BATCH_ROUNDS_THEN_REST and ps[1]["items"] == 40 and ps[1]["pairing"] == N.PAIRING_WAVE and ps[0]["pairing"] != N.PAIRING_WAVE
            assert N.plan_workspace_items(n, k, True, L) == 9 * 40 > max(p["workspace_first"] + p["workspace_items"] for p in ps)
            if prereserve:
                ctx.reserve(max(p["workspace_first"] + p["workspace_items"] for p in ps))
            got, st, bits = _device_call(N, ctx, dev, b.sigs, b.msgs, b.pks, n, k)
            assert got == want and bits == [int(x) for x in got]
            tab = N.KeyTable(ctx, capacity_hint=n * k)
            first, errs = tab.append(b.pks, n * k, pk_format=1, validate=False)
            assert first == 0
            got2, st2, bits2 = _device_call(N, ctx, dev, b.sigs, b.msgs, None, n, k, table=tab, idx=list(range(n * k)))
            assert got2 == want and st2 == st
            tab.close()
        finally:
            ctx.close()
