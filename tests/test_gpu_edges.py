"""GPU parity, second layer: the edge cases the reference's checks admit but its tests never feed (infinity keys in
Signature::verify, infinity / undecodable members in verify_multiple and aggregate_verify), KeyValidate on garbage, the
resident key table against the byte-format path, batched signature aggregation, context sharing between threads, argument
validation of the C ABI, the config-5 shard size, and a seeded randomised sweep. Everything goes through the C ABI and is
compared with the oracle bit for bit."""
import ctypes as C
import random
import threading

import numpy as np
import pytest

import helpers
import orc

pytestmark = pytest.mark.gpu

G1_INF_C = bytes([0xC0]) + bytes(47)
G1_INF_U = bytes([0x40]) + bytes(95)


@pytest.fixture(scope="module")
def mb():
    from milagro_bls_amd import batch, _native
    _native.default_context()
    return batch


@pytest.fixture(scope="module")
def N():
    from milagro_bls_amd import _native
    return _native


def _keys(rnd, n):
    sks = [rnd.randrange(1, helpers.R) for _ in range(n)]
    pk96 = orc.batch_sk_to_pk(b"".join(s.to_bytes(32, "big") for s in sks), n, 1, nthreads=8)
    return sks, [pk96[96 * i:96 * i + 96] for i in range(n)]


# ------------------------------------------------------------------------------------------------ Signature::verify, pk = infinity
@pytest.mark.usefixtures("engine")
def test_verify_batch_with_infinite_public_keys(mb):
    """reference src/signature.rs:27-40 has no infinity check on the key: (sig, msg, pk = infinity) reaches the pairing, where an
    infinite argument contributes 1, so the item verifies iff e(sig, -G1) = 1 iff sig = infinity (and sig must be in G2)."""
    rnd = random.Random(11)
    sks, pks = _keys(rnd, 4)
    msgs = [rnd.randbytes(32) for _ in range(8)]
    sig = lambda i, m: orc.g2_compress(orc.sign(m, sks[i]))
    items = [   # (sig, msg, pk96)
        (sig(0, msgs[0]), msgs[0], pks[0]),              # valid
        (sig(0, msgs[1]), msgs[1], G1_INF_U),            # real signature, infinite key -> false
        (helpers.G2_INF, msgs[2], G1_INF_U),             # infinite signature, infinite key -> true (both pairings are 1)
        (helpers.G2_INF, msgs[3], pks[1]),               # infinite signature, real key -> false
        (sig(1, msgs[4]), msgs[4], pks[1]),              # valid
        (sig(2, msgs[5]), msgs[5], pks[3]),              # wrong key
    ]
    want = [orc.verify(orc.g2_from_compressed(s)[1], m, p) for s, m, p in items]
    assert want == [True, False, True, False, True, False]
    n = len(items)
    sigs = b"".join(i[0] for i in items); ms = b"".join(i[1] for i in items)
    got_u, _ = mb.verify_batch(sigs, ms, b"".join(i[2] for i in items), n, pk_format=1)
    got_c, _ = mb.verify_batch(sigs, ms, b"".join(orc.g1_compress(i[2]) for i in items), n, pk_format=0)
    assert got_u == want and got_c == want
    # the same items through fast_aggregate_verify (one-key sets): there the infinite key IS rejected (src/aggregates.rs:196-198)
    got_f, st = mb.fast_aggregate_verify_batch(sigs, ms, b"".join(i[2] for i in items), n, 1, pk_format=1)
    want_f = [orc.fast_aggregate_verify(orc.g2_from_compressed(s)[1], m, [p]) for s, m, p in items]
    assert got_f == want_f == [True, False, False, False, True, False] and st[2] & 0x08


# ------------------------------------------------------------------------------------------------ verify_multiple / aggregate_verify edges
def _vm_sets(rnd, n):
    sks, pks = _keys(rnd, n)
    msgs = [rnd.randbytes(32) for _ in range(n)]
    sigs = [orc.g2_compress(orc.sign(m, s)) for m, s in zip(msgs, sks)]
    rands = [rnd.randrange(1, 1 << 63) for _ in range(n)]
    return sks, pks, msgs, sigs, rands


def _vm_gpu(N, sigs, apks, msgs, rands):
    ctx = N.default_context()
    n = len(sigs)
    rr = (C.c_uint64 * max(1, n))(*rands)
    return bool(N.lib().mbls_verify_multiple_aggregate_signatures(ctx.handle, N.cbuf(b"".join(sigs)), N.cbuf(b"".join(apks)), N.cbuf(b"".join(msgs)),
                                                                 32, None, rr, n))


def _vm_orc(sigs, apks, msgs, rands):
    dec = [orc.g2_from_compressed(s) for s in sigs]
    if any(e for e, _ in dec):
        return None
    return orc.verify_multiple([(d[1], a, m) for d, a, m in zip(dec, apks, msgs)], rands)


@pytest.mark.usefixtures("engine")
def test_verify_multiple_edge_members_vs_oracle(N, vectors):
    """reference src/aggregates.rs:261-316 with members its tests never contain: an infinite signature, an infinite
    aggregate key, both at once in one set, a signature outside G2, a wrong key -- against the oracle with the same scalars."""
    rnd = random.Random(12)
    sks, pks, msgs, sigs, rands = _vm_sets(rnd, 7)
    base = (list(sigs), list(pks), list(msgs))
    assert _vm_gpu(N, *base, rands) is True and _vm_orc(*base, rands) is True

    def variant(f):
        s, a, m = list(sigs), list(pks), list(msgs)
        f(s, a, m)
        got, want = _vm_gpu(N, s, a, m, rands), _vm_orc(s, a, m, rands)
        assert got == want, (got, want)
        return got
    assert variant(lambda s, a, m: s.__setitem__(2, helpers.G2_INF)) is False                    # a set loses its signature
    assert variant(lambda s, a, m: a.__setitem__(3, G1_INF_U)) is False                          # a set loses its key
    assert variant(lambda s, a, m: (s.__setitem__(4, helpers.G2_INF), a.__setitem__(4, G1_INF_U))) is True   # the set contributes 1 on both sides
    assert variant(lambda s, a, m: a.__setitem__(1, pks[0])) is False
    probe = bytes.fromhex(vectors["model"]["g2_subgroup_probes"][0]["compressed"])
    assert variant(lambda s, a, m: s.__setitem__(6, probe)) is False                              # src/aggregates.rs:274-276
    # all sets infinite on both sides: the product of no pairings is 1
    assert _vm_gpu(N, [helpers.G2_INF] * 3, [G1_INF_U] * 3, msgs[:3], rands[:3]) is True
    assert _vm_orc([helpers.G2_INF] * 3, [G1_INF_U] * 3, msgs[:3], rands[:3]) is True
    # a member whose bytes do not decode: a reference caller could not have built the object; the ABI answers false
    bad_sig = bytes([sigs[0][0] & 0x7F]) + sigs[0][1:]
    assert _vm_gpu(N, [bad_sig] + sigs[1:], pks, msgs, rands) is False
    bad_pk = bytes(48) + bytes([1]) + bytes(47)        # (0, 2^376): not on the curve
    assert _vm_gpu(N, sigs, [bad_pk] + pks[1:], msgs, rands) is False


def _vm_gpu_rng(N, sigs, apks, msgs, rands):
    """mbls_verify_multiple_aggregate_signatures_rng with a source that hands out `rands` in order -> (bool, scalars asked for, calls)"""
    asked = []

    def draw(_user, out, count):
        for i in range(count):
            out[i] = rands[i]
        asked.append(int(count))
    cb = N.SCALAR_SOURCE(draw)
    n = len(sigs)
    got = N.lib().mbls_verify_multiple_aggregate_signatures_rng(N.default_context().handle, N.cbuf(b"".join(sigs)), N.cbuf(b"".join(apks)), N.cbuf(b"".join(msgs)),
                                                                32, None, n, cb, None)
    return bool(got), sum(asked), len(asked)


@pytest.mark.usefixtures("engine")
def test_verify_multiple_rng_entry_keeps_the_reference_order(N, vectors):
    """reference src/aggregates.rs:261-316 in one call: the same bool as the entry that takes the scalars (and as the oracle), and the source is asked
    exactly once for the scalars of the sets in front of the first signature outside G2 (:272-287) -- all of them when there is none."""
    rnd = random.Random(41)
    sks, pks, msgs, sigs, rands = _vm_sets(rnd, 9)
    assert _vm_gpu_rng(N, sigs, pks, msgs, rands) == (True, 9, 1) and _vm_orc(sigs, pks, msgs, rands) is True
    probe = bytes.fromhex(vectors["model"]["g2_subgroup_probes"][0]["compressed"])

    def variant(f):
        s, a, m = list(sigs), list(pks), list(msgs)
        f(s, a, m)
        got = _vm_gpu_rng(N, s, a, m, rands)
        assert got[0] == _vm_gpu(N, s, a, m, rands) == bool(_vm_orc(s, a, m, rands)), got
        return got
    assert variant(lambda s, a, m: a.__setitem__(1, pks[0])) == (False, 9, 1)                       # rejected by the pairing check: every scalar drawn
    assert variant(lambda s, a, m: s.__setitem__(8, sigs[7])) == (False, 9, 1)
    assert variant(lambda s, a, m: s.__setitem__(5, probe)) == (False, 5, 1)                        # :274-276: sets 0..4 had their scalars drawn
    assert variant(lambda s, a, m: (s.__setitem__(5, probe), s.__setitem__(7, probe))) == (False, 5, 1)
    assert variant(lambda s, a, m: s.__setitem__(0, probe)) == (False, 0, 0)                        # nothing drawn, the source never called
    assert variant(lambda s, a, m: s.__setitem__(2, helpers.G2_INF)) == (False, 9, 1)               # infinity is in G2 (passes :274), the pairing check rejects
    assert variant(lambda s, a, m: (s.__setitem__(4, helpers.G2_INF), a.__setitem__(4, G1_INF_U))) == (True, 9, 1)
    assert variant(lambda s, a, m: a.__setitem__(3, G1_INF_U)) == (False, 9, 1)
    # bytes that do not decode stop the loop like a signature outside G2 (a reference caller could not have built the object)
    bad_sig = bytes([sigs[3][0] & 0x7F]) + sigs[3][1:]
    assert _vm_gpu_rng(N, sigs[:3] + [bad_sig] + sigs[4:], pks, msgs, rands) == (False, 3, 1)
    # a zero from the source fails the batch (the mirrors draw until nonzero, :280-287); an empty batch is true and asks for nothing
    assert _vm_gpu_rng(N, sigs, pks, msgs, rands[:4] + [0] + rands[5:]) == (False, 9, 1)
    assert _vm_gpu_rng(N, [], [], [], []) == (True, 0, 0)
    assert N.lib().mbls_verify_multiple_aggregate_signatures_rng(N.default_context().handle, N.cbuf(b"".join(sigs)), N.cbuf(b"".join(pks)), N.cbuf(b"".join(msgs)),
                                                                 32, None, 9, N.SCALAR_SOURCE(0), None) == 0          # no source: refused
    # any 64-bit nonzero scalars, messages behind an offset table
    moff = (C.c_uint64 * 10)(*[32 * i for i in range(10)])
    big = [(1 << 63), (1 << 64) - 1] + rands[2:]
    cb = N.SCALAR_SOURCE(lambda _u, out, count: [out.__setitem__(i, big[i]) for i in range(count)] and None)
    assert N.lib().mbls_verify_multiple_aggregate_signatures_rng(N.default_context().handle, N.cbuf(b"".join(sigs)), N.cbuf(b"".join(pks)), N.cbuf(b"".join(msgs)),
                                                                 0, moff, 9, cb, None) == 1


@pytest.mark.parametrize("n", [2560, 2561, 20000, 33000])
def test_verify_multiple_rng_entry_large_batches(N, n):
    """the same entry at the last size whose signature chain runs on lane pairs (2 560 sets: k_sig2 / k_blind_sig2_d) and the first on single lanes, where the
    chains run side by side with two lanes per message (20 000 sets), and one after the other (33 000: more than half a round)"""
    import torch, bench
    ctx = N.default_context(); dev = torch.device("cuda:0")
    d_sigs, d_msgs, d_pks, _ = bench.build_inputs(ctx, dev, n, 1, N.PK_UNCOMPRESSED, rank=5, negatives=False)
    sigs, msgs, apks = bytes(d_sigs.cpu().numpy().tobytes()), bytes(d_msgs.cpu().numpy().tobytes()), bytes(d_pks.cpu().numpy().tobytes())
    rnd = random.Random(n)
    rands = [rnd.randrange(1, 1 << 64) for _ in range(n)]
    rr = (C.c_uint64 * n)(*rands)
    asked = []

    def draw(_user, out, count):
        C.memmove(out, rr, 8 * count); asked.append(int(count))
    cb = N.SCALAR_SOURCE(draw)
    call = lambda s: N.lib().mbls_verify_multiple_aggregate_signatures_rng(ctx.handle, N.cbuf(s), N.cbuf(apks), N.cbuf(msgs), 32, None, n, cb, None)
    assert call(sigs) == 1 and asked == [n]
    assert N.lib().mbls_verify_multiple_aggregate_signatures(ctx.handle, N.cbuf(sigs), N.cbuf(apks), N.cbuf(msgs), 32, None, rr, n) == 1
    k = n - 7
    swapped = sigs[:96 * k] + sigs[96 * (k - 1):96 * k] + sigs[96 * (k + 1):]                        # set k carries its neighbour's signature: in G2, wrong
    asked.clear()
    assert call(swapped) == 0 and asked == [n]
    assert N.lib().mbls_verify_multiple_aggregate_signatures(ctx.handle, N.cbuf(swapped), N.cbuf(apks), N.cbuf(msgs), 32, None, rr, n) == 0


def test_verify_multiple_scalar_requirements(N):
    """The blinding scalars are the security of the batch check: NULL is refused and a zero scalar fails the call (the reference
    draws until nonzero, src/aggregates.rs:280-287)."""
    import torch
    rnd = random.Random(13)
    sks, pks, msgs, sigs, rands = _vm_sets(rnd, 3)
    ctx = N.default_context()
    dev = torch.device("cuda:0")
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_s, d_a, d_m = t(b"".join(sigs)), t(b"".join(pks)), t(b"".join(msgs))
    d_res = torch.full((8,), 7, dtype=torch.uint8, device=dev); d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = N.lib().mbls_verify_multiple_aggregate_signatures_device(ctx.handle, d_s.data_ptr(), d_a.data_ptr(), d_m.data_ptr(), 32, None, None, 3, d_res.data_ptr(), d_st.data_ptr(), None)
    assert rc == N.ERR_ARGUMENT                                  # no scalars at all: refused on the host
    d_r = torch.tensor([rands[0], 0, rands[2]], dtype=torch.int64, device=dev)
    rc = N.lib().mbls_verify_multiple_aggregate_signatures_device(ctx.handle, d_s.data_ptr(), d_a.data_ptr(), d_m.data_ptr(), 32, None, d_r.data_ptr(), 3, d_res.data_ptr(), d_st.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0 and int(d_res[0].item()) == 0 and (int(d_st[0].item()) & 0x80)      # a zero scalar: the device entry only enqueues; result 0 + MBLS_ST_BAD_SCALAR
    rr0 = (C.c_uint64 * 3)(rands[0], 0, rands[2])
    assert N.lib().mbls_verify_multiple_aggregate_signatures(ctx.handle, N.cbuf(b"".join(sigs)), N.cbuf(b"".join(pks)), N.cbuf(b"".join(msgs)), 32, None, rr0, 3) == 0
    d_r = torch.tensor(rands, dtype=torch.int64, device=dev)
    rc = N.lib().mbls_verify_multiple_aggregate_signatures_device(ctx.handle, d_s.data_ptr(), d_a.data_ptr(), d_m.data_ptr(), 32, None, d_r.data_ptr(), 3, d_res.data_ptr(), d_st.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0 and int(d_res[0].item()) == 1 and int(d_st[0].item()) & ~0x20 == 0
    # forged pair (sig1 + D, sig2 - D): passes an unblinded check, must fail the blinded one
    D = orc.sign(b"d" * 32, 12345)
    f1 = orc.g2_compress(orc.g2_add(orc.g2_from_compressed(sigs[0])[1], D))
    f2 = orc.g2_compress(orc.g2_add(orc.g2_from_compressed(sigs[1])[1], orc.g2_mul(D, helpers.R - 1)))
    assert _vm_gpu(N, [f1, f2, sigs[2]], pks, msgs, rands) is False
    assert _vm_gpu(N, sigs, pks, msgs, rands) is True
    assert _vm_gpu(N, sigs, pks, msgs, [rands[0], 0, rands[2]]) is False
    # 2^63 is what i64::MIN.abs() wraps to in a release build of the reference (src/aggregates.rs:285): any nonzero 64-bit scalar works
    assert _vm_gpu(N, sigs, pks, msgs, [1 << 63, (1 << 64) - 1, 1]) is True


@pytest.mark.usefixtures("engine")
def test_aggregate_verify_edge_members_vs_oracle(N, vectors):
    """reference src/aggregates.rs:130-170 with infinite / invalid members, against the oracle."""
    from milagro_bls_amd import AggregateSignature, PublicKey, Signature
    rnd = random.Random(14)
    n = 5
    sks, pks = _keys(rnd, n)
    msgs = [rnd.randbytes(20 + 3 * i) for i in range(n)]             # ragged message lengths
    sig_pts = [orc.sign(m, s) for m, s in zip(msgs, sks)]
    agg = sig_pts[0]
    for p in sig_pts[1:]:
        agg = orc.g2_add(agg, p)
    agg_c = orc.g2_compress(agg)

    def both(sigc, ms, ks):
        got = AggregateSignature(sigc).aggregate_verify(ms, [PublicKey(k) for k in ks])
        e, pt = orc.g2_from_compressed(sigc)
        want = False if e else orc.aggregate_verify(pt, ms, ks)
        assert got == want, (got, want)
        return got
    assert both(agg_c, msgs, pks) is True
    assert both(agg_c, msgs, pks[:-1] + [G1_INF_U]) is False                      # a key replaced by infinity: its pairing drops out
    # drop signer 4 from the aggregate and give it the infinite key: the remaining product is complete again
    agg4 = sig_pts[0]
    for p in sig_pts[1:4]:
        agg4 = orc.g2_add(agg4, p)
    assert both(orc.g2_compress(agg4), msgs, pks[:4] + [G1_INF_U]) is True
    assert both(helpers.G2_INF, msgs, pks) is False                               # infinite signature against real keys
    assert both(helpers.G2_INF, msgs, [G1_INF_U] * n) is True                      # nothing on either side
    probe = bytes.fromhex(vectors["model"]["g2_subgroup_probes"][2]["compressed"])
    assert both(probe, msgs, pks) is False                                        # src/aggregates.rs:137-139
    assert both(agg_c, msgs[::-1], pks) is False
    assert AggregateSignature(agg_c).aggregate_verify(msgs[:-1], [PublicKey(k) for k in pks]) is False      # length mismatch, :132-134
    assert AggregateSignature(agg_c).aggregate_verify([], []) is False


# ------------------------------------------------------------------------------------------------ KeyValidate on garbage
def test_pk_decode_batch_validate_on_random_blobs(mb):
    """PublicKey::from_bytes (decode + KeyValidate, reference src/keys.rs:140-147, :181-186) on 256 random / garbage encodings:
    error class and decoded bytes equal the oracle's (which runs the full [r]P test)."""
    rnd = random.Random(15)
    sks, pks = _keys(rnd, 24)
    blobs = [bytes([rnd.choice([0x80, 0xA0, 0xC0, 0x00, 0xE0, 0x9f]) | rnd.getrandbits(5)]) + rnd.randbytes(47) for _ in range(200)]
    blobs += [orc.g1_compress(p) for p in pks]                           # valid keys in G1
    blobs += [G1_INF_C, bytes([0x80]) + bytes(47), bytes([0xA0]) + bytes(47)]     # infinity; (0, +-2): on the curve, outside G1
    while len(blobs) < 256:                                              # x-only randomness: on-curve points are ~half, almost none in G1
        blobs.append(bytes([0x80 | rnd.getrandbits(5) & 0x19]) + rnd.randbytes(47))
    out, errs = mb.pk_decode_batch(b"".join(blobs), len(blobs), validate=True)
    n_ok = 0
    for i, bl in enumerate(blobs):
        e, pt = orc.pk_from_bytes(bl)
        assert errs[i] == e, (i, bl.hex(), errs[i], e)
        if e == 0:
            assert out[96 * i:96 * i + 96] == pt
            n_ok += 1
    assert n_ok == 24
    # the unchecked decode of the same blobs accepts strictly more
    _, errs_u = mb.pk_decode_batch(b"".join(blobs), len(blobs), validate=False)
    assert sum(e == 0 for e in errs_u) > n_ok + 20


# ------------------------------------------------------------------------------------------------ resident key table
@pytest.mark.usefixtures("engine")
def test_keytable_indexed_verification_matches_byte_path_and_oracle(mb, N):
    rnd = random.Random(16)
    pool_n, n, k = 40, 130, 6
    sks, pks = _keys(rnd, pool_n)
    tab = N.KeyTable(capacity_hint=8)                                   # small on purpose: appends must grow the table
    comp = [orc.g1_compress(p) for p in pks]
    first, errs = tab.append(b"".join(comp[:25]), 25, pk_format=0, validate=True)
    assert first == 0 and errs == [0] * 25
    first, errs = tab.append(b"".join(pks[25:]), pool_n - 25, pk_format=1, validate=False)
    assert first == 25 and errs == [0] * (pool_n - 25) and len(tab) == pool_n
    got, e = tab.get(0, pool_n)
    assert e == [0] * pool_n and got == b"".join(pks)                   # as_uncompressed_bytes round trip of every entry
    # special entries: infinity (unchecked), a rejected key (infinity under KeyValidate), a key outside G1, undecodable bytes
    s_first, s_errs = tab.append(G1_INF_C, 1, pk_format=0, validate=False)
    assert s_errs == [0]
    IDX_INF = s_first
    v_first, v_errs = tab.append(G1_INF_C + bytes([0x80]) + bytes(47) + bytes([0x80]) + b"\xff" * 47, 3, pk_format=0, validate=True)
    assert v_errs == [N.ERR_INVALID_POINT] * 3                          # src/keys.rs:334-350
    IDX_BAD = v_first
    idxs, sigs, msgs = [], [], []
    for i in range(n):
        idx = rnd.sample(range(pool_n), k)
        m = rnd.randbytes(32)
        sigs.append(orc.g2_compress(orc.sign(m, sum(sks[j] for j in idx) % helpers.R)))
        msgs.append(m); idxs.append(idx)
    kinds = ["valid"] * n
    for i in range(3, n, 5):
        kind = ["wrong_key", "inf_member", "bad_member", "oob_index", "flip_msg", "dup_key"][(i // 5) % 6]
        kinds[i] = kind
        if kind == "wrong_key":
            idxs[i][0] = next(j for j in range(pool_n) if j not in idxs[i])
        elif kind == "inf_member":
            pass                                                        # rebuilt below: k-1 signers + the infinity entry
        elif kind == "bad_member":
            idxs[i][2] = IDX_BAD + (i % 3)
        elif kind == "oob_index":
            idxs[i][1] = len(tab) + 5
        elif kind == "flip_msg":
            msgs[i] = bytes([msgs[i][0] ^ 2]) + msgs[i][1:]
        elif kind == "dup_key":
            idxs[i][1] = idxs[i][0]
    # rebuild the inf_member items properly: k-1 signers + the infinity entry (sum unchanged) -> valid
    for i in range(n):
        if kinds[i] == "inf_member":
            idx = rnd.sample(range(pool_n), k - 1)
            m = msgs[i]
            sigs[i] = orc.g2_compress(orc.sign(m, sum(sks[j] for j in idx) % helpers.R))
            idxs[i] = idx[:2] + [IDX_INF] + idx[2:]
    flat = [j for idx in idxs for j in idx]
    got, st = mb.fast_aggregate_verify_batch_indexed(tab, b"".join(sigs), b"".join(msgs), flat, n, k)
    # byte-format path over the same keys (an invalid / out-of-range entry has no byte form: substitute undecodable bytes)
    all96, all_e = tab.get(0, len(tab))
    undec = bytes(48) + bytes([1]) + bytes(47)
    key_bytes = lambda j: undec if (j >= len(tab) or all_e[j]) else all96[96 * j:96 * j + 96]
    pkb = b"".join(key_bytes(j) for j in flat)
    got_b, st_b = mb.fast_aggregate_verify_batch(b"".join(sigs), b"".join(msgs), pkb, n, k, pk_format=1)
    want = orc.batch_fast_aggregate_verify(b"".join(sigs), b"".join(msgs), pkb, n, k, 1, nthreads=8)
    assert got == got_b == want
    assert st == st_b
    for i in range(n):
        exp = kinds[i] in ("valid", "inf_member")
        assert got[i] == exp, (i, kinds[i])
        if kinds[i] in ("bad_member", "oob_index"):
            assert st[i] & 0x04
    # ragged index lists incl. an empty one
    counts = [k, 0, k, 2, k]
    off = [0]
    rag = []
    for i, c in enumerate(counts):
        rag += idxs[i][:c]; off.append(off[-1] + c)
    got_r, st_r = mb.fast_aggregate_verify_batch_indexed(tab, b"".join(sigs[:5]), b"".join(msgs[:5]), rag, 5, offsets=off)
    assert got_r[0] == got[0] and got_r[2] == got[2] and got_r[4] == got[4] and got_r[1] is False and st_r[1] & 0x10 and got_r[3] is False
    tab.close()


def test_key_sum_routines_with_divergent_lanes(mb, N):
    """The generated key-sum routines (96-byte keys; table indices) when every lane of a wave meets something different: ragged set sizes
    (0..9 keys), and sets drawn from {G, 2G, -G, Q, infinity, an undecodable key} so that doublings (the running sum equals the next key),
    sums passing through infinity, infinite and undecodable keys fall on arbitrary lanes and positions. AggregatePublicKey::aggregate
    (src/aggregates.rs:29-39) per set against the oracle, then fast_aggregate_verify over the same sets through both key forms."""
    rnd = random.Random(33)
    R = helpers.R
    sk_of = {"g": 1, "g2": 2, "mg": R - 1, "q": 0xABCDEF, "inf": 0}
    pk_of = {name: orc.sk_to_pk(sk) if sk else G1_INF_U for name, sk in sk_of.items()}
    undec = bytes(48) + bytes([1]) + bytes(47)                          # (0, 2^376): not on the curve
    n = 150
    sets = []
    for i in range(n):
        cnt = rnd.randrange(10)
        sets.append([rnd.choice(["g", "g", "g2", "mg", "q", "inf", "bad"] if i % 5 else ["g", "g2", "mg", "q"]) for _ in range(cnt)])
    sets[3] = ["g", "g"]; sets[4] = ["g", "g", "g2", "g2", "g2"]; sets[5] = ["g", "mg"]; sets[6] = []; sets[7] = ["q"]
    key_bytes = lambda name: undec if name == "bad" else pk_of[name]
    flat = b"".join(key_bytes(x) for s_ in sets for x in s_)
    offsets = [0]
    for s_ in sets:
        offsets.append(offsets[-1] + len(s_))
    apks, st = mb.aggregate_public_keys_batch(flat, n, pk_format=1, pk_offsets=offsets)
    for i, s_ in enumerate(sets):
        good = [pk_of[x] for x in s_ if x != "bad"]                      # an undecodable key is skipped and reported
        e, want = orc.aggregate_pks(good) if good else (0, G1_INF_U)
        assert apks[96 * i:96 * i + 96] == want, (i, s_)
        assert bool(st[i] & 0x04) == ("bad" in s_) and bool(st[i] & 0x10) == (len(s_) == 0), (i, s_, hex(st[i]))
    # the same sets verified: signatures by the sum of the secret keys (an undecodable key makes the item fail)
    msgs = [rnd.randbytes(32) for _ in range(n)]
    sigs = [orc.g2_compress(orc.sign(msgs[i], sum(sk_of[x] for x in s_ if x != "bad") % R or 1)) for i, s_ in enumerate(sets)]
    got_b, st_b = mb.fast_aggregate_verify_batch(b"".join(sigs), b"".join(msgs), flat, n, pk_format=1, pk_offsets=offsets)
    tab = N.KeyTable()
    names = ["g", "g2", "mg", "q", "inf"]
    first, errs = tab.append(b"".join(pk_of[x] for x in names), len(names), pk_format=1, validate=False)
    assert errs == [0] * len(names)
    idx = [names.index(x) if x != "bad" else 1000 + i for i, s_ in enumerate(sets) for x in s_]      # outside the table = undecodable
    got_i, st_i = mb.fast_aggregate_verify_batch_indexed(tab, b"".join(sigs), b"".join(msgs), idx, n, offsets=offsets)
    for i, s_ in enumerate(sets):
        total = sum(sk_of[x] for x in s_ if x != "bad") % R
        expect = len(s_) > 0 and "bad" not in s_ and total != 0
        assert got_b[i] == got_i[i] == expect, (i, s_, hex(st_b[i]), hex(st_i[i]))
        assert (st_b[i] & 0x3C) == (st_i[i] & 0x3C), (i, s_, hex(st_b[i]), hex(st_i[i]))
    assert sum(got_b) > n // 4


@pytest.mark.usefixtures("engine")
def test_keytable_device_entry_at_batch_size(N):
    """2^13 x 128 keys through the indexed device entry against the byte-format device entry: identical results, bitmap and status."""
    import torch
    import bench
    ctx = N.default_context()
    dev = torch.device("cuda:0")
    n, k = 1 << 13, 128
    inp = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=5, return_indices=True)
    d_sigs, d_msgs, d_pks, expect, d_idx, tab = inp
    outs = []
    for mode in ("bytes", "indexed"):
        d_res = torch.zeros(n, dtype=torch.uint8, device=dev); d_bm = torch.zeros(n // 64, dtype=torch.int64, device=dev)
        d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        if mode == "bytes":
            ctx.check(N.lib().mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED,
                                                                      None, n, k, d_res.data_ptr(), d_bm.data_ptr(), d_st.data_ptr(), None))
        else:
            ctx.check(N.lib().mbls_fast_aggregate_verify_batch_indexed_device(ctx.handle, tab.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_idx.data_ptr(),
                                                                              None, n, k, d_res.data_ptr(), d_bm.data_ptr(), d_st.data_ptr(), None))
        torch.cuda.synchronize()
        outs.append((d_res.cpu(), d_bm.cpu(), d_st.cpu()))
    assert torch.equal(outs[0][0], expect) and torch.equal(outs[1][0], expect)
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.usefixtures("engine")
def test_key_buffer_alignment_paths_agree(N):
    """The generated key-sum routine fetches keys with 16-byte loads and is used for 4-byte aligned key buffers; any other address goes
    through the compiled lane body. The same 96-byte keys at offsets 0, 1, 2 and 4 inside a device buffer must give identical results."""
    import torch
    import bench
    ctx = N.default_context()
    dev = torch.device("cuda:0")
    n, k = 256, 16
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=9)[:4]
    flat = d_pks.reshape(-1)
    outs = []
    for off in (0, 1, 2, 4):
        buf = torch.zeros(flat.numel() + 8, dtype=torch.uint8, device=dev)
        buf[off:off + flat.numel()] = flat
        d_res = torch.zeros(n, dtype=torch.uint8, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        ctx.check(N.lib().mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, buf.data_ptr() + off, N.PK_UNCOMPRESSED,
                                                                  None, n, k, d_res.data_ptr(), None, d_st.data_ptr(), None))
        torch.cuda.synchronize()
        outs.append((d_res.cpu(), d_st.cpu()))
    for r, st in outs:
        assert torch.equal(r, expect) and torch.equal(st, outs[0][1])


# ------------------------------------------------------------------------------------------------ batched AggregateSignature::aggregate
def test_aggregate_signatures_batch_vs_oracle(mb):
    rnd = random.Random(17)
    n, k = 70, 5
    sks = [rnd.randrange(1, helpers.R) for _ in range(n * k)]
    msgs = [rnd.randbytes(32) for _ in range(n)]
    pts = [orc.sign(msgs[i // k], sks[i]) for i in range(n * k)]
    sigs = [orc.g2_compress(p) for p in pts]
    sigs[7] = helpers.G2_INF                                   # an infinite member
    pts[7] = orc.g2_from_compressed(helpers.G2_INF)[1]
    sigs[11] = sigs[10]; pts[11] = pts[10]                     # a doubled member (the complete addition doubles)
    sigs[16] = orc.g2_compress(orc.g2_mul(pts[15], helpers.R - 1)); pts[16] = orc.g2_from_compressed(sigs[16])[1]   # inverse pair -> passes through infinity
    out, errs = mb.aggregate_signatures_batch(b"".join(sigs), n, k)
    assert errs == [0] * n
    for i in range(n):
        acc = pts[k * i]
        for p in pts[k * i + 1:k * i + k]:
            acc = orc.g2_add(acc, p)
        assert out[96 * i:96 * i + 96] == orc.g2_compress(acc), i
    # an undecodable member -> the Signature::from_bytes error of that member; ragged sets incl. an empty one -> infinity
    bad = list(sigs[:6]); bad[4] = bytes([bad[4][0] & 0x7F]) + bad[4][1:]
    out2, errs2 = mb.aggregate_signatures_batch(b"".join(bad), 3, offsets=[0, 3, 3, 6])
    assert errs2 == [0, 0, 2] and out2[96:192] == helpers.G2_INF and out2[:96] != bytes(96) and out2[192:] == bytes(96)
    # the aggregate verifies: sum of k signatures on one message under the k keys
    pk96 = orc.batch_sk_to_pk(b"".join(s.to_bytes(32, "big") for s in sks[20:25]), 5, 1)
    got, _ = mb.fast_aggregate_verify_batch(out[96 * 4:96 * 5], msgs[4], pk96, 1, 5, pk_format=1)
    assert got == [True]


# ------------------------------------------------------------------------------------------------ threads, argument validation
def test_one_context_shared_by_threads(mb):
    """Every entry point takes the context's lock (the reference's functions are re-entrant): four threads hammer
    one context with different batches; each must get its own answers."""
    batches = [helpers.make_batch(40 + 13 * t, 3, fmt=t % 2, seed=300 + t) for t in range(4)]
    want = [orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, b.fmt, nthreads=4) for b in batches]
    errors = []

    def work(t):
        b = batches[t]
        for _ in range(4):
            got, _ = mb.fast_aggregate_verify_batch(b.sigs, b.msgs, b.pks, b.n, b.k, pk_format=b.fmt)
            if got != want[t]:
                errors.append(t)
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not errors


def test_argument_validation(mb, N):
    ctx = N.default_context()
    b = helpers.make_batch(4, 3, fmt=0, seed=31, negatives=False)
    res = N.outbuf(4)
    bad_off = (C.c_uint32 * 5)(0, 3, 2, 6, 9)                          # not non-decreasing
    rc = N.lib().mbls_fast_aggregate_verify_batch(ctx.handle, N.cbuf(b.sigs), N.cbuf(b.msgs), 32, None, N.cbuf(b.pks), 0, bad_off, 4, 0, res, None)
    assert rc == N.ERR_ARGUMENT and "offsets" in ctx.last_error()
    rc = N.lib().mbls_fast_aggregate_verify_batch(ctx.handle, N.cbuf(b.sigs), N.cbuf(b.msgs), 32, None, N.cbuf(b.pks), 7, None, 4, 3, res, None)
    assert rc == N.ERR_ARGUMENT
    rc = N.lib().mbls_fast_aggregate_verify_batch(ctx.handle, None, N.cbuf(b.msgs), 32, None, N.cbuf(b.pks), 0, None, 4, 3, res, None)
    assert rc == N.ERR_ARGUMENT
    # the context still works afterwards
    assert mb.fast_aggregate_verify_batch(b.sigs, b.msgs, b.pks, 4, 3)[0] == [True] * 4


# ------------------------------------------------------------------------------------------------ config 5 shard size
def test_config5_shard_2_17_items_128_keys(N):
    """BASELINE configs[4]: 2^20 items over 8 GPUs = 2^17 items x 128 keys per GPU. One shard through the device entry point:
    bitmap and results by construction (sign -> aggregate -> verify round trip, every 16th item corrupted in five ways), status
    classes, and a 64-item subsample pinned to the oracle."""
    import torch
    import bench
    ctx = N.default_context()
    dev = torch.device("cuda:0")
    n, k = 1 << 17, 128
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=6)
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev); d_bm = torch.zeros(n // 64, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.check(N.lib().mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None,
                                                              n, k, d_res.data_ptr(), d_bm.data_ptr(), d_st.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(d_res.cpu(), expect)
    bits = d_bm.cpu().numpy().view(np.uint64)
    unpacked = ((bits[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).reshape(-1).astype(np.uint8)
    assert (unpacked == expect.numpy()).all()
    st = d_st.cpu().numpy()
    assert (st[expect.numpy() == 1] & 0x5F == 0).all()                 # accepted items carry no rejection bit
    sel = list(range(n - 64, n))                                        # the last wave of the last round of workgroups
    sub = lambda t: t[sel].cpu().numpy().tobytes()
    got = orc.batch_fast_aggregate_verify(sub(d_sigs), sub(d_msgs), sub(d_pks), 64, k, 1, nthreads=8)
    assert got == [bool(x) for x in expect[sel].tolist()]


# ------------------------------------------------------------------------------------------------ seeded randomised sweep
@pytest.mark.usefixtures("engine")
@pytest.mark.parametrize("seed", [100, 101, 102])
def test_randomised_sweep_vs_oracle(mb, seed):
    """scripts/stress_parity.py as a test: many shapes (n around the wave size, odd key counts, both formats), seven rejection
    classes, every accept bit against the oracle."""
    for n, k, fmt in ((1, 3, 1), (63, 5, 0), (65, 2, 1), (129, 7, 1), (500, 4, 1), (200, 16, 0)):
        b = helpers.make_batch(n, k, fmt=fmt, seed=seed * 7 + n, pool_n=64)
        got, st = mb.fast_aggregate_verify_batch(b.sigs, b.msgs, b.pks, b.n, b.k, pk_format=fmt)
        want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, fmt, nthreads=8)
        assert got == want == b.expect, (seed, n, k, fmt)


# ------------------------------------------------------------------------------------------------ per-item message lengths
@pytest.mark.usefixtures("engine")
def test_ragged_message_lengths_vs_oracle(mb, N):
    """The reference takes any `msg: &[u8]` per call (src/aggregates.rs:177; its tests sign 0 .. 133 700 bytes, :436): the batch entries
    take one message buffer + an offset table. Lengths 0 .. 300 and one 133 700-byte item, fast_aggregate_verify and Signature::verify,
    byte keys and table indices, against the oracle item by item; a corrupted length must fail exactly its item."""
    rnd = random.Random(77)
    k = 3
    sks, pks = _keys(rnd, 8)
    lens = [0, 1, 31, 32, 33, 55, 56, 63, 64, 65, 119, 120, 127, 128, 129, 255, 256, 300, 133700, 7]
    n = len(lens)
    msgs = [rnd.randbytes(l) if l != 133700 else bytes([42]) * l for l in lens]
    idx = [rnd.sample(range(8), k) for _ in range(n)]
    sig192 = [orc.sign(msgs[i], sum(sks[j] for j in idx[i]) % helpers.R) for i in range(n)]
    sigs = [orc.g2_compress(s) for s in sig192]
    # negatives: a signature over the message minus its last byte / plus one byte
    msgs_v = list(msgs)
    msgs_v[5] = msgs[5][:-1]; msgs_v[9] = msgs[9] + b"\x00"; msgs_v[18] = msgs[18][:-1]
    want = [orc.fast_aggregate_verify(sig192[i], msgs_v[i], [pks[j] for j in idx[i]]) for i in range(n)]
    assert want == [i not in (5, 9, 18) for i in range(n)]
    offs = [0]
    for m in msgs_v:
        offs.append(offs[-1] + len(m))
    flat_pk = b"".join(b"".join(pks[j] for j in idx[i]) for i in range(n))
    got, st = mb.fast_aggregate_verify_batch(b"".join(sigs), b"".join(msgs_v), flat_pk, n, k, pk_format=1, msg_offsets=offs)
    assert got == want
    assert all((s & 0x40) != 0 for i, s in enumerate(st) if not want[i])
    # the same through a key table
    tab = N.KeyTable(capacity_hint=8)
    first, errs = tab.append(b"".join(pks), 8, pk_format=1, validate=True)
    assert not any(errs)
    got2, _ = mb.fast_aggregate_verify_batch_indexed(tab, b"".join(sigs), b"".join(msgs_v), [first + j for ix in idx for j in ix], n, k, msg_offsets=offs)
    assert got2 == want
    # a table that does not start at 0 (a shard of a larger buffer): only msgs[offs[0] .. offs[n]) is read
    pad = 13
    offs2 = [o + pad for o in offs]
    got3, _ = mb.fast_aggregate_verify_batch(b"".join(sigs), bytes(pad) + b"".join(msgs_v), flat_pk, n, k, pk_format=1, msg_offsets=offs2)
    assert got3 == want
    # Signature::verify over the same ragged messages (one key each)
    sig1 = [orc.g2_compress(orc.sign(msgs[i], sks[i % 8])) for i in range(n)]
    want1 = [orc.verify(orc.g2_from_compressed(sig1[i])[1], msgs_v[i], pks[i % 8]) for i in range(n)]
    got4, _ = mb.verify_batch(b"".join(sig1), b"".join(msgs_v), b"".join(pks[i % 8] for i in range(n)), n, pk_format=1, msg_offsets=offs)
    assert got4 == want1 == want
    # host entries refuse a table that runs backwards; the device entry rejects exactly that item
    bad = list(offs); bad[3], bad[4] = bad[4], bad[3]
    with pytest.raises(N.MblsError):
        mb.fast_aggregate_verify_batch(b"".join(sigs), b"".join(msgs_v), flat_pk, n, k, pk_format=1, msg_offsets=bad)
    import torch
    dev = torch.device("cuda:0")
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_s, d_m, d_p = t(b"".join(sigs)), t(b"".join(msgs_v)), t(flat_pk)
    d_off = torch.tensor(bad, dtype=torch.int64, device=dev)
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx = N.default_context()
    ctx.check(N.lib().mbls_fast_aggregate_verify_batch_device(ctx.handle, d_s.data_ptr(), d_m.data_ptr(), 0, d_off.data_ptr(), d_p.data_ptr(), 1, None, n, k,
                                                              d_res.data_ptr(), None, d_st.data_ptr(), None))
    torch.cuda.synchronize()
    res, stt = d_res.cpu().tolist(), d_st.cpu().tolist()
    assert (stt[3] & 0x100) and res[3] == 0
    # bad[3] > bad[4] only breaks item 3's own range; items 2 and 4 now cover different bytes and fail their pairing; the rest stand
    assert [bool(r) for i, r in enumerate(res) if i not in (2, 3, 4)] == [w for i, w in enumerate(want) if i not in (2, 3, 4)]


def test_verify_multiple_ragged_messages(N):
    """verify_multiple_aggregate_signatures with messages of different lengths (reference src/aggregates.rs:261-316 takes `&[u8]` per set)"""
    from milagro_bls_amd import api
    rnd = random.Random(5)
    sks, pks = _keys(rnd, 4)
    msgs = [b"", b"a", rnd.randbytes(100), bytes([42]) * 133700]
    sets, osets = [], []
    for i in range(4):
        s192 = orc.sign(msgs[i], sks[i])
        sets.append((api.AggregateSignature.from_bytes(orc.g2_compress(s192)), api.AggregatePublicKey.from_public_key(api.PublicKey.from_uncompressed_bytes(pks[i])), msgs[i]))
        osets.append((s192, pks[i], msgs[i]))
    assert orc.verify_multiple(osets, [3, 5, 7, 11]) is True
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(random.Random(1), sets) is True
    bad = list(sets); bad[2] = (sets[2][0], sets[2][1], msgs[2][:-1])
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(random.Random(1), bad) is False


# ------------------------------------------------------------------------------------------------ key table: ordering across streams
@pytest.mark.usefixtures("engine")
def test_keytable_append_and_verify_on_different_streams(N):
    """An append made on stream A must be visible to a verification enqueued on stream B right behind it (no host synchronisation in
    between), also when the append makes the table grow (the records move): the library orders them with an event / drains the device
    around the move (include/mbls.h, key table section)."""
    import torch
    dev = torch.device("cuda:0")
    ctx = N.default_context()
    rnd = random.Random(21)
    n, k, pool = 256, 4, 192
    sks, pks = _keys(rnd, pool)
    idx = [rnd.sample(range(pool), k) for _ in range(n)]
    msgs = [rnd.randbytes(32) for _ in range(n)]
    sigs = orc.batch_sign(b"".join((sum(sks[j] for j in ix) % helpers.R).to_bytes(32, "big") for ix in idx), b"".join(msgs), n, nthreads=8)
    want = [True] * n
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_s, d_m = t(sigs), t(b"".join(msgs))
    d_idx = torch.tensor([j for ix in idx for j in ix], dtype=torch.int32, device=dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for rep in range(3):
        tab = N.KeyTable(ctx, capacity_hint=64)                 # three appends of 64: the second and third make it grow
        d_errs = torch.full((pool,), 9, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(n, dtype=torch.uint8, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        # something long on stream A first, so that the appends behind it are still pending when stream B's work is enqueued
        d_p = t(b"".join(pks))
        ms = C.c_float()
        for c0 in range(0, pool, 64):
            st = sa if (c0 // 64) % 2 == 0 else None            # alternate between stream A and the null stream
            tab.append_device(d_p.data_ptr() + 96 * c0, 64, d_errs.data_ptr() + c0, pk_format=1, validate=True,
                              stream=C.c_void_p(st.cuda_stream) if st is not None else None)
        ctx.check(N.lib().mbls_fast_aggregate_verify_batch_indexed_device(ctx.handle, tab.handle, d_s.data_ptr(), d_m.data_ptr(), 32, None, d_idx.data_ptr(), None,
                                                                          n, k, d_res.data_ptr(), None, d_st.data_ptr(), C.c_void_p(sb.cuda_stream)))
        torch.cuda.synchronize()
        assert d_errs.cpu().tolist() == [0] * pool
        assert [bool(x) for x in d_res.cpu().tolist()] == want, "rep %d" % rep
        got, errs = tab.get(0, pool)
        assert got == b"".join(pks) and not any(errs)
        tab.close()


def test_keytable_outlives_its_context(N):
    """mbls_keytable_destroy after mbls_ctx_destroy must not touch freed memory (either order of destruction is allowed)"""
    ctx = N.Context(0)
    tab = N.KeyTable(ctx, capacity_hint=4)
    rnd = random.Random(3)
    _, pks = _keys(rnd, 2)
    first, errs = tab.append(b"".join(pks), 2, pk_format=1, validate=True)
    assert first == 0 and not any(errs) and len(tab) == 2
    ctx.close()
    assert len(tab) == 0
    tab.close()


# ------------------------------------------------------------------------------------------------ several devices behind one handle
def test_multi_keytable_append_is_all_or_nothing(N):
    """mbls_multi_keytable_append when one replica cannot follow: the replicas that did append drop the new records again, so a failed call
    changes no table and the indices stay the same on every device. Forced here by appending to replica 1 alone first (through
    mbls_multi_keytable_replica), which makes the replicas' first indices disagree."""
    import ctypes as C
    rnd = random.Random(31)
    _, pks = _keys(rnd, 6)
    m = N.MultiContext([0, 0])
    tab = N.MultiKeyTable(m, capacity_hint=8)
    first, errs = tab.append(b"".join(pks[:3]), 3, pk_format=1, validate=True)
    assert first == 0 and not any(errs) and len(tab) == 3
    rep = [N.lib().mbls_multi_keytable_replica(tab.handle, g) for g in (0, 1)]
    assert rep[0] and rep[1] and not N.lib().mbls_multi_keytable_replica(tab.handle, 2)
    size = lambda g: int(N.lib().mbls_keytable_size(rep[g]))
    assert (size(0), size(1)) == (3, 3)
    e1 = N.outbuf(1); f1 = C.c_uint64(0)
    assert N.lib().mbls_keytable_append(rep[1], N.cbuf(pks[3]), 1, 1, 1, C.byref(f1), e1) == 0 and size(1) == 4
    with pytest.raises(N.MblsError) as ei:
        tab.append(b"".join(pks[4:6]), 2, pk_format=1, validate=True)
    assert "disagree" in str(ei.value)
    assert (size(0), size(1)) == (3, 4)          # both replicas are where they were before the failed call
    tab.close(); m.close()


@pytest.mark.usefixtures("engine")
def test_multi_device_handle_matches_single_context_and_oracle(mb, N):
    """mbls_multi_* (device list + contiguous shards, one host thread per device) on the one GPU of this box with
    device_ids = {0, 0} and {0, 0, 0}: odd item counts, ragged key sets and ragged messages, byte keys and a replicated key table --
    results and status words identical to the single-context entry and to the oracle."""
    rnd = random.Random(8)
    pool = 24
    sks, pks = _keys(rnd, pool)
    n = 203                                                      # odd: shards of 101 + 102, 67 + 68 + 68
    cnts = [rnd.randrange(0, 6) for _ in range(n)]
    cnts[0] = 0; cnts[n - 1] = 5
    lens = [rnd.randrange(0, 90) for _ in range(n)]
    idx = [rnd.sample(range(pool), c) for c in cnts]
    msgs = [rnd.randbytes(l) for l in lens]
    sig192 = [orc.sign(msgs[i], (sum(sks[j] for j in idx[i]) % helpers.R) or 1) for i in range(n)]
    sigs = [orc.g2_compress(s) for s in sig192]
    for i in range(5, n, 9):                                     # negatives: flipped message byte / infinity signature / signature outside G2
        kind = (i // 9) % 3
        if kind == 0 and lens[i]:
            msgs[i] = bytes([msgs[i][0] ^ 1]) + msgs[i][1:]
        elif kind == 1:
            sigs[i] = helpers.G2_INF; sig192[i] = None
        else:
            sigs[i] = bytes.fromhex(helpers.load_vectors()["model"]["g2_subgroup_probes"][i % 3]["compressed"]); sig192[i] = None
    want = []
    for i in range(n):
        s = sig192[i] if sig192[i] is not None else orc.g2_from_compressed(sigs[i])[1]
        want.append(orc.fast_aggregate_verify(s, msgs[i], [pks[j] for j in idx[i]]))
    assert any(want) and not all(want) and want[0] is False
    koff, moff = [0], [0]
    for i in range(n):
        koff.append(koff[-1] + cnts[i]); moff.append(moff[-1] + len(msgs[i]))
    flat_pk = b"".join(b"".join(pks[j] for j in ix) for ix in idx)
    flat_idx = [j for ix in idx for j in ix]
    one, st_one = mb.fast_aggregate_verify_batch(b"".join(sigs), b"".join(msgs), flat_pk, n, pk_format=1, pk_offsets=koff, msg_offsets=moff)
    assert one == want
    for devs in ([0, 0], [0, 0, 0]):
        m = N.MultiContext(devs)
        assert len(m) == len(devs)
        m.reserve(n)
        got, st = mb.multi_fast_aggregate_verify_batch(m, b"".join(sigs), b"".join(msgs), flat_pk, n, pk_format=1, pk_offsets=koff, msg_offsets=moff)
        assert got == want and st == st_one
        tab = N.MultiKeyTable(m, capacity_hint=4)              # grows on every replica
        first, errs = tab.append(b"".join(pks), pool, pk_format=1, validate=True)
        assert first == 0 and not any(errs) and len(tab) == pool
        got2, st2 = mb.multi_fast_aggregate_verify_batch_indexed(m, tab, b"".join(sigs), b"".join(msgs), flat_idx, n, offsets=koff, msg_offsets=moff)
        assert got2 == want and st2 == st_one
        # uniform layout: k keys and 32 bytes per item
        b = helpers.make_batch(37, 3, fmt=1, seed=14)
        got3, _ = mb.multi_fast_aggregate_verify_batch(m, b.sigs, b.msgs, b.pks, b.n, b.k, pk_format=1)
        assert got3 == b.expect
        # Signature::verify shape, more devices than items
        s1 = orc.g2_compress(orc.sign(b"x", sks[0]))
        got4, _ = mb.multi_verify_batch(m, s1, b"x", pks[0], 1, pk_format=1, msg_len=1)
        assert got4 == [True]
        assert mb.multi_fast_aggregate_verify_batch(m, b"", b"", b"", 0, 1, pk_format=1)[0] == []
        tab.close(); m.close()


def test_signing_in_chunks_other_message_lengths_and_streams(mb):
    """mbls_sign_batch_device / mbls_sk_to_pk_batch_device (reference src/signature.rs:17-21, src/keys.rs:124-137) beyond one chunk of the
    four-lane signing pipeline (65 536 signatures) and of the table-driven key derivation (131 072 keys), with a message length that is not
    32, on a stream of the caller's: every signature verifies against its key, a sample is the oracle's, and the empty message signs too."""
    import torch
    from milagro_bls_amd import _native as N
    ctx = N.default_context(); lib = N.lib(); dev = torch.device("cuda:0")
    n, mlen = 131072 + 77, 45
    g = torch.Generator(device="cpu"); g.manual_seed(99)
    sks = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g); sks[:, 0] &= 0x3F; sks[:, 31] |= 1
    msgs = torch.randint(0, 256, (n, mlen), dtype=torch.uint8, generator=g)
    d_sk, d_msg = sks.to(dev), msgs.to(dev)
    d_sig = torch.zeros((n, 96), dtype=torch.uint8, device=dev); d_pk = torch.zeros((n, 48), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx.check(lib.mbls_sign_batch_device(ctx.handle, d_sk.data_ptr(), d_msg.data_ptr(), mlen, n, d_sig.data_ptr(), C.c_void_p(st.cuda_stream)))
        ctx.check(lib.mbls_sk_to_pk_batch_device(ctx.handle, d_sk.data_ptr(), 0, n, d_pk.data_ptr(), C.c_void_p(st.cuda_stream)))
    st.synchronize()
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.check(lib.mbls_verify_batch_device(ctx.handle, d_sig.data_ptr(), d_msg.data_ptr(), mlen, None, d_pk.data_ptr(), 0, n, d_res.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    assert bool(d_res.all().item())
    pick = [0, 1, 65535, 65536, 65537, 131071, 131072, n - 1]
    sub = lambda t: b"".join(bytes(t[i].cpu().numpy().tobytes()) for i in pick)
    assert sub(d_sig) == orc.batch_sign(sub(d_sk), sub(d_msg), len(pick), msg_len=mlen, nthreads=8)
    assert sub(d_pk) == orc.batch_sk_to_pk(sub(d_sk), len(pick), 0, nthreads=8)
    sk1 = (12345).to_bytes(32, "big")
    assert mb.sign_batch(sk1, b"", 1, msg_len=0) == orc.batch_sign(sk1, b"", 1, msg_len=0)
