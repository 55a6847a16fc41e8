"""The reference's own unit tests, restated against the host-side mirror of its API (milagro_bls_amd.api), running
on the GPU through the C ABI. Each test names the reference test it mirrors (file:line)."""
import random

import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import milagro_bls_amd as m
    m.default_context()
    return m


def keypair(api, rnd):
    sk = api.SecretKey.from_bytes(rnd.randrange(1, helpers.R).to_bytes(32, "big"))
    return api.Keypair(sk, api.PublicKey.from_secret_key(sk))


def kps_from_bytes(api, lists):
    out = []
    for b in lists:
        sk = api.SecretKey.from_bytes(bytes(b))
        out.append(api.Keypair(sk, api.PublicKey.from_secret_key(sk)))
    return out


def test_compression_round_trips(api, vectors):
    # src/amcl_utils.rs:81-145
    for h in vectors["reference"]["g1_compressed_round_trip"]["hex"]:
        assert api.PublicKey.from_bytes_unchecked(bytes.fromhex(h)).as_bytes().hex() == h
    for h in vectors["reference"]["g2_compressed_round_trip"]["hex"]:
        assert api.Signature.from_bytes(bytes.fromhex(h)).as_bytes().hex() == h
    assert api.AggregateSignature.new().as_bytes() == helpers.G2_INF
    assert api.AggregateSignature.from_bytes(helpers.G2_INF).as_bytes() == helpers.G2_INF


@pytest.mark.usefixtures("engine")
def test_basic_sign_verify(api):
    # src/signature.rs:63-86
    kp = keypair(api, random.Random(1))
    for m in (b"", b"a", b"an example"):
        sig = api.Signature.new(m, kp.sk)
        assert sig.verify(m, kp.pk)
        new_sig = api.Signature.from_bytes(sig.as_bytes())
        assert new_sig.as_bytes() == sig.as_bytes() and new_sig.verify(m, kp.pk)


@pytest.mark.usefixtures("engine")
def test_verification_failure_message(api):
    # src/signature.rs:89-100
    kp = keypair(api, random.Random(2))
    sig = api.Signature.new(b"Some msg", kp.sk)
    assert sig.verify(b"Other msg", kp.pk) is False and sig.verify(b"", kp.pk) is False


@pytest.mark.usefixtures("engine")
def test_readme(api, vectors):
    # src/signature.rs:103-125, src/keys.rs:311-330
    rd = vectors["reference"]["readme_sk"]
    sk = api.SecretKey.from_bytes(bytes(rd["bytes"]))
    assert sk.as_bytes() == bytes(rd["bytes"])                                 # src/keys.rs:214-223
    pk = api.PublicKey.from_secret_key(sk)
    sig = api.Signature.new(b"cats", sk)
    assert sig.verify(b"cats", pk)
    assert pk.as_bytes().hex() == vectors["model"]["readme"]["pk"] and sig.as_bytes().hex() == vectors["model"]["readme"]["sig"]
    assert sig.verify(b"cats", api.PublicKey.from_bytes(pk.as_bytes()))


def test_public_key_serialization(api, vectors):
    # src/keys.rs:226-282
    rnd = random.Random(3)
    for _ in range(4):
        pk = keypair(api, rnd).pk
        assert api.PublicKey.from_bytes(pk.as_bytes()).as_bytes() == pk.as_bytes()
        assert api.PublicKey.from_uncompressed_bytes(pk.as_uncompressed_bytes()).as_uncompressed_bytes() == pk.as_uncompressed_bytes()
    inf = api.PublicKey.from_bytes_unchecked(bytes([192]) + bytes(47))
    rec = api.PublicKey.from_uncompressed_bytes(inf.as_uncompressed_bytes())
    assert rec == inf and rec.is_infinity()
    for n in (1, 95, 97, 0):
        with pytest.raises(api.AmclError) as e:
            api.PublicKey.from_uncompressed_bytes(bytes([1]) * n)
        assert e.value.code == api.AmclError.InvalidG1Size
    with pytest.raises(api.AmclError) as e:
        api.PublicKey.from_uncompressed_bytes(bytes(47) + b"\x01" + bytes(47) + b"\x01")
    assert e.value.code == api.AmclError.InvalidPoint
    with pytest.raises(api.AmclError) as e:
        api.PublicKey.from_bytes(bytes(47))
    assert e.value.code == api.AmclError.InvalidG1Size
    with pytest.raises(api.AmclError) as e:
        api.Signature.from_bytes(bytes(95))
    assert e.value.code == api.AmclError.InvalidG2Size


def test_secret_key_from_bytes(api):
    # src/keys.rs:285-297
    for data, code in ((b"", api.AmclError.InvalidSecretKeySize), (bytes([1]) * 33, api.AmclError.InvalidSecretKeySize),
                       (bytes(32), api.AmclError.InvalidSecretKeyRange), (bytes([255]) * 32, api.AmclError.InvalidSecretKeyRange)):
        with pytest.raises(api.AmclError) as e:
            api.SecretKey.from_bytes(data)
        assert e.value.code == code
    sk = api.SecretKey.random(random.Random(5))
    assert len(sk.as_bytes()) == 32 and 0 < sk.as_raw() < helpers.R               # src/keys.rs:300-303
    with pytest.raises(api.AmclError):
        api.SecretKey.key_generate(bytes(31))                                     # src/keys.rs:46-48


def test_key_validate(api):
    # src/keys.rs:334-350
    zt = bytes([128]) + bytes(47)
    with pytest.raises(api.AmclError) as e:
        api.PublicKey.from_bytes(zt)
    assert e.value.code == api.AmclError.InvalidPoint
    assert api.PublicKey.from_bytes_unchecked(zt).key_validate() is False
    with pytest.raises(api.AmclError) as e:
        api.PublicKey.from_bytes(bytes([196]) + bytes(47))
    assert e.value.code == api.AmclError.InvalidPoint


@pytest.mark.usefixtures("engine")
def test_empty_and_split_zero_fast_aggregate_verify(api):
    # src/aggregates.rs:384-410
    agg = api.AggregateSignature.new()
    assert agg.fast_aggregate_verify(bytes(32), []) is False
    pk = api.PublicKey.from_secret_key(api.SecretKey.from_bytes((1).to_bytes(32, "big")))
    neg = api.PublicKey.from_secret_key(api.SecretKey.from_bytes(bytes.fromhex("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000000")))
    assert agg.fast_aggregate_verify(bytes(32), [pk, neg]) is False
    with pytest.raises(api.AmclError) as e:
        api.AggregatePublicKey.aggregate([])
    assert e.value.code == api.AmclError.AggregateEmptyPoints


def helper_test_aggregate_public_keys(api, control_kp, signing_kps, non_signing_kps, messages):
    # src/aggregates.rs:423-530
    subset = signing_kps[:-1]
    for message in messages:
        agg = api.AggregateSignature.new(); pks = []
        for kp in signing_kps:
            sig = api.Signature.new(message, kp.sk)
            assert sig.verify(message, kp.pk) and not sig.verify(message, control_kp.pk)
            agg.add(sig); pks.append(kp.pk)
        apk = api.AggregatePublicKey.into_aggregate(pks)
        assert agg.fast_aggregate_verify_pre_aggregated(message, apk)
        assert agg.fast_aggregate_verify_pre_aggregated(message, api.AggregatePublicKey.aggregate(pks[::-1]))
        dbl = agg.clone(); dbl.add(api.Signature.new(message, signing_kps[0].sk))
        assert not dbl.fast_aggregate_verify_pre_aggregated(message, apk)
        dist = api.AggregateSignature.new()
        for i, kp in enumerate(signing_kps):
            dist.add(api.Signature.new(b"different_msg!1" if i == 0 else message, kp.sk))
        assert not dist.fast_aggregate_verify_pre_aggregated(message, apk)
        sup = agg.clone(); sup.add(api.Signature.new(message, non_signing_kps[0].sk))
        assert not sup.fast_aggregate_verify_pre_aggregated(message, apk)
        sub_pks = [kp.pk for kp in subset]
        assert not agg.fast_aggregate_verify_pre_aggregated(message, api.AggregatePublicKey.aggregate(sub_pks))
        sub_pks.append(signing_kps[-1].pk)
        assert agg.fast_aggregate_verify_pre_aggregated(message, api.AggregatePublicKey.aggregate(sub_pks))
        assert not agg.fast_aggregate_verify_pre_aggregated(message, api.AggregatePublicKey.aggregate([kp.pk for kp in non_signing_kps]))
        assert agg.fast_aggregate_verify(message, pks)
    return agg, apk


@pytest.mark.usefixtures("engine")
def test_known_aggregate_public_keys(api, vectors):
    # src/aggregates.rs:555-609, including the 133 700-byte message, and the golden bytes of the model
    kk = vectors["reference"]["known_keys"]
    control = kps_from_bytes(api, kk["control"])[0]
    signing = kps_from_bytes(api, kk["signing"]); non_signing = kps_from_bytes(api, kk["non_signing"])
    msgs = [helpers.expand_msg(h) for h in kk["messages_hex"]]
    agg, apk = helper_test_aggregate_public_keys(api, control, signing, non_signing, msgs)
    last = vectors["model"]["aggregate_scenarios"][-1]
    assert agg.as_bytes().hex() == last["agg_sig"] and apk.point.hex() == last["agg_pk_uncompressed"]


def test_random_aggregate_public_keys(api):
    # src/aggregates.rs:533-552
    rnd = random.Random(7)
    helper_test_aggregate_public_keys(api, keypair(api, rnd), [keypair(api, rnd) for _ in range(6)], [keypair(api, rnd) for _ in range(6)], [b"Small msg"])


def test_add_aggregate_public_key_and_signature(api):
    # src/aggregates.rs:612-685
    rnd = random.Random(8)
    kps = [keypair(api, rnd) for _ in range(4)]
    a12 = api.AggregatePublicKey.aggregate([kps[0].pk, kps[1].pk]); a34 = api.AggregatePublicKey.aggregate([kps[2].pk, kps[3].pk])
    a1234 = api.AggregatePublicKey.aggregate([k.pk for k in kps])
    a12.add_aggregate(a34)
    assert a12 == a1234
    msg = bytes([1]) * 32
    sigs = [api.Signature.new(msg, k.sk) for k in kps]
    full = api.AggregateSignature.aggregate(sigs)
    s12 = api.AggregateSignature.new(); s12.add(sigs[0]); s12.add(sigs[1])
    s34 = api.AggregateSignature.new(); s34.add(sigs[2]); s34.add(sigs[3])
    s12.add_aggregate(s34)
    assert s12 == full and s12.fast_aggregate_verify_pre_aggregated(msg, a1234)
    assert api.AggregatePublicKey.from_public_key(kps[0].pk).point == kps[0].pk.point            # src/aggregates.rs:931-939
    assert api.AggregateSignature.from_signature(sigs[0]).point == sigs[0].point                  # src/aggregates.rs:942-950


def _sets(api, rnd, n, m, wrong_key=False):
    sets = []
    wrong = api.SecretKey.from_bytes(bytes([1]) * 32)
    for i in range(n):
        msg = bytes([i]) * 32; agg = api.AggregateSignature.new(); pks = []
        for _ in range(m):
            kp = keypair(api, rnd)
            agg.add(api.Signature.new(msg, wrong if wrong_key else kp.sk)); pks.append(kp.pk)
        sets.append((agg, api.AggregatePublicKey.into_aggregate(pks), msg))
    return sets


@pytest.mark.usefixtures("engine")
def test_verify_multiple_signatures(api, vectors):
    # src/aggregates.rs:688-805 (n = 10 sets x m = 3 keys) + the model's golden sets with pinned blinding scalars
    rnd = random.Random(9)
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(rnd, _sets(api, rnd, 10, 3)) is True
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(rnd, _sets(api, rnd, 10, 3, wrong_key=True)) is False
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(rnd, []) is True     # empty iterator: see DESIGN.md (unpinned)
    import ctypes as C
    from milagro_bls_amd import _native as N
    vm = vectors["model"]["verify_multiple"]
    for name in ("valid", "invalid"):
        s = vm[name]["sets"]
        rr = (C.c_uint64 * len(s))(*vm["rands"])
        got = N.lib().mbls_verify_multiple_aggregate_signatures(N.default_context().handle, N.cbuf(b"".join(bytes.fromhex(x["sig"]) for x in s)),
                                                                N.cbuf(b"".join(bytes.fromhex(x["apk"]) for x in s)),
                                                                N.cbuf(b"".join(bytes.fromhex(x["msg"]) for x in s)), 32, None, rr, len(s))
        assert bool(got) is vm[name]["result"]
    # a set whose signature is outside G2 fails the whole batch (src/aggregates.rs:274-276)
    sets = _sets(api, rnd, 3, 2)
    sets[1] = (api.AggregateSignature.from_bytes(bytes.fromhex(vectors["model"]["g2_subgroup_probes"][0]["compressed"])), sets[1][1], sets[1][2])
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(rnd, sets) is False


@pytest.mark.usefixtures("engine")
def test_external_eth2_spec_cases_on_the_gpu(api, vectors):
    """round 5: the published Eth2 BLS cases (sign, verify, aggregate, fast_aggregate_verify, aggregate_verify on the three standard keys; tests/golden/vectors.json
    "external", each string reproduced by the big-integer model) through the API mirror on the GPU, on every engine"""
    ext = vectors["external"]
    sks = [api.SecretKey.from_bytes(bytes.fromhex(kp["sk"])) for kp in ext["eth2_sk_to_pk"]]
    pks = [api.PublicKey.from_secret_key(sk) for sk in sks]
    assert [pk.as_bytes().hex() for pk in pks] == [kp["pk"] for kp in ext["eth2_sk_to_pk"]]
    by_hex = {kp["sk"]: i for i, kp in enumerate(ext["eth2_sk_to_pk"])}
    tampered = lambda h: bytes.fromhex(h)[:-4] + b"\xff" * 4

    def decode(cls, b):
        try:
            return cls.from_bytes(b)
        except api.AmclError:
            return None
    sig_ab = []
    for e in ext["eth2_sign_cases"] + [ext["eth2_sign"]]:
        i, msg = by_hex[e["sk"]], bytes.fromhex(e["msg"])
        sig = api.Signature.new(msg, sks[i])
        assert sig.as_bytes().hex() == e["sig"]
        assert sig.verify(msg, pks[i]) is True and sig.verify(msg, pks[(i + 1) % 3]) is False
        bad = decode(api.Signature, tampered(e["sig"]))
        assert bad is None or bad.verify(msg, pks[i]) is False
        if e["msg"] == "ab" * 32:
            sig_ab.append((i, sig))
    for e in ext["eth2_fast_aggregate_verify"]:
        k, msg = e["n_keys"], bytes.fromhex(e["msg"])
        agg = api.AggregateSignature.from_bytes(bytes.fromhex(e["sig"]))
        assert agg.fast_aggregate_verify(msg, pks[:k]) is True
        if k < 3:
            assert agg.fast_aggregate_verify(msg, pks[:k + 1]) is False
        bad = decode(api.AggregateSignature, tampered(e["sig"]))
        assert bad is None or bad.fast_aggregate_verify(msg, pks[:k]) is False
    assert api.AggregateSignature.new().fast_aggregate_verify(bytes.fromhex("ab" * 32), []) is False
    agg = api.AggregateSignature.new()
    for _, sg in sorted(sig_ab, key=lambda t: t[0]):
        agg.add(sg)
    assert agg.as_bytes().hex() == ext["eth2_fast_aggregate_verify"][2]["sig"]                    # aggregate_0xabab...
    av = ext["eth2_aggregate_verify"]
    msgs = [bytes.fromhex(m) for m in av["msgs"]]
    assert api.AggregateSignature.from_bytes(bytes.fromhex(av["sig"])).aggregate_verify(msgs, pks) is True
    bad = decode(api.AggregateSignature, tampered(av["sig"]))
    assert bad is None or bad.aggregate_verify(msgs, pks) is False
    assert api.AggregateSignature.new().aggregate_verify([], []) is False


class _CountingRng:
    """random.Random that counts the bytes drawn through getrandbits(8) (what the mirror's scalar loop uses)"""

    def __init__(self, seed, zero_first=0):
        self.r = random.Random(seed); self.n = 0; self.zero_first = zero_first

    def getrandbits(self, k):
        assert k == 8
        self.n += 1
        if self.zero_first:                      # the first draws come out as the all-zero scalar: retried (src/aggregates.rs:281)
            self.zero_first -= 1
            return 0
        return self.r.getrandbits(8)


def test_verify_multiple_draws_scalars_in_the_reference_order(api, vectors):
    """src/aggregates.rs:272-287: the reference tests set i's signature for the subgroup BEFORE it draws rand[i] and returns at the first
    signature outside G2 -- a rejected batch of 5 sets with the bad signature at position 2 has consumed exactly 2 x 8 random bytes; an accepted
    or pairing-rejected batch 8 bytes per set; a zero draw is retried (8 more bytes)."""
    rnd = random.Random(31)
    sets = _sets(api, rnd, 5, 2)
    rng = _CountingRng(1)
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(rng, sets) is True and rng.n == 5 * 8
    rng = _CountingRng(2, zero_first=8)
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(rng, sets) is True and rng.n == 6 * 8
    bad = list(sets)
    bad[2] = (api.AggregateSignature.from_bytes(bytes.fromhex(vectors["model"]["g2_subgroup_probes"][0]["compressed"])), sets[2][1], sets[2][2])
    rng = _CountingRng(3)
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(rng, bad) is False and rng.n == 2 * 8
    bad[0] = bad[2]
    rng = _CountingRng(4)
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(rng, bad) is False and rng.n == 0
    wrong = list(sets); wrong[4] = (sets[3][0], sets[4][1], sets[4][2])          # in G2, wrong signature: rejected by the pairing check, every scalar drawn
    rng = _CountingRng(5)
    assert api.AggregateSignature.verify_multiple_aggregate_signatures(rng, wrong) is False and rng.n == 5 * 8


@pytest.mark.parametrize("devs", [[0], [0, 0], [0, 0, 0]])
def test_verify_multiple_over_several_devices_draws_in_the_reference_order_too(api, vectors, devs):
    """mbls_multi_verify_multiple_aggregate_signatures_rng (one context per listed device: {0} = an RCCL communicator of one rank, {0, 0} / {0, 0, 0} = the host join on
    one GPU): every device tests its shard's signatures first, the scalars are asked for ONCE for the sets in front of the first bad signature of the whole batch --
    same bool and the same number of random bytes as the one-device call, wherever the bad signature's shard lies; more devices than sets leave empty shards."""
    from milagro_bls_amd import _native as N
    rnd = random.Random(32)
    sets = _sets(api, rnd, 7, 2)
    probe = api.AggregateSignature.from_bytes(bytes.fromhex(vectors["model"]["g2_subgroup_probes"][0]["compressed"]))
    m = N.MultiContext(devs)
    try:
        vm = lambda rng, s: api.AggregateSignature.verify_multiple_aggregate_signatures(rng, s, devices=m)
        rng = _CountingRng(1)
        assert vm(rng, sets) is True and rng.n == 7 * 8
        rng = _CountingRng(2, zero_first=8)
        assert vm(rng, sets) is True and rng.n == 8 * 8
        for pos in (0, 2, 3, 6):                              # first / a middle shard / the last set
            bad = list(sets); bad[pos] = (probe, sets[pos][1], sets[pos][2])
            rng = _CountingRng(3)
            assert vm(rng, bad) is False and rng.n == pos * 8, pos
            one = _CountingRng(3)
            assert api.AggregateSignature.verify_multiple_aggregate_signatures(one, bad) is False and one.n == rng.n
        wrong = list(sets); wrong[5] = (sets[4][0], sets[5][1], sets[5][2])        # in G2, wrong signature: the pairing check rejects, every scalar was drawn
        rng = _CountingRng(5)
        assert vm(rng, wrong) is False and rng.n == 7 * 8
        rng = _CountingRng(6)
        assert vm(rng, sets[:2]) is True and rng.n == 2 * 8          # fewer sets than devices ({0, 0, 0}): an empty shard contributes (1, infinity)
        assert vm(_CountingRng(7), []) is True
        # a generator that raises: the batch fails closed and the exception comes out afterwards
        class Boom(_CountingRng):
            def getrandbits(self, k):
                if self.n >= 20:
                    raise RuntimeError("rng broke")
                return super().getrandbits(k)
        with pytest.raises(RuntimeError):
            vm(Boom(8), sets)
    finally:
        m.close()


@pytest.mark.usefixtures("engine")
def test_aggregate_verify(api, vectors):
    # src/aggregates.rs:808-929
    rnd = random.Random(10)
    n = 10
    for repeat in (False, True):
        msgs = [bytes([i]) * 32 for i in range(n)]
        if repeat:
            msgs[-1] = bytes(32)
        kps = [keypair(api, rnd) for _ in range(n)]
        agg = api.AggregateSignature.aggregate([api.Signature.new(m, k.sk) for m, k in zip(msgs, kps)])
        assert agg.aggregate_verify(msgs, [k.pk for k in kps]) is True      # repeated message still verifies (:859-860)
    partial = api.AggregateSignature.aggregate([api.Signature.new(m, k.sk) for m, k in list(zip(msgs, kps))[:-1]])
    assert partial.aggregate_verify(msgs, [k.pk for k in kps]) is False
    kp = keypair(api, rnd); msg = bytes([1]) * 32
    one = api.AggregateSignature.from_signature(api.Signature.new(msg, kp.sk))
    assert one.aggregate_verify([msg], [kp.pk, kp.pk]) is False and one.aggregate_verify([msg, msg], [kp.pk]) is False
    assert one.aggregate_verify([], []) is False
    av = vectors["model"]["aggregate_verify"]
    pks = [api.PublicKey.from_uncompressed_bytes(bytes.fromhex(p)) for p in av["pks_uncompressed"]]
    for name in ("valid", "msg_repeat", "missing_signature"):
        c = av[name]
        assert api.AggregateSignature.from_bytes(bytes.fromhex(c["sig"])).aggregate_verify([bytes.fromhex(m) for m in c["msgs"]], pks) is c["result"]
