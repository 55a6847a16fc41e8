"""Pin the CPU oracle (oracle/bls_oracle.c): every golden vector the reference's tests hold for the path
(tests/golden/vectors.json "reference"), the external RFC 9380 / Eth2 pins, and the Python model's outputs."""
import subprocess
import sys
import os

import pytest

import helpers
import orc

ERR = {"InvalidG1Size": orc.ERR_G1_SIZE, "InvalidG2Size": orc.ERR_G2_SIZE, "InvalidPoint": orc.ERR_POINT}


def test_constants_are_regenerable():
    # committed constants == generator output (a hand-edited constant cannot go unnoticed)
    assert subprocess.call([sys.executable, os.path.join(helpers.ROOT, "oracle", "gen_constants.py"), "--check"]) == 0


def test_reference_g1_round_trip(vectors):
    # reference src/amcl_utils.rs:81-99 + benches/bls381_benches.rs:36-37 (full from_bytes on the third)
    for h in vectors["reference"]["g1_compressed_round_trip"]["hex"]:
        b = bytes.fromhex(h)
        e, pt = orc.g1_from_compressed(b)
        assert e == 0 and orc.g1_compress(pt) == b
        e, pt2 = orc.pk_from_bytes(b)
        assert e == 0 and pt2 == pt and orc.g1_key_validate(pt)
        e, pt3 = orc.g1_from_uncompressed(pt)
        assert e == 0 and pt3 == pt


def test_reference_g2_round_trip(vectors):
    # reference src/amcl_utils.rs:118-145
    for h in vectors["reference"]["g2_compressed_round_trip"]["hex"]:
        b = bytes.fromhex(h)
        e, pt = orc.g2_from_compressed(b)
        assert e == 0 and orc.g2_compress(pt) == b and orc.g2_subgroup_check(pt)


def test_reference_infinity_round_trips():
    # reference src/amcl_utils.rs:102-115
    e, pt = orc.g1_from_compressed(bytes([0xC0]) + bytes(47))
    assert e == 0 and pt == bytes([0x40]) + bytes(95) and orc.g1_compress(pt) == bytes([0xC0]) + bytes(47)
    e, pt = orc.g2_from_compressed(bytes([0xC0]) + bytes(95))
    assert e == 0 and orc.g2_compress(pt) == bytes([0xC0]) + bytes(95)


def test_reference_structural_cases(vectors):
    s = vectors["reference"]["structural"]
    # src/keys.rs:250-258
    e, pt = orc.g1_from_compressed(bytes.fromhex(s["pk_infinity_unchecked_ok"]["compressed"]))
    assert e == 0
    e2, pt2 = orc.g1_from_uncompressed(pt)
    assert e2 == 0 and pt2 == pt and pt[0] == 0x40
    # src/keys.rs:344-350
    assert orc.pk_from_bytes(bytes.fromhex(s["pk_infinity_bad_flags"]["compressed"]))[0] == orc.ERR_POINT
    # src/keys.rs:334-341
    b = bytes.fromhex(s["pk_zero_two"]["compressed"])
    assert orc.pk_from_bytes(b)[0] == orc.ERR_POINT and orc.g1_from_compressed(b)[0] == 0
    # src/keys.rs:261-273
    for n in s["pk_uncompressed_bad_sizes"]["sizes"]:
        assert orc.g1_from_uncompressed(bytes([1]) * n)[0] == orc.ERR_G1_SIZE
    # src/keys.rs:276-282
    assert orc.g1_from_uncompressed(bytes.fromhex(s["pk_uncompressed_off_curve"]["uncompressed"]))[0] == orc.ERR_POINT
    # decompress size errors: src/amcl_utils.rs:54-56, :70-72
    assert orc.g1_from_compressed(bytes(47))[0] == orc.ERR_G1_SIZE and orc.g2_from_compressed(bytes(95))[0] == orc.ERR_G2_SIZE
    # src/aggregates.rs:392-410
    z = s["split_zero"]
    pk1 = orc.sk_to_pk(int(z["sk_one"], 16)); pkm = orc.sk_to_pk(int(z["sk_minus_one"], 16))
    inf_sig = orc.g2_from_compressed(helpers.G2_INF)[1]
    assert orc.g1_compress(pkm).hex() == vectors["model"]["minus_g1"]
    assert orc.fast_aggregate_verify(inf_sig, bytes.fromhex(z["msg"]), [pk1, pkm]) is False
    # src/aggregates.rs:384-389
    assert orc.fast_aggregate_verify(inf_sig, bytes(32), []) is False
    assert orc.aggregate_pks([])[0] == orc.ERR_EMPTY          # src/aggregates.rs:30-32


def test_external_rfc9380_and_eth2(vectors):
    ext = vectors["external"]
    dst = bytes.fromhex(ext["rfc9380_J_10_1"]["dst"])
    for v in ext["rfc9380_J_10_1"]["vectors"]:
        h = orc.hash_to_g2(bytes.fromhex(v["msg"]), dst)
        want = b"".join(int(v[k], 16).to_bytes(48, "big") for k in ("x_c1", "x_c0", "y_c1", "y_c0"))
        assert h == want
    e = ext["eth2_sign"]
    sig = orc.sign(bytes.fromhex(e["msg"]), int(e["sk"], 16))
    assert orc.g2_compress(sig).hex() == e["sig"]
    assert orc.verify(sig, bytes.fromhex(e["msg"]), orc.sk_to_pk(int(e["sk"], 16)))
    for kp in ext["eth2_sk_to_pk"]:
        assert orc.g1_compress(orc.sk_to_pk(int(kp["sk"], 16))).hex() == kp["pk"]


def _tampered(sig_hex):
    return bytes.fromhex(sig_hex)[:-4] + b"\xff" * 4          # the Eth2 cases' "tampered signature": last four bytes replaced


def test_external_eth2_spec_cases(vectors):
    """round 5: the published Eth2 BLS cases sign / verify / aggregate / fast_aggregate_verify / aggregate_verify on the three standard keys, where the reference's
    semantics and the specification's coincide (the specification's extra KeyValidate of every key is not part of src/aggregates.rs:177-215)"""
    ext = vectors["external"]
    std = [int(kp["sk"], 16) for kp in ext["eth2_sk_to_pk"]]
    pks = [orc.sk_to_pk(sk) for sk in std]
    sigs = {}
    for e in ext["eth2_sign_cases"] + [ext["eth2_sign"]]:
        sk, msg = int(e["sk"], 16), bytes.fromhex(e["msg"])
        sig = orc.sign(msg, sk)
        assert orc.g2_compress(sig).hex() == e["sig"]
        sigs[(sk, msg)] = sig
        # verify_valid / verify_wrong_pubkey / verify_tampered_signature
        assert orc.verify(sig, msg, orc.sk_to_pk(sk)) is True
        assert orc.verify(sig, msg, orc.sk_to_pk(std[(std.index(sk) + 1) % 3])) is False
        err, bad = orc.g2_from_compressed(_tampered(e["sig"]))
        assert err != 0 or orc.verify(bad, msg, orc.sk_to_pk(sk)) is False
    for e in ext["eth2_fast_aggregate_verify"]:
        k, msg = e["n_keys"], bytes.fromhex(e["msg"])
        err, sig = orc.g2_from_compressed(bytes.fromhex(e["sig"])); assert err == 0
        assert orc.fast_aggregate_verify(sig, msg, pks[:k]) is True
        if k < 3:
            assert orc.fast_aggregate_verify(sig, msg, pks[:k + 1]) is False                  # fast_aggregate_verify_extra_pubkey
        err, bad = orc.g2_from_compressed(_tampered(e["sig"]))
        assert err != 0 or orc.fast_aggregate_verify(bad, msg, pks[:k]) is False               # ..._tampered_signature
    assert orc.fast_aggregate_verify(orc.g2_from_compressed(helpers.G2_INF)[1], bytes.fromhex("ab" * 32), []) is False   # ..._na_pubkeys_and_infinity_signature
    # aggregate_0xabab...: the sum of the three signatures on ab.. IS the three-key fast_aggregate_verify signature
    agg = None
    for sk in std:
        s1 = sigs[(sk, bytes.fromhex("ab" * 32))]
        agg = s1 if agg is None else orc.g2_add(agg, s1)
    assert orc.g2_compress(agg).hex() == ext["eth2_fast_aggregate_verify"][2]["sig"]
    av = ext["eth2_aggregate_verify"]
    msgs = [bytes.fromhex(m) for m in av["msgs"]]
    err, sig = orc.g2_from_compressed(bytes.fromhex(av["sig"])); assert err == 0
    assert orc.aggregate_verify(sig, msgs, pks) is True                                         # aggregate_verify_valid
    err, bad = orc.g2_from_compressed(_tampered(av["sig"]))
    assert err != 0 or orc.aggregate_verify(bad, msgs, pks) is False                            # aggregate_verify_tampered_signature
    assert orc.aggregate_verify(orc.g2_from_compressed(helpers.G2_INF)[1], [], []) is False       # aggregate_verify_na_pubkeys_and_infinity_signature


def test_model_hash_to_g2(vectors):
    for v in vectors["model"]["hash_to_g2"]:
        assert orc.g2_compress(orc.hash_to_g2(helpers.expand_msg(v["msg"]))).hex() == v["compressed"], v["msg"][:20]


def test_model_keys_and_readme(vectors):
    kk = vectors["reference"]["known_keys"]
    for group in ("control", "signing", "non_signing"):
        for b, want in zip(kk[group], vectors["model"]["known_pks"][group]):
            pk = orc.sk_to_pk(int.from_bytes(bytes(b), "big"))
            assert pk.hex() == want["uncompressed"] and orc.g1_compress(pk).hex() == want["compressed"]
    rd = vectors["reference"]["readme_sk"]
    sk = int.from_bytes(bytes(rd["bytes"]), "big")
    pk = orc.sk_to_pk(sk)
    sig = orc.sign(b"cats", sk)
    assert orc.g1_compress(pk).hex() == vectors["model"]["readme"]["pk"] and orc.g2_compress(sig).hex() == vectors["model"]["readme"]["sig"]
    # reference src/signature.rs:103-125
    assert orc.verify(sig, b"cats", pk) and orc.verify(sig, b"cats", orc.pk_from_bytes(orc.g1_compress(pk))[1])


def test_model_aggregate_scenarios(vectors):
    # the seven assertions of helper_test_aggregate_public_keys (reference src/aggregates.rs:423-530) on the fixed keys
    kk = vectors["reference"]["known_keys"]
    sign_pks = [bytes.fromhex(p["uncompressed"]) for p in vectors["model"]["known_pks"]["signing"]]
    non_pks = [bytes.fromhex(p["uncompressed"]) for p in vectors["model"]["known_pks"]["non_signing"]]
    ctrl = bytes.fromhex(vectors["model"]["known_pks"]["control"][0]["uncompressed"])
    for sc in vectors["model"]["aggregate_scenarios"]:
        msg = helpers.expand_msg(sc["msg"])
        dec = lambda h: orc.g2_from_compressed(bytes.fromhex(h))[1]
        sks = [int.from_bytes(bytes(b), "big") for b in kk["signing"]]
        for s, want in zip(sks, sc["individual_sigs"]):
            assert orc.g2_compress(orc.sign(msg, s)).hex() == want
        assert [orc.verify(dec(sc["individual_sigs"][i]), msg, sign_pks[i]) for i in range(2)] == sc["individual_verify_own_key"] == [True, True]
        assert orc.verify(dec(sc["individual_sigs"][0]), msg, ctrl) is False
        agg = dec(sc["agg_sig"])
        e, apk = orc.aggregate_pks(sign_pks)
        assert apk.hex() == sc["agg_pk_uncompressed"]
        acc = orc.g2_from_compressed(helpers.G2_INF)[1]
        for h in sc["individual_sigs"]:
            acc = orc.g2_add(acc, dec(h))
        assert orc.g2_compress(acc).hex() == sc["agg_sig"]
        pre = orc.fast_aggregate_verify_pre_aggregated
        assert pre(agg, msg, apk) is sc["full_set"] is True
        assert pre(agg, msg, orc.aggregate_pks(sign_pks[::-1])[1]) is sc["reversed_set"] is True
        assert pre(dec(sc["double_signed_sig"]), msg, apk) is sc["double_signed"] is False
        assert pre(dec(sc["distinct_msg_sig"]), msg, apk) is sc["distinct_msg"] is False
        assert pre(dec(sc["super_set_sig"]), msg, apk) is sc["super_set"] is False
        assert pre(agg, msg, orc.aggregate_pks(sign_pks[:-1])[1]) is sc["subset"] is False
        assert pre(agg, msg, orc.aggregate_pks(non_pks)[1]) is sc["non_signing"] is False
        assert orc.fast_aggregate_verify(agg, msg, sign_pks) is True


def test_model_subgroup_probes(vectors):
    for p in vectors["model"]["g2_subgroup_probes"]:
        e, pt = orc.g2_from_compressed(bytes.fromhex(p["compressed"]))
        assert e == 0 and orc.g2_subgroup_check(pt) is p["in_g2"]
    for p in vectors["model"]["g1_subgroup_probes"]:
        e, pt = orc.g1_from_compressed(bytes.fromhex(p["compressed"]))
        assert e == 0 and pt.hex() == p["uncompressed"] and orc.g1_key_validate(pt) is p["key_validate"]
        assert (orc.pk_from_bytes(bytes.fromhex(p["compressed"]))[0] == 0) is p["key_validate"]
    for h in vectors["model"]["g1_bad_compressed"]:
        assert orc.g1_from_compressed(bytes.fromhex(h))[0] == orc.ERR_POINT


def test_model_verify_multiple(vectors):
    vm = vectors["model"]["verify_multiple"]
    for name in ("valid", "invalid"):
        sets = [(orc.g2_from_compressed(bytes.fromhex(s["sig"]))[1], bytes.fromhex(s["apk"]), bytes.fromhex(s["msg"])) for s in vm[name]["sets"]]
        assert orc.verify_multiple(sets, vm["rands"]) is vm[name]["result"]


def test_model_aggregate_verify(vectors):
    av = vectors["model"]["aggregate_verify"]
    pks = [bytes.fromhex(p) for p in av["pks_uncompressed"]]
    for name in ("valid", "msg_repeat", "missing_signature"):
        c = av[name]
        sig = orc.g2_from_compressed(bytes.fromhex(c["sig"]))[1]
        assert orc.aggregate_verify(sig, [bytes.fromhex(m) for m in c["msgs"]], pks) is c["result"]
    sig = orc.g2_from_compressed(bytes.fromhex(av["valid"]["sig"]))[1]
    msgs = [bytes.fromhex(m) for m in av["valid"]["msgs"]]
    assert orc.aggregate_verify(sig, msgs[:1], pks[:2]) is False and orc.aggregate_verify(sig, msgs[:2], pks[:1]) is False   # src/aggregates.rs:132-134
    assert orc.aggregate_verify(sig, [], []) is False


def test_model_fast_aggregate_verify_batch(vectors):
    fb = vectors["model"]["fast_aggregate_verify_batch"]
    items, k = fb["items"], fb["k"]
    for fmt, key in ((orc.PK_COMPRESSED, "pks_compressed"), (orc.PK_UNCOMPRESSED, "pks_uncompressed")):
        got = orc.batch_fast_aggregate_verify(b"".join(bytes.fromhex(i["sig"]) for i in items), b"".join(bytes.fromhex(i["msg"]) for i in items),
                                              b"".join(bytes.fromhex(h) for i in items for h in i[key]), len(items), k, fmt, nthreads=4)
        assert got == [i["result"] for i in items]


def test_oracle_batch_matches_construction():
    b = helpers.make_batch(24, 5, fmt=0, seed=3)
    assert orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, 0, nthreads=8) == b.expect
    b = helpers.make_batch(12, 3, fmt=1, seed=4)
    assert orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, 1, nthreads=8) == b.expect
