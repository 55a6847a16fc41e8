"""CPU-only: the lane bodies of the HIP kernels, compiled as plain C++ (tests/host_emul), against the oracle and the
golden vectors. This is the safety net that lets kernel arithmetic be changed without a GPU; the GPU parity tests
(test_gpu_*.py) repeat the same comparisons through the C ABI on the real device."""
import ctypes as C
import random

import helpers
import orc
from helpers import cb, ob


def test_fp_mul_sqr(emul):
    rnd = random.Random(1)
    n = 300
    A = [rnd.randrange(helpers.P) for _ in range(n)]; B = [rnd.randrange(helpers.P) for _ in range(n)]
    A[:4] = [0, 1, helpers.P - 1, helpers.P - 1]; B[:4] = [5, helpers.P - 1, helpers.P - 1, 2]
    a = b"".join(x.to_bytes(48, "big") for x in A); b = b"".join(x.to_bytes(48, "big") for x in B)
    o = ob(48 * n); emul.emul_fp_mul(cb(a), cb(b), C.c_uint64(n), o, 0)
    assert all(int.from_bytes(bytes(o)[48 * i:48 * i + 48], "big") == A[i] * B[i] % helpers.P for i in range(n))
    emul.emul_fp_mul(cb(a), cb(b), C.c_uint64(n), o, 1)
    assert all(int.from_bytes(bytes(o)[48 * i:48 * i + 48], "big") == A[i] * A[i] % helpers.P for i in range(n))


def test_hash_to_g2_golden(emul, vectors):
    for v in vectors["model"]["hash_to_g2"]:
        m = helpers.expand_msg(v["msg"])
        if len(m) > 1000:
            continue    # long messages: same code path, covered once on the GPU
        o = ob(96); emul.emul_hash_to_g2(cb(m), len(m), C.c_uint64(1), o)
        assert bytes(o).hex() == v["compressed"], v["msg"][:16]


def test_sign_and_sk_to_pk(emul, vectors):
    e = vectors["external"]["eth2_sign"]
    o = ob(96); emul.emul_sign(cb(bytes.fromhex(e["sk"])), cb(bytes.fromhex(e["msg"])), 32, C.c_uint64(1), o)
    assert bytes(o).hex() == e["sig"]
    for e in vectors["external"]["eth2_sign_cases"]:
        o = ob(96); emul.emul_sign(cb(bytes.fromhex(e["sk"])), cb(bytes.fromhex(e["msg"])), 32, C.c_uint64(1), o)
        assert bytes(o).hex() == e["sig"]
    for kp in vectors["external"]["eth2_sk_to_pk"]:
        o = ob(48); emul.emul_sk_to_pk(cb(bytes.fromhex(kp["sk"])), 0, C.c_uint64(1), o)
        assert bytes(o).hex() == kp["pk"]


def test_codec_and_validation(emul, vectors):
    ref = vectors["reference"]
    for h in ref["g1_compressed_round_trip"]["hex"]:
        out, err = ob(96), ob(1); emul.emul_g1_decode(cb(bytes.fromhex(h)), 0, 1, C.c_uint64(1), out, err)
        assert err[0] == 0 and bytes(out) == orc.g1_from_compressed(bytes.fromhex(h))[1]
        o48 = ob(48); emul.emul_g1_compress(out, C.c_uint64(1), o48, err)
        assert err[0] == 0 and bytes(o48).hex() == h
    for h in ref["g2_compressed_round_trip"]["hex"]:
        err, g2 = ob(1), ob(1); emul.emul_g2_check(cb(bytes.fromhex(h)), C.c_uint64(1), err, g2)
        assert err[0] == 0 and g2[0] == 1
    for p in vectors["model"]["g2_subgroup_probes"]:
        err, g2 = ob(1), ob(1); emul.emul_g2_check(cb(bytes.fromhex(p["compressed"])), C.c_uint64(1), err, g2)
        assert err[0] == 0 and bool(g2[0]) is p["in_g2"]
    for p in vectors["model"]["g1_subgroup_probes"]:
        ok = ob(1); emul.emul_g1_key_validate(cb(bytes.fromhex(p["uncompressed"])), C.c_uint64(1), ok)
        assert bool(ok[0]) is p["key_validate"]
    s = ref["structural"]
    for name, validate, want in (("pk_infinity_bad_flags", 1, 3), ("pk_zero_two", 1, 3), ("pk_zero_two", 0, 0), ("pk_infinity_unchecked_ok", 0, 0)):
        out, err = ob(96), ob(1); emul.emul_g1_decode(cb(bytes.fromhex(s[name]["compressed"])), 0, validate, C.c_uint64(1), out, err)
        assert err[0] == want, name
    out, err = ob(96), ob(1); emul.emul_g1_decode(cb(bytes.fromhex(s["pk_uncompressed_off_curve"]["uncompressed"])), 1, 0, C.c_uint64(1), out, err)
    assert err[0] == 3
    for h in vectors["model"]["g1_bad_compressed"]:
        out, err = ob(96), ob(1); emul.emul_g1_decode(cb(bytes.fromhex(h)), 0, 0, C.c_uint64(1), out, err)
        assert err[0] == 3


def test_group_ops_edge_cases(emul):
    g = orc.sk_to_pk(1); g2 = orc.sk_to_pk(2); mg = orc.sk_to_pk(helpers.R - 1); inf = bytes([0x40]) + bytes(95)
    for a, b in ((g, g), (g, mg), (g, g2), (inf, g), (g, inf), (inf, inf)):
        out, err = ob(96), ob(1); emul.emul_g1_add(cb(a), cb(b), C.c_uint64(1), out, err)
        assert err[0] == 0 and bytes(out) == orc.g1_add(a, b)
    s1 = orc.g2_compress(orc.sign(b"m", 3)); s2 = orc.g2_compress(orc.sign(b"m", 4)); sm = orc.g2_compress(orc.sign(b"m", helpers.R - 3))
    for a, b in ((s1, s1), (s1, sm), (s1, s2), (helpers.G2_INF, s1), (s1, helpers.G2_INF), (helpers.G2_INF, helpers.G2_INF)):
        out, err = ob(96), ob(1); emul.emul_g2_add(cb(a), cb(b), C.c_uint64(1), out, err)
        want = orc.g2_compress(orc.g2_add(orc.g2_from_compressed(a)[1], orc.g2_from_compressed(b)[1]))
        assert err[0] == 0 and bytes(out) == want
    # aggregation with repeated keys and inverse pairs (complete addition semantics, reference src/aggregates.rs:34-37)
    keys = [g, g, mg, g2, g2, g2]
    out, st = ob(96), (C.c_uint32 * 1)()
    emul.emul_aggregate(cb(b"".join(keys)), 1, None, C.c_uint64(1), len(keys), out, st)
    assert bytes(out) == orc.aggregate_pks(keys)[1]


def _emul_verify(emul, b, mode=0, offsets=None):
    res = ob(b.n); st = (C.c_uint32 * max(1, b.n))()
    off = (C.c_uint32 * len(offsets))(*offsets) if offsets else None
    emul.emul_verify_batch(cb(b.sigs), cb(b.msgs), 32, cb(b.pks), b.fmt, off, C.c_uint64(b.n), b.k, mode, res, st)
    return [bool(x) for x in bytes(res)[:b.n]], list(st)


def test_pipeline_matches_oracle_and_construction(emul):
    for fmt in (0, 1):
        b = helpers.make_batch(16, 4, fmt=fmt, seed=10 + fmt)
        got, st = _emul_verify(emul, b)
        assert got == b.expect == orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, fmt, nthreads=8)
        flag = {"sig_not_in_g2": 0x02, "apk_infinity": 0x08, "bad_sig_bytes": 0x01, "bad_pk_bytes": 0x04, "flip_msg": 0x40, "wrong_key": 0x40}
        for kind, s in zip(b.kinds, st):
            if kind in flag:
                assert s & flag[kind], (kind, s)


def test_pipeline_golden_batch(emul, vectors):
    fb = vectors["model"]["fast_aggregate_verify_batch"]
    items = fb["items"]
    b = helpers.Batch(); b.n, b.k, b.fmt = len(items), fb["k"], 0
    b.sigs = b"".join(bytes.fromhex(i["sig"]) for i in items); b.msgs = b"".join(bytes.fromhex(i["msg"]) for i in items)
    b.pks = b"".join(bytes.fromhex(h) for i in items for h in i["pks_compressed"])
    assert _emul_verify(emul, b)[0] == [i["result"] for i in items]


def test_pipeline_ragged_and_empty_sets(emul):
    b = helpers.make_batch(6, 3, fmt=0, seed=21, negatives=False)
    # ragged: item 2 loses all its keys (empty set -> false, reference src/aggregates.rs:179-181), item 4 loses one key (-> false)
    counts = [3, 3, 0, 3, 2, 3]
    pks = b"".join(b.pks[48 * 3 * i:48 * 3 * i + 48 * c] for i, c in enumerate(counts))
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    rb = helpers.Batch(); rb.n, rb.k, rb.fmt, rb.sigs, rb.msgs, rb.pks = 6, 0, 0, b.sigs, b.msgs, pks
    got, st = _emul_verify(emul, rb, offsets=offs)
    assert got == [True, True, False, True, False, True] and st[2] & 0x10


def test_verify_mode_single_key(emul):
    b = helpers.make_batch(8, 1, fmt=0, seed=30)
    got, _ = _emul_verify(emul, b, mode=1)
    assert got == orc.batch_verify(b.sigs, b.msgs, b.pks, b.n, nthreads=4)


def test_hash_to_field_register_only_path():
    """hash_fields_to_ws (digests and message blocks by value, what the hashing kernel runs before its generated routine) against the
    array-based expand_message_xmd_256 + fp_from_two_digests, for several message lengths incl. 0 and the multi-block 200"""
    import ctypes as C, random
    emu = helpers.load_emulator()
    rnd = random.Random(17)
    for mlen in (0, 1, 32, 55, 56, 64, 200):
        n = 3
        msgs = bytes(rnd.randrange(256) for _ in range(max(1, mlen * n)))
        out = (C.c_uint32 * (48 * n))(); ref = (C.c_uint32 * (48 * n))()
        emu.emul_hash_fields(msgs, C.c_uint32(mlen), C.c_uint64(n), out, ref)
        assert list(out) == list(ref), mlen
        assert any(out)
