"""The wave-cooperative engine on the GPU (mbls_coop.h): the one-wave-per-item pairing check against the one-lane-per-item kernels and the
oracle on the same batches (both settings of mbls_ctx_set_coop_max_items), and the cooperative levels of the n-pairing paths' trees."""
import ctypes as C
import random

import pytest

import helpers
import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mb():
    from milagro_bls_amd import batch, _native
    _native.default_context()
    return batch


@pytest.fixture(scope="module")
def N():
    from milagro_bls_amd import _native
    return _native


@pytest.mark.parametrize("fmt", [0, 1])
def test_both_pairing_paths_agree_with_the_oracle(mb, N, fmt):
    """every rejection class of helpers.make_batch (message bit, wrong key, signature outside G2, infinity signature, apk = infinity,
    undecodable signature / key) through the cooperative pairing check and through k_miller / k_final: same results, same status words"""
    ctx = N.default_context()
    b = helpers.make_batch(96, 5, fmt=fmt, seed=41 + fmt)
    want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, fmt, nthreads=8)
    assert want == b.expect
    outs = []
    try:
        for lim in (0, 1 << 20):
            ctx.set_coop_max_items(lim)
            outs.append(mb.fast_aggregate_verify_batch(b.sigs, b.msgs, b.pks, b.n, b.k, pk_format=fmt))
    finally:
        ctx.reset_tuning()
    assert outs[0][0] == outs[1][0] == want
    assert outs[0][1] == outs[1][1]


def test_cooperative_verify_with_infinite_members(mb, N):
    """Signature::verify has no infinity checks (reference src/signature.rs:27-40): pairs with an infinite member contribute 1 -- the
    cooperative program masks the G1 argument of such a pair once (tools/gen_coop.py masked_p)"""
    ctx = N.default_context()
    rnd = random.Random(3)
    sks = [rnd.randrange(1, helpers.R) for _ in range(2)]
    pks = [orc.sk_to_pk(s) for s in sks]
    inf_pk = bytes([0x40]) + bytes(95)
    msgs = [rnd.randbytes(32) for _ in range(5)]
    sig = lambda i, m: orc.g2_compress(orc.sign(m, sks[i]))
    items = [(sig(0, msgs[0]), msgs[0], pks[0]), (sig(0, msgs[1]), msgs[1], inf_pk), (helpers.G2_INF, msgs[2], inf_pk), (helpers.G2_INF, msgs[3], pks[1]),
             (sig(1, msgs[4]), msgs[4], pks[0])]
    want = [orc.verify(orc.g2_from_compressed(s)[1], m, p) for s, m, p in items]
    assert want == [True, False, True, False, False]
    try:
        for lim in (0, 1 << 20):
            ctx.set_coop_max_items(lim)
            got, _ = mb.verify_batch(b"".join(i[0] for i in items), b"".join(i[1] for i in items), b"".join(i[2] for i in items), len(items), pk_format=1)
            assert got == want, lim
    finally:
        ctx.reset_tuning()


def test_tree_levels_on_both_engines(N):
    """verify_multiple over 4 500 one-key sets: the first level of each tree has 2 250 pairs (one lane per product, k_f12_tree_d / k_g2_tree_d),
    every level below runs one wave per product (programs f12mul / g2add). All valid -> true; any single corrupted set -> false; the
    first 40 sets also against the oracle with the same scalars."""
    import torch
    import bench
    from milagro_bls_amd import batch
    ctx = N.default_context()
    dev = torch.device("cuda:0")
    n = 4500
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, 1, N.PK_UNCOMPRESSED, rank=9, negatives=False)
    g = torch.Generator(device="cpu"); g.manual_seed(11)
    rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
    args = (d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, 1)
    assert batch.verify_multiple_sets_device(*args, pk_format=N.PK_UNCOMPRESSED) is True
    for i in (0, 2249, 2250, 4499, 1234):
        d_msgs[i, 7] ^= 1
        assert batch.verify_multiple_sets_device(*args, pk_format=N.PK_UNCOMPRESSED) is False, i
        d_msgs[i, 7] ^= 1
    assert batch.verify_multiple_sets_device(*args, pk_format=N.PK_UNCOMPRESSED) is True
    m = 40
    sets = [(orc.g2_from_compressed(d_sigs[i].cpu().numpy().tobytes())[1], d_pks[i, 0].cpu().numpy().tobytes(), d_msgs[i].cpu().numpy().tobytes()) for i in range(m)]
    rr = [int(x) for x in rands[:m].cpu().tolist()]
    assert orc.verify_multiple(sets, rr) is True
    assert batch.verify_multiple_sets_device(d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), m, 1, pk_format=N.PK_UNCOMPRESSED) is True
    # the same scalars on shifted sets: a genuine mismatch between scalars and sets must fail in both
    rr2 = rr[1:] + rr[:1]
    assert orc.verify_multiple(sets, rr2) is True                    # any nonzero scalars verify valid sets
    sets_bad = [(sets[1][0], sets[0][1], sets[0][2])] + sets[1:]
    assert orc.verify_multiple(sets_bad, rr) is False


def test_message_phase_of_the_pipeline_against_golden_and_oracle(mb, vectors):
    """H(m) as the verification pipeline's own message phase leaves it -- the generated one-lane routine (mode 1), the cooperative
    program hashg2 (mode 2) and the two-lanes-per-message form of batches below half a round (mode 3: k_hash2) -- against the committed
    golden vectors (incl. the 133 700-byte message) and the oracle on 200 / 333 random messages"""
    rnd = random.Random(17)
    for v in vectors["model"]["hash_to_g2"]:
        m = helpers.expand_msg(v["msg"])
        for mode in (1, 2, 3):
            assert mb.hash_to_g2_batch(m, 1, msg_len=len(m), mode=mode).hex() == v["compressed"], (mode, v["msg"][:16])
    msgs = rnd.randbytes(32 * 200)
    want = orc.batch_hash_to_g2(msgs, 200)
    assert mb.hash_to_g2_batch(msgs, 200, mode=1) == want
    assert mb.hash_to_g2_batch(msgs, 200, mode=2) == want
    assert mb.hash_to_g2_batch(msgs, 200, mode=3) == want
    msgs = rnd.randbytes(32 * 333)                      # an odd count: the last workgroup holds 13 messages on 26 lanes
    assert mb.hash_to_g2_batch(msgs, 333, mode=3) == orc.batch_hash_to_g2(msgs, 333)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 130])
def test_packed_programs_agree_with_the_oracle(mb, N, n):
    """Two items per wave in the pairing check (pairing2x2) and four in the message phase (hashg2x4), forced for batch sizes that leave the
    last wave partly empty (n odd, n mod 4 != 0): the verdicts and status words of the one-item-per-wave programs and of the oracle --
    every rejection class of helpers.make_batch --, and H(m) itself through the message-phase probe."""
    ctx = N.default_context()
    b = helpers.make_batch(n, 3, fmt=1, seed=400 + n)
    want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, 1, nthreads=8)
    assert want == b.expect
    outs = {}
    try:
        for name, pack in (("packed", (0, 1 << 62, 0)), ("plain", (1 << 62, 1 << 62, 1 << 62)), ("pairing packed", (0, 1 << 62, 1 << 62)), ("hash packed", (1 << 62, 1 << 62, 0))):
            ctx.set_coop_packing(*pack)
            outs[name] = mb.fast_aggregate_verify_batch(b.sigs, b.msgs, b.pks, b.n, b.k, pk_format=1)
            assert mb.hash_to_g2_batch(b.msgs, n, mode=2) == orc.batch_hash_to_g2(b.msgs, n), name
    finally:
        ctx.reset_tuning()
    for name, (got, st) in outs.items():
        assert got == want and st == outs["plain"][1], name
