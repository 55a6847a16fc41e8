"""The exchange steps of the multi-device handle (include/mbls.h, mbls_multi_*): the packed accept bitmap of mbls_multi_fast_aggregate_verify_bitmap and the
partial records of mbls_multi_verify_multiple_aggregate_signatures travel between the devices as RCCL all-gathers (librccl.so.1 opened at run time,
one communicator per handle) -- or through host memory where RCCL cannot make a communicator. This box has ONE GPU: a handle over {0} takes the RCCL path
(a communicator of one rank: ncclCommInitAll, ncclGroupStart / ncclAllGather / ncclGroupEnd on device buffers), a handle over {0, 0} must fall back to the
host join (RCCL wants one rank per device) and say so. Both against the single-context entry and the oracle. More than one DEVICE: unmeasured on hardware."""
import random

import pytest

import helpers
import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from milagro_bls_amd import batch, _native
    _native.default_context()
    return batch, _native


def _batch(n, seed):
    b = helpers.make_batch(n, 3, fmt=1, seed=seed, nthreads=8)
    want = orc.batch_fast_aggregate_verify(b.sigs, b.msgs, b.pks, b.n, b.k, 1, nthreads=8)
    assert want == b.expect
    return b, want


@pytest.mark.parametrize("devs,want_rccl", [([0], True), ([0, 0], False), ([0, 0, 0], False)])
def test_accept_bitmap_gathered_between_the_devices(env, devs, want_rccl):
    mb, N = env
    m = N.MultiContext(devs)
    note = m.exchange_note
    assert m.rccl_active is want_rccl, note
    if want_rccl:
        assert "RCCL all-gather" in note
    else:
        assert "host join" in note and "more than once" in note
    for n in (1, 63, 64, 65, 200, 331):                           # shards of whole bitmap words; the last word partly filled; fewer words than devices
        b, want = _batch(n, seed=100 + n)
        bits, st, words = mb.multi_fast_aggregate_verify_bitmap(m, b.sigs, b.msgs, b.pks, n, b.k, pk_format=1)
        assert bits == want, (devs, n)
        one, st_one = mb.fast_aggregate_verify_batch(b.sigs, b.msgs, b.pks, n, b.k, pk_format=1)
        assert one == want and st == st_one
        assert len(words) == (n + 63) // 64 and all(w >> 64 == 0 for w in words)
        if n % 64:
            assert words[-1] >> (n % 64) == 0                      # nothing beyond item n - 1
    # ragged keys and messages through offset tables
    rnd = random.Random(3)
    n = 150
    b, want = _batch(n, seed=9)
    koff = [3 * i for i in range(n + 1)]
    moff = [32 * i for i in range(n + 1)]
    bits, st, _ = mb.multi_fast_aggregate_verify_bitmap(m, b.sigs, b.msgs, b.pks, n, pk_format=1, pk_offsets=koff, msg_offsets=moff)
    assert bits == want
    m.close()


@pytest.mark.parametrize("devs", [[0], [0, 0]])
def test_verify_multiple_records_gathered_between_the_devices(env, devs):
    """src/aggregates.rs:261-316 over the handle: the shards' 896-byte records are all-gathered (RCCL for {0}, host for {0, 0}), the first device joins them"""
    import ctypes as C
    mb, N = env
    rnd = random.Random(12)
    n, k = 9, 2
    sks = [[rnd.randrange(1, helpers.R) for _ in range(k)] for _ in range(n)]
    msgs = [rnd.randbytes(32) for _ in range(n)]
    sets = []
    for i in range(n):
        apk = orc.aggregate_pks([orc.sk_to_pk(s) for s in sks[i]])[1]
        sig = orc.sign(msgs[i], sum(sks[i]) % helpers.R)
        sets.append((sig, apk, msgs[i]))
    rands = [rnd.randrange(1, 1 << 63) for _ in range(n)]
    assert orc.verify_multiple(sets, rands) is True
    m = N.MultiContext(devs)

    def run(ss):
        rr = (C.c_uint64 * n)(*rands)
        return N.lib().mbls_multi_verify_multiple_aggregate_signatures(m.handle, N.cbuf(b"".join(orc.g2_compress(s[0]) for s in ss)), N.cbuf(b"".join(s[1] for s in ss)),
                                                                       N.cbuf(b"".join(s[2] for s in ss)), 32, None, rr, n)
    assert run(sets) == 1
    bad = list(sets); bad[7] = (sets[6][0], sets[7][1], sets[7][2])
    assert orc.verify_multiple(bad, rands) is False and run(bad) == 0
    m.close()
