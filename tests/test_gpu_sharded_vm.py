"""verify_multiple_aggregate_signatures cut into shards ("one exchange step"; reference src/aggregates.rs:261-316 is one
loop over one iterator -- its product of pairings and its sum of blinded signatures are associative): mbls_verify_multiple_partial_device per
shard, the records joined by mbls_verify_multiple_finish_device, and the same through the multi-device handle with device_ids = {0, 0} /
{0, 0, 0} -- against the one-device call and the oracle with the same scalars."""
import ctypes as C
import random

import pytest

import helpers
import orc

pytestmark = pytest.mark.gpu

G1_INF_U = bytes([0x40]) + bytes(95)


@pytest.fixture(scope="module")
def N():
    from milagro_bls_amd import _native
    _native.default_context()
    return _native


def _sets(rnd, n):
    sks = [rnd.randrange(1, helpers.R) for _ in range(n)]
    pk96 = orc.batch_sk_to_pk(b"".join(s.to_bytes(32, "big") for s in sks), n, 1, nthreads=8)
    pks = [pk96[96 * i:96 * i + 96] for i in range(n)]            # 96-byte decoded keys
    msgs = [rnd.randbytes(32) for _ in range(n)]
    sigs = [orc.g2_compress(orc.sign(m, s)) for m, s in zip(msgs, sks)]
    rands = [rnd.randrange(1, 1 << 64) for _ in range(n)]
    return sks, pks, msgs, sigs, rands


def _one_device(N, sigs, apks, msgs, rands):
    n = len(sigs)
    rr = (C.c_uint64 * max(1, n))(*rands)
    return bool(N.lib().mbls_verify_multiple_aggregate_signatures(N.default_context().handle, N.cbuf(b"".join(sigs)), N.cbuf(b"".join(apks)),
                                                                 N.cbuf(b"".join(msgs)), 32, None, rr, n))


def _sharded(N, sigs, apks, msgs, rands, cuts, spoil=None):
    """the shards [cuts[j], cuts[j+1]) through the device entries on one context: every record lands in one device buffer, then the join"""
    import torch
    from milagro_bls_amd import batch
    dev = torch.device("cuda:0")
    t = lambda b: torch.frombuffer(bytearray(b if b else b"\0"), dtype=torch.uint8).to(dev)
    G = len(cuts) - 1
    recs = torch.zeros(max(1, G) * N.VM_PARTIAL_BYTES, dtype=torch.uint8, device=dev)
    keep = []
    for g in range(G):
        lo, hi = cuts[g], cuts[g + 1]
        d_s, d_a, d_m = t(b"".join(sigs[lo:hi])), t(b"".join(apks[lo:hi])), t(b"".join(msgs[lo:hi]))
        d_r = torch.tensor([x - (1 << 64) if x >= (1 << 63) else x for x in rands[lo:hi]] or [0], dtype=torch.int64, device=dev)
        keep += [d_s, d_a, d_m, d_r]
        batch.verify_multiple_partial_device(d_s.data_ptr(), d_m.data_ptr(), d_r.data_ptr(), hi - lo, recs.data_ptr() + g * N.VM_PARTIAL_BYTES, d_apks=d_a.data_ptr())
    if spoil is not None:
        torch.cuda.synchronize()
        spoil(recs)
    return batch.verify_multiple_finish_device(recs.data_ptr(), G)


def _oracle(sigs, apks, msgs, rands):
    dec = [orc.g2_from_compressed(s) for s in sigs]
    if any(e for e, _ in dec):
        return None
    return orc.verify_multiple([(d[1], a, m) for d, a, m in zip(dec, apks, msgs)], rands)


@pytest.mark.usefixtures("engine")
def test_sharded_verify_multiple_edge_members_vs_oracle(N, vectors):
    from milagro_bls_amd import batch
    rnd = random.Random(77)
    sks, pks, msgs, sigs, rands = _sets(rnd, 9)
    probe = bytes.fromhex(vectors["model"]["g2_subgroup_probes"][0]["compressed"])
    D = orc.sign(b"d" * 32, 4242)
    f0 = orc.g2_compress(orc.g2_add(orc.g2_from_compressed(sigs[0])[1], D))
    f8 = orc.g2_compress(orc.g2_add(orc.g2_from_compressed(sigs[8])[1], orc.g2_mul(D, helpers.R - 1)))
    variants = {
        "all valid": (lambda s, a, m: None, True),
        "a set loses its signature": (lambda s, a, m: s.__setitem__(2, helpers.G2_INF), False),
        "a set loses its key": (lambda s, a, m: a.__setitem__(3, G1_INF_U), False),
        "a set infinite on both sides": (lambda s, a, m: (s.__setitem__(4, helpers.G2_INF), a.__setitem__(4, G1_INF_U)), True),
        "wrong key": (lambda s, a, m: a.__setitem__(1, pks[0]), False),
        "signature outside G2": (lambda s, a, m: s.__setitem__(6, probe), False),                   # src/aggregates.rs:274-276
        "wrong message in the last shard": (lambda s, a, m: m.__setitem__(8, bytes(32)), False),
        # (sig_0 + D, sig_8 - D): the two halves of the forgery sit in different shards; only the blinding across shards catches it
        "forged pair across shards": (lambda s, a, m: (s.__setitem__(0, f0), s.__setitem__(8, f8)), False),
    }
    m2 = N.MultiContext([0, 0]); m3 = N.MultiContext([0, 0, 0])
    try:
        for name, (f, want) in variants.items():
            s, a, m = list(sigs), list(pks), list(msgs)
            f(s, a, m)
            assert _oracle(s, a, m, rands) is want, name
            assert _one_device(N, s, a, m, rands) is want, name
            for cuts in ([0, 9], [0, 4, 9], [0, 1, 1, 8, 9], [0, 0, 9, 9]):                          # one shard, two, an empty one inside, empty ones at the ends
                assert _sharded(N, s, a, m, rands, cuts) is want, (name, cuts)
            for mc in (m2, m3):
                assert batch.multi_verify_multiple_aggregate_signatures(mc, b"".join(s), b"".join(a), b"".join(m), rands, 9) is want, name
        # undecodable members: the ABI answers false wherever they sit
        bad_sig = bytes([sigs[5][0] & 0x7F]) + sigs[5][1:]
        assert _sharded(N, sigs[:5] + [bad_sig] + sigs[6:], pks, msgs, rands, [0, 4, 9]) is False
        assert batch.multi_verify_multiple_aggregate_signatures(m2, b"".join(sigs[:5] + [bad_sig] + sigs[6:]), b"".join(pks), b"".join(msgs), rands, 9) is False
        # a zero scalar in one shard fails the whole check (src/aggregates.rs:280-287 draws until nonzero)
        assert _sharded(N, sigs, pks, msgs, rands[:7] + [0] + rands[8:], [0, 4, 9]) is False
        # ragged messages through the handle: offsets are absolute, every shard stages its own slice
        lens = [0, 5, 300, 32, 1, 64, 33, 7, 129]
        rm = [rnd.randbytes(l) for l in lens]
        rs = [orc.g2_compress(orc.sign(x, k)) for x, k in zip(rm, sks)]
        off = [0]
        for x in rm:
            off.append(off[-1] + len(x))
        for mc in (m2, m3):
            assert batch.multi_verify_multiple_aggregate_signatures(mc, b"".join(rs), b"".join(pks), b"".join(rm), rands, 9, msg_len=0, msg_offsets=off) is True
            assert batch.multi_verify_multiple_aggregate_signatures(mc, b"".join(rs), b"".join(pks), b"".join(rm[:-1]) + bytes(129), rands, 9, msg_len=0, msg_offsets=off) is False
        # no sets at all / no records at all: the empty iterator
        assert batch.multi_verify_multiple_aggregate_signatures(m2, b"", b"", b"", [], 0) is True
        assert batch.verify_multiple_finish_device(0, 0) is True
    finally:
        m2.close(); m3.close()


def test_a_record_that_is_not_one_fails_the_check(N):
    rnd = random.Random(78)
    sks, pks, msgs, sigs, rands = _sets(rnd, 4)
    assert _sharded(N, sigs, pks, msgs, rands, [0, 2, 4]) is True

    def zero_second(recs):
        recs[N.VM_PARTIAL_BYTES:] = 0                             # what the buffer holds before k_vm_export has spoken
    assert _sharded(N, sigs, pks, msgs, rands, [0, 2, 4], spoil=zero_second) is False

    def swap_sums(recs):                                          # shard 1's signature sum replaced by shard 0's: the product no longer matches
        a = recs[576:576 + 288].clone()
        recs[N.VM_PARTIAL_BYTES + 576:N.VM_PARTIAL_BYTES + 576 + 288] = a
    assert _sharded(N, sigs, pks, msgs, rands, [0, 2, 4], spoil=swap_sums) is False


def test_sharded_verify_multiple_4500_sets_from_their_keys(N):
    """three uneven shards of 4 500 sets x 2 wire-format keys (the shards' key sums run on the device first, src/aggregates.rs:29-39): the first
    tree levels of every shard run one lane per product, the join of the three records one wave per product"""
    import torch
    import bench
    from milagro_bls_amd import batch
    ctx = N.default_context()
    dev = torch.device("cuda:0")
    n, k = 4500, 2
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=5, negatives=False)
    g = torch.Generator(device="cpu"); g.manual_seed(12)
    rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
    cuts = [0, 100, 2900, 4500]
    recs = torch.zeros(3 * N.VM_PARTIAL_BYTES, dtype=torch.uint8, device=dev)

    def run():
        for j in range(3):
            lo, hi = cuts[j], cuts[j + 1]
            batch.verify_multiple_partial_device(d_sigs[lo:].data_ptr(), d_msgs[lo:].data_ptr(), rands[lo:].data_ptr(), hi - lo, recs.data_ptr() + j * N.VM_PARTIAL_BYTES,
                                                 d_pks=d_pks[lo:].data_ptr(), k=k, pk_format=N.PK_UNCOMPRESSED)
        return batch.verify_multiple_finish_device(recs.data_ptr(), 3)
    assert run() is True
    assert batch.verify_multiple_sets_device(d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, k, pk_format=N.PK_UNCOMPRESSED) is True
    for i in (0, 99, 100, 2899, 4499):
        d_msgs[i, 3] ^= 4
        assert run() is False, i
        d_msgs[i, 3] ^= 4
    assert run() is True


def test_verify_multiple_above_2_14_sets_side_stream_trees(N):
    """more than 2^14 sets: the chains run one after the other on the caller's stream, and the signatures' sum tree + the Miller loop of (S, -G1)
    run on a second stream beside the product tree (npairing_finish, side_s_chain) -- one call and one shard + join, valid and corrupted"""
    import torch
    import bench
    from milagro_bls_amd import batch
    ctx = N.default_context()
    dev = torch.device("cuda:0")
    n = 20000
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, 1, N.PK_UNCOMPRESSED, rank=6, negatives=False)
    g = torch.Generator(device="cpu"); g.manual_seed(13)
    rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
    rec = torch.zeros(N.VM_PARTIAL_BYTES, dtype=torch.uint8, device=dev)

    def one():
        return batch.verify_multiple_sets_device(d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, 1, pk_format=N.PK_UNCOMPRESSED)

    def shard():
        batch.verify_multiple_partial_device(d_sigs.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, rec.data_ptr(), d_pks=d_pks.data_ptr(), k=1, pk_format=N.PK_UNCOMPRESSED)
        return batch.verify_multiple_finish_device(rec.data_ptr(), 1)
    for f in (one, shard):
        assert f() is True
        for i in (0, 9999, 19999):
            d_sigs_backup = d_msgs[i, 9].item()
            d_msgs[i, 9] ^= 2
            assert f() is False, (f.__name__, i)
            d_msgs[i, 9] = d_sigs_backup
        assert f() is True
    # a signature swapped between two sets: the sum of the signatures is unchanged only without blinding
    a, b = d_sigs[5].clone(), d_sigs[15000].clone()
    d_sigs[5], d_sigs[15000] = b, a
    assert one() is False and shard() is False
    d_sigs[5], d_sigs[15000] = a, b
    assert one() is True


def test_many_records_join_through_both_tree_engines(N):
    """5 000 records -- four real shards among 4 996 empty ones (an empty shard contributes (1, infinity, 0)): the first levels of the join's
    product and sum trees run one lane per product, the rest one wave per product; same bool as the four shards alone, and as one call"""
    import torch
    from milagro_bls_amd import batch
    rnd = random.Random(79)
    sks, pks, msgs, sigs, rands = _sets(rnd, 8)
    dev = torch.device("cuda:0")
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    G = 5000
    P = N.VM_PARTIAL_BYTES
    for spoil in (False, True):
        m = list(msgs)
        if spoil:
            m[5] = bytes(32)
        recs = torch.zeros(G * P, dtype=torch.uint8, device=dev)
        empty = torch.zeros(P, dtype=torch.uint8, device=dev)
        batch.verify_multiple_partial_device(0, 0, 0, 0, empty.data_ptr(), d_apks=0)        # n = 0: no buffers needed
        torch.cuda.synchronize()
        recs.view(G, P)[:] = empty
        where = [0, 1234, 2500, 4999]
        keep = []
        for j, g in enumerate(where):
            lo, hi = 2 * j, 2 * j + 2
            d = [t(b"".join(sigs[lo:hi])), t(b"".join(pks[lo:hi])), t(b"".join(m[lo:hi])),
                 torch.tensor([x - (1 << 64) if x >= (1 << 63) else x for x in rands[lo:hi]], dtype=torch.int64, device=dev)]
            keep += d
            batch.verify_multiple_partial_device(d[0].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 2, recs.data_ptr() + g * P, d_apks=d[1].data_ptr())
        assert batch.verify_multiple_finish_device(recs.data_ptr(), G) is (not spoil)
        assert _one_device(N, sigs, pks, m, rands) is (not spoil)


def test_verify_multiple_over_key_table_indices(N):
    """sets named by indices into a resident key table (mbls_verify_multiple_sets_indexed_device): the same bool as the sets given by their key
    bytes -- one call and two shards + join --, an index outside the table rejects the check with MBLS_ST_BAD_PK_ENCODING"""
    import torch
    import bench
    from milagro_bls_amd import batch
    ctx = N.default_context()
    dev = torch.device("cuda:0")
    n, k = 3000, 3
    d_sigs, d_msgs, d_pks, expect, d_idx, table = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=8, negatives=False, return_indices=True)
    d_idx = d_idx.to(torch.int32).contiguous()
    g = torch.Generator(device="cpu"); g.manual_seed(14)
    rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
    args = (table, d_sigs.data_ptr(), d_idx.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, k)
    by_bytes = lambda: batch.verify_multiple_sets_device(d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, k, pk_format=N.PK_UNCOMPRESSED)
    assert batch.verify_multiple_sets_indexed_device(*args) is True and by_bytes() is True
    d_msgs[77, 1] ^= 1
    assert batch.verify_multiple_sets_indexed_device(*args) is False and by_bytes() is False
    d_msgs[77, 1] ^= 1
    # two shards through the record form
    recs = torch.zeros(2 * N.VM_PARTIAL_BYTES, dtype=torch.uint8, device=dev)
    for j, (lo, hi) in enumerate(((0, 1000), (1000, n))):
        batch.verify_multiple_sets_indexed_device(table, d_sigs[lo:].data_ptr(), d_idx[lo:].data_ptr(), d_msgs[lo:].data_ptr(), rands[lo:].data_ptr(), hi - lo, k,
                                                  d_partial=recs.data_ptr() + j * N.VM_PARTIAL_BYTES)
    assert batch.verify_multiple_finish_device(recs.data_ptr(), 2) is True
    # an index outside the table
    keep = int(d_idx[5, 1].item())
    d_idx[5, 1] = 0x7FFFFFF0
    st = torch.zeros(1, dtype=torch.int32, device=dev); res = torch.full((8,), 7, dtype=torch.uint8, device=dev)
    batch.verify_multiple_sets_indexed_device(*args, d_result=res.data_ptr(), d_status=st.data_ptr())
    torch.cuda.synchronize()
    assert int(res[0].item()) == 0 and (int(st[0].item()) & 0x04)
    d_idx[5, 1] = keep
    assert batch.verify_multiple_sets_indexed_device(*args) is True
