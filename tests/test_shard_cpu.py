"""World-size-2 gloo test of the multi-GPU path's host logic: contiguous shards, per-rank bitmap, one all-gather.
The per-shard verifier here is the oracle (no GPU in this test); the GPU run replaces it by the HIP pipeline."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, k, q):
    import sys
    sys.path.insert(0, os.path.join(helpers.ROOT, "oracle")); sys.path.insert(0, helpers.ROOT)
    import orc
    from milagro_bls_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = helpers.make_batch(n, k, fmt=0, seed=77, nthreads=2)
    lo, hi = shard.shard_range(n, rank, world)
    m = hi - lo
    res = orc.batch_fast_aggregate_verify(b.sigs[96 * lo:96 * hi], b.msgs[32 * lo:32 * hi], b.pks[48 * k * lo:48 * k * hi], m, k, 0, nthreads=2)
    words = shard.pack_bits(torch.tensor(res, dtype=torch.uint8))
    full = shard.all_gather_bitmap(words, world)
    got = torch.cat([shard.unpack_bits(full[r * words.numel():(r + 1) * words.numel()], m) for r in range(world)])
    if rank == 0:
        q.put((got.tolist(), b.expect))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_sharded_verify_and_bitmap_gather():
    n, k, world = 128, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, expect = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [bool(x) for x in got] == expect


def test_pack_unpack_layout():
    from milagro_bls_amd import shard
    r = torch.zeros(130, dtype=torch.uint8); r[0] = 1; r[63] = 1; r[64] = 1; r[129] = 1
    w = shard.pack_bits(r)
    assert w.numel() == 3 and (int(w[0]) & 1) == 1 and int(w[0]) < 0 and int(w[1]) == 1 and int(w[2]) == 2
    assert torch.equal(shard.unpack_bits(w, 130), r)
    assert shard.shard_range(10, 0, 3) == (0, 3) and shard.shard_range(10, 2, 3) == (6, 10)
