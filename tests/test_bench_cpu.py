"""CPU test of bench.py's multi-rank plumbing (world > 1) under gloo with a stub verifier: the timed region (warm-up, barrier +
sync fences, K steps), the per-step all-gather of the bitmap, the MAX-reduce of the elapsed time and the MIN-reduce of the
correctness flag -- so that the first real multi-GPU run cannot fail on anything but the GPU work itself."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, fail_rank):
    import sys
    sys.path.insert(0, helpers.ROOT)
    import bench
    from milagro_bls_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000                                                       # not a multiple of 64: the last bitmap word is partial
    words = shard.bitmap_words(n)
    expect = torch.ones(n, dtype=torch.uint8); expect[7::16] = 0
    res = expect.clone()
    if rank == fail_rank:
        res[123] ^= 1                                              # this rank's verifier disagrees with the expectation
    bm = shard.pack_bits(res)
    d_all = torch.zeros(words * world, dtype=torch.int64)
    calls = []

    def step():                                                    # stub of the per-step work: "verify", then the gather
        time.sleep(0.01 * (rank + 1))                              # ranks take different times: the MAX must come back
        calls.append(1)
        shard.all_gather_bitmap(bm, world, out=d_all)
    elapsed = bench.timed_steps(step, steps=3, warmup=2, world=world, sync=lambda: None)
    ok = bench.check_bitmap(res, bm, expect) and bench.check_gathered(d_all, world, words, bm, rank)
    ok_all = bench.reduce_all_ok(ok, world)
    q.put((rank, len(calls), elapsed, ok, ok_all, [int(x) for x in d_all[:2]], [int(x) for x in d_all[words:words + 2]]))
    dist.barrier(); dist.destroy_process_group()


def _run(fail_rank):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_timed_region_and_reductions_two_ranks():
    r0, r1 = _run(fail_rank=-1)
    assert r0[1] == r1[1] == 5                                     # 2 warm-up + 3 timed steps on every rank
    assert r0[2] == r1[2] and r0[2] >= 3 * 0.02 * 0.9              # MAX over ranks: the slower rank's time, identical everywhere
    assert r0[3] and r1[3] and r0[4] and r1[4]
    assert r0[5] == r1[5] and r0[6] == r1[6]                       # every rank holds the whole gathered bitmap


def test_one_bad_rank_fails_the_whole_job():
    r0, r1 = _run(fail_rank=1)
    assert r0[3] is True and r1[3] is False
    assert r0[4] is False and r1[4] is False                       # MIN-reduce: rank 0 prints ok = false, every rank exits 3


def test_single_rank_paths_need_no_process_group():
    import sys
    sys.path.insert(0, helpers.ROOT)
    import bench
    n = []
    assert bench.timed_steps(lambda: n.append(1), 4, 1, 1, lambda: None) >= 0 and len(n) == 5
    assert bench.reduce_max(1.5, 1) == 1.5 and bench.reduce_all_ok(True, 1) is True


def _bench(*argv, env=None):
    import json, subprocess, sys
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py")] + list(argv), env=e, capture_output=True, text=True, timeout=600)
    lines = []
    for l in r.stdout.splitlines():
        try:
            lines.append(json.loads(l))
        except ValueError:
            pass
    return r.returncode, lines, r.stderr


def test_gpus_flag_without_a_launcher_spawns_the_ranks():
    """`python bench.py --gpus 2` with no torchrun around it: the parent starts two rank processes itself (before touching any GPU) and relays
    rank 0's line -- n_gpus = 2, one ms_per_step per rank (the stub's ranks sleep 5 and 10 ms a step, the job time is the slower one's)."""
    rc, lines, err = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--stub")
    assert rc == 0, err
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["stub"] is True and d["bitmap_matches_expectation"] is True
    assert len(d["ms_per_step_per_rank"]) == 2 and d["ms_per_step_per_rank"][1] >= d["ms_per_step_per_rank"][0] * 0.9
    assert d["ms_per_step"] >= 10.0 * 0.9 and d["ms_per_step"] >= max(d["ms_per_step_per_rank"]) * 0.95


def test_a_failing_rank_or_a_wrong_world_size_is_an_error_exit_not_a_one_gpu_line():
    rc, lines, err = _bench("--gpus", "2", "--steps", "2", "--warmup", "0", "--stub", env={"MBLS_STUB_FAIL_RANK": "1"})
    assert rc == 3 and (not lines or lines[0]["bitmap_matches_expectation"] is False)
    # a launcher that started a different number of ranks than --gpus says
    rc, lines, err = _bench("--gpus", "4", "--stub", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 5 and not lines and "WORLD_SIZE" in err
    # more GPUs asked for than the machine has (no GPU here): refused before anything is started
    import torch
    if torch.cuda.device_count() < 2:
        rc, lines, err = _bench("--gpus", "2")
        assert rc == 4 and not lines and "visible" in err


def test_eight_ranks_print_what_a_scale_record_needs():
    """`bench.py --gpus 8 --stub`: the line a SCALE record is made from at N = 8 -- n_gpus, one ms_per_step per rank, the gathered bitmap checked on every rank, the
    process group's own rank count and the time of the one collective, and configs[4]'s shard leg (timed at 8 ranks only)."""
    rc, lines, err = _bench("--gpus", "8", "--steps", "2", "--warmup", "1", "--stub")
    assert rc == 0, err
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 8 and d["stub"] is True and d["scaling"] == "weak" and d["bitmap_matches_expectation"] is True
    assert len(d["ms_per_step_per_rank"]) == 8 and all(x > 0 for x in d["ms_per_step_per_rank"])
    assert d["ms_per_step"] >= max(d["ms_per_step_per_rank"]) * 0.95           # the job's time is the slowest rank's
    c = d["collective"]
    assert c["backend"] == "gloo" and c["group_world_size"] == 8 and c["group_rank_count_matches_n_gpus"] is True and c["all_gather_ms_per_step"] > 0
    leg = d["configs4_shard_leg"]
    assert leg["bitmap_matches_expectation"] is True and leg["steps"] == 2 and leg["value"] > 0


def test_a_rank_that_dies_ends_the_job_with_an_error_and_no_line():
    """one of eight ranks exits in the middle of the timed region (fresh child processes only; nothing is re-executed): the others would wait in the gather for
    ever -- the parent ends them after its grace period, prints no result line and leaves with a non-zero code"""
    import time as _t
    t0 = _t.time()
    rc, lines, err = _bench("--gpus", "8", "--steps", "3", "--warmup", "0", "--stub", env={"MBLS_STUB_DIE_RANK": "5"})
    assert rc != 0 and not lines
    assert _t.time() - t0 < 240
