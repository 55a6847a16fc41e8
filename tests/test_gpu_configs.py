"""BASELINE.json configs at (or near) their stated sizes on the GPU, through size-independent properties: inputs are
signed on the device by the product's own kernels (bench.build_inputs), so every accepted item is a
sign -> aggregate -> verify round trip, every corrupted item must be rejected, and a subsample is pinned to the oracle."""
import ctypes as C

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import bench
    from milagro_bls_amd import _native as N
    ctx = N.default_context()
    return torch, bench, N, ctx


def test_config3_fast_aggregate_verify_128_keys_both_formats(env):
    # configs[2] AS NAMED: 2^16 items x 128 public keys in the 96-byte form the headline is measured on (26 ms of GPU time; building and signing the batch
    # on the device takes longer); the 48-byte wire form at 2^13 items (its decompression is 8 x the work per item)
    torch, bench, N, ctx = env
    import orc
    dev = torch.device("cuda:0")
    for fmt in (N.PK_UNCOMPRESSED, N.PK_COMPRESSED):
        n, k = (1 << 16) if fmt == N.PK_UNCOMPRESSED else (1 << 13), 128
        d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, fmt, rank=3)
        d_res = torch.zeros(n, dtype=torch.uint8, device=dev); d_bm = torch.zeros(n // 64, dtype=torch.int64, device=dev)
        d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        ctx.check(N.lib().mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), fmt, None,
                                                                  n, k, d_res.data_ptr(), d_bm.data_ptr(), d_st.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(d_res.cpu(), expect)
        st = d_st.cpu().numpy()
        bad = np.arange(7, n, 16)
        want_flag = [0x40, 0x40, 0x02, 0x40, 0x08]     # msg bit, wrong key, sig not in G2, infinity sig (pairing fails), apk = infinity
        for j, i in enumerate(bad[:40]):
            assert st[i] & want_flag[j % 5], (i, j % 5, st[i])
        m = 48
        pkb = 96 if fmt == N.PK_UNCOMPRESSED else 48
        got = orc.batch_fast_aggregate_verify(d_sigs[:m].cpu().numpy().tobytes(), d_msgs[:m].cpu().numpy().tobytes(),
                                              d_pks[:m].cpu().numpy().tobytes(), m, k, fmt, nthreads=8)
        assert got == [bool(x) for x in expect[:m].tolist()]


def test_config4_verify_multiple_2_14_sets_128_keys(env):
    # configs[3]: verify_multiple_aggregate_signatures, 2^14 sets x 128 keys: all valid -> true, one corrupted set -> false
    torch, bench, N, ctx = env
    from milagro_bls_amd import batch
    dev = torch.device("cuda:0")
    n, k = 1 << 14, 128
    d_sigs, d_msgs, d_pks, expect = bench.build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank=4, negatives=False)
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    rands = torch.randint(1, (1 << 62), (n,), dtype=torch.int64, generator=g).to(dev)
    args = (d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), n, k)
    assert batch.verify_multiple_sets_device(*args, pk_format=N.PK_UNCOMPRESSED) is True
    d_msgs[n // 3, 5] ^= 0x10
    assert batch.verify_multiple_sets_device(*args, pk_format=N.PK_UNCOMPRESSED) is False
    d_msgs[n // 3, 5] ^= 0x10
    assert batch.verify_multiple_sets_device(*args, pk_format=N.PK_UNCOMPRESSED) is True
    # a signature outside G2 anywhere in the batch -> false (reference src/aggregates.rs:274-276)
    probe = bytes.fromhex(helpers.load_vectors()["model"]["g2_subgroup_probes"][1]["compressed"])
    keep = d_sigs[n - 1].clone()
    d_sigs[n - 1] = torch.frombuffer(bytearray(probe), dtype=torch.uint8).to(dev)
    assert batch.verify_multiple_sets_device(*args, pk_format=N.PK_UNCOMPRESSED) is False
    d_sigs[n - 1] = keep
    # small case against the oracle with the same blinding scalars
    import orc
    m = 6
    sets = []
    apks, _ = batch.aggregate_public_keys_batch(d_pks[:m].cpu().numpy().tobytes(), m, k, pk_format=N.PK_UNCOMPRESSED)
    for i in range(m):
        sets.append((orc.g2_from_compressed(d_sigs[i].cpu().numpy().tobytes())[1], apks[96 * i:96 * i + 96], d_msgs[i].cpu().numpy().tobytes()))
    rr = [int(x) for x in rands[:m].cpu().tolist()]
    assert orc.verify_multiple(sets, rr) is True
    assert batch.verify_multiple_sets_device(d_sigs.data_ptr(), d_pks.data_ptr(), d_msgs.data_ptr(), rands.data_ptr(), m, k, pk_format=N.PK_UNCOMPRESSED) is True


def test_bench_two_ranks_sharing_the_gpu_real_verifier():
    """bench.py's multi-rank path with the REAL verifier: two rank processes on this box's one GPU (MBLS_BENCH_SHARE_GPU=1: both on device 0, gathers through
    gloo because RCCL wants one rank per device) -- per-rank inputs, the bitmap gather inside the step, the gathered-bitmap check, MAX / MIN reductions, the
    launcher-free spawn -- and stdout carries exactly ONE line (libraries that print to the C stdout, like RCCL's banner, are kept off it)."""
    import json
    import os
    import subprocess
    import sys
    env = dict(os.environ, MBLS_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--items", "4096",
                        "--no-cpu-baseline", "--no-variants"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["bitmap_matches_expectation"] is True and len(d["ms_per_step_per_rank"]) == 2 and "share_gpu_test" in d
    # what makes a SCALE line check itself: the process group's own rank count against n_gpus and the time of the one collective, timed alone
    c = d["collective"]
    assert c["group_world_size"] == 2 and c["group_rank_count_matches_n_gpus"] is True and c["all_gather_ms_per_step"] > 0 and c["backend"] == "gloo"
    # one rank, with the in-process multi-device leg (it makes an RCCL communicator): still one line on stdout
    p = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--items", "4096", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    d1 = json.loads(lines[0])
    assert len(lines) == 1 and d1["n_gpus"] == 1 and d1["collective"] is None
    assert d1["multi_legs_ok"] is True and all(l["ran"] and l["results_match"] for l in d1["multi_handle_leg"])
    k = d1["valu_issue"]["kernels"]["k_miller"]
    assert 0 < k["mac_frac"] < 1.0 and 0.5 < k["mac_share_of_valu"] < 0.8 and k["mac_peak"] > 0          # (4 096 items: not the kernels' operating point; the keys are what is checked)


def test_bench_multi_device_leg_in_a_child_process():
    """the in-process multi-device legs over more than one device run in a child process with a time limit (bench.multi_leg_in_child): a hang or a crash
    there costs that leg, not the line. On this box: the child with two contexts on device 0 (host join), and what a child that cannot answer leaves behind."""
    import os
    import bench
    os.environ["MBLS_MULTI_LEG_DEVICES"] = "0,0"
    try:
        leg = bench.multi_leg_in_child(2, 2048, 4)
    finally:
        del os.environ["MBLS_MULTI_LEG_DEVICES"]
    assert leg.get("results_match") is True and leg["devices"] == [0, 0] and leg["items"] == 4096, leg
    assert leg["bitmap_gather"]["gathered_bitmap_matches"] is True
    gone = bench.multi_leg_in_child(2, 2048, 4, timeout_s=0.01)
    assert "error" in gone and "results_match" not in gone
