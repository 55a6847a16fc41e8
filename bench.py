#!/usr/bin/env python3
"""bench.py -- fast_aggregate_verify/s (128 pubkeys, 32-byte message) on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (mbls_fast_aggregate_verify_batch_device: aggregate -> sig -> hash -> Miller ->
final exponentiation -> bitmap) over one batch of 2^16 synthetic (signature, message, 128-pubkey set) items per GPU
(BASELINE.json configs[2]), inputs resident in HBM, followed -- for N > 1 -- by the RCCL all-gather of the accept
bitmap. Weak scaling: every rank verifies its own 2^16 items, no data-path collective.
Inputs are produced on the device by the product's own signing / key kernels (the oracle is used only for the
cpu_baseline leg and a small cross-check of the bitmap). PyTorch is plumbing: device buffers, stream, RCCL.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from milagro_bls_amd import _native as N  # noqa: E402
from milagro_bls_amd import shard  # noqa: E402

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
SEED = 0x6D626C73          # "mbls" 
POOL = 4096
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
# a point of E'(Fp2) outside G2 (tests/golden/vectors.json model.g2_subgroup_probes[0]) for the "sig not in G2" negatives
NOT_IN_G2_HEX = None


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def make_pool(seed):
    st = seed
    sks = []
    for _ in range(POOL):
        v = 0
        for _ in range(4):
            st, w = splitmix64(st)
            v = (v << 64) | w
        sks.append(1 + v % (R - 1))
    return sks


def build_inputs(ctx, dev, n, k, fmt, rank, negatives=True):
    """Synthetic batch on the device. Item i uses the k pool keys {(a_i + j*s_i) mod POOL}, s_i odd (distinct keys);
    sig_i = [sum sk mod r] H(msg_i). Every 16th item (i % 16 == 7) is corrupted, cycling over five rejection classes."""
    lib = N.lib()
    pool = make_pool(SEED)
    pool_b = np.frombuffer(b"".join(s.to_bytes(32, "big") for s in pool), dtype=np.uint8).reshape(POOL, 32)
    d_pool_sk = torch.from_numpy(pool_b.copy()).to(dev)
    pkb = 48 if fmt == N.PK_COMPRESSED else 96
    d_pool_pk = torch.empty((POOL, pkb), dtype=torch.uint8, device=dev)
    ctx.check(lib.mbls_sk_to_pk_batch_device(ctx.handle, d_pool_sk.data_ptr(), fmt, POOL, d_pool_pk.data_ptr(), None))
    rng = np.random.default_rng(SEED + 1000 * rank)
    a = rng.integers(0, POOL, size=n, dtype=np.int64)
    s = rng.integers(0, POOL // 2, size=n, dtype=np.int64) * 2 + 1
    idx = (a[:, None] + np.arange(k, dtype=np.int64)[None, :] * s[:, None]) % POOL          # [n, k]
    msgs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    # aggregate secret keys: limb-wise sums (k * 2^32 fits easily in u64), then one Python mod per item
    limbs = np.array([[(sk >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for sk in pool], dtype=np.uint64)
    agg = np.zeros((n, 8), dtype=np.uint64)
    for c0 in range(0, n, 8192):
        agg[c0:c0 + 8192] = limbs[idx[c0:c0 + 8192]].sum(axis=1)
    agg_sk = np.zeros((n, 32), dtype=np.uint8)
    for i in range(n):
        v = 0
        row = agg[i]
        for j in range(8):
            v += int(row[j]) << (32 * j)
        agg_sk[i] = np.frombuffer((v % R).to_bytes(32, "big"), dtype=np.uint8)
    d_idx = torch.from_numpy(idx).to(dev)
    d_msgs = torch.from_numpy(msgs).to(dev)
    d_agg = torch.from_numpy(agg_sk).to(dev)
    d_sigs = torch.empty((n, 96), dtype=torch.uint8, device=dev)
    ctx.check(lib.mbls_sign_batch_device(ctx.handle, d_agg.data_ptr(), d_msgs.data_ptr(), 32, n, d_sigs.data_ptr(), None))
    d_pks = d_pool_pk[d_idx.reshape(-1)].reshape(n, k, pkb).contiguous()
    expect = torch.ones(n, dtype=torch.uint8)
    if negatives:
        bad = np.arange(7, n, 16)
        kinds = np.arange(len(bad)) % 5
        expect[bad] = 0
        b0 = torch.from_numpy(bad[kinds == 0]).to(dev)                     # flip one message bit
        d_msgs[b0, 0] ^= 1
        b1 = bad[kinds == 1]                                               # replace key 0 by the next pool key (sum changes)
        if len(b1):
            other = (idx[b1, 0] + 1) % POOL
            d_pks[torch.from_numpy(b1).to(dev), 0] = d_pool_pk[torch.from_numpy(other).to(dev)]
        b2 = torch.from_numpy(bad[kinds == 2]).to(dev)                     # signature on the curve but outside G2
        if len(b2):
            with open(os.path.join(ROOT, "tests", "golden", "vectors.json")) as f:
                probe = json.load(f)["model"]["g2_subgroup_probes"][0]["compressed"]
            d_sigs[b2] = torch.frombuffer(bytearray(bytes.fromhex(probe)), dtype=torch.uint8).to(dev)
        b3 = torch.from_numpy(bad[kinds == 3]).to(dev)                     # canonical infinity signature
        if len(b3):
            infs = torch.zeros(96, dtype=torch.uint8); infs[0] = 0xC0
            d_sigs[b3] = infs.to(dev)
        b4 = bad[kinds == 4]                                               # last key := -(sum of the others): apk = infinity
        if len(b4) and k >= 2:
            t4 = torch.from_numpy(b4).to(dev)
            part = d_pks[t4, :k - 1].contiguous()
            apk = np.zeros((len(b4), 96), dtype=np.uint8)
            ctx.check(lib.mbls_aggregate_public_keys_batch(ctx.handle, N.cbuf(part.cpu().numpy().tobytes()), fmt, None, len(b4), k - 1,
                                                           apk.ctypes.data_as(C.c_void_p), None))
            P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
            neg = np.zeros((len(b4), 96), dtype=np.uint8)
            for r in range(len(b4)):
                y = int.from_bytes(apk[r, 48:].tobytes(), "big")
                neg[r] = np.frombuffer(apk[r, :48].tobytes() + ((P - y) % P).to_bytes(48, "big"), dtype=np.uint8)
            if fmt == N.PK_COMPRESSED:
                comp = np.zeros((len(b4), 48), dtype=np.uint8); errs = np.zeros(len(b4), dtype=np.uint8)
                ctx.check(lib.mbls_pk_compress_batch(ctx.handle, neg.ctypes.data_as(C.c_void_p), len(b4), comp.ctypes.data_as(C.c_void_p),
                                                     errs.ctypes.data_as(C.c_void_p)))
                neg = comp
            d_pks[t4, k - 1] = torch.from_numpy(neg).to(dev)
        elif len(b4):
            expect[b4] = 1
    torch.cuda.synchronize()
    return d_sigs, d_msgs, d_pks, expect


def cpu_baseline(d_sigs, d_msgs, d_pks, expect, k, fmt, seconds_target=12.0):
    """The oracle (CPU port, not Milagro itself -- the reference cannot be built here) on a bounded sample of the same
    batch, all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    cores = os.cpu_count() or 1
    threads = min(cores, 256)
    probe = min(4 * threads, d_sigs.shape[0])
    pkb = d_pks.shape[2]

    def run(m):
        sg = d_sigs[:m].cpu().numpy().tobytes(); ms = d_msgs[:m].cpu().numpy().tobytes(); pk = d_pks[:m].cpu().numpy().tobytes()
        t = time.perf_counter()
        got = orc.batch_fast_aggregate_verify(sg, ms, pk, m, k, fmt, nthreads=threads)
        dt = time.perf_counter() - t
        assert got == [bool(x) for x in expect[:m].tolist()], "oracle disagrees with the constructed expectation"
        return dt
    dt = run(probe)
    m = int(min(d_sigs.shape[0], max(probe, probe * seconds_target / max(dt, 1e-3))))
    m = max(threads, (m // threads) * threads)
    dt = run(m)
    return {"value": m / dt, "unit": "fast_aggregate_verify/s", "cores": threads, "kind": "port",
            "sample": "%d items (k=%d, %s keys) of the GPU batch, oracle/bls_oracle.c, %d threads, %.1f s" % (
                m, k, "compressed" if fmt == N.PK_COMPRESSED else "uncompressed", threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--items", type=int, default=1 << 16, help="items per GPU per step (default 2^16, BASELINE configs[2])")
    ap.add_argument("--keys", type=int, default=128)
    ap.add_argument("--pk-format", choices=["uncompressed", "compressed"], default="uncompressed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    ctx = N.Context(dev.index)
    lib = N.lib()
    n, k = args.items, args.keys
    fmt = N.PK_UNCOMPRESSED if args.pk_format == "uncompressed" else N.PK_COMPRESSED
    pkb = 96 if fmt == N.PK_UNCOMPRESSED else 48
    bytes_per_item = k * pkb + 96 + 32 + 1          # algorithmic bytes in + 1 out

    t_in = time.perf_counter()
    d_sigs, d_msgs, d_pks, expect = build_inputs(ctx, dev, n, k, fmt, rank)
    t_in = time.perf_counter() - t_in
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
    words = (n + 63) // 64
    d_bm = torch.zeros(words, dtype=torch.int64, device=dev)
    d_all = torch.zeros(words * world, dtype=torch.int64, device=dev) if world > 1 else None
    ctx.reserve(n)
    if fmt == N.PK_COMPRESSED:
        ctx.check(lib.mbls_ctx_reserve_keys(ctx.handle, n * k))
    stream = torch.cuda.current_stream(dev)

    def step():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, d_pks.data_ptr(), fmt, None,
                                                              n, k, d_res.data_ptr(), d_bm.data_ptr(), None, C.c_void_p(stream.cuda_stream)))
        if world > 1:
            shard.all_gather_bitmap(d_bm, world, out=d_all)  # the only collective: RCCL gather of the accept bitmap

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    lib.mbls_enable_phase_timing(ctx.handle, 0)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was timed: the accept bitmap must equal the expectation known by construction
    got = d_res.cpu()
    ok = bool(torch.equal(got, expect))
    bits = d_bm.cpu().numpy().view(np.uint64)
    unpacked = ((bits[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & 1).reshape(-1)[:n].astype(np.uint8)
    ok = ok and bool((unpacked == expect.numpy()).all())
    if world > 1:
        okt = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    # per-kernel timing with HIP events on the launch stream (separate, untimed pass) -> roofline of the dominant kernel
    lib.mbls_enable_phase_timing(ctx.handle, 1)
    phase = np.zeros(N.N_PHASES, dtype=np.float64)
    reps = max(1, min(3, args.steps))
    for _ in range(reps):
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, d_pks.data_ptr(), fmt, None,
                                                              n, k, d_res.data_ptr(), d_bm.data_ptr(), None, C.c_void_p(stream.cuda_stream)))
        ms = (C.c_float * N.N_PHASES)()
        lib.mbls_last_phase_ms(ctx.handle, ms)
        phase += np.array(list(ms))
    phase /= reps
    lib.mbls_enable_phase_timing(ctx.handle, 0)
    dom = int(np.argmax(phase))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n * world * args.steps / elapsed
        achieved = n * bytes_per_item / (phase[dom] * 1e-3) / 1e9          # algorithmic GB/s through the dominant kernel
        # integer-ALU calibration (the bound that actually applies): dependent Fp multiplications, all lanes busy
        cal_ms = C.c_float()
        ctx.check(lib.mbls_fp_mul_bench(ctx.handle, 1 << 19, 2000, C.byref(cal_ms)))
        fpmul_peak = (1 << 19) * 2000 / (cal_ms.value * 1e-3)
        # Fp multiplications per item: census of the lane bodies (tests/host_emul counts fp_mul + fp_sqr calls), committed
        census = None
        cf = os.path.join(ROOT, "profiles", "fpmul_census.json")
        if os.path.exists(cf):
            with open(cf) as f:
                census = json.load(f).get("k%d_%s" % (k, args.pk_format))
        valu = {"unit": "Fp mul/s", "peak_measured": fpmul_peak, "fp_mul_per_item": census,
                "achieved": (value / world) * census if census else None,
                "frac": (value / world) * census / fpmul_peak if census else None,
                "note": "peak = k_fp_mul_bench (dependent Montgomery multiplications with the 32-bit-limb fp_mul, 2^19 lanes) "
                        "measured in this run; fp_mul_per_item from profiles/fpmul_census.json (emulator census of fp_mul+fp_sqr "
                        "calls, Karatsuba equivalents). The Fp2 routines use fewer instructions per product than that reference "
                        "multiplier (28-bit digits), so this is a progress gauge; issue_utilisation is the hardware-side figure"}
        # VALU issue utilisation of the dominant kernel from the committed SQ counters (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)
        sqf = os.path.join(ROOT, "profiles", "r01_sq_counters.json")
        if os.path.exists(sqf):
            try:
                with open(sqf) as f:
                    sq = json.load(f).get("k_" + N.PHASE_NAMES[dom])
                valu["issue_utilisation"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
            except Exception:
                pass
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                with open(tf) as f:
                    traffic = json.load(f).get(N.PHASE_NAMES[dom])
            except Exception:
                traffic = None
        out = {
            "metric": "fast_aggregate_verify/sec (128 pubkeys, 32B msg)", "value": value, "unit": "fast_aggregate_verify/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 (12 x 32-bit Montgomery limbs; 14 x 28-bit digits with 64-bit column sums inside the multiplication routines)", "data": "synthetic",
            "config": {"workload": "configs[2]: batch 2^%d fast_aggregate_verify, %d pubkeys each, per GPU" % (int(np.log2(n)), k),
                       "items_per_gpu": n, "keys_per_item": k, "msg_bytes": 32, "pk_format": args.pk_format,
                       "negatives": "every 16th item corrupted (msg bit / wrong key / sig not in G2 / infinity sig / apk = infinity)",
                       "parallelism": "items sharded over %d GPU(s), RCCL all-gather of the accept bitmap" % world},
            "bitmap_matches_expectation": ok,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_" + N.PHASE_NAMES[dom], "kernel_ms": float(phase[dom]),
                         "algorithmic_bytes_per_item": bytes_per_item,
                         "note": "integer-ALU bound path: HBM fraction is reported as measured, see valu_roofline"},
            "valu_roofline": valu,
            "phase_ms": {nm: float(v) for nm, v in zip(N.PHASE_NAMES, phase)},
            "input_build_s": t_in,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(d_sigs, d_msgs, d_pks, expect, k, fmt)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
