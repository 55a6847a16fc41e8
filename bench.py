#!/usr/bin/env python3
"""bench.py -- fast_aggregate_verify/s (128 pubkeys, 32-byte message) on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (mbls_fast_aggregate_verify_batch_device: aggregate -> sig -> hash -> Miller ->
final exponentiation -> bitmap) over one batch of 2^16 synthetic (signature, message, 128-pubkey set) items per GPU
(BASELINE.json configs[2]), inputs resident in HBM, followed -- for N > 1 -- by the RCCL all-gather of the accept
bitmap. Weak scaling: every rank verifies its own 2^16 items, no data-path collective.
Inputs are produced on the device by the product's own signing / key kernels (the oracle is used only for the
cpu_baseline leg). PyTorch is plumbing: device buffers, stream, RCCL.

Every figure in the JSON line is measured by THIS run unless its key says otherwise (`*_from_profile` objects carry the
file and commit they were read from and are never part of `roofline` / `valu_issue` proper).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
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
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
SEED = 0x6D626C73          # "mbls" 
POOL = 4096
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
PHASE_REPS = 5             # calls of the per-kernel timing pass that count (after 2 discarded ones)
PEAK_CLOCK_GHZ = 2.4       # MI355X peak engine clock (the guide's figure): only used to express ns per instruction as "clocks at peak"
N_SIMD = 256 * 4           # 256 CUs x 4 SIMDs


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


def build_inputs(ctx, dev, n, k, fmt, rank, negatives=True, return_indices=False):
    """Synthetic batch on the device. Item i uses the k pool keys {(a_i + j*s_i) mod POOL}, s_i odd (distinct keys);
    sig_i = [sum sk mod r] H(msg_i). Every 16th item (i % 16 == 7) is corrupted, cycling over five rejection classes.
    return_indices: additionally a resident key table holding the pool (+ the crafted keys of the apk = infinity items) and the
    [n, k] uint32 table indices of every item's keys -- the same batch in the indexed representation."""
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
    table = None
    if return_indices:
        table = N.KeyTable(ctx, capacity_hint=POOL + n // 64)
        d_errs = torch.zeros(POOL, dtype=torch.uint8, device=dev)
        table.append_device(d_pool_pk.data_ptr(), POOL, d_errs.data_ptr(), pk_format=fmt, validate=False)
        torch.cuda.synchronize()
        assert int(d_errs.max().item()) == 0
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
            idx[b1, 0] = other
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
            neg = np.zeros((len(b4), 96), dtype=np.uint8)
            for r in range(len(b4)):
                y = int.from_bytes(apk[r, 48:].tobytes(), "big")
                neg[r] = np.frombuffer(apk[r, :48].tobytes() + ((P - y) % P).to_bytes(48, "big"), dtype=np.uint8)
            if table is not None:
                first, errs = table.append(neg.tobytes(), len(b4), pk_format=N.PK_UNCOMPRESSED, validate=False)
                assert not any(errs)
                idx[b4, k - 1] = first + np.arange(len(b4))
            if fmt == N.PK_COMPRESSED:
                comp = np.zeros((len(b4), 48), dtype=np.uint8); errs = np.zeros(len(b4), dtype=np.uint8)
                ctx.check(lib.mbls_pk_compress_batch(ctx.handle, neg.ctypes.data_as(C.c_void_p), len(b4), comp.ctypes.data_as(C.c_void_p),
                                                     errs.ctypes.data_as(C.c_void_p)))
                neg = comp
            d_pks[t4, k - 1] = torch.from_numpy(neg).to(dev)
        elif len(b4):
            expect[b4] = 1
    torch.cuda.synchronize()
    if return_indices:
        return d_sigs, d_msgs, d_pks, expect, torch.from_numpy(idx.astype(np.uint32).view(np.int32)).to(dev).contiguous(), table
    return d_sigs, d_msgs, d_pks, expect


# ---------------------------------------------------------------------------------------------- the timed region
def timed_steps(step, steps, warmup, world, sync):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + device synchronisation on both sides; the
    elapsed time is the MAX over ranks. `sync` synchronises the device (a no-op in the CPU test of this function)."""
    def fence():
        sync()
        if world > 1:
            dist.barrier()
            sync()
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    own = time.perf_counter() - t0
    LAST_OWN_ELAPSED[0] = own
    return reduce_max(own, world)


LAST_OWN_ELAPSED = [0.0]          # this rank's own time over the last timed region (the returned value is the MAX over ranks)


def gather_per_rank(x, world):
    """every rank's value, in rank order (per-rank ms_per_step of the JSON line)"""
    if world == 1:
        return [float(x)]
    t = torch.tensor([x], dtype=torch.float64, device=_reduce_device())
    out = torch.zeros(world, dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.cpu().tolist()]


def _reduce_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def reduce_max(x, world):
    if world == 1:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_reduce_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_all_ok(ok, world):
    if world == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_reduce_device())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def check_bitmap(results_u8, bitmap_words, expect_u8):
    """results and the packed bitmap must both equal the expectation known by construction"""
    n = expect_u8.numel()
    ok = bool(torch.equal(results_u8.cpu(), expect_u8))
    return ok and bool(torch.equal(shard.unpack_bits(bitmap_words.cpu(), n), expect_u8))


def check_gathered(all_words, world, words_per_rank, own_words, rank):
    """the gathered bitmap holds this rank's words at its slot"""
    mine = all_words[rank * words_per_rank:(rank + 1) * words_per_rank]
    return bool(torch.equal(mine.cpu(), own_words.cpu()))


def collective_figure(gather, world, sync, reps=20):
    """What a SCALE record needs to check itself: the backend and the number of ranks the process group REALLY has (RCCL's own count under 'nccl'), and the time of
    the one collective of a step -- the all-gather of the accept bitmap -- timed alone: `reps` gathers between two fences, MAX over ranks."""
    if world == 1:
        return None
    for _ in range(2):
        gather()
    sync(); dist.barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        gather()
    sync()
    ms = reduce_max((time.perf_counter() - t0) / reps * 1e3, world)
    return {"backend": dist.get_backend(), "group_world_size": dist.get_world_size(), "group_rank_count_matches_n_gpus": dist.get_world_size() == world,
            "all_gather_ms_per_step": ms, "reps": reps}


# ---------------------------------------------------------------------------------------------- CPU baseline (the oracle)
def _oracle_native():
    """oracle/bls_oracle.c built -O3 -march=native for THIS host (the shipped oracle/_build library is a portable build)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    out = os.path.join(ROOT, "oracle", "_build", "libbls_oracle_native.so")
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=gnu11", "-w", "-shared", "-o", out, os.path.join(ROOT, "oracle", "bls_oracle.c"), "-lpthread"])
        orc.LIB_PATH = out
        orc._lib = None
        flags = "-O3 -march=native"
    except Exception:
        flags = "portable build (native rebuild failed)"
    return orc, flags


def usable_cores():
    """threads the CPU leg can really run at once: the affinity set, capped by the cgroup CPU quota when the container has one (a
    256-thread affinity mask over an 8-CPU quota runs 8 threads' worth of work, however many are started)"""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:        # cgroup v1
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = float(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(d_sigs, d_msgs, d_pks, expect, k, fmt, seconds_target=10.0):
    """The oracle (a CPU restatement -- "port" -- NOT Milagro/amcl: the reference cannot be built here, and amcl is several times
    faster per core than this plain C) on a bounded sample of the same batch: all cores of this process's affinity set, one thread,
    and BASELINE config 1 (single Signature::verify, one thread)."""
    orc, flags = _oracle_native()
    threads, quota = usable_cores()

    def run(m, nthreads):
        sg = d_sigs[:m].cpu().numpy().tobytes(); ms = d_msgs[:m].cpu().numpy().tobytes(); pk = d_pks[:m].cpu().numpy().tobytes()
        t = time.perf_counter()
        got = orc.batch_fast_aggregate_verify(sg, ms, pk, m, k, fmt, nthreads=nthreads)
        dt = time.perf_counter() - t
        assert got == [bool(x) for x in expect[:m].tolist()], "oracle disagrees with the constructed expectation"
        return dt
    fmt_name = "compressed" if fmt == N.PK_COMPRESSED else "uncompressed"
    # one thread
    dt = run(8, 1)
    m1 = int(max(8, min(d_sigs.shape[0], 8 * 4.0 / max(dt, 1e-3))))
    dt1 = run(m1, 1)
    one = {"value": m1 / dt1, "unit": "fast_aggregate_verify/s", "cores": 1, "kind": "port",
           "sample": "%d items (k=%d, %s keys) of the GPU batch, oracle/bls_oracle.c %s, 1 thread, %.1f s" % (m1, k, fmt_name, flags, dt1)}
    # all cores
    probe = min(2 * threads, d_sigs.shape[0])
    dt = run(probe, threads)
    m = int(min(d_sigs.shape[0], max(probe, probe * seconds_target / max(dt, 1e-3))))
    m = max(threads, (m // threads) * threads)
    dt = run(m, threads)
    allc = {"value": m / dt, "unit": "fast_aggregate_verify/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "sample": "%d items (k=%d, %s keys) of the GPU batch, oracle/bls_oracle.c %s (CPU restatement, not Milagro), %d threads "
                      "= this process's affinity set (os.cpu_count() = %d), %.1f s" % (m, k, fmt_name, flags, threads, os.cpu_count() or 0, dt)}
    # BASELINE config 1: single Signature::verify, msg "Some msg", one thread (reference benches/bls381_benches.rs:104-112)
    sk = 0x263dbd792f5b1be47ed85f8938c0f29586af0d3ac7b977f21c278fe1462040e3
    msg = b"Some msg"
    sig = orc.sign(msg, sk); pk = orc.sk_to_pk(sk)
    reps, t = 0, time.perf_counter()
    while time.perf_counter() - t < 2.0:
        assert orc.verify(sig, msg, pk)
        reps += 1
    cfg1 = {"value": reps / (time.perf_counter() - t), "unit": "Signature::verify/s", "cores": 1, "kind": "port",
            "sample": "configs[0]: %d x verify of one (sig, 'Some msg', pk), 1 thread" % reps}
    return allc, one, cfg1


# ---------------------------------------------------------------------------------------------- instruction-level figure
def valu_issue_figure(lib, ctx, phase_ms, n_items, agg_kernel="k_aggregate"):
    """Hardware-side efficiency of the generated kernels from numbers this run measures plus the generators' exact instruction
    counts (profiles/instr_census.json, written by tools/instr_census.py from the same generators that emit the routines):
    achieved = wave-instructions per item x items / 64 lanes / kernel time / 1024 SIMDs;
    ceiling  = the rate one SIMD issues this instruction mix at, measured live by mbls_valu_bench (8 waves per SIMD): v_mad_u64_u32
    and plain 32-bit VALU timed separately and blended by the routines' mix. Only the generated routines are counted (not the few
    hundred compiler-scheduled instructions around them), so `frac` is a lower bound. agg_kernel: "k_aggregate" (the census row for
    128 96-byte keys per item) -- pass None when the timed keys had another format."""
    cf = os.path.join(ROOT, "profiles", "instr_census.json")
    if not os.path.exists(cf):
        return None
    with open(cf) as f:
        census = json.load(f)
    rate, by_waves = {}, {}
    for mode, name in ((0, "v_mad_u64_u32"), (1, "plain_valu"), (7, "v_mad_i64_i32")):
        best = 0.0
        for w in (1, 2, 4, 8):                                      # the best rate any occupancy reaches (two calls each, the faster one)
            r_w = 0.0
            for _ in range(2):
                ms = C.c_float()
                ctx.check(lib.mbls_valu_bench(ctx.handle, mode, w, 4000, C.byref(ms)))
                r_w = max(r_w, w * 4000 * 128 / (ms.value * 1e-3))  # wave-instructions per second per SIMD
            by_waves["%s@%dw" % (name, w)] = r_w
            best = max(best, r_w)
        rate[name] = best
    # the rate of a bare loop of the Fp2 product routine (1 281 instructions, 980 multiply-accumulates) on ONE wave per SIMD -- the occupancy
    # the pipeline kernels run at: the reference for "how much does everything around the products cost in issue rate"
    r_loop = 0.0
    for mode, instr, name in ((2, 1281, "fp2_product_loop@1w"), (3, 923, "fp_pair_product_loop@1w")):
        r_m = 0.0
        for _ in range(3):                                          # ~10 ms each: long enough for the clock to settle under this load
            ms = C.c_float()
            ctx.check(lib.mbls_valu_bench(ctx.handle, mode, 1, 500, C.byref(ms)))
            r_m = max(r_m, 500 * 8 * instr / (ms.value * 1e-3))
        rate[name] = r_m
        r_loop = max(r_loop, r_m)
    out = {"unit": "wave-instructions/s/SIMD", "ceiling_measured": rate, "ceiling_by_waves_per_simd": by_waves, "kernels": {},
           "note": "ceiling = the best rate mbls_valu_bench reaches at 1, 2, 4 or 8 waves per SIMD, measured live (a calibration stream of multiply-accumulates / "
                   "of add-with-carry pairs; a kernel whose mix runs at a higher clock can still come within a percent of 1); instruction counts = generated "
                   "routines only (exact), cold paths excluded: frac is a lower bound"}
    for kern, phase in (("k_miller", "miller"), ("k_final", "final"), (agg_kernel, "aggregate")):
        c = census["per_item"].get(kern)
        if not c or phase_ms.get(phase, 0) <= 0:
            continue
        total, mad = c["valu"], c["mad_u64_u32"]
        ach = total * n_items / 64.0 / (phase_ms[phase] * 1e-3) / N_SIMD
        blend = total / (mad / rate["v_mad_u64_u32"] + (total - mad) / rate["plain_valu"])
        ceil = max(blend, r_loop)                                       # whichever calibration stream issues faster
        ns = 1e9 / ach                                                  # SIMD time per wave-instruction
        # mac_frac = the MAC-only roofline the review asked for: multiply-accumulate wave-instructions / kernel time against the best rate a bare stream of
        # v_mad_u64_u32 / v_mad_i64_i32 reaches at any occupancy (measured above; mac_frac_vs_r02_ubench_0p571: against the 0.571 G wave-instructions/s/SIMD of
        # profiles/r02_ubench.txt's "mad+addc dependent" row at 8 waves, the figure the round-5 review used) -- 1 - mac_frac is what everything that is not a multiply-accumulate (reduction masks and shifts, carry passes,
        # additions between products, register moves, memory instructions, calls) and the one-wave issue rate cost together
        mac_ach = mad * n_items / 64.0 / (phase_ms[phase] * 1e-3) / N_SIMD
        mac_peak = max(max(by_waves["v_mad_u64_u32@%dw" % w], by_waves["v_mad_i64_i32@%dw" % w]) for w in (1, 2, 4, 8))       # the best any occupancy reaches, either opcode
        out["kernels"][kern] = {"valu_wave_instr_per_item": total, "of_which_v_mad_u64_u32": mad, "achieved": ach, "ceiling": ceil, "ceiling_blend": blend, "ceiling_product_loop": r_loop, "frac": ach / ceil,
                                "mac_achieved": mac_ach, "mac_peak": mac_peak, "mac_frac": mac_ach / mac_peak, "mac_share_of_valu": mad / float(total),
                                "mac_frac_vs_r02_ubench_0p571": mac_ach / 0.571e9,
                                "ns_per_valu_instr": ns, "clk_per_valu_instr": ns * PEAK_CLOCK_GHZ,
                                "clk_note": "ns x %.1f GHz peak clock; the clock under this instruction mix is lower, see DESIGN.md section 4" % PEAK_CLOCK_GHZ}
    return out


def _med_ms(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


def other_configs(ctx, lib, dev, sptr):
    """BASELINE configs[1] (2^16 x Signature::verify) and configs[3] (verify_multiple, 2^14 sets x 128 keys, against the same sets one by
    one), and the small-batch latency of the hot path: median of 5 enqueue + synchronise calls, inputs resident in HBM, results checked"""
    from milagro_bls_amd import batch
    out = {"_unit": "ms per call (median of 5 after 2 warm-ups), inputs resident in HBM"}
    n = 1 << 16
    c_sigs, c_msgs, c_pks, c_exp = build_inputs(ctx, dev, n, 1, N.PK_COMPRESSED, rank=11)
    c_res = torch.zeros(n, dtype=torch.uint8, device=dev)

    def f_c2():
        ctx.check(lib.mbls_verify_batch_device(ctx.handle, c_sigs.data_ptr(), c_msgs.data_ptr(), 32, None, c_pks.data_ptr(), N.PK_COMPRESSED, n, c_res.data_ptr(), None, None, sptr))
    t = _med_ms(f_c2)
    out["configs[1] 2^16 x Signature::verify (compressed key)"] = {"ms": t, "verify_per_s": n / t * 1e3, "correct": bool(torch.equal(c_res.cpu(), c_exp))}
    k = 128
    for nn in (1 << 14, 1 << 16):
        v_sigs, v_msgs, v_pks, v_exp = build_inputs(ctx, dev, nn, k, N.PK_UNCOMPRESSED, rank=12, negatives=False)
        g = torch.Generator(device="cpu"); g.manual_seed(7)
        rands = torch.randint(1, (1 << 62), (nn,), dtype=torch.int64, generator=g).to(dev)
        d_r = torch.full((8,), 7, dtype=torch.uint8, device=dev)

        def f_c4():
            batch.verify_multiple_sets_device(v_sigs.data_ptr(), v_pks.data_ptr(), v_msgs.data_ptr(), rands.data_ptr(), nn, k, pk_format=N.PK_UNCOMPRESSED,
                                              d_result=d_r.data_ptr(), stream=sptr)
        t4 = _med_ms(f_c4)
        ok_true = int(d_r[0].item()) == 1
        v_msgs[nn // 3, 5] ^= 0x10; f_c4(); torch.cuda.synchronize(); ok_false = int(d_r[0].item()) == 0; v_msgs[nn // 3, 5] ^= 0x10
        v_res = torch.zeros(nn, dtype=torch.uint8, device=dev)

        def f_one():
            ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, v_sigs.data_ptr(), v_msgs.data_ptr(), 32, None, v_pks.data_ptr(), N.PK_UNCOMPRESSED, None, nn, k,
                                                                  v_res.data_ptr(), None, None, sptr))
        t1 = _med_ms(f_one)
        name = "configs[3] verify_multiple 2^14 sets x 128 keys" if nn == (1 << 14) else "verify_multiple 2^16 sets x 128 keys"
        out[name] = {"ms": t4, "sets_per_s": nn / t4 * 1e3, "same_sets_one_by_one_ms": t1, "correct": bool(ok_true and ok_false and v_res.all().item())}
        if nn == (1 << 14):
            lat = {}
            for m in (1, 64, 1024, 4096):
                l_res = torch.zeros(m, dtype=torch.uint8, device=dev)

                def f_l():
                    ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, v_sigs.data_ptr(), v_msgs.data_ptr(), 32, None, v_pks.data_ptr(), N.PK_UNCOMPRESSED, None, m, k,
                                                                          l_res.data_ptr(), None, None, sptr))
                lat[str(m)] = _med_ms(f_l)
                assert bool(l_res.all().item())
            out["latency fast_aggregate_verify (128 keys) by batch size"] = lat
        else:
            # batches ABOVE a round (nn = 2^16 here): the two-track routes of verify_pipeline -- the remainder beside the last round, equal halves -- on the same
            # items repeated (items are independent); every result checked
            above = {}
            for m in (69632, 73728, 100000, 150000):
                rep = -(-m // nn)
                b_s = v_sigs.repeat(rep, 1)[:m].contiguous(); b_m = v_msgs.repeat(rep, 1)[:m].contiguous(); b_p = v_pks.repeat(rep, 1, 1)[:m].contiguous()
                b_res = torch.zeros(m, dtype=torch.uint8, device=dev)

                def f_b():
                    ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, b_s.data_ptr(), b_m.data_ptr(), 32, None, b_p.data_ptr(), N.PK_UNCOMPRESSED, None, m, k,
                                                                          b_res.data_ptr(), None, None, sptr))
                t_b = _med_ms(f_b)
                above[str(m)] = {"ms": t_b, "per_s": m / t_b * 1e3, "correct": bool(b_res.all().item())}
                del b_s, b_m, b_p
            out["batches above a round (two tracks), 128 keys"] = dict(above, correct=all(v["correct"] for v in above.values()))
    return out


def criterion_shapes():
    """The reference's own criterion points for the path, through the API mirror (host objects in, bool out: PCIe and synchronisation included, like a caller of
    the drop-in sees them): `Verify a Signature` (benches/bls381_benches.rs:104-112) and `multiple-signatures-verification-30` (:179-245: n = 10 aggregate
    signatures x m = 3 keys through verify_multiple_aggregate_signatures with a fresh rng per iteration), beside the same 10 sets verified one by one."""
    import random
    import milagro_bls_amd as m
    rnd = random.Random(20)
    sk = m.SecretKey.from_bytes(rnd.randrange(1, R).to_bytes(32, "big")); pk = m.PublicKey.from_secret_key(sk)
    msg = b"Some msg"; sig = m.Signature.new(msg, sk)                  # benches/bls381_benches.rs:86-90
    out = {"_unit": "ms per call, median of 5 after 2 warm-ups, host objects in / bool out (PCIe + synchronisation included)"}
    ok = sig.verify(msg, pk)
    out["Signature::verify (one signature)"] = _med_ms(lambda: sig.verify(msg, pk))
    n, mk = 10, 3
    sets = []
    for i in range(n):
        agg = m.AggregateSignature.new(); pks = []
        mi = bytes([i]) * 32
        for _ in range(mk):
            s1 = m.SecretKey.from_bytes(rnd.randrange(1, R).to_bytes(32, "big"))
            agg.add(m.Signature.new(mi, s1)); pks.append(m.PublicKey.from_secret_key(s1))
        sets.append((agg, m.AggregatePublicKey.into_aggregate(pks), mi))
    ok = ok and m.AggregateSignature.verify_multiple_aggregate_signatures(random.Random(1), sets)
    out["verify_multiple_aggregate_signatures (10 sets x 3 keys)"] = _med_ms(lambda: m.AggregateSignature.verify_multiple_aggregate_signatures(random.Random(1), sets))
    ok = ok and all(a.fast_aggregate_verify_pre_aggregated(mm, apk) for a, apk, mm in sets)
    out["the same 10 sets one by one (fast_aggregate_verify_pre_aggregated)"] = _med_ms(lambda: [a.fast_aggregate_verify_pre_aggregated(mm, apk) for a, apk, mm in sets])
    bad = list(sets); bad[4] = (sets[3][0], sets[4][1], sets[4][2])
    ok = ok and not m.AggregateSignature.verify_multiple_aggregate_signatures(random.Random(1), bad)
    out["correct"] = bool(ok)
    return out


def aggregate_verify_leg(ctx, lib, dev, sptr, n=1 << 14, kp=4):
    """n x AggregateSignature::aggregate_verify in one call (reference src/aggregates.rs:130-170): n items of kp (message, key)
    pairs each, distinct 32-byte messages, signatures aggregated on the device. All valid, then one message flipped: only that item fails."""
    pool = make_pool(SEED)
    pool_b = np.frombuffer(b"".join(s.to_bytes(32, "big") for s in pool), dtype=np.uint8).reshape(POOL, 32)
    d_pool_sk = torch.from_numpy(pool_b.copy()).to(dev)
    d_pool_pk = torch.empty((POOL, 96), dtype=torch.uint8, device=dev)
    ctx.check(lib.mbls_sk_to_pk_batch_device(ctx.handle, d_pool_sk.data_ptr(), N.PK_UNCOMPRESSED, POOL, d_pool_pk.data_ptr(), None))
    rng = np.random.default_rng(SEED + 77)
    total = n * kp
    idx = torch.from_numpy(rng.integers(0, POOL, size=total, dtype=np.int64)).to(dev)
    d_msgs = torch.from_numpy(rng.integers(0, 256, size=(total, 32), dtype=np.uint8)).to(dev)
    d_sk = d_pool_sk[idx].contiguous(); d_pks = d_pool_pk[idx].contiguous()
    d_psig = torch.empty((total, 96), dtype=torch.uint8, device=dev)
    ctx.check(lib.mbls_sign_batch_device(ctx.handle, d_sk.data_ptr(), d_msgs.data_ptr(), 32, total, d_psig.data_ptr(), None))
    d_sigs = torch.empty((n, 96), dtype=torch.uint8, device=dev); d_errs = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.check(lib.mbls_aggregate_signatures_batch_device(ctx.handle, d_psig.data_ptr(), None, n, kp, total, d_sigs.data_ptr(), d_errs.data_ptr(), None))
    torch.cuda.synchronize()
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev)

    def f():
        ctx.check(lib.mbls_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), None, kp, total, n,
                                                         d_res.data_ptr(), None, sptr))
    t = _med_ms(f)
    ok = bool(d_res.all().item()) and int(d_errs.max().item()) == 0
    j = n // 3
    d_msgs[j * kp + 1, 4] ^= 8; f(); torch.cuda.synchronize()
    ok = ok and int(d_res[j].item()) == 0 and int(d_res.sum().item()) == n - 1
    d_msgs[j * kp + 1, 4] ^= 8
    return {"ms": t, "aggregate_verify_per_s": n / t * 1e3, "pairs_per_s": total / t * 1e3, "correct": ok}


def keyops_figures(ctx, lib, dev, n=1 << 16):
    """The operations either side of the path (the reference's criterion groups benches/bls381_benches.rs:10-84 (de)compression, :115-147
    signing, :247-268 key generation; KeyValidate = src/keys.rs:182): batches of n through the device entries where they exist (host
    entries otherwise, PCIe included and labelled so), median of 3 after one warm-up."""
    out = {"_unit": "ms per batch of %d (median of 3 after 1 warm-up)" % n}
    g = torch.Generator(device="cpu"); g.manual_seed(99)
    sks = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g); sks[:, 0] &= 0x3F; sks[:, 31] |= 1
    msgs = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
    d_sk, d_msg = sks.to(dev), msgs.to(dev)
    d_sig = torch.empty((n, 96), dtype=torch.uint8, device=dev); d_pk = torch.empty((n, 48), dtype=torch.uint8, device=dev)
    ctx.reserve(4 * n)
    t = _med_ms(lambda: ctx.check(lib.mbls_sign_batch_device(ctx.handle, d_sk.data_ptr(), d_msg.data_ptr(), 32, n, d_sig.data_ptr(), None)), reps=3, warm=1)
    out["Signature::new (device entry)"] = {"ms": t, "per_s": n / t * 1e3}
    t = _med_ms(lambda: ctx.check(lib.mbls_sk_to_pk_batch_device(ctx.handle, d_sk.data_ptr(), 0, n, d_pk.data_ptr(), None)), reps=3, warm=1)
    out["PublicKey::from_secret_key (device entry, compressed out)"] = {"ms": t, "per_s": n / t * 1e3}
    pk48 = d_pk.cpu().numpy(); pk96 = np.zeros((n, 96), dtype=np.uint8); errs = np.zeros(n, dtype=np.uint8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for validate, name in ((0, "PublicKey::from_bytes_unchecked (decompress; host entry, PCIe included)"), (1, "PublicKey::from_bytes (decompress + KeyValidate; host entry, PCIe included)")):
        t = _med_ms(lambda: ctx.check(lib.mbls_pk_decode_batch(ctx.handle, vp(pk48), 0, validate, n, vp(pk96), vp(errs))), reps=3, warm=1)
        out[name] = {"ms": t, "per_s": n / t * 1e3, "correct": not errs.any()}
    back = np.zeros((n, 48), dtype=np.uint8)
    t = _med_ms(lambda: ctx.check(lib.mbls_pk_compress_batch(ctx.handle, vp(pk96), n, vp(back), vp(errs))), reps=3, warm=1)
    out["PublicKey::as_bytes (compress; host entry, PCIe included)"] = {"ms": t, "per_s": n / t * 1e3, "correct": bool((back == pk48).all())}
    sig = d_sig.cpu().numpy(); ing2 = np.zeros(n, dtype=np.uint8)
    t = _med_ms(lambda: ctx.check(lib.mbls_sig_check_batch(ctx.handle, vp(sig), n, vp(errs), vp(ing2))), reps=3, warm=1)
    out["Signature::from_bytes + subgroup check (host entry, PCIe included)"] = {"ms": t, "per_s": n / t * 1e3, "correct": bool(ing2.all()) and not errs.any()}
    # AggregatePublicKey::add / AggregateSignature::add (one addition per call in the reference's benches): sums of 2
    out2 = np.zeros((n // 2, 96), dtype=np.uint8); e2 = np.zeros(n // 2, dtype=np.uint8)
    t = _med_ms(lambda: ctx.check(lib.mbls_aggregate_signatures_batch(ctx.handle, vp(sig), None, n // 2, 2, vp(out2), vp(e2))), reps=3, warm=1)
    out["AggregateSignature::add (sets of 2; host entry, PCIe included)"] = {"ms": t, "per_s": (n // 2) / t * 1e3, "correct": not e2.any()}
    return out


def multi_handle_leg(lib, table, d_sigs, d_msgs, d_idx, expect, n, k, devices):
    """The in-process multi-GPU entry (include/mbls.h, mbls_multi_*): ONE process, one context + host thread per listed device, the caller's
    HOST buffers cut into contiguous shards, results written in place. n items per device (weak scaling, like the per-rank legs), key
    indices into a replicated key table -- PCIe included (uploads of 96 + 32 + 4 k bytes per item, download of 1)."""
    from milagro_bls_amd import batch
    G = len(devices)
    sigs = np.tile(d_sigs.cpu().numpy(), (G, 1)); msgs = np.tile(d_msgs.cpu().numpy(), (G, 1)); idx = np.tile(d_idx.cpu().numpy(), (G, 1))
    want = np.tile(expect.numpy(), G)
    # the rank's key table (the pool + the crafted keys of the apk = infinity items) read back and replicated: same indices everywhere
    tsize = len(table)
    keys96 = np.zeros((tsize, 96), dtype=np.uint8); kerr = np.zeros(tsize, dtype=np.uint8)
    table.ctx.check(lib.mbls_keytable_get(table.handle, 0, tsize, keys96.ctypes.data_as(C.c_void_p), kerr.ctypes.data_as(C.c_void_p)))
    assert not kerr.any()
    m = N.MultiContext(devices)
    try:
        m.reserve(n * G)
        tab = N.MultiKeyTable(m, capacity_hint=tsize)
        first, errs = tab.append(keys96.tobytes(), tsize, pk_format=N.PK_UNCOMPRESSED, validate=False)
        assert first == 0 and not any(errs)
        res = np.zeros(n * G, dtype=np.uint8)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)

        def call():
            m.check(lib.mbls_multi_fast_aggregate_verify_batch_indexed(m.handle, tab.handle, vp(sigs), vp(msgs), 32, None, vp(idx), None, n * G, k, vp(res), None))
        call()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
        t = float(np.median(ts))
        # the handle's exchange step: the accept bitmap packed on every device and all-gathered BETWEEN the devices (native RCCL when the handle has a communicator:
        # one rank on a one-GPU box, xGMI on a node), on a small batch of byte keys -- every device ends with the whole bitmap, the first one's copy is checked
        nb = min(4096, n) * G
        pk96 = keys96[idx[:nb].reshape(-1)].reshape(nb, -1)
        words = np.zeros((nb + 63) // 64, dtype=np.uint64)

        def gather():
            m.check(lib.mbls_multi_fast_aggregate_verify_bitmap(m.handle, vp(sigs), vp(msgs), 32, None, vp(pk96), N.PK_UNCOMPRESSED, None, nb, k, vp(words), None))
        gather()
        tg = time.perf_counter(); gather(); tg = time.perf_counter() - tg
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:nb]
        gathered_ok = bool((bits == want[:nb]).all())
        note, active = m.exchange_note, m.rccl_active
        tab.close()
    finally:
        m.close()
    return {"devices": list(devices), "items": n * G, "ms_per_call": t * 1e3, "value": n * G / t, "unit": "fast_aggregate_verify/s",
            "what": "mbls_multi_fast_aggregate_verify_batch_indexed: one process, host buffers in, results out (PCIe included), %d items per device" % n,
            "results_match": bool((res == want).all()) and gathered_ok,
            "bitmap_gather": {"entry": "mbls_multi_fast_aggregate_verify_bitmap", "items": nb, "ms_per_call": tg * 1e3, "rccl_active": active, "exchange": note,
                              "gathered_bitmap_matches": gathered_ok}}


def multi_leg_child(args):
    """`bench.py --multi-leg G` (started by multi_leg_in_child): the same synthetic batch as rank 0's, the leg over devices 0 .. G-1, its JSON on stdout"""
    protect_stdout()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = N.Context(0); lib = N.lib()
    n, k = args.items, args.keys
    d_sigs, d_msgs, d_pks, expect, d_idx, table = build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, 0, return_indices=True)
    devices = list(range(args.multi_leg))
    if os.environ.get("MBLS_MULTI_LEG_DEVICES"):          # tests on a one-GPU box: "0,0" = two contexts on device 0 (host join: RCCL wants one rank per device)
        devices = [int(x) for x in os.environ["MBLS_MULTI_LEG_DEVICES"].split(",")]
    leg = multi_handle_leg(lib, table, d_sigs, d_msgs, d_idx, expect, n, k, devices)
    emit_result(json.dumps(leg))
    return 0


def multi_leg_in_child(G, n, k, timeout_s=600):
    cmd = [sys.executable, os.path.abspath(__file__), "--multi-leg", str(G), "--items", str(n), "--keys", str(k)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"devices": list(range(G)), "error": "no answer within %d s (child process ended)" % timeout_s}
    except OSError as e:
        return {"devices": list(range(G)), "error": "cannot start the child process: %s" % e}
    for l in reversed(r.stdout.decode(errors="replace").splitlines()):
        try:
            d = json.loads(l)
            if isinstance(d, dict) and "devices" in d:
                return d
        except ValueError:
            pass
    return {"devices": list(range(G)), "error": "child exit code %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])}


def config5_leg(ctx, lib, dev, sptr, rank, world, k):
    """2^17 items on this rank (2^20 over 8 GPUs) + the bitmap gather: 3 timed steps"""
    n = 1 << 17
    d_sigs, d_msgs, d_pks, expect = build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank + 100)
    ctx.reserve(n)
    words = n // 64
    d_res = torch.zeros(n, dtype=torch.uint8, device=dev); d_bm = torch.zeros(words, dtype=torch.int64, device=dev)
    d_all = torch.zeros(words * world, dtype=torch.int64, device=dev)

    def step():
        ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_pks.data_ptr(), N.PK_UNCOMPRESSED, None,
                                                              n, k, d_res.data_ptr(), d_bm.data_ptr(), None, sptr))
        shard.all_gather_bitmap(d_bm, world, out=d_all)
    steps = 3
    elapsed = timed_steps(step, steps, 1, world, torch.cuda.synchronize)
    ok = reduce_all_ok(check_bitmap(d_res, d_bm, expect) and check_gathered(d_all, world, words, d_bm, rank), world)
    return {"workload": "configs[4]: 2^20 fast_aggregate_verify sharded over 8 GPUs (2^17 per rank), RCCL gather of the accept bitmap",
            "value": n * world * steps / elapsed, "unit": "fast_aggregate_verify/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "bitmap_matches_expectation": ok}


def sharded_verify_multiple_leg(ctx, lib, dev, sptr, rank, world, k, nn=1 << 14):
    """BASELINE configs[3] cut into shards ("one exchange step"): every rank runs 2^14 sets x k keys up to its 896-byte
    record (mbls_verify_multiple_partial_device), the records are all-gathered (RCCL), every rank joins them in rank order
    (mbls_verify_multiple_finish_device) and holds the same bool. On one GPU the same sets go through as two shards + join, beside the one-call time."""
    from milagro_bls_amd import batch
    P = N.VM_PARTIAL_BYTES
    v_sigs, v_msgs, v_pks, _ = build_inputs(ctx, dev, nn, k, N.PK_UNCOMPRESSED, rank=12 + 31 * rank, negatives=False)
    g = torch.Generator(device="cpu"); g.manual_seed(7 + rank)
    rands = torch.randint(1, (1 << 62), (nn,), dtype=torch.int64, generator=g).to(dev)
    shards = world if world > 1 else 2
    recs = torch.zeros(shards * P, dtype=torch.uint8, device=dev)
    mine = torch.zeros(P, dtype=torch.uint8, device=dev)
    d_r = torch.full((8,), 7, dtype=torch.uint8, device=dev)

    def step():
        if world > 1:
            batch.verify_multiple_partial_device(v_sigs.data_ptr(), v_msgs.data_ptr(), rands.data_ptr(), nn, mine.data_ptr(), d_pks=v_pks.data_ptr(), k=k,
                                                 pk_format=N.PK_UNCOMPRESSED, stream=sptr)
            shard.all_gather_records(mine, world, out=recs)
        else:
            h = nn // 2
            for j, (lo, cnt) in enumerate(((0, h), (h, nn - h))):
                batch.verify_multiple_partial_device(v_sigs[lo:].data_ptr(), v_msgs[lo:].data_ptr(), rands[lo:].data_ptr(), cnt, recs.data_ptr() + j * P,
                                                     d_pks=v_pks[lo:].data_ptr(), k=k, pk_format=N.PK_UNCOMPRESSED, stream=sptr)
        batch.verify_multiple_finish_device(recs.data_ptr(), shards, d_result=d_r.data_ptr(), stream=sptr)
    steps = 5
    elapsed = timed_steps(step, steps, 2, world, torch.cuda.synchronize)
    ok_true = int(d_r[0].item()) == 1
    if rank == world - 1:                                                  # one corrupted set on the last rank must turn every rank's bool
        v_msgs[nn // 3, 5] ^= 0x10
    step(); torch.cuda.synchronize()
    ok_false = int(d_r[0].item()) == 0
    if rank == world - 1:
        v_msgs[nn // 3, 5] ^= 0x10
    ok = reduce_all_ok(ok_true and ok_false, world)
    return {"workload": "verify_multiple_aggregate_signatures, %d sets x %d keys per GPU, %d shard(s) + one join%s" % (
                nn, k, shards, " (RCCL all-gather of %d x %d bytes)" % (world, P) if world > 1 else " on one device"),
            "ms_per_step": elapsed / steps * 1e3, "sets_per_s": nn * world * steps / elapsed, "steps": steps, "correct": ok}


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def spawn_ranks(n, argv, stub=False, timeout_s=3600):
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment), relay rank 0's JSON line, return the worst exit code. The parent never touches a GPU (torch.cuda.device_count() does not
    initialise one on this image) and never re-executes itself. Fewer devices than ranks, a rank that cannot start or a missing line is an
    error exit -- never a silent one-GPU run."""
    if not stub and os.environ.get("MBLS_BENCH_SHARE_GPU") != "1":
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d but only %d device(s) visible" % (n, have), file=sys.stderr)
            return 4
    env0 = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    procs = []
    try:
        for r in range(n):
            env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    except OSError as e:
        print("bench.py: cannot start rank %d: %s" % (len(procs), e), file=sys.stderr)
        for p in procs:
            p.kill()
        return 6
    t0 = time.time()
    codes = [None] * n
    t_failed, grace_s = None, 20.0
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed and t_failed is None:
            t_failed = time.time()
        # one rank is gone: the others would wait in a collective for ever -- but a failure every rank has agreed on (a parity mismatch is reduced over the
        # ranks, all of them then leave with the same code) ends them by itself within moments: they get a grace period before they are killed
        if (t_failed is not None and time.time() - t_failed > grace_s) or time.time() - t0 > timeout_s:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill(); codes[i] = p.wait() or 9
            break
        time.sleep(0.2)
    out0 = procs[0].stdout.read().decode() if procs[0].stdout else ""
    line = None
    for l in out0.splitlines():
        try:
            d = json.loads(l)
            if isinstance(d, dict) and d.get("n_gpus") == n:
                line = l
        except ValueError:
            pass
    worst = max(abs(c) for c in codes)
    if line is None:
        print("bench.py: rank 0 printed no result line for %d GPUs (exit codes %s)" % (n, codes), file=sys.stderr)
        return worst or 7
    emit_result(line)
    return worst


def stub_rank(args, rank, world):
    """--stub: the multi-rank plumbing of this file under gloo on the CPU with a stand-in for the verifier (tests/test_bench_cpu.py). Same
    timed region, gather, reductions and JSON shape as the real path; "stub": true marks the line as no measurement."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    words = shard.bitmap_words(n)
    expect = torch.ones(n, dtype=torch.uint8); expect[7::16] = 0
    res = expect.clone()
    if os.environ.get("MBLS_STUB_FAIL_RANK") == str(rank):
        res[5] ^= 1
    bm = shard.pack_bits(res)
    d_all = torch.zeros(words * world, dtype=torch.int64)

    calls = [0]

    def step():
        time.sleep(0.005 * (rank + 1))
        calls[0] += 1
        if os.environ.get("MBLS_STUB_DIE_RANK") == str(rank) and calls[0] == 2:
            os._exit(17)                               # a rank that dies in the middle of the job (tests/test_bench_cpu.py): no line, a non-zero exit, no hang
        if world > 1:
            shard.all_gather_bitmap(bm, world, out=d_all)
    elapsed = timed_steps(step, args.steps, args.warmup, world, lambda: None)
    per_rank = gather_per_rank(LAST_OWN_ELAPSED[0] / args.steps * 1e3, world)
    coll = collective_figure(lambda: shard.all_gather_bitmap(bm, world, out=d_all), world, lambda: None, reps=5)
    # configs[4] as the real path times it at 8 ranks (config5_leg): the rank's own shard, then the gather -- here with the stand-in verifier
    shard_leg = None
    if world == 8:
        e5 = timed_steps(step, 2, 0, world, lambda: None)
        shard_leg = {"workload": "configs[4] (stub): %d items per rank over 8 ranks + the bitmap gather" % n, "value": n * world * 2 / e5, "unit": "fast_aggregate_verify/s",
                     "ms_per_step": e5 / 2 * 1e3, "steps": 2,
                     "bitmap_matches_expectation": reduce_all_ok(check_gathered(d_all, world, words, bm, rank), world)}
    ok = check_bitmap(res, bm, expect) and (world == 1 or check_gathered(d_all, world, words, bm, rank))
    # the exchange step of the sharded verify_multiple leg: one byte record per rank, gathered in rank order
    mine = torch.full((N.VM_PARTIAL_BYTES,), rank + 1, dtype=torch.uint8)
    recs = shard.all_gather_records(mine, world, out=torch.zeros(world * N.VM_PARTIAL_BYTES, dtype=torch.uint8))
    ok = ok and all(bool((recs[r * N.VM_PARTIAL_BYTES:(r + 1) * N.VM_PARTIAL_BYTES] == r + 1).all()) for r in range(world))
    ok = reduce_all_ok(ok, world)
    if rank == 0:
        emit_result(json.dumps({"metric": "fast_aggregate_verify/sec (128 pubkeys, 32B msg)", "stub": True, "value": n * world * args.steps / elapsed,
                          "unit": "fast_aggregate_verify/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "ms_per_step_per_rank": per_rank, "bitmap_matches_expectation": ok, "higher_is_better": True, "scaling": "weak",
                          "collective": coll, "configs4_shard_leg": shard_leg}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    return 0 if ok else 3


_RESULT_FD = None


def protect_stdout():
    """stdout carries ONE JSON line (the driver parses it). Libraries under this process write to the C stdout as well -- RCCL prints a version banner when a
    communicator is made, buffered until exit --, so file descriptor 1 is pointed at stderr for the whole run and the result line goes to the original one."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit_result(line):
    sys.stdout.flush()
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, (line + "\n").encode())


def git_head():
    """the commit this tree is at: from git where .git exists, else from .git_head (written by the post-commit hook scripts/post-commit.sh installs; the GPU
    box receives the tree without .git) -- labelled, because a file can lag one commit behind"""
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, ".git_head")) as f:
            return f.read().strip() + " (.git_head)"
    except Exception:
        return None


def library_source_hash():
    """content hash of the sources the loaded libmbls_hip.so was built from (milagro_bls_amd/build.py writes it next to the library)"""
    try:
        with open(N.LIB_PATH + ".srchash") as f:
            return f.read().strip()
    except Exception:
        return None


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the PMC passes (scripts/profile_round.sh, separate FETCH_SIZE / WRITE_SIZE passes, corrected as the guide
    prescribes) -- taken into roofline.traffic only when that profile was collected on THIS build (same source hash), else (None, reason)"""
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(p) as f:
            d = json.load(f)
    except Exception:
        return None, "no profiles/hbm_traffic.json"
    have, want = library_source_hash(), d.get("_source_hash")
    if not want:
        return None, "profiles/hbm_traffic.json carries no source hash (collected before round 5)"
    if have != want:
        return None, "profiles/hbm_traffic.json was collected on another build (source hash %s..., this library %s...)" % (want[:12], (have or "?")[:12])
    v = d.get(kernel)
    return (v, "PMC passes of this build (%s)" % d.get("_collected", "scripts/profile_round.sh")) if v is not None else (None, "kernel not in the profile")


def from_profile(fname, key):
    """a figure that was NOT measured by this run: read from a committed profile, labelled with its source"""
    p = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(p):
        return None
    try:
        with open(p) as f:
            d = json.load(f)
        v = d.get(key)
        if v is None:
            return None
        return {"value": v, "source": "profiles/%s (%s)" % (fname, d.get("_collected", "collected in an earlier run")), "measured_by_this_run": False}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--items", type=int, default=1 << 16, help="items per GPU per step (default 2^16, BASELINE configs[2]; 131072 = the config-5 shard)")
    ap.add_argument("--keys", type=int, default=128)
    ap.add_argument("--pk-format", choices=["uncompressed", "compressed", "indexed"], default="uncompressed",
                    help="how the timed path receives the keys: 96-byte decoded form (default: what the reference's fast_aggregate_verify takes), "
                         "48-byte wire form, or indices into a resident key table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the compressed-key and indexed-key legs")
    ap.add_argument("--multi-leg", type=int, default=0, help=argparse.SUPPRESS)      # internal: run ONLY the in-process multi-device leg over this many devices, print it
    ap.add_argument("--stub", action="store_true", help="CPU test of the multi-rank plumbing only: gloo backend and a stand-in verifier; the line it prints is "
                                                         "labelled a stub and is not a measurement")
    args = ap.parse_args()
    protect_stdout()

    if args.multi_leg:
        sys.exit(multi_leg_child(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process only starts the ranks (before anything here has touched a GPU) and relays rank 0's line
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.stub))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(5)
    if args.stub:
        sys.exit(stub_rank(args, rank, world))
    # MBLS_BENCH_SHARE_GPU=1 (test mode, labelled in the line): every rank on device 0 and the gathers through gloo -- RCCL wants one rank per device, and the
    # builder's GPU box has one; this runs the real verifier through the whole multi-rank path (per-rank inputs, gather, checks, reductions) without RCCL
    share_gpu = world > 1 and os.environ.get("MBLS_BENCH_SHARE_GPU") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    ctx = N.Context(dev.index)
    lib = N.lib()
    n, k = args.items, args.keys
    words = (n + 63) // 64
    stream = torch.cuda.current_stream(dev)
    sptr = C.c_void_p(stream.cuda_stream)

    t_in = time.perf_counter()
    d_sigs, d_msgs, d_pks, expect, d_idx, table = build_inputs(ctx, dev, n, k, N.PK_UNCOMPRESSED, rank, return_indices=True)
    t_in = time.perf_counter() - t_in
    ctx.reserve(n)

    def make_step(kind, d_res, d_bm, d_all, d_pk_c=None):
        if kind == "indexed":
            def verify():
                ctx.check(lib.mbls_fast_aggregate_verify_batch_indexed_device(ctx.handle, table.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, d_idx.data_ptr(),
                                                                              None, n, k, d_res.data_ptr(), d_bm.data_ptr(), None, sptr))
        else:
            fmt = N.PK_UNCOMPRESSED if kind == "uncompressed" else N.PK_COMPRESSED
            pk = d_pks if kind == "uncompressed" else d_pk_c

            def verify():
                ctx.check(lib.mbls_fast_aggregate_verify_batch_device(ctx.handle, d_sigs.data_ptr(), d_msgs.data_ptr(), 32, None, pk.data_ptr(), fmt, None,
                                                                      n, k, d_res.data_ptr(), d_bm.data_ptr(), None, sptr))

        def step():
            verify()
            if world > 1:
                shard.all_gather_bitmap(d_bm, world, out=d_all)  # the only collective: RCCL gather of the accept bitmap
        return step, verify

    def compressed_keys():
        flat = d_pks.reshape(-1, 96)
        out = torch.empty((flat.shape[0], 48), dtype=torch.uint8, device=dev)
        # compress on the device in slices through the host-buffer helper (input generation, untimed)
        sl = 1 << 18
        for c0 in range(0, flat.shape[0], sl):
            part = flat[c0:c0 + sl].cpu().numpy()
            comp = np.zeros((part.shape[0], 48), dtype=np.uint8); errs = np.zeros(part.shape[0], dtype=np.uint8)
            ctx.check(lib.mbls_pk_compress_batch(ctx.handle, part.ctypes.data_as(C.c_void_p), part.shape[0], comp.ctypes.data_as(C.c_void_p), errs.ctypes.data_as(C.c_void_p)))
            assert not errs.any()
            out[c0:c0 + sl] = torch.from_numpy(comp).to(dev)
        return out.reshape(n, k, 48).contiguous()

    own_ms = {}

    def run_leg(kind, steps, warmup):
        d_res = torch.zeros(n, dtype=torch.uint8, device=dev)
        d_bm = torch.zeros(words, dtype=torch.int64, device=dev)
        d_all = torch.zeros(words * world, dtype=torch.int64, device=dev) if world > 1 else None
        d_pk_c = None
        if kind == "compressed":
            d_pk_c = compressed_keys()
            ctx.check(lib.mbls_ctx_reserve_keys(ctx.handle, n * k))
        step, verify = make_step(kind, d_res, d_bm, d_all, d_pk_c)
        lib.mbls_enable_phase_timing(ctx.handle, 0)
        elapsed = timed_steps(step, steps, warmup, world, torch.cuda.synchronize)
        own_ms[kind] = LAST_OWN_ELAPSED[0] / steps * 1e3
        ok = check_bitmap(d_res, d_bm, expect)
        if world > 1:
            ok = ok and check_gathered(d_all, world, words, d_bm, rank)
        ok = reduce_all_ok(ok, world)
        # per-kernel timing with HIP events on the launch stream (separate, untimed pass): 2 discarded warm-ups (the events make every
        # call synchronise, which lets the clocks settle differently from the free-running timed loop), then the median of 5
        lib.mbls_enable_phase_timing(ctx.handle, 1)
        rows = []
        for it in range(2 + PHASE_REPS):
            verify()
            ms = (C.c_float * N.N_PHASES)()
            lib.mbls_last_phase_ms(ctx.handle, ms)
            if it >= 2:
                rows.append(list(ms))
        phase = np.median(np.array(rows, dtype=np.float64), axis=0)
        lib.mbls_enable_phase_timing(ctx.handle, 0)
        return elapsed, ok, {nm: float(v) for nm, v in zip(N.PHASE_NAMES, phase)}

    elapsed, ok, phase_ms = run_leg(args.pk_format, args.steps, args.warmup)
    variants = {}
    if not args.no_variants and world == 1:
        for kind in ("compressed", "indexed", "uncompressed"):
            if kind == args.pk_format:
                continue
            e2, ok2, ph2 = run_leg(kind, max(2, min(3, args.steps)), 1)
            st2 = max(2, min(3, args.steps))
            pkb2 = {"compressed": 48 * k, "uncompressed": 96 * k, "indexed": 4 * k}[kind]
            variants[kind] = {"value": n * st2 / e2, "unit": "fast_aggregate_verify/s", "ms_per_step": e2 / st2 * 1e3, "bitmap_matches_expectation": ok2,
                              "algorithmic_bytes_per_item": pkb2 + 96 + 32 + 1, "phase_ms": ph2}
            ok = ok and ok2

    per_rank_ms = gather_per_rank(own_ms[args.pk_format], world)
    coll = None
    if world > 1:
        _bm = torch.zeros(words, dtype=torch.int64, device=dev); _all = torch.zeros(words * world, dtype=torch.int64, device=dev)
        coll = collective_figure(lambda: shard.all_gather_bitmap(_bm, world, out=_all), world, torch.cuda.synchronize)
    # one step = the kernels of one in-order stream (+ the gather): their event-timed sum against this rank's own step time, on every rank
    phase_sum_own = sum(phase_ms.values())
    phase_gap = abs(phase_sum_own - own_ms[args.pk_format]) / own_ms[args.pk_format]
    worst_gap = reduce_max(phase_gap, world)
    other = {}
    multi_legs = None
    if not args.no_variants and world == 1:
        other = other_configs(ctx, lib, dev, sptr)
        other["aggregate_verify_batch 2^14 items x 4 (message, key) pairs"] = aggregate_verify_leg(ctx, lib, dev, sptr)
        other["operations either side of the path (2^16 each)"] = keyops_figures(ctx, lib, dev)
        other["the reference's criterion shapes (benches/bls381_benches.rs:104-112, 179-245)"] = criterion_shapes()
        ok = ok and all(v.get("correct", True) for v in other.values() if isinstance(v, dict))
        ok = ok and all(v.get("correct", True) for v in other["operations either side of the path (2^16 each)"].values() if isinstance(v, dict))
        # the in-process multi-GPU entry over 1, 2, 4, 8 of the visible devices (one process, host buffers: PCIe included)
        have = torch.cuda.device_count()
        multi_legs = []
        for G in (1, 2, 4, 8):
            if G > have:
                break
            # one device: here. More than one (RCCL between the devices of ONE process: never met hardware in the builder's runs): in a child process with a
            # time limit, so that a hang or a crash there costs this leg and not the line; a leg that RAN and disagrees with the expectation still fails the run
            leg = multi_handle_leg(lib, table, d_sigs, d_msgs, d_idx, expect, n, k, [0]) if G == 1 else multi_leg_in_child(G, n, k, timeout_s=240)
            # a leg that timed out, crashed or could not start carries 'error' and no 'results_match': it did NOT run -- said so per leg and in the line
            # ('multi_legs_ok' below), never counted as a pass
            leg["ran"] = "results_match" in leg
            multi_legs.append(leg)
            if leg["ran"]:
                ok = ok and bool(leg["results_match"])
            else:
                break                                      # a leg that could not run (time limit, crash): the larger ones are not tried -- the line must not wait for them
    vm_leg = None
    if n == (1 << 16) and not args.no_variants:
        vm_leg = sharded_verify_multiple_leg(ctx, lib, dev, sptr, rank, world, k)
        ok = ok and vm_leg["correct"]
    shard_leg = None
    if world == 8 and n == (1 << 16):
        # BASELINE configs[4] as it is named: 2^20 items over 8 GPUs = 2^17 per rank (the default legs above are configs[2] per GPU)
        shard_leg = config5_leg(ctx, lib, dev, sptr, rank, world, k)

    if rank == 0:
        pkb = {"compressed": 48 * k, "uncompressed": 96 * k, "indexed": 4 * k}[args.pk_format]
        bytes_per_item = pkb + 96 + 32 + 1          # algorithmic bytes in + 1 out
        ms_per_step = elapsed / args.steps * 1e3
        value = n * world * args.steps / elapsed
        dom = max((nm for nm in phase_ms), key=lambda nm: phase_ms[nm])
        achieved = n * bytes_per_item / (phase_ms[dom] * 1e-3) / 1e9          # algorithmic GB/s through the dominant kernel
        cfg_name = ("configs[2]" if world == 1 else "configs[2] on every GPU (weak scaling: %d x 2^16 items)" % world) if n == (1 << 16) else (
            "the configs[4] shard (2^20 items / 8 GPUs)" if n == (1 << 17) else "custom size")
        phase_sum = sum(phase_ms.values())
        phase_pass = "consistent" if worst_gap <= 0.02 + (0.02 if world > 1 else 0.0) else "inconsistent"
        traffic, traffic_note = measured_traffic(dom)
        out = {
            "metric": "fast_aggregate_verify/sec (128 pubkeys, 32B msg)", "value": value, "unit": "fast_aggregate_verify/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "ms_per_step_per_rank": per_rank_ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (12 x 32-bit Montgomery limbs; 14 x 28-bit digits with 64-bit column sums inside the multiplication routines)", "data": "synthetic",
            "config": {"workload": "%s: batch of %d fast_aggregate_verify, %d pubkeys each, per GPU" % (cfg_name, n, k),
                       "items_per_gpu": n, "keys_per_item": k, "msg_bytes": 32, "pk_format": args.pk_format,
                       "negatives": "every 16th item corrupted (msg bit / wrong key / sig not in G2 / infinity sig / apk = infinity)",
                       "parallelism": "items sharded over %d GPU(s), RCCL all-gather of the accept bitmap" % world},
            "bitmap_matches_expectation": ok,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note, "kernel": "k_" + dom, "kernel_ms": phase_ms[dom],
                         "algorithmic_bytes_per_item": bytes_per_item,
                         "note": "VALU-issue-bound path (see valu_issue): the HBM fraction is reported as measured; traffic = HBM bytes per launch of the dominant "
                                 "kernel from the PMC counters (separate passes, never inside a timed run), attached only when collected on this very build"},
            "traffic_from_profile": from_profile("hbm_traffic.json", dom),
            "valu_issue": valu_issue_figure(lib, ctx, phase_ms, n, {"uncompressed": "k_aggregate", "indexed": "k_aggregate_indexed"}.get(args.pk_format) if k == 128 else None),
            "phase_ms": phase_ms,
            "phase_notes": "sig = signature decoding: the subgroup test psi(sig) = [x] sig is read off the Miller loop's running point (k_sig_verdict, counted under miller)",
            "phase_pass": {"verdict": phase_pass, "sum_phase_ms": phase_sum, "ms_per_step": ms_per_step, "worst_rank_gap": worst_gap,
                           "method": "median of %d event-timed calls after 2 discarded ones; 'consistent' = on EVERY rank the kernels of its in-order stream sum to that rank's own "
                                     "timed step within 2 %% (4 %% with the bitmap gather inside the step, world > 1); worst_rank_gap = the largest relative gap" % PHASE_REPS},
            "variants": variants,
            "other_configs": other,
            "multi_handle_leg": multi_legs,
            # true only when every leg over the visible devices RAN and matched; false when one ran and disagreed (exit code 3) or produced no result at all
            "multi_legs_ok": (all(l.get("ran") and l.get("results_match") for l in multi_legs) if multi_legs is not None else None),
            "configs4_shard_leg": shard_leg,
            "collective": coll,
            "sharded_verify_multiple_leg": vm_leg,
            "input_build_s": t_in,
            "head": git_head(), "library_source_hash": library_source_hash(),
        }
        if share_gpu:
            out["share_gpu_test"] = "all %d ranks on ONE device, gathers through gloo: a functional test of the multi-rank path, not a scaling measurement" % world
        if not args.no_cpu_baseline and world == 1:
            allc, one, cfg1 = cpu_baseline(d_sigs, d_msgs, d_pks, expect, k, N.PK_UNCOMPRESSED)
            out["cpu_baseline"] = allc
            out["cpu_baseline_1t"] = one
            out["cpu_config1"] = cfg1
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        emit_result(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
