"""Shared test helpers: golden vectors, the host emulator of the lane bodies, synthetic batches (built with the
oracle, which is the checker here -- never the thing under test)."""
import ctypes as C
import json
import os
import random
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "vectors.json")
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
G2_INF = bytes([0xC0]) + bytes(95)


def load_vectors():
    with open(GOLDEN) as f:
        return json.load(f)


def expand_msg(h):
    return bytes([42]) * 133700 if h == "2a*133700" else bytes.fromhex(h)


def load_emulator():
    """Build (amdclang++, plain C++) and load tests/host_emul/libmbls_emul.so."""
    d = os.path.join(ROOT, "tests", "host_emul")
    so = os.path.join(d, "libmbls_emul.so")
    src = [os.path.join(d, "mbls_emul.cpp")] + [os.path.join(ROOT, "milagro_bls_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "milagro_bls_amd", "csrc")) if f.endswith((".h", ".inc"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        cxx = os.environ.get("MBLS_EMUL_CXX", "/opt/rocm/lib/llvm/bin/clang++")
        subprocess.check_call([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src[0]])
    return C.CDLL(so)


def cb(x):
    x = bytes(x)
    return (C.c_uint8 * max(1, len(x))).from_buffer_copy(x if x else b"\0")


def ob(n):
    return (C.c_uint8 * max(1, n))()


class Batch:
    pass


NOT_IN_G2 = None


def make_batch(n, k, fmt=0, seed=1, pool_n=64, negatives=True, nthreads=8):
    """Synthetic fast_aggregate_verify batch in wire format + the expectation by construction.
    Negative kinds cycle over items with i % 4 == 3."""
    import orc
    rnd = random.Random(seed)
    pool = [rnd.randrange(1, R) for _ in range(pool_n)]
    sz = 48 if fmt == 0 else 96
    pkb = orc.batch_sk_to_pk(b"".join(s.to_bytes(32, "big") for s in pool), pool_n, fmt, nthreads=nthreads)
    pk = [pkb[sz * j:sz * j + sz] for j in range(pool_n)]
    msgs = [rnd.randbytes(32) for _ in range(n)]
    idxs = [rnd.sample(range(pool_n), k) for _ in range(n)]
    aggs = [sum(pool[j] for j in idx) % R for idx in idxs]
    sigs = orc.batch_sign(b"".join(a.to_bytes(32, "big") for a in aggs), b"".join(msgs), n, nthreads=nthreads)
    sigs = [sigs[96 * i:96 * i + 96] for i in range(n)]
    keys = [[pk[j] for j in idx] for idx in idxs]
    expect = [True] * n
    kinds = ["valid"] * n
    if negatives:
        order = ["flip_msg", "wrong_key", "sig_not_in_g2", "sig_infinity", "apk_infinity", "bad_sig_bytes", "bad_pk_bytes"]
        c = 0
        for i in range(n):
            if i % 4 != 3:
                continue
            kind = order[c % len(order)]; c += 1
            if kind == "apk_infinity" and k < 2:
                kind = "flip_msg"
            kinds[i] = kind; expect[i] = False
            if kind == "flip_msg":
                msgs[i] = bytes([msgs[i][0] ^ 1]) + msgs[i][1:]
            elif kind == "wrong_key":
                other = [j for j in range(pool_n) if j not in idxs[i]][0]
                keys[i][0] = pk[other]
            elif kind == "sig_not_in_g2":
                sigs[i] = bytes.fromhex(load_vectors()["model"]["g2_subgroup_probes"][c % 3]["compressed"])
            elif kind == "sig_infinity":
                sigs[i] = G2_INF
            elif kind == "apk_infinity":
                e, partial = orc.aggregate_pks([orc.sk_to_pk(pool[j]) for j in idxs[i][:-1]])
                neg = orc.g1_mul(partial, R - 1)
                keys[i][-1] = orc.g1_compress(neg) if fmt == 0 else neg
            elif kind == "bad_sig_bytes":
                sigs[i] = bytes([sigs[i][0] & 0x7F]) + sigs[i][1:]      # compression flag cleared
            elif kind == "bad_pk_bytes":
                keys[i][0] = bytes([0x80 if fmt == 0 else 0x00]) + b"\xff" * (sz - 1)   # x >= p
    b = Batch()
    b.n, b.k, b.fmt = n, k, fmt
    b.sigs = b"".join(sigs); b.msgs = b"".join(msgs); b.pks = b"".join(b"".join(ks) for ks in keys)
    b.expect = expect; b.kinds = kinds
    return b
