"""ctypes binding of the CPU oracle (oracle/bls_oracle.c). TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
the product package milagro_bls_amd/."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libbls_oracle.so")

OK, ERR_G1_SIZE, ERR_G2_SIZE, ERR_POINT, ERR_EMPTY = 0, 1, 2, 3, 4
PK_COMPRESSED, PK_UNCOMPRESSED = 0, 1

_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("bls_oracle.c", "bls_oracle.h", "orc_constants.h")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_init()
    return _lib


def _buf(n):
    return (C.c_uint8 * n)()


def _b(x):
    return (C.c_uint8 * len(x)).from_buffer_copy(bytes(x)) if len(x) else (C.c_uint8 * 1)()


def fp_mul(a, b):
    o = _buf(48); lib().orc_fp_mul(_b(a), _b(b), o); return bytes(o)


def fp_inv(a):
    o = _buf(48); lib().orc_fp_inv(_b(a), o); return bytes(o)


def fp_sqrt(a):
    o = _buf(48); ok = lib().orc_fp_sqrt(_b(a), o); return bytes(o) if ok else None


def op_counts(reset=True):
    m, s = C.c_uint64(), C.c_uint64()
    lib().orc_op_counts(C.byref(m), C.byref(s), int(reset))
    return m.value, s.value


def g1_from_compressed(b):
    o = _buf(96); e = lib().orc_g1_from_compressed(_b(b), C.c_size_t(len(b)), o); return e, (bytes(o) if e == 0 else None)


def g1_from_uncompressed(b):
    o = _buf(96); e = lib().orc_g1_from_uncompressed(_b(b), C.c_size_t(len(b)), o); return e, (bytes(o) if e == 0 else None)


def pk_from_bytes(b):
    o = _buf(96); e = lib().orc_pk_from_bytes(_b(b), C.c_size_t(len(b)), o); return e, (bytes(o) if e == 0 else None)


def g1_key_validate(pk):
    return bool(lib().orc_g1_key_validate(_b(pk)))


def g1_compress(pk):
    o = _buf(48); e = lib().orc_g1_compress(_b(pk), o); assert e == 0, e; return bytes(o)


def g2_from_compressed(b):
    o = _buf(192); e = lib().orc_g2_from_compressed(_b(b), C.c_size_t(len(b)), o); return e, (bytes(o) if e == 0 else None)


def g2_compress(sig):
    o = _buf(96); e = lib().orc_g2_compress(_b(sig), o); assert e == 0, e; return bytes(o)


def g2_subgroup_check(sig):
    return bool(lib().orc_g2_subgroup_check(_b(sig)))


def g1_add(a, b):
    o = _buf(96); e = lib().orc_g1_add(_b(a), _b(b), o); assert e == 0, e; return bytes(o)


def g2_add(a, b):
    o = _buf(192); e = lib().orc_g2_add(_b(a), _b(b), o); assert e == 0, e; return bytes(o)


def g1_mul(a, k):
    o = _buf(96); e = lib().orc_g1_mul(_b(a), _b(int(k).to_bytes(32, "big")), o); assert e == 0, e; return bytes(o)


def g2_mul(a, k):
    o = _buf(192); e = lib().orc_g2_mul(_b(a), _b(int(k).to_bytes(32, "big")), o); assert e == 0, e; return bytes(o)


def aggregate_pks(pks):
    o = _buf(96); e = lib().orc_aggregate_pks(_b(b"".join(pks)), C.c_size_t(len(pks)), o); return e, (bytes(o) if e == 0 else None)


def sk_to_pk(sk):
    o = _buf(96); lib().orc_sk_to_pk(_b(int(sk).to_bytes(32, "big")), o); return bytes(o)


def hash_to_g2(msg, dst=None):
    o = _buf(192)
    if dst is None:
        lib().orc_hash_to_g2(_b(msg), C.c_size_t(len(msg)), None, C.c_size_t(0), o)
    else:
        lib().orc_hash_to_g2(_b(msg), C.c_size_t(len(msg)), _b(dst), C.c_size_t(len(dst)), o)
    return bytes(o)


def sign(msg, sk):
    o = _buf(192); lib().orc_sign(_b(msg), C.c_size_t(len(msg)), _b(int(sk).to_bytes(32, "big")), o); return bytes(o)


def verify(sig, msg, pk):
    return bool(lib().orc_verify(_b(sig), _b(msg), C.c_size_t(len(msg)), _b(pk)))


def fast_aggregate_verify_pre_aggregated(sig, msg, apk):
    return bool(lib().orc_fast_aggregate_verify_pre_aggregated(_b(sig), _b(msg), C.c_size_t(len(msg)), _b(apk)))


def fast_aggregate_verify(sig, msg, pks):
    return bool(lib().orc_fast_aggregate_verify(_b(sig), _b(msg), C.c_size_t(len(msg)), _b(b"".join(pks)), C.c_size_t(len(pks))))


def aggregate_verify(sig, msgs, pks):
    lens = (C.c_size_t * max(1, len(msgs)))(*[len(m) for m in msgs])
    return bool(lib().orc_aggregate_verify(_b(sig), _b(b"".join(msgs)), lens, C.c_size_t(len(msgs)),
                                           _b(b"".join(pks)), C.c_size_t(len(pks))))


def verify_multiple(sets, rands):
    """sets = [(sig192, apk96, msg)], rands = nonzero 63-bit ints."""
    n = len(sets)
    lens = (C.c_size_t * max(1, n))(*[len(s[2]) for s in sets])
    rr = (C.c_uint64 * max(1, n))(*rands)
    return bool(lib().orc_verify_multiple(_b(b"".join(s[0] for s in sets)), _b(b"".join(s[1] for s in sets)),
                                          _b(b"".join(s[2] for s in sets)), lens, rr, C.c_size_t(n)))


def batch_fast_aggregate_verify(sigs, msgs, pks, n, k, pk_fmt=PK_COMPRESSED, msg_len=32, nthreads=1):
    """Wire-format batch: sigs n*96, msgs n*msg_len, pks n*k*(48|96) -> list of bools."""
    o = _buf(max(1, n))
    lib().orc_batch_fast_aggregate_verify(_b(sigs), _b(msgs), C.c_size_t(msg_len), _b(pks), pk_fmt,
                                          C.c_size_t(n), C.c_size_t(k), o, nthreads)
    return [bool(x) for x in bytes(o)[:n]]


def batch_verify(sigs, msgs, pks, n, msg_len=32, nthreads=1):
    o = _buf(max(1, n))
    lib().orc_batch_verify(_b(sigs), _b(msgs), C.c_size_t(msg_len), _b(pks), C.c_size_t(n), o, nthreads)
    return [bool(x) for x in bytes(o)[:n]]


def batch_sign(sks, msgs, n, msg_len=32, nthreads=1):
    """sks: n*32 big-endian, msgs: n*msg_len -> n*96 compressed signatures."""
    o = _buf(96 * max(1, n))
    lib().orc_batch_sign(_b(sks), _b(msgs), C.c_size_t(msg_len), C.c_size_t(n), o, nthreads)
    return bytes(o)[:96 * n]


def batch_sk_to_pk(sks, n, pk_fmt=PK_COMPRESSED, nthreads=1):
    sz = 48 if pk_fmt == PK_COMPRESSED else 96
    o = _buf(sz * max(1, n))
    lib().orc_batch_sk_to_pk(_b(sks), C.c_size_t(n), pk_fmt, o, nthreads)
    return bytes(o)[:sz * n]


def batch_hash_to_g2(msgs, n, msg_len=32):
    o = _buf(96 * max(1, n))
    lib().orc_batch_hash_to_g2(_b(msgs), C.c_size_t(msg_len), C.c_size_t(n), o)
    return bytes(o)[:96 * n]
