"""Batch entry points (host buffers in, results out) and device-pointer entry points for callers that keep
their inputs resident in HBM (bench.py uses torch tensors only as device memory + stream plumbing)."""
import ctypes as C

from . import _native as N


def _c():
    return N.default_context()


def _moff(msg_offsets):
    """n + 1 byte offsets into the message buffer (messages of any length each), or None for msg_len bytes per item"""
    return None if msg_offsets is None else (C.c_uint64 * len(msg_offsets))(*msg_offsets)


def fast_aggregate_verify_batch(sigs, msgs, pks, n, k=None, pk_format=N.PK_COMPRESSED, msg_len=32, pk_offsets=None, ctx=None, msg_offsets=None):
    """n x AggregateSignature::fast_aggregate_verify (reference src/aggregates.rs:177-215).
    Returns (results: list[bool], status: list[int])."""
    ctx = ctx or _c()
    res = N.outbuf(n)
    st = (C.c_uint32 * max(1, n))()
    off = None
    if pk_offsets is not None:
        off = (C.c_uint32 * len(pk_offsets))(*pk_offsets)
        k = 0
    ctx.check(N.lib().mbls_fast_aggregate_verify_batch(ctx.handle, N.cbuf(sigs), N.cbuf(msgs), msg_len, _moff(msg_offsets), N.cbuf(pks), pk_format,
                                                       off, n, k, res, st))
    return [bool(x) for x in bytes(res)[:n]], list(st)[:n]


def fast_aggregate_verify_batch_indexed(table, sigs, msgs, key_idx, n, k=None, msg_len=32, offsets=None, ctx=None, msg_offsets=None):
    """The same over a resident key table: item i uses table entries key_idx[k*i : k*i+k] (or key_idx[offsets[i]:offsets[i+1]])."""
    ctx = ctx or table.ctx
    res = N.outbuf(n)
    st = (C.c_uint32 * max(1, n))()
    idx = (C.c_uint32 * max(1, len(key_idx)))(*key_idx)
    off = None
    if offsets is not None:
        off = (C.c_uint32 * len(offsets))(*offsets)
        k = 0
    ctx.check(N.lib().mbls_fast_aggregate_verify_batch_indexed(ctx.handle, table.handle, N.cbuf(sigs), N.cbuf(msgs), msg_len, _moff(msg_offsets), idx, off, n, k, res, st))
    return [bool(x) for x in bytes(res)[:n]], list(st)[:n]


def aggregate_signatures_batch(sigs96, n, k=None, offsets=None, ctx=None):
    """n x AggregateSignature::aggregate (reference src/aggregates.rs:100-106) -> (compressed sums, errs)"""
    ctx = ctx or _c()
    out, errs = N.outbuf(96 * n), N.outbuf(n)
    off = None
    if offsets is not None:
        off = (C.c_uint32 * len(offsets))(*offsets)
        k = 0
    ctx.check(N.lib().mbls_aggregate_signatures_batch(ctx.handle, N.cbuf(sigs96), off, n, k, out, errs))
    return bytes(out)[:96 * n], list(bytes(errs)[:n])


def aggregate_verify_batch(sigs, msgs, pks96, n, k=None, pair_offsets=None, msg_len=32, msg_offsets=None, ctx=None):
    """n x AggregateSignature::aggregate_verify (reference src/aggregates.rs:130-170): the (message, key) pairs of all items back to back;
    item i owns pairs [pair_offsets[i], pair_offsets[i+1]) or k each. Returns (results, status)."""
    ctx = ctx or _c()
    res = N.outbuf(n)
    st = (C.c_uint32 * max(1, n))()
    off = None
    if (k is None) == (pair_offsets is None):
        raise ValueError("aggregate_verify_batch: give exactly one of k (pairs per item) and pair_offsets (n + 1 entries)")
    if pair_offsets is not None:
        if len(pair_offsets) != n + 1:
            raise ValueError("aggregate_verify_batch: pair_offsets must have n + 1 = %d entries, got %d" % (n + 1, len(pair_offsets)))
        if pair_offsets and not 0 <= pair_offsets[-1] <= 0xFFFFFFFF:
            raise ValueError("aggregate_verify_batch: pair indices are 32-bit")
        off = (C.c_uint32 * len(pair_offsets))(*pair_offsets)
        k = 0
    elif not 0 <= k * n <= 0xFFFFFFFF:
        raise ValueError("aggregate_verify_batch: pair indices are 32-bit")
    ctx.check(N.lib().mbls_aggregate_verify_batch(ctx.handle, N.cbuf(sigs), N.cbuf(msgs), msg_len, _moff(msg_offsets), N.cbuf(pks96), off, k, n, res, st))
    return [bool(x) for x in bytes(res)[:n]], list(st)[:n]


def verify_batch(sigs, msgs, pks, n, pk_format=N.PK_COMPRESSED, msg_len=32, ctx=None, msg_offsets=None):
    """n x Signature::verify (reference src/signature.rs:27-40)."""
    ctx = ctx or _c()
    res = N.outbuf(n)
    st = (C.c_uint32 * max(1, n))()
    ctx.check(N.lib().mbls_verify_batch(ctx.handle, N.cbuf(sigs), N.cbuf(msgs), msg_len, _moff(msg_offsets), N.cbuf(pks), pk_format, n, res, st))
    return [bool(x) for x in bytes(res)[:n]], list(st)[:n]


def pk_decode_batch(data, n, in_format=N.PK_COMPRESSED, validate=True, ctx=None):
    ctx = ctx or _c()
    out, errs = N.outbuf(96 * n), N.outbuf(n)
    ctx.check(N.lib().mbls_pk_decode_batch(ctx.handle, N.cbuf(data), in_format, int(validate), n, out, errs))
    return bytes(out)[:96 * n], list(bytes(errs)[:n])


def pk_compress_batch(data96, n, ctx=None):
    ctx = ctx or _c()
    out, errs = N.outbuf(48 * n), N.outbuf(n)
    ctx.check(N.lib().mbls_pk_compress_batch(ctx.handle, N.cbuf(data96), n, out, errs))
    return bytes(out)[:48 * n], list(bytes(errs)[:n])


def sig_check_batch(data96, n, ctx=None):
    ctx = ctx or _c()
    errs, g2 = N.outbuf(n), N.outbuf(n)
    ctx.check(N.lib().mbls_sig_check_batch(ctx.handle, N.cbuf(data96), n, errs, g2))
    return list(bytes(errs)[:n]), [bool(x) for x in bytes(g2)[:n]]


def sign_batch(sks32, msgs, n, msg_len=32, ctx=None):
    ctx = ctx or _c()
    out = N.outbuf(96 * n)
    ctx.check(N.lib().mbls_sign_batch(ctx.handle, N.cbuf(sks32), N.cbuf(msgs), msg_len, n, out))
    return bytes(out)[:96 * n]


def sk_to_pk_batch(sks32, n, out_format=N.PK_COMPRESSED, ctx=None):
    ctx = ctx or _c()
    sz = 48 if out_format == N.PK_COMPRESSED else 96
    out = N.outbuf(sz * n)
    ctx.check(N.lib().mbls_sk_to_pk_batch(ctx.handle, N.cbuf(sks32), out_format, n, out))
    return bytes(out)[:sz * n]


def hash_to_g2_batch(msgs, n, msg_len=32, ctx=None, mode=0):
    """n x hash_to_curve_g2, compressed; mode 0: the stand-alone lane body, 1 / 2: the verification pipeline's message phase with one lane /
    one wave per item (include/mbls.h, mbls_hash_to_g2_batch_mode)"""
    ctx = ctx or _c()
    out = N.outbuf(96 * n)
    ctx.check(N.lib().mbls_hash_to_g2_batch_mode(ctx.handle, N.cbuf(msgs), msg_len, n, out, mode))
    return bytes(out)[:96 * n]


def aggregate_public_keys_batch(pks, n, k=None, pk_format=N.PK_COMPRESSED, pk_offsets=None, ctx=None):
    ctx = ctx or _c()
    out = N.outbuf(96 * n)
    st = (C.c_uint32 * max(1, n))()
    off = None
    if pk_offsets is not None:
        off = (C.c_uint32 * len(pk_offsets))(*pk_offsets)
        k = 0
    ctx.check(N.lib().mbls_aggregate_public_keys_batch(ctx.handle, N.cbuf(pks), pk_format, off, n, k, out, st))
    return bytes(out)[:96 * n], list(st)[:n]


def fp_mul_batch(a48, b48, n, square=False, ctx=None, op=None):
    """field probe; op as in include/mbls.h (default: 1 if square else 0)"""
    ctx = ctx or _c()
    out = N.outbuf(48 * n)
    ctx.check(N.lib().mbls_fp_mul_batch(ctx.handle, N.cbuf(a48), N.cbuf(b48), n, out, int(square) if op is None else op))
    return bytes(out)[:48 * n]


def fp_mul_bench(n_lanes, iters, ctx=None):
    """Integer-ALU calibration: `iters` dependent Fp multiplications on each of n_lanes lanes -> elapsed ms."""
    ctx = ctx or _c()
    ms = C.c_float()
    ctx.check(N.lib().mbls_fp_mul_bench(ctx.handle, n_lanes, iters, C.byref(ms)))
    return ms.value


def verify_multiple_sets_device(d_sigs, d_pks, d_msgs, d_rands, n, k, pk_format=N.PK_COMPRESSED, msg_len=32, stream=None, ctx=None, d_result=None, d_status=None):
    """AggregateSignature::verify_multiple_aggregate_signatures (reference src/aggregates.rs:261-316) over n sets given by
    their k wire-format keys each, everything resident on the device (raw device pointers / ints). The C entry only enqueues: the bool
    arrives in the device byte d_result (and the OR of the sets' status bits in the device word d_status). Without d_result this wrapper
    provides both (torch tensors), synchronises and returns the bool."""
    ctx = ctx or _c()
    if d_result is not None:
        ctx.check(N.lib().mbls_verify_multiple_sets_device(ctx.handle, d_sigs, d_pks, pk_format, None, k, d_msgs, msg_len, None, d_rands, n,
                                                           d_result, d_status, stream))
        return None
    import torch
    res = torch.full((8,), 7, dtype=torch.uint8, device="cuda")
    ctx.check(N.lib().mbls_verify_multiple_sets_device(ctx.handle, d_sigs, d_pks, pk_format, None, k, d_msgs, msg_len, None, d_rands, n,
                                                       res.data_ptr(), None, stream))
    torch.cuda.synchronize()
    v = int(res[0].item())
    assert v in (0, 1)
    return bool(v)


def verify_multiple_sets_indexed_device(table, d_sigs, d_key_idx, d_msgs, d_rands, n, k, msg_len=32, stream=None, ctx=None, d_result=None, d_status=None, d_partial=None,
                                        d_offsets=None):
    """verify_multiple over sets named by indices into a resident KeyTable (raw device pointers). With d_result or d_partial the call only enqueues;
    without both it synchronises and returns the bool."""
    ctx = ctx or _c()
    f = N.lib().mbls_verify_multiple_sets_indexed_device
    if d_result is not None or d_partial is not None:
        ctx.check(f(ctx.handle, table.handle, d_sigs, d_key_idx, d_offsets, k, d_msgs, msg_len, None, d_rands, n, d_result, d_status, d_partial, stream))
        return None
    import torch
    res = torch.full((8,), 7, dtype=torch.uint8, device="cuda")
    ctx.check(f(ctx.handle, table.handle, d_sigs, d_key_idx, d_offsets, k, d_msgs, msg_len, None, d_rands, n, res.data_ptr(), d_status, None, stream))
    torch.cuda.synchronize()
    v = int(res[0].item())
    assert v in (0, 1)
    return bool(v)


def verify_multiple_partial_device(d_sigs, d_msgs, d_rands, n, d_partial, d_apks=None, d_pks=None, k=0, pk_format=N.PK_COMPRESSED, msg_len=32, stream=None, ctx=None):
    """One shard of a verify_multiple that is spread over several devices or processes (include/mbls.h): the shard's
    Miller product, signature sum and status bits as one N.VM_PARTIAL_BYTES record at the device address d_partial. Enqueues only."""
    ctx = ctx or _c()
    ctx.check(N.lib().mbls_verify_multiple_partial_device(ctx.handle, d_sigs, d_apks, d_pks, pk_format, None, k, d_msgs, msg_len, None, d_rands, n, d_partial, stream))


def verify_multiple_finish_device(d_partials, n_partials, d_result=None, d_status=None, stream=None, ctx=None):
    """The records of all shards (n_partials x N.VM_PARTIAL_BYTES at d_partials, the same order on every participant) -> the bool of the
    one-device call over the concatenated sets. With d_result the call only enqueues; without, it synchronises and returns the bool."""
    ctx = ctx or _c()
    if d_result is not None:
        ctx.check(N.lib().mbls_verify_multiple_finish_device(ctx.handle, d_partials, n_partials, d_result, d_status, stream))
        return None
    import torch
    res = torch.full((8,), 7, dtype=torch.uint8, device="cuda")
    ctx.check(N.lib().mbls_verify_multiple_finish_device(ctx.handle, d_partials, n_partials, res.data_ptr(), d_status, stream))
    torch.cuda.synchronize()
    v = int(res[0].item())
    assert v in (0, 1)
    return bool(v)


def multi_verify_multiple_aggregate_signatures(mctx, sigs, apks, msgs, rands, n, msg_len=32, msg_offsets=None):
    """verify_multiple_aggregate_signatures (reference src/aggregates.rs:261-316) sharded over the devices of a MultiContext: host buffers
    (sigs 96 B each, decoded aggregate keys 96 B each, messages, 64-bit scalars) -> bool"""
    r = (C.c_uint64 * max(1, n))(*rands)
    return bool(N.lib().mbls_multi_verify_multiple_aggregate_signatures(mctx.handle, N.cbuf(sigs), N.cbuf(apks), N.cbuf(msgs), msg_len, _moff(msg_offsets), r, n))


def multi_fast_aggregate_verify_batch(mctx, sigs, msgs, pks, n, k=None, pk_format=N.PK_COMPRESSED, msg_len=32, pk_offsets=None, msg_offsets=None):
    """fast_aggregate_verify_batch sharded over the devices of a MultiContext (include/mbls.h, mbls_multi_*)"""
    res = N.outbuf(n)
    st = (C.c_uint32 * max(1, n))()
    off = None
    if pk_offsets is not None:
        off = (C.c_uint32 * len(pk_offsets))(*pk_offsets)
        k = 0
    mctx.check(N.lib().mbls_multi_fast_aggregate_verify_batch(mctx.handle, N.cbuf(sigs), N.cbuf(msgs), msg_len, _moff(msg_offsets), N.cbuf(pks), pk_format,
                                                              off, n, k, res, st))
    return [bool(x) for x in bytes(res)[:n]], list(st)[:n]


def multi_fast_aggregate_verify_bitmap(mctx, sigs, msgs, pks, n, k=None, pk_format=N.PK_COMPRESSED, msg_len=32, pk_offsets=None, msg_offsets=None):
    """The same with the results as ONE packed accept bitmap that every device of the handle ends up holding (all-gather between the devices: RCCL when
    mctx.rccl_active, host memory otherwise). -> (bits, status, words): bits[i] = item i accepted, words = the ceil(n / 64) bitmap words of the first device."""
    words = (C.c_uint64 * max(1, (n + 63) // 64))()
    st = (C.c_uint32 * max(1, n))()
    off = None
    if pk_offsets is not None:
        off = (C.c_uint32 * len(pk_offsets))(*pk_offsets)
        k = 0
    mctx.check(N.lib().mbls_multi_fast_aggregate_verify_bitmap(mctx.handle, N.cbuf(sigs), N.cbuf(msgs), msg_len, _moff(msg_offsets), N.cbuf(pks), pk_format,
                                                               off, n, k, words, st))
    w = list(words)[:(n + 63) // 64]
    return [bool((w[i // 64] >> (i % 64)) & 1) for i in range(n)], list(st)[:n], w


def multi_verify_batch(mctx, sigs, msgs, pks, n, pk_format=N.PK_COMPRESSED, msg_len=32, msg_offsets=None):
    res = N.outbuf(n)
    st = (C.c_uint32 * max(1, n))()
    mctx.check(N.lib().mbls_multi_verify_batch(mctx.handle, N.cbuf(sigs), N.cbuf(msgs), msg_len, _moff(msg_offsets), N.cbuf(pks), pk_format, n, res, st))
    return [bool(x) for x in bytes(res)[:n]], list(st)[:n]


def multi_fast_aggregate_verify_batch_indexed(mctx, mtable, sigs, msgs, key_idx, n, k=None, msg_len=32, offsets=None, msg_offsets=None):
    res = N.outbuf(n)
    st = (C.c_uint32 * max(1, n))()
    idx = (C.c_uint32 * max(1, len(key_idx)))(*key_idx)
    off = None
    if offsets is not None:
        off = (C.c_uint32 * len(offsets))(*offsets)
        k = 0
    mctx.check(N.lib().mbls_multi_fast_aggregate_verify_batch_indexed(mctx.handle, mtable.handle, N.cbuf(sigs), N.cbuf(msgs), msg_len, _moff(msg_offsets),
                                                                      idx, off, n, k, res, st))
    return [bool(x) for x in bytes(res)[:n]], list(st)[:n]
