"""Host-side mirror of the reference's public API (reference src/lib.rs:17-22) over the C ABI of
libmbls_hip.so: same type and method names, argument meaning and error behaviour, so that the parity tests
read like the reference's own tests. Every numeric operation runs in the HIP kernels (no CPU arithmetic
here apart from SecretKey's HKDF key derivation, which the reference also does on the host, src/keys.rs:45-77).

    reference                                   here
    ---------                                   ----
    SecretKey / PublicKey / Keypair             src/keys.rs:28-204
    Signature                                   src/signature.rs:9-51
    AggregatePublicKey / AggregateSignature     src/aggregates.rs:17-334
    AmclError                                   amcl::errors::AmclError (src/amcl_utils.rs:11)
"""
import ctypes as C
import hashlib
import hmac
import os

from . import _native as N

G1_BYTES = 48            # reference src/lib.rs:20
G2_BYTES = 96
SECRET_KEY_BYTES = 32
CURVE_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
KEY_SALT = b"BLS-SIG-KEYGEN-SALT-"   # reference src/keys.rs:24
L = 48                                # reference src/keys.rs:26
_G2_INFINITY = bytes([0xC0]) + bytes(95)


class AmclError(Exception):
    """Mirror of the AmclError variants the reference uses."""
    InvalidG1Size = N.ERR_INVALID_G1_SIZE
    InvalidG2Size = N.ERR_INVALID_G2_SIZE
    InvalidPoint = N.ERR_INVALID_POINT
    AggregateEmptyPoints = N.ERR_AGGREGATE_EMPTY_POINTS
    InvalidSecretKeySize = N.ERR_INVALID_SECRET_KEY_SIZE
    InvalidSecretKeyRange = N.ERR_INVALID_SECRET_KEY_RANGE
    _NAMES = {1: "InvalidG1Size", 2: "InvalidG2Size", 3: "InvalidPoint", 4: "AggregateEmptyPoints",
              5: "InvalidSecretKeySize", 6: "InvalidSecretKeyRange"}

    def __init__(self, code):
        super().__init__(self._NAMES.get(code, "DeviceError(%d)" % code))
        self.code = code

    def __eq__(self, other):
        return isinstance(other, AmclError) and other.code == self.code

    def __hash__(self):
        return hash(self.code)


def hkdf_extract(salt, ikm):
    """HKDF-Extract with SHA-256 (RFC 5869 section 2.2; amcl HASH256::hkdf_extract, reference src/keys.rs:62)"""
    return hmac.new(bytes(salt) if salt else bytes(32), bytes(ikm), hashlib.sha256).digest()


def hkdf_expand(prk, info, length):
    """HKDF-Expand with SHA-256 (RFC 5869 section 2.3; amcl HASH256::hkdf_extend, reference src/keys.rs:67)"""
    okm, t, i = b"", b"", 1
    while len(okm) < length:
        t = hmac.new(prk, t + bytes(info) + bytes([i]), hashlib.sha256).digest()
        okm += t
        i += 1
    return okm[:length]


def _ctx():
    return N.default_context()


def _raise(rc):
    if rc != N.OK:
        if rc >= N.ERR_DEVICE:
            raise N.MblsError(rc, _ctx().last_error())
        raise AmclError(rc)


class SecretKey:
    """reference src/keys.rs:28-113. Host-only object; the scalar is sent to the GPU only for signing."""

    def __init__(self, x):
        self._x = int(x)

    @classmethod
    def random(cls, rng=None):
        ikm = os.urandom(32) if rng is None else bytes(rng.getrandbits(8) for _ in range(32))
        return cls.key_generate(ikm, b"")

    @classmethod
    def key_generate(cls, ikm, key_info=b""):
        """KeyGenerate, reference src/keys.rs:45-77 (HKDF-SHA256 with the salt-rehash loop)."""
        if len(ikm) < 32:
            raise AmclError(AmclError.InvalidSecretKeySize)
        sk, salt = 0, KEY_SALT
        while sk == 0:
            salt = hashlib.sha256(salt).digest()
            prk = hkdf_extract(salt, bytes(ikm) + b"\x00")
            okm = hkdf_expand(prk, bytes(key_info) + bytes([0, L]), L)
            sk = int.from_bytes(okm, "big") % CURVE_ORDER
        return cls(sk)

    @classmethod
    def from_bytes(cls, data):
        """reference src/keys.rs:80-82, error cases pinned by tests at src/keys.rs:285-297."""
        data = bytes(data)
        if len(data) != SECRET_KEY_BYTES:
            raise AmclError(AmclError.InvalidSecretKeySize)
        x = int.from_bytes(data, "big")
        if x == 0 or x >= CURVE_ORDER:
            raise AmclError(AmclError.InvalidSecretKeyRange)
        return cls(x)

    def as_bytes(self):
        return self._x.to_bytes(SECRET_KEY_BYTES, "big")

    def as_raw(self):
        return self._x

    def __eq__(self, other):
        return isinstance(other, SecretKey) and self.as_bytes() == other.as_bytes()


class PublicKey:
    """reference src/keys.rs:116-187. `point` is the 96-byte uncompressed form (amcl's layout is private)."""

    def __init__(self, point):
        self.point = bytes(point)

    @classmethod
    def from_secret_key(cls, sk):
        out = N.outbuf(96)
        _raise(N.lib().mbls_pk_from_secret_key(_ctx().handle, N.cbuf(sk.as_bytes()), 32, out))
        return cls(bytes(out))

    @classmethod
    def from_bytes(cls, data):
        out = N.outbuf(96)
        _raise(N.lib().mbls_pk_from_bytes(_ctx().handle, N.cbuf(data), len(data), out))
        return cls(bytes(out))

    @classmethod
    def from_bytes_unchecked(cls, data):
        out = N.outbuf(96)
        _raise(N.lib().mbls_pk_from_bytes_unchecked(_ctx().handle, N.cbuf(data), len(data), out))
        return cls(bytes(out))

    @classmethod
    def from_uncompressed_bytes(cls, data):
        out = N.outbuf(96)
        _raise(N.lib().mbls_pk_from_uncompressed_bytes(_ctx().handle, N.cbuf(data), len(data), out))
        return cls(bytes(out))

    def as_bytes(self):
        out = N.outbuf(48)
        _raise(N.lib().mbls_pk_as_bytes(_ctx().handle, N.cbuf(self.point), out))
        return bytes(out)

    def as_uncompressed_bytes(self):
        return self.point

    def key_validate(self):
        return bool(N.lib().mbls_pk_key_validate(_ctx().handle, N.cbuf(self.point)))

    def is_infinity(self):
        return self.point[0] == 0x40

    def __eq__(self, other):
        return isinstance(other, PublicKey) and self.point == other.point


class Keypair:
    """reference src/keys.rs:189-204."""

    def __init__(self, sk, pk):
        self.sk, self.pk = sk, pk

    @classmethod
    def random(cls, rng=None):
        sk = SecretKey.random(rng)
        return cls(sk, PublicKey.from_secret_key(sk))


class Signature:
    """reference src/signature.rs:9-51. `point` is the 96-byte compressed form."""

    def __init__(self, point):
        self.point = bytes(point)

    @classmethod
    def new(cls, msg, sk):
        out = N.outbuf(96)
        _raise(N.lib().mbls_sign(_ctx().handle, N.cbuf(msg), len(msg), N.cbuf(sk.as_bytes()), 32, out))
        return cls(bytes(out))

    def verify(self, msg, pk):
        return bool(N.lib().mbls_verify(_ctx().handle, N.cbuf(self.point), N.cbuf(msg), len(msg), N.cbuf(pk.point)))

    @classmethod
    def from_bytes(cls, data):
        out = N.outbuf(96)
        _raise(N.lib().mbls_sig_from_bytes(_ctx().handle, N.cbuf(data), len(data), out))
        return cls(bytes(out))

    def as_bytes(self):
        return self.point

    def __eq__(self, other):
        return isinstance(other, Signature) and self.point == other.point


class AggregatePublicKey:
    """reference src/aggregates.rs:17-78."""

    def __init__(self, point):
        self.point = bytes(point)

    @classmethod
    def aggregate(cls, keys):
        if len(keys) == 0:
            raise AmclError(AmclError.AggregateEmptyPoints)
        out = N.outbuf(96)
        _raise(N.lib().mbls_aggregate_public_keys(_ctx().handle, N.cbuf(b"".join(k.point for k in keys)), len(keys), out))
        return cls(bytes(out))

    into_aggregate = aggregate

    @classmethod
    def from_public_key(cls, key):
        return cls(key.point)

    def add(self, public_key):
        out = N.outbuf(96)
        _raise(N.lib().mbls_aggregate_public_key_add(_ctx().handle, N.cbuf(self.point), N.cbuf(public_key.point), out))
        self.point = bytes(out)

    def add_aggregate(self, other):
        self.add(other)

    def is_infinity(self):
        return self.point[0] == 0x40

    def __eq__(self, other):
        return isinstance(other, AggregatePublicKey) and self.point == other.point


class AggregateSignature:
    """reference src/aggregates.rs:83-334."""

    def __init__(self, point=_G2_INFINITY):
        self.point = bytes(point)

    @classmethod
    def new(cls):
        return cls(_G2_INFINITY)

    @classmethod
    def aggregate(cls, signatures):
        """reference src/aggregates.rs:100-106: one batched launch (segmented G2 sum) instead of one addition per call"""
        signatures = list(signatures)
        if not signatures:
            return cls.new()
        out, errs = N.outbuf(96), N.outbuf(1)
        _raise(N.lib().mbls_aggregate_signatures_batch(_ctx().handle, N.cbuf(b"".join(s.point for s in signatures)), None, 1, len(signatures), out, errs))
        _raise(bytes(errs)[0])
        return cls(bytes(out))

    @classmethod
    def from_signature(cls, signature):
        return cls(signature.point)

    def clone(self):
        return AggregateSignature(self.point)

    def add(self, signature):
        out = N.outbuf(96)
        _raise(N.lib().mbls_aggregate_signature_add(_ctx().handle, N.cbuf(self.point), N.cbuf(signature.point), out))
        self.point = bytes(out)

    def add_aggregate(self, other):
        self.add(other)

    def aggregate_verify(self, msgs, public_keys):
        lens = (C.c_size_t * max(1, len(msgs)))(*[len(m) for m in msgs])
        return bool(N.lib().mbls_aggregate_verify(_ctx().handle, N.cbuf(self.point), N.cbuf(b"".join(msgs)), lens, len(msgs),
                                                   N.cbuf(b"".join(k.point for k in public_keys)), len(public_keys)))

    def fast_aggregate_verify(self, msg, public_keys):
        return bool(N.lib().mbls_fast_aggregate_verify(_ctx().handle, N.cbuf(self.point), N.cbuf(msg), len(msg),
                                                        N.cbuf(b"".join(k.point for k in public_keys)), len(public_keys)))

    def fast_aggregate_verify_pre_aggregated(self, msg, aggregate_public_key):
        return bool(N.lib().mbls_fast_aggregate_verify_pre_aggregated(_ctx().handle, N.cbuf(self.point), N.cbuf(msg), len(msg),
                                                                       N.cbuf(aggregate_public_key.point)))

    @staticmethod
    def verify_multiple_aggregate_signatures(rng, signature_sets, devices=None):
        """reference src/aggregates.rs:261-316. devices (optional, not in the reference): a _native.MultiContext -- the sets are then cut into one shard per device
        (mbls_multi_verify_multiple_aggregate_signatures_rng: same bool, same draws). `rng` must offer getrandbits (random.Random); the blinding scalars
        are drawn exactly as at :280-287: 8 random bytes, big-endian i64, absolute value, retry on zero -- and IN THE REFERENCE'S ORDER: its loop
        tests set i's signature for the subgroup (:272-275) before it draws rand[i], and returns at the first signature outside G2, so a rejected
        batch leaves the caller's generator where the reference would (mbls_verify_multiple_aggregate_signatures_rng tests the signatures first and
        asks for the scalars of the sets in front of the first bad one only). Messages may have any lengths (`&[u8]` per set in the reference): one buffer + an offset table."""
        sets = list(signature_sets)
        if not sets:
            return bool(N.lib().mbls_verify_multiple_aggregate_signatures(_ctx().handle, None, None, None, 0, None, None, 0))
        failed = []

        def draw(_user, out, count):                    # src/aggregates.rs:280-287, for the sets the reference's loop reaches a draw for
            try:
                for i in range(count):
                    r = 0
                    while r == 0:
                        v = int.from_bytes(bytes(rng.getrandbits(8) for _ in range(8)), "big", signed=True)
                        r = abs(v) & 0xFFFFFFFFFFFFFFFF
                    out[i] = r
            except BaseException as e:                  # an exception must not unwind through the C frames: the batch fails, the error is raised afterwards
                failed.append(e)
                for i in range(count):
                    out[i] = 0
        offs = [0]
        for s in sets:
            offs.append(offs[-1] + len(s[2]))
        moff = (C.c_uint64 * len(offs))(*offs)
        cb = N.SCALAR_SOURCE(draw)
        entry, handle = ((N.lib().mbls_verify_multiple_aggregate_signatures_rng, _ctx().handle) if devices is None else
                         (N.lib().mbls_multi_verify_multiple_aggregate_signatures_rng, devices.handle))
        ok = bool(entry(handle, N.cbuf(b"".join(s[0].point for s in sets)), N.cbuf(b"".join(s[1].point for s in sets)),
                        N.cbuf(b"".join(bytes(s[2]) for s in sets)), 0, moff, len(sets), cb, None))
        if failed:
            raise failed[0]
        return ok

    @classmethod
    def from_bytes(cls, data):
        out = N.outbuf(96)
        _raise(N.lib().mbls_sig_from_bytes(_ctx().handle, N.cbuf(data), len(data), out))
        return cls(bytes(out))

    def as_bytes(self):
        return self.point

    def __eq__(self, other):
        return isinstance(other, AggregateSignature) and self.point == other.point
