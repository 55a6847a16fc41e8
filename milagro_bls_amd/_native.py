"""ctypes loader for libmbls_hip.so. There is no CPU fallback: if the HIP library is missing, or no GPU
is present when a context is created, this raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MBLS_LIB", os.path.join(HERE, "libmbls_hip.so"))

# error codes (include/mbls.h)
OK = 0
ERR_INVALID_G1_SIZE = 1
ERR_INVALID_G2_SIZE = 2
ERR_INVALID_POINT = 3
ERR_AGGREGATE_EMPTY_POINTS = 4
ERR_INVALID_SECRET_KEY_SIZE = 5
ERR_INVALID_SECRET_KEY_RANGE = 6
ERR_DEVICE = 100
ERR_ARGUMENT = 101
PK_COMPRESSED, PK_UNCOMPRESSED = 0, 1
VM_PARTIAL_BYTES = 896          # include/mbls.h MBLS_VM_PARTIAL_BYTES
N_PHASES = 6
PHASE_NAMES = ("aggregate", "sig", "hash", "miller", "final", "pack")

_lib = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p
class Limits(C.Structure):
    """include/mbls.h mbls_limits"""
    _fields_ = [(n, C.c_uint64) for n in ("round_items", "coop_max_items", "coop_hash_max_items", "coop_pack_min_items", "coop_pack_max_items", "coop_hash_pack_min_items",
                                          "split_max_items", "fork_max_items", "hash2_max_items", "tracks_min_rest", "tracks_side_max")]


class PassPlan(C.Structure):
    """include/mbls.h mbls_pass_plan"""
    _fields_ = [(n, C.c_uint64) for n in ("first_item", "items", "workspace_first", "workspace_items")] + \
               [(n, C.c_uint32) for n in ("track", "stage", "pairing", "message", "front", "sig_subgroup_from_miller_loop")]


class BatchPlan(C.Structure):
    """include/mbls.h mbls_batch_plan"""
    _fields_ = [("mode", C.c_uint32), ("n_passes", C.c_uint32), ("passes", PassPlan * 3)]


PAIRING_WAVE, PAIRING_LANE, PAIRING_LANES2, PAIRING_LANES4, PAIRING_WAVE_X2 = 0, 1, 2, 4, 5
MESSAGE_LANE, MESSAGE_LANES2, MESSAGE_WAVE, MESSAGE_WAVE_X4 = 1, 2, 3, 4
FRONT_IN_A_ROW, FRONT_MESSAGE_BESIDE, FRONT_ALL_BESIDE = 0, 1, 2
BATCH_ONE_PASS, BATCH_ROUNDS_THEN_REST, BATCH_ROUND_BESIDE_REST, BATCH_TWO_HALVES = 0, 1, 2, 3


def default_limits(round_items=65536):
    """the library's default routing limits for a device whose round is round_items lanes (pure: no GPU)"""
    L = Limits()
    lib().mbls_default_limits(round_items, C.byref(L))
    return L


def plan_batch(n, limits=None):
    """what mbls_fast_aggregate_verify_batch_device would do with n items under `limits` (pure: no GPU) -> (mode, [pass dicts])"""
    L = limits if limits is not None else default_limits()
    b = BatchPlan()
    rc = lib().mbls_plan_batch(C.byref(L), n, C.byref(b))
    if rc != OK:
        raise MblsError(rc, "mbls_plan_batch")
    return b.mode, [{f: getattr(b.passes[i], f) for f, _ in PassPlan._fields_} for i in range(b.n_passes)]


def plan_workspace_items(n, k, split_layout=True, limits=None):
    """workspace items the plan of n items of k keys needs (pure: no GPU); split_layout: uniform 96-byte keys or key-table indices"""
    L = limits if limits is not None else default_limits()
    return int(lib().mbls_plan_workspace_items(C.byref(L), n, k, 1 if split_layout else 0))


# mbls_scalar_source (include/mbls.h): void draw(void* user, uint64_t* out, uint64_t count)
SCALAR_SOURCE = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64)
SIGNATURES = {
    "mbls_ctx_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "mbls_ctx_destroy": (None, [vp]),
    "mbls_ctx_reserve": (C.c_int, [vp, C.c_uint64]),
    "mbls_ctx_reserve_keys": (C.c_int, [vp, C.c_uint64]),
    "mbls_ctx_set_coop_max_items": (C.c_int, [vp, C.c_uint64]),
    "mbls_ctx_set_coop_hash_max_items": (C.c_int, [vp, C.c_uint64]),
    "mbls_ctx_set_coop_packing": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mbls_ctx_set_round_items": (C.c_int, [vp, C.c_uint64]),
    "mbls_ctx_reset_tuning": (C.c_int, [vp]),
    "mbls_ctx_set_lane_shaping": (C.c_int, [vp, C.c_uint64, C.c_uint64]),
    "mbls_ctx_set_tracks": (C.c_int, [vp, C.c_uint64, C.c_uint64]),
    "mbls_ctx_set_secret_ops": (C.c_int, [vp, C.c_int]),
    "mbls_default_limits": (None, [C.c_uint64, vp]),
    "mbls_ctx_get_limits": (C.c_int, [vp, vp]),
    "mbls_plan_batch": (C.c_int, [vp, C.c_uint64, vp]),
    "mbls_plan_workspace_items": (C.c_uint64, [vp, C.c_uint64, C.c_uint32, C.c_int]),
    "mbls_last_error": (C.c_char_p, [vp]),
    "mbls_fast_aggregate_verify_batch_device": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, C.c_int, vp, C.c_uint64, C.c_uint32, vp, vp, vp, vp]),
    "mbls_fast_aggregate_verify_batch": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, C.c_int, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "mbls_verify_batch_device": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, C.c_int, C.c_uint64, vp, vp, vp, vp]),
    "mbls_verify_batch": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, C.c_int, C.c_uint64, vp, vp]),
    "mbls_aggregate_verify_batch": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, vp, C.c_uint32, C.c_uint64, vp, vp]),
    "mbls_aggregate_verify_batch_device": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp, vp]),
    "mbls_keytable_create": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "mbls_keytable_destroy": (None, [vp]),
    "mbls_keytable_size": (C.c_uint64, [vp]),
    "mbls_keytable_append": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), vp]),
    "mbls_keytable_append_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), vp, vp]),
    "mbls_keytable_get": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, vp]),
    "mbls_fast_aggregate_verify_batch_indexed_device": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp, vp]),
    "mbls_fast_aggregate_verify_batch_indexed": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "mbls_aggregate_signatures_batch": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "mbls_aggregate_signatures_batch_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint64, vp, vp, vp]),
    "mbls_pk_from_bytes": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "mbls_pk_from_bytes_unchecked": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "mbls_pk_from_uncompressed_bytes": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "mbls_pk_as_bytes": (C.c_int, [vp, vp, vp]),
    "mbls_pk_key_validate": (C.c_int, [vp, vp]),
    "mbls_pk_from_secret_key": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "mbls_sig_from_bytes": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "mbls_sign": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, vp]),
    "mbls_verify": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "mbls_aggregate_public_keys": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "mbls_aggregate_public_key_add": (C.c_int, [vp, vp, vp, vp]),
    "mbls_aggregate_signature_add": (C.c_int, [vp, vp, vp, vp]),
    "mbls_fast_aggregate_verify": (C.c_int, [vp, vp, vp, C.c_size_t, vp, C.c_size_t]),
    "mbls_fast_aggregate_verify_pre_aggregated": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "mbls_aggregate_verify": (C.c_int, [vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t]),
    "mbls_verify_multiple_aggregate_signatures": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_size_t]),
    "mbls_verify_multiple_aggregate_signatures_device": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint64, vp, vp, vp]),
    "mbls_verify_multiple_aggregate_signatures_rng": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, C.c_size_t, SCALAR_SOURCE, vp]),
    "mbls_verify_multiple_sets_device": (C.c_int, [vp, vp, vp, C.c_int, vp, C.c_uint32, vp, C.c_uint32, vp, vp, C.c_uint64, vp, vp, vp]),
    "mbls_verify_multiple_sets_indexed_device": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint32, vp, C.c_uint32, vp, vp, C.c_uint64, vp, vp, vp, vp]),
    "mbls_verify_multiple_partial_device": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, C.c_uint32, vp, C.c_uint32, vp, vp, C.c_uint64, vp, vp]),
    "mbls_verify_multiple_finish_device": (C.c_int, [vp, vp, C.c_uint64, vp, vp, vp]),
    "mbls_multi_verify_multiple_aggregate_signatures": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_size_t]),
    "mbls_multi_verify_multiple_aggregate_signatures_rng": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, C.c_size_t, SCALAR_SOURCE, vp]),
    "mbls_pk_decode_batch": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_uint64, vp, vp]),
    "mbls_pk_compress_batch": (C.c_int, [vp, vp, C.c_uint64, vp, vp]),
    "mbls_sig_check_batch": (C.c_int, [vp, vp, C.c_uint64, vp, vp]),
    "mbls_sign_batch": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint64, vp]),
    "mbls_sign_batch_device": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint64, vp, vp]),
    "mbls_sk_to_pk_batch": (C.c_int, [vp, vp, C.c_int, C.c_uint64, vp]),
    "mbls_sk_to_pk_batch_device": (C.c_int, [vp, vp, C.c_int, C.c_uint64, vp, vp]),
    "mbls_hash_to_g2_batch": (C.c_int, [vp, vp, C.c_uint32, C.c_uint64, vp]),
    "mbls_hash_to_g2_batch_mode": (C.c_int, [vp, vp, C.c_uint32, C.c_uint64, vp, C.c_int]),
    "mbls_aggregate_public_keys_batch": (C.c_int, [vp, vp, C.c_int, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "mbls_fp_mul_batch": (C.c_int, [vp, vp, vp, C.c_uint64, vp, C.c_int]),
    "mbls_fp_mul_bench": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_float)]),
    "mbls_valu_bench": (C.c_int, [vp, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "mbls_multi_create": (C.c_int, [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]),
    "mbls_multi_destroy": (None, [vp]),
    "mbls_multi_device_count": (C.c_int, [vp]),
    "mbls_multi_last_error": (C.c_char_p, [vp]),
    "mbls_multi_context": (vp, [vp, C.c_int]),
    "mbls_multi_reserve": (C.c_int, [vp, C.c_uint64]),
    "mbls_multi_rccl_active": (C.c_int, [vp]),
    "mbls_multi_exchange_note": (C.c_char_p, [vp]),
    "mbls_multi_fast_aggregate_verify_bitmap": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, C.c_int, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "mbls_multi_device_bitmap": (vp, [vp, C.c_int]),
    "mbls_multi_fast_aggregate_verify_batch": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, C.c_int, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "mbls_multi_verify_batch": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, C.c_int, C.c_uint64, vp, vp]),
    "mbls_multi_keytable_create": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "mbls_multi_keytable_destroy": (None, [vp]),
    "mbls_multi_keytable_size": (C.c_uint64, [vp]),
    "mbls_multi_keytable_replica": (vp, [vp, C.c_int]),
    "mbls_multi_keytable_append": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), vp]),
    "mbls_multi_fast_aggregate_verify_batch_indexed": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "mbls_enable_phase_timing": (C.c_int, [vp, C.c_int]),
    "mbls_last_phase_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
}


class MblsError(RuntimeError):
    def __init__(self, code, what=""):
        super().__init__("mbls error %d %s" % (code, what))
        self.code = code


def lib():
    """Load libmbls_hip.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / HSA runtime. Two HIP runtimes in one process
        # do not coexist (the second one finds no GPU), so when torch is installed it is imported first: our
        # library's NEEDED libamdhip64.so.7 then resolves to the copy torch already loaded. Set MBLS_STANDALONE=1
        # to skip this and run against /opt/rocm's runtime alone.
        if os.environ.get("MBLS_STANDALONE", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libmbls_hip.so not built: run `python -m milagro_bls_amd.build` (HIP extension is mandatory)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


class Context:
    """One mbls_ctx per GPU."""

    def __init__(self, device_id=0):
        self._h = vp()
        rc = lib().mbls_ctx_create(C.byref(self._h), device_id)
        if rc != OK:
            raise MblsError(rc, "mbls_ctx_create failed (no MI355X / HIP device available?)")

    def close(self):
        if self._h:
            lib().mbls_ctx_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def last_error(self):
        return lib().mbls_last_error(self._h).decode()

    def check(self, rc):
        if rc != OK:
            raise MblsError(rc, self.last_error())

    def reserve(self, n):
        self.check(lib().mbls_ctx_reserve(self._h, n))

    def set_coop_max_items(self, n):
        """batches up to n items take the one-wave-per-item pairing check (0: never)"""
        self.check(lib().mbls_ctx_set_coop_max_items(self._h, n))

    def set_coop_hash_max_items(self, n):
        self.check(lib().mbls_ctx_set_coop_hash_max_items(self._h, n))

    def set_lane_shaping(self, split_max_items, fork_max_items):
        """one-lane path below a full round: two lanes per item in the Miller phase up to split_max_items, front phases side by side up to fork_max_items"""
        self.check(lib().mbls_ctx_set_lane_shaping(self._h, split_max_items, fork_max_items))

    def set_tracks(self, min_rest_items, side_max_items=16384):
        """batches of q rounds + r items with r >= min_rest_items: the last round and the remainder on two tracks side by side -- the remainder beside the round
        up to side_max_items, two equal halves above (min_rest_items = 0: never)"""
        self.check(lib().mbls_ctx_set_tracks(self._h, min_rest_items, side_max_items))

    def limits(self):
        """the context's current routing limits (setters and environment applied)"""
        L = Limits()
        self.check(lib().mbls_ctx_get_limits(self._h, C.byref(L)))
        return L

    def set_secret_ops(self, variable_time):
        """False (default): signing / sk -> pk look their tables up by scan + selection (constant-time access); True: by key-dependent address (throw-away keys only)"""
        self.check(lib().mbls_ctx_set_secret_ops(self._h, int(bool(variable_time))))

    def reset_tuning(self):
        """every routing parameter (engine crossovers, packing, round, lane shaping) back to the library's defaults"""
        self.check(lib().mbls_ctx_reset_tuning(self._h))

    def set_round_items(self, items):
        """items per round of the one-lane kernels (0: the device's CUs x 4 x 64); a batch's remainder above whole rounds is routed on its own"""
        self.check(lib().mbls_ctx_set_round_items(self._h, items))

    def set_coop_packing(self, pairing_min_items, pairing_max_items, hash_min_items):
        """pairing_min < n <= pairing_max: two items per wave in the pairing check; n > hash_min: four per wave in the message phase"""
        self.check(lib().mbls_ctx_set_coop_packing(self._h, pairing_min_items, pairing_max_items, hash_min_items))


class KeyTable:
    """Resident table of decoded public keys in HBM (include/mbls.h, mbls_keytable_*): decode once, verify by index."""

    def __init__(self, ctx=None, capacity_hint=0):
        self.ctx = ctx or default_context()
        self._h = vp()
        self.ctx.check(lib().mbls_keytable_create(self.ctx.handle, capacity_hint, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().mbls_keytable_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def __len__(self):
        return int(lib().mbls_keytable_size(self._h))

    def append(self, pks, n, pk_format=PK_COMPRESSED, validate=True):
        """-> (first_index, errs): errs[i] = OK / ERR_* as PublicKey::from_bytes[_unchecked] / from_uncompressed_bytes would return"""
        first = C.c_uint64(0)
        errs = outbuf(n)
        self.ctx.check(lib().mbls_keytable_append(self._h, cbuf(pks), pk_format, int(validate), n, C.byref(first), errs))
        return int(first.value), list(bytes(errs)[:n])

    def append_device(self, d_pks, n, d_errs, pk_format=PK_COMPRESSED, validate=True, stream=None):
        first = C.c_uint64(0)
        self.ctx.check(lib().mbls_keytable_append_device(self._h, d_pks, pk_format, int(validate), n, C.byref(first), d_errs, stream))
        return int(first.value)

    def get(self, first, n=1):
        out, errs = outbuf(96 * n), outbuf(n)
        self.ctx.check(lib().mbls_keytable_get(self._h, first, n, out, errs))
        return bytes(out)[:96 * n], list(bytes(errs)[:n])


class MultiContext:
    """Several GPUs behind one handle (include/mbls.h, mbls_multi_*): contiguous shards, one context and one host thread per device."""

    def __init__(self, device_ids):
        ids = (C.c_int * len(device_ids))(*device_ids)
        self._h = vp()
        rc = lib().mbls_multi_create(C.byref(self._h), ids, len(device_ids))
        if rc != OK:
            raise MblsError(rc, "mbls_multi_create failed")

    def close(self):
        if self._h:
            lib().mbls_multi_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def __len__(self):
        return int(lib().mbls_multi_device_count(self._h))

    def check(self, rc):
        if rc != OK:
            raise MblsError(rc, lib().mbls_multi_last_error(self._h).decode())

    def reserve(self, n):
        self.check(lib().mbls_multi_reserve(self._h, n))

    @property
    def rccl_active(self):
        """True: the handle's exchange steps (accept bitmap, verify_multiple's partial records) are RCCL all-gathers between the devices"""
        return bool(lib().mbls_multi_rccl_active(self._h))

    @property
    def exchange_note(self):
        return lib().mbls_multi_exchange_note(self._h).decode()


class MultiKeyTable:
    """A key table replicated on every device of a MultiContext (same indices everywhere)."""

    def __init__(self, mctx, capacity_hint=0):
        self.m = mctx
        self._h = vp()
        mctx.check(lib().mbls_multi_keytable_create(mctx.handle, capacity_hint, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().mbls_multi_keytable_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def __len__(self):
        return int(lib().mbls_multi_keytable_size(self._h))

    def append(self, pks, n, pk_format=PK_COMPRESSED, validate=True):
        first = C.c_uint64(0)
        errs = outbuf(n)
        self.m.check(lib().mbls_multi_keytable_append(self._h, cbuf(pks), pk_format, int(validate), n, C.byref(first), errs))
        return int(first.value), list(bytes(errs)[:n])


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("MBLS_DEVICE", "0")))
    return _default_ctx


def cbuf(data):
    """bytes -> ctypes buffer (kept alive by the caller)."""
    data = bytes(data)
    return (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")


def outbuf(n):
    return (C.c_uint8 * max(1, n))()
