/*
 * mbls.h -- C ABI of libmbls_hip.so: MI355X (gfx950) batch BLS12-381 signature verification.
 *
 * This is the drop-in boundary for the verification path of sigp/milagro_bls. The reference has no FFI of
 * its own (it is a Rust library over the `amcl` crate); these entry points are what a Rust shim re-creating
 * reference src/lib.rs:17-22 would bind. Each entry cites the reference interface it
 * replaces. Plain pointers and sizes only; the caller owns every buffer.
 *
 * Wire formats (ZCash BLS12-381 serialization, as in the reference):
 *   signature / aggregate signature : 96 bytes, compressed G2      (Signature::as_bytes, src/signature.rs:49-51)
 *   public key, compressed          : 48 bytes                      (PublicKey::as_bytes, src/keys.rs:158-160)
 *   public key, uncompressed        : 96 bytes x||y                 (PublicKey::as_uncompressed_bytes, src/keys.rs:163-165)
 *   secret key                      : 32 bytes big-endian           (SecretKey::as_bytes, src/keys.rs:85-87)
 * A decoded PublicKey / AggregatePublicKey object is represented by its 96-byte uncompressed form, a decoded
 * Signature / AggregateSignature by its 96-byte compressed form (amcl's in-memory layout is private).
 *
 * There is NO CPU fallback: every function runs HIP kernels and returns MBLS_ERR_DEVICE if the GPU is
 * unavailable. `*_device` variants take device pointers (inputs already resident in HBM) and a hipStream_t
 * (passed as void*); the others take host pointers and stage through the context's device buffers.
 */
#ifndef MBLS_H
#define MBLS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* error codes: the AmclError variants the reference uses (src/amcl_utils.rs:55,71; src/keys.rs:47,143,293;
   src/aggregates.rs:31) plus device errors */
#define MBLS_OK 0
#define MBLS_ERR_INVALID_G1_SIZE 1          /* AmclError::InvalidG1Size */
#define MBLS_ERR_INVALID_G2_SIZE 2          /* AmclError::InvalidG2Size */
#define MBLS_ERR_INVALID_POINT 3            /* AmclError::InvalidPoint */
#define MBLS_ERR_AGGREGATE_EMPTY_POINTS 4   /* AmclError::AggregateEmptyPoints */
#define MBLS_ERR_INVALID_SECRET_KEY_SIZE 5  /* AmclError::InvalidSecretKeySize */
#define MBLS_ERR_INVALID_SECRET_KEY_RANGE 6 /* AmclError::InvalidSecretKeyRange */
#define MBLS_ERR_DEVICE 100                 /* HIP failure / no GPU */
#define MBLS_ERR_ARGUMENT 101

#define MBLS_G1_BYTES 48                    /* reference src/lib.rs:20 G1_BYTES */
#define MBLS_G2_BYTES 96                    /* reference src/lib.rs:20 G2_BYTES */
#define MBLS_SECRET_KEY_BYTES 32            /* reference src/lib.rs:20 SECRET_KEY_BYTES */
#define MBLS_PK_COMPRESSED 0
#define MBLS_PK_UNCOMPRESSED 1

/* per-item status bits reported by the batch verifiers (why an item was rejected) */
#define MBLS_ST_BAD_SIG_ENCODING 0x01u
#define MBLS_ST_SIG_NOT_IN_G2 0x02u
#define MBLS_ST_BAD_PK_ENCODING 0x04u
#define MBLS_ST_APK_INFINITY 0x08u
#define MBLS_ST_NO_KEYS 0x10u
#define MBLS_ST_PK_INFINITY 0x20u
#define MBLS_ST_PAIRING_FAILED 0x40u
#define MBLS_ST_BAD_SCALAR 0x80u            /* verify_multiple: a zero blinding scalar */
#define MBLS_ST_BAD_MSG_RANGE 0x100u        /* msg_offsets[i+1] < msg_offsets[i] (device entries; host entries refuse the call) */

typedef struct mbls_ctx mbls_ctx;

/* ---- context: one per GPU (one process per GPU in multi-GPU runs) ----
 * Thread safety: every entry point takes the context's lock, so a context may be shared by any number of threads (the
 * reference's functions are pure and re-entrant); calls on one context run one after the other.
 * Device-pointer entries only enqueue work (none of them synchronises with the host): a later call on another stream waits
 * (on the device) for the workspace of the earlier one. For concurrent streams of work create one context per stream.
 * ONE EXCEPTION, by construction: a call whose batch is larger than any the context has seen first GROWS the workspace
 * (hipFree + hipMalloc inside mbls_ctx_reserve: both drain the device). Reserve the largest batch up front with
 * mbls_ctx_reserve / mbls_ctx_reserve_keys and no *_device entry ever synchronises. */
int mbls_ctx_create(mbls_ctx** out, int device_id);
void mbls_ctx_destroy(mbls_ctx* ctx);
/* pre-allocate the HBM workspace for batches of up to max_items items (6 384 bytes per item; avoids allocation in timed regions) */
int mbls_ctx_reserve(mbls_ctx* ctx, uint64_t max_items);
/* pre-allocate the staging area for max_keys decompressed public keys (compressed wire format, 96 bytes per key) */
int mbls_ctx_reserve_keys(mbls_ctx* ctx, uint64_t max_keys);
const char* mbls_last_error(mbls_ctx* ctx);
/* Small batches are latency-bound (one lane per item walks 14 M dependent instructions whatever the batch size), so batches of up to
 * max_items items run their pairing check -- Miller loop + final exponentiation -- with ONE WAVE per item, the item's field values
 * shared by the 64 lanes (mbls_coop.h); the message phase after hash_to_field does the same. Same results, bit for bit.
 * Defaults 5120 / 3584: the measured crossovers (between them the message phase runs on lane pairs; environment, read by mbls_ctx_create and therefore also by every context of an
 * mbls_multi handle: MBLS_COOP_MAX_ITEMS, MBLS_COOP_HASH_MAX_ITEMS); 0 = never. */
int mbls_ctx_set_coop_max_items(mbls_ctx* ctx, uint64_t max_items);
int mbls_ctx_reset_tuning(mbls_ctx* ctx);      /* every routing parameter of this section back to its default (environment overrides included) */
int mbls_ctx_set_coop_hash_max_items(mbls_ctx* ctx, uint64_t max_items);     /* the same for the message phase (never above the limit above) */
/* Within those limits, batches of pairing_min_items < n <= pairing_max_items items run the pairing check with two items per wave, and of
 * more than hash_min_items the message phase with four: more steps per wave, fewer per item -- it pays where it saves a round of waves.
 * Defaults (1024, 2048] and 768 (measured); min = UINT64_MAX: never. */
int mbls_ctx_set_coop_packing(mbls_ctx* ctx, uint64_t pairing_min_items, uint64_t pairing_max_items, uint64_t hash_min_items);
/* One ROUND of the one-lane-per-item kernels is one wave on every SIMD: CUs x 4 x 64 items (65 536 on MI355X; the kernels hold 512 registers
 * per lane, so a SIMD runs one wave at a time). A batch of q rounds + r items would cost q + 1 rounds of every kernel: the library runs the
 * q rounds and then the r items as a batch of their own, which takes the route of an r-item batch (one wave per item up to the limits
 * above). items = 0 restores the device's value; any multiple of 64 is accepted (tests use small rounds to exercise the cut). */
int mbls_ctx_set_round_items(mbls_ctx* ctx, uint64_t items);
/* Shaping of the one-lane path below a full round. Up to split_max_items items (default and maximum: half a round) the two pairs of an item's
 * Miller loop are walked on TWO lanes by the one-pair routine (6.6 ms instead of 11.3 ms; the product and the signature's subgroup verdict
 * follow as separate small kernels); up to fork_max_items items (default: three quarters of a round) the three front phases -- key sum, signature decoding,
 * message hashing -- are enqueued side by side on the context's own streams instead of one after the other. 0 = never. Same results, bit for
 * bit (environment for new contexts: MBLS_SPLIT_MAX_ITEMS, MBLS_FORK_MAX_ITEMS). */
int mbls_ctx_set_lane_shaping(mbls_ctx* ctx, uint64_t split_max_items, uint64_t fork_max_items);
/* Batches above a round, n = q rounds + r items. r < min_rest_items: the remainder follows the rounds as a batch of its own (mbls_ctx_set_round_items).
 * r >= min_rest_items: after the q - 1 whole rounds in front, the LAST round and the remainder run on TWO TRACKS side by side -- each on its own part of the
 * workspace and its own streams, so that the SIMDs one leaves idle take waves of the other --: up to side_max_items the round on one track and the remainder (on
 * the lane-pair forms, whatever its size) on the other, above it two equal halves of (round + r) / 2 items. Defaults 3 584 and a quarter of a round (measured:
 * 69 632 items 35.0 -> 33.7 ms, 73 728 items 40.0 -> 33.8 ms, 100 000 items 51.6 -> 46.5 ms); min_rest_items = 0: never. Same results, bit for bit (environment for new contexts:
 * MBLS_TRACKS_MIN_REST, MBLS_TRACKS_SIDE_MAX). */
int mbls_ctx_set_tracks(mbls_ctx* ctx, uint64_t min_rest_items, uint64_t side_max_items);
/* 0 (default): lookups that depend on a secret key are scans with selection; 1: the variable-time forms (see "SECRET KEYS ON THE DEVICE" below) */
int mbls_ctx_set_secret_ops(mbls_ctx* ctx, int variable_time);

/* ---- routing, as data (pure functions: no GPU, no context) -------------------------------------------------
 * Which kernels a batch of n items takes is decided by the limits above; mbls_plan_batch states the decision without running anything -- the SAME
 * function the verification entries call (verify_pipeline), so the table in DESIGN.md section 5 is checkable on a machine without a GPU
 * (tests/test_plan_cpu.py). mbls_default_limits fills the defaults for a device with round_items = CUs x 4 x 64 (65 536 on MI355X);
 * mbls_ctx_get_limits reads a context's current ones (setters and environment applied). The plan is a function of n alone; two forms follow from it and from
 * the keys per item: with the pairing check on waves (MBLS_PAIRING_WAVE*) the signature phase runs on lane pairs (k_sig2), and an item's key sum of k >= 32
 * keys, k a multiple of 8, uniform layout, as eight partial sums on lanes of their own (k_apk_combine; workspace n + 8 n items instead of workspace_items). */
typedef struct mbls_limits {
    uint64_t round_items, coop_max_items, coop_hash_max_items, coop_pack_min_items, coop_pack_max_items, coop_hash_pack_min_items,
             split_max_items, fork_max_items, hash2_max_items, tracks_min_rest, tracks_side_max;
} mbls_limits;
enum { MBLS_PAIRING_WAVE = 0,        /* one wave per item walks Miller loop + final exponentiation (program pairing2) */
       MBLS_PAIRING_WAVE_X2 = 5,     /* ... two items per wave (pairing2x2) */
       MBLS_PAIRING_LANE = 1,        /* one lane per item: k_miller (two-pair loop) + k_final -- the headline kernels */
       MBLS_PAIRING_LANES2 = 2,      /* two lanes per item: k_miller_split + k_final2 */
       MBLS_PAIRING_LANES4 = 4 };    /* four lanes per item in the Miller phase (k_miller_split4), two in the final exponentiation (k_final2) */
enum { MBLS_MESSAGE_LANE = 1, MBLS_MESSAGE_LANES2 = 2, MBLS_MESSAGE_WAVE = 3, MBLS_MESSAGE_WAVE_X4 = 4 };   /* k_hash / k_hash2 / hashg2 / hashg2x4 */
enum { MBLS_FRONT_IN_A_ROW = 0,      /* key sum -> signature -> message phase on one stream */
       MBLS_FRONT_MESSAGE_BESIDE = 1,/* message phase on a side stream beside key sum -> signature */
       MBLS_FRONT_ALL_BESIDE = 2 };  /* all three side by side */
typedef struct mbls_pass_plan {      /* one pass of the pipeline over a contiguous range of items */
    uint64_t first_item, items, workspace_first, workspace_items;
    uint32_t track;                  /* 0: the caller's stream; 1: the second track, side by side with track 0 of the same stage */
    uint32_t stage;                  /* passes of one stage run side by side, stages one after the other */
    uint32_t pairing, message, front, sig_subgroup_from_miller_loop;
} mbls_pass_plan;
enum { MBLS_BATCH_ONE_PASS = 0, MBLS_BATCH_ROUNDS_THEN_REST = 1, MBLS_BATCH_ROUND_BESIDE_REST = 2, MBLS_BATCH_TWO_HALVES = 3 };
typedef struct mbls_batch_plan { uint32_t mode, n_passes; mbls_pass_plan pass[3]; } mbls_batch_plan;
void mbls_default_limits(uint64_t round_items, mbls_limits* out);
int mbls_ctx_get_limits(mbls_ctx* ctx, mbls_limits* out);
/* the plan of mbls_fast_aggregate_verify_batch[_indexed]_device / mbls_verify_batch_device for n items (MBLS_ERR_ARGUMENT for n = 0 or null pointers) */
int mbls_plan_batch(const mbls_limits* limits, uint64_t n, mbls_batch_plan* out);
/* the workspace items that plan needs for items of k keys each -- split_layout != 0: uniform 96-byte keys (4-byte aligned) or key-table indices, the layouts whose
 * key sums on the wave engine take eight lanes per item (n + 8 n items for such a pass instead of its workspace_items). This is what the verification entries
 * reserve BEFORE they queue the first pass of a plan (no pass ever grows the workspace under another in flight); mbls_ctx_reserve(ctx, this) beforehand keeps
 * every allocation out of the call. 0 for n = 0 or a null pointer. */
uint64_t mbls_plan_workspace_items(const mbls_limits* limits, uint64_t n, uint32_t k, int split_layout);

/* ---- the hot path -------------------------------------------------------------------------------------
 * Batch of n independent AggregateSignature::fast_aggregate_verify calls (reference src/aggregates.rs:177-215):
 * item i = (sigs[96 i..], its message, its public keys). The reference takes any `msg: &[u8]` per call: messages are
 * either msg_len bytes each, contiguous (msg_offsets == NULL: item i's message is msgs[msg_len i ..]), or of any length
 * each: item i's message is msgs[msg_offsets[i] .. msg_offsets[i+1]) with msg_offsets of n + 1 non-decreasing entries
 * (msg_len is then ignored; host entries refuse a table that runs backwards or holds a message of 2^32 bytes or more,
 * device entries reject such an item with MBLS_ST_BAD_MSG_RANGE). Keys are either k per item, contiguous
 * (pk_offsets == NULL), or ragged: item i owns keys [pk_offsets[i], pk_offsets[i+1]) of `pks`.
 * results[i] = 1/0 exactly as the reference function returns true/false, including its check order:
 * empty key list -> 0, signature outside G2 -> 0, aggregate key = infinity -> 0, pairing check.
 * bitmap (optional, ceil(n/64) words): bit (i%64) of word i/64 = results[i]. status (optional): MBLS_ST_* bits. */
int mbls_fast_aggregate_verify_batch_device(mbls_ctx* ctx, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_len,
                                            const uint64_t* d_msg_offsets, const uint8_t* d_pks, int pk_format,
                                            const uint32_t* d_pk_offsets, uint64_t n, uint32_t k, uint8_t* d_results,
                                            uint64_t* d_bitmap, uint32_t* d_status, void* stream);
int mbls_fast_aggregate_verify_batch(mbls_ctx* ctx, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len,
                                     const uint64_t* msg_offsets, const uint8_t* pks, int pk_format,
                                     const uint32_t* pk_offsets, uint64_t n, uint32_t k, uint8_t* results, uint32_t* status);
/* Batch of n Signature::verify calls (reference src/signature.rs:27-40): one key per item, no infinity check. */
int mbls_verify_batch_device(mbls_ctx* ctx, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_len,
                             const uint64_t* d_msg_offsets, const uint8_t* d_pks, int pk_format, uint64_t n,
                             uint8_t* d_results, uint64_t* d_bitmap, uint32_t* d_status, void* stream);
int mbls_verify_batch(mbls_ctx* ctx, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* msg_offsets,
                      const uint8_t* pks, int pk_format, uint64_t n, uint8_t* results, uint32_t* status);

/* ---- resident key table ---------------------------------------------------------------------------------
 * The on-device analogue of the decoded PublicKey objects a reference caller keeps (src/keys.rs:116-120; callers cache
 * decoded keys through as_uncompressed_bytes / from_uncompressed_bytes, src/keys.rs:163-175): keys are decoded (and
 * optionally KeyValidate'd) ONCE into HBM as affine Montgomery limbs, and a verification names its keys by table index --
 * what `&[&PublicKey]` is in the reference's fast_aggregate_verify (src/aggregates.rs:177). Per use this removes the byte
 * decoding, the Montgomery conversion and the on-curve check (5 of 16 multiplications per key), makes the compressed
 * 48-byte wire format a one-time cost, and cuts the per-item input to 96 + msg_len + 4 k bytes.
 * A table belongs to the context it was created with (same GPU, same lock). Entries are never removed; indices are stable.
 * Ordering: an append made through mbls_keytable_append_device on one stream is seen by verifications and mbls_keytable_get on any
 * other stream (they wait for it on the device). Destroying the context first releases its tables' records; the handles stay
 * valid for mbls_keytable_destroy only. */
typedef struct mbls_keytable mbls_keytable;
int mbls_keytable_create(mbls_ctx* ctx, uint64_t capacity_hint, mbls_keytable** out);
void mbls_keytable_destroy(mbls_keytable* t);
uint64_t mbls_keytable_size(const mbls_keytable* t);
/* n x PublicKey::from_bytes (pk_format compressed, validate = 1), from_bytes_unchecked (compressed, 0) or
 * from_uncompressed_bytes (uncompressed, 0): entry first_index + i holds key i; errs[i] = MBLS_OK / MBLS_ERR_* exactly as
 * the reference constructor would return. A key that failed is stored as an invalid entry: every item that names it is
 * rejected with MBLS_ST_BAD_PK_ENCODING (the reference caller would hold no PublicKey to pass). */
int mbls_keytable_append(mbls_keytable* t, const uint8_t* pks, int pk_format, int validate, uint64_t n, uint64_t* first_index, uint8_t* errs);
int mbls_keytable_append_device(mbls_keytable* t, const uint8_t* d_pks, int pk_format, int validate, uint64_t n, uint64_t* first_index,
                                uint8_t* d_errs, void* stream);
/* PublicKey::as_uncompressed_bytes of n consecutive entries (src/keys.rs:163-165); errs[i] = MBLS_ERR_INVALID_POINT for invalid entries */
int mbls_keytable_get(mbls_keytable* t, uint64_t first_index, uint64_t n, uint8_t* pks96, uint8_t* errs);
/* The hot path over table indices: item i uses entries key_idx[k i .. k i + k) (offsets == NULL) or
 * key_idx[offsets[i] .. offsets[i+1]). An index >= mbls_keytable_size counts as an undecodable key. Same results, bitmap and
 * status words as mbls_fast_aggregate_verify_batch over the same keys in wire format. */
int mbls_fast_aggregate_verify_batch_indexed_device(mbls_ctx* ctx, const mbls_keytable* t, const uint8_t* d_sigs, const uint8_t* d_msgs,
                                                    uint32_t msg_len, const uint64_t* d_msg_offsets, const uint32_t* d_key_idx,
                                                    const uint32_t* d_offsets, uint64_t n, uint32_t k,
                                                    uint8_t* d_results, uint64_t* d_bitmap, uint32_t* d_status, void* stream);
int mbls_fast_aggregate_verify_batch_indexed(mbls_ctx* ctx, const mbls_keytable* t, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len,
                                             const uint64_t* msg_offsets, const uint32_t* key_idx, const uint32_t* offsets, uint64_t n,
                                             uint32_t k, uint8_t* results, uint32_t* status);

/* ---- several GPUs behind one handle ------------------------------------------------------------------------
 * Items are independent (reference src/aggregates.rs:177-215 keeps no state between calls), so a batch shards embarrassingly:
 * device g of G verifies items [n g / G, n (g + 1) / G). One context and one host thread per listed device; every thread stages its
 * own shard from the caller's buffers and writes its results into them in place (in a one-process-per-GPU deployment -- bench.py --
 * the same partition is milagro_bls_amd/shard.py and the accept bitmap is gathered with RCCL through torch.distributed).
 * THE EXCHANGE STEPS of the handle -- the packed accept bitmap of mbls_multi_fast_aggregate_verify_bitmap, the partial records of
 * mbls_multi_verify_multiple_aggregate_signatures -- are RCCL all-gathers between the devices' buffers (over xGMI on an MI355X node): the
 * handle opens librccl.so.1 at run time (no link-time dependency) and makes one communicator over its devices (ncclCommInitAll). Where
 * that is not possible -- RCCL absent, the same device listed twice (RCCL wants one rank per device), MBLS_MULTI_NO_RCCL set -- the same
 * records travel through host memory instead: same results, never a restart; mbls_multi_rccl_active / mbls_multi_exchange_note tell which.
 * A device id may be listed more than once (two contexts then share that GPU). Calls on one handle are serialised. */
typedef struct mbls_multi mbls_multi;
int mbls_multi_create(mbls_multi** out, const int* device_ids, int n_devices);
void mbls_multi_destroy(mbls_multi* m);
int mbls_multi_device_count(const mbls_multi* m);
const char* mbls_multi_last_error(mbls_multi* m);
mbls_ctx* mbls_multi_context(mbls_multi* m, int i);            /* the i-th device's context (for the scalar API, reserve, ...) */
int mbls_multi_reserve(mbls_multi* m, uint64_t max_items);     /* workspace for batches of up to max_items items in total */
int mbls_multi_rccl_active(const mbls_multi* m);               /* 1: exchange steps are RCCL all-gathers between the devices; 0: through host memory */
const char* mbls_multi_exchange_note(mbls_multi* m);           /* which, and why (e.g. "host join: device 0 is listed more than once ...") */
/* same arguments, results and status words as mbls_fast_aggregate_verify_batch / mbls_verify_batch (host buffers) */
int mbls_multi_fast_aggregate_verify_batch(mbls_multi* m, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len,
                                           const uint64_t* msg_offsets, const uint8_t* pks, int pk_format, const uint32_t* pk_offsets,
                                           uint64_t n, uint32_t k, uint8_t* results, uint32_t* status);
int mbls_multi_verify_batch(mbls_multi* m, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* msg_offsets,
                            const uint8_t* pks, int pk_format, uint64_t n, uint8_t* results, uint32_t* status);
/* n x fast_aggregate_verify with the results as ONE packed accept bitmap that EVERY device of the handle ends up holding (bit i % 64 of word i / 64 =
 * item i): device g verifies the items of words [g W, (g + 1) W), W = ceil(ceil(n / 64) / G), packs them on the device, and the words are all-gathered
 * between the devices (RCCL when active, see above). `bitmap` (host, ceil(n / 64) words; may be NULL) receives the first device's copy;
 * mbls_multi_device_bitmap(m, g) is device g's own copy (a device pointer to G W words, valid until the next call on the handle). Other arguments
 * as mbls_multi_fast_aggregate_verify_batch. */
int mbls_multi_fast_aggregate_verify_bitmap(mbls_multi* m, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len,
                                            const uint64_t* msg_offsets, const uint8_t* pks, int pk_format, const uint32_t* pk_offsets,
                                            uint64_t n, uint32_t k, uint64_t* bitmap, uint32_t* status);
const uint64_t* mbls_multi_device_bitmap(mbls_multi* m, int g);
/* verify_multiple_aggregate_signatures over the devices of the handle: device g runs sets [n g / G, n (g + 1) / G) up to its partial record
 * (mbls_verify_multiple_partial_device), the G records are all-gathered between the devices (RCCL when active), the first device joins them and
 * runs the tail: the same bool as mbls_verify_multiple_aggregate_signatures on one device, same arguments. */
int mbls_multi_verify_multiple_aggregate_signatures(mbls_multi* m, const uint8_t* sigs96, const uint8_t* apks96, const uint8_t* msgs,
                                                    uint32_t msg_len, const uint64_t* msg_offsets, const uint64_t* rands, size_t n);
/* a key table replicated on every device of the handle: same indices everywhere */
typedef struct mbls_multi_keytable mbls_multi_keytable;
int mbls_multi_keytable_create(mbls_multi* m, uint64_t capacity_hint, mbls_multi_keytable** out);
void mbls_multi_keytable_destroy(mbls_multi_keytable* t);
uint64_t mbls_multi_keytable_size(const mbls_multi_keytable* t);
/* All or nothing: when one device fails (or the replicas disagree) the replicas that did append drop the new records again, the
 * indices stay the same on every device and the error names the device (mbls_multi_last_error). */
int mbls_multi_keytable_append(mbls_multi_keytable* t, const uint8_t* pks, int pk_format, int validate, uint64_t n, uint64_t* first_index, uint8_t* errs);
/* replica i of the table (the table of mbls_multi_context(m, i)); owned by the handle */
mbls_keytable* mbls_multi_keytable_replica(mbls_multi_keytable* t, int i);
int mbls_multi_fast_aggregate_verify_batch_indexed(mbls_multi* m, const mbls_multi_keytable* t, const uint8_t* sigs, const uint8_t* msgs,
                                                   uint32_t msg_len, const uint64_t* msg_offsets, const uint32_t* key_idx, const uint32_t* offsets,
                                                   uint64_t n, uint32_t k, uint8_t* results, uint32_t* status);

/* ---- scalar API, 1:1 with the reference's methods (each runs the batch kernels with n = 1) ---- */
/* PublicKey::from_bytes (src/keys.rs:140-147): compressed decode + KeyValidate -> 96-byte decoded key */
int mbls_pk_from_bytes(mbls_ctx* ctx, const uint8_t* bytes, size_t len, uint8_t pk_out[96]);
/* PublicKey::from_bytes_unchecked (src/keys.rs:150-155) */
int mbls_pk_from_bytes_unchecked(mbls_ctx* ctx, const uint8_t* bytes, size_t len, uint8_t pk_out[96]);
/* PublicKey::from_uncompressed_bytes (src/keys.rs:170-175) */
int mbls_pk_from_uncompressed_bytes(mbls_ctx* ctx, const uint8_t* bytes, size_t len, uint8_t pk_out[96]);
/* PublicKey::as_bytes (src/keys.rs:158-160) */
int mbls_pk_as_bytes(mbls_ctx* ctx, const uint8_t pk[96], uint8_t out[48]);
/* PublicKey::key_validate (src/keys.rs:181-186) -> 1/0 */
int mbls_pk_key_validate(mbls_ctx* ctx, const uint8_t pk[96]);
/* PublicKey::from_secret_key (src/keys.rs:124-137); sk range is checked like SecretKey::from_bytes (src/keys.rs:80-82) */
int mbls_pk_from_secret_key(mbls_ctx* ctx, const uint8_t* sk, size_t sk_len, uint8_t pk_out[96]);
/* Signature::from_bytes / AggregateSignature::from_bytes (src/signature.rs:43-46, src/aggregates.rs:319-322) */
int mbls_sig_from_bytes(mbls_ctx* ctx, const uint8_t* bytes, size_t len, uint8_t sig_out[96]);
/* Signature::new (src/signature.rs:17-21) */
int mbls_sign(mbls_ctx* ctx, const uint8_t* msg, size_t msg_len, const uint8_t* sk, size_t sk_len, uint8_t sig_out[96]);
/* Signature::verify (src/signature.rs:27-40) -> 1/0 */
int mbls_verify(mbls_ctx* ctx, const uint8_t sig[96], const uint8_t* msg, size_t msg_len, const uint8_t pk[96]);
/* AggregatePublicKey::aggregate / into_aggregate (src/aggregates.rs:29-56): n decoded keys -> decoded aggregate */
int mbls_aggregate_public_keys(mbls_ctx* ctx, const uint8_t* pks96, size_t n, uint8_t apk_out[96]);
/* AggregatePublicKey::add / add_aggregate (src/aggregates.rs:68-77) */
int mbls_aggregate_public_key_add(mbls_ctx* ctx, const uint8_t a[96], const uint8_t b[96], uint8_t out[96]);
/* AggregateSignature::add / add_aggregate (src/aggregates.rs:114-124); AggregateSignature::new() is 0xC0||0.. */
int mbls_aggregate_signature_add(mbls_ctx* ctx, const uint8_t a[96], const uint8_t b[96], uint8_t out[96]);
/* AggregateSignature::fast_aggregate_verify (src/aggregates.rs:177-215) -> 1/0 */
int mbls_fast_aggregate_verify(mbls_ctx* ctx, const uint8_t sig[96], const uint8_t* msg, size_t msg_len,
                               const uint8_t* pks96, size_t n_pks);
/* AggregateSignature::fast_aggregate_verify_pre_aggregated (src/aggregates.rs:223-253) -> 1/0 */
int mbls_fast_aggregate_verify_pre_aggregated(mbls_ctx* ctx, const uint8_t sig[96], const uint8_t* msg, size_t msg_len,
                                              const uint8_t apk[96]);
/* AggregateSignature::aggregate_verify (src/aggregates.rs:130-170): n messages of msg_lens[i] bytes, concatenated */
int mbls_aggregate_verify(mbls_ctx* ctx, const uint8_t sig[96], const uint8_t* msgs, const size_t* msg_lens, size_t n_msgs,
                          const uint8_t* pks96, size_t n_pks);
/* n x AggregateSignature::aggregate_verify (src/aggregates.rs:130-170) in one call. The (message, key) pairs of all items lie back to back:
 * pair j = (message j, key pks96 + 96 j); messages are msg_len bytes each or msgs[msg_offsets[j] .. msg_offsets[j+1]) (total_pairs + 1
 * offsets); item i owns the pairs [pair_offsets[i], pair_offsets[i+1]) (n + 1 offsets starting at 0) or k each (pair_offsets == NULL) --
 * "as many messages as keys" (src/aggregates.rs:131) holds by construction, an item without pairs is false (MBLS_ST_NO_KEYS). results[i] =
 * 1/0, status[i] = the MBLS_ST_* bits of the item (its signature's and its pairs' ORed). One lane per pair walks the key decode, the
 * message phase and a one-pair Miller loop; the (sig_i, -G1) pairs ride the same launch; a product tree per item; one final exponentiation
 * per item. The device entry only enqueues (total_pairs = pair_offsets[n] must be given: it sizes the launches). */
int mbls_aggregate_verify_batch(mbls_ctx* ctx, const uint8_t* sigs96, const uint8_t* msgs, uint32_t msg_len, const uint64_t* msg_offsets,
                                const uint8_t* pks96, const uint32_t* pair_offsets, uint32_t k, uint64_t n, uint8_t* results, uint32_t* status);
int mbls_aggregate_verify_batch_device(mbls_ctx* ctx, const uint8_t* d_sigs96, const uint8_t* d_msgs, uint32_t msg_len,
                                       const uint64_t* d_msg_offsets, const uint8_t* d_pks96, const uint32_t* d_pair_offsets, uint32_t k,
                                       uint64_t total_pairs, uint64_t n, uint8_t* d_results, uint32_t* d_status, void* stream);
/* AggregateSignature::verify_multiple_aggregate_signatures (src/aggregates.rs:261-316): n sets of
 * (aggregate signature, aggregate public key, message); rands[i] = the NONZERO blinding scalars (63 bits in the
 * reference) drawn from the caller's RNG exactly as at src/aggregates.rs:280-287 -- the reference owns that loop, here
 * the caller does. The scalars are the security of the batch check: rands == NULL is MBLS_ERR_ARGUMENT, and a zero
 * scalar (which would drop its set from the check) makes the check fail: 0 from the bool form; from the *_device forms
 * *d_result = 0 and MBLS_ST_BAD_SCALAR in the status word. One bool for the whole batch.
 * The *_device forms ONLY ENQUEUE (no host synchronisation anywhere): *d_result (one byte of device memory) receives 1 / 0,
 * *d_status (optional, one device word) the OR of the MBLS_ST_* bits of all sets -- a signature outside G2 (the reference's early
 * `return false`, src/aggregates.rs:274-276), an undecodable member or a zero scalar give 0 through it. */
int mbls_verify_multiple_aggregate_signatures(mbls_ctx* ctx, const uint8_t* sigs96, const uint8_t* apks96,
                                              const uint8_t* msgs, uint32_t msg_len, const uint64_t* msg_offsets,
                                              const uint64_t* rands, size_t n);
int mbls_verify_multiple_aggregate_signatures_device(mbls_ctx* ctx, const uint8_t* d_sigs96, const uint8_t* d_apks96,
                                              const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_msg_offsets,
                                              const uint64_t* d_rands, uint64_t n, uint8_t* d_result, uint32_t* d_status,
                                              void* stream);
/* The reference's own shape, generator included (src/aggregates.rs:261-316): its loop tests set i's signature for the subgroup
 * (:272-275) BEFORE it draws rand[i] from the caller's rng (:280-287) and returns at the first signature outside G2, so a rejected
 * batch leaves the rng after exactly as many draws as sets came before the bad one. This entry keeps that order in ONE call: the
 * signatures are decoded and tested first (beside the message phase), the host reads the verdicts, `draw(user, out, count)` is
 * called at most once for the `count` scalars of the sets in front of the first bad signature (count = n when there is none;
 * not called for count = 0 or n = 0) and must fill out[0 .. count) with NONZERO scalars in set order; what follows does not repeat
 * the subgroup test. Same bool as mbls_verify_multiple_aggregate_signatures with the same scalars. `draw` runs on the calling
 * thread while the context is locked and the call's staging buffers are in use: it must not call back into the library with
 * this context (another context is fine). include/milagro_bls.hpp, rust/src/lib.rs and milagro_bls_amd/api.py draw as :280-287 does. */
typedef void (*mbls_scalar_source)(void* user, uint64_t* out, uint64_t count);
int mbls_verify_multiple_aggregate_signatures_rng(mbls_ctx* ctx, const uint8_t* sigs96, const uint8_t* apks96,
                                              const uint8_t* msgs, uint32_t msg_len, const uint64_t* msg_offsets,
                                              size_t n, mbls_scalar_source draw, void* user);
/* The several-device form (mbls_multi_verify_multiple_aggregate_signatures) with the reference's RNG order and no second subgroup test (what mbls_verify_multiple_aggregate_signatures_rng is to one device): every device decodes and
 * tests its shard's signatures first, the host finds the first bad signature of the WHOLE batch, `draw` is asked ONCE for the scalars of the sets in front of it
 * (reference src/aggregates.rs:272-287) and -- every signature good -- the devices go on from the points they hold to their records, the exchange and the join. */
int mbls_multi_verify_multiple_aggregate_signatures_rng(mbls_multi* m, const uint8_t* sigs96, const uint8_t* apks96, const uint8_t* msgs,
                                                        uint32_t msg_len, const uint64_t* msg_offsets, size_t n, mbls_scalar_source draw, void* user);

/* The same for sets given by their keys in wire format (BASELINE configs[3]: 2^14 sets x 128 keys): set i owns k keys
 * (or [pk_offsets[i], pk_offsets[i+1])), AggregatePublicKey::aggregate (src/aggregates.rs:29-39) runs on the device first. */
int mbls_verify_multiple_sets_device(mbls_ctx* ctx, const uint8_t* d_sigs96, const uint8_t* d_pks, int pk_format,
                                     const uint32_t* d_pk_offsets, uint32_t k, const uint8_t* d_msgs, uint32_t msg_len,
                                     const uint64_t* d_msg_offsets, const uint64_t* d_rands, uint64_t n, uint8_t* d_result,
                                     uint32_t* d_status, void* stream);

/* The same for sets named by indices into a resident key table (mbls_keytable_*; the deployment's form: a set is a list of validator indices):
 * set i owns d_key_idx[d_offsets[i] .. d_offsets[i+1]) or k indices each; an index outside the table or an invalid record rejects the check
 * (MBLS_ST_BAD_PK_ENCODING in the status word). With d_partial != NULL the call produces the shard record of the section below instead of
 * d_result / d_status (which may then be NULL). Enqueues only. */
int mbls_verify_multiple_sets_indexed_device(mbls_ctx* ctx, const mbls_keytable* t, const uint8_t* d_sigs96, const uint32_t* d_key_idx,
                                             const uint32_t* d_offsets, uint32_t k, const uint8_t* d_msgs, uint32_t msg_len,
                                             const uint64_t* d_msg_offsets, const uint64_t* d_rands, uint64_t n, uint8_t* d_result,
                                             uint32_t* d_status, uint8_t* d_partial, void* stream);

/* verify_multiple over several devices or processes ("one exchange step"; the reference's function is one loop over
 * one iterator, src/aggregates.rs:261-316 -- the product of pairings and the sum of blinded signatures it accumulates are associative, so
 * the sets may be cut into shards): every participant runs mbls_verify_multiple_partial_device over ITS sets and gets one
 * MBLS_VM_PARTIAL_BYTES record (the shard's Miller product, its sum of [r_i] sig_i, the OR of its status words; opaque, in the library's
 * own number format: exchange it only between builds of the same library); the records are exchanged (RCCL all-gather of G x 896 bytes, or
 * through the host), and mbls_verify_multiple_finish_device joins G records -- any G >= 0, in any order as long as every participant uses
 * the same -- into the bool the one-device call returns for the concatenated sets. Keys: d_apks96 (one aggregate key per set) or, when
 * that is NULL, d_pks / pk_format / d_pk_offsets / k as in mbls_verify_multiple_sets_device. An empty shard (n = 0) is a valid
 * participant. Both entries only enqueue. */
#define MBLS_VM_PARTIAL_BYTES 896
int mbls_verify_multiple_partial_device(mbls_ctx* ctx, const uint8_t* d_sigs96, const uint8_t* d_apks96, const uint8_t* d_pks, int pk_format,
                                        const uint32_t* d_pk_offsets, uint32_t k, const uint8_t* d_msgs, uint32_t msg_len,
                                        const uint64_t* d_msg_offsets, const uint64_t* d_rands, uint64_t n, uint8_t* d_partial, void* stream);
int mbls_verify_multiple_finish_device(mbls_ctx* ctx, const uint8_t* d_partials, uint64_t n_partials, uint8_t* d_result, uint32_t* d_status,
                                       void* stream);

/* ---- batch helpers used to build inputs and caches on the device ---- */
/* n x PublicKey::from_bytes[_unchecked] / from_uncompressed_bytes: errs[i] = MBLS_OK / MBLS_ERR_* per key */
int mbls_pk_decode_batch(mbls_ctx* ctx, const uint8_t* in, int in_format, int validate, uint64_t n, uint8_t* out96, uint8_t* errs);
int mbls_pk_compress_batch(mbls_ctx* ctx, const uint8_t* in96, uint64_t n, uint8_t* out48, uint8_t* errs);
/* n x Signature::from_bytes: errs[i]; in_g2 (optional) = subgroup_check_g2 per signature */
int mbls_sig_check_batch(mbls_ctx* ctx, const uint8_t* in96, uint64_t n, uint8_t* errs, uint8_t* in_g2);
/* SECRET KEYS ON THE DEVICE (signing, sk -> pk; reference src/signature.rs:17-21, src/keys.rs:124-137 -- amcl's g1mul / g2mul select table entries in
 * constant time). Every table lookup that depends on a key is a SCAN WITH SELECTION: signing reads all eight records of the lane's window table in every
 * window and keeps its own with v_cndmask (the generated routine's constant-time form, tools/gen_tower_d.py blind_scan_ct), sk -> pk reads all 16 multiples
 * [d 16^j] G1 of every window and keeps record d_j (k_sk_select) -- no address, no instruction stream and no memory-operation count depends on a key; the
 * scalar's other uses (digit extraction, sign / zero handling) are selections as well. mbls_ctx_set_secret_ops(ctx, 1) (environment MBLS_UNSAFE_SECRET_OPS
 * for new contexts) switches both to the faster forms that read ONE record at a key-dependent address -- for building test and bench inputs from throw-away
 * keys only (measured at 2^16: signing 14.5 instead of 15.3 ms, sk -> pk 0.9 instead of 2.7 ms). This is NOT a claim of resistance against power or fault analysis, and a GPU shared with an attacker's kernels
 * is not a place for long-term keys either way. What the calls leave behind is wiped: the staged keys, the digit / selection buffers and the workspace slots
 * of the partial products and tables are zeroed on the stream before the call's workspace is released.
 * n x Signature::new / PublicKey::from_secret_key. Secret keys are NOT range-checked here: any 32-byte big-endian value gives [sk mod r] H(msg) /
 * [sk mod r] G1. The device entries use the context's workspace (four items per signature, in chunks of 65 536 signatures) and wait for its
 * previous user like the verification entries; they only enqueue. */
int mbls_sign_batch(mbls_ctx* ctx, const uint8_t* sks32, const uint8_t* msgs, uint32_t msg_len, uint64_t n, uint8_t* sigs96);
int mbls_sign_batch_device(mbls_ctx* ctx, const uint8_t* d_sks32, const uint8_t* d_msgs, uint32_t msg_len, uint64_t n, uint8_t* d_sigs96, void* stream);
int mbls_sk_to_pk_batch(mbls_ctx* ctx, const uint8_t* sks32, int out_format, uint64_t n, uint8_t* pks);
int mbls_sk_to_pk_batch_device(mbls_ctx* ctx, const uint8_t* d_sks32, int out_format, uint64_t n, uint8_t* d_pks, void* stream);
/* n x hash_to_curve_g2 (src/amcl_utils.rs:33-35), compressed output */
int mbls_hash_to_g2_batch(mbls_ctx* ctx, const uint8_t* msgs, uint32_t msg_len, uint64_t n, uint8_t* out96);
/* the same through the verification pipeline's own message phase, for the parity tests: mode 0 = the stand-alone lane body (as above),
 * 1 = the generated one-lane-per-item routine of the batch path, 2 = the one-wave-per-item program of small batches, 3 = two lanes per
 * message (the form batches between the wave engine's limit and half a round take) */
int mbls_hash_to_g2_batch_mode(mbls_ctx* ctx, const uint8_t* msgs, uint32_t msg_len, uint64_t n, uint8_t* out96, int mode);
/* n x AggregateSignature::aggregate (src/aggregates.rs:100-106): set i sums its k signatures (or the signatures
 * [offsets[i], offsets[i+1]) of sigs96), starting from infinity (an empty set gives 0xC0 || 0..). errs[i] = MBLS_OK or the
 * Signature::from_bytes error of the first member that does not decode. No subgroup check, like the reference. */
int mbls_aggregate_signatures_batch(mbls_ctx* ctx, const uint8_t* sigs96, const uint32_t* offsets, uint64_t n, uint32_t k, uint8_t* out96, uint8_t* errs);
int mbls_aggregate_signatures_batch_device(mbls_ctx* ctx, const uint8_t* d_sigs96, const uint32_t* d_offsets, uint64_t n_sets, uint32_t k,
                                           uint64_t total_sigs, uint8_t* d_out96, uint8_t* d_errs, void* stream);
/* n x AggregatePublicKey::aggregate over wire-format keys -> decoded aggregate keys */
int mbls_aggregate_public_keys_batch(mbls_ctx* ctx, const uint8_t* pks, int pk_format, const uint32_t* pk_offsets,
                                     uint64_t n, uint32_t k, uint8_t* apks96, uint32_t* status);
/* field probe for the parity tests of the hand-written routines, on canonical 48-byte big-endian values. op: 0 out = a*b,
 * 1 a^2, 2 Fp2 product and 3 Fp2 square over element pairs (2i, 2i+1) = (real, imaginary), 4 a^-1 (0 -> 0),
 * 5 a^((p-3)/4), 6 the paired-product routine on elements 2i and 2i+1 */
int mbls_fp_mul_batch(mbls_ctx* ctx, const uint8_t* a48, const uint8_t* b48, uint64_t n, uint8_t* out48, int op);
/* integer-ALU calibration: runs `iters` dependent Fp multiplications per lane on n lanes, returns elapsed ms */
int mbls_fp_mul_bench(mbls_ctx* ctx, uint64_t n_lanes, uint32_t iters, float* ms_out);

/* VALU issue-rate calibration for bench.py: every SIMD runs waves_per_simd (1..8) waves of `iters` x 128 instructions; mode 0:
 * v_mad_u64_u32, mode 1: v_add_co/v_addc chains. Returns elapsed ms (rate = waves_per_simd * iters * 128 / ms per SIMD).
 * mode 2: `iters` x 8 calls' worth of the generated Fp2 product routine inlined back to back (8 x 1 281 instructions per iteration, 980
 * multiply-accumulates each; use waves_per_simd <= 4): the rate of a kernel made of nothing but products. mode 3: the same with the paired
 * Fp product of the key-sum routines (8 x 923 instructions, 784 multiply-accumulates each). modes 4-6 (scripts/dbg/class_vs_mix.py): 8 multiply-accumulates + 4 plain
 * operations interleaved / in two blocks (192 instructions per iteration), the plain operations alone (128); mode 7: v_mad_i64_i32 on eight accumulators (128). */
int mbls_valu_bench(mbls_ctx* ctx, int mode, uint32_t waves_per_simd, uint32_t iters, float* ms_out);

/* ---- instrumentation: per-kernel HIP-event timing of the last *_device verify call (ms), for bench.py ---- */
#define MBLS_N_PHASES 6
int mbls_enable_phase_timing(mbls_ctx* ctx, int on);
int mbls_last_phase_ms(mbls_ctx* ctx, float ms[MBLS_N_PHASES]);   /* aggregate, sig, hash, miller, final, pack */

#ifdef __cplusplus
}
#endif
#endif
