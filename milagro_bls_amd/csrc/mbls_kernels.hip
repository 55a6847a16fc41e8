// mbls_kernels.hip -- gfx950 kernels and the C ABI of libmbls_hip.so (see include/mbls.h).
//
// One item per lane, one wave (64 lanes) per workgroup: the kernels are long integer carry-chain programs
// with no intra-workgroup cooperation, so single-wave workgroups give the dispatcher the finest granularity
// to fill 256 CUs x 4 SIMDs. The pipeline is split into phase kernels that hand their state over through a
// limb-major struct-of-arrays workspace in HBM (mbls_lanes.h); each phase is thousands of Fp
// multiplications per lane, so the hand-over traffic (<= 1.2 KB per item per phase) and the launch gaps are
// noise. No MFMA (carry-chain integer work). LDS is not used for tiling (no data is shared between lanes) but as the
// home of each lane's loop-carried state in k_miller / k_final (36 KB per wave, lane-interleaved so that a wave's access is
// conflict-free), which keeps that state out of the register file between uses and off the HBM-backed stack.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <new>
#include <mutex>
#include <vector>
#include <dlfcn.h>
#include <unistd.h>
#include "mbls_ops.h"
#include "mbls_coop.h"
#include "../../include/mbls.h"

// The product library exists only with the generated routines: the host side below selects kernels (the fused subgroup verdict of
// k_miller / k_sig_verdict, k_blind_*_d, the tree levels, k_coop) whose bodies ARE those routines. The development switches that used to
// replace them by the compiled lane bodies would leave empty or unfused kernels behind the same launches -- wrong answers, not slower
// ones -- so such a build is refused. (The compiled lane bodies remain what tests/host_emul compiles as plain C++.)
#if defined(MBLS_NO_ASM) || defined(MBLS_NO_FP2_ASM) || defined(MBLS_NO_LDS_STATE)
#error "libmbls_hip.so cannot be built with MBLS_NO_ASM / MBLS_NO_FP2_ASM / MBLS_NO_LDS_STATE: the host side depends on the generated routines"
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !MBLS_DEVICE_ASM
#error "device pass without the generated routines"
#endif

// RCCL: the library is opened at run time (mbls_multi_create), never linked -- and its headers are not needed to BUILD either: the few opaque types and enumerators
// the dlopen'ed entry points take are declared here when <rccl/rccl.h> is absent (values are RCCL's / NCCL's stable ABI: ncclSuccess = 0, ncclUint8 = 1)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
#endif

#define MBLS_SLOT_S 25            // 6 Fp: Jacobian G2 accumulator for verify_multiple
#define MBLS_SLOT_T 31            // 12 Fp: the running points of the generated Miller loop (packed, 2^392 domain; tools/gen_tower_d.py T_SLOT)
#define MBLS_SLOT_G2TMP 43        // 6 Fp: scratch of the generated subgroup-test routine (tools/gen_tower_d.py G2_SLOTS)
#define MBLS_SLOT_KREC 49         // 60 Fp: the six compressed powers of the final exponentiation (tools/gen_tower_d.py K_SLOT, K_REC); verify_multiple's
                                  // signature phase keeps its table of 1..8 times the signature in 49..96 (BL_TAB)
#define MBLS_SLOT_G1TAB 109       // 24 Fp: verify_multiple's table of 1..8 times the aggregate key (tools/gen_tower_d.py G1B_TAB)
#define MBLS_SLOT_TOTAL 133
#define WG 64
// The pipeline kernels are built for one wave per SIMD (512 registers per lane): a batch of 2^16 items is exactly one wave
// per SIMD on 256 CUs, and the hot loops are generated straight-line routines that already issue at the VALU rate with a
// single wave, so a 256-register / two-wave variant is slower at every batch size (larger batches
// simply run as successive rounds of workgroups).
#define MBLS_LB __launch_bounds__(WG, 1)

static __device__ __forceinline__ uint64_t gid() { return (uint64_t)blockIdx.x * WG + threadIdx.x; }

// ------------------------------------------------------------------------------------------------ pipeline kernels
__global__ void MBLS_LB k_aggregate(mbls_ws ws, const uint8_t* pks, const uint32_t* offsets, uint32_t k, int fmt, int mode,
                                                   uint32_t* status, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    const uint32_t pkb = fmt == MBLS_PK_COMPRESSED ? 48u : 96u;
    uint64_t first = offsets ? offsets[i] : (uint64_t)k * i; uint32_t cnt = offsets ? offsets[i + 1] - offsets[i] : k;
    uint32_t bad = 0;
    if (offsets && offsets[i + 1] < offsets[i]) { cnt = 0; bad = MBLS_ST_BAD_PK_ENCODING; }     // a non-monotonic offset table never becomes a read
    uint32_t st; lane_aggregate(ws, i, pks + pkb * first, cnt, fmt, mode, &st); st |= bad; if (st) atomicOr(status + i, st);
}
// the generated routines (96-byte keys at 4-byte aligned addresses; table indices)
__global__ void MBLS_LB k_aggregate_raw_d(mbls_ws ws, const uint8_t* pks, const uint32_t* offsets, uint32_t k, int mode, uint32_t* status, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    uint64_t first = offsets ? offsets[i] : (uint64_t)k * i; uint32_t cnt = offsets ? offsets[i + 1] - offsets[i] : k;
    uint32_t bad = 0;
    if (offsets && offsets[i + 1] < offsets[i]) { cnt = 0; bad = MBLS_ST_BAD_PK_ENCODING; }
#if MBLS_DEVICE_ASM
    uint32_t st = lane_aggregate_d<false>(ws, i, pks + 96 * first, cnt, mode, threadIdx.x) | bad;
    if (st) atomicOr(status + i, st);
#endif
}
__global__ void MBLS_LB k_aggregate_indexed_d(mbls_ws ws, const uint32_t* recs, uint64_t tsize, const uint32_t* idx, const uint32_t* offsets, uint32_t k,
                                              int mode, uint32_t* status, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    uint64_t first = offsets ? offsets[i] : (uint64_t)k * i; uint32_t cnt = offsets ? offsets[i + 1] - offsets[i] : k;
    uint32_t bad = 0;
    if (offsets && offsets[i + 1] < offsets[i]) { cnt = 0; bad = MBLS_ST_BAD_PK_ENCODING; }
#if MBLS_DEVICE_ASM
    uint32_t st = lane_aggregate_d<true>(ws, i, idx + first, cnt, mode, threadIdx.x, recs, tsize) | bad;
    if (st) atomicOr(status + i, st);
#endif
}
// one key per lane, nothing but a square root: with the 8-entry window table (112 AGPRs) the kernel fits 256 registers and two
// waves share a SIMD -- the plain 32-bit third of the instruction stream then issues at twice the rate (profiles/r02_ubench.txt)
// Small batches (the wave engine's: latency): the key sum of an item is cut into MBLS_KEY_SPLIT partial sums on lanes of their own (k_aggregate_raw_d /
// k_aggregate_indexed_d over n * MBLS_KEY_SPLIT sub-items, whose keys lie back to back exactly like items of k / MBLS_KEY_SPLIT keys; their sums in the APK slots of
// workspace items n + i * MBLS_KEY_SPLIT + q, their status words in st_sub) and this kernel adds the partial sums of item i -- AggregatePublicKey::aggregate
// (reference src/aggregates.rs:29-39) sums from infinity in any order to the same point --, folds the status words and applies the apk = infinity test.
#define MBLS_KEY_SPLIT 8u
__global__ void MBLS_LB k_apk_combine(mbls_ws ws, uint64_t n, int mode, const uint32_t* st_sub, uint32_t* status) {
    uint64_t i = gid(); if (i >= n) return;
    const uint64_t t0 = n + i * MBLS_KEY_SPLIT;
    g1j acc; acc.x = ws_ld(ws, MBLS_SLOT_APK, t0); acc.y = ws_ld(ws, MBLS_SLOT_APK + 1, t0); acc.z = ws_ld(ws, MBLS_SLOT_APK + 2, t0);
    uint32_t st = st_sub[i * MBLS_KEY_SPLIT];
    for (uint32_t q = 1; q < MBLS_KEY_SPLIT; q++) {
        g1j p; p.x = ws_ld(ws, MBLS_SLOT_APK, t0 + q); p.y = ws_ld(ws, MBLS_SLOT_APK + 1, t0 + q); p.z = ws_ld(ws, MBLS_SLOT_APK + 2, t0 + q);
        g1_add(&acc, &acc, &p);
        st |= st_sub[i * MBLS_KEY_SPLIT + q];
    }
    if (mode == MBLS_MODE_FAST_AGGREGATE && g1_is_inf(&acc)) st |= MBLS_ST_APK_INFINITY;
    ws_st(ws, MBLS_SLOT_APK, i, acc.x); ws_st(ws, MBLS_SLOT_APK + 1, i, acc.y); ws_st(ws, MBLS_SLOT_APK + 2, i, acc.z);
    if (st) atomicOr(status + i, st);
}
__global__ void __launch_bounds__(WG, 2) k_pk_decompress(const uint8_t* pks48, uint64_t nkeys, uint32_t* keys_xy, uint8_t* flags) {
    uint64_t j = gid(); if (j >= nkeys) return;
    lane_pk_decompress(j, pks48, keys_xy, flags);
}
__global__ void MBLS_LB k_aggregate_decoded(mbls_ws ws, const uint32_t* keys_xy, const uint8_t* flags, uint32_t k, int mode, uint32_t* status, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    uint32_t st; lane_aggregate_decoded(ws, i, keys_xy + 24 * (uint64_t)k * i, flags + (uint64_t)k * i, k, mode, &st); if (st) atomicOr(status + i, st);
}
// Resident key table (the on-device analogue of the decoded PublicKey objects a reference caller holds, src/keys.rs:116-120):
// item i sums the table entries idx[k*i .. k*i+k) (or idx[offsets[i] .. offsets[i+1])).
__global__ void MBLS_LB k_aggregate_indexed(mbls_ws ws, const uint32_t* recs, uint64_t tsize, const uint32_t* idx, const uint32_t* offsets, uint32_t k,
                                            int mode, uint32_t* status, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    uint64_t first = offsets ? offsets[i] : (uint64_t)k * i; uint32_t cnt = offsets ? offsets[i + 1] - offsets[i] : k;
    uint32_t bad = 0;
    if (offsets && offsets[i + 1] < offsets[i]) { cnt = 0; bad = MBLS_ST_BAD_PK_ENCODING; }
    uint32_t st; lane_aggregate_indexed(ws, i, recs, tsize, idx + first, cnt, mode, &st); st |= bad; if (st) atomicOr(status + i, st);
}
// n x PublicKey::from_bytes[_unchecked] / from_uncompressed_bytes into table records (affine Montgomery limbs + flags)
__global__ void MBLS_LB k_keytable_append(const uint8_t* in, int fmt, int validate, uint64_t n, uint32_t* recs, uint8_t* errs) {
    uint64_t i = gid(); if (i < n) op_keytable_append(i, in, fmt, validate, recs, errs);
}
__global__ void MBLS_LB k_keytable_export(const uint32_t* recs, uint64_t first, uint64_t n, uint8_t* out96, uint8_t* errs) {
    uint64_t i = gid(); if (i < n) op_keytable_export(i, recs + 32 * first, out96, errs);
}
// batched AggregateSignature::aggregate (src/aggregates.rs:100-106): decode one signature per lane, then one set per lane
__global__ void MBLS_LB k_g2_decode_affine(const uint8_t* sigs96, uint64_t n, uint32_t* xy, uint8_t* flags) {
    uint64_t i = gid(); if (i < n) op_g2_decode_affine(i, sigs96, xy, flags);
}
__global__ void MBLS_LB k_g2_sum(const uint32_t* xy, const uint8_t* flags, const uint32_t* offsets, uint32_t k, uint64_t n, uint8_t* out96, uint8_t* errs) {
    uint64_t i = gid(); if (i >= n) return;
    uint64_t first = offsets ? offsets[i] : (uint64_t)k * i; uint32_t cnt = offsets ? offsets[i + 1] - offsets[i] : k;
    if (offsets && offsets[i + 1] < offsets[i]) cnt = 0;
    op_g2_sum(i, xy + 48 * first, flags + first, cnt, out96, errs);
}
__global__ void MBLS_LB k_sig(mbls_ws ws, const uint8_t* sigs, uint32_t* status, uint64_t n, int check) {
    __shared__ uint32_t spill[154 * 64];          // 11 spill slots of 14 dwords per lane for the generated subgroup test
    uint64_t i = gid(); if (i >= n) return;
    uint32_t st = 0; lane_sig(ws, i, sigs + 96 * i, &st, (MBLS_LDS uint32_t*)spill, threadIdx.x, true, check != 0);
    if (st) atomicOr(status + i, st);             // may run beside k_aggregate on another stream
}
// item i's message: msgs[mlen i ..] or, with an offset table of n + 1 entries, msgs[moff[i] .. moff[i+1]) (any length below 2^32; a range
// that runs backwards is never read: the item is rejected with MBLS_ST_BAD_MSG_RANGE)
__global__ void MBLS_LB k_hash(mbls_ws ws, const uint8_t* msgs, uint32_t mlen, const uint64_t* moff, uint32_t* status, uint64_t n) {
    __shared__ uint32_t spill[154 * 64];          // the same for the addition / cofactor-clearing routine
    uint64_t i = gid(); if (i >= n) return;
    const uint8_t* m = msgs + (uint64_t)mlen * i; uint32_t len = mlen;
    if (moff) {
        const uint64_t a = moff[i], b = moff[i + 1];
        const bool bad = b < a || b - a > 0xFFFFFFFFull;
        m = msgs + (bad ? 0 : a); len = bad ? 0u : (uint32_t)(b - a);
        if (bad) atomicOr(status + i, MBLS_ST_BAD_MSG_RANGE);
    }
    lane_hash(ws, i, m, len, (MBLS_LDS uint32_t*)spill, threadIdx.x, true);
}
// The message phase with TWO lanes per message (batches of at most half a round): lane t = 2 i + e has workspace item t of its own, runs
// hash_to_field of message i (both lanes: 19 SHA-256 compressions are 2 % of the phase) and ONE map_to_curve -- u_e --, then the even lane adds
// the neighbour's point, clears the cofactor and leaves H in item 2 i; k_h_compact_a / _b move it to item i (through slots 19..24: item i's
// slots 7..12 may still be the working slots of another pair's lane while this kernel runs).
__global__ void MBLS_LB k_hash2(mbls_ws ws, const uint8_t* msgs, uint32_t mlen, const uint64_t* moff, uint32_t* status, uint64_t n) {
    __shared__ uint32_t spill[154 * 64];
    const uint64_t t = gid(); if (t >= 2 * n) return;
    const uint64_t i = t >> 1;
    const uint8_t* m = msgs + (uint64_t)mlen * i; uint32_t len = mlen;
    if (moff) {
        const uint64_t a = moff[i], b = moff[i + 1];
        const bool bad = b < a || b - a > 0xFFFFFFFFull;
        m = msgs + (bad ? 0 : a); len = bad ? 0u : (uint32_t)(b - a);
        if (bad && !(t & 1)) atomicOr(status + i, MBLS_ST_BAD_MSG_RANGE);
    }
#if MBLS_DEVICE_ASM
    hash_fields_to_ws(ws.w, ws.stride, t, m, len, (uint32_t)(t & 1));
    g2_hash2_d_call(ws, t, (MBLS_LDS uint32_t*)spill, threadIdx.x);
#endif
}
__global__ void __launch_bounds__(WG) k_h_compact_a(mbls_ws ws, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    for (int w = 0; w < 72; w++) ws.w[((uint64_t)(19 * 12 + w)) * ws.stride + i] = ws.w[((uint64_t)(MBLS_SLOT_H * 12 + w)) * ws.stride + 2 * i];
}
__global__ void __launch_bounds__(WG) k_h_compact_b(mbls_ws ws, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    for (int w = 0; w < 72; w++) ws.w[((uint64_t)(MBLS_SLOT_H * 12 + w)) * ws.stride + i] = ws.w[((uint64_t)(19 * 12 + w)) * ws.stride + i];
}
// hash_to_field alone (SHA-256 / expand_message_xmd, one lane per item): u0, u1 into slots 31, 32 / 37, 38 -- the input of the cooperative
// engine's hashg2 program (small batches: one WAVE per item walks the maps, the addition and the cofactor clearing)
__global__ void MBLS_LB k_hash_fields(mbls_ws ws, const uint8_t* msgs, uint32_t mlen, const uint64_t* moff, uint32_t* status, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    const uint8_t* m = msgs + (uint64_t)mlen * i; uint32_t len = mlen;
    if (moff) {
        const uint64_t a = moff[i], b = moff[i + 1];
        const bool bad = b < a || b - a > 0xFFFFFFFFull;
        m = msgs + (bad ? 0 : a); len = bad ? 0u : (uint32_t)(b - a);
        if (bad) atomicOr(status + i, MBLS_ST_BAD_MSG_RANGE);
    }
#if MBLS_DEVICE_ASM
    hash_fields_to_ws(ws.w, ws.stride, i, m, len);
#endif
}
__global__ void MBLS_LB k_miller(mbls_ws ws, uint64_t n) {
    // one wave per SIMD = 4 waves per CU: each wave can park 36 KB of loop state in LDS (144 of the 160 KB)
    __shared__ uint32_t tstore[154 * 64];         // 11 spill slots of 14 dwords per lane for the generated loop (the running points are in HBM)
    uint64_t i = gid(); if (i >= n) return;
    lane_miller(ws, i, (MBLS_LDS uint32_t*)tstore, threadIdx.x, true);
}
// between k_miller and k_final when k_sig only decoded: the signature's subgroup test from the loop's running point (lane_sig_verdict)
__global__ void MBLS_LB k_sig_verdict(mbls_ws ws, uint32_t* status, uint64_t n, uint64_t t_item_offset, int t_pair) {
    uint64_t i = gid(); if (i >= n) return;
    const uint32_t st = lane_sig_verdict(ws, i, i + t_item_offset, t_pair);
    if (st) status[i] |= st;
}
// Batches that fill at most half of the SIMDs (2 n <= one round): the two pairs of item i on TWO lanes -- lane i walks (H_i, apk_i), lane n + i
// walks (sig_i, -G1) -- each with the generated one-pair loop (6.6 ms instead of 11.3 for the two-pair loop), workspace item = lane. The
// product f_i f_(n+i) is one level of the generated product tree (k_f12_tree_d(2 n, n)), and the signature's subgroup verdict reads the running
// point of lane n + i (k_sig_verdict with t_item_offset = n, t_pair = 1): the one-pair routine walks the same bits of |x| with the same
// incomplete formulas (Q in homogeneous form with Z = 1), so it ends with [|x|] sig exactly like pair 0 of the two-pair loop.
// lpp = 2 (batches of at most a quarter of a round): TWO lanes per pair -- lanes 2 q, 2 q + 1 walk pair q together, their products in pairs
// (miller_loop_single_pair_d): four lanes per item.
template <int LPP> MBLS_FN void miller_split_body(const mbls_ws& ws, uint64_t n, uint32_t* spill) {
    const uint64_t lane_id = gid();
    uint64_t t = LPP == 2 ? lane_id >> 1 : lane_id; if (t >= 2 * n) return;
    const bool sigpair = t >= n;                                   // branch-free operand selection: the routine is called from uniform control flow
    const uint64_t i = sigpair ? t - n : t;
    g2j h; h.x = ws_ld2(ws, MBLS_SLOT_H, i); h.y = ws_ld2(ws, MBLS_SLOT_H + 2, i); h.z = ws_ld2(ws, MBLS_SLOT_H + 4, i);
    g1j a; a.x = ws_ld(ws, MBLS_SLOT_APK, i); a.y = ws_ld(ws, MBLS_SLOT_APK + 1, i); a.z = ws_ld(ws, MBLS_SLOT_APK + 2, i);
    const fp2 sx = ws_ld2(ws, MBLS_SLOT_SIG, i), sy = ws_ld2(ws, MBLS_SLOT_SIG + 2, i);
    mbls_pair pr, ps;
    pr.skip = g2_is_inf(&h) | g1_is_inf(&a);
    g2h_from_jacobian(&pr.q, &h); g1arg_from_jacobian(&pr.p, &a);
    ps.skip = fp2_is_zero(sy);
    g2h_from_affine(&ps.q, sx, sy); g1arg_from_affine(&ps.p, fp_load_const(MBLS_G1_X), fp_load_const(MBLS_G1_NEG_Y));
    pr.skip = sigpair ? ps.skip : pr.skip;
    pr.q.x = fp2_select(sigpair, ps.q.x, pr.q.x); pr.q.y = fp2_select(sigpair, ps.q.y, pr.q.y); pr.q.z = fp2_select(sigpair, ps.q.z, pr.q.z);
    pr.p.px = fp_select(sigpair, ps.p.px, pr.p.px); pr.p.py = fp_select(sigpair, ps.p.py, pr.p.py); pr.p.pz3 = fp_select(sigpair, ps.p.pz3, pr.p.pz3);
    pr.t = pr.q;
    fp12 f;
#if MBLS_DEVICE_ASM
    if (LPP == 2) miller_loop_single_pair_d(&f, &pr, ws.w, ws.stride, t, (MBLS_LDS uint32_t*)spill, threadIdx.x);
    else miller_loop_single_d(&f, &pr, ws.w, ws.stride, t, (MBLS_LDS uint32_t*)spill, threadIdx.x);
#endif
    const fp2* c = &f.c0.c0;
    for (int s = 0; s < 6; s++) ws_st2(ws, MBLS_SLOT_F + 2 * s, t, c[s]);
}
__global__ void MBLS_LB k_miller_split(mbls_ws ws, uint64_t n) {
    __shared__ uint32_t spill[154 * 64];
    miller_split_body<1>(ws, n, spill);
}
__global__ void MBLS_LB k_miller_split4(mbls_ws ws, uint64_t n) {      // four lanes per item: two per pair
    __shared__ uint32_t spill[154 * 64];
    miller_split_body<2>(ws, n, spill);
}
__global__ void MBLS_LB k_final(mbls_ws ws, uint32_t* status, uint8_t* results, uint64_t n) {
    __shared__ uint32_t accstore[154 * 64];       // spill slots of the generated exponentiation routine / the Fp12 parked by the one-shot products
    uint64_t i = gid(); if (i >= n) return;
    uint32_t st = status[i]; uint8_t r; lane_final(ws, i, &st, &r, (MBLS_LDS uint32_t*)accstore, threadIdx.x, true); status[i] = st; results[i] = r;
}

// The final exponentiation with TWO lanes per item (batches of at most half a round: the SIMDs a one-lane launch would leave idle do half of
// every compressed squaring): lanes 2 j, 2 j + 1 of a workgroup = item 32 blockIdx + j. Both lanes end with the same value; the even one reports.
__global__ void MBLS_LB k_final2(mbls_ws ws, uint32_t* status, uint8_t* results, uint64_t n) {
    __shared__ uint32_t accstore[154 * 64];       // one column per ITEM: the two lanes of an item park identical values in it
    const uint64_t t = gid(); if (t >= 2 * n) return;
    const uint64_t i = t >> 1;
#if MBLS_DEVICE_ASM
    uint32_t st = status[i]; uint8_t r;
    lane_final2(ws, i, &st, &r, (MBLS_LDS uint32_t*)accstore, threadIdx.x);
    if ((t & 1) == 0) { status[i] = st; results[i] = r; }
#endif
}

// accept bitmap: one 64-bit word per wave via ballot
__global__ void MBLS_LB k_pack(const uint8_t* results, uint64_t* bitmap, uint64_t n) {
    uint64_t i = gid();
    bool bit = (i < n) && results[i];
    uint64_t m = __ballot(bit);
    if (threadIdx.x == 0) bitmap[blockIdx.x] = m;
}

// ------------------------------------------------------------------------------------------------ n-pairing kernels
// (aggregate_verify, reference src/aggregates.rs:130-170; verify_multiple, src/aggregates.rs:261-316)
// item i: f_i = Miller(H_i, P_i) with P_i = [r_i] pk_i (r_i = 1 when rands == NULL: aggregate_verify only); S_i = [r_i] sig_i.
// The G1 and the G2 halves are kernels of their own so that they can run side by side on two streams (sets below 2^14 leave most SIMDs
// idle); both OR their bits into status[i] atomically. A zero scalar would drop set i from the check: it is flagged, never used.
// k_blind_g1 (compiled) serves aggregate_verify (rands == NULL: no multiplication); verify_multiple runs the generated k_blind_*_d.
__global__ void MBLS_LB k_blind_g1(mbls_ws ws, const uint8_t* pks96, const uint64_t* rands, uint32_t* status, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    uint32_t st = 0;
    g1j p;
    if (pks96) {
        fp x, y; bool inf; int e = g1_decode_uncompressed(&x, &y, &inf, pks96 + 96 * i);
        if (e) { st |= MBLS_ST_BAD_PK_ENCODING; inf = true; }
        p.x = x; p.y = y; p.z = fp_one(); if (inf) g1_set_inf(&p);
    } else {          // aggregate key left in the workspace by k_aggregate (which already set the status bits)
        p.x = ws_ld(ws, MBLS_SLOT_APK, i); p.y = ws_ld(ws, MBLS_SLOT_APK + 1, i); p.z = ws_ld(ws, MBLS_SLOT_APK + 2, i);
    }
    if (rands) {
        uint32_t k[2] = {(uint32_t)rands[i], (uint32_t)(rands[i] >> 32)};
        if ((k[0] | k[1]) == 0) st |= MBLS_ST_BAD_SCALAR;
        g1_mul(&p, &p, k, 64);
    }
    ws_st(ws, MBLS_SLOT_APK, i, p.x); ws_st(ws, MBLS_SLOT_APK + 1, i, p.y); ws_st(ws, MBLS_SLOT_APK + 2, i, p.z);
    if (st) atomicOr(status + i, st);
}
// [r_i] pk_i with the generated windowed routine (g1_blind_routine); pks96 == NULL: the aggregate key k_aggregate left in slots 0..2
__global__ void MBLS_LB k_blind_g1_d(mbls_ws ws, const uint8_t* pks96, const uint64_t* rands, uint32_t* status, uint64_t n) {
#if MBLS_DEVICE_ASM
    uint64_t i = gid(); if (i >= n) return;
    uint32_t st = 0;
    if (pks96) {
        fp x, y; bool inf; int e = g1_decode_uncompressed(&x, &y, &inf, pks96 + 96 * i);
        if (e) { st |= MBLS_ST_BAD_PK_ENCODING; inf = true; }
        g1j p; p.x = x; p.y = y; p.z = fp_one(); if (inf) g1_set_inf(&p);
        ws_st(ws, MBLS_SLOT_APK, i, p.x); ws_st(ws, MBLS_SLOT_APK + 1, i, p.y); ws_st(ws, MBLS_SLOT_APK + 2, i, p.z);
    }
    const uint64_t r = rands[i];
    if (r == 0) st |= MBLS_ST_BAD_SCALAR;
    g1_blind_d_call(ws, i, threadIdx.x, r);
    if (st) atomicOr(status + i, st);
#endif
}
// the same with the generated routines (decode inlined like k_sig; subgroup test + windowed [r] sig: g2_blind_routine): no lane-private memory
__global__ void MBLS_LB k_blind_sig_d(mbls_ws ws, const uint8_t* sigs96, const uint64_t* rands, uint32_t* status, uint64_t n) {
#if MBLS_DEVICE_ASM
    __shared__ uint32_t spill[154 * 64];
    uint64_t i = gid(); if (i >= n) return;
    uint32_t st = 0;
    bool inf;
    if (sigs96) {
        fp2 x, y;
        int e = g2_decode_compressed_t<true>(&x, &y, &inf, sigs96 + 96 * i);
        if (e) { st |= MBLS_ST_BAD_SIG_ENCODING; inf = true; }
        if (inf) { x = fp2_zero(); y = fp2_zero(); }
        ws_st2(ws, MBLS_SLOT_SIG, i, x); ws_st2(ws, MBLS_SLOT_SIG + 2, i, y);
    } else      // sigs96 == NULL: k_sig decoded AND tested the signatures already (they are in the slots; y = 0: infinity): only [r] sig is left to do
        inf = fp2_is_zero(ws_ld2(ws, MBLS_SLOT_SIG + 2, i));
    const uint64_t r = rands[i];
    if (r == 0) st |= MBLS_ST_BAD_SCALAR;
    // an infinite (or undecodable) signature is (0, 0) in the slots: with the scalar 0 every window digit is 0 and the sum stays at infinity
    const uint32_t fl = g2_blind_d_call(ws, i, (MBLS_LDS uint32_t*)spill, threadIdx.x, inf ? 0 : r, sigs96 ? 0u : 1u);
    if (!((fl & 1u) | inf)) st |= MBLS_ST_SIG_NOT_IN_G2;
    if (st) atomicOr(status + i, st);
#endif
}
// The signature chain of SMALL verify_multiple batches (their Miller loops run on waves: the signatures are the critical path) with TWO lanes per signature:
// lanes 2 j, 2 j + 1 of a workgroup = item 32 blockIdx + j; both decode, both walk the generated routine on identical values, products in pairs.
// k_sig2 = k_sig with the subgroup test (the one-call entry's first phase); k_blind_sig2_d = k_blind_sig_d.
__global__ void MBLS_LB k_sig2(mbls_ws ws, const uint8_t* sigs96, uint32_t* status, uint64_t n) {
#if MBLS_DEVICE_ASM
    __shared__ uint32_t spill[154 * 64];          // one column per ITEM: the two lanes of an item park identical values in it
    const uint64_t t = gid(); if (t >= 2 * n) return;
    const uint64_t i = t >> 1;
    uint32_t st = 0;
    fp2 x, y; bool inf;
    int e = g2_decode_compressed_t<true>(&x, &y, &inf, sigs96 + 96 * i);
    if (e) { st |= MBLS_ST_BAD_SIG_ENCODING; inf = true; }
    if (inf) { x = fp2_zero(); y = fp2_zero(); }
    ws_st2(ws, MBLS_SLOT_SIG, i, x); ws_st2(ws, MBLS_SLOT_SIG + 2, i, y);
    const uint32_t fl = g2_subgroup2_d_call(ws, i, (MBLS_LDS uint32_t*)spill, threadIdx.x);
    if (!((fl & 1u) | inf)) st |= MBLS_ST_SIG_NOT_IN_G2;
    if (st && !(t & 1)) atomicOr(status + i, st);
#endif
}
__global__ void MBLS_LB k_blind_sig2_d(mbls_ws ws, const uint8_t* sigs96, const uint64_t* rands, uint32_t* status, uint64_t n) {
#if MBLS_DEVICE_ASM
    __shared__ uint32_t spill[154 * 64];
    const uint64_t t = gid(); if (t >= 2 * n) return;
    const uint64_t i = t >> 1;
    uint32_t st = 0;
    bool inf;
    if (sigs96) {
        fp2 x, y;
        int e = g2_decode_compressed_t<true>(&x, &y, &inf, sigs96 + 96 * i);
        if (e) { st |= MBLS_ST_BAD_SIG_ENCODING; inf = true; }
        if (inf) { x = fp2_zero(); y = fp2_zero(); }
        ws_st2(ws, MBLS_SLOT_SIG, i, x); ws_st2(ws, MBLS_SLOT_SIG + 2, i, y);
    } else
        inf = fp2_is_zero(ws_ld2(ws, MBLS_SLOT_SIG + 2, i));
    const uint64_t r = rands[i];
    if (r == 0) st |= MBLS_ST_BAD_SCALAR;
    const uint32_t fl = g2_blind2_d_call(ws, i, (MBLS_LDS uint32_t*)spill, threadIdx.x, inf ? 0 : r, sigs96 ? 0u : 1u);
    if (!((fl & 1u) | inf)) st |= MBLS_ST_SIG_NOT_IN_G2;
    if (st && !(t & 1)) atomicOr(status + i, st);
#endif
}
// f_i = Miller(H_i, P_i) for i < n; lane n (if with_sig) computes Miller(S, -G1) with S read from slot S of item `s_item`.
// The loop itself is the generated single-pair routine (mbls_pairing.h, miller_loop_single_d).
template <int LPP> MBLS_FN void miller_single_body(const mbls_ws& ws, uint64_t n, int with_sig, uint64_t s_item, uint32_t* spill) {
    uint64_t i = LPP == 2 ? gid() >> 1 : gid(); if (i > n || (i == n && !with_sig)) return;
    // branch-free operand selection (lane n takes S and the constant -G1): the generated routine is called from uniform control flow
    const bool last = (i == n);
    const int qslot = last ? MBLS_SLOT_S : MBLS_SLOT_H;
    const uint64_t qitem = last ? s_item : i;
    g2j h; h.x = ws_ld2(ws, qslot, qitem); h.y = ws_ld2(ws, qslot + 2, qitem); h.z = ws_ld2(ws, qslot + 4, qitem);
    g1j a; a.x = ws_ld(ws, MBLS_SLOT_APK, i); a.y = ws_ld(ws, MBLS_SLOT_APK + 1, i); a.z = ws_ld(ws, MBLS_SLOT_APK + 2, i);
    a.x = fp_select(last, fp_load_const(MBLS_G1_X), a.x); a.y = fp_select(last, fp_load_const(MBLS_G1_NEG_Y), a.y); a.z = fp_select(last, fp_one(), a.z);
    mbls_pair pr;
    pr.skip = g2_is_inf(&h) | g1_is_inf(&a);
    g2h_from_jacobian(&pr.q, &h); g1arg_from_jacobian(&pr.p, &a);
    pr.t = pr.q;
    fp12 f;
#if MBLS_DEVICE_ASM
    if (LPP == 2) miller_loop_single_pair_d(&f, &pr, ws.w, ws.stride, i, (MBLS_LDS uint32_t*)spill, threadIdx.x);
    else miller_loop_single_d(&f, &pr, ws.w, ws.stride, i, (MBLS_LDS uint32_t*)spill, threadIdx.x);
#else
    miller_loop(&f, &pr, 1);
#endif
    const fp2* c = &f.c0.c0;
    for (int s = 0; s < 6; s++) ws_st2(ws, MBLS_SLOT_F + 2 * s, i, c[s]);
}
__global__ void MBLS_LB k_miller_single(mbls_ws ws, uint64_t n, int with_sig, uint64_t s_item) {
    __shared__ uint32_t spill[154 * 64];
    miller_single_body<1>(ws, n, with_sig, s_item, spill);
}
// two lanes per Miller loop (half a round of pairs or less): lanes 2 i, 2 i + 1 walk pair i together, their products in pairs
__global__ void MBLS_LB k_miller_single2(mbls_ws ws, uint64_t n, int with_sig, uint64_t s_item) {
    __shared__ uint32_t spill[154 * 64];
    miller_single_body<2>(ws, n, with_sig, s_item, spill);
}
// tree levels with one lane per product: item i <- item i (op) item i + half, for i + half < m (generated routines, tools/gen_tower_d.py)
__global__ void MBLS_LB k_f12_tree_d(mbls_ws ws, uint64_t m, uint64_t half) {
#if MBLS_DEVICE_ASM
    __shared__ uint32_t spill[154 * 64];
    uint64_t i = gid(); if (i + half >= m || i >= half) return;
    tree_level_d_call<false>(ws, i, half, (MBLS_LDS uint32_t*)spill, threadIdx.x);
#endif
}
__global__ void MBLS_LB k_g2_tree_d(mbls_ws ws, uint64_t m, uint64_t half) {
#if MBLS_DEVICE_ASM
    __shared__ uint32_t spill[154 * 64];
    uint64_t i = gid(); if (i + half >= m || i >= half) return;
    tree_level_d_call<true>(ws, i, half, (MBLS_LDS uint32_t*)spill, threadIdx.x);
#endif
}
// ---- batched AggregateVerify (reference src/aggregates.rs:130-170, n calls at once). Workspace items: [0, n) the (sig_i, -G1) pairs,
// [n, n + T) the T (message, key) pairs of all items back to back (item i owns pairs [off[i], off[i+1]) or k each), [n + T, 2 n + T) staging.
// item i's signature (slots SIG of item i, written by k_sig: affine, y = 0 = infinity) -> the operands of its (sig, -G1) pair in the slots
// k_miller_single reads: H = the signature (Jacobian), APK = -G1
__global__ void MBLS_LB k_sigpair_setup(mbls_ws ws, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    g2j s; s.x = ws_ld2(ws, MBLS_SLOT_SIG, i); s.y = ws_ld2(ws, MBLS_SLOT_SIG + 2, i); s.z = fp2_one();
    if (fp2_is_zero(s.y)) g2_set_inf(&s);
    ws_st2(ws, MBLS_SLOT_H, i, s.x); ws_st2(ws, MBLS_SLOT_H + 2, i, s.y); ws_st2(ws, MBLS_SLOT_H + 4, i, s.z);
    ws_st(ws, MBLS_SLOT_APK, i, fp_load_const(MBLS_G1_X)); ws_st(ws, MBLS_SLOT_APK + 1, i, fp_load_const(MBLS_G1_NEG_Y)); ws_st(ws, MBLS_SLOT_APK + 2, i, fp_one());
}
// pair j -> the item that owns it (ragged layouts; one lane per item walks its range)
__global__ void MBLS_LB k_pair_item_map(const uint32_t* off, uint64_t n, uint64_t total, uint32_t* map) {
    uint64_t i = gid(); if (i >= n) return;
    const uint64_t a = off[i], b = off[i + 1];
    if (b < a || b > total) return;
    for (uint64_t j = a; j < b; j++) map[j] = (uint32_t)i;
}
// one level of the per-item product trees over the pairs' Miller values: pair j (workspace item j of the view `wp`) takes its partner j + half
// when both lie in the same item's range and j is a multiple of 2 half from the start of that range. The generated tree routine, called with
// the lanes of the wave that have no product this level switched off.
__global__ void MBLS_LB k_f12_seg_tree_d(mbls_ws wp, const uint32_t* map, const uint32_t* off, uint32_t k, uint64_t n, uint64_t total, uint64_t half) {
#if MBLS_DEVICE_ASM
    __shared__ uint32_t spill[154 * 64];
    uint64_t j = gid(); if (j >= total) return;
    uint64_t lo, hi;
    if (off) {
        // map[j] = 0xFFFFFFFF: no item owns pair j (a table that does not start at 0, ends below total or runs backwards) -- never an index
        const uint32_t i = map[j]; if (i >= n) return;
        lo = off[i]; hi = off[i + 1];
        if (!(lo <= j && j < hi && hi <= total)) return;
    } else { lo = (j / k) * k; hi = lo + k; }
    const uint64_t r = j - lo;
    if (r % (2 * half) != 0 || j + half >= hi) return;
    tree_level_d_call<false>(wp, j, half, (MBLS_LDS uint32_t*)spill, threadIdx.x);
#endif
}
// item i: the product of its pairs (left in the first pair of its range by the tree) -> the staging item n + T + i (1 for an empty range);
// and the item's status = its signature's | the OR of its pairs' | MBLS_ST_NO_KEYS for an empty range (src/aggregates.rs:131-133)
__global__ void MBLS_LB k_f12_seg_gather(mbls_ws ws, const uint32_t* off, const uint32_t* map, uint32_t k, uint64_t n, uint64_t total, uint32_t* st_item, const uint32_t* st_pair) {
    uint64_t i = gid(); if (i >= n) return;
    uint64_t lo = off ? off[i] : (uint64_t)k * i, hi = off ? off[i + 1] : lo + k;
    uint32_t st = 0;
    if (hi < lo || hi > total) { st |= MBLS_ST_BAD_PK_ENCODING; hi = lo; }        // an offset table that runs backwards never becomes a read
    if (hi == lo) st |= MBLS_ST_NO_KEYS;
    // a pair of this range that another item claims as well (ranges overlap where a table ran backwards): that item's tree may have raced with this one's,
    // so the item is rejected -- an item that is NOT flagged owned every one of its pairs alone
    for (uint64_t j = lo; j < hi; j++) { st |= st_pair[j]; if (map && map[j] != (uint32_t)i) st |= MBLS_ST_BAD_PK_ENCODING; }
    fp12 f; fp12_set_one(&f);
    const fp* one = &f.c0.c0.c0;
    for (int t = 0; t < 12; t++) ws_st(ws, MBLS_SLOT_F + t, n + total + i, hi > lo ? ws_ld(ws, MBLS_SLOT_F + t, n + lo) : one[t]);
    if (st) atomicOr(st_item + i, st);
}
__global__ void MBLS_LB k_status_or(const uint32_t* status, uint64_t n, uint32_t* out) {
    uint64_t i = gid(); uint32_t v = (i < n) ? status[i] : 0;
    if (__ballot(v != 0)) { if (v) atomicOr(out, v); }
}
// verify_multiple over several devices ("one exchange step"): what one shard contributes is the product of its
// sets' Miller values (slot F of item 0), its sum of blinded signatures (slot S of item 0) and the OR of its status words -- a
// MBLS_VM_PARTIAL_BYTES record in the workspace's own number format (opaque: only mbls_verify_multiple_finish_device of the same build reads
// it). An empty shard contributes (1, infinity, 0).
#define MBLS_VM_PARTIAL_WORDS (MBLS_VM_PARTIAL_BYTES / 4)
#define MBLS_VM_MAGIC 0x4d564d31u
static_assert(MBLS_VM_PARTIAL_WORDS >= 18 * 12 + 2, "partial record too small");
__global__ void MBLS_LB k_vm_export(mbls_ws ws, const uint32_t* st_or, int empty, uint32_t* out) {
    const uint32_t t = threadIdx.x;
    if (blockIdx.x) return;
    if (empty) {
        if (t == 0) {
            fp12 f; fp12_set_one(&f); const fp* one = &f.c0.c0.c0;
            for (int a = 0; a < 12; a++) for (int j = 0; j < 12; j++) out[12 * a + j] = one[a][j];
            for (int j = 144; j < 216; j++) out[j] = 0;                   // Z = 0: infinity
            out[216] = 0; out[217] = MBLS_VM_MAGIC;
            for (int j = 218; j < (int)MBLS_VM_PARTIAL_WORDS; j++) out[j] = 0;
        }
        return;
    }
    for (uint32_t j = t; j < 216; j += WG) {                             // word j of the record = limb j % 12 of slot F + j / 12 (F and S are adjacent)
        const uint32_t slot = MBLS_SLOT_F + j / 12, limb = j % 12;
        out[j] = ws.w[((uint64_t)slot * 12 + limb) * ws.stride];
    }
    if (t == 0) { out[216] = st_or[0]; out[217] = MBLS_VM_MAGIC; }
    for (uint32_t j = 218 + t; j < MBLS_VM_PARTIAL_WORDS; j += WG) out[j] = 0;
}
static_assert(MBLS_SLOT_S == MBLS_SLOT_F + 12, "k_vm_export / k_vm_import copy slots F and S as one range");
// record g -> slots F, S of item g; the status words ORed into st_or[0]; a record that is not one (wrong magic) makes the check fail
__global__ void MBLS_LB k_vm_import(mbls_ws ws, const uint32_t* in, uint64_t G, uint32_t* st_or) {
    const uint64_t g = gid(); if (g >= G) return;
    const uint32_t* r = in + g * MBLS_VM_PARTIAL_WORDS;
    for (uint32_t j = 0; j < 216; j++) ws.w[((uint64_t)(MBLS_SLOT_F + j / 12) * 12 + j % 12) * ws.stride + g] = r[j];
    uint32_t st = r[216];
    if (r[217] != MBLS_VM_MAGIC) st |= MBLS_ST_PAIRING_FAILED | MBLS_ST_BAD_SIG_ENCODING;
    if (st) atomicOr(st_or, st);
}
// the signature k_sig decoded into slots 3..6 of item `item` (affine; y = 0: infinity) -> slot S of the same item in Jacobian form: the
// (sig, -G1) pair of aggregate_verify (reference src/aggregates.rs:158-164)
__global__ void MBLS_LB k_sigslot_to_s(mbls_ws ws, uint64_t item) {
    if (gid() != 0) return;
    g2j s; s.x = ws_ld2(ws, MBLS_SLOT_SIG, item); s.y = ws_ld2(ws, MBLS_SLOT_SIG + 2, item); s.z = fp2_one();
    if (fp2_is_zero(s.y)) g2_set_inf(&s);
    ws_st2(ws, MBLS_SLOT_S, item, s.x); ws_st2(ws, MBLS_SLOT_S + 2, item, s.y); ws_st2(ws, MBLS_SLOT_S + 4, item, s.z);
}

// ------------------------------------------------------------------------------------------------ auxiliary kernels
__global__ void MBLS_LB k_g1_decode(const uint8_t* in, int fmt, int validate, uint64_t n, uint8_t* out96, uint8_t* err) { uint64_t i = gid(); if (i < n) op_g1_decode(i, in, fmt, validate, out96, err); }
__global__ void MBLS_LB k_g1_key_validate(const uint8_t* in96, uint64_t n, uint8_t* ok) { uint64_t i = gid(); if (i < n) op_g1_key_validate(i, in96, ok); }
__global__ void MBLS_LB k_g1_compress(const uint8_t* in96, uint64_t n, uint8_t* out48, uint8_t* err) { uint64_t i = gid(); if (i < n) op_g1_compress(i, in96, out48, err); }
// n x Signature::from_bytes (reference src/signature.rs:43-46) + the subgroup test verify performs (:29): the decoder inlined as in k_sig and the GENERATED subgroup
// routine on the item's workspace slots (psi(P) = [x] P; no lane-private memory). err[i] = the decoder's code; in_g2 (optional): 1 for points of G2 and for infinity.
__global__ void MBLS_LB k_g2_check(mbls_ws ws, const uint8_t* in96, uint64_t n, uint8_t* err, uint8_t* in_g2) {
#if MBLS_DEVICE_ASM
    __shared__ uint32_t spill[154 * 64];
    uint64_t i = gid(); if (i >= n) return;
    fp2 x, y; bool inf;
    const int e = g2_decode_compressed_t<true>(&x, &y, &inf, in96 + 96 * i);
    err[i] = (uint8_t)e;
    if (!in_g2) return;                               // (uniform: a kernel argument)
    if (e) inf = true;
    if (inf) { x = fp2_zero(); y = fp2_zero(); }
    ws_st2(ws, MBLS_SLOT_SIG, i, x); ws_st2(ws, MBLS_SLOT_SIG + 2, i, y);
    const uint32_t fl = g2_group_d_call<false>(ws, i, (MBLS_LDS uint32_t*)spill, threadIdx.x);
    in_g2[i] = (!e && ((fl & 1u) | (inf ? 1u : 0u))) ? 1 : 0;
#endif
}
__global__ void MBLS_LB k_g2_add(const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out, uint8_t* err) { uint64_t i = gid(); if (i < n) op_g2_add(i, a, b, out, err); }
__global__ void MBLS_LB k_g1_add(const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out, uint8_t* err) { uint64_t i = gid(); if (i < n) op_g1_add(i, a, b, out, err); }
__global__ void MBLS_LB k_sk_to_pk(const uint8_t* sks, int fmt, uint64_t n, uint8_t* out) { uint64_t i = gid(); if (i < n) op_sk_to_pk(i, sks, fmt, out); }
__global__ void MBLS_LB k_fp_mul(const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out, int op) { uint64_t i = gid(); if (i < n) op_fp_mul(i, n, a, b, out, op); }
// PublicKey::from_secret_key as a key sum: [sk] G1 = sum_j [d_j 16^j] G1 over the 64 hexadecimal digits of sk, every term a record of a
// fixed 64 x 16 table (record 16 j + d; d = 0 is the point at infinity) -- the indexed key-sum routine does the rest, no doubling at all
__global__ void MBLS_LB k_sk_digits(const uint8_t* sks32, uint64_t n, uint32_t* idx) {
    uint64_t i = gid(); if (i >= n) return;
    const uint8_t* b = sks32 + 32 * i;
    for (uint32_t j = 0; j < 64; j++) {
        const uint32_t d = (b[31 - (j >> 1)] >> (4 * (j & 1))) & 15u;
        idx[64 * i + j] = 16 * j + d;
    }
}
// CONSTANT-TIME sk -> pk (the default; reference src/keys.rs:124-137 -- amcl's g1mul selects in constant time): lane (key i, window j) reads ALL 16 records
// [d 16^j] G1 of its window and keeps record d_j by selection, so neither an address nor the instruction stream depends on the key; the selected records form a
// per-batch table sel[64 i + j] that the key-sum routine then walks with the public indices 64 i + j (idx).
__global__ void __launch_bounds__(WG) k_sk_select(const uint8_t* sks32, uint64_t n, const uint32_t* gtab, uint32_t* sel, uint32_t* idx) {
    const uint64_t t = gid(); if (t >= 64 * n) return;
    const uint64_t i = t >> 6; const uint32_t j = (uint32_t)(t & 63);
    const uint32_t d = (sks32[32 * i + 31 - (j >> 1)] >> (4 * (j & 1))) & 15u;
    const uint4* rec = (const uint4*)(gtab + (size_t)16 * j * MBLS_KEYREC_DWORDS);
    uint4 acc[MBLS_KEYREC_DWORDS / 4];
#pragma unroll
    for (int q = 0; q < MBLS_KEYREC_DWORDS / 4; q++) acc[q] = rec[q];
    for (uint32_t e = 1; e < 16; e++) {
        // the selection as mask arithmetic on an OPAQUE mask: with a plain `take ? v : acc` the compiler guards the loads with the condition (s_and_saveexec +
        // s_cbranch_execz around global_load: only the lanes whose digit is e would touch record e -- exactly the access pattern this kernel exists to avoid).
        // tests/test_build_cpu.py checks the ISA: 16 x 8 unconditional 16-byte loads, no branch between the first load and the store.
        uint32_t m = 0u - (uint32_t)(e == d);
        asm volatile("" : "+v"(m));
#pragma unroll
        for (int q = 0; q < MBLS_KEYREC_DWORDS / 4; q++) {
            const uint4 v = rec[(size_t)e * (MBLS_KEYREC_DWORDS / 4) + q];
            acc[q].x = (v.x & m) | (acc[q].x & ~m); acc[q].y = (v.y & m) | (acc[q].y & ~m);
            acc[q].z = (v.z & m) | (acc[q].z & ~m); acc[q].w = (v.w & m) | (acc[q].w & ~m);
        }
    }
    uint4* out = (uint4*)(sel + t * MBLS_KEYREC_DWORDS);
#pragma unroll
    for (int q = 0; q < MBLS_KEYREC_DWORDS / 4; q++) out[q] = acc[q];
    idx[t] = (uint32_t)t;
}
__global__ void MBLS_LB k_apk_export_fmt(mbls_ws ws, uint64_t n, int fmt, uint8_t* out) {
    uint64_t i = gid(); if (i >= n) return;
    g1j a; a.x = ws_ld(ws, MBLS_SLOT_APK, i); a.y = ws_ld(ws, MBLS_SLOT_APK + 1, i); a.z = ws_ld(ws, MBLS_SLOT_APK + 2, i);
    fp x, y; bool inf; g1_to_affine(&x, &y, &inf, &a);
    if (fmt == MBLS_PK_COMPRESSED) g1_encode_compressed(out + 48 * i, x, y, inf); else g1_encode_uncompressed(out + 96 * i, x, y, inf);
}
__global__ void MBLS_LB k_apk_export(mbls_ws ws, uint64_t n, uint8_t* out96) { uint64_t i = gid(); if (i < n) op_apk_export(ws, i, out96); }
// H(m) as the message phase of the pipeline left it in workspace slots 7..12 (Jacobian) -> 96 compressed bytes (the probe mbls_hash_to_g2_batch)
// Signature::new (reference src/signature.rs:17-21) on the generated routines. psi acts on G2 as multiplication by the curve parameter
// x = -y, y = 0xd201000000010000, and r = y^4 - y^2 + 1 < y^4: with sk mod r = a0 + a1 y + a2 y^2 + a3 y^3 (0 <= a_j < y < 2^64),
// [sk] H = sum_j [a_j] (-1)^j psi^j(H) -- four 64-bit multiplications that run side by side on four lanes (items i, n + i, 2n + i, 3n + i)
// with verify_multiple's windowed routine (g2_blind_routine without its subgroup test), then two levels of the G2 sum tree.
// (64-bit limbs in named variables, every limb loop unrolled: nothing here is indexed at run time, so nothing lives in lane-private memory)
MBLS_FN uint64_t div_step_y(uint64_t* rem, uint64_t limb) {       // (rem : limb) / y with rem < y: the quotient limb, the remainder left in *rem (restoring, one bit a step)
    const uint64_t y = MBLS_X_ABS;
    uint64_t r = *rem, q = 0;
    for (int b = 63; b >= 0; b--) {
        const uint64_t top = r >> 63;
        r = (r << 1) | ((limb >> b) & 1u);
        const bool ge = top | (r >= y);
        r = ge ? r - y : r; q = (q << 1) | (ge ? 1u : 0u);
    }
    *rem = r; return q;
}
MBLS_FN void scalar_base_y_digits(uint64_t a[4], const uint8_t* sk32) {
    uint64_t k3 = 0, k2 = 0, k1 = 0, k0 = 0;                        // big-endian bytes -> limbs, k0 the least significant
#pragma unroll
    for (int j = 0; j < 8; j++) { k3 = (k3 << 8) | sk32[j]; k2 = (k2 << 8) | sk32[8 + j]; k1 = (k1 << 8) | sk32[16 + j]; k0 = (k0 << 8) | sk32[24 + j]; }
    const uint64_t r0 = 0xffffffff00000001ull, r1 = 0x53bda402fffe5bfeull, r2 = 0x3339d80809a1d805ull, r3 = 0x73eda753299d7d48ull;      // the group order
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {                            // any 32-byte value: 2^256 < 5 r, at most four subtractions
        const uint64_t d0 = k0 - r0; const uint64_t b0 = k0 < r0;
        const uint64_t t1 = k1 - r1; const uint64_t d1 = t1 - b0; const uint64_t b1 = (k1 < r1) | (t1 < b0);
        const uint64_t t2 = k2 - r2; const uint64_t d2 = t2 - b1; const uint64_t b2 = (k2 < r2) | (t2 < b1);
        const uint64_t t3 = k3 - r3; const uint64_t d3 = t3 - b2; const uint64_t b3 = (k3 < r3) | (t3 < b2);
        k0 = b3 ? k0 : d0; k1 = b3 ? k1 : d1; k2 = b3 ? k2 : d2; k3 = b3 ? k3 : d3;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {                                  // k <- k / y, a[t] = k mod y
        uint64_t rem = 0;
        k3 = div_step_y(&rem, k3); k2 = div_step_y(&rem, k2); k1 = div_step_y(&rem, k1); k0 = div_step_y(&rem, k0);
        a[t] = rem;
    }
}
__global__ void MBLS_LB k_sign_blind(mbls_ws ws, const uint8_t* sks32, uint64_t n, int ct) {
#if MBLS_DEVICE_ASM
    __shared__ uint32_t spill[154 * 64];
    uint64_t t = gid(); if (t >= 4 * n) return;
    const uint64_t i = t % n; const uint32_t j = (uint32_t)(t / n);
    g2j h; h.x = ws_ld2(ws, MBLS_SLOT_H, i); h.y = ws_ld2(ws, MBLS_SLOT_H + 2, i); h.z = ws_ld2(ws, MBLS_SLOT_H + 4, i);
    fp2 x, y; bool inf; g2_to_affine(&x, &y, &inf, &h);           // H(m) is never infinity in practice; (0, 0) with the scalar 0 if it is
    g2j q; q.x = x; q.y = y; q.z = fp2_one();
    for (uint32_t e = 0; e < 3; e++) { g2j u; g2_psi(&u, &q); const bool on = e < j; q.x = fp2_select(on, u.x, q.x); q.y = fp2_select(on, u.y, q.y); }
    q.y = fp2_select((j & 1u) != 0, fp2_neg(q.y), q.y);
    if (inf) { q.x = fp2_zero(); q.y = fp2_zero(); }
    ws_st2(ws, MBLS_SLOT_SIG, t, q.x); ws_st2(ws, MBLS_SLOT_SIG + 2, t, q.y);
    uint64_t a[4]; scalar_base_y_digits(a, sks32 + 32 * i);
    const uint64_t r = j == 0 ? a[0] : j == 1 ? a[1] : j == 2 ? a[2] : a[3];
    // ct (the default): the window tables are read by scan + selection, never at an address that depends on the key (mbls_ctx_set_secret_ops)
    if (ct) (void)g2_blind_ct_d_call(ws, t, (MBLS_LDS uint32_t*)spill, threadIdx.x, inf ? 0 : r, 1u);
    else (void)g2_blind_d_call(ws, t, (MBLS_LDS uint32_t*)spill, threadIdx.x, inf ? 0 : r, 1u);
#endif
}
__global__ void MBLS_LB k_s_export(mbls_ws ws, uint64_t n, uint8_t* out96) {
    uint64_t i = gid(); if (i >= n) return;
    g2j h; h.x = ws_ld2(ws, MBLS_SLOT_S, i); h.y = ws_ld2(ws, MBLS_SLOT_S + 2, i); h.z = ws_ld2(ws, MBLS_SLOT_S + 4, i);
    g2_encode_jacobian(out96 + 96 * i, &h);
}
__global__ void MBLS_LB k_h_export(mbls_ws ws, uint64_t n, uint8_t* out96) {
    uint64_t i = gid(); if (i >= n) return;
    g2j h; h.x = ws_ld2(ws, MBLS_SLOT_H, i); h.y = ws_ld2(ws, MBLS_SLOT_H + 2, i); h.z = ws_ld2(ws, MBLS_SLOT_H + 4, i);
    g2_encode_jacobian(out96 + 96 * i, &h);
}
__global__ void MBLS_LB k_fp_mul_bench(uint32_t* sink, uint32_t iters, uint64_t n) {
    uint64_t i = gid(); if (i >= n) return;
    fp a = fp_load_const(MBLS_G1_X), b = fp_load_const(MBLS_G1_Y);
    a[0] ^= (uint32_t)i;
    for (uint32_t it = 0; it < iters; it++) a = fp_mul(a, b);
    uint32_t acc = 0;
    for (int j = 0; j < 12; j++) acc ^= a[j];
    sink[i] = acc;
}

// VALU issue-rate calibration (bench.py, valu_issue): 128 instructions per iteration, 8 independent chains per lane.
// mode 0: v_mad_u64_u32 with the carry-out alternating between VCC and SGPR pairs (what the multiplication routines issue);
// mode 1: v_add_co / v_addc chains (the class every other integer instruction of the routines issues at).
#define MBLS_REP4(x) x x x x
#define MBLS_VB_V8(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define MBLS_VALU_BENCH_CLOBBERS "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", MBLS_VB_V8(1), MBLS_VB_V8(2), MBLS_VB_V8(3), MBLS_VB_V8(4), MBLS_VB_V8(5), \
    MBLS_VB_V8(6), MBLS_VB_V8(7), MBLS_VB_V8(8), MBLS_VB_V8(9), MBLS_VB_V8(10), MBLS_VB_V8(11), "v120", "v121", "v122", "v123", "v124", "v125", \
    "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "vcc", "scc"
#define MBLS_REP16(x) MBLS_REP4(MBLS_REP4(x))
__global__ void __launch_bounds__(WG) k_valu_bench(uint32_t* sink, uint32_t iters, int mode) {
    uint32_t a = (threadIdx.x * 2654435761u + blockIdx.x) | 1u, b = a ^ 0x9e3779b9u;
    uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
    uint32_t h0 = a, h1 = b, h2 = a + 5, h3 = b + 7;
    for (uint32_t i = 0; i < iters; i++) {
        if (mode == 0) {
            MBLS_REP16(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, s[22:23], %8, %9, %3\n"
                                    "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, s[22:23], %8, %9, %7\n"
                                    : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc", "s20", "s21", "s22", "s23");)
        } else if (mode == 2) {
#if MBLS_DEVICE_ASM
            // the Fp2 product routine itself, eight times back to back (8 x 1 281 instructions, 980 of them multiply-accumulates): what a kernel
            // made of nothing but products would issue at -- the reference the generated kernels' rates are quoted against
            MBLS_REP4(asm volatile(MBLS_FP2_MUL_D_ASM MBLS_FP2_MUL_D_ASM ::: MBLS_VALU_BENCH_CLOBBERS);)
#endif
        } else if (mode == 3) {
#if MBLS_DEVICE_ASM
            // the same with the paired Fp product of the key-sum routines (8 x 923 instructions, 784 multiply-accumulates each)
            MBLS_REP4(asm volatile(MBLS_FP_MULPAIR_D_ASM MBLS_FP_MULPAIR_D_ASM ::: MBLS_VALU_BENCH_CLOBBERS);)
#endif
        } else if (mode == 4) {
            // class model against mix, the experiment: 8 multiply-accumulates and 4 plain two-operand operations INTERLEAVED 2 : 1 (the kernels'
            // proportion) ...
            MBLS_REP16(asm volatile("v_mad_u64_u32 %0, vcc, %12, %13, %0\n v_mad_u64_u32 %1, s[20:21], %12, %13, %1\n v_add_u32_e32 %8, %12, %8\n"
                                    "v_mad_u64_u32 %2, vcc, %12, %13, %2\n v_mad_u64_u32 %3, s[22:23], %12, %13, %3\n v_and_b32_e32 %9, %13, %9\n"
                                    "v_mad_u64_u32 %4, vcc, %12, %13, %4\n v_mad_u64_u32 %5, s[20:21], %12, %13, %5\n v_add_u32_e32 %10, %13, %10\n"
                                    "v_mad_u64_u32 %6, vcc, %12, %13, %6\n v_mad_u64_u32 %7, s[22:23], %12, %13, %7\n v_xor_b32_e32 %11, %12, %11\n"
                                    : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3)
                                    : "v"(a), "v"(b) : "vcc", "s20", "s21", "s22", "s23");)
        } else if (mode == 5) {
            // ... and the SAME twelve instructions in two homogeneous blocks (what a per-class model of the stream assumes)
            MBLS_REP16(asm volatile("v_mad_u64_u32 %0, vcc, %12, %13, %0\n v_mad_u64_u32 %1, s[20:21], %12, %13, %1\n"
                                    "v_mad_u64_u32 %2, vcc, %12, %13, %2\n v_mad_u64_u32 %3, s[22:23], %12, %13, %3\n"
                                    "v_mad_u64_u32 %4, vcc, %12, %13, %4\n v_mad_u64_u32 %5, s[20:21], %12, %13, %5\n"
                                    "v_mad_u64_u32 %6, vcc, %12, %13, %6\n v_mad_u64_u32 %7, s[22:23], %12, %13, %7\n"
                                    "v_add_u32_e32 %8, %12, %8\n v_and_b32_e32 %9, %13, %9\n v_add_u32_e32 %10, %13, %10\n v_xor_b32_e32 %11, %12, %11\n"
                                    : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3)
                                    : "v"(a), "v"(b) : "vcc", "s20", "s21", "s22", "s23");)
        } else if (mode == 7) {
            // the scans' own instruction: v_mad_i64_i32 on eight accumulators (the peak of bench.py's mac_frac: whichever of mode 0 and this one issues faster)
            MBLS_REP16(asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_mad_i64_i32 %1, s[20:21], %8, %9, %1\n v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_mad_i64_i32 %3, s[22:23], %8, %9, %3\n"
                                    "v_mad_i64_i32 %4, vcc, %8, %9, %4\n v_mad_i64_i32 %5, s[20:21], %8, %9, %5\n v_mad_i64_i32 %6, vcc, %8, %9, %6\n v_mad_i64_i32 %7, s[22:23], %8, %9, %7\n"
                                    : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc", "s20", "s21", "s22", "s23");)
        } else if (mode == 6) {
            // ... and the four plain operations alone
            MBLS_REP16(asm volatile("v_add_u32_e32 %0, %4, %0\n v_and_b32_e32 %1, %5, %1\n v_add_u32_e32 %2, %5, %2\n v_xor_b32_e32 %3, %4, %3\n"
                                    "v_add_u32_e32 %0, %5, %0\n v_and_b32_e32 %1, %4, %1\n v_add_u32_e32 %2, %4, %2\n v_xor_b32_e32 %3, %5, %3\n"
                                    : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));)
        } else {
            MBLS_REP16(asm volatile("v_add_co_u32_e64 %0, vcc, %4, %0\n v_add_co_u32_e64 %2, s[20:21], %5, %2\n v_addc_co_u32_e64 %1, vcc, %5, %1, vcc\n v_addc_co_u32_e64 %3, s[20:21], %4, %3, s[20:21]\n"
                                    "v_add_co_u32_e64 %0, vcc, %5, %0\n v_add_co_u32_e64 %2, s[20:21], %4, %2\n v_addc_co_u32_e64 %1, vcc, %4, %1, vcc\n v_addc_co_u32_e64 %3, s[20:21], %5, %3, s[20:21]\n"
                                    : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : "vcc", "s20", "s21");)
        }
    }
    sink[(uint64_t)blockIdx.x * WG + threadIdx.x] = (uint32_t)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) ^ h0 ^ h1 ^ h2 ^ h3;
}

// ------------------------------------------------------------------------------------------------ host side
// Every extern "C" entry takes the context's (recursive) lock: a context may be shared between threads, calls are serialised.
// Host-buffer entries stage through grow-only device buffers and two streams that the context owns (no allocation or stream
// creation per call once the sizes have been seen); device-pointer entries only enqueue, and order their use of the
// workspace against earlier calls on other streams with an event.
#define MBLS_N_STAGE 10
// measured crossovers (scripts/throughput_vs_n.py, 128 keys, device-resident): one wave per item wins up to ~5 k items (10.3 ms at 4 096, ~11.8 at
// 5 120); the lane path with four lanes per item in the Miller loop and two in the message phase and the final exponentiation (products in pairs)
// needs 12.2-12.7 ms for anything up to a quarter of a round
#define MBLS_DEFAULT_COOP_MAX_ITEMS 5120
#define MBLS_DEFAULT_COOP_HASH_MAX_ITEMS 3584           /* above n / 4 + 2 n / 64 > 1 024 waves: the four-per-wave message phase needs a second round beside the key sums and the signatures (3 600 items 8.0 ms, 3 648 items 9.9 ms); the lane-pair form takes 8.8 - 9.0 */
#define MBLS_DEFAULT_TRACKS_MIN_REST 3584
struct mbls_ctx {
    std::recursive_mutex mu;
    int device = 0;
    uint64_t cap = 0;                  // items the workspace can hold
    uint32_t* d_w = nullptr;           // workspace limbs
    uint32_t* d_status = nullptr;      // per-item status (when the caller passes none)
    uint8_t* d_results = nullptr;
    uint32_t* d_scalar = nullptr;      // small scratch words
    uint32_t* d_gtab = nullptr;        // [64 * 16][MBLS_KEYREC_DWORDS]: [d 16^j] G1 (sk -> pk), built on first use
    uint32_t* d_skidx = nullptr; uint64_t skidx_cap = 0;      // the table indices of a batch of secret keys
    uint32_t* d_sksel = nullptr; uint64_t sksel_cap = 0;      // constant-time sk -> pk: the records each (key, window) selected, [chunk * 64][MBLS_KEYREC_DWORDS]
    bool secret_ops_fast = false;      // false (default): table lookups that depend on a secret key are scans with selection (mbls_ctx_set_secret_ops)
    uint64_t key_cap = 0;              // decompressed-key staging (compressed wire format): capacity in keys
    uint32_t* d_keys_xy = nullptr;     // [key_cap][24] affine Montgomery coordinates
    uint8_t* d_key_flags = nullptr;
    struct { void* p; size_t cap; } stage[MBLS_N_STAGE] = {};
    hipStream_t hs_a = nullptr, hs_b = nullptr, hs_c = nullptr, hs_d = nullptr;      // streams of the host-buffer entry points (hs_d: the signature phase while hs_b uploads keys)
    hipEvent_t hs_ev = nullptr, hs_ev2 = nullptr, hs_ev3 = nullptr;
    // the second TRACK of the verification pipeline (verify_pipeline): a batch between one and two rounds runs as two halves side by side, each with its own part
    // of the workspace, its own main stream and its own side streams for the front phases
    hipStream_t t1_s = nullptr, t1_b = nullptr, t1_c = nullptr;
    hipEvent_t t1_ev = nullptr, t1_ev2 = nullptr, t1_ev3 = nullptr;
    hipEvent_t ws_ev = nullptr; hipStream_t ws_stream = nullptr; bool ws_pending = false;   // last asynchronous user of the workspace
    bool timing = false;
    hipEvent_t ev[MBLS_N_PHASES + 1] = {};
    float phase_ms[MBLS_N_PHASES] = {};
    std::vector<struct mbls_keytable*> tables;     // the key tables created on this context (orphaned when it is destroyed)
    coop_prog coop[10] = {};           // the cooperative engine's microprograms in HBM (mbls_coop.h): pairing2, vmtail, f12mul, g2add, smiller, vmfinal, hashg2, miller1,
                                       // pairing2x2 (two items per wave), hashg2x4 (four)
    uint32_t* d_coop = nullptr;
    // measured crossovers (scripts/coop_sweep.py, 128 keys, device-resident): one wave per item for the pairing check wins up to ~10 k items
    // (19.7 ms at 10 240 against 22.5), for the message phase as well up to ~6 k (13.3 ms at 6 144 against 14.0; four items per wave above 768)
    uint64_t coop_hash_max_items = MBLS_DEFAULT_COOP_HASH_MAX_ITEMS;
    uint64_t coop_max_items = MBLS_DEFAULT_COOP_MAX_ITEMS;
    // within (pack_min, pack_max] a wave serves two items (pairing check) / four items (message phase) side by side: more steps per wave, fewer per item
    uint64_t coop_pack_min_items = 1024, coop_pack_max_items = 2048, coop_hash_pack_min_items = 768;
    // one ROUND of the one-lane kernels = one wave on every SIMD (512 registers per lane: one wave per SIMD) = CUs x 4 x 64 items. A batch of
    // q rounds + r items would cost q + 1 rounds of every kernel; the r items are cut off and take the route of an r-item batch instead
    uint64_t round_items = 65536;
    // batches of at most split_max_items items (default: half a round) on the one-lane path walk their two Miller pairs on two lanes (k_miller_split);
    // up to fork_max_items items the three front phases (keys | signature | message) run side by side on the context's streams
    uint64_t split_max_items = 32768, fork_max_items = 49152, hash2_max_items = 20480;
    // a batch of one to two rounds whose remainder over the round is at least tracks_min_rest items runs as two equal halves on two tracks (0: never)
    uint64_t tracks_min_rest = MBLS_DEFAULT_TRACKS_MIN_REST;
    uint64_t tracks_side_max = 16384;  // remainders up to this many items run BESIDE the last round (a track of their own) instead of as one of two equal halves
    char err[256] = {};
};
struct mbls_keytable {
    mbls_ctx* c = nullptr;             // nullptr: the context was destroyed first (the records went with it)
    uint32_t* d_recs = nullptr;        // [cap][MBLS_KEYREC_DWORDS]
    uint64_t size = 0, cap = 0;
    hipEvent_t ev = nullptr; hipStream_t ev_stream = nullptr; bool pending = false;   // the last asynchronous append
};
typedef std::lock_guard<std::recursive_mutex> mbls_lock;

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return MBLS_ERR_DEVICE; } } while (0)
#define ARGFAIL(ctx, what) do { snprintf((ctx)->err, sizeof((ctx)->err), "invalid argument: %s", what); return MBLS_ERR_ARGUMENT; } while (0)

static inline unsigned nblk(uint64_t n) { return (unsigned)((n + WG - 1) / WG); }
// the message phase of n items on stream s: one lane per item (k_hash), or -- small batches -- hash_to_field per lane and the rest one wave per item
// pair_ok: the caller's workspace view holds 2 n items and nothing else uses slots 7..24 / 31..42 of items [0, 2 n) meanwhile -- batches of at most
// half a round then take two lanes per message (k_hash2)
enum { HASH_FORM_AUTO = 0, HASH_FORM_LANE, HASH_FORM_WAVE, HASH_FORM_PAIR };
static void launch_hash(mbls_ctx* c, mbls_ws ws, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff, uint32_t* st, uint64_t n, hipStream_t s, bool pair_ok = false,
                        int form = HASH_FORM_AUTO);
// the per-item key sum from wire-format keys: the generated routine for 96-byte keys (its 16-byte loads want 4-byte alignment),
// the compiled lane body for 48-byte keys summed in place and for unaligned buffers
static void launch_aggregate(mbls_ws ws, const uint8_t* d_pks, const uint32_t* d_off, uint32_t k, int fmt, int mode, uint32_t* st, uint64_t n, hipStream_t s) {
    if (fmt == MBLS_PK_UNCOMPRESSED && (((uintptr_t)d_pks) & 3u) == 0)
        hipLaunchKernelGGL(k_aggregate_raw_d, dim3(nblk(n)), dim3(WG), 0, s, ws, d_pks, d_off, k, mode, st, n);
    else
        hipLaunchKernelGGL(k_aggregate, dim3(nblk(n)), dim3(WG), 0, s, ws, d_pks, d_off, k, fmt, mode, st, n);
}

// one launch of the cooperative kernel: program `prog` on n_items items (64 / lpi of them per wave)
enum { COOP_PAIRING2 = 0, COOP_VMTAIL, COOP_F12MUL, COOP_G2ADD, COOP_SMILLER, COOP_VMFINAL, COOP_HASHG2, COOP_MILLER1, COOP_PAIRING2X2, COOP_HASHG2X4 };
static void coop_run(mbls_ctx* c, int prog, mbls_ws ws, uint64_t first_item, uint64_t item_step, uint64_t partner_step, uint64_t n_items, uint32_t* st, uint8_t* res,
                     int res_mode, hipStream_t s) {
    const coop_prog& pg = c->coop[prog];
    const uint64_t ipw = 64 / pg.lpi;
    const dim3 grid((unsigned)((n_items + ipw - 1) / ipw));
    const size_t lds = COOP_LDS_BYTES(pg);
    if (prog == COOP_HASHG2X4)
        hipLaunchKernelGGL(k_coop_pow_x4, grid, dim3(64), lds, s, pg, ws, first_item, item_step, partner_step, n_items, st, res, res_mode);
    else if (prog == COOP_HASHG2)
        hipLaunchKernelGGL(k_coop_pow, grid, dim3(64), lds, s, pg, ws, first_item, item_step, partner_step, n_items, st, res, res_mode);
    else if (prog == COOP_PAIRING2X2)
        hipLaunchKernelGGL(k_coop_x2, grid, dim3(64), lds, s, pg, ws, first_item, item_step, partner_step, n_items, st, res, res_mode);
    else
        hipLaunchKernelGGL(k_coop, grid, dim3(64), lds, s, pg, ws, first_item, item_step, partner_step, n_items, st, res, res_mode);
}
// which form the message phase of n items takes: lane pairs (k_hash2) where the caller reserved 2 n workspace items and the batch is above the wave engine's range
// but at most hash2_max_items (5/16 of a round: 2 n lanes for the messages beside n for the keys, with the signatures behind them, are still resident together; at a
// third of a round and above the doubled message phase pushes the key sums behind it: 20 480 items 16.1 -> 14.6 ms, 21 845 items 16.3 -> 16.7); one wave per item
// (or four items per wave) up to coop_hash_max_items -- the measured crossover, for every caller --; one lane per item otherwise
static int hash_form(const mbls_ctx* c, uint64_t n, bool pair_ok, bool no_waves = false) {
    const bool waves = !no_waves && n <= c->coop_hash_max_items && n <= c->coop_max_items;
    if (pair_ok && !waves && n <= c->split_max_items && n <= c->hash2_max_items) return HASH_FORM_PAIR;
    return waves ? HASH_FORM_WAVE : HASH_FORM_LANE;
}
static void launch_hash(mbls_ctx* c, mbls_ws ws, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff, uint32_t* st, uint64_t n, hipStream_t s, bool pair_ok, int form) {
    if (form == HASH_FORM_AUTO) form = hash_form(c, n, pair_ok);
    if (form == HASH_FORM_PAIR) {
        hipLaunchKernelGGL(k_hash2, dim3(nblk(2 * n)), dim3(WG), 0, s, ws, d_msgs, msg_len, d_moff, st, n);
        hipLaunchKernelGGL(k_h_compact_a, dim3(nblk(n)), dim3(WG), 0, s, ws, n);
        hipLaunchKernelGGL(k_h_compact_b, dim3(nblk(n)), dim3(WG), 0, s, ws, n);
    } else if (form == HASH_FORM_WAVE) {
        hipLaunchKernelGGL(k_hash_fields, dim3(nblk(n)), dim3(WG), 0, s, ws, d_msgs, msg_len, d_moff, st, n);
        coop_run(c, n > c->coop_hash_pack_min_items ? COOP_HASHG2X4 : COOP_HASHG2, ws, (uint64_t)0, (uint64_t)1, (uint64_t)0, n, (uint32_t*)nullptr, (uint8_t*)nullptr, COOP_RES_ITEM, s);
    } else
        hipLaunchKernelGGL(k_hash, dim3(nblk(n)), dim3(WG), 0, s, ws, d_msgs, msg_len, d_moff, st, n);
}
static void ctx_free(mbls_ctx* c) {
    (void)hipSetDevice(c->device);
    if (c->d_w) (void)hipFree(c->d_w);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_results) (void)hipFree(c->d_results);
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    if (c->d_gtab) (void)hipFree(c->d_gtab);
    if (c->d_skidx) (void)hipFree(c->d_skidx);
    if (c->d_sksel) (void)hipFree(c->d_sksel);
    if (c->d_keys_xy) (void)hipFree(c->d_keys_xy);
    if (c->d_key_flags) (void)hipFree(c->d_key_flags);
    if (c->d_coop) (void)hipFree(c->d_coop);
    for (int i = 0; i < MBLS_N_STAGE; i++) if (c->stage[i].p) (void)hipFree(c->stage[i].p);
    for (int i = 0; i <= MBLS_N_PHASES; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->hs_ev) (void)hipEventDestroy(c->hs_ev);
    if (c->hs_ev2) (void)hipEventDestroy(c->hs_ev2);
    if (c->hs_ev3) (void)hipEventDestroy(c->hs_ev3);
    if (c->t1_ev) (void)hipEventDestroy(c->t1_ev);
    if (c->t1_ev2) (void)hipEventDestroy(c->t1_ev2);
    if (c->t1_ev3) (void)hipEventDestroy(c->t1_ev3);
    if (c->t1_s) (void)hipStreamDestroy(c->t1_s);
    if (c->t1_b) (void)hipStreamDestroy(c->t1_b);
    if (c->t1_c) (void)hipStreamDestroy(c->t1_c);
    if (c->ws_ev) (void)hipEventDestroy(c->ws_ev);
    if (c->hs_a) (void)hipStreamDestroy(c->hs_a);
    if (c->hs_b) (void)hipStreamDestroy(c->hs_b);
    if (c->hs_c) (void)hipStreamDestroy(c->hs_c);
    if (c->hs_d) (void)hipStreamDestroy(c->hs_d);
    delete c;
}
static void ctx_default_tuning(mbls_ctx* c);
extern "C" int mbls_ctx_create(mbls_ctx** out, int device_id) {
    if (!out) return MBLS_ERR_ARGUMENT;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device_id < 0 || device_id >= cnt) return MBLS_ERR_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return MBLS_ERR_DEVICE;
    mbls_ctx* c = new (std::nothrow) mbls_ctx();
    if (!c) return MBLS_ERR_DEVICE;
    c->device = device_id;
    bool ok = true;
    for (int i = 0; i <= MBLS_N_PHASES; i++) ok = ok && hipEventCreate(&c->ev[i]) == hipSuccess;
    ok = ok && hipMalloc(&c->d_scalar, 64) == hipSuccess;
    // the side streams of the front phases: when the three chains do not fit the chip side by side (above a quarter of a round), the message phase -- the
    // longest of them -- is to get its SIMDs first and the signature phase -- the shortest -- last (32 768 items: the front takes 4.4 ms instead of 5.0)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);           // (numerically: lowest priority, highest priority)
    if (getenv("MBLS_NO_STREAM_PRIORITIES")) prio_lo = prio_hi = 0;
    ok = ok && hipStreamCreateWithFlags(&c->hs_a, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithPriority(&c->hs_b, hipStreamNonBlocking, prio_lo) == hipSuccess &&
         hipStreamCreateWithPriority(&c->hs_c, hipStreamNonBlocking, prio_hi) == hipSuccess && hipStreamCreateWithPriority(&c->hs_d, hipStreamNonBlocking, prio_lo) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->hs_ev, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->hs_ev2, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->hs_ev3, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ws_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->t1_s, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithPriority(&c->t1_b, hipStreamNonBlocking, prio_lo) == hipSuccess &&
         hipStreamCreateWithPriority(&c->t1_c, hipStreamNonBlocking, prio_hi) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->t1_ev, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->t1_ev2, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->t1_ev3, hipEventDisableTiming) == hipSuccess;
    if (ok) {       // the cooperative engine's programs: one upload per context
#define COOP_SRC(P) {MBLS_COOP_##P##_STEPS, MBLS_COOP_##P##_ROWS, MBLS_COOP_##P##_CONSTS}
#define COOP_CNT(P) {2 * MBLS_COOP_##P##_NSTEPS, 512 * MBLS_COOP_##P##_NROWS, 15 * (MBLS_COOP_##P##_NCONSTS ? MBLS_COOP_##P##_NCONSTS : 1)}
        const int NP = 10;
        const uint32_t* src[NP][3] = {COOP_SRC(PAIRING2), COOP_SRC(VMTAIL), COOP_SRC(F12MUL), COOP_SRC(G2ADD), COOP_SRC(SMILLER), COOP_SRC(VMFINAL), COOP_SRC(HASHG2),
                                      COOP_SRC(MILLER1), COOP_SRC(PAIRING2X2), COOP_SRC(HASHG2X4)};
        const size_t cnt[NP][3] = {COOP_CNT(PAIRING2), COOP_CNT(VMTAIL), COOP_CNT(F12MUL), COOP_CNT(G2ADD), COOP_CNT(SMILLER), COOP_CNT(VMFINAL), COOP_CNT(HASHG2),
                                   COOP_CNT(MILLER1), COOP_CNT(PAIRING2X2), COOP_CNT(HASHG2X4)};
        const uint32_t nconst[NP] = {MBLS_COOP_PAIRING2_NCONSTS, MBLS_COOP_VMTAIL_NCONSTS, MBLS_COOP_F12MUL_NCONSTS, MBLS_COOP_G2ADD_NCONSTS, MBLS_COOP_SMILLER_NCONSTS,
                                     MBLS_COOP_VMFINAL_NCONSTS, MBLS_COOP_HASHG2_NCONSTS, MBLS_COOP_MILLER1_NCONSTS, MBLS_COOP_PAIRING2X2_NCONSTS, MBLS_COOP_HASHG2X4_NCONSTS};
#define COOP_DIM(P) {MBLS_COOP_##P##_LPI, MBLS_COOP_##P##_NSLOTS}
        const uint32_t dims[NP][2] = {COOP_DIM(PAIRING2), COOP_DIM(VMTAIL), COOP_DIM(F12MUL), COOP_DIM(G2ADD), COOP_DIM(SMILLER), COOP_DIM(VMFINAL), COOP_DIM(HASHG2),
                                      COOP_DIM(MILLER1), COOP_DIM(PAIRING2X2), COOP_DIM(HASHG2X4)};
        size_t total = 0;
        for (int p = 0; p < NP; p++) for (int a = 0; a < 3; a++) total += (cnt[p][a] + 3) & ~(size_t)3;       // 16-byte aligned pieces (the rows are read as uint4)
        ok = hipMalloc(&c->d_coop, total * 4) == hipSuccess;
        size_t at = 0;
        for (int p = 0; p < NP && ok; p++) {
            const uint32_t* dp[3];
            for (int a = 0; a < 3 && ok; a++) {
                dp[a] = c->d_coop + at;
                ok = hipMemcpy(c->d_coop + at, src[p][a], cnt[p][a] * 4, hipMemcpyHostToDevice) == hipSuccess;
                at += (cnt[p][a] + 3) & ~(size_t)3;
            }
            c->coop[p].steps = dp[0]; c->coop[p].rows = dp[1]; c->coop[p].consts = dp[2]; c->coop[p].nconsts = nconst[p];
            c->coop[p].lpi = dims[p][0]; c->coop[p].nslots = dims[p][1];
        }
        ctx_default_tuning(c);
    }
    if (!ok) { ctx_free(c); return MBLS_ERR_DEVICE; }
    *out = c; return MBLS_OK;
}
// batches of up to max_items items run their pairing check one wave per item (mbls_coop.h); 0 = always one lane per item
extern "C" int mbls_ctx_set_coop_max_items(mbls_ctx* c, uint64_t max_items) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu); c->coop_max_items = max_items; return MBLS_OK;
}
extern "C" int mbls_ctx_set_coop_hash_max_items(mbls_ctx* c, uint64_t max_items) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu); c->coop_hash_max_items = max_items; return MBLS_OK;
}
extern "C" int mbls_ctx_set_coop_packing(mbls_ctx* c, uint64_t pairing_min_items, uint64_t pairing_max_items, uint64_t hash_min_items) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu); c->coop_pack_min_items = pairing_min_items; c->coop_pack_max_items = pairing_max_items; c->coop_hash_pack_min_items = hash_min_items; return MBLS_OK;
}
// every routing parameter back to its default (what mbls_ctx_create sets, environment included)
static void ctx_default_tuning(mbls_ctx* c) {
    c->coop_max_items = MBLS_DEFAULT_COOP_MAX_ITEMS; c->coop_hash_max_items = MBLS_DEFAULT_COOP_HASH_MAX_ITEMS;
    c->coop_pack_min_items = 1024; c->coop_pack_max_items = 2048; c->coop_hash_pack_min_items = 768;
    hipDeviceProp_t prop;
    c->round_items = 65536;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->round_items = (uint64_t)prop.multiProcessorCount * 4 * WG;
    c->split_max_items = c->round_items / 2; c->fork_max_items = c->round_items / 4 * 3; c->hash2_max_items = c->round_items / 16 * 5;
    const char* e;
    if ((e = getenv("MBLS_COOP_MAX_ITEMS"))) c->coop_max_items = strtoull(e, nullptr, 10);
    if ((e = getenv("MBLS_COOP_HASH_MAX_ITEMS"))) c->coop_hash_max_items = strtoull(e, nullptr, 10);
    if ((e = getenv("MBLS_SPLIT_MAX_ITEMS"))) c->split_max_items = strtoull(e, nullptr, 10);
    if ((e = getenv("MBLS_FORK_MAX_ITEMS"))) c->fork_max_items = strtoull(e, nullptr, 10);
    if ((e = getenv("MBLS_HASH2_MAX_ITEMS"))) c->hash2_max_items = strtoull(e, nullptr, 10);
    c->secret_ops_fast = getenv("MBLS_UNSAFE_SECRET_OPS") != nullptr;
    c->tracks_min_rest = MBLS_DEFAULT_TRACKS_MIN_REST; c->tracks_side_max = c->round_items / 4;
    if ((e = getenv("MBLS_TRACKS_MIN_REST"))) c->tracks_min_rest = strtoull(e, nullptr, 10);
    if ((e = getenv("MBLS_TRACKS_SIDE_MAX"))) c->tracks_side_max = strtoull(e, nullptr, 10);
}
extern "C" int mbls_ctx_reset_tuning(mbls_ctx* c) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    ctx_default_tuning(c); return MBLS_OK;
}
// items per round of the one-lane kernels (default: CUs x 4 SIMDs x 64 lanes); batches above it have their remainder routed as a batch of its own
extern "C" int mbls_ctx_set_round_items(mbls_ctx* c, uint64_t items) {
    if (!c || (items && items % WG)) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!items) {
        hipDeviceProp_t prop;
        HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
        items = (uint64_t)prop.multiProcessorCount * 4 * WG;
    }
    c->round_items = items; c->split_max_items = items / 2; c->fork_max_items = items / 4 * 3; c->hash2_max_items = items / 16 * 5; c->tracks_side_max = items / 4; return MBLS_OK;
}
// one-lane path: batches of up to split_max_items items walk their two Miller pairs on two lanes (never above half a round); up to
// fork_max_items items the three front phases run side by side. Defaults: round / 2 and 3/4 of a round (measured: side by side costs 26.6 ms
// at 57 344 items and 30.2 at 65 535, in a row 26.1 and 26.4; at 32 768 it is 17.1 against 19.3); 0 = never.
extern "C" int mbls_ctx_set_lane_shaping(mbls_ctx* c, uint64_t split_max_items, uint64_t fork_max_items) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    c->split_max_items = split_max_items > c->round_items / 2 ? c->round_items / 2 : split_max_items;
    c->fork_max_items = fork_max_items; return MBLS_OK;
}
extern "C" void mbls_ctx_destroy(mbls_ctx* c) {
    if (!c) return;
    {
        mbls_lock lk(c->mu); (void)hipSetDevice(c->device); (void)hipDeviceSynchronize();
        for (mbls_keytable* t : c->tables) {      // tables that outlive their context become empty shells: mbls_keytable_destroy only frees the handle
            if (t->d_recs) (void)hipFree(t->d_recs);
            if (t->ev) (void)hipEventDestroy(t->ev);
            t->d_recs = nullptr; t->ev = nullptr; t->size = t->cap = 0; t->c = nullptr;
        }
        c->tables.clear();
    }
    ctx_free(c);
}
extern "C" const char* mbls_last_error(mbls_ctx* c) { return c ? c->err : "null context"; }
extern "C" int mbls_ctx_reserve(mbls_ctx* c, uint64_t max_items) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t want = ((max_items + 1 + WG - 1) / WG) * WG;     // +1: the extra (sig, -G1) lane of the n-pairing paths
    if (want <= c->cap) return MBLS_OK;
    if (c->d_w) { (void)hipFree(c->d_w); (void)hipFree(c->d_status); (void)hipFree(c->d_results); c->d_w = nullptr; c->d_status = nullptr; c->d_results = nullptr; c->cap = 0; }
    HIPCHK(c, hipMalloc(&c->d_w, (size_t)MBLS_SLOT_TOTAL * 12 * want * 4));
    HIPCHK(c, hipMalloc(&c->d_status, want * 4));
    HIPCHK(c, hipMalloc(&c->d_results, want));
    c->cap = want; return MBLS_OK;
}
// staging for decompressed keys (compressed wire format with a uniform key count)
static int reserve_keys(mbls_ctx* c, uint64_t nkeys) {
    if (nkeys <= c->key_cap) return MBLS_OK;
    if (c->d_keys_xy) { (void)hipFree(c->d_keys_xy); (void)hipFree(c->d_key_flags); c->d_keys_xy = nullptr; c->d_key_flags = nullptr; c->key_cap = 0; }
    HIPCHK(c, hipMalloc(&c->d_keys_xy, nkeys * 96));
    HIPCHK(c, hipMalloc(&c->d_key_flags, nkeys));
    c->key_cap = nkeys; return MBLS_OK;
}
extern "C" int mbls_ctx_reserve_keys(mbls_ctx* c, uint64_t max_keys) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return reserve_keys(c, max_keys);
}
extern "C" int mbls_enable_phase_timing(mbls_ctx* c, int on) { if (!c) return MBLS_ERR_ARGUMENT; mbls_lock lk(c->mu); c->timing = on != 0; return MBLS_OK; }
extern "C" int mbls_last_phase_ms(mbls_ctx* c, float ms[MBLS_N_PHASES]) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    for (int i = 0; i < MBLS_N_PHASES; i++) ms[i] = c->phase_ms[i];
    return MBLS_OK;
}

// a view of one of the context's grow-only staging buffers (host-buffer entry points; the lock makes the reuse safe)
struct sbuf {
    mbls_ctx* c; int slot; void* p = nullptr;
    sbuf(mbls_ctx* c_, int slot_) : c(c_), slot(slot_) {}
    hipError_t alloc(size_t bytes) {
        if (!bytes) bytes = 1;
        auto& s = c->stage[slot];
        if (s.cap < bytes) {
            if (s.p) { (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
            size_t want = bytes < 4096 ? 4096 : bytes + bytes / 8;
            hipError_t e = hipMalloc(&s.p, want); if (e != hipSuccess) return e;
            s.cap = want;
        }
        p = s.p; return hipSuccess;
    }
    hipError_t up(const void* h, size_t bytes) { hipError_t e = alloc(bytes); if (e != hipSuccess || !bytes) return e; return hipMemcpy(p, h, bytes, hipMemcpyHostToDevice); }
    hipError_t down(void* h, size_t bytes) { return bytes ? hipMemcpy(h, p, bytes, hipMemcpyDeviceToHost) : hipSuccess; }
    template <typename T> T* as() { return (T*)p; }
};

// cross-stream ordering of the shared workspace: a call on stream s first waits for the last asynchronous user on another stream
static int ws_acquire(mbls_ctx* c, hipStream_t s) {
    if (c->ws_pending && c->ws_stream != s) HIPCHK(c, hipStreamWaitEvent(s, c->ws_ev, 0));
    return MBLS_OK;
}
static int ws_release(mbls_ctx* c, hipStream_t s) {
    HIPCHK(c, hipEventRecord(c->ws_ev, s)); c->ws_stream = s; c->ws_pending = true; return MBLS_OK;
}

// where an item's public keys come from
struct keysrc {
    const uint8_t* d_pks = nullptr; int fmt = MBLS_PK_UNCOMPRESSED; const uint32_t* d_off = nullptr;      // wire bytes
    const uint32_t* d_recs = nullptr; uint64_t tsize = 0; const uint32_t* d_idx = nullptr; bool indexed = false;   // key table
    const mbls_keytable* tab = nullptr;
};
// readers of a key table on stream s wait (on the device) for the last asynchronous append made on another stream
static int table_acquire(mbls_ctx* c, const mbls_keytable* t, hipStream_t s) {
    if (t && t->pending && t->ev_stream != s) HIPCHK(c, hipStreamWaitEvent(s, t->ev, 0));
    return MBLS_OK;
}

// ---- routing as data (include/mbls.h, mbls_plan_batch): the decisions of the verification pipeline as pure functions of the limits and the batch size. The
// pipeline below acts on exactly these structures, so what mbls_plan_batch reports is what runs.
static mbls_limits ctx_limits(const mbls_ctx* c) {
    mbls_limits L;
    L.round_items = c->round_items; L.coop_max_items = c->coop_max_items; L.coop_hash_max_items = c->coop_hash_max_items;
    L.coop_pack_min_items = c->coop_pack_min_items; L.coop_pack_max_items = c->coop_pack_max_items; L.coop_hash_pack_min_items = c->coop_hash_pack_min_items;
    L.split_max_items = c->split_max_items; L.fork_max_items = c->fork_max_items; L.hash2_max_items = c->hash2_max_items;
    L.tracks_min_rest = c->tracks_min_rest; L.tracks_side_max = c->tracks_side_max;
    return L;
}
extern "C" void mbls_default_limits(uint64_t round_items, mbls_limits* out) {
    if (!out) return;
    if (!round_items) round_items = 65536;
    out->round_items = round_items; out->coop_max_items = MBLS_DEFAULT_COOP_MAX_ITEMS; out->coop_hash_max_items = MBLS_DEFAULT_COOP_HASH_MAX_ITEMS;
    out->coop_pack_min_items = 1024; out->coop_pack_max_items = 2048; out->coop_hash_pack_min_items = 768;
    out->split_max_items = round_items / 2; out->fork_max_items = round_items / 4 * 3; out->hash2_max_items = round_items / 16 * 5;
    out->tracks_min_rest = MBLS_DEFAULT_TRACKS_MIN_REST; out->tracks_side_max = round_items / 4;
}
// one pass over n items. no_waves: the pass stays on the lane kernels (a remainder beside a round); fork_max: front phases side by side up to this many items
// (~0: the limits' own); keys_later: the host entries' two-part form (their signature phase always has a stream of its own)
static void plan_pass(const mbls_limits& L, uint64_t n, bool no_waves, uint64_t fork_max, bool keys_later, bool timing, mbls_pass_plan* p) {
    const uint64_t coop_max = no_waves ? 0 : L.coop_max_items, coop_hash_max = no_waves ? 0 : L.coop_hash_max_items;
    const bool split = n > coop_max && n <= L.split_max_items && 2 * n <= L.round_items;       // two lanes per item in the Miller phase
    // (a batch whose pairing check runs on waves but whose message phase does not takes the lane-pair message phase too: 2 n items of workspace)
    const bool hash_pairs = split || (n <= coop_max && n > coop_hash_max && n <= L.split_max_items && 4 * n <= L.round_items);
    p->items = n; p->workspace_items = hash_pairs ? 2 * n : n;
    if (n <= coop_max) p->pairing = (n > L.coop_pack_min_items && n <= L.coop_pack_max_items) ? MBLS_PAIRING_WAVE_X2 : MBLS_PAIRING_WAVE;
    else if (split) p->pairing = 4 * n <= L.round_items ? MBLS_PAIRING_LANES4 : MBLS_PAIRING_LANES2;
    else p->pairing = MBLS_PAIRING_LANE;
    // the message phase: lane pairs (k_hash2) where 2 n workspace items are there and the batch is above the wave engine's range but at most hash2_max_items (5/16 of
    // a round: 2 n lanes for the messages beside n for the keys, with the signatures behind them, are still resident together; at a third of a round and above the
    // doubled message phase pushes the key sums behind it: 20 480 items 16.1 -> 14.6 ms, 21 845 items 16.3 -> 16.7); one wave per item (four items per wave above the
    // packing limit) up to coop_hash_max_items -- the measured crossover, for every caller --; one lane per item otherwise
    const bool waves = n <= coop_hash_max && n <= coop_max;
    if (hash_pairs && !waves && n <= L.split_max_items && n <= L.hash2_max_items) p->message = MBLS_MESSAGE_LANES2;
    else if (waves) p->message = n > L.coop_hash_pack_min_items ? MBLS_MESSAGE_WAVE_X4 : MBLS_MESSAGE_WAVE;
    else p->message = MBLS_MESSAGE_LANE;
    p->sig_subgroup_from_miller_loop = n > coop_max ? 1u : 0u;     // one lane (or lane pairs) per item: the signature's subgroup test comes out of the Miller loop
    const bool fork = !timing && n <= (fork_max == ~0ull ? L.fork_max_items : fork_max);
    // above a quarter of a round the three chains no longer fit the chip side by side (and the message phase runs on n lanes, its longest form): the signature
    // phase -- the shortest -- then follows the key sum on the caller's stream, beside the message phase: 32 768 items 17.1 -> 16.5 ms
    const bool sig_side = fork && (keys_later || 4 * n <= L.round_items);
    p->front = !fork ? MBLS_FRONT_IN_A_ROW : sig_side ? MBLS_FRONT_ALL_BESIDE : MBLS_FRONT_MESSAGE_BESIDE;
}
// The batch as the caller sees it. n = q rounds + r items (0 < r < round): a batch of q rounds + r items would cost q + 1 rounds of every kernel, so
//   r < tracks_min_rest: the q rounds as one launch per kernel, then the r items as a batch of their own that takes the route of its size (wave engine up to coop_max_items);
//   r >= tracks_min_rest: the q - 1 rounds in front, then the LAST round and the remainder on TWO TRACKS side by side, each with its own part of the workspace and its
//   own streams, so that the SIMDs one leaves idle take waves of the other -- up to tracks_side_max (and a quarter of a round) the round on track 0 and the remainder,
//   on the lane-pair forms whatever its size, on track 1 (69 120 ... 76 000 items 33.7 ms against 34.5 ... 40 in a row; scripts/dbg/rest_probe.py, rest_probe2.py),
//   above it two equal halves of (R + r) / 2 items (100 000 items 51.6 -> 46.5 ms; scripts/dbg/tracks_probe.py).
// (The phase timers describe a single pass: no cut while they are on -- the caller passes one_pass.)
static void plan_batch(const mbls_limits& L, uint64_t n, bool one_pass, bool timing, mbls_batch_plan* b) {
    memset(b, 0, sizeof(*b));
    const uint64_t R = L.round_items;
    auto pass = [&](int i, uint64_t first, uint64_t items, uint32_t stage, uint32_t trk, uint64_t ws_first, bool no_waves, uint64_t fork_max) {
        plan_pass(L, items, no_waves, fork_max, false, timing, &b->pass[i]);
        b->pass[i].first_item = first; b->pass[i].stage = stage; b->pass[i].track = trk; b->pass[i].workspace_first = ws_first;
    };
    if (one_pass || !R || n <= R || n % R == 0) { b->mode = MBLS_BATCH_ONE_PASS; b->n_passes = 1; pass(0, 0, n, 0, 0, 0, false, ~0ull); return; }
    const uint64_t r = n % R;
    const bool two = L.tracks_min_rest && r >= L.tracks_min_rest && (R + r) / 2 > L.split_max_items && (R + r) / 2 > L.coop_max_items;   // (halves take one workspace item per item)
    if (!two) {
        b->mode = MBLS_BATCH_ROUNDS_THEN_REST; b->n_passes = 2;
        pass(0, 0, n - r, 0, 0, 0, false, ~0ull); pass(1, n - r, r, 1, 0, 0, false, ~0ull);
        return;
    }
    const bool side = r <= L.tracks_side_max && 4 * r <= R;
    const uint64_t lo = n - r - R;                                   // the whole rounds in front: one launch per kernel
    const uint64_t half = side ? R : ((R + r) / 2 + WG - 1) / WG * WG;   // items [lo, lo + half) on track 0, the rest on track 1 (cut at a bitmap word)
    int i = 0; uint32_t stage = 0;
    if (lo) { pass(i++, 0, lo, stage++, 0, 0, false, ~0ull); }
    b->mode = side ? MBLS_BATCH_ROUND_BESIDE_REST : MBLS_BATCH_TWO_HALVES;
    // the halves' front phases side by side only while a half is at most 19/32 of a round and no round runs in front (scripts/dbg/tracks_probe.py: r = 6 144 ... 10 240
    // 36.7 against 37.7 ms; r = 18 432 ... 30 720 in a row 38.0 ... 40.1 against 38.8 ... 42.0; behind a round 64.4 against 67.1 at r = 10 240 ... 20 480); in side
    // mode each part follows the rule of its own size
    const uint64_t fm = side ? ~0ull : ((lo == 0 && half <= R / 32 * 19) ? L.fork_max_items : 0);
    pass(i++, lo, half, stage, 0, 0, false, fm);
    pass(i++, lo + half, n - lo - half, stage, 1, half, side, fm);
    b->n_passes = (uint32_t)i;
}
extern "C" int mbls_plan_batch(const mbls_limits* limits, uint64_t n, mbls_batch_plan* out) {
    if (!limits || !out || !n) return MBLS_ERR_ARGUMENT;
    plan_batch(*limits, n, false, false, out); return MBLS_OK;
}
// the workspace items the plan of n items needs when every item has k keys in a layout that allows the eight-lane key sum (see pass_key_split): what
// verify_pipeline reserves before it queues the first pass -- mbls_ctx_reserve(ctx, this) beforehand keeps every allocation out of the call
extern "C" uint64_t mbls_plan_workspace_items(const mbls_limits* limits, uint64_t n, uint32_t k, int split_layout) {
    if (!limits || !n) return 0;
    mbls_batch_plan bp; plan_batch(*limits, n, false, false, &bp);
    uint64_t need = 0;
    for (uint32_t i = 0; i < bp.n_passes; i++) {
        const mbls_pass_plan& pp = bp.pass[i];
        const bool on_waves = pp.pairing == MBLS_PAIRING_WAVE || pp.pairing == MBLS_PAIRING_WAVE_X2;
        const bool ksplit = split_layout && on_waves && k >= 4 * MBLS_KEY_SPLIT && k % MBLS_KEY_SPLIT == 0;
        const uint64_t w = ksplit && pp.items * (1 + MBLS_KEY_SPLIT) > pp.workspace_items ? pp.items * (1 + MBLS_KEY_SPLIT) : pp.workspace_items;
        if (pp.workspace_first + w > need) need = pp.workspace_first + w;
    }
    return need;
}
#ifdef MBLS_COOP_PROFILE
extern "C" int mbls_coop_profile_read(unsigned long long out[64], int reset) {     // dev builds only (not declared in mbls.h)
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mbls_coop_prof), 64 * 8) != hipSuccess) return MBLS_ERR_DEVICE;
    if (reset) { unsigned long long z[64] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(mbls_coop_prof), z, 64 * 8) != hipSuccess) return MBLS_ERR_DEVICE; }
    return MBLS_OK;
}
#endif
extern "C" int mbls_ctx_get_limits(mbls_ctx* c, mbls_limits* out) {
    if (!c || !out) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu); *out = ctx_limits(c); return MBLS_OK;
}
// A TRACK = what one pass of the pipeline owns besides the caller's stream: where its items start in the workspace (and in the status words / key staging that go
// with it), the side streams of its front phases and the events that join them. Track 0 is the context's own set; verify_pipeline runs a second one beside it.
struct track {
    uint64_t ws_off = 0;
    hipStream_t sb = nullptr, sc = nullptr, sd = nullptr;
    hipEvent_t ev2 = nullptr, ev3 = nullptr;
    bool ws_sync = true;              // order the pass against the workspace's previous user and record its own end (false: the caller does both around its tracks)
    uint64_t fork_max = ~0ull;        // front phases side by side up to this many items (~0: the context's fork_max_items)
    bool no_waves = false;            // the pass stays on the lane kernels whatever its size (a remainder BESIDE a round: the wave engine's 40 KB-of-LDS waves would wait
                                      // for SIMDs the round's 512-register waves hold, and its latency advantage is worth nothing next to a 26 ms round)
};
// The eight-lane key sum of a pass on the wave engine (see verify_pipeline_one) and the workspace items the pass then needs: n items + 8 n partial sums. Both are
// functions of the pass plan, the key source and the keys per item -- verify_pipeline sizes the workspace of a whole plan with them BEFORE the first pass is queued.
static bool pass_key_split(const mbls_pass_plan& pp, const keysrc& ks, uint32_t k) {
    const bool on_waves = pp.pairing == MBLS_PAIRING_WAVE || pp.pairing == MBLS_PAIRING_WAVE_X2;
    const bool staged = !ks.indexed && ks.fmt == MBLS_PK_COMPRESSED && !ks.d_off && k > 1;
    return on_waves && !staged && !ks.d_off && k >= 4 * MBLS_KEY_SPLIT && k % MBLS_KEY_SPLIT == 0 &&
           (ks.indexed || (ks.fmt == MBLS_PK_UNCOMPRESSED && (((uintptr_t)ks.d_pks) & 3u) == 0));
}
static uint64_t pass_ws_items(const mbls_pass_plan& pp, const keysrc& ks, uint32_t k, uint64_t n) {
    const uint64_t split = pass_key_split(pp, ks, k) ? n + n * MBLS_KEY_SPLIT : 0;
    return split > pp.workspace_items ? split : pp.workspace_items;
}
static track track0(mbls_ctx* c) { track t; t.sb = c->hs_b; t.sc = c->hs_c; t.sd = c->hs_d; t.ev2 = c->hs_ev2; t.ev3 = c->hs_ev3; return t; }
static int verify_pipeline_one(mbls_ctx* c, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff, const keysrc& ks,
                               uint64_t n, uint32_t k, int mode, uint8_t* d_results, uint64_t* d_bitmap,
                               uint32_t* d_status, hipStream_t s, int part = 0, const track* tkp = nullptr, bool no_growth = false) {
    const int fmt = ks.fmt; const uint32_t* d_off = ks.d_off;
    if (!c || (fmt != MBLS_PK_COMPRESSED && fmt != MBLS_PK_UNCOMPRESSED)) return MBLS_ERR_ARGUMENT;
    if (n == 0) return MBLS_OK;
    const bool have_keys = ks.indexed ? (ks.d_idx != nullptr) : (ks.d_pks != nullptr);
    if (!d_sigs || (!d_msgs && msg_len && !d_moff) || !d_results || (!have_keys && (k || d_off) && part != 1)) ARGFAIL(c, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    const track tk = tkp ? *tkp : track0(c);
    bool tm = c->timing;
    // what this pass does is decided by plan_pass (the function mbls_plan_batch reports from): forms of the pairing check and of the message phase, workspace, forks
    mbls_pass_plan pp; plan_pass(ctx_limits(c), n, tk.no_waves, tk.fork_max, part != 0, tm, &pp);
    const bool on_waves = pp.pairing == MBLS_PAIRING_WAVE || pp.pairing == MBLS_PAIRING_WAVE_X2;
    const bool split = pp.pairing == MBLS_PAIRING_LANES2 || pp.pairing == MBLS_PAIRING_LANES4;     // lane pairs / quads in the Miller phase, pairs in the final exponentiation
    const bool hash_pairs = pp.workspace_items == 2 * n;
    const int hform = pp.message == MBLS_MESSAGE_LANE ? HASH_FORM_LANE : pp.message == MBLS_MESSAGE_LANES2 ? HASH_FORM_PAIR : HASH_FORM_WAVE;
    bool staged = !ks.indexed && (fmt == MBLS_PK_COMPRESSED) && !d_off && k > 1;   // lane-per-key decompression, then the per-item sums
    // batches on the wave engine are latency: the key sum of 128 keys is a 2 ms chain on one lane -- eight lanes take an eighth of the keys each (items of k / 8
    // keys lie back to back just like that) and k_apk_combine adds their sums: 0.4 ms
    // (pass_key_split: the same predicate verify_pipeline sizes the workspace of a several-pass plan with -- a pass that is not the first of its call finds its
    // space reserved and never grows the workspace under a pass in flight: `no_growth` passes that would not fit fall back to the one-lane key sum)
    bool key_split = pass_key_split(pp, ks, k);
    if (key_split && no_growth && tk.ws_off + pass_ws_items(pp, ks, k, n) > c->cap) key_split = false;
    const uint64_t nsub = key_split ? n * MBLS_KEY_SPLIT : 0;
    const uint64_t ws_items = key_split && n + nsub > pp.workspace_items ? n + nsub : pp.workspace_items;
    if (no_growth && tk.ws_off + ws_items > c->cap) ARGFAIL(c, "internal: a later pass of a plan would have to grow the workspace");
    int rc = mbls_ctx_reserve(c, tk.ws_off + ws_items); if (rc) return rc;        // (a pass on a second track finds its space reserved: no growth under the first)
    mbls_ws ws; ws.w = c->d_w + tk.ws_off; ws.stride = c->cap;
    uint32_t* st = d_status ? d_status : c->d_status + tk.ws_off;
    unsigned g = nblk(n);
    if (staged) { rc = reserve_keys(c, (tk.ws_off + n) * (uint64_t)k); if (rc) return rc; }
    uint32_t* const keys_xy = staged ? c->d_keys_xy + 24 * (uint64_t)k * tk.ws_off : nullptr;
    uint8_t* const key_flags = staged ? c->d_key_flags + (uint64_t)k * tk.ws_off : nullptr;
    // The status words are zeroed and every phase ORs its bits in (atomically), so the three phases before the Miller loop can run in
    // any order -- and, for batches that leave most SIMDs idle (n <= 2^14: at most a quarter of the one-wave-per-SIMD slots), side by
    // side on the context's own streams: keys | signature | message, joined before the Miller loop (latency 35.7 -> ~30 ms).
    // part 1 / part 2 (host-buffer entry points): the signature and message phases are queued first (part 1, the keys may be
    // null), the caller then uploads the keys on another stream and makes this one wait, and part 2 queues the rest.
    const bool keys_later = part != 0;
    const bool fused_sig = pp.sig_subgroup_from_miller_loop != 0;     // the subgroup test of the signature comes out of the Miller loop, k_sig only decodes
    const bool fork = pp.front != MBLS_FRONT_IN_A_ROW;
    // (host-buffer entries: hs_b carries the key upload, so their signature phase has a stream of its own)
    const bool sig_side = pp.front == MBLS_FRONT_ALL_BESIDE;
    hipStream_t s_sig = sig_side ? (part == 0 ? tk.sb : tk.sd) : s, s_msg = fork ? tk.sc : s;
    if (part != 2) {
        if (tk.ws_sync) { rc = ws_acquire(c, s); if (rc) return rc; }
        HIPCHK(c, hipMemsetAsync(st, 0, 4 * n, s));
        if (fork) {
            HIPCHK(c, hipEventRecord(tk.ev2, s));
            if (s_sig != s) HIPCHK(c, hipStreamWaitEvent(s_sig, tk.ev2, 0));
            HIPCHK(c, hipStreamWaitEvent(s_msg, tk.ev2, 0));
        }
    }
    // batches on the wave engine are latency: their signature chain (decode + subgroup ladder, 2.1 ms on one lane) is the longest of the three front chains --
    // two lanes per signature there (k_sig2: 1.6 ms)
    auto launch_sig = [&]() {
        if (on_waves && !fused_sig) hipLaunchKernelGGL(k_sig2, dim3(nblk(2 * n)), dim3(WG), 0, s_sig, ws, d_sigs, st, n);
        else hipLaunchKernelGGL(k_sig, dim3(g), dim3(WG), 0, s_sig, ws, d_sigs, st, n, fused_sig ? 0 : 1);
    };
    if (part == 1) {
        launch_sig();
        launch_hash(c, ws, d_msgs, msg_len, d_moff, st, n, s_msg, hash_pairs, hform);
        HIPCHK(c, hipGetLastError());
        return MBLS_OK;
    }
    if (tm) HIPCHK(c, hipEventRecord(c->ev[0], s));
    if (ks.indexed) { rc = table_acquire(c, ks.tab, s); if (rc) return rc; }
    if (key_split) {
        mbls_ws wsub = ws; wsub.w += n;                                  // sub-item t = workspace item n + t (APK slots only; nothing else uses them up there)
        uint32_t* st_sub = c->d_status + tk.ws_off + n;                  // their status words: behind the items' own
        HIPCHK(c, hipMemsetAsync(st_sub, 0, 4 * nsub, s));
        if (ks.indexed)
            hipLaunchKernelGGL(k_aggregate_indexed_d, dim3(nblk(nsub)), dim3(WG), 0, s, wsub, ks.d_recs, ks.tsize, ks.d_idx, (const uint32_t*)nullptr, k / MBLS_KEY_SPLIT,
                               MBLS_MODE_VERIFY, st_sub, nsub);
        else
            hipLaunchKernelGGL(k_aggregate_raw_d, dim3(nblk(nsub)), dim3(WG), 0, s, wsub, ks.d_pks, (const uint32_t*)nullptr, k / MBLS_KEY_SPLIT, MBLS_MODE_VERIFY, st_sub, nsub);
        hipLaunchKernelGGL(k_apk_combine, dim3(g), dim3(WG), 0, s, ws, n, mode, (const uint32_t*)st_sub, st);
    } else if (ks.indexed)
        hipLaunchKernelGGL(k_aggregate_indexed_d, dim3(g), dim3(WG), 0, s, ws, ks.d_recs, ks.tsize, ks.d_idx, d_off, k, mode, st, n);
    else if (staged) {
        hipLaunchKernelGGL(k_pk_decompress, dim3(nblk(n * (uint64_t)k)), dim3(WG), 0, s, ks.d_pks, n * (uint64_t)k, keys_xy, key_flags);
        hipLaunchKernelGGL(k_aggregate_decoded, dim3(g), dim3(WG), 0, s, ws, (const uint32_t*)keys_xy, (const uint8_t*)key_flags, k, mode, st, n);
    } else
        launch_aggregate(ws, ks.d_pks, d_off, k, fmt, mode, st, n, s);
    if (tm) HIPCHK(c, hipEventRecord(c->ev[1], s));
    if (!keys_later) launch_sig();
    if (tm) HIPCHK(c, hipEventRecord(c->ev[2], s));
    if (!keys_later) launch_hash(c, ws, d_msgs, msg_len, d_moff, st, n, s_msg, hash_pairs, hform);
    if (tm) HIPCHK(c, hipEventRecord(c->ev[3], s));
    if (fork) {      // join
        if (s_sig != s) { HIPCHK(c, hipEventRecord(tk.ev2, s_sig)); HIPCHK(c, hipStreamWaitEvent(s, tk.ev2, 0)); }
        HIPCHK(c, hipEventRecord(tk.ev3, s_msg)); HIPCHK(c, hipStreamWaitEvent(s, tk.ev3, 0));
    }
    if (on_waves) {
        // small batch: one WAVE per item walks the Miller loop and the final exponentiation with its lanes side by side (mbls_coop.h)
        coop_run(c, pp.pairing == MBLS_PAIRING_WAVE_X2 ? COOP_PAIRING2X2 : COOP_PAIRING2, ws, (uint64_t)0, (uint64_t)1, (uint64_t)0, n, st, d_results, COOP_RES_ITEM, s);
        if (tm) { HIPCHK(c, hipEventRecord(c->ev[4], s)); HIPCHK(c, hipEventRecord(c->ev[5], s)); }
    } else {
        if (split) {
            const int lpp = pp.pairing == MBLS_PAIRING_LANES4 ? 2 : 1;         // a quarter of a round or less: two lanes per pair, products in pairs
            if (lpp == 2) hipLaunchKernelGGL(k_miller_split4, dim3(nblk(4 * n)), dim3(WG), 0, s, ws, n);
            else hipLaunchKernelGGL(k_miller_split, dim3(nblk(2 * n)), dim3(WG), 0, s, ws, n);
            hipLaunchKernelGGL(k_f12_tree_d, dim3(g), dim3(WG), 0, s, ws, 2 * n, n);
            hipLaunchKernelGGL(k_sig_verdict, dim3(g), dim3(WG), 0, s, ws, st, n, n, 1);
        } else {
            hipLaunchKernelGGL(k_miller, dim3(g), dim3(WG), 0, s, ws, n);
            if (fused_sig) hipLaunchKernelGGL(k_sig_verdict, dim3(g), dim3(WG), 0, s, ws, st, n, (uint64_t)0, 0);
        }
        if (tm) HIPCHK(c, hipEventRecord(c->ev[4], s));
        if (split) hipLaunchKernelGGL(k_final2, dim3(nblk(2 * n)), dim3(WG), 0, s, ws, st, d_results, n);      // two lanes per item here too
        else hipLaunchKernelGGL(k_final, dim3(g), dim3(WG), 0, s, ws, st, d_results, n);
        if (tm) HIPCHK(c, hipEventRecord(c->ev[5], s));
    }
    if (d_bitmap) hipLaunchKernelGGL(k_pack, dim3(g), dim3(WG), 0, s, d_results, d_bitmap, n);
    if (tm) {
        HIPCHK(c, hipEventRecord(c->ev[6], s));
        HIPCHK(c, hipEventSynchronize(c->ev[6]));
        for (int i = 0; i < MBLS_N_PHASES; i++) HIPCHK(c, hipEventElapsedTime(&c->phase_ms[i], c->ev[i], c->ev[i + 1]));
    }
    HIPCHK(c, hipGetLastError());
    return tk.ws_sync ? ws_release(c, s) : MBLS_OK;
}
// Items [lo, n) of a batch as a batch of their own: uniform layouts advance the base pointers, offset tables are absolute (their slice goes
// with the unmoved base); lo is a multiple of 64, so the bitmap advances by whole words.
static int verify_pipeline_from(mbls_ctx* c, uint64_t lo, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff, const keysrc& ks,
                                uint64_t n, uint32_t k, int mode, uint8_t* d_results, uint64_t* d_bitmap, uint32_t* d_status, hipStream_t s, const track* tk = nullptr,
                                bool no_growth = false) {
    keysrc t = ks;
    const size_t unit = ks.fmt == MBLS_PK_COMPRESSED ? 48 : 96;
    if (ks.d_off) t.d_off = ks.d_off + lo;
    else { if (ks.d_pks) t.d_pks = ks.d_pks + unit * (uint64_t)k * lo; if (ks.d_idx) t.d_idx = ks.d_idx + (uint64_t)k * lo; }
    return verify_pipeline_one(c, d_sigs + 96 * lo, (d_moff || !d_msgs) ? d_msgs : d_msgs + (uint64_t)msg_len * lo, msg_len, d_moff ? d_moff + lo : nullptr, t,
                               n - lo, k, mode, d_results + lo, d_bitmap ? d_bitmap + lo / 64 : nullptr, d_status ? d_status + lo : nullptr, s, 0, tk, no_growth);
}
// The batch as the caller sees it: plan_batch decides (see there), this function carries the plan out. Passes of one stage run side by side -- track 0 on the
// caller's stream, track 1 on the context's second set of streams, forked from and joined to the caller's stream --, stages one after the other.
static int verify_pipeline(mbls_ctx* c, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff, const keysrc& ks,
                           uint64_t n, uint32_t k, int mode, uint8_t* d_results, uint64_t* d_bitmap,
                           uint32_t* d_status, hipStream_t s, int part = 0) {
    if (!c || part != 0 || n == 0)
        return verify_pipeline_one(c, d_sigs, d_msgs, msg_len, d_moff, ks, n, k, mode, d_results, d_bitmap, d_status, s, part);
    mbls_batch_plan bp; plan_batch(ctx_limits(c), n, c->timing, c->timing, &bp);
    if (bp.mode == MBLS_BATCH_ONE_PASS)
        return verify_pipeline_one(c, d_sigs, d_msgs, msg_len, d_moff, ks, n, k, mode, d_results, d_bitmap, d_status, s, 0);
    // one growth of the workspace (and of the staging of decompressed keys) for the whole plan: nothing may be reallocated under a pass in flight
    // (a pass on the wave engine may take the eight-lane key sum: n + 8 n items -- pass_ws_items, the figure the pass itself acts on; the staging of decompressed keys
    // is indexed by ITEM, so it follows the items' extent, not the partial sums')
    uint64_t need = 0, need_items = 0;
    for (uint32_t i = 0; i < bp.n_passes; i++) {
        const uint64_t e = bp.pass[i].workspace_first + pass_ws_items(bp.pass[i], ks, k, bp.pass[i].items); if (e > need) need = e;
        const uint64_t ei = bp.pass[i].workspace_first + bp.pass[i].items; if (ei > need_items) need_items = ei;
    }
    int rc = mbls_ctx_reserve(c, need); if (rc) return rc;
    if (!ks.indexed && ks.fmt == MBLS_PK_COMPRESSED && !ks.d_off && k > 1) { rc = reserve_keys(c, need_items * (uint64_t)k); if (rc) return rc; }
    if (bp.mode == MBLS_BATCH_ROUNDS_THEN_REST) {
        rc = verify_pipeline_one(c, d_sigs, d_msgs, msg_len, d_moff, ks, bp.pass[0].items, k, mode, d_results, d_bitmap, d_status, s, 0, nullptr, true); if (rc) return rc;
        return verify_pipeline_from(c, bp.pass[1].first_item, d_sigs, d_msgs, msg_len, d_moff, ks, n, k, mode, d_results, d_bitmap, d_status, s, nullptr, true);
    }
    // two tracks: [rounds in front,] then two passes side by side
    uint32_t i = 0;
    if (bp.n_passes == 3) { rc = verify_pipeline_one(c, d_sigs, d_msgs, msg_len, d_moff, ks, bp.pass[0].items, k, mode, d_results, d_bitmap, d_status, s, 0, nullptr, true); if (rc) return rc; i = 1; }
    const mbls_pass_plan& pa = bp.pass[i]; const mbls_pass_plan& pb = bp.pass[i + 1];
    const bool side = bp.mode == MBLS_BATCH_ROUND_BESIDE_REST;
    const uint64_t fm = side ? ~0ull : (pa.front == MBLS_FRONT_IN_A_ROW ? 0 : c->fork_max_items);
    rc = ws_acquire(c, s); if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->t1_ev, s)); HIPCHK(c, hipStreamWaitEvent(c->t1_s, c->t1_ev, 0));
    track ta = track0(c); ta.ws_sync = false; ta.fork_max = fm;
    track tb; tb.no_waves = side; tb.fork_max = fm; tb.ws_off = pb.workspace_first; tb.sb = c->t1_b; tb.sc = c->t1_c; tb.sd = c->t1_b; tb.ev2 = c->t1_ev2; tb.ev3 = c->t1_ev3; tb.ws_sync = false;
    const uint64_t lo = pa.first_item, mid = pb.first_item;
    if (side) {          // the round first: its kernels fill the chip, the remainder's waves take what they leave between them
        rc = verify_pipeline_from(c, lo, d_sigs, d_msgs, msg_len, d_moff, ks, mid, k, mode, d_results, d_bitmap, d_status, s, &ta, true);
        if (!rc) rc = verify_pipeline_from(c, mid, d_sigs, d_msgs, msg_len, d_moff, ks, n, k, mode, d_results, d_bitmap, d_status, c->t1_s, &tb, true);
    } else {
        rc = verify_pipeline_from(c, mid, d_sigs, d_msgs, msg_len, d_moff, ks, n, k, mode, d_results, d_bitmap, d_status, c->t1_s, &tb, true);
        if (!rc) rc = verify_pipeline_from(c, lo, d_sigs, d_msgs, msg_len, d_moff, ks, mid, k, mode, d_results, d_bitmap, d_status, s, &ta, true);
    }
    // join: the caller's stream ends when both tracks have (also on an error path: nothing of the call stays in flight unordered)
    hipError_t e1 = hipEventRecord(c->t1_ev, c->t1_s), e2 = hipStreamWaitEvent(s, c->t1_ev, 0);
    if (rc) return rc;
    HIPCHK(c, e1); HIPCHK(c, e2);
    return ws_release(c, s);
}
extern "C" int mbls_ctx_set_tracks(mbls_ctx* c, uint64_t min_rest_items, uint64_t side_max_items) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu); c->tracks_min_rest = min_rest_items; c->tracks_side_max = side_max_items; return MBLS_OK;
}
extern "C" int mbls_fast_aggregate_verify_batch_device(mbls_ctx* c, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_len,
        const uint64_t* d_moff, const uint8_t* d_pks, int fmt, const uint32_t* d_off, uint64_t n, uint32_t k, uint8_t* d_results, uint64_t* d_bitmap,
        uint32_t* d_status, void* stream) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    keysrc ks; ks.d_pks = d_pks; ks.fmt = fmt; ks.d_off = d_off;
    return verify_pipeline(c, d_sigs, d_msgs, msg_len, d_moff, ks, n, k, MBLS_MODE_FAST_AGGREGATE, d_results, d_bitmap, d_status, (hipStream_t)stream);
}
extern "C" int mbls_verify_batch_device(mbls_ctx* c, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff,
        const uint8_t* d_pks, int fmt, uint64_t n, uint8_t* d_results, uint64_t* d_bitmap, uint32_t* d_status, void* stream) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    keysrc ks; ks.d_pks = d_pks; ks.fmt = fmt;
    return verify_pipeline(c, d_sigs, d_msgs, msg_len, d_moff, ks, n, 1, MBLS_MODE_VERIFY, d_results, d_bitmap, d_status, (hipStream_t)stream);
}

// offsets of a ragged key / signature table handed over in host memory: non-decreasing, and the total fits the index type
static bool offsets_ok(const uint32_t* off, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) if (off[i + 1] < off[i]) return false;
    return true;
}
// message ranges: non-decreasing, every message shorter than 2^32 bytes
static bool msg_offsets_ok(const uint64_t* off, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) if (off[i + 1] < off[i] || off[i + 1] - off[i] > 0xFFFFFFFFull) return false;
    return true;
}

// Host buffers in, results out. The keys (or key indices) are 99 % of the bytes: they are uploaded on a second stream while the
// signature and message phases run.
static int verify_host(mbls_ctx* c, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* moff, const uint8_t* pks, int fmt,
                       const mbls_keytable* tab, const uint32_t* idx, const uint32_t* off, uint64_t n, uint32_t k, int mode,
                       uint8_t* results, uint32_t* status) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (n == 0) return MBLS_OK;
    if (moff && !msg_offsets_ok(moff, n)) ARGFAIL(c, "msg_offsets must be non-decreasing, messages below 2^32 bytes");
    const uint64_t msg_first = moff ? moff[0] : 0;                       // the bytes this call uploads: msgs[msg_first .. msg_first + msg_total)
    const size_t msg_total = moff ? (size_t)(moff[n] - moff[0]) : (size_t)msg_len * n;
    if (!sigs || (!msgs && msg_total) || !results) ARGFAIL(c, "null buffer");
    if (fmt != MBLS_PK_COMPRESSED && fmt != MBLS_PK_UNCOMPRESSED) ARGFAIL(c, "pk_format");
    if (tab && tab->c != c) ARGFAIL(c, "key table belongs to another context");
    if (off && !offsets_ok(off, n)) ARGFAIL(c, "offsets must be non-decreasing");
    HIPCHK(c, hipSetDevice(c->device));
    // like the messages, a key offset table may start anywhere (a shard of a larger batch): only keys [off[0], off[n]) are uploaded
    const uint64_t key_first = off ? off[0] : 0;
    uint64_t total_keys = off ? (uint64_t)(off[n] - off[0]) : (uint64_t)k * n;
    const bool indexed = tab != nullptr;
    if (total_keys && !(indexed ? (const void*)idx : (const void*)pks)) ARGFAIL(c, "null key buffer");
    size_t unit = indexed ? 4 : (fmt == MBLS_PK_COMPRESSED ? 48 : 96);
    sbuf ds(c, 0), dm(c, 1), dp(c, 2), doff(c, 3), dr(c, 4), dst(c, 5), dmo(c, 6);
    HIPCHK(c, ds.up(sigs, 96 * n)); HIPCHK(c, dm.up(msgs ? msgs + msg_first : nullptr, msg_total));
    if (off) HIPCHK(c, doff.up(off, 4 * (n + 1)));
    const uint64_t* d_moff = nullptr; const uint8_t* d_msgs = dm.as<uint8_t>();
    if (moff) {                                                          // the table as given; the uploaded bytes start at msgs[moff[0]]
        HIPCHK(c, dmo.up(moff, 8 * (n + 1))); d_moff = dmo.as<uint64_t>(); d_msgs -= msg_first;
    }
    HIPCHK(c, dr.alloc(n)); HIPCHK(c, dst.alloc(4 * n)); HIPCHK(c, dp.alloc(unit * total_keys));
    bool tm = c->timing; c->timing = false;              // the phase timers assume the plain order
    keysrc ks; ks.fmt = fmt; ks.d_off = off ? doff.as<uint32_t>() : nullptr; ks.indexed = indexed;
    if (indexed) { ks.d_recs = tab->d_recs; ks.tsize = tab->size; ks.tab = tab; }
    const uint64_t R = c->round_items;
    const uint64_t n_all = n;
    if (R && n > R && n % R) n -= n % R;                 // the rounds first (parts 1 and 2 below); the remainder as a full pass once the keys are there
    int rc = verify_pipeline(c, ds.as<uint8_t>(), d_msgs, msg_len, d_moff, ks, n, k, mode, dr.as<uint8_t>(), nullptr, dst.as<uint32_t>(), c->hs_a, 1);
    if (!rc) {
        hipError_t e1 = hipSuccess;      // issued after the first two phases were queued: a copy from pageable memory may block the host
        if (total_keys) e1 = hipMemcpyAsync(dp.p, indexed ? (const void*)(idx + key_first) : (const void*)(pks + unit * key_first), unit * total_keys, hipMemcpyHostToDevice, c->hs_b);
        if (e1 == hipSuccess) e1 = hipEventRecord(c->hs_ev, c->hs_b);
        if (e1 == hipSuccess) e1 = hipStreamWaitEvent(c->hs_a, c->hs_ev, 0);
        if (e1 != hipSuccess) { snprintf(c->err, sizeof(c->err), "key upload failed: %s", hipGetErrorString(e1)); rc = MBLS_ERR_DEVICE; }
        if (!rc) {
            if (indexed) ks.d_idx = dp.as<uint32_t>() - key_first; else ks.d_pks = dp.as<uint8_t>() - unit * key_first;     // the table's offsets are absolute
            rc = verify_pipeline(c, ds.as<uint8_t>(), d_msgs, msg_len, d_moff, ks, n, k, mode, dr.as<uint8_t>(), nullptr, dst.as<uint32_t>(), c->hs_a, 2);
            if (!rc && n_all > n)
                rc = verify_pipeline_from(c, n, ds.as<uint8_t>(), d_msgs, msg_len, d_moff, ks, n_all, k, mode, dr.as<uint8_t>(), nullptr, dst.as<uint32_t>(), c->hs_a);
        }
    }
    n = n_all;
    c->timing = tm;
    if (rc) {         // part 1 may be running on the context's streams: nothing of this call is left in flight when it returns an error
        (void)hipStreamSynchronize(c->hs_a); (void)hipStreamSynchronize(c->hs_b); (void)hipStreamSynchronize(c->hs_c); (void)hipStreamSynchronize(c->hs_d);
        c->ws_pending = false;
        return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->hs_a));
    c->ws_pending = false;
    HIPCHK(c, dr.down(results, n));
    if (status) HIPCHK(c, dst.down(status, 4 * n));
    return MBLS_OK;
}
extern "C" int mbls_fast_aggregate_verify_batch(mbls_ctx* c, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* moff,
        const uint8_t* pks, int fmt, const uint32_t* off, uint64_t n, uint32_t k, uint8_t* results, uint32_t* status) {
    return verify_host(c, sigs, msgs, msg_len, moff, pks, fmt, nullptr, nullptr, off, n, k, MBLS_MODE_FAST_AGGREGATE, results, status);
}
extern "C" int mbls_verify_batch(mbls_ctx* c, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* moff, const uint8_t* pks, int fmt,
        uint64_t n, uint8_t* results, uint32_t* status) {
    return verify_host(c, sigs, msgs, msg_len, moff, pks, fmt, nullptr, nullptr, nullptr, n, 1, MBLS_MODE_VERIFY, results, status);
}

// ---- resident key table
extern "C" int mbls_keytable_create(mbls_ctx* c, uint64_t capacity_hint, mbls_keytable** out) {
    if (!c || !out) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    mbls_keytable* t = new (std::nothrow) mbls_keytable();
    if (!t) return MBLS_ERR_DEVICE;
    t->c = c; t->cap = capacity_hint ? capacity_hint : 1024;
    hipError_t e = hipMalloc(&t->d_recs, t->cap * MBLS_KEYREC_DWORDS * 4);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&t->ev, hipEventDisableTiming);
    if (e != hipSuccess) { if (t->d_recs) (void)hipFree(t->d_recs); delete t; HIPCHK(c, e); }
    try { c->tables.push_back(t); } catch (...) { (void)hipFree(t->d_recs); (void)hipEventDestroy(t->ev); delete t; return MBLS_ERR_DEVICE; }
    *out = t; return MBLS_OK;
}
// Either order of destruction is fine: a table whose context went first is an empty shell (mbls_ctx_destroy released its records).
extern "C" void mbls_keytable_destroy(mbls_keytable* t) {
    if (!t) return;
    if (t->c) {
        mbls_ctx* c = t->c;
        mbls_lock lk(c->mu); (void)hipSetDevice(c->device); (void)hipDeviceSynchronize();
        if (t->d_recs) (void)hipFree(t->d_recs);
        if (t->ev) (void)hipEventDestroy(t->ev);
        for (size_t i = 0; i < c->tables.size(); i++) if (c->tables[i] == t) { c->tables.erase(c->tables.begin() + i); break; }
    }
    delete t;
}
// undo of an append (mbls_multi_keytable_append: a replica failed): later appends overwrite the dropped records
static void keytable_truncate(mbls_keytable* t, uint64_t size) {
    if (!t || !t->c) return;
    mbls_lock lk(t->c->mu);
    if (size < t->size) t->size = size;
}
extern "C" uint64_t mbls_keytable_size(const mbls_keytable* t) { if (!t || !t->c) return 0; mbls_lock lk(t->c->mu); return t->size; }
static int keytable_grow(mbls_keytable* t, uint64_t need, hipStream_t s) {
    mbls_ctx* c = t->c;
    if (need <= t->cap) return MBLS_OK;
    uint64_t ncap = t->cap * 2 > need ? t->cap * 2 : need;
    uint32_t* nr = nullptr;
    HIPCHK(c, hipMalloc(&nr, ncap * MBLS_KEYREC_DWORDS * 4));
    // appends in flight on other streams must have written the old records before they are copied, and verifications in flight may
    // still read them when they are freed: the device is drained on both sides of the copy (growth is rare; reserve with capacity_hint)
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && t->size) e = hipMemcpyAsync(nr, t->d_recs, t->size * MBLS_KEYREC_DWORDS * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) t->pending = false;
    if (e != hipSuccess) { (void)hipFree(nr); HIPCHK(c, e); }
    (void)hipFree(t->d_recs); t->d_recs = nr; t->cap = ncap;
    return MBLS_OK;
}
extern "C" int mbls_keytable_append_device(mbls_keytable* t, const uint8_t* d_pks, int fmt, int validate, uint64_t n, uint64_t* first_index,
                                           uint8_t* d_errs, void* stream) {
    if (!t || !t->c) return MBLS_ERR_ARGUMENT;
    mbls_ctx* c = t->c;
    mbls_lock lk(c->mu);
    if ((fmt != MBLS_PK_COMPRESSED && fmt != MBLS_PK_UNCOMPRESSED) || (n && (!d_pks || !d_errs))) ARGFAIL(c, "keytable_append");
    if (t->size + n > 0xFFFFFFFFull) ARGFAIL(c, "key table indices are 32-bit");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = keytable_grow(t, t->size + n, s); if (rc) return rc;
    if (n) {
        // an earlier append on another stream may still be writing its own records: nothing to order (disjoint), but the event below
        // replaces the earlier one, so this stream inherits the earlier append's completion first
        if (t->pending && t->ev_stream != s) HIPCHK(c, hipStreamWaitEvent(s, t->ev, 0));
        hipLaunchKernelGGL(k_keytable_append, dim3(nblk(n)), dim3(WG), 0, s, d_pks, fmt, validate, n, t->d_recs + t->size * MBLS_KEYREC_DWORDS, d_errs);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(t->ev, s)); t->ev_stream = s; t->pending = true;     // readers on other streams wait for this (table_acquire)
    }
    if (first_index) *first_index = t->size;      // published only once everything of the append is in the stream
    t->size += n; return MBLS_OK;
}
static int map_dec_err_g1(int e) { return e == MBLS_DEC_OK ? MBLS_OK : (e == MBLS_DEC_SIZE ? MBLS_ERR_INVALID_G1_SIZE : MBLS_ERR_INVALID_POINT); }
static int map_dec_err_g2(int e) { return e == MBLS_DEC_OK ? MBLS_OK : (e == MBLS_DEC_SIZE ? MBLS_ERR_INVALID_G2_SIZE : MBLS_ERR_INVALID_POINT); }
extern "C" int mbls_keytable_append(mbls_keytable* t, const uint8_t* pks, int fmt, int validate, uint64_t n, uint64_t* first_index, uint8_t* errs) {
    if (!t || !t->c) return MBLS_ERR_ARGUMENT;
    mbls_ctx* c = t->c;
    mbls_lock lk(c->mu);
    if ((fmt != MBLS_PK_COMPRESSED && fmt != MBLS_PK_UNCOMPRESSED) || (n && (!pks || !errs))) ARGFAIL(c, "keytable_append");
    HIPCHK(c, hipSetDevice(c->device));
    sbuf di(c, 0), de(c, 1);
    HIPCHK(c, di.up(pks, (fmt ? 96 : 48) * n)); HIPCHK(c, de.alloc(n));
    int rc = mbls_keytable_append_device(t, di.as<uint8_t>(), fmt, validate, n, first_index, de.as<uint8_t>(), c->hs_a); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, de.down(errs, n));
    if (t->ev_stream == c->hs_a) t->pending = false;
    for (uint64_t i = 0; i < n; i++) errs[i] = (uint8_t)map_dec_err_g1(errs[i]);
    return MBLS_OK;
}
extern "C" int mbls_keytable_get(mbls_keytable* t, uint64_t first, uint64_t n, uint8_t* pks96, uint8_t* errs) {
    if (!t || !t->c) return MBLS_ERR_ARGUMENT;
    mbls_ctx* c = t->c;
    mbls_lock lk(c->mu);
    if (first + n > t->size || (n && (!pks96 || !errs))) ARGFAIL(c, "keytable_get range");
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    sbuf dout(c, 0), de(c, 1); HIPCHK(c, dout.alloc(96 * n)); HIPCHK(c, de.alloc(n));
    { int rc = table_acquire(c, t, c->hs_a); if (rc) return rc; }
    hipLaunchKernelGGL(k_keytable_export, dim3(nblk(n)), dim3(WG), 0, c->hs_a, (const uint32_t*)t->d_recs, first, n, dout.as<uint8_t>(), de.as<uint8_t>());
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, dout.down(pks96, 96 * n)); HIPCHK(c, de.down(errs, n));
    for (uint64_t i = 0; i < n; i++) errs[i] = (uint8_t)map_dec_err_g1(errs[i]);
    return MBLS_OK;
}
extern "C" int mbls_fast_aggregate_verify_batch_indexed_device(mbls_ctx* c, const mbls_keytable* t, const uint8_t* d_sigs, const uint8_t* d_msgs,
        uint32_t msg_len, const uint64_t* d_moff, const uint32_t* d_idx, const uint32_t* d_off, uint64_t n, uint32_t k, uint8_t* d_results, uint64_t* d_bitmap,
        uint32_t* d_status, void* stream) {
    if (!c || !t) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (t->c != c) ARGFAIL(c, "key table belongs to another context");
    keysrc ks; ks.indexed = true; ks.d_recs = t->d_recs; ks.tsize = t->size; ks.d_idx = d_idx; ks.d_off = d_off; ks.tab = t;
    return verify_pipeline(c, d_sigs, d_msgs, msg_len, d_moff, ks, n, k, MBLS_MODE_FAST_AGGREGATE, d_results, d_bitmap, d_status, (hipStream_t)stream);
}
extern "C" int mbls_fast_aggregate_verify_batch_indexed(mbls_ctx* c, const mbls_keytable* t, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len,
        const uint64_t* moff, const uint32_t* idx, const uint32_t* off, uint64_t n, uint32_t k, uint8_t* results, uint32_t* status) {
    if (!c || !t) return MBLS_ERR_ARGUMENT;
    return verify_host(c, sigs, msgs, msg_len, moff, nullptr, MBLS_PK_UNCOMPRESSED, t, idx, off, n, k, MBLS_MODE_FAST_AGGREGATE, results, status);
}

// ---- batch helpers
extern "C" int mbls_pk_decode_batch(mbls_ctx* c, const uint8_t* in, int fmt, int validate, uint64_t n, uint8_t* out96, uint8_t* errs) {
    if (!c || !in || !out96 || !errs || (fmt != 0 && fmt != 1)) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    sbuf di(c, 0), dout(c, 1), de(c, 2); HIPCHK(c, di.up(in, (fmt ? 96 : 48) * n)); HIPCHK(c, dout.alloc(96 * n)); HIPCHK(c, de.alloc(n));
    hipLaunchKernelGGL(k_g1_decode, dim3(nblk(n)), dim3(WG), 0, c->hs_a, di.as<uint8_t>(), fmt, validate, n, dout.as<uint8_t>(), de.as<uint8_t>());
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, dout.down(out96, 96 * n)); HIPCHK(c, de.down(errs, n));
    for (uint64_t i = 0; i < n; i++) errs[i] = (uint8_t)map_dec_err_g1(errs[i]);
    return MBLS_OK;
}
extern "C" int mbls_pk_compress_batch(mbls_ctx* c, const uint8_t* in96, uint64_t n, uint8_t* out48, uint8_t* errs) {
    if (!c || !in96 || !out48 || !errs) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    sbuf di(c, 0), dout(c, 1), de(c, 2); HIPCHK(c, di.up(in96, 96 * n)); HIPCHK(c, dout.alloc(48 * n)); HIPCHK(c, de.alloc(n));
    hipLaunchKernelGGL(k_g1_compress, dim3(nblk(n)), dim3(WG), 0, c->hs_a, di.as<uint8_t>(), n, dout.as<uint8_t>(), de.as<uint8_t>());
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, dout.down(out48, 48 * n)); HIPCHK(c, de.down(errs, n));
    for (uint64_t i = 0; i < n; i++) errs[i] = (uint8_t)map_dec_err_g1(errs[i]);
    return MBLS_OK;
}
extern "C" int mbls_sig_check_batch(mbls_ctx* c, const uint8_t* in96, uint64_t n, uint8_t* errs, uint8_t* in_g2) {
    if (!c || !in96 || !errs) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    sbuf di(c, 0), de(c, 1), dg(c, 2); HIPCHK(c, di.up(in96, 96 * n)); HIPCHK(c, de.alloc(n)); if (in_g2) HIPCHK(c, dg.alloc(n));
    int rc = mbls_ctx_reserve(c, n); if (rc) return rc;               // the subgroup routine works on the items' workspace slots
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    rc = ws_acquire(c, c->hs_a); if (rc) return rc;
    hipLaunchKernelGGL(k_g2_check, dim3(nblk(n)), dim3(WG), 0, c->hs_a, ws, di.as<uint8_t>(), n, de.as<uint8_t>(), in_g2 ? dg.as<uint8_t>() : (uint8_t*)nullptr);
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); c->ws_pending = false; HIPCHK(c, de.down(errs, n)); if (in_g2) HIPCHK(c, dg.down(in_g2, n));
    for (uint64_t i = 0; i < n; i++) errs[i] = (uint8_t)map_dec_err_g2(errs[i]);
    return MBLS_OK;
}
static void g2_tree_levels(mbls_ctx* c, mbls_ws ws, uint64_t m, int levels, hipStream_t s);
// Secret-dependent words do not outlive the call that made them: workspace slots [slot_lo, slot_hi) of items 0..m-1 are zeroed on the
// stream (word-major layout: one row of `stride` items per word, so this is a 2-D fill of m items x 12 (slot_hi - slot_lo) rows).
// NOTE the signing / key-derivation kernels are NOT constant-time: k_aggregate_indexed_d gathers table records at secret-dependent
// addresses and the window digits of k_sign_blind select per-lane records (include/mbls.h says so at mbls_sign_batch).
static hipError_t ws_wipe(mbls_ctx* c, int slot_lo, int slot_hi, uint64_t m, hipStream_t s) {
    return hipMemset2DAsync(c->d_w + (size_t)slot_lo * 12 * c->cap, c->cap * 4, 0, m * 4, (size_t)(slot_hi - slot_lo) * 12, s);
}
// [sk] H(msg) for n (secret key, message) pairs: the pipeline's message phase, the four-lane windowed multiplication (k_sign_blind), two
// levels of the G2 sum tree, compression -- in chunks of MBLS_SIGN_CHUNK signatures (four workspace items each). Enqueues only.
#define MBLS_SIGN_CHUNK 65536ull
extern "C" int mbls_sign_batch_device(mbls_ctx* c, const uint8_t* d_sks, const uint8_t* d_msgs, uint32_t msg_len, uint64_t n, uint8_t* d_sigs, void* stream) {
    if (!c || !d_sks || !d_sigs || (!d_msgs && msg_len)) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const uint64_t chunk = n < MBLS_SIGN_CHUNK ? n : MBLS_SIGN_CHUNK;
    int rc = mbls_ctx_reserve(c, 4 * chunk); if (rc) return rc;
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    rc = ws_acquire(c, s); if (rc) return rc;
    for (uint64_t lo = 0; lo < n; lo += chunk) {
        const uint64_t m = n - lo < chunk ? n - lo : chunk;
        HIPCHK(c, hipMemsetAsync(c->d_status, 0, 4 * m, s));
        launch_hash(c, ws, d_msgs + (uint64_t)msg_len * lo, msg_len, nullptr, c->d_status, m, s);
        hipLaunchKernelGGL(k_sign_blind, dim3(nblk(4 * m)), dim3(WG), 0, s, ws, d_sks + 32 * lo, m, c->secret_ops_fast ? 0 : 1);
        g2_tree_levels(c, ws, 4 * m, 2, s);
        hipLaunchKernelGGL(k_s_export, dim3(nblk(m)), dim3(WG), 0, s, ws, m, d_sigs + 96 * lo);
    }
    HIPCHK(c, hipGetLastError());
    // the four partial products [a_j] psi^j(H) (slots SIG, S) and their window tables (KREC..) are functions of the secret key
    HIPCHK(c, ws_wipe(c, MBLS_SLOT_SIG, MBLS_SLOT_SIG + 4, 4 * chunk, s));
    HIPCHK(c, ws_wipe(c, MBLS_SLOT_S, MBLS_SLOT_S + 6, 4 * chunk, s));
    HIPCHK(c, ws_wipe(c, MBLS_SLOT_KREC, MBLS_SLOT_KREC + 48 + 6, 4 * chunk, s));        // the tables and (constant-time form) the record each window selected: 49..102
    return ws_release(c, s);
}
extern "C" int mbls_sign_batch(mbls_ctx* c, const uint8_t* sks, const uint8_t* msgs, uint32_t msg_len, uint64_t n, uint8_t* sigs) {
    if (!c || !sks || !sigs || (!msgs && msg_len)) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    sbuf dk(c, 0), dm(c, 1), dout(c, 2); HIPCHK(c, dk.up(sks, 32 * n)); HIPCHK(c, dm.up(msgs, (size_t)msg_len * n)); HIPCHK(c, dout.alloc(96 * n));
    int rc = mbls_sign_batch_device(c, dk.as<uint8_t>(), dm.as<uint8_t>(), msg_len, n, dout.as<uint8_t>(), c->hs_a); if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(dk.p, 0, 32 * n, c->hs_a));                  // the staged secret keys
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, dout.down(sigs, 96 * n)); return MBLS_OK;
}
// the fixed table of sk -> pk: the 1 024 multiples come from the compiled ladder (k_sk_to_pk) once per context
static int ensure_gtab(mbls_ctx* c) {
    if (c->d_gtab) return MBLS_OK;
    std::vector<uint8_t> sc(32 * 1024, 0);
    for (int j = 0; j < 64; j++) for (int d = 0; d < 16; d++) sc[32 * (16 * j + d) + 31 - (j >> 1)] = (uint8_t)(d << (4 * (j & 1)));
    HIPCHK(c, hipDeviceSynchronize());       // once per context: nothing in flight may still be reading the staging buffers used here
    sbuf dk(c, 7), dp(c, 8), de(c, 9);      // not the staging slots of the host entries that call this with their inputs already uploaded
    HIPCHK(c, dk.up(sc.data(), sc.size())); HIPCHK(c, dp.alloc(96 * 1024)); HIPCHK(c, de.alloc(1024));
    uint32_t* tab = nullptr; HIPCHK(c, hipMalloc(&tab, 1024 * MBLS_KEYREC_DWORDS * 4));
    hipLaunchKernelGGL(k_sk_to_pk, dim3(nblk(1024)), dim3(WG), 0, c->hs_a, dk.as<uint8_t>(), MBLS_PK_UNCOMPRESSED, (uint64_t)1024, dp.as<uint8_t>());
    hipLaunchKernelGGL(k_keytable_append, dim3(nblk(1024)), dim3(WG), 0, c->hs_a, dp.as<uint8_t>(), MBLS_PK_UNCOMPRESSED, 0, (uint64_t)1024, tab, de.as<uint8_t>());
    hipError_t e = hipStreamSynchronize(c->hs_a);
    std::vector<uint8_t> errs(1024, 1);
    if (e == hipSuccess) e = de.down(errs.data(), 1024);
    bool ok = e == hipSuccess;
    for (int t = 0; ok && t < 1024; t++) ok = errs[t] == 0;
    if (!ok) { (void)hipFree(tab); snprintf(c->err, sizeof(c->err), "building the generator table failed"); return MBLS_ERR_DEVICE; }
    c->d_gtab = tab; return MBLS_OK;
}
#define MBLS_SKPK_CHUNK 131072ull
#define MBLS_SKPK_CT_CHUNK 32768ull          /* constant-time form: 64 selected records of 128 bytes per key are staged (256 MB per chunk) */
extern "C" int mbls_sk_to_pk_batch_device(mbls_ctx* c, const uint8_t* d_sks, int fmt, uint64_t n, uint8_t* d_pks, void* stream) {
    if (!c || !d_sks || !d_pks || (fmt != 0 && fmt != 1)) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_gtab(c); if (rc) return rc;
    const bool ct = !c->secret_ops_fast;
    const uint64_t cmax = ct ? MBLS_SKPK_CT_CHUNK : MBLS_SKPK_CHUNK;
    const uint64_t chunk = n < cmax ? n : cmax;
    rc = mbls_ctx_reserve(c, chunk); if (rc) return rc;
    if (c->skidx_cap < chunk) {
        if (c->d_skidx) { (void)hipFree(c->d_skidx); c->d_skidx = nullptr; c->skidx_cap = 0; }
        HIPCHK(c, hipMalloc(&c->d_skidx, chunk * 64 * 4)); c->skidx_cap = chunk;
    }
    if (ct && c->sksel_cap < chunk) {
        if (c->d_sksel) { (void)hipFree(c->d_sksel); c->d_sksel = nullptr; c->sksel_cap = 0; }
        HIPCHK(c, hipMalloc(&c->d_sksel, chunk * 64 * MBLS_KEYREC_DWORDS * 4)); c->sksel_cap = chunk;
    }
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    rc = ws_acquire(c, s); if (rc) return rc;
    for (uint64_t lo = 0; lo < n; lo += chunk) {
        const uint64_t m = n - lo < chunk ? n - lo : chunk;
        HIPCHK(c, hipMemsetAsync(c->d_status, 0, 4 * m, s));
        if (ct) {
            // the records are chosen by scan + selection (k_sk_select), the key sum then walks sel[64 i + j]: public indices, public addresses
            hipLaunchKernelGGL(k_sk_select, dim3(nblk(64 * m)), dim3(WG), 0, s, d_sks + 32 * lo, m, (const uint32_t*)c->d_gtab, c->d_sksel, c->d_skidx);
            hipLaunchKernelGGL(k_aggregate_indexed_d, dim3(nblk(m)), dim3(WG), 0, s, ws, (const uint32_t*)c->d_sksel, (uint64_t)(64 * m), (const uint32_t*)c->d_skidx,
                               (const uint32_t*)nullptr, 64u, MBLS_MODE_VERIFY, c->d_status, m);
        } else {
            hipLaunchKernelGGL(k_sk_digits, dim3(nblk(m)), dim3(WG), 0, s, d_sks + 32 * lo, m, c->d_skidx);
            hipLaunchKernelGGL(k_aggregate_indexed_d, dim3(nblk(m)), dim3(WG), 0, s, ws, (const uint32_t*)c->d_gtab, (uint64_t)1024, (const uint32_t*)c->d_skidx,
                               (const uint32_t*)nullptr, 64u, MBLS_MODE_VERIFY, c->d_status, m);
        }
        hipLaunchKernelGGL(k_apk_export_fmt, dim3(nblk(m)), dim3(WG), 0, s, ws, m, fmt, d_pks + (uint64_t)(fmt ? 96 : 48) * lo);
    }
    HIPCHK(c, hipGetLastError());
    if (ct) HIPCHK(c, hipMemsetAsync(c->d_sksel, 0, chunk * 64 * MBLS_KEYREC_DWORDS * 4, s));            // the selected multiples are functions of the keys
    else HIPCHK(c, hipMemsetAsync(c->d_skidx, 0, chunk * 64 * 4, s));            // the hexadecimal digits of the secret keys
    return ws_release(c, s);
}
// 1: table lookups that depend on a secret key (the window tables of signing, the generator table of sk -> pk) go back to the faster forms that read ONE record
// at an address computed from the key's digits -- for building test and bench inputs from throw-away keys only. 0 (default): scan + selection.
extern "C" int mbls_ctx_set_secret_ops(mbls_ctx* c, int variable_time) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu); c->secret_ops_fast = variable_time != 0; return MBLS_OK;
}
extern "C" int mbls_sk_to_pk_batch(mbls_ctx* c, const uint8_t* sks, int fmt, uint64_t n, uint8_t* pks) {
    if (!c || !sks || !pks || (fmt != 0 && fmt != 1)) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    sbuf dk(c, 0), dout(c, 1); HIPCHK(c, dk.up(sks, 32 * n)); HIPCHK(c, dout.alloc((fmt ? 96 : 48) * n));
    int rc = mbls_sk_to_pk_batch_device(c, dk.as<uint8_t>(), fmt, n, dout.as<uint8_t>(), c->hs_a); if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(dk.p, 0, 32 * n, c->hs_a));                  // the staged secret keys
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, dout.down(pks, (fmt ? 96 : 48) * n)); return MBLS_OK;
}
extern "C" int mbls_hash_to_g2_batch_mode(mbls_ctx* c, const uint8_t* msgs, uint32_t msg_len, uint64_t n, uint8_t* out96, int mode);
extern "C" int mbls_hash_to_g2_batch(mbls_ctx* c, const uint8_t* msgs, uint32_t msg_len, uint64_t n, uint8_t* out96) {
    return mbls_hash_to_g2_batch_mode(c, msgs, msg_len, n, out96, 0);
}
extern "C" int mbls_hash_to_g2_batch_mode(mbls_ctx* c, const uint8_t* msgs, uint32_t msg_len, uint64_t n, uint8_t* out96, int mode) {
    if (!c || !out96 || (!msgs && msg_len) || mode < 0 || mode > 3) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    sbuf dm(c, 0), dout(c, 1); HIPCHK(c, dm.up(msgs, (size_t)msg_len * n)); HIPCHK(c, dout.alloc(96 * n));
    {                     // the pipeline's message phase (mode 0: the form a batch of this size takes by itself; mode 1: one lane per item, the generated routine;
                          // mode 2: one wave per item, program hashg2; mode 3: two lanes per item), then its H
        const bool pair0 = mode == 0 && 2 * n <= c->round_items;
        int rc = mbls_ctx_reserve(c, (mode == 3 || pair0) ? 2 * n : n); if (rc) return rc;
        mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
        rc = ws_acquire(c, c->hs_a); if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(c->d_status, 0, 4 * n, c->hs_a));
        // the form is named, not routed: mode 1 one lane per item, mode 2 one wave per item (four items per wave above the packing limit), mode 3 two lanes per item
        launch_hash(c, ws, dm.as<uint8_t>(), msg_len, nullptr, c->d_status, n, c->hs_a, mode == 3 || pair0,
                    mode == 0 ? HASH_FORM_AUTO : mode == 1 ? HASH_FORM_LANE : mode == 2 ? HASH_FORM_WAVE : HASH_FORM_PAIR);
        hipLaunchKernelGGL(k_h_export, dim3(nblk(n)), dim3(WG), 0, c->hs_a, ws, n, dout.as<uint8_t>());
    }
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); c->ws_pending = false; HIPCHK(c, dout.down(out96, 96 * n)); return MBLS_OK;
}
extern "C" int mbls_aggregate_public_keys_batch(mbls_ctx* c, const uint8_t* pks, int fmt, const uint32_t* off, uint64_t n, uint32_t k,
                                                uint8_t* apks96, uint32_t* status) {
    if (!c || !apks96 || (fmt != 0 && fmt != 1)) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    if (off && !offsets_ok(off, n)) ARGFAIL(c, "offsets must be non-decreasing");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = mbls_ctx_reserve(c, n); if (rc) return rc;
    uint64_t total = off ? off[n] : (uint64_t)k * n;
    if (total && !pks) ARGFAIL(c, "null key buffer");
    sbuf dp(c, 0), doff(c, 1), dout(c, 2); HIPCHK(c, dp.up(pks, (fmt ? 96 : 48) * total)); if (off) HIPCHK(c, doff.up(off, 4 * (n + 1))); HIPCHK(c, dout.alloc(96 * n));
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    hipStream_t s = c->hs_a;
    rc = ws_acquire(c, s); if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_status, 0, 4 * n, s));      // k_aggregate ORs its bits in
    launch_aggregate(ws, dp.as<uint8_t>(), off ? doff.as<uint32_t>() : (const uint32_t*)nullptr, k, fmt, MBLS_MODE_FAST_AGGREGATE, c->d_status, n, s);
    hipLaunchKernelGGL(k_apk_export, dim3(nblk(n)), dim3(WG), 0, s, ws, n, dout.as<uint8_t>());
    HIPCHK(c, hipStreamSynchronize(s)); c->ws_pending = false;
    HIPCHK(c, dout.down(apks96, 96 * n));
    if (status) HIPCHK(c, hipMemcpy(status, c->d_status, 4 * n, hipMemcpyDeviceToHost));
    return MBLS_OK;
}
extern "C" int mbls_aggregate_signatures_batch_device(mbls_ctx* c, const uint8_t* d_sigs, const uint32_t* d_off, uint64_t n, uint32_t k, uint64_t total,
                                                      uint8_t* d_out96, uint8_t* d_errs, void* stream) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    if (!d_out96 || !d_errs || (total && !d_sigs)) ARGFAIL(c, "null buffer");
    if (!d_off && total != (uint64_t)k * n) ARGFAIL(c, "total_sigs != n_sets * k");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    // the decoded points pass through two of the context's staging buffers: ordered against calls on other streams like the workspace
    // (a buffer that has to grow is freed, which drains the device first)
    int rc = ws_acquire(c, s); if (rc) return rc;
    sbuf dxy(c, 7), dfl(c, 8); HIPCHK(c, dxy.alloc(192 * total)); HIPCHK(c, dfl.alloc(total));
    if (total) hipLaunchKernelGGL(k_g2_decode_affine, dim3(nblk(total)), dim3(WG), 0, s, d_sigs, total, dxy.as<uint32_t>(), dfl.as<uint8_t>());
    hipLaunchKernelGGL(k_g2_sum, dim3(nblk(n)), dim3(WG), 0, s, (const uint32_t*)dxy.as<uint32_t>(), (const uint8_t*)dfl.as<uint8_t>(), d_off, k, n, d_out96, d_errs);
    HIPCHK(c, hipGetLastError());
    return ws_release(c, s);
}
extern "C" int mbls_aggregate_signatures_batch(mbls_ctx* c, const uint8_t* sigs, const uint32_t* off, uint64_t n, uint32_t k, uint8_t* out96, uint8_t* errs) {
    if (!c || !out96 || !errs) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    if (off && !offsets_ok(off, n)) ARGFAIL(c, "offsets must be non-decreasing");
    uint64_t total = off ? off[n] : (uint64_t)k * n;
    if (total && !sigs) ARGFAIL(c, "null signature buffer");
    HIPCHK(c, hipSetDevice(c->device));
    sbuf di(c, 0), doff(c, 1), dout(c, 2), de(c, 3);
    HIPCHK(c, di.up(sigs, 96 * total)); if (off) HIPCHK(c, doff.up(off, 4 * (n + 1))); HIPCHK(c, dout.alloc(96 * n)); HIPCHK(c, de.alloc(n));
    int rc = mbls_aggregate_signatures_batch_device(c, di.as<uint8_t>(), off ? doff.as<uint32_t>() : nullptr, n, k, total, dout.as<uint8_t>(), de.as<uint8_t>(), c->hs_a);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, dout.down(out96, 96 * n)); HIPCHK(c, de.down(errs, n));
    for (uint64_t i = 0; i < n; i++) errs[i] = (uint8_t)map_dec_err_g2(errs[i]);
    return MBLS_OK;
}
extern "C" int mbls_fp_mul_batch(mbls_ctx* c, const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out, int square) {
    if (!c || !a || !b || !out) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (!n) return MBLS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    sbuf da(c, 0), db(c, 1), dout(c, 2); HIPCHK(c, da.up(a, 48 * n)); HIPCHK(c, db.up(b, 48 * n)); HIPCHK(c, dout.alloc(48 * n));
    hipLaunchKernelGGL(k_fp_mul, dim3(nblk(n)), dim3(WG), 0, c->hs_a, da.as<uint8_t>(), db.as<uint8_t>(), n, dout.as<uint8_t>(), square);
    HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, dout.down(out, 48 * n)); return MBLS_OK;
}
extern "C" int mbls_fp_mul_bench(mbls_ctx* c, uint64_t n_lanes, uint32_t iters, float* ms_out) {
    if (!c || !ms_out || !n_lanes) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    sbuf sink(c, 0); HIPCHK(c, sink.alloc(4 * n_lanes));
    hipLaunchKernelGGL(k_fp_mul_bench, dim3(nblk(n_lanes)), dim3(WG), 0, c->hs_a, sink.as<uint32_t>(), 16u, n_lanes);   // warm-up
    HIPCHK(c, hipEventRecord(c->ev[0], c->hs_a));
    hipLaunchKernelGGL(k_fp_mul_bench, dim3(nblk(n_lanes)), dim3(WG), 0, c->hs_a, sink.as<uint32_t>(), iters, n_lanes);
    HIPCHK(c, hipEventRecord(c->ev[1], c->hs_a)); HIPCHK(c, hipEventSynchronize(c->ev[1]));
    HIPCHK(c, hipEventElapsedTime(ms_out, c->ev[0], c->ev[1]));
    return MBLS_OK;
}

extern "C" int mbls_valu_bench(mbls_ctx* c, int mode, uint32_t waves_per_simd, uint32_t iters, float* ms_out) {
    if (!c || !ms_out || !waves_per_simd || waves_per_simd > 8 || mode < 0 || mode > 7) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipDeviceProp_t prop; HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    unsigned blocks = (unsigned)prop.multiProcessorCount * 4u * waves_per_simd;      // waves_per_simd waves on every SIMD
    sbuf sink(c, 0); HIPCHK(c, sink.alloc((size_t)4 * WG * blocks));
    hipLaunchKernelGGL(k_valu_bench, dim3(blocks), dim3(WG), 0, c->hs_a, sink.as<uint32_t>(), 64u, mode);     // warm-up
    HIPCHK(c, hipEventRecord(c->ev[0], c->hs_a));
    hipLaunchKernelGGL(k_valu_bench, dim3(blocks), dim3(WG), 0, c->hs_a, sink.as<uint32_t>(), iters, mode);
    HIPCHK(c, hipEventRecord(c->ev[1], c->hs_a)); HIPCHK(c, hipEventSynchronize(c->ev[1]));
    HIPCHK(c, hipEventElapsedTime(ms_out, c->ev[0], c->ev[1]));
    return MBLS_OK;
}

// ---- scalar API (n = 1 batches)
static int sk_check(const uint8_t* sk, size_t len) {     // SecretKey::from_bytes, reference src/keys.rs:80-82 and tests :285-297
    static const uint8_t R_BE[32] = {0x73,0xed,0xa7,0x53,0x29,0x9d,0x7d,0x48,0x33,0x39,0xd8,0x08,0x09,0xa1,0xd8,0x05,0x53,0xbd,0xa4,0x02,0xff,0xfe,0x5b,0xfe,0xff,0xff,0xff,0xff,0x00,0x00,0x00,0x01};
    if (!sk || len != 32) return MBLS_ERR_INVALID_SECRET_KEY_SIZE;
    bool zero = true; for (int i = 0; i < 32; i++) if (sk[i]) zero = false;
    if (zero || memcmp(sk, R_BE, 32) >= 0) return MBLS_ERR_INVALID_SECRET_KEY_RANGE;
    return MBLS_OK;
}
extern "C" int mbls_pk_from_bytes(mbls_ctx* c, const uint8_t* bytes, size_t len, uint8_t pk_out[96]) {
    if (!c || !pk_out) return MBLS_ERR_ARGUMENT;
    if (!bytes || len != 48) return MBLS_ERR_INVALID_G1_SIZE;         // decompress_g1, reference src/amcl_utils.rs:54-56
    uint8_t e; int rc = mbls_pk_decode_batch(c, bytes, MBLS_PK_COMPRESSED, 1, 1, pk_out, &e); return rc ? rc : e;
}
extern "C" int mbls_pk_from_bytes_unchecked(mbls_ctx* c, const uint8_t* bytes, size_t len, uint8_t pk_out[96]) {
    if (!c || !pk_out) return MBLS_ERR_ARGUMENT;
    if (!bytes || len != 48) return MBLS_ERR_INVALID_G1_SIZE;
    uint8_t e; int rc = mbls_pk_decode_batch(c, bytes, MBLS_PK_COMPRESSED, 0, 1, pk_out, &e); return rc ? rc : e;
}
extern "C" int mbls_pk_from_uncompressed_bytes(mbls_ctx* c, const uint8_t* bytes, size_t len, uint8_t pk_out[96]) {
    if (!c || !pk_out) return MBLS_ERR_ARGUMENT;
    if (!bytes || len != 96) return MBLS_ERR_INVALID_G1_SIZE;         // reference src/keys.rs:171-173
    uint8_t e; int rc = mbls_pk_decode_batch(c, bytes, MBLS_PK_UNCOMPRESSED, 0, 1, pk_out, &e); return rc ? rc : e;
}
extern "C" int mbls_pk_as_bytes(mbls_ctx* c, const uint8_t pk[96], uint8_t out[48]) {
    if (!c || !pk || !out) return MBLS_ERR_ARGUMENT;
    uint8_t e; int rc = mbls_pk_compress_batch(c, pk, 1, out, &e); return rc ? rc : e;
}
extern "C" int mbls_pk_key_validate(mbls_ctx* c, const uint8_t pk[96]) {
    if (!c || !pk) return 0;
    mbls_lock lk(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return 0;
    sbuf di(c, 0), dk(c, 1); if (di.up(pk, 96) != hipSuccess || dk.alloc(1) != hipSuccess) return 0;
    hipLaunchKernelGGL(k_g1_key_validate, dim3(1), dim3(WG), 0, c->hs_a, di.as<uint8_t>(), (uint64_t)1, dk.as<uint8_t>());
    uint8_t ok = 0; if (hipStreamSynchronize(c->hs_a) != hipSuccess || dk.down(&ok, 1) != hipSuccess) return 0;
    return ok;
}
extern "C" int mbls_pk_from_secret_key(mbls_ctx* c, const uint8_t* sk, size_t sk_len, uint8_t pk_out[96]) {
    if (!c || !pk_out) return MBLS_ERR_ARGUMENT;
    int e = sk_check(sk, sk_len); if (e) return e;
    return mbls_sk_to_pk_batch(c, sk, MBLS_PK_UNCOMPRESSED, 1, pk_out);
}
extern "C" int mbls_sig_from_bytes(mbls_ctx* c, const uint8_t* bytes, size_t len, uint8_t sig_out[96]) {
    if (!c || !sig_out) return MBLS_ERR_ARGUMENT;
    if (!bytes || len != 96) return MBLS_ERR_INVALID_G2_SIZE;         // decompress_g2, reference src/amcl_utils.rs:70-72
    uint8_t e; int rc = mbls_sig_check_batch(c, bytes, 1, &e, nullptr); if (rc) return rc;
    if (e) return e;
    memcpy(sig_out, bytes, 96); return MBLS_OK;
}
extern "C" int mbls_sign(mbls_ctx* c, const uint8_t* msg, size_t msg_len, const uint8_t* sk, size_t sk_len, uint8_t sig_out[96]) {
    if (!c || !sig_out) return MBLS_ERR_ARGUMENT;
    if (msg_len > 0xFFFFFFFFull) return MBLS_ERR_ARGUMENT;
    int e = sk_check(sk, sk_len); if (e) return e;
    return mbls_sign_batch(c, sk, msg, (uint32_t)msg_len, 1, sig_out);
}
extern "C" int mbls_verify(mbls_ctx* c, const uint8_t sig[96], const uint8_t* msg, size_t msg_len, const uint8_t pk[96]) {
    uint8_t r = 0; if (!c || !sig || !pk || msg_len > 0xFFFFFFFFull) return 0;
    if (mbls_verify_batch(c, sig, msg, (uint32_t)msg_len, nullptr, pk, MBLS_PK_UNCOMPRESSED, 1, &r, nullptr)) return 0;
    return r;
}
extern "C" int mbls_aggregate_public_keys(mbls_ctx* c, const uint8_t* pks96, size_t n, uint8_t apk_out[96]) {
    if (!c || !apk_out) return MBLS_ERR_ARGUMENT;
    if (n == 0) return MBLS_ERR_AGGREGATE_EMPTY_POINTS;               // reference src/aggregates.rs:30-32
    if (n > 0xFFFFFFFFull) return MBLS_ERR_ARGUMENT;
    uint32_t st = 0; int rc = mbls_aggregate_public_keys_batch(c, pks96, MBLS_PK_UNCOMPRESSED, nullptr, 1, (uint32_t)n, apk_out, &st);
    if (rc) return rc;
    return (st & MBLS_ST_BAD_PK_ENCODING) ? MBLS_ERR_INVALID_POINT : MBLS_OK;
}
extern "C" int mbls_aggregate_public_key_add(mbls_ctx* c, const uint8_t a[96], const uint8_t b[96], uint8_t out[96]) {
    if (!c || !a || !b || !out) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    sbuf da(c, 0), db(c, 1), dout(c, 2), de(c, 3); HIPCHK(c, da.up(a, 96)); HIPCHK(c, db.up(b, 96)); HIPCHK(c, dout.alloc(96)); HIPCHK(c, de.alloc(1));
    hipLaunchKernelGGL(k_g1_add, dim3(1), dim3(WG), 0, c->hs_a, da.as<uint8_t>(), db.as<uint8_t>(), (uint64_t)1, dout.as<uint8_t>(), de.as<uint8_t>());
    uint8_t e; HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, dout.down(out, 96)); HIPCHK(c, de.down(&e, 1));
    return map_dec_err_g1(e);
}
extern "C" int mbls_aggregate_signature_add(mbls_ctx* c, const uint8_t a[96], const uint8_t b[96], uint8_t out[96]) {
    if (!c || !a || !b || !out) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    sbuf da(c, 0), db(c, 1), dout(c, 2), de(c, 3); HIPCHK(c, da.up(a, 96)); HIPCHK(c, db.up(b, 96)); HIPCHK(c, dout.alloc(96)); HIPCHK(c, de.alloc(1));
    hipLaunchKernelGGL(k_g2_add, dim3(1), dim3(WG), 0, c->hs_a, da.as<uint8_t>(), db.as<uint8_t>(), (uint64_t)1, dout.as<uint8_t>(), de.as<uint8_t>());
    uint8_t e; HIPCHK(c, hipStreamSynchronize(c->hs_a)); HIPCHK(c, dout.down(out, 96)); HIPCHK(c, de.down(&e, 1));
    return map_dec_err_g2(e);
}
extern "C" int mbls_fast_aggregate_verify(mbls_ctx* c, const uint8_t sig[96], const uint8_t* msg, size_t msg_len, const uint8_t* pks96, size_t n_pks) {
    uint8_t r = 0; if (!c || !sig || msg_len > 0xFFFFFFFFull || n_pks > 0xFFFFFFFFull) return 0;
    if (n_pks == 0) return 0;                                         // reference src/aggregates.rs:179-181
    if (mbls_fast_aggregate_verify_batch(c, sig, msg, (uint32_t)msg_len, nullptr, pks96, MBLS_PK_UNCOMPRESSED, nullptr, 1, (uint32_t)n_pks, &r, nullptr)) return 0;
    return r;
}
extern "C" int mbls_fast_aggregate_verify_pre_aggregated(mbls_ctx* c, const uint8_t sig[96], const uint8_t* msg, size_t msg_len, const uint8_t apk[96]) {
    uint8_t r = 0; if (!c || !sig || !apk || msg_len > 0xFFFFFFFFull) return 0;
    // identical checks with a one-key set: sig in G2, key != infinity, pairing (reference src/aggregates.rs:223-253)
    if (mbls_fast_aggregate_verify_batch(c, sig, msg, (uint32_t)msg_len, nullptr, apk, MBLS_PK_UNCOMPRESSED, nullptr, 1, 1, &r, nullptr)) return 0;
    return r;
}

// Product / sum trees of the n-pairing paths: m values in slot F (slot S) of items 0..m-1 -> item 0. A level with many pairs is one lane
// per product (k_f12_tree_d / k_g2_tree_d: the chip is full of them); below MBLS_COOP_TREE_PAIRS pairs a level is one WAVE per product
// (mbls_coop.h, programs f12mul / g2add: 14 / 28 steps instead of a 25 k-instruction dependent chain).
#define MBLS_COOP_TREE_PAIRS 2048
static void f12_tree(mbls_ctx* c, mbls_ws ws, uint64_t m, hipStream_t s) {
    while (m > 1) {
        const uint64_t half = (m + 1) / 2, pairs = m - half;
        if (pairs > MBLS_COOP_TREE_PAIRS) hipLaunchKernelGGL(k_f12_tree_d, dim3(nblk(half)), dim3(WG), 0, s, ws, m, half);
        else coop_run(c, COOP_F12MUL, ws, (uint64_t)0, (uint64_t)1, half, pairs, (uint32_t*)nullptr, (uint8_t*)nullptr, COOP_RES_ITEM, s);
        m = half;
    }
}
static void g2_tree_levels(mbls_ctx* c, mbls_ws ws, uint64_t m, int levels, hipStream_t s) {
    while (m > 1 && levels-- > 0) {
        const uint64_t half = (m + 1) / 2, pairs = m - half;
        if (pairs > MBLS_COOP_TREE_PAIRS) hipLaunchKernelGGL(k_g2_tree_d, dim3(nblk(half)), dim3(WG), 0, s, ws, m, half);
        else coop_run(c, COOP_G2ADD, ws, (uint64_t)0, (uint64_t)1, half, pairs, (uint32_t*)nullptr, (uint8_t*)nullptr, COOP_RES_ITEM, s);
        m = half;
    }
}
static void g2_tree(mbls_ctx* c, mbls_ws ws, uint64_t m, hipStream_t s) { g2_tree_levels(c, ws, m, 64, s); }
// n-pairing product check shared by aggregate_verify and verify_multiple (reference src/aggregates.rs:158-169, :307-315). On entry the
// workspace holds, for items 0..n-1, H_i (slot H) and P_i (slot APK); S (the (S, -G1) pair's G2 point) in slot S of item 0; the OR of
// every status word in d_scalar[0]. Everything is enqueued on s: one Miller loop per lane, the product tree, and ONE wave for the tail --
// the Miller loop of (S, -G1), the product, the single final exponentiation, the comparison and the status bits (program vmtail).
// late_status: fold the sets' status words into d_scalar[0] only HERE, after the wait for the signature chain -- the sets' Miller loops then start as
// soon as the keys and the messages are ready and do not wait for the (longer) signature chain, whose bits only the tail needs
// d_partial: stop before the tail and write the shard's record instead (k_vm_export); s_miller_ev then only says "the signature chain is done"
// side_s_chain (batches that fill the chip and run their chains one after the other): the signatures' sum tree and the Miller loop of (S, -G1) are
// enqueued HERE on a second stream, beside the product tree -- both trees are short kernels on a shrinking number of lanes, and the one-wave
// Miller loop hides behind them
// m one-pair Miller loops over items [0, m) of ws (k_miller_single's layout), one lane per pair; half a round or less: two lanes per pair, products in
// pairs (4.9 ms instead of 6.7). Above a round the whole rounds go first and what is left of the last one follows as a launch of its own in the form
// its size allows (81 920 pairs: 6.7 + 4.9 ms instead of two rounds)
static void launch_miller_single(mbls_ctx* c, mbls_ws ws, uint64_t m, hipStream_t s) {
    const uint64_t R = c->round_items;
    const uint64_t full = m > R ? (m / R) * R : 0, rest = m - full;
    if (full) hipLaunchKernelGGL(k_miller_single, dim3(nblk(full)), dim3(WG), 0, s, ws, full, 0, (uint64_t)0);
    if (!rest) return;
    mbls_ws wr = ws; wr.w += full;
    if (2 * rest <= R) hipLaunchKernelGGL(k_miller_single2, dim3(nblk(2 * rest)), dim3(WG), 0, s, wr, rest, 0, (uint64_t)0);
    else hipLaunchKernelGGL(k_miller_single, dim3(nblk(rest)), dim3(WG), 0, s, wr, rest, 0, (uint64_t)0);
}
static int npairing_finish(mbls_ctx* c, uint64_t n, hipStream_t s, uint8_t* d_result, hipEvent_t s_miller_ev = nullptr, bool late_status = false,
                           uint32_t* d_partial = nullptr, bool side_s_chain = false) {
    const bool s_miller_done = s_miller_ev != nullptr || side_s_chain;
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    if (2 * n <= c->coop_max_items)     // few pairs: one WAVE per Miller loop (program miller1, ~0.9 ms) instead of one lane (6.6 ms)
        coop_run(c, COOP_MILLER1, ws, (uint64_t)0, (uint64_t)1, (uint64_t)0, n, (uint32_t*)nullptr, (uint8_t*)nullptr, COOP_RES_ITEM, s);
    else
        launch_miller_single(c, ws, n, s);
    if (side_s_chain) {
        HIPCHK(c, hipEventRecord(c->hs_ev2, s)); HIPCHK(c, hipStreamWaitEvent(c->hs_b, c->hs_ev2, 0));
        g2_tree(c, ws, n, c->hs_b);
        if (!d_partial) coop_run(c, COOP_SMILLER, ws, (uint64_t)0, (uint64_t)1, (uint64_t)0, (uint64_t)1, (uint32_t*)nullptr, (uint8_t*)nullptr, COOP_RES_ITEM, c->hs_b);
        HIPCHK(c, hipEventRecord(c->hs_ev, c->hs_b));
        s_miller_ev = c->hs_ev;
    }
    f12_tree(c, ws, n, s);
    // s_miller_done: the Miller value of (S, -G1) is left in slots 97..108 of item 0 by program smiller, running beside the chains
    if (s_miller_done) HIPCHK(c, hipStreamWaitEvent(s, s_miller_ev, 0));
    if (late_status) hipLaunchKernelGGL(k_status_or, dim3(nblk(n)), dim3(WG), 0, s, c->d_status, n, c->d_scalar);
    if (d_partial) hipLaunchKernelGGL(k_vm_export, dim3(1), dim3(WG), 0, s, ws, (const uint32_t*)c->d_scalar, 0, d_partial);
    else coop_run(c, s_miller_done ? COOP_VMFINAL : COOP_VMTAIL, ws, (uint64_t)0, (uint64_t)1, (uint64_t)0, (uint64_t)1, c->d_scalar, d_result, COOP_RES_BATCH, s);
    HIPCHK(c, hipGetLastError());
    return MBLS_OK;
}
extern "C" int mbls_aggregate_verify(mbls_ctx* c, const uint8_t sig[96], const uint8_t* msgs, const size_t* msg_lens, size_t n_msgs,
                                     const uint8_t* pks96, size_t n_pks) {
    if (!c || !sig) return 0;
    if (n_msgs != n_pks || n_pks == 0) return 0;                      // reference src/aggregates.rs:132-134
    if (!msg_lens || !pks96) return 0;
    mbls_lock lk(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return 0;
    uint64_t n = n_pks;
    if (mbls_ctx_reserve(c, n)) return 0;
    std::vector<uint64_t> off;
    try { off.resize(n + 1); } catch (...) { return 0; }
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        if (msg_lens[i] > 0xFFFFFFFFull) return 0;
        off[i] = total; total += msg_lens[i];
    }
    off[n] = total;
    if (total && !msgs) return 0;
    sbuf dm(c, 0), doff(c, 1), dp(c, 3), dsig(c, 4); int result = 0;
    if (dm.up(msgs, total) != hipSuccess || doff.up(off.data(), 8 * (n + 1)) != hipSuccess ||
        dp.up(pks96, 96 * n) != hipSuccess || dsig.up(sig, 96) != hipSuccess) return 0;
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    hipStream_t s = c->hs_a;
    if (ws_acquire(c, s)) return 0;
    (void)hipMemsetAsync(c->d_scalar, 0, 64, s);
    (void)hipMemsetAsync(c->d_status, 0, 4 * n, s);
    if (hipMemsetAsync(c->d_results, 0, 1, s) != hipSuccess) return 0;      // false until the tail kernel has spoken
    // three chains side by side: the signature (decode + subgroup test: k_sig on item 0's slots, then into slot S), the keys, the messages
    (void)hipEventRecord(c->hs_ev, s); (void)hipStreamWaitEvent(c->hs_b, c->hs_ev, 0); (void)hipStreamWaitEvent(c->hs_c, c->hs_ev, 0);
    hipLaunchKernelGGL(k_sig2, dim3(1), dim3(WG), 0, c->hs_b, ws, (const uint8_t*)dsig.as<uint8_t>(), c->d_status, (uint64_t)1);      // (two lanes: the chain is latency)
    hipLaunchKernelGGL(k_sigslot_to_s, dim3(1), dim3(WG), 0, c->hs_b, ws, (uint64_t)0);
    (void)hipEventRecord(c->hs_ev2, c->hs_b);                                   // the signature's status bits and S exist
    // ... and its Miller loop (S, -G1) runs on one wave right away, beside the message phase (the longest chain) and the pairs' own loops: the tail then only
    // multiplies and exponentiates (program vmfinal instead of vmtail)
    coop_run(c, COOP_SMILLER, ws, (uint64_t)0, (uint64_t)1, (uint64_t)0, (uint64_t)1, (uint32_t*)nullptr, (uint8_t*)nullptr, COOP_RES_ITEM, c->hs_b);
    (void)hipEventRecord(c->hs_ev, c->hs_b);
    hipLaunchKernelGGL(k_blind_g1, dim3(nblk(n)), dim3(WG), 0, s, ws, dp.as<uint8_t>(), (const uint64_t*)nullptr, c->d_status, n);   // r_i = 1: no blinding in AggregateVerify
    launch_hash(c, ws, dm.as<uint8_t>(), 0u, (const uint64_t*)doff.as<uint64_t>(), c->d_status, n, c->hs_c);
    (void)hipEventRecord(c->hs_ev3, c->hs_c);
    (void)hipStreamWaitEvent(s, c->hs_ev2, 0); (void)hipStreamWaitEvent(s, c->hs_ev3, 0);
    hipLaunchKernelGGL(k_status_or, dim3(nblk(n)), dim3(WG), 0, s, c->d_status, n, c->d_scalar);
    // a signature outside G2 or an undecodable member makes the tail answer false (reference src/aggregates.rs:137-139): no host round trip
    if (npairing_finish(c, n, s, c->d_results, c->hs_ev)) return 0;
    uint8_t r = 0;
    if (hipStreamSynchronize(s) != hipSuccess) return 0;
    c->ws_pending = false;
    if (hipMemcpy(&r, c->d_results, 1, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    result = r;
    return result;
}
// n x AggregateSignature::aggregate_verify (reference src/aggregates.rs:130-170): item i = (sig_i, its (message, key) pairs). The pairs of
// all items lie back to back: pair j = (message j of d_msgs / d_moff, key d_pks96 + 96 j); item i owns pairs [d_pair_off[i], d_pair_off[i+1])
// or k each. One lane per pair for the key, the message phase and the one-pair Miller loop; the (sig_i, -G1) pairs ride the same Miller
// launch; a product tree per item; one final exponentiation per item. Enqueues only.
extern "C" int mbls_aggregate_verify_batch_device(mbls_ctx* c, const uint8_t* d_sigs, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff,
        const uint8_t* d_pks96, const uint32_t* d_pair_off, uint32_t k, uint64_t total, uint64_t n, uint8_t* d_results, uint32_t* d_status, void* stream) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (n == 0) return MBLS_OK;
    if (!d_sigs || !d_results || (total && (!d_pks96 || (!d_msgs && msg_len && !d_moff)))) ARGFAIL(c, "null buffer");
    if (!d_pair_off && total != (uint64_t)k * n) ARGFAIL(c, "total_pairs != n * k");
    if (total > 0xFFFFFFFFull) ARGFAIL(c, "pair indices are 32-bit");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const uint64_t M = 2 * n + total;
    int rc = mbls_ctx_reserve(c, M); if (rc) return rc;
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    mbls_ws wp = ws; wp.w += n;                                   // the pairs' view: item j of wp = workspace item n + j
    rc = ws_acquire(c, s); if (rc) return rc;
    uint32_t* st_item = d_status ? d_status : c->d_status;
    uint32_t* st_pair = c->d_status + n;                         // (cap >= 2 n + T words)
    uint32_t* map = nullptr;
    sbuf dmap(c, 9);
    if (d_pair_off) {      // the staging buffer is reused between calls: every entry the map kernel does not write must read as "no owner"
        HIPCHK(c, dmap.alloc(4 * (total ? total : 1))); map = dmap.as<uint32_t>();
        HIPCHK(c, hipMemsetAsync(map, 0xFF, 4 * (total ? total : 1), s));
    }
    HIPCHK(c, hipMemsetAsync(st_item, 0, 4 * n, s));
    if (st_item != c->d_status) HIPCHK(c, hipMemsetAsync(c->d_status, 0, 4 * n, s));
    if (total) HIPCHK(c, hipMemsetAsync(st_pair, 0, 4 * total, s));
    HIPCHK(c, hipMemsetAsync(d_results, 0, n, s));                // false until k_final has spoken
    // three chains side by side while they leave SIMDs idle: signatures (items [0, n)), keys and messages (items [n, n + T) through wp)
    const bool fork = n + 2 * total <= c->fork_max_items;
    hipStream_t s_sig = fork ? c->hs_b : s, s_msg = fork ? c->hs_c : s;
    if (fork) { HIPCHK(c, hipEventRecord(c->hs_ev, s)); HIPCHK(c, hipStreamWaitEvent(s_sig, c->hs_ev, 0)); HIPCHK(c, hipStreamWaitEvent(s_msg, c->hs_ev, 0)); }
    hipLaunchKernelGGL(k_sig, dim3(nblk(n)), dim3(WG), 0, s_sig, ws, d_sigs, st_item, n, 1);
    hipLaunchKernelGGL(k_sigpair_setup, dim3(nblk(n)), dim3(WG), 0, s_sig, ws, n);
    if (total) {
        if (d_pair_off) hipLaunchKernelGGL(k_pair_item_map, dim3(nblk(n)), dim3(WG), 0, s, d_pair_off, n, total, map);
        hipLaunchKernelGGL(k_blind_g1, dim3(nblk(total)), dim3(WG), 0, s, wp, d_pks96, (const uint64_t*)nullptr, st_pair, total);      // decode only: r_i = 1
        launch_hash(c, wp, d_msgs, msg_len, d_moff, st_pair, total, s_msg);
    }
    if (fork) {
        HIPCHK(c, hipEventRecord(c->hs_ev2, s_sig)); HIPCHK(c, hipEventRecord(c->hs_ev3, s_msg));
        HIPCHK(c, hipStreamWaitEvent(s, c->hs_ev2, 0)); HIPCHK(c, hipStreamWaitEvent(s, c->hs_ev3, 0));
    }
    // one Miller loop per pair, the signatures' pairs included: items [0, n + T)
    if (2 * (n + total) <= c->coop_max_items)
        coop_run(c, COOP_MILLER1, ws, (uint64_t)0, (uint64_t)1, (uint64_t)0, n + total, (uint32_t*)nullptr, (uint8_t*)nullptr, COOP_RES_ITEM, s);
    else
        launch_miller_single(c, ws, n + total, s);
    // per-item product trees over the pairs, then item i <- (sig pair) x (its pairs' product)
    uint64_t kmax = d_pair_off ? total : k;                      // a ragged layout may hold one long range: levels up to the whole list
    for (uint64_t half = 1; half < kmax; half *= 2)
        hipLaunchKernelGGL(k_f12_seg_tree_d, dim3(nblk(total)), dim3(WG), 0, s, wp, (const uint32_t*)map, d_pair_off, k, n, total, half);
    hipLaunchKernelGGL(k_f12_seg_gather, dim3(nblk(n)), dim3(WG), 0, s, ws, d_pair_off, (const uint32_t*)map, k, n, total, st_item, (const uint32_t*)st_pair);
    hipLaunchKernelGGL(k_f12_tree_d, dim3(nblk(n)), dim3(WG), 0, s, ws, 2 * n + total, n + total);
    if (n <= c->split_max_items && 2 * n <= c->round_items) hipLaunchKernelGGL(k_final2, dim3(nblk(2 * n)), dim3(WG), 0, s, ws, st_item, d_results, n);
    else hipLaunchKernelGGL(k_final, dim3(nblk(n)), dim3(WG), 0, s, ws, st_item, d_results, n);
    HIPCHK(c, hipGetLastError());
    return ws_release(c, s);
}
extern "C" int mbls_aggregate_verify_batch(mbls_ctx* c, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* moff,
        const uint8_t* pks96, const uint32_t* pair_off, uint32_t k, uint64_t n, uint8_t* results, uint32_t* status) {
    if (!c) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    if (n == 0) return MBLS_OK;
    if (!sigs || !results) ARGFAIL(c, "null buffer");
    if (pair_off && (!offsets_ok(pair_off, n) || pair_off[0] != 0)) ARGFAIL(c, "pair_offsets must start at 0 and be non-decreasing");
    const uint64_t total = pair_off ? pair_off[n] : (uint64_t)k * n;
    if (moff && !msg_offsets_ok(moff, total)) ARGFAIL(c, "msg_offsets must be non-decreasing, messages below 2^32 bytes");
    const uint64_t msg_first = moff ? moff[0] : 0;
    const size_t msg_total = moff ? (size_t)(moff[total] - moff[0]) : (size_t)msg_len * total;
    if (total && (!pks96 || (!msgs && msg_total))) ARGFAIL(c, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    sbuf ds(c, 0), dm(c, 1), dp(c, 2), doff(c, 3), dr(c, 4), dst(c, 5), dmo(c, 6);
    HIPCHK(c, ds.up(sigs, 96 * n)); HIPCHK(c, dm.up(msgs ? msgs + msg_first : nullptr, msg_total)); HIPCHK(c, dp.up(pks96, 96 * total));
    if (pair_off) HIPCHK(c, doff.up(pair_off, 4 * (n + 1)));
    const uint8_t* d_msgs = dm.as<uint8_t>();
    if (moff) { HIPCHK(c, dmo.up(moff, 8 * (total + 1))); d_msgs -= msg_first; }
    HIPCHK(c, dr.alloc(n)); HIPCHK(c, dst.alloc(4 * n));
    int rc = mbls_aggregate_verify_batch_device(c, ds.as<uint8_t>(), d_msgs, msg_len, moff ? dmo.as<uint64_t>() : nullptr, dp.as<uint8_t>(),
                                                pair_off ? doff.as<uint32_t>() : nullptr, k, total, n, dr.as<uint8_t>(), dst.as<uint32_t>(), c->hs_a);
    if (rc) { (void)hipStreamSynchronize(c->hs_a); (void)hipStreamSynchronize(c->hs_b); (void)hipStreamSynchronize(c->hs_c); c->ws_pending = false; return rc; }
    HIPCHK(c, hipStreamSynchronize(c->hs_a));
    c->ws_pending = false;
    HIPCHK(c, dr.down(results, n));
    if (status) HIPCHK(c, dst.down(status, 4 * n));
    return MBLS_OK;
}
static int verify_multiple_impl(mbls_ctx* c, const uint8_t* d_sigs, const uint8_t* d_apks, const uint8_t* d_pks, int pk_format,
        const uint32_t* d_pk_offsets, uint32_t k, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff, const uint64_t* d_rands, uint64_t n,
        uint8_t* d_result, uint32_t* d_status_or, void* stream, uint32_t* d_partial = nullptr, const mbls_keytable* tab = nullptr, const uint32_t* d_idx = nullptr,
        bool sigs_resident = false, bool hash_enqueued = false) {
    // sigs_resident: k_sig (decode + subgroup test) has run over these signatures on this workspace and the host has seen every one pass: d_sigs is not read.
    // hash_enqueued (batches whose chains run side by side only): the message phase is already on the context's message stream.
    if (!c || (!d_result && !d_partial)) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0 && d_partial) {     // an empty shard contributes (1, infinity, no status bits)
        mbls_ws none; none.w = nullptr; none.stride = 0;
        hipLaunchKernelGGL(k_vm_export, dim3(1), dim3(WG), 0, s, none, (const uint32_t*)nullptr, 1, d_partial);
        HIPCHK(c, hipGetLastError());
        return MBLS_OK;
    }
    if (n == 0) {     // empty iterator: S' = infinity, product of no pairings = 1 -> e(inf, -G1) = 1 -> true
        HIPCHK(c, hipMemsetAsync(d_result, 1, 1, s));
        if (d_status_or) HIPCHK(c, hipMemsetAsync(d_status_or, 0, 4, s));
        return MBLS_OK;
    }
    if (!d_rands) ARGFAIL(c, "verify_multiple without blinding scalars is forgeable: rands must not be NULL");
    if ((!d_sigs && !sigs_resident) || (!d_msgs && msg_len && !d_moff)) ARGFAIL(c, "null buffer");
    // batches of at most half a round: two lanes per message in the message phase (items [0, 2 n), slots no other chain touches)
    const bool pair_hash = n <= c->split_max_items && 2 * n <= c->round_items;
    int rc = mbls_ctx_reserve(c, pair_hash ? 2 * n : n); if (rc) return rc;
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    rc = ws_acquire(c, s); if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_scalar, 0, 64, s));
    if (!sigs_resident) HIPCHK(c, hipMemsetAsync(c->d_status, 0, 4 * n, s));      // (resident: cleared before k_sig, and the message phase may be writing its bits)
    if (d_partial) HIPCHK(c, hipMemsetAsync(d_partial, 0, MBLS_VM_PARTIAL_BYTES, s));       // not a record until k_vm_export has spoken
    else HIPCHK(c, hipMemsetAsync(d_result, 0, 1, s));                      // false until the tail kernel has spoken (fail closed)
    // Three independent chains: keys (aggregate, [r]apk), signatures (decode, subgroup check, [r]sig, sum tree), messages (hash). Below
    // half a round of lanes each of them leaves at least half the SIMDs idle, so they run side by side on the context's streams and join before
    // the Miller loops (2^15 sets: 13.4 ms instead of 18.3 one after the other); closer to a full round the three chains only get in each
    // other's way (65 535 sets: 24.7 ms side by side, 21.2 in a row) and stay on the caller's stream.
    const bool fork = 2 * n <= c->round_items;
    hipStream_t s_sig = fork ? c->hs_b : s, s_msg = fork ? c->hs_c : s;
    if (fork) { HIPCHK(c, hipEventRecord(c->hs_ev, s)); HIPCHK(c, hipStreamWaitEvent(s_sig, c->hs_ev, 0)); HIPCHK(c, hipStreamWaitEvent(s_msg, c->hs_ev, 0)); }
    // side by side, the message phase is the chain the sets' Miller loops wait for: it is ENQUEUED first (behind the ~20 launches of the signature chain's sum tree
    // it started 0.26 ms late: 2^14 sets 9.47 -> 9.2 ms)
    const bool hash_first = fork && !hash_enqueued;
    if (hash_first) launch_hash(c, ws, d_msgs, msg_len, d_moff, c->d_status, n, s_msg, pair_hash);
    if (tab) {      // sets given by indices into a resident key table: the indexed key sum (the keys were decoded and validated once)
        rc = table_acquire(c, tab, s); if (rc) return rc;
        hipLaunchKernelGGL(k_aggregate_indexed_d, dim3(nblk(n)), dim3(WG), 0, s, ws, (const uint32_t*)tab->d_recs, tab->size, d_idx, d_pk_offsets, k, MBLS_MODE_VERIFY, c->d_status, n);
    } else if (!d_apks)    // sets given by their wire-format keys: AggregatePublicKey::aggregate on the device first (src/aggregates.rs:29-39)
        launch_aggregate(ws, d_pks, d_pk_offsets, k, pk_format, MBLS_MODE_VERIFY, c->d_status, n, s);
    hipLaunchKernelGGL(k_blind_g1_d, dim3(nblk(n)), dim3(WG), 0, s, ws, tab ? (const uint8_t*)nullptr : d_apks, d_rands, c->d_status, n);
    // small batches (their Miller loops run one WAVE per pair, see npairing_finish): the signature chain is what the call waits for -- two lanes per signature
    if (2 * n <= c->coop_max_items)
        hipLaunchKernelGGL(k_blind_sig2_d, dim3(nblk(2 * n)), dim3(WG), 0, s_sig, ws, sigs_resident ? (const uint8_t*)nullptr : d_sigs, d_rands, c->d_status, n);
    else
        hipLaunchKernelGGL(k_blind_sig_d, dim3(nblk(n)), dim3(WG), 0, s_sig, ws, sigs_resident ? (const uint8_t*)nullptr : d_sigs, d_rands, c->d_status, n);
    const bool side_s_chain = !fork && s != c->hs_b;                  // the sum tree waits for the product tree's company (npairing_finish)
    if (!side_s_chain) g2_tree(c, ws, n, s_sig);
    if (fork) {     // S is complete: its Miller loop runs on one wave beside the other chains and the sets' Miller loops (most SIMDs are idle)
        HIPCHK(c, hipEventRecord(c->hs_ev2, s_sig));                 // the status bits and S exist
        if (!d_partial)                                              // (a shard's S joins the other shards' first: mbls_verify_multiple_finish_device)
            coop_run(c, COOP_SMILLER, ws, (uint64_t)0, (uint64_t)1, (uint64_t)0, (uint64_t)1, (uint32_t*)nullptr, (uint8_t*)nullptr, COOP_RES_ITEM, s_sig);
        HIPCHK(c, hipEventRecord(c->hs_ev, s_sig));                  // ... and its Miller value (awaited just before the tail)
    }
    if (!(hash_enqueued && fork) && !hash_first) launch_hash(c, ws, d_msgs, msg_len, d_moff, c->d_status, n, s_msg, pair_hash);
    if (fork) {      // the sets' Miller loops need the keys (this stream) and the messages; the signature chain is awaited before the tail (hs_ev)
        HIPCHK(c, hipEventRecord(c->hs_ev3, s_msg));
        HIPCHK(c, hipStreamWaitEvent(s, c->hs_ev3, 0));
    } else
        hipLaunchKernelGGL(k_status_or, dim3(nblk(n)), dim3(WG), 0, s, c->d_status, n, c->d_scalar);
    // a set whose signature is outside G2 (reference src/aggregates.rs:274-276), an undecodable member or a zero scalar makes the tail
    // answer false: the status bits are folded in on the device, the call only enqueues
    rc = npairing_finish(c, n, s, d_result, fork ? c->hs_ev : nullptr, fork, d_partial, side_s_chain); if (rc) return rc;
    if (d_status_or) HIPCHK(c, hipMemcpyAsync(d_status_or, c->d_scalar, 4, hipMemcpyDeviceToDevice, s));
    return ws_release(c, s);
}
// verify_multiple over sets named by indices into a resident key table (the deployment's form: validator keys are decoded once, a set is a list of
// validator indices): set i owns indices [d_offsets[i], d_offsets[i+1]) of d_key_idx, or k each. d_partial (optional) instead of d_result: the shard form.
extern "C" int mbls_verify_multiple_sets_indexed_device(mbls_ctx* c, const mbls_keytable* t, const uint8_t* d_sigs, const uint32_t* d_key_idx, const uint32_t* d_offsets,
        uint32_t k, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff, const uint64_t* d_rands, uint64_t n, uint8_t* d_result, uint32_t* d_status_or,
        uint8_t* d_partial, void* stream) {
    if (!c || !t || (!d_result && !d_partial)) return MBLS_ERR_ARGUMENT;
    if (n && !d_key_idx) return MBLS_ERR_ARGUMENT;
    if (((uintptr_t)d_partial) & 3) return MBLS_ERR_ARGUMENT;
    {
        mbls_lock lk(c->mu);
        if (t->c != c) ARGFAIL(c, "key table belongs to another context");
    }
    return verify_multiple_impl(c, d_sigs, nullptr, nullptr, MBLS_PK_UNCOMPRESSED, d_offsets, k, d_msgs, msg_len, d_moff, d_rands, n, d_partial ? nullptr : d_result,
                                d_partial ? nullptr : d_status_or, stream, (uint32_t*)d_partial, t, d_key_idx);
}
// One shard of a verify_multiple that is spread over several devices or processes: everything up to the shard's
// Miller product and signature sum, left as one MBLS_VM_PARTIAL_BYTES record in device memory. Enqueues only.
extern "C" int mbls_verify_multiple_partial_device(mbls_ctx* c, const uint8_t* d_sigs, const uint8_t* d_apks, const uint8_t* d_pks, int pk_format,
        const uint32_t* d_pk_offsets, uint32_t k, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff, const uint64_t* d_rands, uint64_t n,
        uint8_t* d_partial, void* stream) {
    if (!c || !d_partial) return MBLS_ERR_ARGUMENT;
    if (n && !d_apks && !d_pks) return MBLS_ERR_ARGUMENT;
    if (!d_apks && pk_format != MBLS_PK_COMPRESSED && pk_format != MBLS_PK_UNCOMPRESSED) return MBLS_ERR_ARGUMENT;
    if (((uintptr_t)d_partial) & 3) return MBLS_ERR_ARGUMENT;
    return verify_multiple_impl(c, d_sigs, d_apks, d_apks ? nullptr : d_pks, pk_format, d_apks ? nullptr : d_pk_offsets, k, d_msgs, msg_len, d_moff, d_rands, n,
                                nullptr, nullptr, stream, (uint32_t*)d_partial);
}
// The exchange step's second half: G records (the shards' in any fixed order -- every participant that uses the same order computes the same
// bool) -> product and sum trees over the records, then the tail of the one-device check: the Miller loop of (S, -G1), the product, ONE final
// exponentiation, the comparison, the status bits (program vmtail; reference src/aggregates.rs:307-315). Enqueues only.
extern "C" int mbls_verify_multiple_finish_device(mbls_ctx* c, const uint8_t* d_partials, uint64_t G, uint8_t* d_result, uint32_t* d_status_or, void* stream) {
    if (!c || !d_result) return MBLS_ERR_ARGUMENT;
    mbls_lock lk(c->mu);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipSetDevice(c->device));
    if (G == 0) {      // no shards: the empty iterator
        HIPCHK(c, hipMemsetAsync(d_result, 1, 1, s));
        if (d_status_or) HIPCHK(c, hipMemsetAsync(d_status_or, 0, 4, s));
        return MBLS_OK;
    }
    if (!d_partials || (((uintptr_t)d_partials) & 3)) ARGFAIL(c, "null or unaligned records");
    int rc = mbls_ctx_reserve(c, G); if (rc) return rc;
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    rc = ws_acquire(c, s); if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_scalar, 0, 64, s));
    HIPCHK(c, hipMemsetAsync(d_result, 0, 1, s));
    hipLaunchKernelGGL(k_vm_import, dim3(nblk(G)), dim3(WG), 0, s, ws, (const uint32_t*)d_partials, G, c->d_scalar);
    f12_tree(c, ws, G, s);
    g2_tree(c, ws, G, s);
    coop_run(c, COOP_VMTAIL, ws, (uint64_t)0, (uint64_t)1, (uint64_t)0, (uint64_t)1, c->d_scalar, d_result, COOP_RES_BATCH, s);
    HIPCHK(c, hipGetLastError());
    if (d_status_or) HIPCHK(c, hipMemcpyAsync(d_status_or, c->d_scalar, 4, hipMemcpyDeviceToDevice, s));
    return ws_release(c, s);
}
extern "C" int mbls_verify_multiple_aggregate_signatures_device(mbls_ctx* c, const uint8_t* d_sigs, const uint8_t* d_apks, const uint8_t* d_msgs,
        uint32_t msg_len, const uint64_t* d_moff, const uint64_t* d_rands, uint64_t n, uint8_t* d_result, uint32_t* d_status_or, void* stream) {
    if (!c) return MBLS_ERR_ARGUMENT;
    if (n && !d_apks) return MBLS_ERR_ARGUMENT;
    return verify_multiple_impl(c, d_sigs, d_apks, nullptr, 0, nullptr, 0, d_msgs, msg_len, d_moff, d_rands, n, d_result, d_status_or, stream);
}
extern "C" int mbls_verify_multiple_sets_device(mbls_ctx* c, const uint8_t* d_sigs, const uint8_t* d_pks, int pk_format, const uint32_t* d_pk_offsets,
        uint32_t k, const uint8_t* d_msgs, uint32_t msg_len, const uint64_t* d_moff, const uint64_t* d_rands, uint64_t n, uint8_t* d_result,
        uint32_t* d_status_or, void* stream) {
    if (!c) return MBLS_ERR_ARGUMENT;
    if (pk_format != MBLS_PK_COMPRESSED && pk_format != MBLS_PK_UNCOMPRESSED) return MBLS_ERR_ARGUMENT;
    if (n && !d_pks) return MBLS_ERR_ARGUMENT;
    return verify_multiple_impl(c, d_sigs, nullptr, d_pks, pk_format, d_pk_offsets, k, d_msgs, msg_len, d_moff, d_rands, n, d_result, d_status_or, stream);
}
extern "C" int mbls_verify_multiple_aggregate_signatures(mbls_ctx* c, const uint8_t* sigs96, const uint8_t* apks96, const uint8_t* msgs,
        uint32_t msg_len, const uint64_t* moff, const uint64_t* rands, size_t n) {
    if (!c) return 0;
    if (n == 0) return 1;
    if (moff && !msg_offsets_ok(moff, n)) return 0;
    const uint64_t msg_first = moff ? moff[0] : 0;
    const size_t msg_total = moff ? (size_t)(moff[n] - moff[0]) : (size_t)msg_len * n;
    if (!sigs96 || !apks96 || !rands || (!msgs && msg_total)) return 0;
    mbls_lock lk(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return 0;
    sbuf ds(c, 0), da(c, 1), dm(c, 2), dr(c, 3), dmo(c, 6), dres(c, 4);
    if (ds.up(sigs96, 96 * n) != hipSuccess || da.up(apks96, 96 * n) != hipSuccess || dm.up(msgs ? msgs + msg_first : nullptr, msg_total) != hipSuccess ||
        dr.up(rands, 8 * n) != hipSuccess || (moff && dmo.up(moff, 8 * (n + 1)) != hipSuccess) || dres.alloc(8) != hipSuccess) return 0;
    if (mbls_verify_multiple_aggregate_signatures_device(c, ds.as<uint8_t>(), da.as<uint8_t>(), dm.as<uint8_t>() - msg_first, msg_len,
                                                         moff ? dmo.as<uint64_t>() : nullptr, dr.as<uint64_t>(), n, dres.as<uint8_t>(), nullptr, c->hs_a)) return 0;
    uint8_t r = 0;
    if (hipStreamSynchronize(c->hs_a) != hipSuccess) return 0;
    c->ws_pending = false;
    if (dres.down(&r, 1) != hipSuccess) return 0;
    return r;
}
// The reference's own shape (src/aggregates.rs:261-316): its loop tests set i's signature for the subgroup (:272-275) BEFORE it draws rand[i] from the caller's
// generator (:280-287) and returns at the first signature outside G2 -- so a rejected batch leaves the generator after exactly as many draws as sets came before
// the bad one. One call does the same: the signatures are decoded and tested first (k_sig, beside the message phase when the batch leaves room), the host reads
// the verdicts, `draw` is asked for exactly the scalars the reference would have drawn, and what follows skips the second subgroup test (the points are in the slots).
// The two halves of that call, shared with the several-device form (mbls_multi_verify_multiple_aggregate_signatures_rng): phase 1 stages a shard's sets, decodes and
// tests its signatures (they stay in the workspace), starts the message phase and reads the verdicts back; phase 2 takes the scalars and runs the rest -- up to the
// bool (d_partial == nullptr) or up to the shard's record. The caller holds the context's lock across both.
struct vm_rng_stage {
    sbuf ds, da, dm, dr, dmo, dout;
    const uint8_t* d_msgs = nullptr; const uint64_t* d_moff = nullptr;
    vm_rng_stage(mbls_ctx* c) : ds(c, 0), da(c, 1), dm(c, 2), dr(c, 3), dmo(c, 6), dout(c, 4) {}
};
static void vm_rng_drain(mbls_ctx* c) { (void)hipStreamSynchronize(c->hs_a); (void)hipStreamSynchronize(c->hs_b); (void)hipStreamSynchronize(c->hs_c); c->ws_pending = false; }
// sigs96 / apks96: the shard's own sets; msgs: the buffer the (absolute) offsets of moff[0 .. n] point into, or n x msg_len bytes. -> MBLS_OK and st[0 .. n) (status words)
static int vm_rng_phase1(mbls_ctx* c, vm_rng_stage& g, const uint8_t* sigs96, const uint8_t* apks96, const uint8_t* msgs, uint32_t msg_len, const uint64_t* moff, size_t n,
                         size_t out_bytes, uint32_t* st) {
    if (hipSetDevice(c->device) != hipSuccess) return MBLS_ERR_DEVICE;
    if (g.dout.alloc(out_bytes) != hipSuccess) return MBLS_ERR_DEVICE;
    if (n == 0) return MBLS_OK;
    const uint64_t msg_first = moff ? moff[0] : 0;
    const size_t msg_total = moff ? (size_t)(moff[n] - moff[0]) : (size_t)msg_len * n;
    if (g.ds.up(sigs96, 96 * n) != hipSuccess || g.da.up(apks96, 96 * n) != hipSuccess || g.dm.up(msgs ? msgs + msg_first : nullptr, msg_total) != hipSuccess ||
        g.dr.alloc(8 * n) != hipSuccess || (moff && g.dmo.up(moff, 8 * (n + 1)) != hipSuccess)) return MBLS_ERR_DEVICE;
    g.d_msgs = g.dm.as<uint8_t>() - msg_first; g.d_moff = moff ? g.dmo.as<uint64_t>() : nullptr;
    hipStream_t s = c->hs_a;
    const bool pair_hash = n <= c->split_max_items && 2 * n <= c->round_items, fork = 2 * n <= c->round_items;       // as verify_multiple_impl decides
    int rc = mbls_ctx_reserve(c, pair_hash ? 2 * n : n); if (rc) return rc;
    mbls_ws ws; ws.w = c->d_w; ws.stride = c->cap;
    rc = ws_acquire(c, s); if (rc) return rc;
    hipStream_t s_sig = fork ? c->hs_b : s;
    if (hipMemsetAsync(c->d_status, 0, 4 * n, s) != hipSuccess) { vm_rng_drain(c); return MBLS_ERR_DEVICE; }
    if (fork) { (void)hipEventRecord(c->hs_ev, s); (void)hipStreamWaitEvent(c->hs_b, c->hs_ev, 0); (void)hipStreamWaitEvent(c->hs_c, c->hs_ev, 0); }
    if (2 * n <= c->coop_max_items) hipLaunchKernelGGL(k_sig2, dim3(nblk(2 * n)), dim3(WG), 0, s_sig, ws, (const uint8_t*)g.ds.as<uint8_t>(), c->d_status, (uint64_t)n);
    else hipLaunchKernelGGL(k_sig, dim3(nblk(n)), dim3(WG), 0, s_sig, ws, (const uint8_t*)g.ds.as<uint8_t>(), c->d_status, (uint64_t)n, 1);
    if (fork) launch_hash(c, ws, g.d_msgs, msg_len, g.d_moff, c->d_status, n, c->hs_c, pair_hash);       // the message phase does not wait for the host
    if (hipStreamSynchronize(s_sig) != hipSuccess || hipMemcpy(st, c->d_status, 4 * n, hipMemcpyDeviceToHost) != hipSuccess) { vm_rng_drain(c); return MBLS_ERR_DEVICE; }
    return MBLS_OK;
}
// rands[0 .. n): the shard's scalars. d_partial == false: the bool is left in g.dout (1 byte); true: the shard's record (MBLS_VM_PARTIAL_BYTES). Enqueues on hs_a.
static int vm_rng_phase2(mbls_ctx* c, vm_rng_stage& g, uint32_t msg_len, const uint64_t* rands, size_t n, bool partial) {
    if (hipSetDevice(c->device) != hipSuccess) return MBLS_ERR_DEVICE;
    if (n && g.dr.up(rands, 8 * n) != hipSuccess) return MBLS_ERR_DEVICE;
    return verify_multiple_impl(c, nullptr, g.da.as<uint8_t>(), nullptr, 0, nullptr, 0, g.d_msgs, msg_len, g.d_moff, g.dr.as<uint64_t>(), n, partial ? nullptr : g.dout.as<uint8_t>(),
                                nullptr, c->hs_a, partial ? g.dout.as<uint32_t>() : nullptr, nullptr, nullptr, true, true);
}
extern "C" int mbls_verify_multiple_aggregate_signatures_rng(mbls_ctx* c, const uint8_t* sigs96, const uint8_t* apks96, const uint8_t* msgs,
        uint32_t msg_len, const uint64_t* moff, size_t n, mbls_scalar_source draw, void* user) {
    if (!c) return 0;
    if (n == 0) return 1;                                                   // empty iterator: true, the generator untouched
    if (!draw) return 0;
    if (moff && !msg_offsets_ok(moff, n)) return 0;
    const size_t msg_total = moff ? (size_t)(moff[n] - moff[0]) : (size_t)msg_len * n;
    if (!sigs96 || !apks96 || (!msgs && msg_total)) return 0;
    mbls_lock lk(c->mu);
    std::vector<uint64_t> rands; std::vector<uint32_t> st;
    try { rands.resize(n); st.resize(n); } catch (...) { return 0; }
    vm_rng_stage g(c);
    if (vm_rng_phase1(c, g, sigs96, apks96, msgs, msg_len, moff, n, 8, st.data())) return 0;
    size_t reached = n;                                                     // the sets the reference's loop draws a scalar for
    for (size_t i = 0; i < n; i++)
        if (st[i] & (MBLS_ST_BAD_SIG_ENCODING | MBLS_ST_SIG_NOT_IN_G2)) { reached = i; break; }
    if (reached) draw(user, rands.data(), (uint64_t)reached);
    if (reached < n) { vm_rng_drain(c); return 0; }                         // :273-275 (only the message phase is left to wait for)
    const int rc = vm_rng_phase2(c, g, msg_len, rands.data(), n, false);
    std::fill(rands.begin(), rands.end(), 0);
    if (rc) { vm_rng_drain(c); return 0; }
    uint8_t r = 0;
    if (hipStreamSynchronize(c->hs_a) != hipSuccess) { vm_rng_drain(c); return 0; }
    c->ws_pending = false;
    if (g.dout.down(&r, 1) != hipSuccess) return 0;
    return r;
}

// ------------------------------------------------------------------------------------------------ several GPUs behind one handle
// items are independent (reference src/aggregates.rs:177-215 holds no state between calls), so a batch is cut
// into contiguous shards, one per device; each device has its own context, and one host thread per device stages and verifies its
// shard, writing results straight into the caller's buffers. No device talks to another. Key tables are replicated on every device.
#include <thread>
// RCCL, opened at run time: libmbls_hip.so has no link-time dependency on it (a host without RCCL, or a one-device handle that never exchanges anything, loads
// the library all the same). dlopen finds the copy the process already holds (PyTorch-ROCm ships its own librccl.so.1) before the one under /opt/rocm.
struct rccl_api {
    void* h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static bool rccl_open(rccl_api* a, char* why, size_t why_len) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) { a->h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD); if (a->h) break; }       // a copy the process holds already
    for (const char* nm : names) { if (a->h) break; a->h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); }
    if (!a->h) { snprintf(why, why_len, "librccl.so.1 not found (%s)", dlerror()); return false; }
    a->CommInitAll = (decltype(a->CommInitAll))dlsym(a->h, "ncclCommInitAll");
    a->CommDestroy = (decltype(a->CommDestroy))dlsym(a->h, "ncclCommDestroy");
    a->GroupStart = (decltype(a->GroupStart))dlsym(a->h, "ncclGroupStart");
    a->GroupEnd = (decltype(a->GroupEnd))dlsym(a->h, "ncclGroupEnd");
    a->AllGather = (decltype(a->AllGather))dlsym(a->h, "ncclAllGather");
    a->GetErrorString = (decltype(a->GetErrorString))dlsym(a->h, "ncclGetErrorString");
    if (!a->CommInitAll || !a->CommDestroy || !a->GroupStart || !a->GroupEnd || !a->AllGather || !a->GetErrorString) {
        snprintf(why, why_len, "librccl.so.1 lacks an entry point"); return false;
    }
    return true;
}
// Several devices behind one handle. The exchange step of the paths that have one (the accept bitmap of mbls_multi_fast_aggregate_verify_bitmap, the partial
// records of mbls_multi_verify_multiple_aggregate_signatures) is an RCCL all-gather between the devices' buffers -- over xGMI on an MI355X node -- when the
// handle could set up a communicator; otherwise (RCCL absent, the same device listed twice: RCCL refuses duplicates, MBLS_MULTI_NO_RCCL set) the records travel
// through host memory and `rccl_note` says why. Never a restart, never a different result.
struct mbls_multi {
    std::vector<mbls_ctx*> ctx; char err[256] = {}; std::mutex mu;
    rccl_api rccl; std::vector<ncclComm_t> comms; bool rccl_on = false; char rccl_note[256] = {};
    std::vector<uint64_t*> d_bm_send, d_bm_all; uint64_t bm_words_cap = 0;        // per device: this shard's bitmap words / the gathered bitmap
    std::vector<uint8_t*> d_rec_all;                                              // per device: G partial records of verify_multiple
};
struct mbls_multi_keytable { mbls_multi* m = nullptr; std::vector<mbls_keytable*> tab; };

static void multi_rccl_setup(mbls_multi* m, const int* device_ids, int G) {
    if (getenv("MBLS_MULTI_NO_RCCL")) { snprintf(m->rccl_note, sizeof(m->rccl_note), "host join: MBLS_MULTI_NO_RCCL is set"); return; }
    for (int a = 0; a < G; a++) for (int b = a + 1; b < G; b++) if (device_ids[a] == device_ids[b]) {
        snprintf(m->rccl_note, sizeof(m->rccl_note), "host join: device %d is listed more than once (RCCL wants one rank per device)", device_ids[a]); return;
    }
    char why[160] = {};
    if (!rccl_open(&m->rccl, why, sizeof(why))) { snprintf(m->rccl_note, sizeof(m->rccl_note), "host join: %s", why); return; }
    m->comms.assign((size_t)G, nullptr);
    // RCCL may print a version banner to the C stdout when the first communicator of a process is made. The library does NOT touch the process-wide descriptor 1
    // for that by default (another thread of the host may be writing there in that window): an application that emits a protocol on stdout protects it itself, as
    // bench.py does (protect_stdout), or sets NCCL_DEBUG / RCCL's own switches. MBLS_MULTI_REDIRECT_STDOUT=1 opts a single-threaded host in to the old
    // behaviour: descriptor 1 points at stderr while the communicator is set up.
    const bool redirect = getenv("MBLS_MULTI_REDIRECT_STDOUT") != nullptr;
    int saved = -1;
    if (redirect) { fflush(stdout); saved = dup(1); if (saved >= 0) (void)dup2(2, 1); }
    const ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), G, device_ids);
    if (redirect) { fflush(stdout); if (saved >= 0) { (void)dup2(saved, 1); (void)close(saved); } }
    if (r != ncclSuccess) {
        snprintf(m->rccl_note, sizeof(m->rccl_note), "host join: ncclCommInitAll failed: %s", m->rccl.GetErrorString(r)); m->comms.clear(); return;
    }
    m->rccl_on = true;
    snprintf(m->rccl_note, sizeof(m->rccl_note), "RCCL all-gather over a communicator of %d device(s)", G);
}
extern "C" int mbls_multi_create(mbls_multi** out, const int* device_ids, int n_devices) {
    if (!out || !device_ids || n_devices <= 0) return MBLS_ERR_ARGUMENT;
    mbls_multi* m = new (std::nothrow) mbls_multi();
    if (!m) return MBLS_ERR_DEVICE;
    for (int g = 0; g < n_devices; g++) {          // the same device may be listed more than once (two contexts share it)
        mbls_ctx* c = nullptr;
        int rc = mbls_ctx_create(&c, device_ids[g]);
        if (rc) { for (mbls_ctx* x : m->ctx) mbls_ctx_destroy(x); delete m; return rc; }
        m->ctx.push_back(c);
    }
    try {
        m->d_bm_send.assign((size_t)n_devices, nullptr); m->d_bm_all.assign((size_t)n_devices, nullptr); m->d_rec_all.assign((size_t)n_devices, nullptr);
        multi_rccl_setup(m, device_ids, n_devices);
    } catch (...) { for (mbls_ctx* x : m->ctx) mbls_ctx_destroy(x); delete m; return MBLS_ERR_DEVICE; }
    *out = m; return MBLS_OK;
}
extern "C" void mbls_multi_destroy(mbls_multi* m) {
    if (!m) return;
    for (size_t g = 0; g < m->ctx.size(); g++) {
        (void)hipSetDevice(m->ctx[g]->device); (void)hipDeviceSynchronize();
        if (g < m->comms.size() && m->comms[g]) (void)m->rccl.CommDestroy(m->comms[g]);
        if (m->d_bm_send[g]) (void)hipFree(m->d_bm_send[g]);
        if (m->d_bm_all[g]) (void)hipFree(m->d_bm_all[g]);
        if (m->d_rec_all[g]) (void)hipFree(m->d_rec_all[g]);
    }
    for (mbls_ctx* c : m->ctx) mbls_ctx_destroy(c);
    delete m;
}
// 1: the handle's exchange steps run as RCCL all-gathers between the devices; 0: through host memory (mbls_multi_exchange_note says why)
extern "C" int mbls_multi_rccl_active(const mbls_multi* m) { return m && m->rccl_on ? 1 : 0; }
extern "C" const char* mbls_multi_exchange_note(mbls_multi* m) { return m ? m->rccl_note : "null handle"; }
// Exchange `bytes` bytes per device: device g contributes d_send[g] and ends with all G contributions, in device order, in d_recv[g] (G * bytes). The devices'
// streams `st[g]` carry the operation; on return it has completed everywhere. RCCL when the handle has a communicator, host memory otherwise.
static int multi_allgather(mbls_multi* m, const std::vector<const void*>& d_send, const std::vector<void*>& d_recv, size_t bytes, const std::vector<hipStream_t>& st) {
    const size_t G = m->ctx.size();
    if (m->rccl_on) {
        ncclResult_t r = m->rccl.GroupStart();
        for (size_t g = 0; g < G && r == ncclSuccess; g++) {
            if (hipSetDevice(m->ctx[g]->device) != hipSuccess) { (void)m->rccl.GroupEnd(); snprintf(m->err, sizeof(m->err), "hipSetDevice failed"); return MBLS_ERR_DEVICE; }
            r = m->rccl.AllGather(d_send[g], d_recv[g], bytes, ncclUint8, m->comms[g], st[g]);
        }
        const ncclResult_t r2 = m->rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { snprintf(m->err, sizeof(m->err), "RCCL all-gather failed: %s", m->rccl.GetErrorString(r)); return MBLS_ERR_DEVICE; }
        for (size_t g = 0; g < G; g++) {
            if (hipSetDevice(m->ctx[g]->device) != hipSuccess || hipStreamSynchronize(st[g]) != hipSuccess) { snprintf(m->err, sizeof(m->err), "all-gather: device %d did not complete", m->ctx[g]->device); return MBLS_ERR_DEVICE; }
        }
        return MBLS_OK;
    }
    std::vector<uint8_t> host;
    try { host.resize(G * bytes); } catch (...) { snprintf(m->err, sizeof(m->err), "out of host memory"); return MBLS_ERR_DEVICE; }
    for (size_t g = 0; g < G; g++)
        if (hipSetDevice(m->ctx[g]->device) != hipSuccess || hipStreamSynchronize(st[g]) != hipSuccess ||
            hipMemcpy(host.data() + g * bytes, d_send[g], bytes, hipMemcpyDeviceToHost) != hipSuccess) { snprintf(m->err, sizeof(m->err), "host join: download from device %d failed", m->ctx[g]->device); return MBLS_ERR_DEVICE; }
    for (size_t g = 0; g < G; g++)
        if (hipSetDevice(m->ctx[g]->device) != hipSuccess || hipMemcpy(d_recv[g], host.data(), G * bytes, hipMemcpyHostToDevice) != hipSuccess) { snprintf(m->err, sizeof(m->err), "host join: upload to device %d failed", m->ctx[g]->device); return MBLS_ERR_DEVICE; }
    return MBLS_OK;
}
extern "C" int mbls_multi_device_count(const mbls_multi* m) { return m ? (int)m->ctx.size() : 0; }
extern "C" const char* mbls_multi_last_error(mbls_multi* m) { return m ? m->err : "null handle"; }
extern "C" mbls_ctx* mbls_multi_context(mbls_multi* m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }
extern "C" int mbls_multi_reserve(mbls_multi* m, uint64_t max_items) {
    if (!m) return MBLS_ERR_ARGUMENT;
    const uint64_t G = m->ctx.size();
    for (mbls_ctx* c : m->ctx) { int rc = mbls_ctx_reserve(c, (max_items + G - 1) / G); if (rc) return rc; }
    return MBLS_OK;
}
// shard g of n items over G devices: [n g / G, n (g + 1) / G)  (the same partition as milagro_bls_amd/shard.py)
static inline uint64_t shard_lo(uint64_t n, uint64_t g, uint64_t G) { return (uint64_t)(((unsigned __int128)n * g) / G); }

static int multi_run(mbls_multi* m, const mbls_multi_keytable* mt, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* moff,
                     const uint8_t* pks, int fmt, const uint32_t* idx, const uint32_t* off, uint64_t n, uint32_t k, int mode, uint8_t* results, uint32_t* status) {
    if (!m || (mt && mt->m != m)) return MBLS_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(m->mu);
    const uint64_t G = m->ctx.size();
    const size_t unit = mt ? 4 : (fmt == MBLS_PK_COMPRESSED ? 48 : 96);
    std::vector<int> rcs(G, MBLS_OK);
    std::vector<std::thread> th;
    auto work = [&](uint64_t g) {
        const uint64_t lo = shard_lo(n, g, G), hi = shard_lo(n, g + 1, G), cnt = hi - lo;
        if (!cnt) return;
        // uniform layouts advance the base pointers; offset tables are absolute, so their slice [lo, hi] goes with the unmoved base
        const uint8_t* s_msgs = (moff || !msgs) ? msgs : msgs + (uint64_t)msg_len * lo;
        const uint8_t* s_pks = (off || !pks) ? pks : pks + unit * (uint64_t)k * lo;
        const uint32_t* s_idx = (off || !idx) ? idx : idx + (uint64_t)k * lo;
        rcs[g] = verify_host(m->ctx[g], sigs ? sigs + 96 * lo : nullptr, s_msgs, msg_len, moff ? moff + lo : nullptr, s_pks, fmt,
                             mt ? mt->tab[g] : nullptr, s_idx, off ? off + lo : nullptr, cnt, k, mode, results ? results + lo : nullptr,
                             status ? status + lo : nullptr);
    };
    try { for (uint64_t g = 1; g < G; g++) th.emplace_back(work, g); } catch (...) { for (auto& t : th) t.join(); snprintf(m->err, sizeof(m->err), "cannot start a host thread"); return MBLS_ERR_DEVICE; }
    work(0);
    for (auto& t : th) t.join();
    for (uint64_t g = 0; g < G; g++)
        if (rcs[g]) { snprintf(m->err, sizeof(m->err), "device %d (shard %llu): %s", m->ctx[g]->device, (unsigned long long)g, m->ctx[g]->err); return rcs[g]; }
    return MBLS_OK;
}
extern "C" int mbls_multi_fast_aggregate_verify_batch(mbls_multi* m, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* moff,
        const uint8_t* pks, int fmt, const uint32_t* off, uint64_t n, uint32_t k, uint8_t* results, uint32_t* status) {
    return multi_run(m, nullptr, sigs, msgs, msg_len, moff, pks, fmt, nullptr, off, n, k, MBLS_MODE_FAST_AGGREGATE, results, status);
}
extern "C" int mbls_multi_verify_batch(mbls_multi* m, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* moff, const uint8_t* pks, int fmt,
        uint64_t n, uint8_t* results, uint32_t* status) {
    return multi_run(m, nullptr, sigs, msgs, msg_len, moff, pks, fmt, nullptr, nullptr, n, 1, MBLS_MODE_VERIFY, results, status);
}
// verify_multiple over the devices of the handle: device g takes sets [n g / G, n (g + 1) / G) up to its partial record
// (one host thread per device), the records meet on the first device, which joins them and runs the tail. The same bool as the one-device call.
extern "C" int mbls_multi_verify_multiple_aggregate_signatures(mbls_multi* m, const uint8_t* sigs96, const uint8_t* apks96, const uint8_t* msgs,
        uint32_t msg_len, const uint64_t* moff, const uint64_t* rands, size_t n) {
    if (!m) return 0;
    if (n == 0) return 1;
    if (moff && !msg_offsets_ok(moff, n)) return 0;
    if (!sigs96 || !apks96 || !rands || (!msgs && (moff ? moff[n] != moff[0] : msg_len != 0))) return 0;
    std::lock_guard<std::mutex> lk(m->mu);
    const uint64_t G = m->ctx.size();
    std::vector<int> rcs(G, MBLS_OK); std::vector<std::thread> th;
    std::vector<const void*> d_send(G, nullptr); std::vector<void*> d_recv(G, nullptr); std::vector<hipStream_t> st(G, nullptr);
    auto work = [&](uint64_t g) {
        mbls_ctx* c = m->ctx[g];
        const uint64_t lo = shard_lo(n, g, G), hi = shard_lo(n, g + 1, G), cnt = hi - lo;
        mbls_lock lk2(c->mu);
        if (hipSetDevice(c->device) != hipSuccess) { rcs[g] = MBLS_ERR_DEVICE; return; }
        if (!m->d_rec_all[g] && hipMalloc(&m->d_rec_all[g], G * MBLS_VM_PARTIAL_BYTES) != hipSuccess) { m->d_rec_all[g] = nullptr; rcs[g] = MBLS_ERR_DEVICE; return; }
        const uint64_t first = moff ? moff[lo] : (uint64_t)msg_len * lo;
        const size_t mbytes = moff ? (size_t)(moff[hi] - moff[lo]) : (size_t)msg_len * cnt;
        sbuf ds(c, 0), da(c, 1), dm(c, 2), dr(c, 3), dmo(c, 6), drec(c, 4);
        if (ds.up(sigs96 + 96 * lo, 96 * cnt) != hipSuccess || da.up(apks96 + 96 * lo, 96 * cnt) != hipSuccess ||
            dm.up(msgs ? msgs + first : nullptr, mbytes) != hipSuccess || dr.up(rands + lo, 8 * cnt) != hipSuccess ||
            (moff && dmo.up(moff + lo, 8 * (cnt + 1)) != hipSuccess) || drec.alloc(MBLS_VM_PARTIAL_BYTES) != hipSuccess) { rcs[g] = MBLS_ERR_DEVICE; return; }
        // offset tables are absolute: the staged message bytes start at the shard's first offset
        rcs[g] = mbls_verify_multiple_partial_device(c, ds.as<uint8_t>(), da.as<uint8_t>(), nullptr, 0, nullptr, 0, dm.as<uint8_t>() - (moff ? first : 0), msg_len,
                                                     moff ? dmo.as<uint64_t>() : nullptr, dr.as<uint64_t>(), cnt, drec.as<uint8_t>(), c->hs_a);
        const bool synced = hipStreamSynchronize(c->hs_a) == hipSuccess;
        (void)hipStreamSynchronize(c->hs_b); (void)hipStreamSynchronize(c->hs_c);
        c->ws_pending = false;
        if (!rcs[g] && !synced) rcs[g] = MBLS_ERR_DEVICE;
        d_send[g] = drec.p; d_recv[g] = m->d_rec_all[g]; st[g] = c->hs_a;       // the shard's record stays on its device: the exchange step follows
    };
    try { for (uint64_t g = 1; g < G; g++) th.emplace_back(work, g); } catch (...) { for (auto& t : th) t.join(); snprintf(m->err, sizeof(m->err), "cannot start a host thread"); return 0; }
    work(0);
    for (auto& t : th) t.join();
    for (uint64_t g = 0; g < G; g++)
        if (rcs[g]) { snprintf(m->err, sizeof(m->err), "device %d (shard %llu): %s", m->ctx[g]->device, (unsigned long long)g, m->ctx[g]->err); return 0; }
    // THE exchange step: every device ends with all G records -- an RCCL all-gather of G x 896 bytes between the devices when the handle
    // has a communicator, host memory otherwise; the join then runs on the first device (any of them would reach the same bool)
    if (multi_allgather(m, d_send, d_recv, MBLS_VM_PARTIAL_BYTES, st)) return 0;
    mbls_ctx* c = m->ctx[0];
    mbls_lock lk0(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return 0;
    sbuf dres(c, 4); uint8_t r = 0;
    if (dres.alloc(8) != hipSuccess) return 0;
    if (mbls_verify_multiple_finish_device(c, m->d_rec_all[0], G, dres.as<uint8_t>(), nullptr, c->hs_a)) { (void)hipStreamSynchronize(c->hs_a); c->ws_pending = false; return 0; }
    if (hipStreamSynchronize(c->hs_a) != hipSuccess) return 0;
    c->ws_pending = false;
    if (dres.down(&r, 1) != hipSuccess) return 0;
    return r;
}
// The same over several devices with the reference's RNG ORDER and without a second subgroup test (what mbls_verify_multiple_aggregate_signatures_rng is to one
// device): every device decodes and tests its shard's signatures (they stay in its workspace) beside its message phase; the host reads the verdicts, finds the first
// bad signature of the WHOLE batch, asks `draw` once for the scalars of the sets in front of it (reference src/aggregates.rs:272-287), and -- all signatures good --
// every device goes on to its record, the records are exchanged (RCCL / host) and joined. One host thread per device holds its context across both phases.
#include <condition_variable>
extern "C" int mbls_multi_verify_multiple_aggregate_signatures_rng(mbls_multi* m, const uint8_t* sigs96, const uint8_t* apks96, const uint8_t* msgs,
        uint32_t msg_len, const uint64_t* moff, size_t n, mbls_scalar_source draw, void* user) {
    if (!m) return 0;
    if (n == 0) return 1;                                                   // empty iterator: true, the generator untouched
    if (!draw) return 0;
    if (moff && !msg_offsets_ok(moff, n)) return 0;
    if (!sigs96 || !apks96 || (!msgs && (moff ? moff[n] != moff[0] : msg_len != 0))) return 0;
    std::lock_guard<std::mutex> lk(m->mu);
    const uint64_t G = m->ctx.size();
    std::vector<uint64_t> rands; std::vector<uint32_t> stw;
    try { rands.resize(n); stw.resize(n); } catch (...) { return 0; }
    std::vector<int> rcs(G, MBLS_OK); std::vector<std::thread> th;
    std::vector<const void*> d_send(G, nullptr); std::vector<void*> d_recv(G, nullptr); std::vector<hipStream_t> st(G, nullptr);
    std::mutex bm; std::condition_variable cv; uint64_t arrived = 0; int go = 0;             // go: 1 = run phase 2, -1 = stop (a bad signature, or a failed shard)
    auto work = [&](uint64_t g) {
        mbls_ctx* c = m->ctx[g];
        const uint64_t lo = shard_lo(n, g, G), hi = shard_lo(n, g + 1, G), cnt = hi - lo;
        mbls_lock lk2(c->mu);
        vm_rng_stage sg(c);
        if (hipSetDevice(c->device) != hipSuccess) rcs[g] = MBLS_ERR_DEVICE;
        if (!rcs[g] && !m->d_rec_all[g] && hipMalloc(&m->d_rec_all[g], G * MBLS_VM_PARTIAL_BYTES) != hipSuccess) { m->d_rec_all[g] = nullptr; rcs[g] = MBLS_ERR_DEVICE; }
        // (offset tables are absolute: phase 1 stages the shard's message bytes from its first offset on)
        if (!rcs[g]) rcs[g] = vm_rng_phase1(c, sg, sigs96 + 96 * lo, apks96 + 96 * lo, moff ? msgs : (msgs ? msgs + (uint64_t)msg_len * lo : nullptr), msg_len,
                                            moff ? moff + lo : nullptr, cnt, MBLS_VM_PARTIAL_BYTES, stw.data() + lo);
        int mine;
        {
            std::unique_lock<std::mutex> ul(bm);
            arrived++; cv.notify_all();
            cv.wait(ul, [&] { return go != 0; });
            mine = go;
        }
        if (mine < 0 || rcs[g]) { if (cnt) vm_rng_drain(c); return; }
        rcs[g] = vm_rng_phase2(c, sg, msg_len, rands.data() + lo, cnt, true);
        const bool synced = hipStreamSynchronize(c->hs_a) == hipSuccess;
        (void)hipStreamSynchronize(c->hs_b); (void)hipStreamSynchronize(c->hs_c);
        c->ws_pending = false;
        if (!rcs[g] && !synced) rcs[g] = MBLS_ERR_DEVICE;
        d_send[g] = sg.dout.p; d_recv[g] = m->d_rec_all[g]; st[g] = c->hs_a;       // the shard's record stays on its device: the exchange step follows
    };
    try { for (uint64_t g = 0; g < G; g++) th.emplace_back(work, g); }
    catch (...) {
        { std::lock_guard<std::mutex> gl(bm); go = -1; } cv.notify_all();
        for (auto& t : th) t.join();
        snprintf(m->err, sizeof(m->err), "cannot start a host thread"); return 0;
    }
    bool proceed = true;
    {
        std::unique_lock<std::mutex> ul(bm);
        cv.wait(ul, [&] { return arrived == G; });
    }
    for (uint64_t g = 0; g < G; g++) if (rcs[g]) proceed = false;
    size_t reached = n;
    if (proceed) {
        for (size_t i = 0; i < n; i++)
            if (stw[i] & (MBLS_ST_BAD_SIG_ENCODING | MBLS_ST_SIG_NOT_IN_G2)) { reached = i; break; }
        if (reached) draw(user, rands.data(), (uint64_t)reached);
        if (reached < n) proceed = false;                                   // :273-275
    }
    { std::lock_guard<std::mutex> gl(bm); go = proceed ? 1 : -1; } cv.notify_all();
    for (auto& t : th) t.join();
    std::fill(rands.begin(), rands.end(), 0);
    for (uint64_t g = 0; g < G; g++)
        if (rcs[g]) { snprintf(m->err, sizeof(m->err), "device %d (shard %llu): %s", m->ctx[g]->device, (unsigned long long)g, m->ctx[g]->err); return 0; }
    if (!proceed) return 0;
    if (multi_allgather(m, d_send, d_recv, MBLS_VM_PARTIAL_BYTES, st)) return 0;
    mbls_ctx* c = m->ctx[0];
    mbls_lock lk0(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return 0;
    sbuf dres(c, 5); uint8_t r = 0;
    if (dres.alloc(8) != hipSuccess) return 0;
    if (mbls_verify_multiple_finish_device(c, m->d_rec_all[0], G, dres.as<uint8_t>(), nullptr, c->hs_a)) { (void)hipStreamSynchronize(c->hs_a); c->ws_pending = false; return 0; }
    if (hipStreamSynchronize(c->hs_a) != hipSuccess) return 0;
    c->ws_pending = false;
    if (dres.down(&r, 1) != hipSuccess) return 0;
    return r;
}
// n x fast_aggregate_verify over the handle's devices with the results as ONE packed accept bitmap that every device ends up holding (north_star: "RCCL over
// xGMI used only to gather the final accept bitmap"): device g verifies the items of bitmap words [g W, (g + 1) W), W = ceil(ceil(n / 64) / G), packs its W words
// on the device, and the words are all-gathered between the devices (multi_allgather). `bitmap` (host, ceil(n / 64) words, optional) receives device 0's copy;
// mbls_multi_device_bitmap(m, g) is device g's copy (G W words, valid until the next call on the handle). Host buffers in, like the other handle entries.
extern "C" int mbls_multi_fast_aggregate_verify_bitmap(mbls_multi* m, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len, const uint64_t* moff,
        const uint8_t* pks, int fmt, const uint32_t* off, uint64_t n, uint32_t k, uint64_t* bitmap, uint32_t* status) {
    if (!m) return MBLS_ERR_ARGUMENT;
    if (n == 0) return MBLS_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    const uint64_t G = m->ctx.size();
    const uint64_t words = (n + 63) / 64, W = (words + G - 1) / G;
    const size_t unit = fmt == MBLS_PK_COMPRESSED ? 48 : 96;
    std::vector<int> rcs(G, MBLS_OK); std::vector<std::thread> th;
    std::vector<const void*> d_send(G, nullptr); std::vector<void*> d_recv(G, nullptr); std::vector<hipStream_t> st(G, nullptr);
    auto work = [&](uint64_t g) {
        mbls_ctx* c = m->ctx[g];
        const uint64_t lo = 64 * W * g < n ? 64 * W * g : n, hi = 64 * W * (g + 1) < n ? 64 * W * (g + 1) : n, cnt = hi - lo;
        mbls_lock lk2(c->mu);
        if (hipSetDevice(c->device) != hipSuccess) { rcs[g] = MBLS_ERR_DEVICE; return; }
        if (m->bm_words_cap < W || !m->d_bm_send[g]) {      // (bm_words_cap is raised after the workers have joined)
            if (m->d_bm_send[g]) { (void)hipFree(m->d_bm_send[g]); m->d_bm_send[g] = nullptr; }
            if (m->d_bm_all[g]) { (void)hipFree(m->d_bm_all[g]); m->d_bm_all[g] = nullptr; }
            if (hipMalloc(&m->d_bm_send[g], 8 * W) != hipSuccess || hipMalloc(&m->d_bm_all[g], 8 * W * G) != hipSuccess) { rcs[g] = MBLS_ERR_DEVICE; return; }
        }
        d_send[g] = m->d_bm_send[g]; d_recv[g] = m->d_bm_all[g]; st[g] = c->hs_a;
        if (hipMemsetAsync(m->d_bm_send[g], 0, 8 * W, c->hs_a) != hipSuccess) { rcs[g] = MBLS_ERR_DEVICE; return; }
        if (!cnt) return;
        const uint8_t* s_msgs = (moff || !msgs) ? msgs : msgs + (uint64_t)msg_len * lo;
        const uint8_t* s_pks = (off || !pks) ? pks : pks + unit * (uint64_t)k * lo;
        std::vector<uint8_t> res;
        try { res.resize(cnt); } catch (...) { rcs[g] = MBLS_ERR_DEVICE; return; }
        rcs[g] = verify_host(c, sigs ? sigs + 96 * lo : nullptr, s_msgs, msg_len, moff ? moff + lo : nullptr, s_pks, fmt, nullptr, nullptr, off ? off + lo : nullptr, cnt, k,
                             MBLS_MODE_FAST_AGGREGATE, res.data(), status ? status + lo : nullptr);
        if (rcs[g]) return;
        // the shard's result bytes are still in the context's staging buffer (slot 4 of verify_host): pack them where they are
        hipLaunchKernelGGL(k_pack, dim3(nblk(cnt)), dim3(WG), 0, c->hs_a, (const uint8_t*)c->stage[4].p, m->d_bm_send[g], cnt);
        if (hipGetLastError() != hipSuccess) rcs[g] = MBLS_ERR_DEVICE;
    };
    try { for (uint64_t g = 1; g < G; g++) th.emplace_back(work, g); } catch (...) { for (auto& t : th) t.join(); snprintf(m->err, sizeof(m->err), "cannot start a host thread"); return MBLS_ERR_DEVICE; }
    work(0);
    for (auto& t : th) t.join();
    for (uint64_t g = 0; g < G; g++)
        if (rcs[g]) { snprintf(m->err, sizeof(m->err), "device %d (shard %llu): %s", m->ctx[g]->device, (unsigned long long)g, m->ctx[g]->err); m->bm_words_cap = 0; return rcs[g]; }
    if (m->bm_words_cap < W) m->bm_words_cap = W;
    int rc = multi_allgather(m, d_send, d_recv, 8 * W, st); if (rc) return rc;
    if (bitmap) {
        if (hipSetDevice(m->ctx[0]->device) != hipSuccess || hipMemcpy(bitmap, m->d_bm_all[0], 8 * words, hipMemcpyDeviceToHost) != hipSuccess) {
            snprintf(m->err, sizeof(m->err), "bitmap download failed"); return MBLS_ERR_DEVICE;
        }
    }
    return MBLS_OK;
}
extern "C" const uint64_t* mbls_multi_device_bitmap(mbls_multi* m, int g) { return (m && g >= 0 && g < (int)m->ctx.size()) ? m->d_bm_all[g] : nullptr; }
extern "C" int mbls_multi_keytable_create(mbls_multi* m, uint64_t capacity_hint, mbls_multi_keytable** out) {
    if (!m || !out) return MBLS_ERR_ARGUMENT;
    mbls_multi_keytable* t = new (std::nothrow) mbls_multi_keytable();
    if (!t) return MBLS_ERR_DEVICE;
    t->m = m;
    for (mbls_ctx* c : m->ctx) {
        mbls_keytable* x = nullptr;
        int rc = mbls_keytable_create(c, capacity_hint, &x);
        if (rc) { for (mbls_keytable* y : t->tab) mbls_keytable_destroy(y); delete t; return rc; }
        t->tab.push_back(x);
    }
    *out = t; return MBLS_OK;
}
extern "C" void mbls_multi_keytable_destroy(mbls_multi_keytable* t) {
    if (!t) return;
    for (mbls_keytable* x : t->tab) mbls_keytable_destroy(x);
    delete t;
}
extern "C" mbls_keytable* mbls_multi_keytable_replica(mbls_multi_keytable* t, int i) { return (t && i >= 0 && i < (int)t->tab.size()) ? t->tab[i] : nullptr; }
extern "C" uint64_t mbls_multi_keytable_size(const mbls_multi_keytable* t) { return (t && !t->tab.empty()) ? mbls_keytable_size(t->tab[0]) : 0; }
// the same keys, decoded on every device side by side (the indices are the same everywhere); errs as mbls_keytable_append
extern "C" int mbls_multi_keytable_append(mbls_multi_keytable* t, const uint8_t* pks, int fmt, int validate, uint64_t n, uint64_t* first_index, uint8_t* errs) {
    if (!t || !t->m) return MBLS_ERR_ARGUMENT;
    mbls_multi* m = t->m;
    std::lock_guard<std::mutex> lk(m->mu);
    const size_t G = t->tab.size();
    std::vector<int> rcs(G, MBLS_OK); std::vector<uint64_t> first(G, 0);
    std::vector<std::vector<uint8_t>> e(G);
    std::vector<std::thread> th;
    // all or nothing: every replica's size as it is now -- whatever fails below (a worker that could not be started, an append that failed after its records
    // were already published: the stream synchronisation or the download of errs), ALL replicas go back to it, so that indices stay equal everywhere
    std::vector<uint64_t> size0(G);
    for (size_t g = 0; g < G; g++) size0[g] = t->tab[g]->size;
    auto rollback = [&]() { for (size_t g = 0; g < G; g++) if (t->tab[g]->size > size0[g]) keytable_truncate(t->tab[g], size0[g]); };
    auto work = [&](size_t g) {
        uint8_t* eg = errs;
        if (g) { try { e[g].resize(n ? n : 1); } catch (...) { rcs[g] = MBLS_ERR_DEVICE; return; } eg = e[g].data(); }
        rcs[g] = mbls_keytable_append(t->tab[g], pks, fmt, validate, n, &first[g], eg);
    };
    try { for (size_t g = 1; g < G; g++) th.emplace_back(work, g); }
    catch (...) { for (auto& x : th) x.join(); rollback(); snprintf(m->err, sizeof(m->err), "could not start a worker thread"); return MBLS_ERR_DEVICE; }
    work(0);
    for (auto& x : th) x.join();
    int bad = MBLS_OK;
    for (size_t g = 0; g < G && !bad; g++) {
        if (rcs[g]) { snprintf(m->err, sizeof(m->err), "device %d: %s", m->ctx[g]->device, m->ctx[g]->err); bad = rcs[g]; }
        else if (first[g] != first[0] || (g && n && memcmp(e[g].data(), errs, n) != 0)) { snprintf(m->err, sizeof(m->err), "replicas of the key table disagree (device %d)", m->ctx[g]->device); bad = MBLS_ERR_DEVICE; }
    }
    if (bad) { rollback(); return bad; }      // (the size is host-side state: the dropped records are simply overwritten by the next append)
    if (first_index) *first_index = first[0];
    return MBLS_OK;
}
extern "C" int mbls_multi_fast_aggregate_verify_batch_indexed(mbls_multi* m, const mbls_multi_keytable* t, const uint8_t* sigs, const uint8_t* msgs, uint32_t msg_len,
        const uint64_t* moff, const uint32_t* idx, const uint32_t* off, uint64_t n, uint32_t k, uint8_t* results, uint32_t* status) {
    if (!t) return MBLS_ERR_ARGUMENT;
    return multi_run(m, t, sigs, msgs, msg_len, moff, nullptr, MBLS_PK_UNCOMPRESSED, idx, off, n, k, MBLS_MODE_FAST_AGGREGATE, results, status);
}
