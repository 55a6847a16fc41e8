// mbls_lanes.h -- the per-lane bodies of the pipeline kernels and the HBM workspace layout between them.
//
// Pipeline for one (sig, msg, pubkey-set) item per lane -- the data-parallel restatement of
// AggregateSignature::fast_aggregate_verify (reference src/aggregates.rs:177-215), Signature::verify
// (src/signature.rs:27-40) and fast_aggregate_verify_pre_aggregated (src/aggregates.rs:223-253):
//   lane_aggregate   decode + sum the item's public keys            (src/aggregates.rs:29-39, :189-198)
//   lane_sig         decode the signature, G2 subgroup check         (src/signature.rs:43-46, aggregates.rs:184)
//   lane_hash        hash_to_curve_g2(msg)                           (src/amcl_utils.rs:33-35)
//   lane_miller      2-pair Miller loop e(sig,-G1) e(H,apk)          (src/amcl_utils.rs:38-39)
//   lane_final       final exponentiation, == 1, fold status         (src/amcl_utils.rs:40-41)
// Rejections are carried as status bits (mbls_curve.h), never as early exits of the whole pipeline.
//
// Workspace (HBM): struct-of-arrays of 32-bit limbs, limb-major: limb j of Fp slot s of item i lives at
// w[(s*12 + j) * stride + i], so every load/store instruction of a wave touches 64 consecutive dwords.
#pragma once
#include "mbls_hash.h"
#include "mbls_pairing.h"

#define MBLS_PK_COMPRESSED 0
#define MBLS_PK_UNCOMPRESSED 1

// verification flavours (which checks the reference function performs)
#define MBLS_MODE_FAST_AGGREGATE 0   // empty-set + apk-infinity checks (src/aggregates.rs:179-198)
#define MBLS_MODE_VERIFY 1           // Signature::verify: one key, no infinity check (src/signature.rs:27-40)

enum {
    MBLS_SLOT_APK = 0,      // 3 Fp: Jacobian X, Y, Z
    MBLS_SLOT_SIG = 3,      // 4 Fp: affine x.c0, x.c1, y.c0, y.c1
    MBLS_SLOT_H = 7,        // 6 Fp: Jacobian
    MBLS_SLOT_F = 13,       // 12 Fp: Miller value
    MBLS_SLOT_COUNT = 25
};
struct mbls_ws { uint32_t* w; uint64_t stride; };

MBLS_FN fp ws_ld(const mbls_ws& ws, int slot, uint64_t i) {
    fp r; const uint32_t* p = ws.w + (uint64_t)slot * 12 * ws.stride + i;
#pragma unroll
    for (int j = 0; j < 12; j++) r[j] = p[(uint64_t)j * ws.stride];
    return r;
}
MBLS_FN void ws_st(const mbls_ws& ws, int slot, uint64_t i, fp v) {
    uint32_t* p = ws.w + (uint64_t)slot * 12 * ws.stride + i;
#pragma unroll
    for (int j = 0; j < 12; j++) p[(uint64_t)j * ws.stride] = v[j];
}
MBLS_FN fp2 ws_ld2(const mbls_ws& ws, int slot, uint64_t i) { fp2 r; r.c0 = ws_ld(ws, slot, i); r.c1 = ws_ld(ws, slot + 1, i); return r; }
MBLS_FN void ws_st2(const mbls_ws& ws, int slot, uint64_t i, const fp2& v) { ws_st(ws, slot, i, v.c0); ws_st(ws, slot + 1, i, v.c1); }

// ---------------------------------------------------------------------------------------------- phases
// Sum of the item's k public keys (wire bytes), starting from infinity like AggregatePublicKey::aggregate.
MBLS_FN void lane_aggregate(const mbls_ws& ws, uint64_t i, const uint8_t* pks, uint32_t k, int fmt, int mode, uint32_t* status) {
    uint32_t st = 0;
    g1j acc; g1_set_inf(&acc);
    if (fmt == MBLS_PK_COMPRESSED) {
        for (uint32_t j = 0; j < k; j++) {
            fp x, y; bool inf;
            int e = g1_decode_compressed(&x, &y, &inf, pks + (uint64_t)48 * j);
            if (e) { st |= MBLS_ST_BAD_PK_ENCODING; inf = true; }
            if (inf) st |= MBLS_ST_PK_INFINITY;
            g1_madd(&acc, &acc, x, y, inf);
        }
    } else {
        // 96-byte keys: dword loads when the buffer allows, and the next key is requested before the current one is used
        // (with one wave per SIMD nothing else hides the memory latency)
        const bool al = (((uintptr_t)pks) & 15u) == 0;
        fp nwx = fp_zero(), nwy = fp_zero();
        if (k) g1_load_words96(&nwx, &nwy, pks, al);
        for (uint32_t j = 0; j < k; j++) {
            fp wx = nwx, wy = nwy;
            if (j + 1 < k) g1_load_words96(&nwx, &nwy, pks + (uint64_t)96 * (j + 1), al);
            fp x, y; bool inf;
            int e = g1_decode_uncompressed_w(&x, &y, &inf, wx, wy);
            if (e) { st |= MBLS_ST_BAD_PK_ENCODING; inf = true; }
            if (inf) st |= MBLS_ST_PK_INFINITY;
            g1_madd_inl(&acc, &acc, x, y, inf);
        }
    }
    if (mode == MBLS_MODE_FAST_AGGREGATE) {
        if (k == 0) st |= MBLS_ST_NO_KEYS;
        if (g1_is_inf(&acc)) st |= MBLS_ST_APK_INFINITY;
    }
    ws_st(ws, MBLS_SLOT_APK, i, acc.x); ws_st(ws, MBLS_SLOT_APK + 1, i, acc.y); ws_st(ws, MBLS_SLOT_APK + 2, i, acc.z);
    *status = st;
}
// The same sum as one generated digit-form routine per key format (tools/gen_tower_d.py, g1_aggregate_d_routine): the loop over the lane's
// keys -- fetch one key ahead, decode, mixed addition with the reference's case handling -- without lane-private memory. `keys`: the
// lane's first 96-byte key (4-byte aligned) or, with a table, its first index. Returns the status bits.
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp_mulpair_d_asm_fn() { asm volatile(MBLS_FP_MULPAIR_D_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp_mul1_d_asm_fn() { asm volatile(MBLS_FP_MUL1_D_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_fp_sqrpair_d_asm_fn() { asm volatile(MBLS_FP_SQRPAIR_D_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g1_aggregate_raw_d_asm_fn() { asm volatile(MBLS_G1_AGGREGATE_RAW_D_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g1_aggregate_indexed_d_asm_fn() { asm volatile(MBLS_G1_AGGREGATE_INDEXED_D_ASM); }
template <bool INDEXED>
MBLS_FN uint32_t lane_aggregate_d(const mbls_ws& ws, uint64_t i, const void* keys, uint32_t cnt, int mode, uint32_t lane,
                                  const uint32_t* recs = nullptr, uint64_t tsize = 0) {
    uint32_t addr = 4u * lane;
    const uint64_t gb = (uint64_t)(uintptr_t)ws.w + 4ull * (i - lane);
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws.stride * 4));
    uint32_t plo = (uint32_t)(uintptr_t)keys, phi = (uint32_t)((uint64_t)(uintptr_t)keys >> 32), c = cnt, fl;
    if (INDEXED) {
        const uint32_t r_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)recs), r_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)(uintptr_t)recs >> 32));
        const uint32_t ts = __builtin_amdgcn_readfirstlane((uint32_t)(tsize > 0xFFFFFFFFull ? 0xFFFFFFFFull : tsize));
        asm volatile(MBLS_ASM_CALL("mbls_g1_aggregate_indexed_d_asm_fn")
                     : "={v251}"(fl), "+{v248}"(plo), "+{v249}"(phi), "+{v250}"(c)
                     : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4), "{s94}"(r_lo), "{s95}"(r_hi), "{s96}"(ts)
                     : MBLS_G1_AGG_D_ASM_CLOBBERS);
    } else {
        asm volatile(MBLS_ASM_CALL("mbls_g1_aggregate_raw_d_asm_fn")
                     : "={v251}"(fl), "+{v248}"(plo), "+{v249}"(phi), "+{v250}"(c)
                     : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                     : MBLS_G1_AGG_D_ASM_CLOBBERS, "s94", "s95", "s96");
    }
    uint32_t st = ((fl & 1u) ? MBLS_ST_PK_INFINITY : 0u) | ((fl & 2u) ? MBLS_ST_BAD_PK_ENCODING : 0u);
    if (mode == MBLS_MODE_FAST_AGGREGATE) {
        if (cnt == 0) st |= MBLS_ST_NO_KEYS;
        if (fl & 4u) st |= MBLS_ST_APK_INFINITY;
    }
    return st;
}
#endif
// Compressed keys are decompressed one key per lane first (one Fp square root each: the dominant cost of the 48-byte
// format, and with n*k lanes the kernel runs at full occupancy), into an array of affine Montgomery coordinates
// (24 dwords per key); the per-item sum then only does the mixed additions.
#define MBLS_KEYFLAG_INF 1u
#define MBLS_KEYFLAG_BAD 2u
MBLS_FN void lane_pk_decompress(uint64_t j, const uint8_t* pks48, uint32_t* keys_xy, uint8_t* flags) {
    fp x, y; bool inf;
    int e = g1_decode_compressed<true>(&x, &y, &inf, pks48 + 48 * j);         // the two-waves-per-SIMD variant of the square root
    uint32_t* o = keys_xy + 24 * j;
#pragma unroll
    for (int t = 0; t < 12; t++) { o[t] = x[t]; o[12 + t] = y[t]; }
    flags[j] = (uint8_t)((e ? MBLS_KEYFLAG_BAD : 0u) | ((inf || e) ? MBLS_KEYFLAG_INF : 0u));
}
MBLS_FN void lane_aggregate_decoded(const mbls_ws& ws, uint64_t i, const uint32_t* keys_xy, const uint8_t* flags, uint32_t k, int mode, uint32_t* status) {
    uint32_t st = 0;
    g1j acc; g1_set_inf(&acc);
    for (uint32_t j = 0; j < k; j++) {
        const uint32_t* o = keys_xy + 24 * (uint64_t)j;
        fp x, y;
#pragma unroll
        for (int t = 0; t < 12; t++) { x[t] = o[t]; y[t] = o[12 + t]; }
        uint32_t f = flags[j];
        if (f & MBLS_KEYFLAG_BAD) st |= MBLS_ST_BAD_PK_ENCODING;
        if (f & MBLS_KEYFLAG_INF) st |= MBLS_ST_PK_INFINITY;
        g1_madd(&acc, &acc, x, y, (f & MBLS_KEYFLAG_INF) != 0);
    }
    if (mode == MBLS_MODE_FAST_AGGREGATE) {
        if (k == 0) st |= MBLS_ST_NO_KEYS;
        if (g1_is_inf(&acc)) st |= MBLS_ST_APK_INFINITY;
    }
    ws_st(ws, MBLS_SLOT_APK, i, acc.x); ws_st(ws, MBLS_SLOT_APK + 1, i, acc.y); ws_st(ws, MBLS_SLOT_APK + 2, i, acc.z);
    *status = st;
}
// Resident key table: one 128-byte record per key = affine Montgomery x (12 dwords), y (12 dwords), flags, 7 dwords of padding
// (a record is one aligned cache line, fetched with six 16-byte loads + one dword). This is what a decoded PublicKey holds in
// memory in the reference (src/keys.rs:116-120): no byte decoding, Montgomery conversion or on-curve check per use.
#define MBLS_KEYREC_DWORDS 32
MBLS_FN void keyrec_load(fp* x, fp* y, uint32_t* flags, const uint32_t* recs, uint64_t tsize, uint32_t id) {
    bool oob = id >= tsize;                                   // an index outside the table counts as an undecodable key
    const uint32_t* q = (const uint32_t*)__builtin_assume_aligned(recs + (uint64_t)MBLS_KEYREC_DWORDS * (oob ? 0 : id), 16);
    fp a, b;
#pragma unroll
    for (int t = 0; t < 12; t++) { a[t] = q[t]; b[t] = q[12 + t]; }
    uint32_t f = q[24];
    *x = a; *y = b; *flags = oob ? (MBLS_KEYFLAG_BAD | MBLS_KEYFLAG_INF) : f;
}
MBLS_FN void lane_aggregate_indexed(const mbls_ws& ws, uint64_t i, const uint32_t* recs, uint64_t tsize, const uint32_t* idx, uint32_t k, int mode, uint32_t* status) {
    uint32_t st = 0;
    g1j acc; g1_set_inf(&acc);
    // two-deep software pipeline (index two keys ahead, record one key ahead): one wave per SIMD has nothing else to hide the
    // gather latency behind
    fp nx = fp_zero(), ny = fp_zero(); uint32_t nf = 0;
    uint32_t id1 = k > 1 ? idx[1] : 0;
    if (k) keyrec_load(&nx, &ny, &nf, recs, tsize, idx[0]);
    for (uint32_t j = 0; j < k; j++) {
        fp x = nx, y = ny; uint32_t f = nf;
        uint32_t id2 = (j + 2 < k) ? idx[j + 2] : 0;
        if (j + 1 < k) keyrec_load(&nx, &ny, &nf, recs, tsize, id1);
        id1 = id2;
        if (f & MBLS_KEYFLAG_BAD) st |= MBLS_ST_BAD_PK_ENCODING;
        if (f & MBLS_KEYFLAG_INF) st |= MBLS_ST_PK_INFINITY;
        g1_madd_inl(&acc, &acc, x, y, (f & MBLS_KEYFLAG_INF) != 0);
    }
    if (mode == MBLS_MODE_FAST_AGGREGATE) {
        if (k == 0) st |= MBLS_ST_NO_KEYS;
        if (g1_is_inf(&acc)) st |= MBLS_ST_APK_INFINITY;
    }
    ws_st(ws, MBLS_SLOT_APK, i, acc.x); ws_st(ws, MBLS_SLOT_APK + 1, i, acc.y); ws_st(ws, MBLS_SLOT_APK + 2, i, acc.z);
    *status = st;
}
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
// The group arithmetic of the two phases below as generated routines (tools/gen_tower_d.py, g2_group_routine): operands and results in
// the workspace, the running point in AGPRs, `spill` = 11 x 14 dwords per lane of LDS, no lane-private memory.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g2_subgroup_d_asm_fn() { asm volatile(MBLS_G2_SUBGROUP_D_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g2_hash_tail_d_asm_fn() { asm volatile(MBLS_G2_HASH_TAIL_D_ASM); }
// two lanes per message (g2_group_routine("hash", two_lane=True); kernel k_hash2): every lane has a workspace item of its own, lanes 2 j and
// 2 j + 1 are one message; H comes back in the EVEN lane's item
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g2_hash_tail2_d_asm_fn() { asm volatile(MBLS_G2_HASH_TAIL2_D_ASM); }
MBLS_FN void g2_hash2_d_call(const mbls_ws& ws, uint64_t t, MBLS_LDS uint32_t* spill, uint32_t lane) {
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + lane);
    const uint64_t gb = (uint64_t)(uintptr_t)ws.w + 4ull * (t - lane) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws.stride * 4));
    uint32_t fl = 0;
    asm volatile(MBLS_ASM_CALL("mbls_g2_hash_tail2_d_asm_fn") : "+{v251}"(fl) : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4) : MBLS_G2_GROUP_D_ASM_CLOBBERS);
}
template <bool HASH>
MBLS_FN uint32_t g2_group_d_call(const mbls_ws& ws, uint64_t i, MBLS_LDS uint32_t* spill, uint32_t lane) {
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + lane);
    const uint64_t gb = (uint64_t)(uintptr_t)ws.w + 4ull * (i - lane) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws.stride * 4));
    uint32_t fl = 0;
    if (HASH)
        asm volatile(MBLS_ASM_CALL("mbls_g2_hash_tail_d_asm_fn") : "+{v251}"(fl) : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4) : MBLS_G2_GROUP_D_ASM_CLOBBERS);
    else
        asm volatile(MBLS_ASM_CALL("mbls_g2_subgroup_d_asm_fn") : "+{v251}"(fl) : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4) : MBLS_G2_GROUP_D_ASM_CLOBBERS);
    return fl;
}
// One level of the n-pairing paths' trees with one lane per product (tools/gen_tower_d.py f12_tree_routine / g2_tree_routine): the lane's
// Miller value (slots 13..24) times / its G2 sum (slots 25..30) plus the one of the item `half` further on.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_f12_tree_d_asm_fn() { asm volatile(MBLS_F12_TREE_D_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g2_tree_d_asm_fn() { asm volatile(MBLS_G2_TREE_D_ASM); }
template <bool G2>
MBLS_FN void tree_level_d_call(const mbls_ws& ws, uint64_t i, uint64_t half, MBLS_LDS uint32_t* spill, uint32_t lane) {
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + lane);
    const uint64_t gb = (uint64_t)(uintptr_t)ws.w + 4ull * (i - lane) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws.stride * 4));
    const uint32_t poff = __builtin_amdgcn_readfirstlane((uint32_t)(half * 4));          // the partner's byte offset (half < 2^30 items)
    if (G2) {
        asm volatile(MBLS_ASM_CALL("mbls_g2_tree_d_asm_fn") : : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4), "{s71}"(poff) : MBLS_G2_TREE_D_ASM_CLOBBERS);
    } else {
        fp f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11;
        asm volatile(MBLS_ASM_CALL("mbls_f12_tree_d_asm_fn")
                     : MBLS_MILLER_D_OUT_REGS(f)
                     : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4), "{s71}"(poff)
                     : MBLS_F12_TREE_D_ASM_CLOBBERS);
        const fp c[12] = {f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11};
#pragma unroll
        for (int t = 0; t < 12; t++) ws_st(ws, MBLS_SLOT_F + t, i, c[t]);
    }
}
// [r] P for the Jacobian G1 point in slots 0..2 (verify_multiple's g1mul(apk_i, r_i), reference src/aggregates.rs:293), back into slots 0..2:
// the generated windowed routine (tools/gen_tower_d.py g1_blind_routine)
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g1_blind_d_asm_fn() { asm volatile(MBLS_G1_BLIND_D_ASM); }
MBLS_FN void g1_blind_d_call(const mbls_ws& ws, uint64_t i, uint32_t lane, uint64_t r) {
    const uint32_t addr = 4u * lane;
    const uint64_t gb = (uint64_t)(uintptr_t)ws.w + 4ull * (i - lane);
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws.stride * 4));
    uint32_t rlo = (uint32_t)r, rhi = (uint32_t)(r >> 32);
    asm volatile(MBLS_ASM_CALL("mbls_g1_blind_d_asm_fn") : "+{v248}"(rlo), "+{v249}"(rhi) : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                 : MBLS_G1_BLIND_D_ASM_CLOBBERS);
}
// verify_multiple's signature phase (reference src/aggregates.rs:274-276, :303) as one generated routine (tools/gen_tower_d.py g2_blind_routine): the
// subgroup test of the signature in slots 3..6, then [r] sig by signed 4-bit windows into slots 25..30. Returns bit 0 = psi(P) = [x]P.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g2_blind_d_asm_fn() { asm volatile(MBLS_G2_BLIND_D_ASM); }
// the same routine with CONSTANT-TIME table access (tools/gen_tower_d.py blind_scan_ct): every window reads all eight records of the lane's table and keeps its
// own by selection -- what signing uses, where the scalar is a secret key (amcl's g2mul selects in constant time, reference src/signature.rs:17-21)
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g2_blind_ct_d_asm_fn() { asm volatile(MBLS_G2_BLIND_CT_D_ASM); }
MBLS_FN uint32_t g2_blind_ct_d_call(const mbls_ws& ws, uint64_t i, MBLS_LDS uint32_t* spill, uint32_t lane, uint64_t r, uint32_t skip_test = 0) {
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + lane);
    const uint64_t gb = (uint64_t)(uintptr_t)ws.w + 4ull * (i - lane) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws.stride * 4));
    uint32_t fl = skip_test, rlo = (uint32_t)r, rhi = (uint32_t)(r >> 32);
    asm volatile(MBLS_ASM_CALL("mbls_g2_blind_ct_d_asm_fn") : "+{v251}"(fl), "+{v248}"(rlo), "+{v249}"(rhi) : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                 : MBLS_G2_BLIND_D_ASM_CLOBBERS);
    return fl;
}
MBLS_FN uint32_t g2_blind_d_call(const mbls_ws& ws, uint64_t i, MBLS_LDS uint32_t* spill, uint32_t lane, uint64_t r, uint32_t skip_test = 0) {
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + lane);
    const uint64_t gb = (uint64_t)(uintptr_t)ws.w + 4ull * (i - lane) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws.stride * 4));
    uint32_t fl = skip_test, rlo = (uint32_t)r, rhi = (uint32_t)(r >> 32);      // skip_test (the same in every lane): the point is known to be in G2
    asm volatile(MBLS_ASM_CALL("mbls_g2_blind_d_asm_fn") : "+{v251}"(fl), "+{v248}"(rlo), "+{v249}"(rhi) : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                 : MBLS_G2_BLIND_D_ASM_CLOBBERS);
    return fl;
}
// TWO LANES PER SIGNATURE (small batches of verify_multiple, where the signature chain is the critical path: k_sig2, k_blind_sig2_d): lanes 2 j and 2 j + 1
// come with the same workspace item, the same LDS column and the same scalar, walk the routine on identical values and take the independent products of
// every doubling / addition in pairs (g2_group_routine("sig", two_lane=True), g2_blind_routine(two_lane=True)). `lane` = threadIdx.x, `i` = THIS lane's item.
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g2_subgroup2_d_asm_fn() { asm volatile(MBLS_G2_SUBGROUP2_D_ASM); }
extern "C" __device__ __attribute__((noinline, used, aligned(64))) void mbls_g2_blind2_d_asm_fn() { asm volatile(MBLS_G2_BLIND2_D_ASM); }
MBLS_FN uint32_t g2_subgroup2_d_call(const mbls_ws& ws, uint64_t i, MBLS_LDS uint32_t* spill, uint32_t lane) {
    const uint32_t col = lane >> 1;
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + col);
    const uint64_t gb = (uint64_t)(uintptr_t)ws.w + 4ull * (i - col) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws.stride * 4));
    uint32_t fl = 0;
    asm volatile(MBLS_ASM_CALL("mbls_g2_subgroup2_d_asm_fn") : "+{v251}"(fl) : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4) : MBLS_G2_GROUP_D_ASM_CLOBBERS);
    return fl;
}
MBLS_FN uint32_t g2_blind2_d_call(const mbls_ws& ws, uint64_t i, MBLS_LDS uint32_t* spill, uint32_t lane, uint64_t r, uint32_t skip_test = 0) {
    const uint32_t col = lane >> 1;
    const uint32_t addr = (uint32_t)(uintptr_t)(spill + col);
    const uint64_t gb = (uint64_t)(uintptr_t)ws.w + 4ull * (i - col) - (uint64_t)(uint32_t)(uintptr_t)spill;
    const uint32_t gb_lo = __builtin_amdgcn_readfirstlane((uint32_t)gb), gb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32));
    const uint32_t st4 = __builtin_amdgcn_readfirstlane((uint32_t)(ws.stride * 4));
    uint32_t fl = skip_test, rlo = (uint32_t)r, rhi = (uint32_t)(r >> 32);
    asm volatile(MBLS_ASM_CALL("mbls_g2_blind2_d_asm_fn") : "+{v251}"(fl), "+{v248}"(rlo), "+{v249}"(rhi) : "{v252}"(addr), "{s68}"(gb_lo), "{s69}"(gb_hi), "{s70}"(st4)
                 : MBLS_G2_BLIND_D_ASM_CLOBBERS);
    return fl;
}
#endif
// hash_to_field: u0 -> workspace slots 31, 32, u1 -> 37, 38 (2^384 domain); the generated routine takes them from there. A function of
// its own: its message schedule and digest arrays then stay out of the kernel's frame.
// swap (k_hash2's odd lanes): u1 into the slots of u0 and vice versa -- the lane's map_to_curve body reads the slots of u0.
MBLS_NOINLINE void hash_fields_to_ws(uint32_t* w, uint64_t stride, uint64_t i, const uint8_t* msg, uint32_t mlen, uint32_t swap = 0) {
    mbls_ws ws; ws.w = w; ws.stride = stride;
    const mbls_u32x8 b0 = expand_xmd_b0(msg, mlen, MBLS_DST_POP, MBLS_DST_POP_LEN);
    mbls_u32x8 prev = 0;
    for (uint32_t k = 0; k < 4; k++) {               // four field elements of 64 bytes each = two blocks each; digests stay in registers
        mbls_u32x8 hi = expand_xmd_block(b0, prev, 2 * k + 1, MBLS_DST_POP, MBLS_DST_POP_LEN);
        mbls_u32x8 lo = expand_xmd_block(b0, hi, 2 * k + 2, MBLS_DST_POP, MBLS_DST_POP_LEN);
        prev = lo;
        ws_st(ws, (int)(31 + (k & 1) + 6 * ((k >> 1) ^ swap)), i, fp_from_two_digests_v(hi, lo));
    }
}
// spill != nullptr (with use_lds): the subgroup test runs as the generated routine on the coordinates just stored
// check = false: decode only -- the Miller loop that follows decides the subgroup test from its own running point (lane_miller)
MBLS_FN void lane_sig(const mbls_ws& ws, uint64_t i, const uint8_t* sig96, uint32_t* status, MBLS_LDS uint32_t* spill = nullptr, uint32_t lane = 0, bool use_lds = false,
                      bool check = true) {
    fp2 x, y; bool inf; uint32_t st = 0;
    // in the kernel (use_lds) the decoder is inlined: its operands then never have an address and stay out of lane-private memory
    int e = use_lds ? g2_decode_compressed_t<true>(&x, &y, &inf, sig96) : g2_decode_compressed(&x, &y, &inf, sig96);
    if (e) { st |= MBLS_ST_BAD_SIG_ENCODING; inf = true; }
    // infinity is stored as y = 0 (no curve point has y = 0: there is no 2-torsion)
    if (inf) { x = fp2_zero(); y = fp2_zero(); }
    ws_st2(ws, MBLS_SLOT_SIG, i, x); ws_st2(ws, MBLS_SLOT_SIG + 2, i, y);
    if (!check) { *status |= st; return; }
    bool in_g2;
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
    if (use_lds) in_g2 = (g2_group_d_call<false>(ws, i, spill, lane) & 1u) | inf;       // infinity passes (psi(O) = [x]O)
    else
#endif
    {
        g2j p; p.x = x; p.y = y; p.z = fp2_one();
        if (inf) g2_set_inf(&p);
        in_g2 = g2_in_subgroup(&p);
    }
    if (!in_g2) st |= MBLS_ST_SIG_NOT_IN_G2;
    *status |= st;
}
MBLS_FN void lane_hash(const mbls_ws& ws, uint64_t i, const uint8_t* msg, uint32_t mlen, MBLS_LDS uint32_t* spill = nullptr, uint32_t lane = 0, bool use_lds = false) {
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
    if (use_lds) {          // hash_to_field here; both map_to_curve evaluations, q0 + q1 and the cofactor clearing as the generated routine
        hash_fields_to_ws(ws.w, ws.stride, i, msg, mlen);
        g2_group_d_call<true>(ws, i, spill, lane);
        return;
    }
#endif
    g2j h; hash_to_g2(&h, msg, mlen, MBLS_DST_POP, MBLS_DST_POP_LEN);
    ws_st2(ws, MBLS_SLOT_H, i, h.x); ws_st2(ws, MBLS_SLOT_H + 2, i, h.y); ws_st2(ws, MBLS_SLOT_H + 4, i, h.z);
}
// use_lds: the kernel provides an LDS home for the running points (an explicit flag: a __shared__ array may sit at LDS
// address 0, so the pointer itself cannot say whether it is there)
MBLS_FN void lane_miller(const mbls_ws& ws, uint64_t i, MBLS_LDS uint32_t* tstore = nullptr, uint32_t lane = 0, bool use_lds = false) {
    mbls_pair pr[2];
    // pair 0: (sig, -G1)
    fp2 sx = ws_ld2(ws, MBLS_SLOT_SIG, i), sy = ws_ld2(ws, MBLS_SLOT_SIG + 2, i);
    pr[0].skip = fp2_is_zero(sy);
    g2h_from_affine(&pr[0].q, sx, sy); pr[0].t = pr[0].q;
    g1arg_from_affine(&pr[0].p, fp_load_const(MBLS_G1_X), fp_load_const(MBLS_G1_NEG_Y));
    // pair 1: (H(msg), apk)
    g2j h; h.x = ws_ld2(ws, MBLS_SLOT_H, i); h.y = ws_ld2(ws, MBLS_SLOT_H + 2, i); h.z = ws_ld2(ws, MBLS_SLOT_H + 4, i);
    g1j a; a.x = ws_ld(ws, MBLS_SLOT_APK, i); a.y = ws_ld(ws, MBLS_SLOT_APK + 1, i); a.z = ws_ld(ws, MBLS_SLOT_APK + 2, i);
    pr[1].skip = g2_is_inf(&h) | g1_is_inf(&a);
    g2h_from_jacobian(&pr[1].q, &h); pr[1].t = pr[1].q;
    g1arg_from_jacobian(&pr[1].p, &a);
    fp12 f;
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
    if (use_lds) miller_loop_verify_d(&f, pr, ws.w, ws.stride, i, tstore, lane); else miller_loop(&f, pr, 2);
#else
    if (use_lds) miller_loop_n<2, true>(&f, pr, tstore, lane, true); else miller_loop(&f, pr, 2);
#endif
    const fp2* c = &f.c0.c0;
    for (int s = 0; s < 6; s++) ws_st2(ws, MBLS_SLOT_F + 2 * s, i, c[s]);
}
// The subgroup test of the signature out of the Miller loop (generated two-pair loop only; runs between k_miller and k_final). Pair 0's running
// point starts at the signature and walks the bits of |x| -- the same doublings and additions as the ladder of psi(P) = [x]P --, so when the
// loop ends it IS [|x|] sig (homogeneous X : Y : Z, workspace slots 31..36): sig is in G2 iff psi(sig) = -T. A signature outside G2 may drive
// the loop's incomplete additions through T = +-sig or T = O (its order then divides a prefix of |x| or a neighbour): every such case ends with
// Z = 0, which is a rejection, and cannot happen for a point of order r. Infinity passes, as in subgroup_check_g2 (reference src/signature.rs:29-31).
// t_item / t_pair: where the loop left the point -- the two-pair loop in pair 0's slots of the item itself; the split form (k_miller_split: the
// signature's pair on a lane of its own, walked by the one-pair routine) in pair 1's slots of item t_item.
MBLS_FN uint32_t lane_sig_verdict(const mbls_ws& ws, uint64_t i, uint64_t t_item, int t_pair) {
    const fp2 qx = ws_ld2(ws, MBLS_SLOT_SIG, i), qy = ws_ld2(ws, MBLS_SLOT_SIG + 2, i);
    // the running point as the loop leaves it: packed words, representatives in (0.5 p, 1.5 p) of the 2^392-domain values -- read as
    // 2^384-domain values they are 2^8 X, 2^8 Y, 2^8 Z: the same projective point
#if MBLS_DEVICE_ASM
    // the generator says where it left the point and in which form (tools/gen_tower_d.py emits these next to the routine)
    static_assert(MBLS_GEN_MILLER_T_DOMAIN_BITS == 392 && MBLS_GEN_MILLER_T_PACKED == 1, "lane_sig_verdict reads packed 2^392-domain words");
    const int T0 = MBLS_GEN_MILLER_T0_SLOT + 6 * t_pair;
#else
    const int T0 = 31 + 6 * t_pair;
#endif
    fp2 X = ws_ld2(ws, T0, t_item), Y = ws_ld2(ws, T0 + 2, t_item), Z = ws_ld2(ws, T0 + 4, t_item);
    X.c0 = fp_reduce_once(X.c0, 0); X.c1 = fp_reduce_once(X.c1, 0); Y.c0 = fp_reduce_once(Y.c0, 0); Y.c1 = fp_reduce_once(Y.c1, 0);
    Z.c0 = fp_reduce_once(Z.c0, 0); Z.c1 = fp_reduce_once(Z.c1, 0);
    const fp2 px = fp2_mul(fp2_conj(qx), fp2_load_const(MBLS_PSI_CX)), py = fp2_mul(fp2_conj(qy), fp2_load_const(MBLS_PSI_CY));     // psi(sig), affine (g2_psi)
    const bool same = !fp2_is_zero(Z) & fp2_eq(fp2_mul(px, Z), X) & fp2_eq(fp2_mul(py, Z), fp2_neg(Y));      // psi(sig) = [x] sig = -[|x|] sig
    return (!same & !fp2_is_zero(qy)) ? MBLS_ST_SIG_NOT_IN_G2 : 0u;
}
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
// two lanes per item (k_final2): both lanes come back with the same value; `result` / `status` as lane_final
MBLS_FN void lane_final2(const mbls_ws& ws, uint64_t i, uint32_t* status, uint8_t* result, MBLS_LDS uint32_t* ls, uint32_t lane) {
    fp12 f;
    final_exp_ws_d2(&f, ws.w, ws.stride, i, ls, lane);
    uint32_t st = *status;
    if (!fp12_is_one(&f)) st |= MBLS_ST_PAIRING_FAILED;
    *status = st;
    const uint32_t reject = MBLS_ST_BAD_SIG_ENCODING | MBLS_ST_SIG_NOT_IN_G2 | MBLS_ST_BAD_PK_ENCODING |
                            MBLS_ST_APK_INFINITY | MBLS_ST_NO_KEYS | MBLS_ST_PAIRING_FAILED | MBLS_ST_BAD_MSG_RANGE;
    *result = (st & reject) ? 0 : 1;
}
#endif
MBLS_FN void lane_final(const mbls_ws& ws, uint64_t i, uint32_t* status, uint8_t* result, MBLS_LDS uint32_t* ls = nullptr, uint32_t lane = 0, bool use_lds = false) {
    fp12 f; fp2* c = &f.c0.c0;
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
    if (use_lds) final_exp_ws_d(&f, ws.w, ws.stride, i, ls, lane);       // the generated routine reads f from the workspace itself
    else
#endif
    {
        for (int s = 0; s < 6; s++) c[s] = ws_ld2(ws, MBLS_SLOT_F + 2 * s, i);
        final_exp(&f, &f);
    }
    uint32_t st = *status;
    if (!fp12_is_one(&f)) st |= MBLS_ST_PAIRING_FAILED;
    *status = st;
    const uint32_t reject = MBLS_ST_BAD_SIG_ENCODING | MBLS_ST_SIG_NOT_IN_G2 | MBLS_ST_BAD_PK_ENCODING |
                            MBLS_ST_APK_INFINITY | MBLS_ST_NO_KEYS | MBLS_ST_PAIRING_FAILED | MBLS_ST_BAD_MSG_RANGE;
    *result = (st & reject) ? 0 : 1;
}
