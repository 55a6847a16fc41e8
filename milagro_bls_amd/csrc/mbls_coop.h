// mbls_coop.h -- the WAVE-COOPERATIVE engine: one item per wave, its Fp values in LDS slots shared by the 64 lanes, the computation a
// sequence of steps in which every lane performs one Fp operation of its own (tools/gen_coop.py is the compiler of the microprograms,
// tools/coop_sim.py the digit-exact CPU model of this kernel).
//
// Why: the pipeline kernels give every item ONE LANE -- the right shape for throughput (2^16 items fill the chip), the wrong one
// when a single value is on the critical path: one lane walks 4 M dependent instructions through a final exponentiation (8 ms), 14 M
// through a whole verification (20 ms), however small the batch. Here the 18 Fp products of a cyclotomic squaring, or the 54 of an
// Fp12 product, sit side by side in ONE step of ~600 instructions. Used for
//   * the tail of verify_multiple_aggregate_signatures / aggregate_verify (reference src/aggregates.rs:307-315, :158-169): the Miller loop of
//     (sum r_i sig_i, -G1), its product with the sets' Miller values, ONE final exponentiation -- program `vmtail`;
//   * the pairing check of small batches (Signature::verify / fast_aggregate_verify with n <= MBLS_COOP_MAX_ITEMS items: reference
//     src/signature.rs:27-40, src/aggregates.rs:177-215), one wave per item -- program `pairing2`;
//   * the last levels of the n-pairing paths' product trees -- program `f12mul`.
// Values are in the digit form of tools/gen_fpd_asm.py (14 signed 28-bit digits, Montgomery radix 2^392); products are the same
// generated scan the pipeline's routines use (mbls_fp_mul1_d_asm_fn's body, inlined). No MFMA (carry-chain integer work), no
// lane-private memory; LDS is the shared register file of the wave.
#pragma once
#include "mbls_lanes.h"
#include "mbls_coop_prog.inc"

#define COOP_SW 16                       // dwords per slot: 14 digits + 2 (16-byte aligned: a slot moves with four 128-bit LDS accesses)
#define COOP_K_END 0
#define COOP_K_MUL 1
#define COOP_K_LIN 2
#define COOP_K_INV 3
#define COOP_K_ISZ 4
#define COOP_K_FLG 5
#define COOP_K_LOADW 6
#define COOP_K_STOREW 7
#define COOP_K_RES 8
#define COOP_K_POW 9                     // S[dst] <- S[src]^((p-3)/4) (the fixed-exponent routine; only in the k_coop<true> instance)
#define COOP_K_SGN 10                    // flag <- sgn0 of the Fp2 value (S[a], S[b]) held as plain integers
#define COOP_RES_ITEM 0                  // RES: fold the pairing bit into status[item], results[item] like lane_final
#define COOP_RES_BATCH 1                 // RES: one bool for the whole batch: the pairing bit and no rejecting bit in the OR of all status words

// lpi: lanes of a step one item uses (64, 32, 16: the wave serves 64 / lpi items side by side); nslots: LDS slots of one item
struct coop_prog { const uint32_t* steps; const uint32_t* rows; const uint32_t* consts; uint32_t nconsts; uint32_t lpi; uint32_t nslots; };
#define COOP_LDS_BYTES(pg) ((pg).lpi == 64 ? (size_t)0 : (size_t)(64 / (pg).lpi) * (((pg).nslots + 1) * COOP_SW * 4 + 64 * 4))      /* dynamic part */
typedef int32_t coop_v16 __attribute__((ext_vector_type(16)));

#define COOP_M28 0x0fffffff
#define COOP_REJECT (MBLS_ST_BAD_SIG_ENCODING | MBLS_ST_SIG_NOT_IN_G2 | MBLS_ST_BAD_PK_ENCODING | MBLS_ST_APK_INFINITY | MBLS_ST_NO_KEYS | MBLS_ST_PAIRING_FAILED | MBLS_ST_BAD_MSG_RANGE)
#define COOP_REJECT_BATCH (MBLS_ST_BAD_SIG_ENCODING | MBLS_ST_SIG_NOT_IN_G2 | MBLS_ST_BAD_PK_ENCODING | MBLS_ST_BAD_MSG_RANGE | MBLS_ST_BAD_SCALAR)

#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
MBLS_CONST int32_t COOP_P28[14] = MBLS_COOP_P28;

// a * b / 2^392 on signed digits: the generated product scan (tools/gen_fpd_asm.py fp_mul1_d_body), inlined. Operands in v0..v13 and
// v28..v41 (preserved), result in v70..v83, accumulator v98..v99; the digits of p and -1/p live in the SGPRs the scan expects.
MBLS_FN void coop_mul(int32_t* r, const int32_t* a, const int32_t* b) {
    coop_v16 va, vb, vr;
#pragma unroll
    for (int j = 0; j < 14; j++) { va[j] = a[j]; vb[j] = b[j]; }
    va[14] = 0; va[15] = 0; vb[14] = 0; vb[15] = 0;
    asm volatile(MBLS_FP_MUL1_D_ASM
                 : "={v[70:85]}"(vr)
                 : "{v[0:15]}"(va), "{v[28:43]}"(vb),
                   "{s40}"(0xfffaaab), "{s41}"(0xfefffff), "{s42}"(0x3ffffb9), "{s43}"(0xfffeb15), "{s44}"(0x6241eab), "{s45}"(0xa0f6b0f), "{s46}"(0xf6730d2),
                   "{s47}"(0xf38512b), "{s56}"(0x4774b84), "{s57}"(0x4bacd76), "{s58}"(0xba7b643), "{s59}"(0xe69a4b1), "{s60}"(0x1ea397f), "{s61}"(0x1a011),
                   "{s64}"(MBLS_COOP_NP28), "{s65}"(COOP_M28)
                 : "v98", "v99", "vcc", "scc");
#pragma unroll
    for (int j = 0; j < 14; j++) r[j] = vr[j];
}
// 64-bit digit sums -> the representative nearest to zero, digits 0..12 in [0, 2^28): carry pass, quotient estimate from the true top
// digit, subtraction pass (tools/coop_sim.py reduce_digits is the same sequence)
MBLS_FN void coop_reduce(int32_t* out, const int64_t* s) {
    int64_t acc = 0; int32_t n[13];
#pragma unroll
    for (int j = 0; j < 13; j++) { acc += s[j]; n[j] = (int32_t)(acc & COOP_M28); acc >>= 28; }
    acc += s[13];
    const int64_t q = (int64_t)__builtin_rint((double)acc * MBLS_COOP_RECIP_PTOP);
    int64_t c = 0;
#pragma unroll
    for (int j = 0; j < 13; j++) { c += (int64_t)n[j] - q * COOP_P28[j]; out[j] = (int32_t)(c & COOP_M28); c >>= 28; }
    c += acc - q * COOP_P28[13];
    out[13] = (int32_t)c;
}
// a reduced value -> [0, p): add p when it is negative
MBLS_FN void coop_canonical(int32_t* d) {
    const int64_t nq = d[13] < 0 ? 1 : 0;
    int64_t c = 0;
#pragma unroll
    for (int j = 0; j < 13; j++) { c += (int64_t)d[j] + nq * COOP_P28[j]; d[j] = (int32_t)(c & COOP_M28); c >>= 28; }
    c += (int64_t)d[13] + nq * COOP_P28[13];
    d[13] = (int32_t)c;
}
// canonical digits (a value below 2^384) -> 12 words
MBLS_FN fp coop_to_words(const int32_t* d) {
    fp w;
#pragma unroll
    for (int q = 0; q < 12; q++) {
        const int j = (32 * q) / 28, off = (32 * q) % 28;
        uint32_t v = (uint32_t)d[j] >> off;
        v |= (uint32_t)d[j + 1] << (28 - off);
        if (28 - off + 28 < 32 && j + 2 < 14) v |= (uint32_t)d[j + 2] << (56 - off);
        w[q] = v;
    }
    return w;
}
// 12 words w -> the 14 digits of w * 2^8 (a 2^384-domain value enters the 2^392 domain), then reduced
MBLS_FN void coop_from_words(int32_t* out, fp w) {
    int64_t s[14];
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const int o = 28 * j - 8;
        uint32_t v;
        if (o < 0) v = w[0] << 8;
        else {
            const int q = o >> 5, r = o & 31;
            v = w[q] >> r;
            if (r > 4 && q + 1 < 12) v |= w[q + 1] << (32 - r);
        }
        s[j] = (int64_t)(v & COOP_M28);
    }
    coop_reduce(out, s);
}

#endif

// One wave per workgroup; workgroup b runs the program on item first_item + b * item_step (workspace addressing of mbls_lanes.h).
// partner_step: LOADW with bit 16 of its workspace slot set reads from item + partner_step instead (the other operand of a tree product).
// POW: the instance that also knows the fixed-exponent step (its routine keeps a window table in AGPRs: the plain instance stays small
// enough for two waves per SIMD).
// LPI: lanes of a step one item uses -- 64: one item per wave, its slots in a static allocation (the latency path); 32 / 16: two / four
// items per wave side by side, each with its own slots and flag words in the dynamic allocation (COOP_LDS_BYTES).
#ifdef MBLS_COOP_PROFILE
// dev build (-DMBLS_COOP_PROFILE, scripts/dbg/coop_prof.py): clocks per step kind and step part, summed by wave 0 of every launch: [kind][0] steps, [1] operands in
// registers, [2] computed, [3] stored and past the barrier
__device__ unsigned long long mbls_coop_prof[16 * 4];
#define COOP_PROF_T(v) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); v = __builtin_readcyclecounter(); } while (0)
#else
#define COOP_PROF_T(v) do { } while (0)
#endif
template <bool POW, int LPI>
__global__ void __launch_bounds__(64) k_coop_t(coop_prog pg, mbls_ws ws, uint64_t first_item, uint64_t item_step, uint64_t partner_step, uint64_t n_items,
                                             uint32_t* status, uint8_t* results, int res_mode) {
#if MBLS_DEVICE_ASM && !defined(MBLS_NO_FP2_ASM)
    extern __shared__ __attribute__((aligned(16))) int32_t coop_lds[];
    __shared__ __attribute__((aligned(16))) int32_t S1[LPI == 64 ? MBLS_COOP_MAX_SLOTS * COOP_SW : 16];
    __shared__ uint32_t flags1[64];
    const uint32_t lane = threadIdx.x;
    constexpr uint32_t ipw = 64u / LPI;
    const uint32_t sub = lane / LPI, ll = lane & (LPI - 1u);
    const uint32_t slot_words = LPI == 64 ? MBLS_COOP_MAX_SLOTS * COOP_SW : (pg.nslots + 1u) * COOP_SW;
    int32_t* const S = LPI == 64 ? S1 : coop_lds + sub * slot_words;         // this lane's item
    uint32_t* const flags = LPI == 64 ? flags1 : (uint32_t*)(coop_lds + ipw * slot_words) + 64u * sub;
    const uint64_t index = (uint64_t)blockIdx.x * ipw + sub;                   // the items beyond n_items (last wave) compute on item 0 and write nothing
    const bool valid = LPI == 64 || index < n_items;
    const uint64_t item = first_item + (valid ? index : 0) * item_step;
    if (LPI == 64) {
        for (uint32_t t = lane; t < MBLS_COOP_MAX_SLOTS * COOP_SW; t += 64) S1[t] = 0;
        flags1[lane] = lane == 1 ? 1u : 0u;
    } else {
        for (uint32_t t = lane; t < ipw * slot_words; t += 64) coop_lds[t] = 0;
        for (uint32_t t = lane; t < ipw * 64u; t += 64) ((uint32_t*)(coop_lds + ipw * slot_words))[t] = (t & 63u) == 1u ? 1u : 0u;
    }
    __syncthreads();
    for (uint32_t t = ll; t < pg.nconsts * 14; t += LPI) S[pg.consts[15 * (t / 14)] * COOP_SW + (t % 14)] = (int32_t)pg.consts[15 * (t / 14) + 1 + (t % 14)];
    __syncthreads();
    const uint4* rows = (const uint4*)pg.rows;
    uint32_t info = pg.steps[0], row = pg.steps[1];
    uint4 m0 = rows[(uint64_t)row * 128 + 2 * ll], m1 = rows[(uint64_t)row * 128 + 2 * ll + 1];
    for (uint32_t step = 0;; step++) {
        const uint32_t kind = info & 0xFF, na = (info >> 8) & 0xF, nb = (info >> 12) & 0xF;
        if (kind == COOP_K_END) break;
#ifdef MBLS_COOP_PROFILE
        unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0; COOP_PROF_T(pt0); pt1 = pt2 = pt0;
#endif
        const uint4 c0 = m0, c1 = m1;
        // the next step's microcode is requested before this step runs (the rows are shared by every wave: L2 hits)
        info = pg.steps[2 * (step + 1)]; row = pg.steps[2 * (step + 1) + 1];
        m0 = rows[(uint64_t)row * 128 + 2 * ll]; m1 = rows[(uint64_t)row * 128 + 2 * ll + 1];
        int32_t cf[8]; uint32_t ix[8];
#pragma unroll
        for (int t = 0; t < 4; t++) { cf[t] = (int32_t)(int8_t)(c0.x >> (8 * t)); cf[4 + t] = (int32_t)(int8_t)(c0.y >> (8 * t)); }
        ix[0] = c0.z & 1023; ix[1] = (c0.z >> 10) & 1023; ix[2] = (c0.z >> 20) & 1023;
        ix[3] = c0.w & 1023; ix[4] = (c0.w >> 10) & 1023; ix[5] = (c0.w >> 20) & 1023;
        ix[6] = c1.x & 1023; ix[7] = (c1.x >> 10) & 1023;
        const uint32_t dst = (c1.x >> 20) & 1023;
        const uint32_t fl = c1.y & 0xFF, fl2 = (c1.y >> 8) & 0xFF, fop = (c1.y >> 16) & 0xFF;
        const bool active = (c1.y >> 31) != 0, writes = active && valid;
        const uint32_t wslot = c1.z;
        if (kind == COOP_K_MUL) {
            int32_t a[14], b[14], r[14];
#pragma unroll
            for (int j = 0; j < 14; j++) { a[j] = cf[0] * S[ix[0] * COOP_SW + j]; b[j] = cf[4] * S[ix[4] * COOP_SW + j]; }
            // the term counts are the same for every lane of the step (uniform branches, statically indexed registers)
#pragma unroll
            for (int t = 1; t < 4; t++) {
                if ((uint32_t)t < na) {
#pragma unroll
                    for (int j = 0; j < 14; j++) a[j] += cf[t] * S[ix[t] * COOP_SW + j];
                }
                if ((uint32_t)t < nb) {
#pragma unroll
                    for (int j = 0; j < 14; j++) b[j] += cf[4 + t] * S[ix[4 + t] * COOP_SW + j];
                }
            }
            COOP_PROF_T(pt1);
            coop_mul(r, a, b);
            COOP_PROF_T(pt2);
#pragma unroll
            for (int j = 0; j < 14; j++) S[dst * COOP_SW + j] = r[j];
        } else if (kind == COOP_K_LIN) {
            int64_t s[14]; int32_t r[14];
            const bool second = fop == 1 && flags[fl] != 0;          // a selection: flag ? terms 4..7 : terms 0..3
#pragma unroll
            for (int j = 0; j < 14; j++) s[j] = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if ((uint32_t)t < na) {
                    const int32_t c = (fop == 1 && second) ? 0 : cf[t];
#pragma unroll
                    for (int j = 0; j < 14; j++) s[j] += (int64_t)c * S[ix[t] * COOP_SW + j];
                }
                if ((uint32_t)t < nb) {
                    const int32_t c = (fop == 1 && !second) ? 0 : cf[4 + t];
#pragma unroll
                    for (int j = 0; j < 14; j++) s[j] += (int64_t)c * S[ix[4 + t] * COOP_SW + j];
                }
            }
            COOP_PROF_T(pt1);
            coop_reduce(r, s);
            COOP_PROF_T(pt2);
#pragma unroll
            for (int j = 0; j < 14; j++) S[dst * COOP_SW + j] = r[j];
        } else if (kind == COOP_K_INV) {
            int32_t d[14], r[14];
#pragma unroll
            for (int j = 0; j < 14; j++) d[j] = S[ix[0] * COOP_SW + j];
            coop_canonical(d);
            fp w = fp_inv(coop_to_words(d));           // the safegcd routine on 12 canonical words of the 2^384 domain (0 -> 0)
            coop_from_words(r, w);
#pragma unroll
            for (int j = 0; j < 14; j++) S[dst * COOP_SW + j] = r[j];
        } else if (POW && kind == COOP_K_POW) {
            int32_t d[14], r[14];
#pragma unroll
            for (int j = 0; j < 14; j++) d[j] = S[ix[0] * COOP_SW + j];
            coop_canonical(d);
            fp w = fp_pow_pm3d4_w4(coop_to_words(d));      // a^((p-3)/4) on 12 canonical words of the 2^384 domain (8-entry window table)
            coop_from_words(r, w);
#pragma unroll
            for (int j = 0; j < 14; j++) S[dst * COOP_SW + j] = r[j];
        } else if (kind == COOP_K_SGN) {
            int32_t d0[14], d1[14];
#pragma unroll
            for (int j = 0; j < 14; j++) { d0[j] = S[ix[0] * COOP_SW + j]; d1[j] = S[ix[1] * COOP_SW + j]; }
            coop_canonical(d0); coop_canonical(d1);
            uint32_t o = 0;
#pragma unroll
            for (int j = 0; j < 14; j++) o |= (uint32_t)d0[j];
            if (active) flags[fl] = ((uint32_t)d0[0] & 1u) | ((o == 0 ? 1u : 0u) & ((uint32_t)d1[0] & 1u));
        } else if (kind == COOP_K_ISZ) {
            uint32_t o = 0;
#pragma unroll
            for (int j = 0; j < 14; j++) o |= (uint32_t)S[ix[0] * COOP_SW + j];
            if (active) flags[fl] = o == 0 ? 1u : 0u;
        } else if (kind == COOP_K_FLG) {
            if (active) {
                const uint32_t x = flags[fl2];
                uint32_t v;
                if (fop == 5) { v = 1; for (uint32_t t = 0; t < wslot; t++) v &= flags[fl2 + t]; }
                else {
                    const uint32_t y = flags[wslot & 63];
                    v = fop == 0 ? (x & y) : fop == 1 ? (x | y) : fop == 2 ? (x & (1u - y)) : fop == 3 ? (x ^ y) : (x | (1u - y));
                }
                flags[fl] = v;
            }
        } else if (kind == COOP_K_LOADW) {
            int32_t r[14];
            const uint64_t it = (wslot & 0x10000u) ? item + partner_step : item;
            fp w = fp_zero();
            if (active) w = ws_ld(ws, (int)(wslot & 0xFFFF), it < ws.stride ? it : item);
            coop_from_words(r, w);
#pragma unroll
            for (int j = 0; j < 14; j++) S[dst * COOP_SW + j] = r[j];
        } else if (kind == COOP_K_STOREW) {
            int32_t d[14];
#pragma unroll
            for (int j = 0; j < 14; j++) d[j] = S[ix[0] * COOP_SW + j];
            coop_canonical(d);
            if (writes) ws_st(ws, (int)wslot, item, coop_to_words(d));
        } else if (kind == COOP_K_RES) {
            if (writes) {
                const bool ok = flags[fl] != 0 && flags[fl2] == 0;
                if (res_mode == COOP_RES_ITEM) {
                    uint32_t st = status[item];
                    if (!ok) st |= MBLS_ST_PAIRING_FAILED;
                    status[item] = st;
                    results[item] = (st & COOP_REJECT) ? 0 : 1;
                } else {
                    const uint32_t st = status[0];         // the OR of every set's status bits (k_status_or)
                    results[0] = (ok && !(st & COOP_REJECT_BATCH)) ? 1 : 0;
                }
            }
        }
        __syncthreads();
#ifdef MBLS_COOP_PROFILE
        COOP_PROF_T(pt3);
        if (pt1 == pt0) pt1 = pt2 = pt3;
        if (blockIdx.x == 0 && lane == 0) {
            mbls_coop_prof[kind * 4 + 0] += 1; mbls_coop_prof[kind * 4 + 1] += pt1 - pt0; mbls_coop_prof[kind * 4 + 2] += pt2 - pt1; mbls_coop_prof[kind * 4 + 3] += pt3 - pt2;
        }
#endif
    }
#endif
}
#define k_coop k_coop_t<false, 64>
#define k_coop_pow k_coop_t<true, 64>
#define k_coop_x2 k_coop_t<false, 32>
#define k_coop_pow_x4 k_coop_t<true, 16>
