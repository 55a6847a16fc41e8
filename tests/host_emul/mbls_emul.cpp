// mbls_emul.cpp -- TEST INFRASTRUCTURE: compiles the lane bodies of the HIP kernels
// (milagro_bls_amd/csrc/mbls_*.h) as plain C++ and runs them one "lane" at a time on the CPU, so the
// CPU-only test suite can check every phase against the oracle before a GPU is involved. It is not a
// fallback: the product library (libmbls_hip.so) contains no CPU path and never loads this file.
#define MBLS_HOST_EMUL 1
#include <stdlib.h>
#include <string.h>
#include "../../milagro_bls_amd/csrc/mbls_ops.h"

static uint64_t emul_phase_ops[5];
extern "C" {
void emul_op_counts(uint64_t* mul, uint64_t* sqr, int reset) { *mul = mbls_cnt_mul; *sqr = mbls_cnt_sqr; if (reset) mbls_cnt_mul = mbls_cnt_sqr = 0; }
// the full fast_aggregate_verify / verify pipeline over n items, same phases and workspace layout as the GPU
void emul_verify_batch(const uint8_t* sigs, const uint8_t* msgs, uint32_t mlen, const uint8_t* pks, int fmt,
                       const uint32_t* offsets, uint64_t n, uint32_t k, int mode, uint8_t* results, uint32_t* status) {
    mbls_ws ws; ws.stride = n ? n : 1; ws.w = (uint32_t*)calloc((size_t)MBLS_SLOT_COUNT * 12 * ws.stride, 4);
    const uint32_t pkb = fmt == MBLS_PK_COMPRESSED ? 48 : 96;
    if (fmt == MBLS_PK_COMPRESSED && !offsets && k > 1) {   // staged path: lane per key, then per-item sums (as on the GPU)
        uint32_t* xy = (uint32_t*)malloc((size_t)n * k * 96); uint8_t* fl = (uint8_t*)malloc((size_t)n * k);
        for (uint64_t j = 0; j < n * k; j++) lane_pk_decompress(j, pks, xy, fl);
        for (uint64_t i = 0; i < n; i++) lane_aggregate_decoded(ws, i, xy + 24 * (uint64_t)k * i, fl + (uint64_t)k * i, k, mode, &status[i]);
        free(xy); free(fl);
    } else
    for (uint64_t i = 0; i < n; i++) {
        uint64_t first = offsets ? offsets[i] : (uint64_t)k * i; uint32_t cnt = offsets ? offsets[i + 1] - offsets[i] : k;
        lane_aggregate(ws, i, pks + pkb * first, cnt, fmt, mode, &status[i]);
    }
    emul_phase_ops[0] = mbls_cnt_mul + mbls_cnt_sqr;
    for (uint64_t i = 0; i < n; i++) lane_sig(ws, i, sigs + 96 * i, &status[i]);
    emul_phase_ops[1] = mbls_cnt_mul + mbls_cnt_sqr;
    for (uint64_t i = 0; i < n; i++) lane_hash(ws, i, msgs + (uint64_t)mlen * i, mlen);
    emul_phase_ops[2] = mbls_cnt_mul + mbls_cnt_sqr;
    for (uint64_t i = 0; i < n; i++) lane_miller(ws, i);
    emul_phase_ops[3] = mbls_cnt_mul + mbls_cnt_sqr;
    for (uint64_t i = 0; i < n; i++) lane_final(ws, i, &status[i], &results[i]);
    emul_phase_ops[4] = mbls_cnt_mul + mbls_cnt_sqr;
    free(ws.w);
}
void emul_phase_counts(uint64_t out[5]) { for (int i = 0; i < 5; i++) out[i] = emul_phase_ops[i]; }
void emul_g1_decode(const uint8_t* in, int fmt, int validate, uint64_t n, uint8_t* out96, uint8_t* err) { for (uint64_t i = 0; i < n; i++) op_g1_decode(i, in, fmt, validate, out96, err); }
void emul_g1_key_validate(const uint8_t* in96, uint64_t n, uint8_t* ok) { for (uint64_t i = 0; i < n; i++) op_g1_key_validate(i, in96, ok); }
void emul_g1_compress(const uint8_t* in96, uint64_t n, uint8_t* out48, uint8_t* err) { for (uint64_t i = 0; i < n; i++) op_g1_compress(i, in96, out48, err); }
void emul_g2_check(const uint8_t* in96, uint64_t n, uint8_t* err, uint8_t* in_g2) { for (uint64_t i = 0; i < n; i++) op_g2_check(i, in96, err, in_g2); }
void emul_g2_add(const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out, uint8_t* err) { for (uint64_t i = 0; i < n; i++) op_g2_add(i, a, b, out, err); }
void emul_g1_add(const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out, uint8_t* err) { for (uint64_t i = 0; i < n; i++) op_g1_add(i, a, b, out, err); }
void emul_sign(const uint8_t* sks, const uint8_t* msgs, uint32_t mlen, uint64_t n, uint8_t* out96) { for (uint64_t i = 0; i < n; i++) op_sign(i, sks, msgs, mlen, out96); }
void emul_sk_to_pk(const uint8_t* sks, int fmt, uint64_t n, uint8_t* out) { for (uint64_t i = 0; i < n; i++) op_sk_to_pk(i, sks, fmt, out); }
void emul_hash_to_g2(const uint8_t* msgs, uint32_t mlen, uint64_t n, uint8_t* out96) { for (uint64_t i = 0; i < n; i++) op_hash_to_g2(i, msgs, mlen, out96); }
// hash_to_field through the register-only function the hashing kernel uses (out: 4 x 12 words per item: u0.c0, u0.c1, u1.c0, u1.c1)
// and through expand_message_xmd_256 + fp_from_two_digests (out_ref)
void emul_hash_fields(const uint8_t* msgs, uint32_t mlen, uint64_t n, uint32_t* out, uint32_t* out_ref) {
    mbls_ws ws; ws.stride = n ? n : 1; ws.w = (uint32_t*)calloc((size_t)49 * 12 * ws.stride, 4);
    for (uint64_t i = 0; i < n; i++) {
        hash_fields_to_ws(ws.w, ws.stride, i, msgs + (uint64_t)mlen * i, mlen);
        const int slot[4] = {31, 32, 37, 38};
        for (int q = 0; q < 4; q++) { fp v = ws_ld(ws, slot[q], i); for (int j = 0; j < 12; j++) out[(4 * i + q) * 12 + j] = v[j]; }
        uint32_t ub[64];
        expand_message_xmd_256(ub, msgs + (uint64_t)mlen * i, mlen, MBLS_DST_POP, MBLS_DST_POP_LEN);
        for (int q = 0; q < 4; q++) { fp v = fp_from_two_digests(ub + 16 * q, ub + 16 * q + 8); for (int j = 0; j < 12; j++) out_ref[(4 * i + q) * 12 + j] = v[j]; }
    }
    free(ws.w);
}
void emul_fp_mul(const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out, int op) { for (uint64_t i = 0; i < n; i++) op_fp_mul(i, n, a, b, out, op); }
void emul_aggregate(const uint8_t* pks, int fmt, const uint32_t* offsets, uint64_t n, uint32_t k, uint8_t* out96, uint32_t* status) {
    mbls_ws ws; ws.stride = n ? n : 1; ws.w = (uint32_t*)calloc((size_t)MBLS_SLOT_COUNT * 12 * ws.stride, 4);
    const uint32_t pkb = fmt == MBLS_PK_COMPRESSED ? 48 : 96;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t first = offsets ? offsets[i] : (uint64_t)k * i; uint32_t cnt = offsets ? offsets[i + 1] - offsets[i] : k;
        lane_aggregate(ws, i, pks + pkb * first, cnt, fmt, MBLS_MODE_FAST_AGGREGATE, &status[i]);
        op_apk_export(ws, i, out96);
    }
    free(ws.w);
}
}
