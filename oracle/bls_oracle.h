/* bls_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE). See bls_oracle.c for scope and pinning status.
 * "Decoded point" = canonical uncompressed bytes: G1 96 B (x||y, infinity 0x40||0..),
 * G2 192 B (x.c1||x.c0||y.c1||y.c0, infinity 0x40||0..). */
#ifndef BLS_ORACLE_H
#define BLS_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* error codes = AmclError variants used by the reference (src/amcl_utils.rs:55,71; src/keys.rs:143; src/aggregates.rs:31) */
#define ORC_OK 0
#define ORC_ERR_G1_SIZE 1   /* AmclError::InvalidG1Size */
#define ORC_ERR_G2_SIZE 2   /* AmclError::InvalidG2Size */
#define ORC_ERR_POINT 3     /* AmclError::InvalidPoint */
#define ORC_ERR_EMPTY 4     /* AmclError::AggregateEmptyPoints */
#define ORC_PK_COMPRESSED 0
#define ORC_PK_UNCOMPRESSED 1

void orc_init(void);
void orc_fp_mul(const uint8_t a[48], const uint8_t b[48], uint8_t out[48]);
void orc_fp_inv(const uint8_t a[48], uint8_t out[48]);
int orc_fp_sqrt(const uint8_t a[48], uint8_t out[48]);
void orc_op_counts(uint64_t *mul, uint64_t *sqr, int reset);

int orc_g1_from_compressed(const uint8_t *in, size_t len, uint8_t out[96]);
int orc_g1_from_uncompressed(const uint8_t *in, size_t len, uint8_t out[96]);
int orc_g1_key_validate(const uint8_t pk[96]);
int orc_pk_from_bytes(const uint8_t *in, size_t len, uint8_t out[96]);
int orc_g1_compress(const uint8_t pk[96], uint8_t out[48]);
int orc_g2_from_compressed(const uint8_t *in, size_t len, uint8_t out[192]);
int orc_g2_compress(const uint8_t sig[192], uint8_t out[96]);
int orc_g2_subgroup_check(const uint8_t sig[192]);

int orc_g1_add(const uint8_t a[96], const uint8_t b[96], uint8_t out[96]);
int orc_g2_add(const uint8_t a[192], const uint8_t b[192], uint8_t out[192]);
int orc_g1_mul(const uint8_t a[96], const uint8_t k32[32], uint8_t out[96]);
int orc_g2_mul(const uint8_t a[192], const uint8_t k32[32], uint8_t out[192]);
int orc_aggregate_pks(const uint8_t *pks, size_t n, uint8_t out[96]);
void orc_sk_to_pk(const uint8_t sk[32], uint8_t out[96]);
void orc_hash_to_g2(const uint8_t *msg, size_t mlen, const uint8_t *dst, size_t dlen, uint8_t out[192]);
void orc_sign(const uint8_t *msg, size_t mlen, const uint8_t sk[32], uint8_t out[192]);

int orc_verify(const uint8_t sig[192], const uint8_t *msg, size_t mlen, const uint8_t pk[96]);
int orc_fast_aggregate_verify_pre_aggregated(const uint8_t sig[192], const uint8_t *msg, size_t mlen, const uint8_t apk[96]);
int orc_fast_aggregate_verify(const uint8_t sig[192], const uint8_t *msg, size_t mlen, const uint8_t *pks, size_t n);
int orc_aggregate_verify(const uint8_t sig[192], const uint8_t *msgs, const size_t *lens, size_t n_msgs, const uint8_t *pks, size_t n_pks);
int orc_verify_multiple(const uint8_t *sigs, const uint8_t *apks, const uint8_t *msgs, const size_t *lens, const uint64_t *rands, size_t n);

void orc_batch_fast_aggregate_verify(const uint8_t *sigs, const uint8_t *msgs, size_t msg_len, const uint8_t *pks, int pk_fmt,
                                     size_t n, size_t k, uint8_t *out, int nthreads);
void orc_batch_verify(const uint8_t *sigs, const uint8_t *msgs, size_t msg_len, const uint8_t *pks, size_t n, uint8_t *out, int nthreads);
void orc_batch_sign(const uint8_t *sks, const uint8_t *msgs, size_t msg_len, size_t n, uint8_t *sigs_out, int nthreads);
void orc_batch_sk_to_pk(const uint8_t *sks, size_t n, int pk_fmt, uint8_t *pks_out, int nthreads);
void orc_batch_hash_to_g2(const uint8_t *msgs, size_t msg_len, size_t n, uint8_t *out96);
#ifdef __cplusplus
}
#endif
#endif
