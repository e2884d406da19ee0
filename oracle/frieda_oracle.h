/*
 * frieda_oracle.h — CPU restatement of frieda's commit / generate_proof / verify path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the reported CPU baseline — never as a fallback for the HIP path.
 *
 * Parity status:
 *   PINNED   — codec, circle domain, twiddles, circle FFT, Merkle (raw Blake2s compression):
 *              reproduces the reference's golden root of src/commit.rs:31-37 on `blob`.
 *   UNPINNED — "parity unpinned": Fiat–Shamir channel, fold scaling, grind, queries, decommit
 *              and verifier follow stwo-prover@19d12d7 (Cargo.toml:12, Cargo.lock:896-898) as
 *              restated from its published algorithm; stwo's source is not vendored under
 *              /root/reference and no reference test holds a known answer for them.  They are
 *              anchored by the reference's own self-consistency tests (src/proof.rs:119-193,
 *              src/lib.rs:52-85) and the FRI degree invariant stwo asserts.
 */
#ifndef FRIEDA_ORACLE_H
#define FRIEDA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FO_P 2147483647u

/* status codes (mirror the product's; a reference `panic!` maps to FO_ERR_INVARIANT) */
#define FO_OK 0
#define FO_ERR_ARG 1
#define FO_ERR_INVARIANT 3

/* ---- field (stwo core/fields/{m31,cm31,qm31}.rs) ---- */
uint32_t fo_m31_add(uint32_t a, uint32_t b);
uint32_t fo_m31_sub(uint32_t a, uint32_t b);
uint32_t fo_m31_mul(uint32_t a, uint32_t b);
uint32_t fo_m31_inv(uint32_t a);
void fo_qm31_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]);

/* ---- codec (src/utils.rs:10-33) ---- */
/* number of felts bytes_to_felt_le yields for `len` bytes */
size_t fo_felt_count(size_t len);
/* src/utils.rs:10-19; out must hold fo_felt_count(len) words */
void fo_bytes_to_felt_le(const uint8_t* data, size_t len, uint32_t* out);
/* src/utils.rs:21-33; returns padded length F' (power of two >= 4) via the reference's f64 rule */
size_t fo_padded_len(size_t n_felts);
/* coef must hold fo_padded_len(fo_felt_count(len)) words; *log_size = log2(F') - 2 */
void fo_polynomial_from_bytes(const uint8_t* data, size_t len, uint32_t* coef, uint32_t* log_size);

/* ---- circle group / domain (stwo core/circle.rs, core/poly/circle/domain.rs) ---- */
void fo_point_from_index(uint32_t index, uint32_t* x, uint32_t* y);
/* CircleDomain::new(Coset::half_odds(n-1)).at(i) */
void fo_circle_domain_at(uint32_t n, uint32_t i, uint32_t* x, uint32_t* y);
uint32_t fo_bit_reverse_index(uint32_t i, uint32_t log_size);
/* ColumnOps::bit_reverse_column of CpuBackend (stwo core/utils.rs bit_reverse), in place on 2^log_size words */
void fo_bit_reverse_column(uint32_t* v, uint32_t log_size);

/* ---- twiddles (stwo backend/cpu/circle.rs::precompute_twiddles on Coset::half_odds(n-1)) ----
 * tw and itw each hold 2^(n-1) words: levels of size N/4, N/8, ..., 1, then the pad value 1. */
void fo_precompute_twiddles(uint32_t n, uint32_t* tw, uint32_t* itw);

/* ---- circle FFT (stwo backend/cpu/circle.rs::evaluate) ----
 * coef: 2^L coefficients; out: 2^n evaluations in bit-reversed domain order. */
void fo_circle_evaluate(const uint32_t* coef, uint32_t L, uint32_t n, const uint32_t* tw, uint32_t* out);

/* ---- reconstruction side (stwo backend/cpu/circle.rs::interpolate, core/fft.rs::ibutterfly) ----
 * block: the 2^L evaluations out[k * 2^L .. (k+1) * 2^L) of the bit-reversed codeword (any k < 2^(n-L)); coef_out: the 2^L
 * coefficients.  k = 0 with L == n is exactly CpuBackend::interpolate on the canonic domain. */
void fo_circle_interpolate_block(const uint32_t* block, uint32_t L, uint32_t n, uint32_t k, const uint32_t* itw, uint32_t* coef_out);
/* reconstruction from R = 2^(L-m) scattered cells of one column: cells[r][2^m] = entries cell_index[r] * 2^m .. of the
 * bit-reversed codeword (m >= 1, distinct cells); tw / itw from fo_precompute_twiddles(n); returns 0 or -1 (bad arguments) */
int fo_reconstruct_cells(const uint32_t* cells, const uint32_t* cell_index, uint32_t R, uint32_t m, uint32_t L, uint32_t n,
                         const uint32_t* tw, const uint32_t* itw, uint32_t* coef_out);
/* inverse of fo_bytes_to_felt_le: felts (each < 2^30) -> the first `len` bytes of the LSB-first bit stream */
/* one column from any >= 2^L + 2 distinct sampled points (vals[i] at bit-reversed position pos[i]); erasure-locator route */
int fo_reconstruct_points(const uint32_t* vals, const uint32_t* pos, uint32_t n_pts, uint32_t L, uint32_t n, uint32_t* coef_out);
void fo_felts_to_bytes(const uint32_t* felts, size_t n_felts, uint8_t* out, size_t len);

/* ---- Merkle (stwo core/vcs/blake2_merkle.rs::hash_node, backend/cpu/blake2s.rs::commit_on_layer) ---- */
void fo_blake2s_compress(const uint32_t h[8], const uint32_t m[16], uint32_t t0, uint32_t t1, uint32_t f0,
                         uint32_t f1, uint32_t out[8]);
/* one layer: out[i] = hash_node(prev ? (prev[2i], prev[2i+1]) : none, [col[i] for col in cols]) */
void fo_merkle_commit_layer(uint32_t log_size, const uint8_t* prev, const uint32_t* const* cols, uint32_t ncols,
                            uint8_t* out);
/* full tree over equal-length columns of 2^log_size; layers_out must hold 32*(2^(log_size+1)-1) bytes laid
 * out leaves first: layer log_size at offset 0, then layer log_size-1, ..., the root last. */
void fo_merkle_commit(const uint32_t* const* cols, uint32_t ncols, uint32_t log_size, uint8_t* layers_out);
/* byte offset of layer `layer_log` inside layers_out for a tree of `log_size` */
size_t fo_merkle_layer_offset(uint32_t log_size, uint32_t layer_log);

/* ---- folds (stwo backend/cpu/fri.rs) ---- columns are SoA QM31 coordinates */
void fo_fold_circle_into_line(uint32_t* const dst[4], const uint32_t* const src[4], uint32_t n, const uint32_t alpha[4]);
/* src on LineDomain(coset) with coset = half_odds(n-1) doubled `n_doublings` times; log size m */
void fo_fold_line(const uint32_t* const src[4], uint32_t line_log_size, uint32_t domain_n, const uint32_t alpha[4],
                  uint32_t* const dst[4]);

/* ---- trait methods frieda's path never calls (stwo backend/cpu/circle.rs::{extend, eval_at_point}, backend/cpu/fri.rs::decompose;
 * restated from the published code of stwo-prover@19d12d7, parity unpinned) ---- */
void fo_circle_extend(const uint32_t* coef, uint32_t log_coef, uint32_t log_size, uint32_t* out);
/* the polynomial of 2^log_coef coefficients at the circle point (px, py) over QM31 */
void fo_circle_eval_at_point(const uint32_t* coef, uint32_t log_coef, const uint32_t px[4], const uint32_t py[4], uint32_t out[4]);
/* eval, g: SoA QM31 columns of 2^log_size entries (bit-reversed order); lambda_out: the decomposition coefficient */
void fo_fri_decompose(const uint32_t* const eval[4], uint32_t log_size, uint32_t* const g[4], uint32_t lambda_out[4]);

/* ---- Fiat–Shamir channel (stwo core/channel/blake2s.rs) ---- */
typedef struct {
    uint8_t digest[32];
    uint64_t n_challenges;
    uint64_t n_sent;
} fo_channel;
void fo_blake2s256(const uint8_t* in, size_t len, uint8_t out[32]);
void fo_channel_init(fo_channel* c);
void fo_channel_mix_u64(fo_channel* c, uint64_t v);
void fo_channel_mix_root(fo_channel* c, const uint8_t root[32]);
void fo_channel_mix_felts(fo_channel* c, const uint32_t* qm31s, size_t n_qm31);
void fo_channel_draw_random_bytes(fo_channel* c, uint8_t out[32]);
void fo_channel_draw_felt(fo_channel* c, uint32_t out[4]);
/* test hook: acceptance bound of draw_base_felts (0 restores 2P); must stay <= 2P */
void fo_test_set_draw_bound(uint32_t bound);
uint32_t fo_channel_trailing_zeros(const fo_channel* c);
uint64_t fo_grind(const fo_channel* c, uint32_t pow_bits);
/* Queries::generate; out must hold n_queries entries; returns the deduplicated count */
size_t fo_queries_generate(fo_channel* c, uint32_t log_domain_size, size_t n_queries, uint32_t* out);

/* ---- API (src/lib.rs:31-43) ---- */
typedef struct {
    uint32_t pow_bits;
    uint32_t log_blowup_factor;
    uint32_t log_last_layer_degree_bound;
    uint32_t n_queries;
} fo_pcs_config;

typedef struct {
    uint32_t* fri_witness; /* QM31s, 4 words each */
    size_t n_fri_witness;
    uint8_t* hash_witness; /* 32 B each */
    size_t n_hash_witness;
    uint32_t* column_witness;
    size_t n_column_witness;
    uint8_t commitment[32];
} fo_layer_proof;

typedef struct {
    fo_layer_proof first_layer;
    fo_layer_proof* inner_layers;
    size_t n_inner_layers;
    uint32_t* last_layer_poly; /* QM31s */
    size_t n_last_layer_poly;
    uint64_t proof_of_work;
    fo_pcs_config pcs_config;
    uint32_t log_size_bound;
    uint32_t* evaluations; /* QM31s */
    size_t n_evaluations;
} fo_proof;

int fo_commit(const uint8_t* data, size_t len, uint32_t log_blowup_factor, uint8_t root[32]);
/* seed may be NULL (Option<u64>::None) */
int fo_commit_and_generate_proof(const uint8_t* data, size_t len, const uint64_t* seed, fo_pcs_config cfg,
                                 uint8_t commitment[32], fo_proof** out);
/* *ok = verify result; returns FO_ERR_INVARIANT where the reference would panic */
int fo_verify(const fo_proof* proof, const uint64_t* seed, int* ok);
void fo_proof_free(fo_proof* p);
fo_proof* fo_proof_clone(const fo_proof* p);

/* canonical little-endian wire image of a proof (layout: DESIGN.md §6); returns bytes written, or the
 * required size when buf == NULL */
size_t fo_proof_serialize(const fo_proof* p, uint8_t* buf, size_t cap);

/* transcript trace of the last fo_commit_and_generate_proof call on this thread (test aid):
 * alphas (QM31 each), layer roots, queries */
typedef struct {
    uint32_t n_layers; /* 1 + n_inner */
    uint32_t alphas[64][4];
    uint8_t roots[64][32];
    uint8_t digest_before_grind[32];
} fo_trace;
const fo_trace* fo_last_trace(void);

#ifdef __cplusplus
}
#endif
#endif
